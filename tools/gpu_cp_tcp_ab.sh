# TCP counters of the backward kernel for A/B library builds (tools/build_cp_variant.sh NAME ...): tools/gpu_cp_tcp_ab.sh OUT NAME1 NAME2 ...
# ("hip" = the product build).  Two --pmc passes per build (requests + latency; stalls), 48 dispatches each.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; shift; mkdir -p $T
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 25"
for k in "$@"; do
  export EPSM_LIB_NAME=libepsm_$k.so
  echo "## build $k" >> $T/summary.txt
  timeout -k 5 200 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum --kernel-include-regex "epsm_backward" --output-format csv -d $T/${k}_d -- $B > $T/${k}_d.log 2>&1; echo "$k pass d rc $?" >> $T/passes.txt
  python tools/summarize_rocprof.py $T/${k}_d >> $T/summary.txt 2>&1
  timeout -k 5 200 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-include-regex "epsm_backward" --output-format csv -d $T/${k}_e -- $B > $T/${k}_e.log 2>&1; echo "$k pass e rc $?" >> $T/passes.txt
  python tools/summarize_rocprof.py $T/${k}_e >> $T/summary.txt 2>&1
  timeout -k 5 200 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-include-regex "epsm_backward" --output-format csv -d $T/${k}_f -- $B > $T/${k}_f.log 2>&1; echo "$k pass f rc $?" >> $T/passes.txt
  python tools/summarize_rocprof.py $T/${k}_f >> $T/summary.txt 2>&1
done
cat $T/passes.txt; grep -v "^$" $T/summary.txt | grep -v "vgpr=" | cut -c1-150
