# Counters of the headline slab under the two placements of the native log (tools/gpu_layout_ab.sh): tools/gpu_layout_counters.sh OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/${1:-layout}; mkdir -p $T
for k in dense interleaved; do
  B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 30 --log-layout $k"
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-include-regex "epsm_backward" --output-format csv -d $T/${k}_rdreq -- $B > $T/${k}_1.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-include-regex "epsm_backward" --output-format csv -d $T/${k}_tcp -- $B > $T/${k}_2.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --kernel-include-regex "epsm_backward" --output-format csv -d $T/${k}_sq -- $B > $T/${k}_3.log 2>&1
  echo "## $k"; for d in rdreq tcp sq; do python3 tools/summarize_rocprof.py $T/${k}_$d | grep -v "^$" | cut -c1-160; done
done
