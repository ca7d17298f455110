# More SQ / TA / TCP counters of the backward kernel, one short pass each (a pass that hangs is cut at 150 s and the rest still run):
# tools/gpu_cp_sq2.sh OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 25"
pass() { n=$1; shift; timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-include-regex "epsm_backward" --output-format csv -d $T/$n -- $B > $T/$n.log 2>&1; echo "pass $n rc $?" >> $T/passes.txt; python tools/summarize_rocprof.py $T/$n >> $T/summary.txt 2>&1; }
: > $T/summary.txt; : > $T/passes.txt
pass a SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES
pass b SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM_RD
# (pass c, TA_* counters: hangs rocprofv3 on this pool -- removed)
pass d TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
pass e TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
cat $T/passes.txt; grep -v "^$" $T/summary.txt | grep -v "vgpr=" | cut -c1-140
