# TCP miss-queue counters of the wavefront tracer's stages: tools/gpu_trace_tcp.sh OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T; : > $T/summary.txt
pass() { n=$1; shift; timeout -k 5 200 rocprofv3 --pmc "$@" --kernel-include-regex "epsm_wf" --output-format csv -d $T/$n -- python tools/prof_trace.py wavefront 100 > $T/$n.log 2>&1; echo "pass $n rc $?" >> $T/summary.txt; python tools/summarize_rocprof.py $T/$n >> $T/summary.txt 2>&1; }
pass a TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass b TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES
grep -v "^$" $T/summary.txt | grep -v "vgpr=" | cut -c1-150
