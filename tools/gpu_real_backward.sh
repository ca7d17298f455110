# The backward kernel on a traced tile: duration and HBM traffic (tools/prof_real_backward.py): tools/gpu_real_backward.sh OUT [variant]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; v=${2:-manifold}; mkdir -p $T
python3 tools/prof_real_backward.py $v
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $T/trace -- python3 tools/prof_real_backward.py $v > $T/trace.log 2>&1
timeout -k 5 200 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-include-regex "epsm_backward" --output-format csv -d $T/pmc1 -- python3 tools/prof_real_backward.py $v > $T/pmc1.log 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --kernel-include-regex "epsm_backward" --output-format csv -d $T/pmc2 -- python3 tools/prof_real_backward.py $v > $T/pmc2.log 2>&1
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU --kernel-include-regex "epsm_backward" --output-format csv -d $T/pmc3 -- python3 tools/prof_real_backward.py $v > $T/pmc3.log 2>&1
python3 tools/summarize_rocprof.py $T | grep -v "^$" | grep "epsm_backward\|TCC\|WRITE\|SQ_\|calls" | cut -c1-150
