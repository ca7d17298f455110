cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-wfpmc}
B="python tools/prof_trace.py wavefront 100 4194304 2"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-include-regex "epsm_wf" --output-format csv -d gpurun_out/${TAG}_1 -- $B > gpurun_out/${TAG}_1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INST_LEVEL_VMEM --kernel-include-regex "epsm_wf" --output-format csv -d gpurun_out/${TAG}_2 -- $B > gpurun_out/${TAG}_2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --kernel-include-regex "epsm_wf" --output-format csv -d gpurun_out/${TAG}_3 -- $B > gpurun_out/${TAG}_3.log 2>&1
for i in 1 2 3; do python tools/summarize_rocprof.py gpurun_out/${TAG}_$i > gpurun_out/${TAG}_$i.txt; tail -2 gpurun_out/${TAG}_$i.log; done
cat gpurun_out/${TAG}_1.txt gpurun_out/${TAG}_2.txt gpurun_out/${TAG}_3.txt
