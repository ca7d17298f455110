cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
prof=${1:-bathroom}
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-scene --profile $prof"
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --kernel-include-regex "epsm_grad" --output-format csv -d gpurun_out/r1r_pmc_f1_$prof -- $B > gpurun_out/r1r_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --kernel-include-regex "epsm_grad" --output-format csv -d gpurun_out/r1r_pmc_f2_$prof -- $B > gpurun_out/r1r_pmc2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "epsm_grad" --output-format csv -d gpurun_out/r1r_pmc_f3_$prof -- $B > gpurun_out/r1r_pmc3.log 2>&1
tail -1 gpurun_out/r1r_pmc3.log
