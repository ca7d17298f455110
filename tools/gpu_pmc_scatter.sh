cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for prof in bathroom specular; do
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-scene --profile $prof"
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --kernel-include-regex "epsm" --output-format csv -d gpurun_out/r1j_pmc_f1_$prof -- $B > gpurun_out/r1j_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --kernel-include-regex "epsm" --output-format csv -d gpurun_out/r1j_pmc_f2_$prof -- $B > gpurun_out/r1j_pmc2.log 2>&1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-scene --profile $prof --two-stage"
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --kernel-include-regex "epsm" --output-format csv -d gpurun_out/r1j_pmc_t1_$prof -- $B > gpurun_out/r1j_pmc3.log 2>&1
done
tail -2 gpurun_out/r1j_pmc3.log
