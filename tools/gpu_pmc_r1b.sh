cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r1b_pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex epsm --output-format csv -d gpurun_out/r1b_pmc_fetch -- $B > gpurun_out/r1b_pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex epsm --output-format csv -d gpurun_out/r1b_pmc_write -- $B > gpurun_out/r1b_pmc2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-include-regex epsm --output-format csv -d gpurun_out/r1b_pmc_sq1 -- $B > gpurun_out/r1b_pmc3.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-include-regex epsm --output-format csv -d gpurun_out/r1b_pmc_sq2 -- $B > gpurun_out/r1b_pmc4.log 2>&1
for p in specular caustic; do python bench.py --steps 5 --warmup 1 --no-cpu-baseline --profile $p > gpurun_out/r1b_bench_$p.log 2>&1; done
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --variant manifold_caustic --profile pool > gpurun_out/r1b_bench_caustic_pool.log 2>&1
cat gpurun_out/r1b_pytest.log; tail -2 gpurun_out/r1b_pmc4.log
