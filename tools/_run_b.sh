cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
EPSM_LIB_NAME=libepsm_recd.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r2f_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2f_pytest.log; tail -3 gpurun_out/r2f_pytest.log
timeout -k 10 1100 bash tools/gpu_variants.sh hip recd > gpurun_out/r2f_variants.log 2>&1; cat gpurun_out/r2f_variants.log
