cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_gpu_tracer_oracle.py tests/test_gpu_parity.py -x -q -s > gpurun_out/r2g_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2g_pytest.log; tail -30 gpurun_out/r2g_pytest.log | cut -c1-1800
