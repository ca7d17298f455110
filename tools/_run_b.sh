cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_color_adjoint.py tests/test_gpu_optim.py -x -q > gpurun_out/r2p_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2p_pytest.log; tail -25 gpurun_out/r2p_pytest.log | cut -c1-400
