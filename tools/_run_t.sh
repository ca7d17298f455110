cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 600 python tools/bench_packed_real.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2t_real.log; cat gpurun_out/r2t_real.log
timeout -k 10 600 python -m pytest tests/test_gpu_tracer.py tests/test_gpu_tracer_oracle.py tests/test_gpu_optim.py -x -q 2>&1 | tail -3
