cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_tangent_scatter.py -x -q > gpurun_out/r2e_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2e_pytest.log; tail -3 gpurun_out/r2e_pytest.log
timeout -k 10 1100 bash tools/gpu_variants.sh hip nt > gpurun_out/r2e_variants.log 2>&1; cat gpurun_out/r2e_variants.log
