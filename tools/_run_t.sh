cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 1100 bash tools/gpu_variants.sh hip nopin 2>&1 | tee gpurun_out/r3a_variants.log
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_tangent_scatter.py tests/test_gpu_tracer.py tests/test_gpu_optim.py -x -q 2>&1 | tail -3
