cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 1100 bash tools/gpu_variants.sh hip > gpurun_out/r2v_variants.log 2>&1; cat gpurun_out/r2v_variants.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
