# wavefront tracer: GPU tests + timings against the one-launch form
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-wf}
python -m pytest tests/test_gpu_tracer.py -x -q 2>&1 | tail -15 > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
python tools/bench_bigscene.py 100 > gpurun_out/${TAG}_big100.log 2>&1; tail -5 gpurun_out/${TAG}_big100.log
python tools/bench_bigscene.py 400 > gpurun_out/${TAG}_big400.log 2>&1; tail -4 gpurun_out/${TAG}_big400.log
python tools/bench_bigscene.py 100 4194304 > gpurun_out/${TAG}_big100_onetile.log 2>&1; tail -4 gpurun_out/${TAG}_big100_onetile.log
