# tracer timings (tools/bench_bigscene.py 100 spheres = 128 k triangles, wavefront form) + the real_scene leg of the bench line,
# after the tracer's GPU tests: tools/gpu_trace_ab.sh OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1.txt; : > $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_tracer.py tests/test_gpu_tracer_oracle.py tests/test_gpu_radiometry.py tests/test_gpu_environment.py -x -q -m gpu 2>&1 | tail -2 >> $OUT || { cat $OUT; exit 1; }
timeout -k 10 300 python tools/bench_bigscene.py 100 2>&1 | grep -E "wavefront|==" >> $OUT
timeout -k 10 300 python tools/bench_real.py 1024 16 2>&1 | tail -3 >> $OUT
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --max-resident-gb 25 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('real_scene', {k: d['real_scene'][k] for k in ('grad_image_ms','trace_and_log_ms','backward_ms')})" >> $OUT
cat $OUT
