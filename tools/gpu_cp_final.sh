# parity of the product library (per-path, pipeline, full size, tracer logs), then its timings on the four workloads: tools/gpu_cp_final.sh OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1.txt; : > $OUT
timeout -k 10 800 python -m pytest tests/test_gpu_backward_per_path.py tests/test_gpu_pipeline.py tests/test_gpu_full_size_packed.py tests/test_gpu_tracer.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 >> $OUT || { cat $OUT; exit 1; }
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1]', 'kernel %.4f ms'%d['stages_ms']['grad'], 'frac %.3f'%d['roofline']['frac'])" >> $OUT; }
for rep in 1 2; do
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 45 2>/dev/null | tail -1 | line "headline"
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --config 2 2>/dev/null | tail -1 | line "config2"
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --config 3 --max-resident-gb 30 2>/dev/null | tail -1 | line "pool"
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --config 5 2>/dev/null | tail -1 | line "config5"
done
cat $OUT
