# small wavefronts in one launch vs two: parity of the small-wavefront forms, then config 5 / config 1 timings with the option on and off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1.txt; : > $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_backward_per_path.py -x -q -m gpu 2>&1 | tail -3 >> $OUT || { cat $OUT; exit 1; }
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1]', 'kernel %.4f ms'%d['stages_ms']['grad'], 'step %.4f ms'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'])" >> $OUT; }
for rep in 1 2; do for c in 5 1; do
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --config $c 2>/dev/null | tail -1 | line "one launch, config $c"
  EPSM_TWO_LAUNCHES=1 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --config $c 2>/dev/null | tail -1 | line "two launches, config $c"
done; done
cat $OUT
