# A/B: fused vs two-stage on a few profiles
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for extra in "" "--two-stage" "--profile specular" "--profile specular --two-stage" "--variant manifold_caustic --profile pool" "--variant manifold_caustic --profile pool --two-stage"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-real-scene $extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$extra]', '%.3e paths/s'%d['value'], '%.2f ms'%d['ms_per_step'], {k: round(v,3) for k,v in d['stages_ms'].items()})"
done
