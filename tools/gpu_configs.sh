# all BASELINE.json configs through bench.py (1 GPU): profiles/rNN_configs.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for c in 1 2 3 4 5; do
  st=10; [ $c -ge 3 ] && [ $c -le 4 ] && st=3
  python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('config $c:', d['config']['workload'])
print('   %.3e paths/s  %.3f ms/step  stages(last slab) %s  roofline.frac %.3f' % (d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['stages_ms'].items()}, d['roofline']['frac']))"
done
