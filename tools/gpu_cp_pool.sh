# pool slab (config 3, manifold_caustic) per variant: tools/gpu_cp_pool.sh NAME...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in "$@"; do
  EPSM_LIB_NAME=libepsm_$k.so python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --config 3 --max-resident-gb 30 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$k pool]', 'kernel %.3f ms'%d['stages_ms']['grad'], 'frac %.3f'%d['roofline']['frac'], 'live frac', d['roofline']['frac_live'])"; done
