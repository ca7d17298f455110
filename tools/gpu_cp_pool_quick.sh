# pool-caustic slab (BASELINE.json configs[2]) only, per variant: tools/gpu_cp_pool_quick.sh NAME1 NAME2 ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in "$@"; do
  EPSM_LIB_NAME=libepsm_$k.so python bench.py --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 30 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$k pool]', 'kernel %.3f ms'%d['stages_ms']['grad'], 'frac %.3f'%d['roofline']['frac'])"; done
