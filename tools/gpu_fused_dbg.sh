cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for dbg in 0 1 9; do
  export EPSM_FUSED_DBG=$dbg
  for extra in "" "--profile specular"; do
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline $extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('dbg $dbg $extra', {k: round(v,3) for k,v in d['stages_ms'].items()})"
  done
done
