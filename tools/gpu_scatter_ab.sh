cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for mode in 0 1 3; do for bits in 11 12; do
  export EPSM_SCATTER_MODE=$mode EPSM_SCATTER_BITS=$bits
  for extra in "" "--profile specular"; do
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline $extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('mode $mode bits $bits $extra', {k: round(v,3) for k,v in d['stages_ms'].items()})"
  done
done; done
