# small wavefronts (config 5: 524 288 paths, K = 2; config 1: 16 384 paths, K = 4) per variant: tools/gpu_cp_small2.sh NAME...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in "$@"; do for c in 5 1; do
  EPSM_LIB_NAME=libepsm_$k.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --config $c 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$k --config $c]', 'kernel %.4f ms'%d['stages_ms']['grad'], 'step %.4f ms'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'])"; done; done
