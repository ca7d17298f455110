# small wavefronts (config 1, config 5) of cp-kernel variants: tools/gpu_cp_small.sh NAME...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1 $2]', 'kernel %.4f ms'%d['stages_ms']['grad'], 'step %.4f ms'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'])"; }
for k in "$@"; do for p in "--config 5" "--config 1"; do EPSM_LIB_NAME=libepsm_$k.so run "$k" "$p"; done; done
