# small wavefronts (configs 1 and 5, the reference's own sizes) per library variant: tools/gpu_small.sh NAME...
# EPSM_SMALL_WAVEFRONT / EPSM_NO_REPLICAS from the caller's environment pick the window form and the flush target.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1 $2 small=$EPSM_SMALL_WAVEFRONT norep=$EPSM_NO_REPLICAS]', 'kernel %.4f ms'%d['stages_ms']['grad'], 'frac %.3f'%d['roofline']['frac'])"; }
for k in "$@"; do for p in "--config 5" "--config 1"; do EPSM_LIB_NAME=libepsm_$k.so run "$k" "$p"; done; done
