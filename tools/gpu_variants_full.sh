# A/B of library variants over the bench profiles, two repetitions: tools/gpu_variants_full.sh hip NAME1 ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-real-scene $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.2f' % d['stages_ms']['grad'], end=' ')"; }
for rep in 1 2; do for k in "$@"; do echo -n "[$k] "; for p in "" "--variant manifold_caustic --profile pool" "--profile specular" "--scene-vertices 7829" "--scene-vertices 1000000" "--config 5"; do EPSM_LIB_NAME=libepsm_$k.so run "$k" "$p"; done; echo; done; done
