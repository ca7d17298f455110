# A/B of cp-kernel variants built with tools/build_cp_variant.sh: tools/gpu_cp_variants.sh NAME1 NAME2 ...  ("hip" = product build)
# Per variant: headline slab (1024x1024 @ 256 spp), config 2, config 3 (pool caustic), config 5 (human-size small wavefront).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1 $2]', 'kernel %.3f ms'%d['stages_ms']['grad'], 'frac %.3f'%d['roofline']['frac'])"; }
for k in "$@"; do for p in "--max-resident-gb 45" "--config 2" "--config 3 --max-resident-gb 30" "--config 5"; do
  EPSM_LIB_NAME=libepsm_$k.so run "$k" "$p"; done; done
