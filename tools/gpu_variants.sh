# A/B of library variants built with tools/build_variant.sh: tools/gpu_variants.sh hip NAME1 NAME2 ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-real-scene $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1 $2]', '%.2f ms'%d['ms_per_step'], 'fused %.2f'%d['stages_ms']['grad'])"; }
for k in "$@"; do for p in "" "--variant manifold_caustic --profile pool"; do EPSM_LIB_NAME=libepsm_$k.so run "$k" "$p"; done; done
