# A/B of library variants built with tools/build_variant.sh: tools/gpu_variants.sh NAME1 NAME2 ...  ("hip" = the product build)
# Per variant: the headline slab (1024x1024 @ 256 spp, 3 resident slabs), config 2 (512x512 @ 64 spp) and config 3's pool profile
# (native packed log; EPSM_AB_SOA=1 adds the reference's tensor layout).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1 $2]', 'kernel %.3f ms'%d['stages_ms']['grad'], 'frac %.3f'%d['roofline']['frac'], 'dense kernel %.3f ms'%d.get('standalone_grad_kernel',{}).get('kernel_ms',0))"; }
for k in "$@"; do for p in "--max-resident-gb 45" "--config 2" "--config 3 --max-resident-gb 30"; do
  EPSM_LIB_NAME=libepsm_$k.so run "$k" "$p"; if [ -n "$EPSM_AB_SOA" ]; then EPSM_LIB_NAME=libepsm_$k.so run "$k" "$p --soa"; fi; done; done
