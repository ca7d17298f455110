# knock-out / A-B builds of the backward kernel (tools/build_cp_variant.sh) on three slabs: headline, pool caustic, dense specular
# usage: tools/gpu_cp_ko.sh NAME1 NAME2 ...   ("hip" = the product build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { EPSM_LIB_NAME=libepsm_$1.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary $3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1 $2]', 'kernel %.3f ms'%d['stages_ms']['grad'], 'frac %.3f'%d['roofline']['frac'])"; }
for k in "$@"; do
  run $k headline "--max-resident-gb 30"
  run $k pool "--config 3 --max-resident-gb 30"
  run $k specular "--config 2 --profile specular"
done
