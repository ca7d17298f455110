# one-launch tracer at small launches, builds NAME...: gradient-only and full trace at 2^19 / 2^20 paths
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "$@"; do for spp in 8 16; do
  echo "== build $v, 256x256 @ $spp spp, one-launch tracer"
  EPSM_LIB_NAME=libepsm_$v.so EPSM_PROF_RES=256 EPSM_PROF_SPP=$spp EPSM_PROF_TRACER=mega timeout -k 10 120 python tools/prof_gradient_only.py manifold 2>&1 | grep "^gradient_only" | cut -c1-75
done; done
