# The two tracer forms at small launches, full and gradient-only trace (clutter, 128 k triangles, 256 x 256 @ spp):
# tools/gpu_small_tracers.sh  -> stdout
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for spp in 8 16 32 64; do for t in wavefront mega; do
  echo "== 256x256 @ $spp spp = $((65536 * spp)) paths, tracer $t"
  EPSM_PROF_RES=256 EPSM_PROF_SPP=$spp EPSM_PROF_TRACER=$t timeout -k 10 120 python tools/prof_gradient_only.py manifold 2>&1 | grep "^gradient_only" | cut -c1-75
done; done
