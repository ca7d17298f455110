# A/B of the wave-packet traversal per stage (tools/build_trace_variant.sh NAME "-D..."): tools/gpu_packet_ab.sh NAME...
# per build: tracer tests, then gradient-only and full trace of the clutter scene (2^24 paths), both variants
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "$@"; do
  echo "== build $v"
  EPSM_LIB_NAME=libepsm_$v.so timeout -k 10 400 python -m pytest tests/test_gpu_tracer.py tests/test_gpu_tracer_oracle.py tests/test_gpu_gradient_only.py -x -q 2>&1 | tail -1
  EPSM_LIB_NAME=libepsm_$v.so timeout -k 10 120 python tools/prof_gradient_only.py manifold 2>&1 | grep "^gradient_only" | cut -c1-75
  EPSM_LIB_NAME=libepsm_$v.so timeout -k 10 120 python tools/prof_gradient_only.py manifold_caustic only 2>&1 | grep "^gradient_only" | cut -c1-75
done
