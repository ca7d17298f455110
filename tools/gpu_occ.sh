cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
EPSM_LIB_NAME=libepsm_hip_rg.so python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for lib in libepsm_hip.so libepsm_hip_rg.so libepsm_hip.so libepsm_hip_rg.so; do
  export EPSM_LIB_NAME=$lib
  for extra in "--two-stage" "--two-stage --profile specular" "--two-stage --variant manifold_caustic --profile pool"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline $extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$lib $extra]', {k: round(v,3) for k,v in d['stages_ms'].items()})"
  done
done
