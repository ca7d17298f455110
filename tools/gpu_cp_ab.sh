# A/B of cp-kernel variants on the four workloads (headline slab, config 2, pool slab, config 5) after the parity tests of the
# product library: tools/gpu_cp_ab.sh OUT NAME...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1.txt; shift; : > $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_backward_per_path.py tests/test_gpu_pipeline.py tests/test_gpu_full_size_packed.py -x -q -m gpu 2>&1 | tail -3 >> $OUT || { cat $OUT; exit 1; }
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1]', 'kernel %.4f ms'%d['stages_ms']['grad'], 'frac %.3f'%d['roofline']['frac'])" >> $OUT; }
for k in "$@"; do
  EPSM_LIB_NAME=libepsm_$k.so python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 45 2>/dev/null | tail -1 | line "$k headline"
  EPSM_LIB_NAME=libepsm_$k.so python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --config 2 2>/dev/null | tail -1 | line "$k config2"
  EPSM_LIB_NAME=libepsm_$k.so python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --config 3 --max-resident-gb 30 2>/dev/null | tail -1 | line "$k pool"
  EPSM_LIB_NAME=libepsm_$k.so python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --config 5 2>/dev/null | tail -1 | line "$k config5"
done
cat $OUT
