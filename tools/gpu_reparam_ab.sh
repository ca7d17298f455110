# prb_reparam: parity tests of the product library, then render_backward timings per variant library: tools/gpu_reparam_ab.sh OUT NAME...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1.txt; shift; : > $OUT
timeout -k 10 700 python -m pytest tests/test_gpu_reparam.py -x -q -m gpu 2>&1 | tail -3 >> $OUT || { cat $OUT; exit 1; }
for k in hip "$@"; do
  echo "[$k]" >> $OUT
  EPSM_LIB_NAME=libepsm_$k.so timeout -k 10 200 python tools/bench_reparam.py 512 16 16 3 2>&1 | tail -1 >> $OUT
  EPSM_LIB_NAME=libepsm_$k.so timeout -k 10 200 python tools/bench_reparam.py 256 16 64 3 2>&1 | tail -1 >> $OUT
done
cat $OUT
