# render_backward timings per variant library (no tests): tools/gpu_reparam_ab2.sh OUT NAME...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1.txt; shift; : > $OUT
for k in "$@"; do
  echo "[$k]" >> $OUT
  EPSM_LIB_NAME=libepsm_$k.so timeout -k 10 200 python tools/bench_reparam.py 512 16 16 3 2>&1 | tail -1 | cut -c1-260 >> $OUT
done
cat $OUT
