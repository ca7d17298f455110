cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in hip noilp; do
  export EPSM_LIB_NAME=libepsm_$k.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1x_$k -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-real-scene > gpurun_out/r1x_$k.log 2>&1
  grep '^{' gpurun_out/r1x_$k.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$k events: grad %.3f ms' % d['stages_ms']['grad'])"
done
