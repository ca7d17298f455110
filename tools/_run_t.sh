cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for K in 1 2 3; do python bench.py --steps 4 --warmup 1 --no-cpu-baseline --config 2 --vertices $K 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('K=$K kernel %.3f ms'%d['stages_ms']['grad'])"; done
for c in 1 5; do python bench.py --steps 10 --warmup 2 --no-cpu-baseline --config $c 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('config $c kernel %.3f ms frac %.3f'%(d['stages_ms']['grad'], d['roofline']['frac']))"; done
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_tangent_scatter.py tests/test_gpu_tracer.py -x -q 2>&1 | tail -3
