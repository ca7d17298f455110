# quick GPU check: parity tests + bench on a few configurations
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-quick}
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/${TAG}_pytest.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-real-scene > gpurun_out/${TAG}_bench.log 2>&1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-real-scene --scene-vertices 7829 > gpurun_out/${TAG}_bench_v7829.log 2>&1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-real-scene --scene-vertices 1000000 > gpurun_out/${TAG}_bench_v1m.log 2>&1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-real-scene --profile specular > gpurun_out/${TAG}_bench_specular.log 2>&1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-real-scene --variant manifold_caustic --profile pool > gpurun_out/${TAG}_bench_caustic_pool.log 2>&1
cat gpurun_out/${TAG}_pytest.log
for f in gpurun_out/${TAG}_bench*.log; do tail -1 $f | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('$f', '%.3e paths/s'%d['value'], '%.3f ms'%d['ms_per_step'], {k: round(v,3) for k,v in d['stages_ms'].items()}, '%.0f GB/s'%d['roofline']['achieved'])
except Exception as e: print('$f', 'FAILED', e)
"; done
