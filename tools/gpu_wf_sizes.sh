cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-wfs}
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for n in 3 10 30 100 400; do
timeout -k 10 200 python tools/bench_bigscene.py $n 4194304 > gpurun_out/${TAG}_big$n.log 2>&1; echo "== $n spheres"; tail -4 gpurun_out/${TAG}_big$n.log | grep -v primal
done
timeout -k 10 200 python tools/bench_bigscene.py 100 > gpurun_out/${TAG}_big100_tiles.log 2>&1; echo "== 100 spheres, 2^20 tiles"; tail -4 gpurun_out/${TAG}_big100_tiles.log | grep -v primal
