cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-bn}
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -3 gpurun_out/${TAG}_bench.err; python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value"))
print(d.get("real_scene"))
PY
for n in 3 10 20; do
timeout -k 10 200 python tools/bench_bigscene.py $n 4194304 > gpurun_out/${TAG}_big$n.log 2>&1; echo "== $n spheres"; grep -E "^\[(mega|wavefront)\]" gpurun_out/${TAG}_big$n.log
done
