# A/B of tracer library variants (tools/build_trace_variant.sh): tools/gpu_trace_ab.sh TAG NAME1 ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-tab}; shift
for rep in 1 2; do
for lib in hip "$@"; do
export EPSM_LIB_NAME=libepsm_$lib.so
for n in 100 400; do
timeout -k 10 300 python tools/bench_bigscene.py $n 4194304 > gpurun_out/${TAG}_${lib}_$n.log 2>&1; echo "== $lib $n: $(grep -E '^\[wavefront\] trace\+sparse' gpurun_out/${TAG}_${lib}_$n.log) | $(grep -E '^\[mega\] trace\+sparse' gpurun_out/${TAG}_${lib}_$n.log)"
done
done
done
