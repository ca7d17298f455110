cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-wfab}; shift
for lib in hip "$@"; do
export EPSM_LIB_NAME=libepsm_$lib.so
timeout -k 10 200 python tools/bench_bigscene.py 100 4194304 > gpurun_out/${TAG}_big100_$lib.log 2>&1; echo "== $lib"; tail -4 gpurun_out/${TAG}_big100_$lib.log | grep -v primal
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_$lib -- python tools/prof_trace.py wavefront 100 > gpurun_out/${TAG}_$lib.log 2>&1
python tools/summarize_trace_bounces.py gpurun_out/${TAG}_$lib > gpurun_out/${TAG}_${lib}_bounces.txt; cat gpurun_out/${TAG}_${lib}_bounces.txt
done
