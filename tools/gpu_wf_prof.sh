cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-wfp}; shift
python tools/prof_trace.py --diff 100 > gpurun_out/${TAG}_diff.log 2>&1; tail -30 gpurun_out/${TAG}_diff.log
for lib in hip "$@"; do
export EPSM_LIB_NAME=libepsm_$lib.so
python tools/bench_bigscene.py 100 4194304 > gpurun_out/${TAG}_big100_$lib.log 2>&1; echo "== $lib"; tail -4 gpurun_out/${TAG}_big100_$lib.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_$lib -- python tools/prof_trace.py wavefront 100 > gpurun_out/${TAG}_$lib.log 2>&1
python tools/summarize_trace_bounces.py gpurun_out/${TAG}_$lib > gpurun_out/${TAG}_${lib}_bounces.txt; cat gpurun_out/${TAG}_${lib}_bounces.txt
done
