cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-wfko}; shift
for lib in "$@"; do
export EPSM_LIB_NAME=libepsm_$lib.so
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_$lib -- python tools/prof_trace.py wavefront 100 > gpurun_out/${TAG}_$lib.log 2>&1
python tools/summarize_trace_bounces.py gpurun_out/${TAG}_$lib > gpurun_out/${TAG}_${lib}_bounces.txt; echo "== $lib"; grep -E "shade|finish" gpurun_out/${TAG}_${lib}_bounces.txt
done
