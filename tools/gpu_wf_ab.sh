# A/B of tracer library variants: tools/gpu_wf_ab.sh TAG lib1 lib2 ...   (hip = the default build, with per-bounce stats)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-wfab}; shift
timeout -k 10 300 python -m pytest tests/test_gpu_tracer.py -x -q 2>&1 | tail -5
python tools/diag_wf_occ.py > gpurun_out/${TAG}_occdiag.log 2>&1; tail -12 gpurun_out/${TAG}_occdiag.log
for lib in hip "$@"; do
export EPSM_LIB_NAME=libepsm_$lib.so
timeout -k 10 200 python tools/bench_bigscene.py 100 4194304 > gpurun_out/${TAG}_big100_$lib.log 2>&1; echo "== $lib"; tail -4 gpurun_out/${TAG}_big100_$lib.log | grep -v primal
done
export EPSM_LIB_NAME=libepsm_hip.so
timeout -k 10 200 python tools/bench_bigscene.py 400 4194304 > gpurun_out/${TAG}_big400_hip.log 2>&1; echo "== hip 400"; tail -4 gpurun_out/${TAG}_big400_hip.log | grep -v primal
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_hip -- python tools/prof_trace.py wavefront 100 > gpurun_out/${TAG}_hip.log 2>&1
python tools/summarize_trace_bounces.py gpurun_out/${TAG}_hip > gpurun_out/${TAG}_hip_bounces.txt; cat gpurun_out/${TAG}_hip_bounces.txt
