# Per-kernel time of the gradient-only backward trace (clutter, 2^24 paths): tools/gpu_prof_gradient_only.sh OUTDIR [variant]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=${1:-gpurun_out/prof_go}; v=${2:-manifold}
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o go -- python3 tools/prof_gradient_only.py $v only > $out/run.log 2>&1
f=$(ls $out/*kernel_stats.csv $out/*/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# %s" % sys.argv[1])
for r in rows:
    if 'epsm' in r['Name']:
        print("%6s calls  avg %9.1f us  total %9.1f us  %s" % (r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3, r['Name'][:110]))
PY
tail -3 $out/run.log
python3 tools/trace_timeline.py $out/go_kernel_trace.csv 4
