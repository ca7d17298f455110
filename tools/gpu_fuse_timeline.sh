# kernel timeline of render_backward on the traced scene with EPSM_TRACE_FUSE_FIRST_HIT (tools/prof_render_backward.py): tools/gpu_fuse_timeline.sh OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/${1:-fuse_tl}; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o rb -- python3 tools/prof_render_backward.py wavefront 4 > $out/run.log 2>&1
python3 tools/trace_timeline.py $(ls $out/rb_kernel_trace.csv $out/*/rb_kernel_trace.csv 2>/dev/null | head -1) 4 | tail -40
