# Tracer, VERDICT r3 item 4b: queue regrouped by the triangle the rays leave (EPSM_WF_REKEY builds: tools/build_trace_variant.sh rkS
# "-DEPSM_WF_REKEY=S") against the product.  tools/gpu_rekey.sh OUT NAME...   ("hip" = product)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$1; shift
echo "# tools/bench_bigscene.py N 4194304 (floor + N tessellated spheres, 512x512 @ 16 spp, max_depth 4, K = 4, one tile; median of 5): [wavefront] lines" > $O
for k in "$@"; do for n in 100 400; do
  echo "== $k, $n spheres" >> $O
  EPSM_LIB_NAME=libepsm_$k.so timeout -k 10 200 python tools/bench_bigscene.py $n 4194304 2>/dev/null | grep -E "^\[wavefront\]" >> $O
done; done
B="python tools/prof_trace.py wavefront 100 4194304 2"
for k in "$@"; do
  EPSM_LIB_NAME=libepsm_$k.so timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rekey_kt_$k -- $B > gpurun_out/rekey_kt_$k.log 2>&1
  echo >> $O; echo "== $k: rocprofv3 --kernel-trace --stats -- $B" >> $O
  python tools/summarize_rocprof.py gpurun_out/rekey_kt_$k | grep "epsm_wf" >> $O
  EPSM_LIB_NAME=libepsm_$k.so timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-include-regex "epsm_wf_(extend|shadow)" --output-format csv -d gpurun_out/rekey_pmc_$k -- $B > gpurun_out/rekey_pmc_$k.log 2>&1
  echo "== $k: --pmc (mean over the dispatches of a kernel); lane utilisation = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)" >> $O
  python tools/summarize_rocprof.py gpurun_out/rekey_pmc_$k | grep -v "^#\|^$" >> $O
done
cat $O
