# Child-parallel traversal A/B (round 5): per-dispatch time and lane utilisation of the closest-hit stage of the bounces >= 1
# (epsm_wf_extend_kernel<false> vs epsm_wf_extend_quad_kernel) on the full trace of the 128 k-triangle scene: tools/gpu_quad_ab.sh OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T; : > $T/summary.txt
for k in hip quad; do
  export EPSM_LIB_NAME=libepsm_$k.so
  echo "## build $k" >> $T/summary.txt
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $T/${k}_trace -- python tools/prof_trace.py wavefront 100 > $T/${k}_trace.log 2>&1
  python tools/summarize_rocprof.py $T/${k}_trace | grep "epsm_wf\|calls" >> $T/summary.txt 2>&1
  timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES --kernel-include-regex "epsm_wf_extend" --output-format csv -d $T/${k}_pmc -- python tools/prof_trace.py wavefront 100 > $T/${k}_pmc.log 2>&1
  python tools/summarize_rocprof.py $T/${k}_pmc >> $T/summary.txt 2>&1
done
grep -v "^$" $T/summary.txt | grep -v "vgpr=" | cut -c1-160
