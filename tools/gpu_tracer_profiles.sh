# Tracer evidence run -> gpurun_out/${TAG}_tracer.txt (copied to profiles/)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r1h}
O=gpurun_out/${TAG}_tracer.txt
echo "# tools/bench_bigscene.py N_SPHERES 4194304: floor + N tessellated spheres (1280 triangles each), 512x512 @ 16 spp = 4 194 304 paths," > $O
echo "# max_depth 4, K = 4, one tile; median of 5" >> $O
for n in 30 100 400; do
timeout -k 10 300 python tools/bench_bigscene.py $n 4194304 > gpurun_out/${TAG}_big$n.log 2>&1; echo "== $n spheres" >> $O; grep -E "^\[(mega|wavefront)\]" gpurun_out/${TAG}_big$n.log >> $O
done
timeout -k 10 200 python tools/bench_real.py 512 64 > gpurun_out/${TAG}_real.log 2>&1; echo "== tools/bench_real.py 512 64 (analytic plate scene, one-launch tracer)" >> $O; tail -2 gpurun_out/${TAG}_real.log >> $O
for m in wavefront mega; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kt_$m -- python tools/prof_trace.py $m 100 > gpurun_out/${TAG}_kt_$m.log 2>&1
echo >> $O; echo "== rocprofv3 --kernel-trace --stats -- python tools/prof_trace.py $m 100  (3 traces of 4 194 304 paths, 128 k triangles)" >> $O
python tools/summarize_rocprof.py gpurun_out/${TAG}_kt_$m | grep -v "^$" >> $O
done
echo >> $O; echo "== per bounce (tools/summarize_trace_bounces.py over the kernel trace above)" >> $O
python tools/summarize_trace_bounces.py gpurun_out/${TAG}_kt_wavefront | grep -v "^#" >> $O
B="python tools/prof_trace.py wavefront 100 4194304 2"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-include-regex "epsm_wf" --output-format csv -d gpurun_out/${TAG}_pmc1 -- $B > gpurun_out/${TAG}_pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM --kernel-include-regex "epsm_wf" --output-format csv -d gpurun_out/${TAG}_pmc2 -- $B > gpurun_out/${TAG}_pmc2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-include-regex "epsm_wf" --output-format csv -d gpurun_out/${TAG}_pmc3 -- $B > gpurun_out/${TAG}_pmc3.log 2>&1
for i in 1 2 3; do echo >> $O; echo "== rocprofv3 --pmc (pass $i) -- $B   (mean over the 4 bounces x 2 traces of a kernel)" >> $O; python tools/summarize_rocprof.py gpurun_out/${TAG}_pmc$i | grep -v "^$" >> $O; done
cat $O | head -60
