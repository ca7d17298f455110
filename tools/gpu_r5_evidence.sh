# Round-5 evidence run: tools/gpu_r5_evidence.sh OUTPREFIX TAG  (smoke, kernel-trace stats, PMC traffic + SQ counters of the
# headline command and of the dense_specular slab, profiles/traffic.json entries, marker trace of render_backward, the bench line)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/${1:-r5}; TAG=${2:-r05_x}; mkdir -p $(dirname $T)
timeout -k 10 300 python __graft_entry__.py smoke > ${T}_smoke.log 2>&1; tail -4 ${T}_smoke.log
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 45"
S="python bench.py --config 2 --profile specular --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_trace -- $B > ${T}_trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_fetch -- $B > ${T}_pmc1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_write -- $B > ${T}_pmc2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_calib -- $B --separate-tangent > ${T}_pmc3.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-include-regex "epsm_backward" --output-format csv -d ${T}_pmc_rdreq -- $B > ${T}_pmc7.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU --kernel-include-regex "epsm_backward" --output-format csv -d ${T}_pmc_sq -- $B > ${T}_pmc5.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-include-regex "epsm_backward" --output-format csv -d ${T}_pmc_tcp -- $B > ${T}_pmc8.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${T}_pmc_gather -- tools/micro/gather128 1 0 > ${T}_pmc6.log 2>&1
# the dense slab (every vertex live): kernel time + traffic
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_spec_trace -- $S > ${T}_spec_trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_spec_fetch -- $S > ${T}_spec1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_spec_write -- $S > ${T}_spec2.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-include-regex "epsm_backward" --output-format csv -d ${T}_spec_rdreq -- $S > ${T}_spec3.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-include-regex "epsm_backward" --output-format csv -d ${T}_spec_sq -- $S > ${T}_spec4.log 2>&1
# stage markers (epsm_mitsuba3_amd/profiler.py -> ROCTX) of render_backward on the traced scene
timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d ${T}_markers -- python tools/prof_render_backward.py wavefront 3 > ${T}_markers.log 2>&1
python tools/summarize_rocprof.py ${T}_trace > ${T}_kernel_stats.txt 2>&1
python tools/summarize_rocprof.py ${T}_spec_trace >> ${T}_kernel_stats.txt 2>&1
{ for d in fetch write calib rdreq sq tcp gather; do python tools/summarize_rocprof.py ${T}_pmc_$d; done; echo "## dense_specular slab (bench.py --config 2 --profile specular)"; for d in fetch write rdreq sq; do python tools/summarize_rocprof.py ${T}_spec_$d; done; } > ${T}_pmc_traffic.txt 2>&1
python tools/make_traffic_json.py ${T}_pmc_fetch ${T}_pmc_write ${T}_pmc_calib --tag $TAG --packed --gather-calib ${T}_pmc_gather --rdreq-dir ${T}_pmc_rdreq --sq-dir ${T}_pmc_sq > ${T}_traffic_entry.json 2>&1
python tools/make_traffic_json.py ${T}_spec_fetch ${T}_spec_write --tag $TAG --packed --profile specular --rdreq-dir ${T}_spec_rdreq --sq-dir ${T}_spec_sq > ${T}_traffic_entry_specular.json 2>&1
cp profiles/traffic.json ${T}_traffic.json
{ echo "# rocprofv3 --marker-trace --kernel-trace --stats -- python tools/prof_render_backward.py wavefront 3"; for f in $(find ${T}_markers -name "*marker_api_stats.csv" -o -name "*marker*stats*.csv" | head -3); do echo "# $f"; cat $f; done; } > ${T}_markers.txt 2>&1
timeout -k 10 900 python bench.py > ${T}_bench.json 2> ${T}_bench.err; cut -c1-600 ${T}_bench.json; tail -12 ${T}_bench.err
head -8 ${T}_kernel_stats.txt; tail -12 ${T}_traffic_entry.json; tail -6 ${T}_traffic_entry_specular.json; head -20 ${T}_markers.txt
