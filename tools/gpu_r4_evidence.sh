# Round-4 evidence run: tools/gpu_r4_evidence.sh OUTPREFIX TAG  (smoke, the bench line, kernel-trace stats,
# PMC traffic + SQ counters of the same command, profiles/traffic.json entry)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/${1:-r4}; mkdir -p $(dirname $T)
timeout -k 10 300 python __graft_entry__.py smoke > ${T}_smoke.log 2>&1; tail -4 ${T}_smoke.log
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 45"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_trace -- $B > ${T}_trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_fetch -- $B > ${T}_pmc1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_write -- $B > ${T}_pmc2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_calib -- $B --separate-tangent > ${T}_pmc3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_req -- $B > ${T}_pmc4.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-include-regex "epsm_backward" --output-format csv -d ${T}_pmc_rdreq -- $B > ${T}_pmc7.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU --kernel-include-regex "epsm_backward" --output-format csv -d ${T}_pmc_sq -- $B > ${T}_pmc5.log 2>&1
# (tools/micro/gather128 is built beforehand and travels with the snapshot; "1 0": eight 16-byte quads per lane, no touch -- the kernel's pattern since the touch was dropped)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${T}_pmc_gather -- tools/micro/gather128 1 0 > ${T}_pmc6.log 2>&1
python tools/summarize_rocprof.py ${T}_trace > ${T}_kernel_stats.txt 2>&1
{ for d in fetch write calib req rdreq sq gather; do python tools/summarize_rocprof.py ${T}_pmc_$d; done; } > ${T}_pmc_traffic.txt 2>&1
python tools/make_traffic_json.py ${T}_pmc_fetch ${T}_pmc_write ${T}_pmc_calib --tag ${2:-r04_l} --packed --gather-calib ${T}_pmc_gather --rdreq-dir ${T}_pmc_rdreq --sq-dir ${T}_pmc_sq > ${T}_traffic_entry.json 2>&1
cp profiles/traffic.json ${T}_traffic.json
timeout -k 10 900 python bench.py > ${T}_bench.json 2> ${T}_bench.err; cut -c1-600 ${T}_bench.json; tail -2 ${T}_bench.err
head -8 ${T}_kernel_stats.txt; tail -20 ${T}_traffic_entry.json
