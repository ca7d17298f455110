# Counters of the primary-ray stage (wave packets) and of the first shade stage: tools/gpu_packet_counters.sh OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; mkdir -p $T
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --kernel-include-regex "epsm_wf_extend_packet|epsm_wf_shade" --output-format csv -d $T/pmc1 -- python3 tools/prof_gradient_only.py manifold only > $T/pmc1.log 2>&1
timeout -k 5 200 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAVES --kernel-include-regex "epsm_wf_extend_packet|epsm_wf_shade" --output-format csv -d $T/pmc2 -- python3 tools/prof_gradient_only.py manifold only > $T/pmc2.log 2>&1
python3 tools/summarize_rocprof.py $T | grep -v "^$" | cut -c1-150
