cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/gpu_fused_dbg.sh
for cfg in "0:" "9:" "0:--two-stage"; do
  dbg=${cfg%%:*}; extra=${cfg#*:}
  export EPSM_FUSED_DBG=$dbg
  tag=$(echo "d${dbg}_${extra}" | tr -d ' -')
  B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM --kernel-include-regex "epsm_grad" --output-format csv -d gpurun_out/r1p_$tag -- $B > gpurun_out/r1p.log 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA --kernel-include-regex "epsm_grad" --output-format csv -d gpurun_out/r1p2_$tag -- $B > gpurun_out/r1p.log 2>&1
done
