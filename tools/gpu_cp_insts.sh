# instruction counts of cp-kernel variants on the headline slab: tools/gpu_cp_insts.sh NAME...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 45"
for k in "$@"; do
  EPSM_LIB_NAME=libepsm_$k.so timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-include-regex "epsm_backward" --output-format csv -d gpurun_out/insts_$k -- $B > gpurun_out/insts_$k.log 2>&1
  echo "== $k"; python tools/summarize_rocprof.py gpurun_out/insts_$k | grep -v "^#\|^kernel\|vgpr=\|^$" | awk '{printf "%s %s  ", $1, $4} END {print ""}'
done
