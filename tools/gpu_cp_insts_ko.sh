# dynamic instruction counts of the headline kernel's stages: SQ counters of knock-out builds (tools/build_cp_variant.sh)
# usage: tools/gpu_cp_insts_ko.sh OUT NAME1 NAME2 ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/$1; shift; mkdir -p $T
for k in "$@"; do
  EPSM_LIB_NAME=libepsm_$k.so timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD --kernel-include-regex "epsm_backward" --output-format csv -d $T/$k -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 30 > $T/$k.log 2>&1
  echo "## $k"; python3 tools/summarize_rocprof.py $T/$k | grep "SQ_" | cut -c1-100
done
