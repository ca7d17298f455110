# SQ counters of cp-kernel variants on the headline slab: tools/gpu_cp_pmc.sh OUTDIR NAME1 NAME2 ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$1; shift; mkdir -p $O
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 45"
for k in "$@"; do
  EPSM_LIB_NAME=libepsm_$k.so timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU --kernel-include-regex "epsm_backward" --output-format csv -d $O/pmc_$k -- $B > $O/pmc_$k.log 2>&1
  echo "== $k"; python tools/summarize_rocprof.py $O/pmc_$k | grep -v "^#"
done
