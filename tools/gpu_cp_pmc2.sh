# More counters of cp-kernel variants on the headline slab, in separate passes: tools/gpu_cp_pmc2.sh OUTDIR NAME1 NAME2 ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$1; shift; mkdir -p $O
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 45"
for k in "$@"; do
  n=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" \
             "FETCH_SIZE WRITE_SIZE"; do
    n=$((n+1))
    EPSM_LIB_NAME=libepsm_$k.so timeout -k 10 300 rocprofv3 --pmc $set --kernel-include-regex "epsm_backward" --output-format csv -d $O/pmc_${k}_$n -- $B > $O/pmc_${k}_$n.log 2>&1
    echo "== $k pass $n"; python tools/summarize_rocprof.py $O/pmc_${k}_$n | grep -v "^#"
  done
done
