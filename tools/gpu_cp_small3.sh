# Where a small wavefront's time goes (config 5: 524 288 paths, K = 2): knock-out builds, window sizes, no replicas, and the
# per-kernel durations of the product against its step time.  tools/gpu_cp_small3.sh OUT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/$1.txt; : > $OUT
one() { python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --config 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1]', 'kernel %.4f ms'%d['stages_ms']['grad'], 'step %.4f ms'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'])" >> $OUT; }
for k in s_base s_norounds s_nosolve s_noemit s_noinsert s_slots1536 s_slots384; do EPSM_LIB_NAME=libepsm_$k.so one $k; done
EPSM_NO_REPLICAS=1 EPSM_LIB_NAME=libepsm_s_base.so one "s_base no replicas"
EPSM_NO_REPLICAS=1 EPSM_LIB_NAME=libepsm_s_slots384.so one "s_slots384 no replicas"
rocprofv3 --kernel-trace --stats -d gpurun_out/$1_prof -o t -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --config 5 > /dev/null 2>&1
python tools/summarize_rocprof.py gpurun_out/$1_prof >> $OUT 2>&1 || ls -R gpurun_out/$1_prof >> $OUT
cat $OUT
