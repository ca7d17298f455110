# prb_reparam on the GPU: the parity tests with their printed numbers, the timing, the kernel trace.  tools/gpu_r3_reparam.sh OUTPREFIX
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/${1:-r3r}; mkdir -p $(dirname $T)
timeout -k 10 600 python -m pytest tests/test_gpu_reparam.py -m gpu -q -s > ${T}_tests.log 2>&1; echo "pytest rc $?" >> ${T}_tests.log
grep -E "grad |error at|passed|failed|^\[0" ${T}_tests.log | cut -c1-700 > ${T}_parity.txt
timeout -k 10 300 python tools/bench_reparam.py 256 16 16 3 > ${T}_bench.txt 2>&1
timeout -k 10 300 python tools/bench_reparam.py 512 16 16 3 >> ${T}_bench.txt 2>&1
timeout -k 10 300 python tools/bench_reparam.py 256 16 64 3 >> ${T}_bench.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_trace -- python tools/bench_reparam.py 512 16 16 3 > ${T}_trace.log 2>&1
python tools/summarize_rocprof.py ${T}_trace > ${T}_kernel_stats.txt 2>&1
cat ${T}_parity.txt ${T}_bench.txt; head -8 ${T}_kernel_stats.txt
