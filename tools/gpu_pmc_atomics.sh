cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in "$@"; do
export EPSM_LIB_NAME=libepsm_$k.so
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-real-scene 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$k]', '%.2f ms'%d['ms_per_step'])"
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum --kernel-include-regex "epsm_grad" --output-format csv -d gpurun_out/atom_$k -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-scene > gpurun_out/atom_$k.log 2>&1
python tools/summarize_rocprof.py gpurun_out/atom_$k | grep -E "ATOMIC|kernel"
done
