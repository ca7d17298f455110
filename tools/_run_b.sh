cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 900 python bench.py > gpurun_out/r2r_bench.json 2> gpurun_out/r2r_bench.err; tail -2 gpurun_out/r2r_bench.err; python -c "
import json; b=json.loads(open('gpurun_out/r2r_bench.json').read().strip().splitlines()[-1])
print('value %.4g'%b['value'], 'ms/step %.2f'%b['ms_per_step'], 'kernel %.3f'%b['roofline']['kernel_ms'], 'frac %.3f'%b['roofline']['frac'], b['roofline']['traffic'], b['config']['resident_bytes_per_gpu']/1e9, b['config']['all_slabs_distinct'], b['config']['slabs_resident_per_gpu'])"
timeout -k 10 600 python bench.py --soa --no-cpu-baseline --steps 5 | python -c "
import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('soa: ms/step %.2f'%b['ms_per_step'], 'kernel %.3f'%b['roofline']['kernel_ms'], b['config']['all_slabs_distinct'], b['roofline']['traffic'])"
