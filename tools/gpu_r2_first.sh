# Round-2 first GPU run: tests, smoke, the new bench line (headline workload), config 2 for comparison with round 1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/${1:-r2a}
python -c "import torch; f,t=torch.cuda.mem_get_info(0); print('HBM free/total GB', f/1e9, t/1e9)" > ${T}_mem.log 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > ${T}_pytest.log 2>&1; echo "pytest rc $?" >> ${T}_pytest.log
tail -5 ${T}_pytest.log
timeout -k 10 300 python __graft_entry__.py smoke > ${T}_smoke.log 2>&1; tail -2 ${T}_smoke.log
timeout -k 10 600 python bench.py --config 2 --steps 20 --warmup 3 > ${T}_bench_cfg2.json 2> ${T}_bench_cfg2.err; cut -c1-600 ${T}_bench_cfg2.json; tail -3 ${T}_bench_cfg2.err
timeout -k 10 900 python bench.py > ${T}_bench.json 2> ${T}_bench.err; cut -c1-1500 ${T}_bench.json; tail -3 ${T}_bench.err
