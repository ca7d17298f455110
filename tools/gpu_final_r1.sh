# Round-1 evidence run: tests, smoke, bench (with cpu baseline), kernel-trace stats, PMC traffic
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=gpurun_out/${1:-r1z}
python -m pytest tests -m gpu -q 2>&1 | tail -3 > ${T}_pytest.log
python __graft_entry__.py smoke > ${T}_smoke.log 2>&1
python bench.py > ${T}_bench.json 2> ${T}_bench.err
python bench.py --two-stage --no-cpu-baseline --no-real-scene > ${T}_bench_two_stage.json 2>> ${T}_bench.err
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-real-scene"
rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_trace -- $B > ${T}_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d ${T}_trace_two_stage -- $B --two-stage > ${T}_trace2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_fetch -- $B > ${T}_pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_write -- $B > ${T}_pmc2.log 2>&1
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_req -- $B > ${T}_pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_fetch2 -- $B --two-stage > ${T}_pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "epsm" --output-format csv -d ${T}_pmc_write2 -- $B --two-stage > ${T}_pmc5.log 2>&1
cat ${T}_pytest.log; tail -2 ${T}_smoke.log; cut -c1-300 ${T}_bench.json
