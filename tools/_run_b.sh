cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 600 python tools/bench_packed_real.py > gpurun_out/r2m_real.log 2>&1; cat gpurun_out/r2m_real.log | grep -v amdgpu.ids
