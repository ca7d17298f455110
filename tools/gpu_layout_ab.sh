# A/B of the native log's two placements (records.alloc_log; ABI v7) on ONE build: the interleaved block per path against the two
# dense arrays of ABI v6.  Headline slab, config 2, config 3 (pool caustic), config 5 (human-size small wavefront); then the traced
# scene (tools/prof_real_backward.py prints trace + backward per layout).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --log-layout $1 $2 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$1 $2]', 'kernel %.3f ms'%d['stages_ms']['grad'], 'frac %.3f'%d['roofline']['frac'], 'step %.2f ms'%d['ms_per_step'])"; }
for p in "--max-resident-gb 45" "--config 2" "--config 3 --max-resident-gb 30" "--config 5"; do for k in dense interleaved; do run $k "$p"; done; done
for k in dense interleaved; do python3 tools/prof_real_backward.py manifold $k; done
for k in dense interleaved; do python3 tools/prof_real_backward.py manifold_caustic $k; done
