cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for p in bathroom specular; do
EPSM_LIB_NAME=libepsm_o3.so python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --max-resident-gb 45 --profile $p 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('[loads only, $p]', 'kernel %.3f ms'%d['stages_ms']['grad'], 'live bytes %.2f GB'%(r['live_algorithmic_bytes']/1e9), '-> %.2f TB/s of live bytes'%(r['live_algorithmic_bytes']/d['stages_ms']['grad']/1e9))"
done
