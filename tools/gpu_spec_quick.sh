cd $GRAFT_REPO_ROOT
for k in w512 w256; do EPSM_LIB_NAME=libepsm_$k.so python bench.py --config 2 --profile specular --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[$k specular]', 'kernel %.3f ms'%d['stages_ms']['grad'])"; done
