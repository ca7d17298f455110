cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-bvh}
run() { timeout -k 10 300 python tools/bench_bigscene.py ${N:-100} 4194304 > gpurun_out/${TAG}_$1.log 2>&1; echo "== $1"; grep -E "^\[(mega|wavefront)\] trace\+sparse|build_bvh|scene build" gpurun_out/${TAG}_$1.log; }
EPSM_SAH_MIN=4 EPSM_LEAF_SIZE=4 run sah4
EPSM_SAH_MIN=6 EPSM_LEAF_SIZE=6 run sah6leaf6
EPSM_SAH_MIN=4 EPSM_LEAF_SIZE=6 run sah4leaf6
EPSM_SAH_MIN=7 EPSM_LEAF_SIZE=7 run sah7leaf7
EPSM_SAH_MIN=3 EPSM_LEAF_SIZE=3 run sah3leaf3
N=400 EPSM_SAH_MIN=4 EPSM_LEAF_SIZE=4 run n400_sah4
N=400 run n400_default
