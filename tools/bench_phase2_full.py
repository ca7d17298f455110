"""BASELINE.json configs[3] names `manifold_hybrid` at 1024 x 1024 @ 256 spp: its SECOND phase (prb_reparam's render_backward,
EPSM/optim.py:113-119) at exactly that size, once -- bench.py's hybrid_phase2 legs run it at 16 spp.
    python tools/bench_phase2_full.py [spp]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = bench.hybrid_phase2_leg(1024, spp, 16, torch.device("cuda", 0))
print(json.dumps(out))
