"""For rocprofv3: the Sinkhorn matcher on epsm_sinkhorn_softmin, RES x RES clouds (default 256), three calls."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from epsm_mitsuba3_amd.matcher import Matcher
res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(res)
render = torch.rand((res * res, 3), generator=g).to(dev)
gt = (render + 0.05 * torch.randn((res * res, 3), generator=g).to(dev)).clamp(0, 1)
m = Matcher(res, dev)
for _ in range(3):
    m.match_Sinkhorn(render, gt)
torch.cuda.synchronize()
print("done")
