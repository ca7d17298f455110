"""Golden vectors of the reference's geomloss-free matcher, ``Matcher.match_sliced_wasserstein``
(EPSM/utils/matcher.py:76-116), run IN PLACE from /root/reference under stubs for what cannot be imported here
(``geomloss`` -- only ``match_Sinkhorn`` uses it -- and ``utils.logger``, image dumps).  Build container only:

    python tests/golden/gen_matcher_golden.py

``tests/golden/matcher_sliced_*.npz``: inputs (rendered / target colours on a res x res grid), the torch seed the call
was made under, the random draws the reference made (PCA basis, slicing directions: captured by wrapping
``torch.pca_lowrank`` / ``torch.rand``) and the returned gradient (res^2, 5).  Only data is written."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("EPSM_REFERENCE_ROOT", "/root/reference")


def load_reference_matcher():
    sys.modules.setdefault("geomloss", types.SimpleNamespace(SamplesLoss=lambda *a, **k: None))
    utils = types.ModuleType("utils"); logger = types.ModuleType("utils.logger"); logger.Logger = object
    sys.modules["utils"], sys.modules["utils.logger"] = utils, logger
    spec = importlib.util.spec_from_file_location("ref_matcher", os.path.join(REF, "EPSM/utils/matcher.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def images(res, seed):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, res), torch.linspace(0, 1, res), indexing="ij")
    blob = lambda cx, cy, s: torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    gt = torch.stack([blob(0.6, 0.4, 0.12), 0.5 * blob(0.3, 0.7, 0.2), 0.2 + 0.1 * xx], dim=-1)
    rd = torch.stack([blob(0.45, 0.5, 0.12), 0.5 * blob(0.35, 0.6, 0.2), 0.2 + 0.1 * yy], dim=-1)
    rd = rd + 0.02 * torch.randn(rd.shape, generator=g)
    return rd.reshape(-1, 3).float(), gt.reshape(-1, 3).float()      # (may leave [0,1]: the matcher clamps)


def main():
    mod = load_reference_matcher()
    for res, seed in ((16, 1), (32, 2)):
        m = mod.Matcher(res, "cpu")
        rd, gt = images(res, seed)
        draws = {}
        real_rand, real_pca = torch.rand, torch.pca_lowrank

        def rand(*a, **k):
            out = real_rand(*a, **k); draws.setdefault("rand", out.clone()); return out

        def pca(*a, **k):
            out = real_pca(*a, **k); draws.setdefault("pca_V", out[2].clone()); return out
        torch.rand, torch.pca_lowrank = rand, pca
        try:
            torch.manual_seed(1000 + seed)
            g = m.match_sliced_wasserstein(rd.clone(), gt.clone())
        finally:
            torch.rand, torch.pca_lowrank = real_rand, real_pca
        path = os.path.join(HERE, f"matcher_sliced_res{res}.npz")
        np.savez_compressed(path, render=rd.numpy(), target=gt.numpy(), seed=np.array(1000 + seed), res=np.array(res),
                            pca_V=draws["pca_V"].numpy(), rand=draws["rand"].numpy(), grad=g.detach().numpy(),
                            num_vectors=np.array(m.num_vectors), num_principle_vectors=np.array(m.num_principle_vectors))
        print(path, g.shape, float(g.abs().max()), os.path.getsize(path))


if __name__ == "__main__":
    main()
