"""Shared helpers of the test-suite: golden loading and the parity metric."""
from __future__ import annotations

import glob
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(prefix="calc_grad_"):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def golden_id(path):
    return os.path.basename(path)[len("calc_grad_"):-len(".npz")]


def load_golden(path, dtype=torch.float32, device="cpu"):
    """-> (variant, path_info, dlduv, dldp, ref) in the reference's path_info format."""
    z = np.load(path)
    K = int(z["K"])
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(device)
    info = [{"cam": f(z["cam"])}]
    for k in range(1, K + 1):
        pre = f"v{k}_"
        info.append({
            "it": k - 1,
            "active": torch.from_numpy(z[pre + "active"].astype(bool)).to(device),
            "bsdf": torch.from_numpy(z[pre + "bsdf"].astype(np.int64)).to(torch.int32).to(device),
            "ismesh": f(z[pre + "ismesh"]),
            "light": f(z[pre + "light"]),
            "active_em": torch.from_numpy(z[pre + "active_em"].astype(bool)).to(device),
            "points": [f(z[pre + "p0"]), f(z[pre + "p1"]), f(z[pre + "p2"]), f(z[pre + "p"])],
            "uv": [f(z[pre + "b0"]), f(z[pre + "b1"])],
            "normal": f(z[pre + "normal"]),
            "normals": [f(z[pre + "n0"]), f(z[pre + "n1"]), f(z[pre + "n2"])],
            "eta": f(z[pre + "eta"]),
            "hf": f(z[pre + "hf"]),
        })
    ref = {k: torch.from_numpy(z[k]) for k in z.files if k.startswith("ref")}
    return str(z["variant"]), info, f(z["dlduv"]), f(z["dldp"]), ref


def stack3(fp, lg, dg):
    """(P+2K, N, 3) float64 stack of the three output lists."""
    return torch.cat([torch.stack([x.detach().cpu().double() for x in lst]) for lst in (fp, lg, dg)], dim=0)


def parity_report(mine: torch.Tensor, truth: torch.Tensor, yard: torch.Tensor | None = None,
                  clip: float = 0.1, rel: float = 2e-4, yard_factor: float = 16.0):
    """Per-path parity of fp32 results against a float64 truth.

    ``mine``, ``truth``, ``yard``: (A, N, 3) stacks (all outputs of every path).
    A path passes when the max-norm error over its outputs is at most
    ``rel * scale + yard_factor * (error of the yardstick fp32 implementation)``
    with ``scale`` = max |truth| over the path's outputs (>= 1e-6).  Components
    whose truth lies within 2 % of the +-clip outlier threshold (epsm.py:932-944)
    are excluded: the clamp is discontinuous and any rounding flips them.
    Returns dict(frac_bad, worst, n_straddle, median_rel).
    """
    mine = mine.double(); truth = truth.double()
    strad = torch.zeros_like(truth, dtype=torch.bool)
    if clip and clip > 0 and np.isfinite(clip):
        strad = (truth.abs() > 0.98 * clip) | ((truth == 0) & (mine.abs() > 0.98 * clip))
        if yard is not None:
            strad |= (yard.double().abs() > 0.98 * clip) & (truth == 0)
    err = torch.where(strad, torch.zeros_like(truth), (mine - truth).abs())
    # a clamp straddler poisons its whole path only through that component, so mask per component
    scale = torch.where(strad, torch.zeros_like(truth), truth.abs()).amax(dim=(0, 2)).clamp_min(1e-6)
    e_path = err.amax(dim=(0, 2))
    tol = rel * scale
    if yard is not None:
        yerr = torch.where(strad, torch.zeros_like(truth), (yard.double() - truth).abs()).amax(dim=(0, 2))
        tol = tol + yard_factor * yerr
    bad = e_path > tol
    return {
        "frac_bad": float(bad.double().mean()),
        "n_bad": int(bad.sum()),
        "worst": float((e_path / scale).max()),
        "median_rel": float((e_path / scale).median()),
        "n_straddle": int(strad.sum()),
        "bad_idx": torch.nonzero(bad).flatten()[:8].tolist(),
    }


COND_GATE = 1e4          # the fp32 tolerance is stated for paths whose system has cond_2 < 1e4
EPS32 = 6e-8


def gated_parity_report(mine: torch.Tensor, truth: torch.Tensor, cond: torch.Tensor, clip: float = 0.1,
                        rel: float = 2e-4, k_eps: float = 4.0, gate: float = COND_GATE):
    """Per-path parity of fp32 results against a float64 truth with the conditioning gate of SURVEY.md 8c.

    ``cond`` (N,): largest cond_2 among the systems ``cur`` a path's outputs use (oracle.binding.oracle_cond).  A path
    passes when its max-norm error is at most ``scale * max(rel, k_eps * eps32 * cond)`` -- the backward-stable bound
    of an fp32 solve with a modest constant, no yardstick term -- with ``scale`` = max |truth| over the path (>= 1e-6).
    Components within 2 % of the +-clip outlier threshold (epsm.py:932-944) are excluded as in ``parity_report``.
    Returns the bad-path fractions INSIDE the gate (cond < gate: the number the tests bound) and OUTSIDE it (reported),
    the share of paths inside, and the worst relative error inside."""
    mine = mine.double(); truth = truth.double(); cond = cond.double()
    strad = torch.zeros_like(truth, dtype=torch.bool)
    if clip and clip > 0 and np.isfinite(clip):
        strad = (truth.abs() > 0.98 * clip) | ((truth == 0) & (mine.abs() > 0.98 * clip))
    err = torch.where(strad, torch.zeros_like(truth), (mine - truth).abs()).amax(dim=(0, 2))
    scale = torch.where(strad, torch.zeros_like(truth), truth.abs()).amax(dim=(0, 2)).clamp_min(1e-6)
    inside = cond < gate
    tol = scale * torch.clamp(k_eps * EPS32 * cond, min=rel)
    bad = err > tol
    n_in, n_out = int(inside.sum()), int((~inside).sum())
    relerr = err / scale
    return {
        "frac_bad_inside": float((bad & inside).sum()) / max(1, n_in),
        "frac_bad_outside": float((bad & ~inside).sum()) / max(1, n_out),
        "n_bad_inside": int((bad & inside).sum()),
        "gate_share": n_in / max(1, n_in + n_out),
        "worst_inside": float(relerr[inside].max()) if n_in else 0.0,
        "median_rel": float(relerr.median()),
        "bad_idx": torch.nonzero(bad & inside).flatten()[:8].tolist(),
    }


def two_routes_report(a, b, lo, hi, rel=2e-4, rel_all=1e-2):
    """Two float32 routes to the same sums over paths whose PER-PATH arithmetic is not bit-identical (the dense
    calc_grad kernel runs epsm_path_core.h, the fused backward kernel epsm_cp_core.h: same algebra, different
    rounding).  Beyond the order of the additions they may differ (i) where a per-path component lies so close to
    the +-clip outlier threshold (epsm.py:932-944, discontinuous) that the two roundings land on different sides
    of it, and (ii) on paths whose system is ill-conditioned -- among millions of paths a handful of those put a
    component 10 % apart right at the threshold, and one such term is up to ``clip`` of absolute difference.
    ``lo`` / ``hi``: route ``b`` run with the threshold moved by -2 % / +2 %: |lo - hi| is, per element, the weight of
    every component inside that band -- the allowance for (i).  For (ii), which no GPU route can price per element:
    beyond the allowance the MEAN difference within ``rel`` of the buffer's magnitude m, 99.9 % of the elements within
    ``rel_all`` of it, none beyond 0.1 m.  (On a fine mesh nearly all elements are within rel * m each --
    ``frac_tight`` is reported; on a 120-vertex mesh every element sums thousands of paths.)
    Returns dict(m, frac_tight, frac_all, worst_rel, mean_excess_rel, allow_share)."""
    a, b, lo, hi = (t.double().flatten() for t in (a, b, lo, hi))
    m = float(b.abs().max())
    allow = (lo - hi).abs()
    d = (a - b).abs()
    excess = (d - allow).clamp_min(0.0)
    return {"m": m, "frac_tight": float((excess <= rel * m).double().mean()), "frac_all": float((excess <= rel_all * m).double().mean()),
            "worst_rel": float(excess.max() / max(m, 1e-300)), "mean_excess_rel": float(excess.mean() / max(m, 1e-300)),
            "allow_share": float(allow.sum() / b.abs().sum().clamp_min(1e-300))}


def assert_two_routes_agree(a, b, lo, hi, name="", rel=2e-4, rel_all=1e-2):
    rep = two_routes_report(a, b, lo, hi, rel=rel, rel_all=rel_all)
    assert rep["m"] > 0, (name, rep)
    assert rep["mean_excess_rel"] <= rel, (name, rep)
    assert rep["frac_all"] >= 0.999 or (a.numel() < 1000 and rep["worst_rel"] <= rel_all * 2), (name, rep)
    assert rep["worst_rel"] <= 0.1, (name, rep)
    return rep


def select_paths(trace, idx):
    """The paths ``idx`` (int64 tensor, ascending) of a per-field PathTrace as a PathTrace of its own: records, rays and
    addressing gathered; ``path_offset`` is dropped (the result is not a pixel-ordered wavefront: use it with explicit
    tangents, not with the in-kernel first-vertex tangent)."""
    import epsm_mitsuba3_amd as epsm
    cut = lambda t: None if t is None else t[idx.to(t.device)].contiguous()
    deep = lambda v: ([cut(x) for x in v] if isinstance(v, (list, tuple)) else (cut(v) if torch.is_tensor(v) and v.shape[:1] == trace.ray_d.shape[:1] else v))
    pi = [{"cam": cut(trace.path_info[0]["cam"])}] + [{k: deep(v) for k, v in r.items()} for r in trace.path_info[1:]]
    si = [{k: deep(v) for k, v in r.items()} for r in trace.scatter_info]
    return epsm.PathTrace(res=trace.res, spp=trace.spp, ray_o=cut(trace.ray_o), ray_d=cut(trace.ray_d), ray_dx=cut(trace.ray_dx),
                          ray_dy=cut(trace.ray_dy), path_info=pi, scatter_info=si, path_offset=0, n_paths_total=int(idx.numel()))
