"""Build of the four-wide BVH (scene.build_bvh) on geometry with mixed triangle scales: the depth of the wide tree is
bounded by construction (the traversal stack holds 16 wide levels, csrc/epsm_trace_core.h kBvhStack)."""
import numpy as np
import torch

from epsm_mitsuba3_amd import scene as S


def _soup(centres, sizes, rng):
    t = centres.shape[0]
    pos = (centres[:, None, :] + rng.normal(size=(t, 3, 3)) * sizes[:, None, None]).reshape(-1, 3)
    return pos, np.arange(3 * t, dtype=np.int64).reshape(t, 3)


def _check(plan, pos, tri, leaf_size=S.LEAF_SIZE):
    nodes = plan["nodes"]; inodes = nodes.view(np.int32)
    T = tri.shape[0]
    assert sorted(plan["order"].tolist()) == list(range(T))                    # every triangle in exactly one leaf
    p = pos[tri[plan["order"]]]                                                # leaf order
    lo_t, hi_t = p.min(axis=1), p.max(axis=1)
    seen = np.zeros(T, dtype=np.int64)
    max_depth = 0
    stack = [(0, 0, np.full(3, -np.inf), np.full(3, np.inf))]
    while stack:
        ni, d, plo, phi = stack.pop()
        max_depth = max(max_depth, d)
        for s in range(4):
            c = int(inodes[ni, 24 + s])
            if c == 0x7fffffff:
                continue
            lo = nodes[ni, [s, 4 + s, 8 + s]].astype(np.float64); hi = nodes[ni, [12 + s, 16 + s, 20 + s]].astype(np.float64)
            assert np.all(lo >= plo - 1e-4 * (1 + np.abs(plo))) and np.all(hi <= phi + 1e-4 * (1 + np.abs(phi)))   # inside the parent's box
            if c < 0:
                a, n = (~c) >> 3, (~c) & 7
                assert 1 <= n <= leaf_size and n == int(inodes[ni, 28 + s])
                assert np.all(lo_t[a:a + n] >= lo - 1e-4 * (1 + np.abs(lo))) and np.all(hi_t[a:a + n] <= hi + 1e-4 * (1 + np.abs(hi)))
                seen[a:a + n] += 1
            else:
                stack.append((c, d + 1, lo, hi))
    assert np.all(seen == 1)
    return max_depth + 1


def _built_depth(pos, tri):
    """build -> boxes through the refit the device uses -> structural check; returns the number of wide levels"""
    plan = S.build_bvh(pos, tri)
    dev = S.DeviceBvh(plan, "cpu")
    dev.refit(torch.from_numpy(pos.astype(np.float32)), torch.from_numpy(tri))
    filled = dict(plan); filled["nodes"] = dev.nodes.numpy()
    return _check(filled, pos.astype(np.float32).astype(np.float64), tri)


def test_teapot_in_a_stadium_builds_inside_the_traversal_stack():
    """20 000 triangles whose sizes span six decades and whose positions are clustered at several scales (this
    raised 'BVH deeper than the traversal stack' once the collapse had gone four-wide)."""
    rng = np.random.default_rng(7)
    t = 20000
    sizes = 10.0 ** rng.uniform(-5, 1, size=t)
    cl = rng.normal(size=(12, 3)) * 10.0 ** rng.uniform(-2, 2, size=(12, 1))
    centres = cl[rng.integers(0, 12, size=t)] + rng.normal(size=(t, 3)) * 10.0 ** rng.uniform(-4, 1, size=(t, 1))
    pos, tri = _soup(centres, sizes, rng)
    assert _built_depth(pos, tri) <= S.kMaxWideDepth


def test_geometric_chain_is_folded_into_sixteen_wide_levels():
    """Triangles of size 2^-i at distance 2^-i from a corner: every SAH plane peels a few of them off, the binary tree is
    a chain -- the collapse must still fit, and the boxes must still hold their triangles."""
    rng = np.random.default_rng(1)
    t = 1500
    s = 2.0 ** (-np.arange(t) / 25.0)
    centres = np.stack([s * 3.0, s * 2.0, s], axis=1)
    pos, tri = _soup(centres, 0.05 * s, rng)
    assert _built_depth(pos, tri) <= S.kMaxWideDepth


def test_uniform_mesh_keeps_the_area_heuristic():
    """On ordinary geometry no subtree is too tall: the collapse is the area-ordered one (same node count as before the bound)."""
    rng = np.random.default_rng(3)
    t = 4000
    pos, tri = _soup(rng.uniform(-1, 1, size=(t, 3)), np.full(t, 0.02), rng)
    assert _built_depth(pos, tri) <= 10
