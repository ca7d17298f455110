"""A stand-in for the path tracer: seeded synthetic wavefronts with the shape of a
real backward trace (camera rays, logged vertices, parameter addressing).  Used by
bench.py and the tests; scene assets of the reference are not available
(README.md:26) and the native tracer is a later row."""
from __future__ import annotations

from typing import List

import torch

from . import dist as _dist
from .integrators import PathTrace
from .synth import (synth_camera_rays, synth_first_hit_triangles, synth_path_info, synth_scatter_info,
                    synth_triangle_table)


class SyntheticScene:
    def __init__(self, res: int = 128, n_vertices: int = 5, n_scene_vertices: int = 7829, n_bsdfs: int = 4,
                 profile: str = "bathroom", device="cuda", coherent: bool = True, tile_paths: int = _dist.TILE_PATHS,
                 shadow_term: bool = True):
        self.res, self.K, self.V, self.B = res, n_vertices, n_scene_vertices, n_bsdfs
        self.profile, self.device, self.coherent = profile, torch.device(device), coherent
        self.tile_paths = tile_paths
        self.shadow_term = shadow_term       # log the occluder record when max_depth <= 3 (epsm.py:609-620)
        self._table = None

    def triangle_table(self) -> torch.Tensor:
        """The scene's ``(T,4) [v0, v1, v2, mode]`` table (include/epsm.h), shared by all tiles."""
        if self._table is None:
            self._table = synth_triangle_table(self.V, self.device, n_bsdfs=self.B)
        return self._table

    def tile(self, t: int, lo: int, hi: int, seed: int, spp: int, K: int, shadow: bool = False, lean: bool = False) -> PathTrace:
        """Tile t = paths [lo, hi) of the wavefront; its content depends only on (seed, t),
        so any rank regenerates the same tile.  ``lean``: drop the fields of the log the gradient path never reads
        (``hf``, the interpolated ``p`` and ``normal``: 36 B per vertex) -- for wavefronts that fill the HBM."""
        n = hi - lo
        s = seed * 7919 + t
        o, d, dx, dy = synth_camera_rays(self.res, spp, seed=seed, device=self.device, lo=lo, hi=hi)
        pi, _, _ = synth_path_info(n, K, seed=s, device=self.device, profile=self.profile)
        p0, p1, p2, b0, b1 = synth_first_hit_triangles(o, d, seed=s)
        pi[0]["cam"] = o
        pi[1]["points"][0], pi[1]["points"][1], pi[1]["points"][2] = p0, p1, p2
        pi[1]["uv"] = [b0, b1]
        if lean:
            for rec in pi[1:]:
                rec["hf"] = None
                rec["normal"] = None
                rec["points"] = rec["points"][:3]
        si = synth_scatter_info(n, K, self.V, seed=s, device=self.device, n_bsdfs=self.B, coherent=self.coherent,
                                res=self.res, spp=spp, path_offset=lo, shadow=shadow, table=self.triangle_table())
        return PathTrace(res=self.res, spp=spp, ray_o=o, ray_d=d, ray_dx=dx, ray_dy=dy, path_info=pi,
                         scatter_info=si, path_offset=lo, n_paths_total=self.res * self.res * spp)

    def iter_traces(self, sensor=2, seed=0, spp=8, max_depth=6, max_log_depth=5, rank=0, world_size=1,
                    sparse_log=False):
        """This rank's tiles (round-robin over ranks) of the backward wavefront, one at a time."""
        max_depth = 6 if max_depth < 0 else min(int(max_depth), 6)
        K = min(self.K, max_log_depth, max_depth)
        n_total = self.res * self.res * spp
        tiles = _dist.tile_ranges(n_total, self.tile_paths)
        shadow = self.shadow_term and max_depth <= 3
        for t in _dist.my_tiles(len(tiles), rank, world_size):
            yield self.tile(t, *tiles[t], seed, spp, K, shadow)

    def trace_paths(self, *args, **kw) -> List[PathTrace]:
        return list(self.iter_traces(*args, **kw))
