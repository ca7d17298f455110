"""Scene-parameter gradient buffers (what ``dr.grad(params[...])`` holds in the
reference after ``render_backward``, epsm.py:84-306).

All buffers are views into ONE flat fp32 allocation so that a multi-GPU backward
pass needs a single RCCL all-reduce, with no packing copies.
"""
from __future__ import annotations

from typing import Optional

import torch


class ParamGrads:
    """``pos (V,3)``: d/d vertex_positions; ``nrm (V,3)``: d/d vertex_normals;
    ``alpha (B)``: d/d per-BSDF roughness; ``cam_origin (3)``: d/d ray origin
    (epsm.py:260-261); ``color (C,3)``: d/d the attached colour parameters (PRBIntegrator).  Vertex indices are global: meshes are concatenated and a
    mesh's rows are ``pos[offset : offset + n_vertices]`` (see ``mesh_slices``)."""

    def __init__(self, n_vertices: int, n_bsdfs: int = 0, device="cuda", mesh_slices: Optional[dict] = None, n_colors: int = 0):
        self.V, self.B, self.C = int(n_vertices), int(n_bsdfs), int(n_colors)
        n = 6 * self.V + self.B + 3 + 3 * self.C
        self.flat = torch.zeros(n, device=device, dtype=torch.float32)
        self.pos = self.flat[: 3 * self.V].view(self.V, 3)
        self.nrm = self.flat[3 * self.V: 6 * self.V].view(self.V, 3)
        self.alpha = self.flat[6 * self.V: 6 * self.V + self.B]
        self.cam_origin = self.flat[6 * self.V + self.B: 6 * self.V + self.B + 3]
        # colour parameters of the hybrid scheme's second phase (diffuse reflectances, emitter radiances): (C,3)
        self.color = self.flat[6 * self.V + self.B + 3:].view(self.C, 3)
        self.mesh_slices = dict(mesh_slices or {})

    def zero_(self):
        self.flat.zero_()
        return self

    def scratch(self) -> "ParamGrads":
        """A zeroed buffer of the same layout (allocated once, cleared on every call): what ONE backward pass
        contributes before it is summed over the ranks and added to the accumulated gradients."""
        s = getattr(self, "_scratch", None)
        if s is None:
            s = self._scratch = ParamGrads(self.V, self.B, device=self.flat.device, mesh_slices=self.mesh_slices, n_colors=self.C)
        return s.zero_()

    def mesh_pos(self, name: str) -> torch.Tensor:
        lo, hi = self.mesh_slices[name]
        return self.pos[lo:hi]

    def mesh_nrm(self, name: str) -> torch.Tensor:
        lo, hi = self.mesh_slices[name]
        return self.nrm[lo:hi]
