"""The reference's integrator surface for the EPSM gradient path.

Mirrors ``EPSMIntegrator`` / ``ManifoldIntegrator`` / ``ManifoldCausticIntegrator``
(src/python/python/ad/integrators/epsm.py:12-306, 744-948, 951-1202) and the plugin
registration the drivers use (``mi.register_integrator`` epsm.py:948,1202;
``mi.load_dict({'type': 'manifold', 'max_depth': d})`` EPSM/optim.py:97-100).

The reference's ``render_backward`` (epsm.py:84-306) is
    trace + log paths  ->  first-vertex tangent  ->  calc_grad  ->  re-trace in
    Backward mode so Dr.Jit scatters into the parameter gradients.
Here the path tracer is whatever object implements ``trace_paths`` (a native
wavefront tracer, or seeded synthetic records for the benchmark); it logs the
vertex INDICES next to the vertex records, so the second trace is replaced by one
scatter kernel.  Everything between the trace and the parameter gradients runs in
three HIP launches (tangent, gradient, scatter) on the current stream.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import dist as _dist
from . import profiler as _prof
from .manifold_grad import OUTLIER_CLIP, calc_grad as _calc_grad, manifold_grad_packed
from .params import ParamGrads
from .records import PackedLog, PackedRecords, PackedScatter
from .tangent_scatter import (backward_pass, backward_pass_packed, first_vertex_tangent, manifold_grad_scatter,
                              scatter)


def sample_tea_32(v0: int, v1: int, rounds: int = 4):
    """drjit's ``sample_tea_32`` on two 32-bit integers (the Tiny Encryption Algorithm, 4 rounds by default) -- what the
    sampler seeds its streams with (src/render/sampler.cpp:115-134) and what ``mi.render`` derives the differential
    seed from (src/python/python/util.py:505-508)."""
    v0 &= 0xFFFFFFFF; v1 &= 0xFFFFFFFF
    s = 0
    for _ in range(rounds):
        s = (s + 0x9E3779B9) & 0xFFFFFFFF
        v0 = (v0 + ((((v1 << 4) + 0xA341316C) ^ (v1 + s) ^ ((v1 >> 5) + 0xC8013EA4)) & 0xFFFFFFFF)) & 0xFFFFFFFF
        v1 = (v1 + ((((v0 << 4) + 0xAD90777D) ^ (v0 + s) ^ ((v0 >> 5) + 0x7E95761E)) & 0xFFFFFFFF)) & 0xFFFFFFFF
    return v0, v1


def render_seeds(seed: int, seed_grad: int = 0):
    """(seed, seed_grad) of one ``mi.render`` call, src/python/python/util.py:505-513: the differential pass gets its OWN
    seed -- ``sample_tea_32(seed, 1)[0]`` unless the caller names one -- because ``grad_in`` is a function of the primal
    image: replaying the very samples whose noise it carries correlates the two factors of the estimator and biases the
    gradient (E[grad] picks up a variance-gradient term).  An explicit ``seed_grad == seed`` is refused with the
    reference's message."""
    if seed_grad == 0:
        seed_grad = sample_tea_32(seed, 1)[0]
    elif seed_grad == seed:
        raise Exception('The primal and differential seed should be different '
                        'to ensure unbiased gradient computation!')
    return seed, seed_grad


@dataclass
class PathTrace:
    """What one logging trace of the backward sensor yields (epsm.py:166-181, 547, 648-654)
    plus the parameter addressing of every logged vertex."""
    res: int                      # film side of the backward sensor
    spp: int
    ray_o: torch.Tensor           # (N,3) primary rays, N = res*res*spp ordered (pixel, sample)
    ray_d: torch.Tensor
    ray_dx: torch.Tensor
    ray_dy: torch.Tensor
    path_info: List[dict]         # [{"cam"}, vertex 1, ..., vertex K]
    scatter_info: List[dict]      # K entries (EpsmScatterRecord fields)
    path_offset: int = 0          # first path of this shard within the full wavefront
    n_paths_total: Optional[int] = None
    log: Optional[PackedLog] = None   # the native log (then path_info / scatter_info may be None)


class EPSMIntegrator:
    """Base class; ``variant`` selects the calc_grad flavour."""
    variant: str = ""

    def __init__(self, props: Optional[dict] = None):
        props = dict(props or {})
        max_depth = props.get("max_depth", 6)           # common.py:31-37
        if max_depth < 0 and max_depth != -1:
            raise Exception("\"max_depth\" must be set to -1 (infinite) or a value >= 0")
        self.max_depth = max_depth            # -1 = infinite (the reference stores 0xffffffff, common.py:37)
        self.rr_depth = props.get("rr_depth", 5)        # common.py:39-41
        if self.rr_depth <= 0:
            raise Exception("\"rr_depth\" must be set to a value greater than zero!")
        # hard-coded in the reference (epsm.py:142,145,549,648,932-944); explicit options here
        self.backward_sensor = props.get("backward_sensor", 2)
        self.backward_spp = props.get("backward_spp", 8)
        self.fuse_tangent = bool(props.get("fuse_tangent", True))   # with `fused`: epsm_backward_pass (one launch per tile)
        self.max_log_depth = min(props.get("max_log_depth", 5), 5)
        self.outlier_clip = props.get("outlier_clip", OUTLIER_CLIP)
        # True: calc_grad and the scatter run as ONE kernel (no dense per-path gradient lists);
        # False: the reference's two-stage shape (calc_grad lists, then scatter)
        self.fused = props.get("fused", True)
        # True: render_backward asks the tracer for the native packed log (one 128-byte record per path vertex) and runs
        # epsm_backward_pass_packed on it; False: the reference's per-field tensors all the way
        self.packed_log = bool(props.get("packed_log", True))
        # True: the backward trace retires a path as soon as nothing behind its last logged vertex can reach calc_grad
        # (EPSM_TRACE_GRADIENT_ONLY, include/epsm_trace.h): identical gradients, the image of that pass -- which the 5-channel
        # branch never uses (epsm.py:729-732) -- is not formed
        self.gradient_only = bool(props.get("gradient_only", True))
        # True: with the native log and the gradient-only trace, the stage that shades a path's FIRST hit also does what the backward
        # pass would do for a path without a chain (EPSM_TRACE_FUSE_FIRST_HIT, include/epsm_trace.h: first-vertex rows, d / d ray.o);
        # such paths -- most of a real wavefront -- are neither logged nor read again.  Same sums.
        self.fuse_first_hit = bool(props.get("fuse_first_hit", True))

    def to_string(self):
        md = 0xFFFFFFFF if self.max_depth < 0 else self.max_depth
        return f"{type(self).__name__}[max_depth = {md}, rr_depth = {self.rr_depth}]"

    __repr__ = to_string

    # -- calc_grad: drop-in -------------------------------------------------
    def calc_grad(self, path_info, dlduv, dldp, Lt=None):
        """epsm.py:745 / 952 -- same arguments and return structure."""
        return _calc_grad(self.variant, path_info, dlduv, dldp, Lt, clip=self.outlier_clip)

    # -- primal -------------------------------------------------------------
    def render(self, scene, sensor=0, seed=0, spp=0, develop=True, evaluate=True):
        """epsm.py:13-82: primal image with two zero channels appended, (H,W,5); the
        5-channel shape is what makes the driver take the EPSM branch (EPSM/optim.py:130)."""
        if not develop:
            raise Exception("develop=True must be specified when invoking AD integrators")
        if not hasattr(scene, "render_primal"):
            raise NotImplementedError("scene object has no render_primal(sensor, seed, spp, max_depth)")
        with _prof.phase("epsm.render.primal"):
            img = scene.render_primal(sensor=sensor, seed=seed, spp=spp, max_depth=self.primal_depth())
        pad = torch.zeros(img.shape[0], img.shape[1], 2, device=img.device, dtype=img.dtype)
        self.primal_image = torch.cat([img[..., :3], pad], dim=-1)
        return self.primal_image

    def primal_depth(self) -> int:
        """The primal pass has no 6-bounce cap (recorded loop, epsm.py:308-501); -1 = as deep as the tracer goes."""
        return 1 << 20 if self.max_depth < 0 else int(self.max_depth)

    # -- backward -----------------------------------------------------------
    def render_backward(self, scene, params: ParamGrads, grad_in: torch.Tensor,
                        sensor=0, seed: int = 0, spp: int = 0) -> None:
        """epsm.py:84-306.  ``sensor`` and ``spp`` are ignored exactly as the reference
        ignores them (epsm.py:142,145): the backward pass uses ``backward_sensor`` and
        ``backward_spp``.  Gradients are ACCUMULATED into ``params`` (as dr.backward does)."""
        if grad_in.shape[-1] == 3:
            # epsm.py:230-234: a 3-channel gradient image is the colour adjoint deltaL = grad_in of the PRB replay, on
            # the sensor and sample count the caller names (the 5-channel branch alone ignores them, epsm.py:142-145).
            # The manifold integrators move geometry through channels 3, 4 only (epsm.py:729-732 leaves the colour
            # adjoint's propagation into geometry commented out): what a 3-channel image can reach are the colour
            # parameters attached to the scene.
            prb = PRBIntegrator({"max_depth": self.max_depth, "rr_depth": self.rr_depth})
            return prb.render_backward(scene, params, grad_in, sensor=sensor, seed=seed, spp=spp)
        rank, world = _dist.world()
        # dr.backward ACCUMULATES into the gradients that are already there.  With more than one rank only THIS
        # call's contribution may be summed over the ranks: what `params` held on entry is already a sum over the
        # ranks (or the caller's own data) and must not be multiplied by the world size.
        target = params.scratch() if world > 1 else params
        tracer = getattr(scene, "iter_traces", None) or scene.trace_paths      # a generator: one tile resident at a time
        kw = {}
        if self.fused and self.fuse_tangent and self.packed_log and getattr(scene, "supports_packed_log", False):
            kw["packed_log"] = True        # the tracer writes the backward kernel's native layout (EpsmPackedLog)
        if self.gradient_only and getattr(scene, "supports_gradient_only", False):
            kw["gradient_only"] = self.variant
            if self.fuse_first_hit and kw.get("packed_log") and getattr(scene, "supports_first_hit_fusion", False):
                kw["first_hit"] = (grad_in if grad_in.is_contiguous() else grad_in.contiguous(), target, self.outlier_clip, True)
        traces = tracer(sensor=self.backward_sensor, seed=seed, spp=self.backward_spp,
                        max_depth=self.tracer_depth(), max_log_depth=self.max_log_depth, rank=rank, world_size=world,
                        sparse_log=True,   # the log is consumed here and nowhere else: skip the zeros of dead bounces
                        **kw)
        if isinstance(traces, PathTrace):
            traces = [traces]
        traces = iter(traces)
        while True:                                # this rank's pixel/sample tiles
            with _prof.phase("epsm.render_backward.trace"):          # (a generator: the tile is traced when asked for)
                trace = next(traces, None)
            if trace is None:
                break
            with _prof.phase("epsm.render_backward.backward_pass"):
                self.backward_from_trace(trace, target, grad_in)
            del trace
        if world > 1:
            with _prof.phase("epsm.render_backward.allreduce"):
                _dist.allreduce_param_grads(target.flat)   # one RCCL all-reduce of the whole buffer
                params.flat += target.flat

    def tracer_depth(self) -> int:
        """``max_depth`` as the tracer takes it: the path loop stops after 6 bounces whatever the integrator says
        (epsm.py:549) and -1 means "no limit" (common.py:31-37)."""
        return 6 if self.max_depth < 0 else min(self.max_depth, 6)

    def backward_from_trace(self, trace: PathTrace, params: ParamGrads, grad_in: torch.Tensor,
                            packed=None, out=None, mark: Optional[Callable[[str], None]] = None,
                            fused: Optional[bool] = None):
        """Tangent -> gradient -> scatter for one tile.  ``packed`` = (PackedRecords, PackedScatter)
        built once for records that stay resident; ``out`` = reusable output tensors; ``mark(name)`` is
        called after each stage (bench.py records HIP events there)."""
        mark = mark or (lambda name: None)
        if isinstance(packed, PackedLog) or getattr(trace, "log", None) is not None:
            # the native log (one 128-byte record per path vertex): one launch, nothing else
            log = packed if isinstance(packed, PackedLog) else trace.log
            mark("tangent")
            # (a log traced under EPSM_TRACE_FUSE_FIRST_HIT: d / d ray.o and the paths without a chain are in the buffers already)
            backward_pass_packed(self.variant, log, grad_in, trace.spp, trace.res, params.pos, params.nrm,
                                 params.alpha if params.B else None, None if log.first_hit_done else params.cam_origin,
                                 clip=self.outlier_clip, path_offset=trace.path_offset)
            mark("grad"); mark("scatter")
            return None
        dev = trace.ray_d.device
        rec, sc = packed if packed is not None else (PackedRecords(trace.path_info, device=dev),
                                                     PackedScatter(trace.scatter_info, device=dev))
        fused = self.fused if fused is None else fused
        if fused and self.fuse_tangent:
            # one launch for the whole tile: tangents live in registers, gradients go into the parameter buffers
            mark("tangent")
            backward_pass(self.variant, rec, sc, trace.ray_o, trace.ray_d, trace.ray_dx, trace.ray_dy, grad_in,
                          trace.spp, trace.res, params.pos, params.nrm, params.alpha if params.B else None,
                          params.cam_origin, clip=self.outlier_clip, path_offset=trace.path_offset)
            mark("grad"); mark("scatter")
            return None
        first = trace.path_info[1]
        dlduv, dldp, grad_o = first_vertex_tangent(
            trace.ray_o, trace.ray_d, trace.ray_dx, trace.ray_dy, grad_in, trace.spp, trace.res,
            first["points"][0], first["points"][1], first["points"][2], first["active"],
            dlduv_width=2, want_origin_grad=True, path_offset=trace.path_offset)
        mark("tangent")
        if fused:
            # gradients go from registers into the parameter buffers
            manifold_grad_scatter(self.variant, rec, sc, dlduv, dldp, params.pos, params.nrm,
                                  params.alpha if params.B else None, clip=self.outlier_clip)
            params.cam_origin += grad_o
            mark("grad")
            mark("scatter")
            return None
        out = manifold_grad_packed(self.variant, rec, dlduv, dldp, clip=self.outlier_clip, dlduv_cols=2, out=out)
        mark("grad")
        scatter(self.variant, rec, sc, *out, params.pos, params.nrm, params.alpha if params.B else None)
        params.cam_origin += grad_o
        mark("scatter")
        return out


class ManifoldIntegrator(EPSMIntegrator):
    variant = "manifold"


class ManifoldCausticIntegrator(EPSMIntegrator):
    variant = "manifold_caustic"


def film_adjoint(film_pos: torch.Tensor, grad_img: torch.Tensor, weight_img: torch.Tensor, rfilter: int) -> torch.Tensor:
    """Adjoint of ImageBlock::put + film.develop (``epsm_film_splat`` / ``epsm_film_develop``) w.r.t. the radiance of
    every sample: image[p] = sum_i w_ip L_i / W_p, so dL_i = sum_p grad[p] w_ip / W_p -- the box filter touches the
    pixel under the sample, the gaussian (stddev 0.5, radius 2, src/rfilters/gaussian.cpp) its 5x5 window.
    ``film_pos (n,2)``, ``grad_img (H,W,3)``, ``weight_img (H,W)`` = W_p of the primal pass; returns ``(n,3)``."""
    H, W = weight_img.shape
    g = grad_img[..., :3] / weight_img.clamp_min(1e-30)[..., None]
    g = torch.where((weight_img > 0)[..., None], g, torch.zeros_like(g))
    px, py = film_pos[:, 0], film_pos[:, 1]
    X, Y = torch.floor(px).long(), torch.floor(py).long()
    if rfilter == 0:                                      # EPSM_RFILTER_BOX
        ok = (X >= 0) & (Y >= 0) & (X < W) & (Y < H)
        return g[Y.clamp(0, H - 1), X.clamp(0, W - 1)] * ok[:, None]
    radius, alpha = 2.0, -1.0 / (2.0 * 0.5 * 0.5)
    bias = math.exp(alpha * radius * radius)
    off = torch.arange(-2, 3, device=film_pos.device)
    xs, ys = X[:, None] + off[None, :], Y[:, None] + off[None, :]             # (n,5)
    dx, dy = (xs.float() + 0.5) - px[:, None], (ys.float() + 0.5) - py[:, None]
    wx = torch.where(dx.abs() <= radius, (torch.exp(alpha * dx * dx) - bias).clamp_min(0), torch.zeros_like(dx))
    wy = torch.where(dy.abs() <= radius, (torch.exp(alpha * dy * dy) - bias).clamp_min(0), torch.zeros_like(dy))
    wx = wx * ((xs >= 0) & (xs < W)); wy = wy * ((ys >= 0) & (ys < H))
    gw = g[ys.clamp(0, H - 1)[:, :, None], xs.clamp(0, W - 1)[:, None, :]]      # (n,5,5,3)
    return (gw * (wy[:, :, None] * wx[:, None, :])[..., None]).sum(dim=(1, 2))


def film_adjoint_reparam(film_pos: torch.Tensor, radiance: torch.Tensor, grad_img: torch.Tensor, accum: torch.Tensor):
    """Adjoint of the gaussian splat + weight division w.r.t. a sample's radiance, its FILM POSITION and the determinant
    of the reparameterisation that multiplies both its value and its weight (common.py:880-920):
        image[p] = sum_i w_ip L_i det_i / sum_i w_ip det_i,   w_ip = f(p - pos_i)
    ``accum (H,W,4)``: the film [r,g,b,w] of the primal pass.  Returns ``dL (n,3)`` and ``adj (n,3)`` =
    [d loss / d pos.x, d loss / d pos.y, d loss / d det] at det = 1.
    On the GPU: ONE kernel (``epsm_film_adjoint_reparam``, include/epsm_trace.h); the torch form below is what it is checked
    against (tests/test_gpu_reparam.py) and what the host build of the tracer runs with."""
    if film_pos.is_cuda:
        import ctypes as C
        from . import _lib
        n = int(film_pos.shape[0])
        fp, rad = film_pos.detach().float().contiguous(), radiance.detach().float().contiguous()
        g, acc = grad_img.detach().float().contiguous(), accum.detach().float().contiguous()
        dL = torch.empty((n, 3), device=film_pos.device, dtype=torch.float32)
        adj = torch.empty((n, 3), device=film_pos.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(film_pos.device).cuda_stream
        _lib.check(_lib.lib().epsm_film_adjoint_reparam(n, fp.data_ptr(), rad.data_ptr(), g.data_ptr(), int(g.shape[-1]), acc.data_ptr(),
                                                         int(acc.shape[1]), int(acc.shape[0]), dL.data_ptr(), adj.data_ptr(),
                                                         C.c_void_p(stream)), "epsm_film_adjoint_reparam")
        return dL, adj
    return film_adjoint_reparam_torch(film_pos, radiance, grad_img, accum)


def film_adjoint_reparam_torch(film_pos: torch.Tensor, radiance: torch.Tensor, grad_img: torch.Tensor, accum: torch.Tensor):
    """``film_adjoint_reparam`` as dense torch operations (the checker of the HIP kernel; the CPU path of the host harness)."""
    H, W = accum.shape[:2]
    Wp = accum[..., 3]
    ok = (Wp > 0)[..., None]
    inv = torch.where(ok, 1.0 / Wp.clamp_min(1e-30)[..., None], torch.zeros_like(accum[..., :1]))
    g = grad_img[..., :3] * inv                                     # grad / W_p
    gi = (g * (accum[..., :3] * inv)).sum(-1)                       # (grad . image) / W_p
    px, py = film_pos[:, 0], film_pos[:, 1]
    X, Y = torch.floor(px).long(), torch.floor(py).long()
    radius, alpha = 2.0, -1.0 / (2.0 * 0.5 * 0.5)
    bias = math.exp(alpha * radius * radius)
    off = torch.arange(-2, 3, device=film_pos.device)
    xs, ys = X[:, None] + off[None, :], Y[:, None] + off[None, :]
    dx, dy = (xs.float() + 0.5) - px[:, None], (ys.float() + 0.5) - py[:, None]

    def weights(d_, inside):
        e = torch.exp(alpha * d_ * d_)
        w = (e - bias).clamp_min(0)
        live = (d_.abs() <= radius) & inside & (w > 0)
        w = torch.where(live, w, torch.zeros_like(w))
        dw = torch.where(live, -2.0 * alpha * d_ * e, torch.zeros_like(w))      # d w / d pos  (d_ = pixel centre - pos)
        return w, dw
    wx, dwx = weights(dx, (xs >= 0) & (xs < W))
    wy, dwy = weights(dy, (ys >= 0) & (ys < H))
    yi, xi = ys.clamp(0, H - 1)[:, :, None], xs.clamp(0, W - 1)[:, None, :]
    gw = g[yi, xi]                                                   # (n,5,5,3)
    A = (gw * radiance[:, None, None, :]).sum(-1) - gi[yi, xi]       # (n,5,5): grad_p . (L_i - image_p) / W_p
    w2 = wy[:, :, None] * wx[:, None, :]
    dL = (gw * w2[..., None]).sum(dim=(1, 2))
    adj = torch.stack([(A * (wy[:, :, None] * dwx[:, None, :])).sum(dim=(1, 2)),
                       (A * (dwy[:, :, None] * wx[:, None, :])).sum(dim=(1, 2)), (A * w2).sum(dim=(1, 2))], dim=1)
    return dL.contiguous(), adj.contiguous()


class PRBIntegrator:
    """Second phase of the reference's ``*_hybrid`` scheme (EPSM/optim.py:87-94, 113-119 switch to ``prb_reparam`` after
    ``thres`` iterations): a 3-channel image and the COLOUR adjoint -- ``render_backward`` takes ``grad_in (H,W,3)``
    (the branch of epsm.py:230-234) and accumulates d sum(image * grad_in) / d theta for the colour parameters attached
    to the scene (``Scene.attach_color``: diffuse reflectances, ``Scene.attach_radiance``: emitters) into
    ``params.color``.  Path replay with detached sampling as in prb.py: one pass of ``epsm_trace_paths_color`` under the
    primal pass's seed returns, per path, the radiance and its derivative sums; the film's adjoint turns ``grad_in``
    into the adjoint radiance of every sample.  This class is ``prb`` (prb.py): colour parameters only, ``reparam`` is
    False and geometry receives nothing.  What ``prb_reparam`` adds on top -- the warp field that makes visibility
    differentiable, i.e. gradients of vertex positions and normals through shading, silhouettes and shadow boundaries
    (ad/reparam.py) -- is the subclass ``PRBReparamIntegrator`` below (csrc/epsm_trace_reparam.h)."""
    reparam = False

    def __init__(self, props: Optional[dict] = None):
        props = dict(props or {})
        max_depth = props.get("max_depth", 6)
        if max_depth < 0 and max_depth != -1:
            raise Exception("\"max_depth\" must be set to -1 (infinite) or a value >= 0")
        self.max_depth = max_depth
        self.rr_depth = props.get("rr_depth", 5)

    def _depth(self) -> int:
        """Depth of BOTH passes: ``epsm_trace_paths_color`` replays at most 6 bounces, and the image the adjoint is paired
        with must come from the same estimator (render() went to the full max_depth, the replay to 6)."""
        return 6 if self.max_depth < 0 else min(int(self.max_depth), 6)

    def to_string(self):
        return f"PRBIntegrator[max_depth = {self.max_depth}, rr_depth = {self.rr_depth}, reparam = False]"

    __repr__ = to_string

    def render(self, scene, sensor=0, seed=0, spp=0, develop=True, evaluate=True):
        if not develop:
            raise Exception("develop=True must be specified when invoking AD integrators")
        self.primal_image = scene.render_primal(sensor=sensor, seed=seed, spp=spp, max_depth=self._depth())[..., :3]
        return self.primal_image

    def render_backward(self, scene, params: ParamGrads, grad_in: torch.Tensor, sensor=0, seed: int = 0, spp: int = 0) -> None:
        """Accumulates into ``params.color`` (one all-reduce of this call's contribution when there are several ranks)."""
        self._color_backward(scene, params, grad_in, sensor, seed, spp)

    def _color_backward(self, scene, params: ParamGrads, grad_in: torch.Tensor, sensor=0, seed: int = 0, spp: int = 0) -> None:
        if not getattr(scene, "color_slots", None):
            if self.reparam:
                return
            if getattr(scene, "has_attached_geometry", lambda: False)():
                raise NotImplementedError(
                    "prb: geometry is attached but no colour parameter is -- `prb` differentiates colours only (prb.py); the "
                    "gradients of vertex positions through visibility are what `prb_reparam` (its warp field: "
                    "csrc/epsm_trace_reparam.h) or the manifold integrators compute")
            return                                  # nothing attached that this phase differentiates
        si = min(sensor, len(scene.sensors) - 1)
        s = scene.sensors[si]
        spp = spp or s.spp
        n_total = s.wavefront_size(spp)
        rank, world = _dist.world()
        import ctypes as C
        from . import _lib
        lib = scene._backend if scene._backend is not None else _lib.lib()
        stream = torch.cuda.current_stream(scene.device).cuda_stream if scene.device.type == "cuda" else None
        accum = torch.zeros((s.height, s.width, 4), device=scene.device, dtype=torch.float32)
        tiles = _dist.tile_ranges(n_total, scene.tile_paths)
        kept = []
        for t in _dist.my_tiles(len(tiles), rank, world):
            lo, hi = tiles[t]
            film_pos, radiance, sums = scene.trace_color(si, seed, spp, self._depth(), lo, hi)
            rc = lib.epsm_film_splat(C.c_int64(hi - lo), C.c_void_p(film_pos.data_ptr()), C.c_void_p(radiance.data_ptr()),
                                     s.width, s.height, s.rfilter, C.c_void_p(accum.data_ptr()), C.c_void_p(stream))
            assert rc == 0, "epsm_film_splat failed"
            kept.append((film_pos, sums))
        if world > 1:
            _dist.allreduce_param_grads(accum)
        g = grad_in.to(scene.device, torch.float32)[: s.height, : s.width, :3]
        values = scene.color_values()                                   # (C,3)
        contrib = torch.zeros_like(values)
        for film_pos, sums in kept:
            dL = film_adjoint(film_pos, g, accum[..., 3], s.rfilter)    # (n,3)
            contrib += (sums * dL[:, None, :]).sum(dim=0)
        contrib = contrib / values.clamp_min(1e-12)
        if world > 1:
            _dist.allreduce_param_grads(contrib)
        params.color += contrib


class PRBReparamIntegrator(PRBIntegrator):
    """``prb_reparam`` (src/python/python/ad/integrators/prb_reparam.py): path replay with detached sampling PLUS the
    reparameterisation of Bangaru et al. that makes visibility differentiable -- ``render_backward`` accumulates
    d sum(image * grad_in) / d vertex positions (and vertex normals) of the attached meshes into ``params.pos`` /
    ``params.nrm`` through silhouettes, shadow boundaries and shading, and the colour adjoint of ``PRBIntegrator`` into
    ``params.color``.  Properties as in prb_reparam.py:226-250: ``reparam_max_depth`` (default: max_depth),
    ``reparam_rays`` (16; at most 64 here), ``reparam_kappa`` (1e5), ``reparam_exp`` (3.0), ``reparam_antithetic`` (False:
    auxiliary rays in mirrored pairs, reparam.py:82-84, 189-196).  The pass replays the estimator of ``Scene.render_primal`` under the
    same seed (csrc/epsm_trace_reparam.h); a box reconstruction filter is refused as in common.py:379-388."""
    reparam = True

    def __init__(self, props: Optional[dict] = None):
        super().__init__(props)
        props = dict(props or {})
        self.reparam_max_depth = props.get("reparam_max_depth", self._depth())
        self.reparam_rays = int(props.get("reparam_rays", 16))
        self.reparam_kappa = float(props.get("reparam_kappa", 1e5))
        self.reparam_exp = float(props.get("reparam_exp", 3.0))
        self.reparam_antithetic = bool(props.get("reparam_antithetic", False))
        if not 1 <= self.reparam_rays <= 64:
            raise ValueError("prb_reparam: 1 <= reparam_rays <= 64")

    def to_string(self):
        return (f"PRBReparamIntegrator[max_depth = {self.max_depth}, rr_depth = {self.rr_depth}, reparam_max_depth = "
                f"{self.reparam_max_depth}, reparam_rays = {self.reparam_rays}]")

    __repr__ = to_string

    def render_backward(self, scene, params: ParamGrads, grad_in: torch.Tensor, sensor=0, seed: int = 0, spp: int = 0) -> None:
        self._color_backward(scene, params, grad_in, sensor, seed, spp)
        cam = bool(getattr(scene, "sensor_attached", False))
        if not scene.has_attached_geometry() and not cam:
            return
        if cam:
            # Moving the sensor by t moves every shape (and every emitter that is one) by -t as the sensor sees it; environment
            # emitters do not care.  So d loss / d sensor position = - sum over ALL vertices of d loss / d vertex position: the
            # same reparameterised pass with every mesh attached, summed (the primary rays' warp field carries the silhouettes).
            if any(e["type"] == 1 for e in scene.emitter_desc):
                raise NotImplementedError("prb_reparam: a `point` emitter's position would have to move with the shapes for the "
                                          "sensor's gradient; scenes with point emitters keep the sensor fixed")
            was = [(m, bool(getattr(m, "pos_attached", False))) for m in scene.meshes]
            for m, _ in was:
                m.pos_attached = True
            scene._refresh_attach_flags(sync_host=False)
            try:
                full = ParamGrads(params.V, params.B, device=params.flat.device, mesh_slices=params.mesh_slices, n_colors=params.C)
                self._geometry_backward(scene, full, grad_in, sensor, seed, spp)
            finally:
                for m, a in was:
                    m.pos_attached = a
                scene._refresh_attach_flags(sync_host=False)
            params.cam_origin -= full.pos.sum(dim=0)
            for m, a in was:                                   # the meshes the caller attached keep their own rows
                if a:
                    lo, hi = params.mesh_slices[m.name]
                    params.pos[lo:hi] += full.pos[lo:hi]
            params.nrm += full.nrm
            return
        self._geometry_backward(scene, params, grad_in, sensor, seed, spp)

    def _geometry_backward(self, scene, params: ParamGrads, grad_in: torch.Tensor, sensor=0, seed: int = 0, spp: int = 0) -> None:
        si = min(sensor, len(scene.sensors) - 1)
        s = scene.sensors[si]
        if s.rfilter == 0:
            raise Exception("ADIntegrator detected the potential for image-space motion due to differentiable shape or camera "
                            "pose parameters. This is, however, incompatible with the box reconstruction filter that is "
                            "currently used. Please specify a smooth reconstruction filter in your scene description (e.g. "
                            "'gaussian', which is actually the default)")          # common.py:379-388
        spp = spp or s.spp
        n_total = s.wavefront_size(spp)
        rank, world = _dist.world()
        import ctypes as C
        from . import _lib
        lib = scene._backend if scene._backend is not None else _lib.lib()
        stream = torch.cuda.current_stream(scene.device).cuda_stream if scene.device.type == "cuda" else None
        # pass 1 (common.py:872-882): the primal estimate of every sample and the film they make
        accum = torch.zeros((s.height, s.width, 4), device=scene.device, dtype=torch.float32)
        # tiles as large as the sharding allows, up to 2^23 paths (7 GB of warp requests): the later stages of a tile carry
        # a fraction of its paths and under-fill the chip in 2^20-path tiles (4.26 M paths: 62 ms in five tiles, 57 ms in one)
        # ... unless the caller set Scene.tile_paths: that is their memory bound and stays an upper bound
        per_rank = -(-n_total // world)
        tile = min(1 << 23, int(scene.tile_paths)) if getattr(scene, "tile_paths_explicit", False) else \
            min(1 << 23, max(int(scene.tile_paths), per_rank))
        tiles = _dist.tile_ranges(n_total, tile)
        mine = list(_dist.my_tiles(len(tiles), rank, world))
        kept = {}
        for t in mine:
            lo, hi = tiles[t]
            with _prof.phase("epsm.prb_reparam.primal_pass"):
                tr = scene._trace(si, seed, spp, self._depth(), 0, lo, hi)
                rc = lib.epsm_film_splat(C.c_int64(hi - lo), C.c_void_p(tr.film_pos.data_ptr()), C.c_void_p(tr.radiance.data_ptr()),
                                         s.width, s.height, s.rfilter, C.c_void_p(accum.data_ptr()), C.c_void_p(stream))
            assert rc == 0, "epsm_film_splat failed"
            kept[t] = (tr.film_pos, tr.radiance.contiguous())
        if world > 1:
            _dist.allreduce_param_grads(accum)
        # pass 2 (common.py:944-955): adjoint radiance, adjoint film position / determinant, the reparameterised replay
        g = grad_in.to(scene.device, torch.float32)[: s.height, : s.width, :3]
        out = params.scratch() if world > 1 else params
        for t in mine:
            lo, hi = tiles[t]
            film_pos, radiance = kept.pop(t)
            with _prof.phase("epsm.prb_reparam.film_adjoint"):
                dL, adj = film_adjoint_reparam(film_pos, radiance, g, accum)
            with _prof.phase("epsm.prb_reparam.reparam_pass"):
                scene.trace_reparam(si, seed, spp, self._depth(), lo, hi, radiance, dL, adj, out.pos, out.nrm,
                                    int(self.reparam_max_depth), self.reparam_rays, self.reparam_kappa, self.reparam_exp,
                                    antithetic=self.reparam_antithetic)
        if world > 1:
            _dist.allreduce_param_grads(out.flat)
            params.flat += out.flat


# -- plugin registry (mi.register_integrator / mi.load_dict) -------------------
_REGISTRY: Dict[str, Callable[[dict], EPSMIntegrator]] = {}


def register_integrator(name: str, constructor: Callable[[dict], EPSMIntegrator]) -> None:
    _REGISTRY[name] = constructor


def load_dict(d: dict) -> EPSMIntegrator:
    d = dict(d)
    kind = d.pop("type")
    if kind not in _REGISTRY:
        raise RuntimeError(f"Plugin \"{kind}\" not found")   # PluginManager error text, plugin.cpp:24-38
    return _REGISTRY[kind](d)


register_integrator("manifold", lambda props: ManifoldIntegrator(props))                  # epsm.py:948
register_integrator("manifold_caustic", lambda props: ManifoldCausticIntegrator(props))   # epsm.py:1202
register_integrator("prb", lambda props: PRBIntegrator(props))
register_integrator("prb_reparam", lambda props: PRBReparamIntegrator(props))    # EPSM/optim.py:89-92 asks for this name
