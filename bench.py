#!/usr/bin/env python3
"""Benchmark of the EPSM manifold-gradient hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (default): the headline of BASELINE.json -- ONE gradient image of the bathroom
configuration, ``manifold``, 1024 x 1024 @ 256 spp = 268 435 456 paths with 5 logged
vertices each (``configs[3]`` without its 8-GPU suffix) -- as 16 DISTINCT
slabs of 2^24 paths (64 image rows each), every slab seeded by its index and resident in
HBM before the timed region.  A "step" is one backward pass over that image: zero the
parameter-gradient buffer -> per slab ONE launch of ``epsm_backward_pass`` (first-vertex
tangent + per-path constraint Jacobian + block solve + adjoint gradients + scatter into
the parameter-gradient buffer) -> one RCCL all-reduce of the buffer when N > 1.

``--gpus N`` (N > 1): strong scaling of the same image -- slab s belongs to rank s % N,
one process per GPU.  Started either by ``torch.distributed.run`` (RANK / WORLD_SIZE /
LOCAL_RANK in the environment) or directly: then THIS process, before it touches the GPU,
starts N children with those variables set, relays rank 0's JSON line and fails if any
child fails.

``--config n`` (1..5) runs BASELINE.json ``configs[n-1]`` instead (``--config 2`` is the
512 x 512 @ 64 spp wavefront of round 1's line); ``--two-stage`` / ``--separate-tangent``
time the reference-shaped pipelines.

Prints ONE JSON line (rank 0).  ``value`` = paths/s over all ranks for the whole step;
``grad_image_ms`` = wall-clock of the step = one gradient image; ``roofline`` prices the
dominant kernel (one slab's ``epsm_backward_cp_kernel`` launch) against the 8 TB/s HBM
peak with the ALGORITHMIC bytes of SURVEY.md 8d (56 + 116 K per path in one launch,
32 + 116 K fused, 32 + 200 K stand-alone) over its own average launch time, measured with
HIP events on the launch stream around every launch of the timed region; next to it
``frac_survey`` (SURVEY 8d's 32 + 116 K), ``live_algorithmic_bytes`` / ``frac_live`` (only the vertices a path's terms reach, from
the flag words) and ``valu_floor_ms`` (the kernel's vector instructions at one per four
clocks and SIMD, from the committed counter run), so that the line says which bound applies;
``cpu_baseline`` times oracle/ (the C restatement: tangent + calc_grad + scatter) on this
box's host cores on a bounded sample of the same records.  Secondary keys, all outside the
timed region (default run only): ``configs_1`` (BASELINE.json configs[1]), ``real_scene``
(render_backward on records the library's own tracer produces: trace + native log + backward
pass; its rate is repeated at top level as ``end_to_end_paths_per_s``) and ``hybrid_phase2`` (prb_reparam's render_backward on the same traced scene: the
reparameterised pass of the hybrid scheme's second phase).
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
SLAB_PATHS = 1 << 24           # paths per resident slab (64 rows of a 1024-wide film at 256 spp)


def algorithmic_bytes_per_path(K: int) -> int:
    # 32 B per path + 116 B in / 84 B out per logged vertex
    return 32 + 200 * K


CONFIGS = {
    # n: (label, variant, profile, res, spp, K, V)
    0: ("BASELINE.json metric: bathroom @ 256 spp, 1024x1024 (configs[3] on the given number of GPUs)", "manifold", "bathroom", 1024, 256, 5, 100000),
    1: ("configs[0]: single glass-sphere caustic, manifold_caustic, 64x64 @ 4 spp", "manifold_caustic", "caustic", 64, 4, 4, 7829),
    2: ("configs[1]: bathroom, manifold, 512x512 @ 64 spp", "manifold", "bathroom", 512, 64, 5, 100000),
    3: ("configs[2]: pool caustic, manifold_caustic, 1024x1024 @ 256 spp", "manifold_caustic", "pool", 1024, 256, 5, 100000),
    4: ("configs[3]: bathroom, manifold (hybrid phase 1), 1024x1024 @ 256 spp sharded over the ranks", "manifold", "bathroom", 1024, 256, 5, 100000),
    5: ("configs[4]: human, manifold, 256x256 @ 8 spp, K=2, 7829-vertex mesh", "manifold", "bathroom", 256, 8, 2, 7829),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=0, choices=sorted(CONFIGS),
                    help="0: the headline (default); n: BASELINE.json configs[n-1]")
    ap.add_argument("--res", type=int, default=None, help="override the preset's film side")
    ap.add_argument("--spp", type=int, default=None, help="override the preset's samples per pixel")
    ap.add_argument("--vertices", type=int, default=None, help="override: logged vertices per path (epsm.py:648)")
    ap.add_argument("--variant", default=None, choices=["manifold", "manifold_caustic"])
    ap.add_argument("--profile", default=None)
    ap.add_argument("--scene-vertices", type=int, default=None, help="override: size V of the scatter target")
    ap.add_argument("--separate-tangent", action="store_true",
                    help="tangent kernel + fused gradient/scatter kernel instead of the single epsm_backward_pass launch")
    ap.add_argument("--soa", action="store_true",
                    help="records in the reference's tensor layout (one (N,3) array per field) instead of the native packed "
                         "log (one 128-byte record per path vertex)")
    ap.add_argument("--log-layout", default=None, choices=["interleaved", "dense"],
                    help="where the native log's rays and records lie (records.alloc_log): two dense arrays (the default) or one "
                         "block of K + 1 cache lines per path (ABI v7; A/B measurements, MEASUREMENTS.md 10.12)")
    ap.add_argument("--two-stage", action="store_true",
                    help="calc_grad lists + separate scatter (the reference's shape) instead of the fused kernel")
    ap.add_argument("--max-resident-gb", type=float, default=0.0,
                    help="cap on the HBM used for resident slabs (0: 85 %% of what is free); when the rank's slabs do not "
                         "fit, the resident ones are re-used round-robin and the JSON line says so")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--real-scene", action="store_true",
                    help="add the secondary leg that traces a real scene (128 k triangles) and runs the backward pass on its records "
                         "(part of the default headline line; this flag adds it to the other configurations)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher + process group + one all-reduce of the parameter-gradient buffer only (gloo on a box "
                         "without GPU): what tests/test_bench_launcher.py runs")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the configs[1] leg after the timed region (the profiling runs want the headline launches only)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="with --gpus N: all N ranks on GPU 0 and the collectives by gloo through the host -- runs the "
                         "sharded path (launcher, slab s -> rank s mod N, scratch buffers, one all-reduce per step) on a "
                         "one-GPU box; the line it prints says so and is not a measurement")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torch.distributed.run
# ---------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def launch_ranks(n: int, argv) -> int:
    """Starts n fresh processes of this script (one per GPU) BEFORE anything here touched the GPU -- a process that
    has initialised HIP must neither fork nor exec.  Rank 0's stdout is relayed; the first failing child ends the
    others (by PID) and makes this process fail."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, rc = b"", 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            p = procs[r]
            try:
                if r == 0:
                    o, _ = p.communicate(timeout=0.5)
                    out0 += o or b""
                else:
                    p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            pending.discard(r)
            if p.returncode != 0:
                rc = p.returncode or 1
                print(f"bench.py: rank {r} exited with code {p.returncode}", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
        if rc:
            for q in list(pending):
                try:
                    procs[q].wait(timeout=30)
                except subprocess.TimeoutExpired:
                    procs[q].kill()
            break
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle's tangent + calc_grad + scatter on the host cores
# ---------------------------------------------------------------------------------------------------------
def cpu_baseline(trace, grad_in, variant, V, B, target_s):
    """Oracle (kind 'port') on every host core, bounded sample: the first n paths of slab 0, all three stages of
    the backward pass (epsm.py:238-297)."""
    from oracle import binding
    from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter, VARIANTS, num_param_grads
    binding.build()
    N = trace.ray_d.shape[0]
    g_cpu = grad_in.detach().cpu().contiguous()

    def host(v, n):
        if isinstance(v, (list, tuple)):
            return [host(x, n) for x in v]
        return v[:n].cpu() if isinstance(v, torch.Tensor) else v

    def timed(n):
        pi = [{k: host(v, n) for k, v in rec.items()} for rec in trace.path_info]
        si = [{k: host(v, n) for k, v in rec.items()} for rec in trace.scatter_info]
        rays = [t[:n].cpu().contiguous() for t in (trace.ray_o, trace.ray_d, trace.ray_dx, trace.ray_dy)]
        rec = PackedRecords(pi, device="cpu")
        sc = PackedScatter(si, device="cpu")
        K = rec.K
        P = num_param_grads(variant, K)
        first = pi[1]
        q = [first["points"][j].contiguous() for j in range(3)]
        act = first["active"].to(torch.uint8).contiguous()
        dlduv64 = torch.empty((n, 2), dtype=torch.float64); dldp64 = torch.empty((n, 3), dtype=torch.float64)
        go = torch.zeros(3, dtype=torch.float64)
        op = torch.empty((P, n, 3)); ol = torch.empty((K, n, 3)); od = torch.empty((K, n, 3))
        gp = torch.zeros((V, 3), dtype=torch.float64); gn = torch.zeros((V, 3), dtype=torch.float64)
        ga = torch.zeros((max(B, 1),), dtype=torch.float64)
        lib = binding._aux()
        t0 = time.perf_counter()
        rc = lib.epsm_oracle_first_vertex_tangent(
            n, int(trace.path_offset), int(trace.spp), int(trace.res), rays[0].data_ptr(), rays[1].data_ptr(),
            rays[2].data_ptr(), rays[3].data_ptr(), g_cpu.data_ptr(), int(g_cpu.shape[1]), int(g_cpu.shape[2]),
            q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), act.data_ptr(), dlduv64.data_ptr(), 2, dldp64.data_ptr(),
            go.data_ptr())
        assert rc == 0
        d2, p3 = dlduv64.float(), dldp64.float()
        cores = lib.epsm_oracle_calc_grad_f32(VARIANTS[variant], n, K, rec.cam.data_ptr(), C.addressof(rec.records),
                                              d2.data_ptr(), 2, 2, p3.data_ptr(), 0.1, op.data_ptr(), ol.data_ptr(),
                                              od.data_ptr(), 0)
        assert cores > 0
        op64, ol64, od64 = op.double(), ol.double(), od.double()
        rc = lib.epsm_oracle_scatter(VARIANTS[variant], n, K, C.addressof(rec.records), C.addressof(sc.records),
                                     sc.table_ptr(), sc.T, op64.data_ptr(), ol64.data_ptr(), od64.data_ptr(),
                                     gp.data_ptr(), gn.data_ptr(), ga.data_ptr(), V, B)
        assert rc == 0
        return time.perf_counter() - t0, cores

    # the first, small run pays the start of the thread team and under-states the rate: size the sample in two steps
    n1 = min(N, 1 << 16)
    dt1, cores = timed(n1)
    for _ in range(2):
        if dt1 >= 0.5 * target_s or n1 >= min(N, 1 << 23):
            break
        n1 = int(min(N, max(n1, n1 / dt1 * target_s), 1 << 23))
        dt1, cores = timed(n1)
    return {"value": n1 / dt1, "unit": "paths/s", "cores": int(cores), "kind": "port",
            "sample": f"tangent + calc_grad + scatter (epsm.py:238-297) on the first {n1} paths of slab 0 of the same "
                      f"workload, oracle/epsm_oracle.c (fp32) + oracle/epsm_oracle_aux.c (fp64), OpenMP on {int(cores)} "
                      f"threads, {dt1:.2f} s",
            # the only number from the reference's OWN code (it cannot travel to the GPU box): BASELINE.md section 2
            "reference_code": {"value": 4.8e4, "unit": "paths/s", "cores": 8,
                               "what": "the reference's calc_grad (torch, CPU) imported in place at survey time, manifold, "
                                       "N = 262 144, K = 5, 8 threads of the build container (BASELINE.md section 2): calc_grad "
                                       "only, not measured in this run"}}


def secondary_config_leg(index, dev, profile_override=None, label_override=None):
    """BASELINE.json configs[index - 1] as a secondary figure next to the headline (round 1's driver line
    was configs[1]): one resident slab, native packed log, the one-launch backward pass timed with events over 10 launches.
    ``profile_override``: the same slab shape with another synthetic profile (``dense_specular``: every vertex of every path
    live, i.e. algorithmic bytes = live bytes and 50 parameter rows per path -- VERDICT r4 item 4)."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.records import PackedLog
    label, variant, profile, res, spp, K, V = CONFIGS[index]
    profile = profile_override or profile
    label = label_override or label
    n = min(res * res * spp, SLAB_PATHS)
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=4, profile=profile, device=dev, tile_paths=n)
    integ = epsm.load_dict({"type": variant, "max_depth": 8})
    trace = scene.tile(0, 0, n, seed=0, spp=spp, K=K, lean=True)
    log = PackedLog.from_trace(trace, device=dev, table=scene.triangle_table(), free=True)
    trace = epsm.PathTrace(res=trace.res, spp=trace.spp, ray_o=None, ray_d=None, ray_dx=None, ray_dy=None, path_info=None,
                           scatter_info=None, path_offset=trace.path_offset, n_paths_total=trace.n_paths_total)
    g = torch.Generator(device=dev).manual_seed(1)
    grad_in = torch.randn((res, res, 5), generator=g, device=dev) * 1e-3
    params = epsm.ParamGrads(V, 4, device=dev)
    for _ in range(2):
        integ.backward_from_trace(trace, params, grad_in, packed=log)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        integ.backward_from_trace(trace, params, grad_in, packed=log)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    alg = (56 + 116 * K) * n
    live = int(live_vertices(log.flags, K, variant).sum())
    out = {"workload": f"{label}: one slab of {n} paths, native packed log", "kernel_ms": ms, "paths_per_s": n / (ms * 1e-3),
           "roofline_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_path": 56 + 116 * K,
           "live_vertices_per_path": live / n,
           "frac_live": (56 * n + 116 * live) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "note": "secondary figure, outside the timed region"}
    # (a counter run exists per (kernel, slab shape, profile): the headline's entry is NOT config 2's -- another film, another coherence)
    traffic, src, _ = lookup_traffic("epsm_backward_pass_packed", n, K, variant, profile) if profile_override else (None, None, None)
    if traffic is not None:
        out["traffic"] = traffic
        out["frac_traffic"] = traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        out["traffic_source"] = src
    return out


def hybrid_phase2_leg(res, spp, rays, dev):
    """Secondary figure, outside the timed region: the hybrid scheme's second phase (EPSM/optim.py:113-119) on the same
    traced scene -- prb_reparam's render_backward (primal replay, film adjoints, the reparameterised pass with its
    auxiliary rays: csrc/epsm_trace_reparam.hip), wall-clock, median of 3."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.exp import clutter
    from epsm_mitsuba3_amd.scene import Scene
    d = clutter.scene_dict(100, res, spp)
    d["sensor0"]["film"]["sample_border"] = True                       # as the reference's sensor 0 (exp/shadow.py:38)
    scene = Scene.from_dict(d, device=dev)
    for i in range(0, 100, 10):
        scene.attach(f"s{i}", positions=True, normals=True)
    integ = epsm.load_dict({"type": "prb_reparam", "max_depth": 3, "reparam_rays": rays})
    params = scene.param_grads()
    g = torch.Generator(device=dev).manual_seed(3)
    grad_in = torch.randn((res, res, 3), generator=g, device=dev) * 1e-2
    fn = lambda: integ.render_backward(scene, params, grad_in, sensor=0, seed=1, spp=spp)
    fn(); out = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        out.append((time.perf_counter() - t) * 1e3)
    ms = sorted(out)[1]
    n = scene.sensors[0].wavefront_size(spp)
    return {"scene": f"exp/clutter.py, {scene.T} triangles, every tenth sphere attached", "integrator": "prb_reparam", "paths": n,
            "max_depth": 3, "reparam_rays": rays, "render_backward_ms": ms, "paths_per_s": n / (ms * 1e-3),
            "note": "gradients of vertex positions / normals through visibility (warp field); secondary figure, outside the timed region"}


def real_scene_leg(variant, res, spp, dev):
    """Secondary figure, outside the timed region: gradient image of a TRACED scene (epsm_mitsuba3_amd/exp/clutter.py,
    the stand-in for the bathroom asset the reference does not ship): render_backward = native tracer with vertex log
    -> tangent + calc_grad + scatter, wall-clock, median of 3."""
    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd.exp import clutter
    scene = clutter.load_scene(dev, n_spheres=100, res=res, spp=spp)
    for i in range(0, 100, 3):
        scene.attach(f"s{i}", positions=True, normals=True)
    integ = epsm.load_dict({"type": variant, "max_depth": clutter.max_depth})
    integ.backward_spp = spp
    params = scene.param_grads()
    g = torch.Generator(device=dev).manual_seed(2)
    grad_in = torch.randn((res, res, 5), generator=g, device=dev) * 1e-3

    def timed(fn, n=3):
        fn(); out = []
        for _ in range(n):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
            out.append((time.perf_counter() - t) * 1e3)
        return sorted(out)[n // 2]
    total = timed(lambda: integ.render_backward(scene, params, grad_in, seed=1))

    # as render_backward traces: native log, gradient-dead paths retired (EPSM_TRACE_GRADIENT_ONLY, include/epsm_trace.h)
    # ... and with the first hit's share of the backward pass done by the stage that shades it (EPSM_TRACE_FUSE_FIRST_HIT: the paths
    # without a chain -- 94 % of this scene's -- are neither logged nor read again); `trace_and_log_ms` includes that work
    kw = dict(sensor=2, seed=1, spp=spp, max_depth=clutter.max_depth, sparse_log=True, packed_log=True, gradient_only=variant,
              first_hit=(grad_in, params, integ.outlier_clip, True) if integ.fuse_first_hit else None)

    def trace_only(**over):
        for tr in scene.iter_traces(**dict(kw, **over)):
            del tr
    trace = timed(trace_only)
    trace_full = timed(lambda: trace_only(gradient_only=None))          # the full trace (what `render` and round 4 ran)
    queues = queues_full = None
    n_tile = min(res * res * spp, scene.WAVEFRONT_TILE_PATHS)          # paths per launch
    if scene.use_wavefront(n_tile):
        # the per-bounce queue lengths (device counters): with the three stages kept for every bounce -- the product hands the last
        # few paths to one launch (EPSM_TRACE_NO_TAIL, include/epsm_trace.h) and then counts nothing behind that point
        scene.wavefront_tail = False
        trace_only(); torch.cuda.synchronize(); queues = scene.wavefront_queue_lengths()
        trace_only(gradient_only=None); torch.cuda.synchronize(); queues_full = scene.wavefront_queue_lengths()
        scene.wavefront_tail = True
    tiles = list(scene.iter_traces(**kw))
    tiles_fused = all(t.log.first_hit_done for t in tiles)

    def backward_only():
        for tr in tiles:
            integ.backward_from_trace(tr, params, grad_in)
    back = timed(backward_only)
    del tiles
    n = res * res * spp
    return {"scene": f"exp/clutter.py: floor + 100 tessellated spheres + area light = {scene.T} triangles", "variant": variant,
            "paths": n, "max_depth": clutter.max_depth, "tracer": "wavefront" if scene.use_wavefront(n_tile) else "one launch",
            "grad_image_ms": total, "trace_and_log_ms": trace, "backward_ms": back, "paths_per_s": n / (total * 1e-3),
            "full_trace_and_log_ms": trace_full, "first_hit_fusion": bool(tiles_fused),
            "paths_alive_into_bounce": {"gradient_only": queues["alive"][1:clutter.max_depth] if queues else None,
                                        "full": queues_full["alive"][1:clutter.max_depth] if queues_full else None},
            "visibility_rays_per_bounce": {"gradient_only": queues["shadow"][:clutter.max_depth] if queues else None,
                                           "full": queues_full["shadow"][:clutter.max_depth] if queues_full else None},
            "note": "render_backward on traced records (trace + native vertex log -> one launch of tangent + calc_grad + scatter "
                    "per tile), wall-clock, median of 3 each: the whole call, the trace alone, the backward pass on resident "
                    "tiles; outside the timed region"}


def live_vertices(flags: torch.Tensor, K: int, variant: str) -> torch.Tensor:
    """Per path, the number of logged vertices whose record the backward pass has to read: ``nv`` of cp::manifold_plan /
    cp::caustic_plan (csrc/epsm_cp_core.h; the term masks of epsm.py:793-802, 852-855, 916-920 / 1172-1183) restated on
    the flag words of the native log (5 bits per vertex: Diffuse, Null, active, active_em, ismesh), and at least the
    first vertex where it is active (the first-vertex tangent reads its triangle)."""
    f = flags.to(torch.int64)
    bit = lambda k, b: ((f >> (5 * (k - 1))) & b) != 0 if 1 <= k <= K else torch.zeros_like(f, dtype=torch.bool)
    valid = torch.ones_like(f, dtype=torch.bool)
    hd = torch.zeros_like(f)
    nv = torch.zeros_like(f)
    for i in range(1, K + 1):
        valid = valid & bit(i, 16)
        hd = hd + bit(i, 1).to(torch.int64)
        valid = valid & (hd < 2)
        if variant == "manifold":
            spec = valid & (hd == 0)
            nv = torch.where(spec & bit(i, 4) & bit(i, 8), torch.full_like(nv, i), nv)
            nv = torch.where(spec & bit(i + 1, 4) & bit(i + 1, 1), torch.full_like(nv, i + 1), nv)
        else:
            nv = torch.where(bit(1, 1) & valid & bit(i + 1, 4) & (bit(i + 1, 1) | bit(i + 1, 2)), torch.full_like(nv, i + 1), nv)
    return torch.maximum(nv, bit(1, 4).to(torch.int64))


def kernel_source_hash() -> str:
    """Fingerprint of the kernel sources: profiles/traffic.json entries carry the one they were measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "epsm_mitsuba3_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if (name.endswith((".hip", ".h")) and not name.startswith(("epsm_trace", "epsm_matcher"))) or name == "Makefile":   # flags count
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def lookup_traffic(kernel, paths, K, variant, profile):
    """HBM bytes per launch from the committed PMC run (profiles/traffic.json; tools/gpu_pmc_traffic.sh collects
    FETCH_SIZE / WRITE_SIZE in their own rocprofv3 --pmc passes and applies the guide's corrections).  An entry counts
    only for the kernel sources it was measured on."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.isfile(tfile):
        return None, "no profiles/traffic.json", None
    src = kernel_source_hash()
    stale = False
    for rec in json.load(open(tfile)):
        if (rec.get("kernel") == kernel and rec.get("paths") == paths and rec.get("K") == K
                and rec.get("variant") == variant and rec.get("profile") == profile):
            if rec.get("src_hash") == src:
                return rec["hbm_bytes_per_launch"], rec.get("source"), rec
            stale = True
    return None, ("stale: the kernel sources changed since the PMC run in profiles/traffic.json" if stale
                  else "no PMC run for this kernel / workload in profiles/traffic.json"), None


_T0 = time.perf_counter()


def _phase(name):
    """Where the wall-clock of a run goes (stderr; the JSON line stays alone on stdout)."""
    print(f"bench.py [{time.perf_counter() - _T0:7.1f} s] {name}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))       # nothing below has run yet: the GPU is untouched
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr)

    label, variant, profile, res, spp, K, V = CONFIGS[args.config]
    variant = args.variant or variant
    profile = args.profile or profile
    res, spp = args.res or res, args.spp or spp
    K, V = args.vertices or K, args.scene_vertices or V
    B = 4
    import torch.distributed as dist

    if args.dry_run:
        # the launcher path without a GPU: process group (gloo), one all-reduce of a parameter-gradient buffer
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from epsm_mitsuba3_amd.params import ParamGrads
        from epsm_mitsuba3_amd import dist as edist
        if world > 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        params = ParamGrads(V, B, device="cpu")
        params.flat.fill_(float(rank + 1))
        edist.allreduce_param_grads(params.flat)
        ok = bool((params.flat == world * (world + 1) / 2).all())
        ranks = dist.get_world_size() if world > 1 else 1
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "manifold_paths_per_s", "dry_run": True, "n_gpus": world, "rccl_ranks": ranks,
                              "allreduce_ok": ok, "allreduce_bytes": params.flat.numel() * 4}))
        sys.exit(0 if ok else 1)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path to measure)")
    if args.rehearse_one_gpu:
        local_rank = 0                       # every rank on GPU 0, collectives by gloo through the host: NOT a measurement
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def all_reduce_(t, op=dist.ReduceOp.SUM):
        if args.rehearse_one_gpu:
            h = t.cpu(); dist.all_reduce(h, op=op); t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd import dist as edist
    from epsm_mitsuba3_amd.records import PackedLog, PackedRecords, PackedScatter, num_param_grads
    _phase("imports done, device ready")

    N_image = res * res * spp                                   # paths of ONE gradient image (all ranks together)
    slab_paths = min(SLAB_PATHS, N_image)
    n_slabs = -(-N_image // slab_paths)
    my_slabs = list(range(rank, n_slabs, world))               # strong scaling: slab s -> rank s % world
    fused = not args.two_stage
    one_launch = fused and not args.separate_tangent
    packed_log = one_launch and not args.soa
    scene = epsm.SyntheticScene(res=res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B,
                                profile=profile, device=dev, tile_paths=slab_paths)
    integ = epsm.load_dict({"type": variant, "max_depth": 8, "fused": fused, "fuse_tangent": not args.separate_tangent})

    # -- this rank's slabs, resident in HBM (content depends only on the slab index) ------------------------
    free_b, total_b = torch.cuda.mem_get_info(dev)
    cap = args.max_resident_gb * 1e9 if args.max_resident_gb > 0 else float("inf")
    # bytes a slab keeps (its final layout) and the most its construction needs on top (the generator's per-field
    # records, before they are repacked / trimmed)
    layout = args.log_layout or "dense"
    keep_per_path = ((4 + 128 * (K + 1)) if layout == "interleaved" else (52 + 128 * K)) if packed_log else (48 + K * 139)
    build_per_path = 48 + K * 200
    slabs, used0 = [], torch.cuda.memory_allocated(dev)
    live0 = None
    for s in my_slabs:
        lo, hi = s * slab_paths, min((s + 1) * slab_paths, N_image)
        n_s = hi - lo
        torch.cuda.empty_cache()
        resident_now = torch.cuda.memory_allocated(dev) - used0
        if slabs and (torch.cuda.mem_get_info(dev)[0] < 1.1 * n_s * (keep_per_path + build_per_path) + (2 << 30)
                      or resident_now + n_s * keep_per_path > cap):
            break                                               # the rest re-uses the resident ones (reported below)
        trace = scene.tile(s, lo, hi, seed=0, spp=spp, K=K, lean=True)
        if packed_log:
            keep_soa = rank == 0 and not slabs            # slab 0 of rank 0 also feeds the CPU baseline and the dense-kernel leg
            packed = PackedLog.from_trace(trace, device=dev, table=scene.triangle_table(), free=not keep_soa, layout=layout)
            if not keep_soa:                              # only the log stays resident: the per-field arrays are released
                trace = epsm.PathTrace(res=trace.res, spp=trace.spp, ray_o=None, ray_d=None, ray_dx=None, ray_dy=None,
                                       path_info=None, scatter_info=None, path_offset=trace.path_offset,
                                       n_paths_total=trace.n_paths_total)
        else:
            packed = (PackedRecords(trace.path_info, device=dev), PackedScatter(trace.scatter_info, device=dev, table=scene.triangle_table()))
        if packed_log and not slabs:
            live0 = int(live_vertices(packed.flags, K, variant).sum())   # slab 0: vertices its paths' terms reach
        slabs.append((trace, packed))
        torch.cuda.synchronize()
    _phase(f"{len(slabs)} slab(s) resident")
    if my_slabs and not slabs:
        raise SystemExit("bench.py: not even one slab fits into HBM")
    # (a rank beyond the number of slabs -- --config 1, 2, 5 have one -- launches nothing and still enters the all-reduce)
    resident_bytes = torch.cuda.memory_allocated(dev) - used0
    n_my = len(my_slabs)
    P = num_param_grads(variant, K)
    out = None
    if args.two_stage:
        n0 = slab_paths
        out = (torch.empty((P, n0, 3), device=dev), torch.empty((K, n0, 3), device=dev), torch.empty((K, n0, 3), device=dev))
    g = torch.Generator(device=dev).manual_seed(1)
    grad_in = torch.randn((res, res, 5), generator=g, device=dev) * 1e-3
    params = epsm.ParamGrads(V, B, device=dev)

    launch_events = []          # (start, after-tangent, after-grad, after-scatter) of every launch of the timed region
    step_events = []

    def step(record=False):
        params.flat.zero_()                      # every backward pass starts from dr.grad == 0 (optim.py: per iteration)
        for j in range(n_my):
            trace, packed = slabs[j % len(slabs)]
            if record:
                evs = [torch.cuda.Event(enable_timing=True)]
                evs[0].record()

                def mark(name, evs=evs):
                    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
                integ.backward_from_trace(trace, params, grad_in, packed=packed, out=out, mark=mark)
                launch_events.append(evs)
            else:
                integ.backward_from_trace(trace, params, grad_in, packed=packed, out=out)
        if record:
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
        edist.allreduce_param_grads(params.flat)
        if record:
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            step_events.append((e0, e1))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    _phase("warm-up done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _phase("timed region done")
    rccl_ranks = 1
    # this rank's own share of a step: its launches (events on the launch stream), without the wait for the slowest rank
    own_ms = (sum(evs[0].elapsed_time(evs[-1]) for evs in launch_events) / args.steps) if launch_events else 0.0
    rank_ms_min = rank_ms_max = own_ms
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        all_reduce_(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        lo = torch.tensor([own_ms], device=dev, dtype=torch.float64); hi = lo.clone()
        all_reduce_(lo, op=dist.ReduceOp.MIN); all_reduce_(hi, op=dist.ReduceOp.MAX)
        rank_ms_min, rank_ms_max = float(lo.item()), float(hi.item())
        probe = torch.ones(1, device=dev)
        all_reduce_(probe)                                       # the number of ranks RCCL actually summed over
        rccl_ranks = int(round(float(probe.item())))
        assert rccl_ranks == dist.get_world_size() == world
    # what the last step left in the parameter-gradient buffer (already summed over the ranks): the same image whatever N
    grad_abs_sum = float(params.flat.double().abs().sum())
    ms_per_step = elapsed / args.steps * 1e3
    value = N_image * args.steps / elapsed

    names = ["tangent", "grad", "scatter"]
    stage_ms = {n: sum(evs[i].elapsed_time(evs[i + 1]) for evs in launch_events) / max(1, len(launch_events))
                for i, n in enumerate(names)}
    stage_ms["allreduce"] = sum(a.elapsed_time(b) for a, b in step_events) / len(step_events)

    # secondary figure, outside the timed region: the stand-alone gradient kernel (calc_grad's
    # dense lists, 32+200K B/path) -- the reference-shaped first stage of --two-stage
    dense_ms = None
    n_slab0 = min(slab_paths, N_image)
    if rank == 0 and not args.two_stage and torch.cuda.mem_get_info(dev)[0] > 1.3 * (P + 2 * K) * n_slab0 * 12:
        from epsm_mitsuba3_amd.manifold_grad import manifold_grad_packed
        d2 = torch.randn((n_slab0, 2), generator=g, device=dev) * 1e-3
        p3 = torch.randn((n_slab0, 3), generator=g, device=dev) * 1e-3
        dout = (torch.empty((P, n_slab0, 3), device=dev), torch.empty((K, n_slab0, 3), device=dev), torch.empty((K, n_slab0, 3), device=dev))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rec0 = PackedRecords(slabs[0][0].path_info, device=dev) if packed_log else slabs[0][1][0]
        manifold_grad_packed(variant, rec0, d2, p3, dlduv_cols=2, out=dout)
        e0.record()
        for _ in range(5):
            manifold_grad_packed(variant, rec0, d2, p3, dlduv_cols=2, out=dout)
        e1.record()
        torch.cuda.synchronize()
        dense_ms = e0.elapsed_time(e1) / 5
        del dout, d2, p3

    result = None
    if rank == 0:
        # 32+116K B/path when the per-path gradients are never written (fused),
        # 32+200K B/path for the stand-alone gradient kernel;
        # one launch (epsm_backward_pass): rays 48 B + image-gradient 8 B per path instead of cam 12 + dlduv 8 + dldp 12
        per_path = ((56 if one_launch else 32) + 116 * K) if fused else algorithmic_bytes_per_path(K)
        alg = per_path * n_slab0
        kernel_ms = stage_ms["grad"]
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        kname = ("epsm_backward_pass_packed" if packed_log else "epsm_backward_pass") if one_launch else ("epsm_grad_scatter_kernel" if fused else "epsm_grad_kernel")
        traffic, traffic_src, traffic_rec = lookup_traffic(kname, n_slab0, K, variant, profile)
        live_alg = (56 * n_slab0 + 116 * live0) if (one_launch and live0 is not None) else None
        valu = (traffic_rec or {}).get("sq_insts_valu")
        distinct = len(slabs) == n_my
        result = {
            "metric": "manifold_paths_per_s", "value": value, "unit": "paths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "grad_image_ms": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "rccl_ranks": rccl_ranks, "grad_abs_sum": grad_abs_sum,
            "rank_launch_ms_per_step": {"min": rank_ms_min, "max": rank_ms_max,
                                        "note": "each rank's own launches per step (HIP events), min / max over the ranks: load imbalance"},
            **({"rehearsal": f"{world} ranks SHARING GPU 0, collectives by gloo through the host: exercises the sharded path, "
                             "measures nothing"} if args.rehearse_one_gpu else {}),
            "config": {"workload": f"{label}: {profile}-like synthetic path records, {variant}, {res}x{res} @ {spp} spp = "
                                   f"{N_image} paths per gradient image, K={K} logged vertices, scatter target V={V} "
                                   f"vertices; {n_slabs} slab(s) of {slab_paths} paths, slab s on rank s % {world}",
                       "variant": variant, "profile": profile, "paths_per_image": N_image, "slabs": n_slabs,
                       "slabs_per_gpu": n_my, "slabs_resident_per_gpu": len(slabs), "all_slabs_distinct": distinct,
                       "resident_bytes_per_gpu": int(resident_bytes), "hbm_total_bytes": int(total_b),
                       "vertices": K, "scene_vertices": V,
                       "sharding": f"{world} rank(s), one all-reduce of the {params.flat.numel() * 4} B "
                                   f"parameter-gradient buffer per step"},
            "stages_ms": stage_ms,
            "pipeline": (("one launch per slab (epsm_backward_pass_packed, native log)" if packed_log else
                          "one launch per slab (epsm_backward_pass)") if one_launch else "tangent + fused") if fused else "two-stage",
            "record_layout": (("packed, interleaved: one block of K + 1 cache lines per path = [rays | 16 B free | K records of 128 B], "
                               "+ flag word (include/epsm.h, EpsmPackedLog)") if layout == "interleaved" else
                              "packed, dense: one 128-byte record per (path, vertex) + rays (N,12) + flag word") if packed_log
                             else "reference tensors: one (N,3) array per field",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # SURVEY.md 8(d)'s own figure for a fused pass, 32 + 116 K B/path (cam + dlduv + dldp = 32 instead of the
                         # rays + image gradient = 56 this launch reads because it computes the tangent itself)
                         "frac_survey": ((32 + 116 * K) * n_slab0 / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if fused else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         # what the memory system sees: the counters' HBM bytes per launch over the same launch time (VERDICT r4:
                         # `frac` prices SURVEY's algorithmic bytes, part of which the kernel never has to move)
                         "frac_traffic": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "kernel": ("epsm_backward_cp_kernel<tangents in kernel> (epsm_backward_pass: tangent + calc_grad + scatter)"
                                    if one_launch else "epsm_backward_cp_kernel (fused calc_grad + scatter)") if fused else "epsm_grad_kernel",
                         "kernel_ms": kernel_ms, "launches_timed": len(launch_events), "paths_per_launch": n_slab0,
                         "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_path": per_path,
                         # SURVEY 8(d)'s figure counts all K vertices of every path; a path's terms stop at its first
                         # diffuse / inactive vertex: what the kernel has to READ is this
                         "live_algorithmic_bytes": live_alg,
                         "frac_live": (live_alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if live_alg else None,
                         # the other bound: vector instructions of one launch (SQ_INSTS_VALU of the committed counter run) at
                         # one per 4 clocks per SIMD (a wave alone on its SIMD; 1024 SIMDs, 2.4 GHz)
                         "valu_insts_per_launch": valu,
                         "valu_floor_ms": (valu / 1024 * 4 / 2.4e9 * 1e3) if valu else None,
                         "kernel_source_hash": kernel_source_hash()},
        }
        if not distinct:
            result["config"]["note"] = (f"only {len(slabs)} of this rank's {n_my} slabs fit into HBM next to each other; "
                                        f"they are processed round-robin to make up the {n_my} launches of a step")
        if dense_ms is not None:
            a2 = algorithmic_bytes_per_path(K) * n_slab0
            result["standalone_grad_kernel"] = {"kernel": "epsm_grad_kernel", "kernel_ms": dense_ms,
                                                "paths_per_s": n_slab0 / (dense_ms * 1e-3),
                                                "achieved": a2 / (dense_ms * 1e-3) / 1e9, "unit": "GB/s",
                                                "frac": a2 / (dense_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "algorithmic_bytes_per_path": a2 // n_slab0,
                                                "note": "outside the timed region; first stage of --two-stage"}
        _phase("dense-kernel leg done")
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(slabs[0][0], grad_in, variant, V, B, args.cpu_seconds)
            _phase("cpu baseline done")
        if world == 1 and args.config == 0 and not (args.soa or args.two_stage or args.separate_tangent or args.no_secondary):
            result["configs_1"] = secondary_config_leg(2, dev)          # BASELINE.json configs[1]: round 1's driver line
            # BASELINE.json configs[2]: `manifold_caustic` on deep specular chains, ONE slab of its sixteen (the slab tools/gpu_cp_ko.sh times)
            result["configs_2"] = secondary_config_leg(3, dev)
            # the workload where SURVEY's algorithmic bytes ARE the live bytes: no diffuse vertex, five constraint vertices and
            # ~50 parameter rows per path
            result["dense_specular"] = secondary_config_leg(2, dev, profile_override="specular",
                                                            label_override="dense_specular: configs[1]'s slab with the `specular` profile (every vertex live)")
        if world == 1 and (args.real_scene or (args.config == 0 and not (args.soa or args.two_stage or args.separate_tangent or args.no_secondary))):
            # end to end on records the library's own tracer produces: trace + native log + backward pass
            del slabs, out
            torch.cuda.empty_cache()
            _phase("configs_1 + dense_specular legs done")
            result["real_scene"] = real_scene_leg(variant, 512, 64, dev)
            _phase("real_scene leg done")
            # `value` counts the backward pass on RESIDENT synthetic records; with the library's own tracer producing them the
            # whole gradient image runs at this rate (first screen of the line, not a nested key)
            result["end_to_end_paths_per_s"] = result["real_scene"]["paths_per_s"]
            result["end_to_end_note"] = ("render_backward on TRACED records (real_scene: trace + native log + backward pass); "
                                         "`value` is the backward pass alone on resident synthetic records")
            torch.cuda.empty_cache()
            result["hybrid_phase2"] = hybrid_phase2_leg(256, 16, 16, dev)
            torch.cuda.empty_cache()
            # the same pass at the film of BASELINE.json configs[3] (`manifold_hybrid`, 1024 x 1024) at 16 spp
            result["hybrid_phase2_1024"] = hybrid_phase2_leg(1024, 16, 16, dev)
            _phase("hybrid_phase2 legs done")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
