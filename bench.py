#!/usr/bin/env python3
"""Benchmark of the EPSM manifold-gradient hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one backward pass of the hot path over one wavefront of synthetic path
records that are already resident in HBM: zero the parameter-gradient buffer ->
first-vertex tangent + per-path constraint Jacobian + block solve + adjoint
gradients + scatter into the parameter-gradient buffer in ONE launch
(``epsm_backward_pass``; ``--separate-tangent``: ``epsm_first_vertex_tangent`` then
``epsm_manifold_grad_scatter``; ``--two-stage``: tangent, ``epsm_manifold_grad``,
``epsm_scatter`` -- the reference's shape) -> one RCCL all-reduce of that buffer
when N > 1.  At N=1 the workload is BASELINE.json ``configs[1]``: bathroom,
``manifold``, 512x512 @ 64 spp -> 16 777 216 paths, 5 logged vertices each
(SURVEY.md 8d).  With N>1 every rank processes its own wavefront of that size
(weak scaling: pixel/sample tiles of a larger image sharded over the GPUs).

Prints ONE JSON line (rank 0).  ``value`` = paths/s over all ranks for the whole
step; ``grad_image_ms`` = wall-clock of the step; ``roofline`` prices the dominant
kernel (the fused gradient+scatter kernel) against the 8 TB/s HBM peak using the
ALGORITHMIC bytes (56 + 116*K per path in one launch, 32 + 116*K fused, 32 + 200*K
stand-alone)
and its own launch time measured with HIP events on the launch stream; ``cpu_baseline`` times oracle/ (the C restatement of
the reference's calc_grad) on this box's host cores on a bounded sample of the same
records.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes_per_path(K: int) -> int:
    # 32 B per path + 116 B in / 84 B out per logged vertex
    return 32 + 200 * K


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--res", type=int, default=512, help="image side (config 2: 512)")
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel (config 2: 64)")
    ap.add_argument("--vertices", type=int, default=5, help="logged vertices per path (epsm.py:648)")
    ap.add_argument("--variant", default="manifold", choices=["manifold", "manifold_caustic"])
    ap.add_argument("--profile", default="bathroom")
    ap.add_argument("--scene-vertices", type=int, default=100000, help="size V of the scatter target")
    ap.add_argument("--separate-tangent", action="store_true",
                    help="tangent kernel + fused gradient/scatter kernel instead of the single epsm_backward_pass launch")
    ap.add_argument("--two-stage", action="store_true",
                    help="calc_grad lists + separate scatter (the reference's shape) instead of the fused kernel")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="preset of BASELINE.json configs[n-1] (0: the flags above; the default flags ARE config 2)")
    ap.add_argument("--slabs", type=int, default=1,
                    help="wavefronts larger than one resident slab: the resident records are processed this many "
                         "times per step (configs 3, 4: 1024x1024 @ 256 spp = 16 slabs of 2^24 paths)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-real-scene", action="store_true",
                    help="skip the secondary leg that traces a real scene (128 k triangles) and runs the backward pass on its records")
    return ap.parse_args()


def cpu_baseline(path_info, variant, target_s):
    """Oracle (kind 'port') on every host core, bounded sample of the same records."""
    from oracle import binding
    from epsm_mitsuba3_amd.records import PackedRecords, VARIANTS, num_param_grads
    binding.build()
    N = path_info[0]["cam"].shape[0]
    g = torch.Generator().manual_seed(0)

    def timed(n):
        sl = slice(0, n)
        pi = [{k: ([x[sl].cpu() for x in v] if isinstance(v, (list, tuple)) else
                   (v[sl].cpu() if isinstance(v, torch.Tensor) else v)) for k, v in rec.items()} for rec in path_info]
        rec = PackedRecords(pi, device="cpu")
        K = rec.K
        d2 = (torch.randn((n, 2), generator=g) * 1e-3).contiguous()
        p = (torch.randn((n, 3), generator=g) * 1e-3).contiguous()
        P = num_param_grads(variant, K)
        op = torch.empty((P, n, 3)); ol = torch.empty((K, n, 3)); od = torch.empty((K, n, 3))
        fn = binding.lib().epsm_oracle_calc_grad_f32
        t0 = time.perf_counter()
        rc = fn(VARIANTS[variant], n, K, rec.cam.data_ptr(), C.addressof(rec.records), d2.data_ptr(),
                2, 2, p.data_ptr(), 0.1, op.data_ptr(), ol.data_ptr(), od.data_ptr(), 0)
        dt = time.perf_counter() - t0
        assert rc > 0
        return dt, rc

    n0 = min(N, 1 << 16)
    dt0, cores = timed(n0)
    n1 = int(min(N, max(n0, n0 / dt0 * target_s), 1 << 23))
    dt1, cores = timed(n1)
    return {"value": n1 / dt1, "unit": "paths/s", "cores": int(cores), "kind": "port",
            "sample": f"calc_grad only (the dominant stage), first {n1} paths of the same wavefront, "
                      f"oracle/epsm_oracle.c fp32 + OpenMP, {dt1:.2f} s"}


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
    trace = timed(lambda: scene.trace_paths(sensor=2, seed=1, spp=spp, max_depth=clutter.max_depth, sparse_log=True))
    n = res * res * spp
    return {"scene": f"exp/clutter.py: floor + 100 tessellated spheres + area light = {scene.T} triangles", "variant": variant,
            "paths": n, "max_depth": clutter.max_depth, "tracer": "wavefront" if scene.use_wavefront() else "one launch",
            "grad_image_ms": total, "trace_and_log_ms": trace, "backward_ms": total - trace, "paths_per_s": n / (total * 1e-3),
            "note": "render_backward on traced records (trace + vertex log -> tangent + calc_grad + scatter), wall-clock, "
                    "median of 3; outside the timed region"}


CONFIGS = {
    # n: (label, variant, profile, res, spp, K, V, slabs of the wavefront when it is run on ONE GPU)
    1: ("configs[0]: single glass-sphere caustic, manifold_caustic, 64x64 @ 4 spp", "manifold_caustic", "caustic", 64, 4, 4, 7829, 1),
    2: ("configs[1]: bathroom, manifold, 512x512 @ 64 spp", "manifold", "bathroom", 512, 64, 5, 100000, 1),
    3: ("configs[2]: pool caustic, manifold_caustic, 1024x1024 @ 256 spp (16 slabs of 2^24 paths)", "manifold_caustic", "pool", 512, 64, 5, 100000, 16),
    4: ("configs[3]: bathroom, manifold (hybrid phase 1), 1024x1024 @ 256 spp sharded over the ranks", "manifold", "bathroom", 512, 64, 5, 100000, 16),
    5: ("configs[4]: human, manifold, 256x256 @ 8 spp, K=2, 7829-vertex mesh", "manifold", "bathroom", 256, 8, 2, 7829, 1),
}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    label = "BASELINE.json configs[1]"
    if args.config:
        label, args.variant, args.profile, args.res, args.spp, args.vertices, args.scene_vertices, args.slabs = CONFIGS[args.config]
        if args.config == 4:
            args.slabs = max(1, args.slabs // world)          # strong scaling of ONE 1024x1024 @ 256 spp image
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path to measure)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import epsm_mitsuba3_amd as epsm
    from epsm_mitsuba3_amd import dist as edist
    from epsm_mitsuba3_amd.records import PackedRecords, PackedScatter, num_param_grads

    K, V, B = args.vertices, args.scene_vertices, 4
    N = args.res * args.res * args.spp              # paths of one gradient image, per rank
    scene = epsm.SyntheticScene(res=args.res, n_vertices=K, n_scene_vertices=V, n_bsdfs=B,
                                profile=args.profile, device=dev, tile_paths=N)
    integ = epsm.load_dict({"type": args.variant, "max_depth": 8, "fused": not args.two_stage,
                            "fuse_tangent": not args.separate_tangent})
    # this rank's wavefront: one resident tile (seeded by rank so shards differ)
    trace = scene.tile(0, 0, N, seed=rank, spp=args.spp, K=K)
    packed = (PackedRecords(trace.path_info, device=dev), PackedScatter(trace.scatter_info, device=dev))
    P = num_param_grads(args.variant, K)
    out = (torch.empty((P, N, 3), device=dev), torch.empty((K, N, 3), device=dev), torch.empty((K, N, 3), device=dev))
    g = torch.Generator(device=dev).manual_seed(1)
    grad_in = torch.randn((args.res, args.res, 5), generator=g, device=dev) * 1e-3
    params = epsm.ParamGrads(V, B, device=dev)

    stage_events = []

    def step(record=False):
        evs = [torch.cuda.Event(enable_timing=True)] if record else None

        def mark(name):
            if record:
                e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        params.flat.zero_()                      # every backward pass starts from dr.grad == 0 (optim.py: per iteration)
        for _ in range(args.slabs - 1):          # earlier slabs of a wavefront that is larger than the resident one
            integ.backward_from_trace(trace, params, grad_in, packed=packed, out=out)
        if record:
            evs[0].record()                      # per-stage times are those of the step's last slab
        integ.backward_from_trace(trace, params, grad_in, packed=packed, out=out, mark=mark)
        edist.allreduce_param_grads(params.flat)
        mark("allreduce")
        if record:
            stage_events.append(evs)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
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
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * N * args.slabs * args.steps / elapsed

    names = ["tangent", "grad", "scatter", "allreduce"]
    stage_ms = {n: sum(evs[i].elapsed_time(evs[i + 1]) for evs in stage_events) / len(stage_events)
                for i, n in enumerate(names)}

    # secondary figure, outside the timed region: the stand-alone gradient kernel (calc_grad's
    # dense lists, 32+200K B/path) -- the reference-shaped first stage of --two-stage
    dense_ms = None
    if rank == 0 and not args.two_stage:
        from epsm_mitsuba3_amd.manifold_grad import manifold_grad_packed
        d2 = torch.randn((N, 2), generator=g, device=dev) * 1e-3
        p3 = torch.randn((N, 3), generator=g, device=dev) * 1e-3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        manifold_grad_packed(args.variant, packed[0], d2, p3, dlduv_cols=2, out=out)
        e0.record()
        for _ in range(5):
            manifold_grad_packed(args.variant, packed[0], d2, p3, dlduv_cols=2, out=out)
        e1.record()
        torch.cuda.synchronize()
        dense_ms = e0.elapsed_time(e1) / 5

    result = None
    if rank == 0:
        fused = not args.two_stage
        # 32+116K B/path when the per-path gradients are never written (fused),
        # 32+200K B/path for the stand-alone gradient kernel
        one_launch = fused and not args.separate_tangent
        # one launch (epsm_backward_pass): rays 48 B + image-gradient 8 B per path instead of cam 12 + dlduv 8 + dldp 12
        alg = ((56 if one_launch else 32) + 116 * K if fused else algorithmic_bytes_per_path(K)) * N
        kernel_ms = stage_ms["grad"]
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tfile):
            for rec in json.load(open(tfile)):
                want = "epsm_backward_pass" if one_launch else ("epsm_grad_scatter_kernel" if fused else "epsm_grad_kernel")
                if (rec["kernel"] == want and rec["paths"] == N
                        and rec["K"] == K and rec["variant"] == args.variant and rec["profile"] == args.profile):
                    traffic, traffic_src = rec["hbm_bytes_per_launch"], rec["source"]
        result = {
            "metric": "manifold_paths_per_s", "value": value, "unit": "paths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "grad_image_ms": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if args.config == 4 else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.profile}-like synthetic path records, {args.variant}, "
                                   f"{args.res}x{args.res} @ {args.spp} spp = {N} resident paths/GPU x {args.slabs} slab(s) "
                                   f"per step, K={K} logged vertices ({label}); scatter target V={V} vertices",
                       "variant": args.variant, "profile": args.profile, "paths_per_gpu": N * args.slabs, "vertices": K,
                       "scene_vertices": V,
                       "sharding": f"{world} x pixel/sample-tile shard, one all-reduce of the {params.flat.numel() * 4} B "
                                   f"parameter-gradient buffer per step"},
            "stages_ms": stage_ms,
            "pipeline": ("one launch (epsm_backward_pass)" if one_launch else "tangent + fused") if fused else "two-stage",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": ("epsm_grad_scatter_kernel<tangents in kernel> (epsm_backward_pass: tangent + calc_grad + scatter)"
                                    if one_launch else "epsm_grad_scatter_kernel (fused calc_grad + scatter)") if fused else "epsm_grad_kernel",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg,
                         "algorithmic_bytes_per_path": alg // N},
        }
        if dense_ms is not None:
            a2 = algorithmic_bytes_per_path(K) * N
            result["standalone_grad_kernel"] = {"kernel": "epsm_grad_kernel", "kernel_ms": dense_ms,
                                                "paths_per_s": N / (dense_ms * 1e-3),
                                                "achieved": a2 / (dense_ms * 1e-3) / 1e9, "unit": "GB/s",
                                                "frac": a2 / (dense_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "algorithmic_bytes_per_path": a2 // N,
                                                "note": "outside the timed region; first stage of --two-stage"}
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(trace.path_info, args.variant, args.cpu_seconds)
        if not args.no_real_scene and world == 1 and args.config in (0, 2):
            del packed, out, trace
            torch.cuda.empty_cache()
            result["real_scene"] = real_scene_leg(args.variant, args.res, args.spp, dev)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
