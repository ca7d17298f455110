#!/usr/bin/env python3
"""Benchmark of the EPSM manifold-gradient hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (``epsm_manifold_grad``: constraint
Jacobian blocks + block solve + adjoint gradients for every path) over one
wavefront of synthetic path records.  At N=1 the workload is BASELINE.json
``configs[1]``: bathroom, ``manifold``, 512x512 @ 64 spp -> 16 777 216 paths
with 5 logged vertices each.  With N>1 every rank processes its
own shard of pixel/sample tiles of the same size (weak scaling); ranks exchange
nothing on the data path of this kernel.

Prints ONE JSON line (rank 0).  ``value`` = paths/s over all ranks with inputs
resident in HBM; ``roofline`` prices the kernel against the 8 TB/s HBM peak
using the ALGORITHMIC bytes (32 + 200*K per path); ``cpu_baseline``
times oracle/ (the C restatement of the reference's calc_grad) on the host cores
of this box on a bounded sample of the same records.
"""
from __future__ import annotations

import argparse
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
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(path_info, dlduv, dldp, variant, target_s):
    """Oracle (kind 'port') on every host core, bounded sample of the same records."""
    from oracle.binding import oracle_calc_grad, build
    from epsm_mitsuba3_amd.synth import path_info_to
    build()
    N = path_info[0]["cam"].shape[0]

    def sample(n):
        sl = slice(0, n)
        out = []
        for rec in path_info:
            r = {}
            for k, v in rec.items():
                if isinstance(v, (list, tuple)):
                    r[k] = [x[sl].cpu() for x in v]
                elif isinstance(v, torch.Tensor):
                    r[k] = v[sl].cpu()
                else:
                    r[k] = v
            out.append(r)
        return out, dlduv[sl].cpu(), dldp[sl].cpu()

    from epsm_mitsuba3_amd.records import PackedRecords, VARIANTS, num_param_grads
    import ctypes as C
    from oracle import binding

    def timed(n):
        pi, d, p = sample(n)
        rec = PackedRecords(pi, device="cpu")
        K = rec.K
        d2 = d.reshape(n, -1).contiguous()
        P = num_param_grads(variant, K)
        op = torch.empty((P, n, 3)); ol = torch.empty((K, n, 3)); od = torch.empty((K, n, 3))
        fn = binding.lib().epsm_oracle_calc_grad_f32
        t0 = time.perf_counter()
        rc = fn(VARIANTS[variant], n, K, rec.cam.data_ptr(), C.addressof(rec.records), d2.data_ptr(),
                d2.shape[1], 2, p.contiguous().data_ptr(), 0.1, op.data_ptr(), ol.data_ptr(), od.data_ptr(), 0)
        dt = time.perf_counter() - t0
        assert rc > 0
        return dt, rc

    n0 = min(N, 1 << 16)
    dt0, cores = timed(n0)
    rate0 = n0 / dt0
    n1 = int(min(N, max(n0, rate0 * target_s), 1 << 23))
    dt1, cores = timed(n1)
    return {"value": n1 / dt1, "unit": "paths/s", "cores": int(cores), "kind": "port",
            "sample": f"first {n1} paths of the same workload, oracle/epsm_oracle.c fp32, {dt1:.2f} s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path to measure)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from epsm_mitsuba3_amd.synth import synth_path_info
    from epsm_mitsuba3_amd.records import PackedRecords, num_param_grads
    from epsm_mitsuba3_amd.manifold_grad import manifold_grad_packed

    N = args.res * args.res * args.spp          # paths of one gradient image (per rank)
    K = args.vertices
    path_info, dlduv, dldp = synth_path_info(N, K, seed=rank, device=dev, profile=args.profile)
    rec = PackedRecords(path_info, device=dev)
    P = num_param_grads(args.variant, K)
    out = (torch.empty((P, N, 3), device=dev), torch.empty((K, N, 3), device=dev),
           torch.empty((K, N, 3), device=dev))

    def step():
        manifold_grad_packed(args.variant, rec, dlduv, dldp, dlduv_cols=2, out=out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps       # HIP events on the launch stream
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * N * args.steps / elapsed

    result = None
    if rank == 0:
        alg = algorithmic_bytes_per_path(K) * N
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        result = {
            "metric": "manifold_paths_per_s", "value": value, "unit": "paths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "grad_image_ms": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"bathroom-like synthetic path records, {args.variant}, "
                                   f"{args.res}x{args.res} @ {args.spp} spp = {N} paths/GPU, K={K} logged vertices "
                                   f"(BASELINE.json configs[1])",
                       "variant": args.variant, "profile": args.profile, "paths_per_gpu": N, "vertices": K,
                       "sharding": f"{world} x pixel/sample-tile shard, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "epsm_grad_kernel", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg},
        }
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(path_info, dlduv, dldp, args.variant, args.cpu_seconds)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
