"""Multi-GPU backward pass: pixel/sample tiles are sharded over the ranks of one
node (one process per GPU) and the parameter-gradient buffer is summed with ONE
all-reduce (RCCL over xGMI when the backend is "nccl").  The
reference has no distributed code at all (single GPU, EPSM/optim.py:9-18).

Paths are independent, so the data path needs no collective; the only shared
state is ``ParamGrads.flat``.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    """(rank, world_size) of the default group, (0, 1) when not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Joins the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*.
    Returns (rank, world_size, local_rank).  ``backend`` defaults to "nccl" (= RCCL on
    ROCm) when a GPU is visible and to "gloo" otherwise."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    r = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if w > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(lr)
            kw["device_id"] = torch.device("cuda", lr)
        dist.init_process_group(backend, rank=r, world_size=w, **kw)
    return r, w, lr


TILE_PATHS = 1 << 20   # paths per tile: ~1 GB of records + outputs at K=5, far below 288 GB of HBM


def tile_ranges(n_paths: int, tile: int = TILE_PATHS) -> List[Tuple[int, int]]:
    """[lo, hi) path ranges of the fixed-size tiles a wavefront is cut into (tiles are
    ordered (pixel-tile, sample) because paths are ordered (pixel, sample))."""
    return [(lo, min(lo + tile, n_paths)) for lo in range(0, n_paths, tile)]


def my_tiles(n_tiles: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> List[int]:
    """Round-robin assignment tile t -> rank t % world: consecutive tiles
    cover neighbouring pixels, so every rank sees a similar mix of path lengths."""
    if rank is None or world_size is None:
        rank, world_size = world()
    return list(range(rank, n_tiles, world_size))


def allreduce_param_grads(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum of the flat parameter-gradient buffer over all ranks."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            # a gloo group over device buffers (rehearsals of the multi-rank path on one GPU): through the host
            host = flat.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat
