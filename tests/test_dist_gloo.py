"""World-size-2 test of the multi-GPU logic on CPU (gloo): tile partitioning, the
flat parameter-gradient buffer and its single all-reduce.  The per-tile arithmetic is
supplied by the oracle here (tests may use it); on GPUs the same orchestration calls
the HIP kernels (tests/test_gpu_pipeline.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from epsm_mitsuba3_amd import dist as edist
from epsm_mitsuba3_amd.params import ParamGrads
from epsm_mitsuba3_amd.synthetic_scene import SyntheticScene


def _ship(*items):
    """Tensors cross the queue BY VALUE (numpy arrays pickle inline).  A torch tensor crosses as a handle the receiver redeems from
    the sender's process -- which may have left by then: FileNotFoundError / ConnectionResetError under load (sanitizer run, xdist)."""
    return tuple(i.detach().cpu().numpy().copy() if torch.is_tensor(i) else i for i in items)


def _unship(item):
    import numpy as np
    return tuple(torch.from_numpy(i) if isinstance(i, np.ndarray) else i for i in item)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


RES, SPP, K, V, B, TILE = 16, 8, 3, 500, 3, 512   # 2048 paths -> 4 tiles


def _grad_image():
    g = torch.Generator().manual_seed(3)
    return torch.randn((RES, RES, 5), generator=g)


def _rank_work(rank, world):
    from _pipeline_oracle import oracle_backward
    scene = SyntheticScene(res=RES, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile="mixed",
                           device="cpu", tile_paths=TILE)
    traces = scene.trace_paths(seed=5, spp=SPP, rank=rank, world_size=world)
    gp, gn, ga, go = oracle_backward("manifold", traces, _grad_image(), V, B)
    params = ParamGrads(V, B, device="cpu")
    params.pos += gp.float(); params.nrm += gn.float(); params.alpha += ga.float(); params.cam_origin += go.float()
    return params, [t.path_offset for t in traces]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    r, w, _ = edist.init_from_env("gloo")
    assert (r, w) == (rank, world) and edist.world() == (rank, world)
    params, offsets = _rank_work(rank, world)
    edist.allreduce_param_grads(params.flat)
    q.put(_ship(rank, offsets, params.flat))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_partition_is_exact():
    tiles = edist.tile_ranges(2048, 512)
    assert tiles == [(0, 512), (512, 1024), (1024, 1536), (1536, 2048)]
    assert edist.tile_ranges(1000, 512) == [(0, 512), (512, 1000)]
    for world in (1, 2, 3, 8):
        seen = sorted(t for r in range(world) for t in edist.my_tiles(11, r, world))
        assert seen == list(range(11))


def test_two_rank_backward_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [_unship(q.get(timeout=180)) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert got[0][1] == [0, 1024] and got[1][1] == [512, 1536]      # round-robin tiles
    assert torch.equal(got[0][2], got[1][2])                        # both ranks hold the sum
    single, _ = _rank_work(0, 1)
    assert float(single.flat.abs().max()) > 0
    assert torch.allclose(got[0][2], single.flat, rtol=1e-5, atol=1e-7 * float(single.flat.abs().max()))


class _OracleIntegrator:
    """``ManifoldIntegrator`` whose per-tile arithmetic is the float64 oracle: what is under test is the
    orchestration of ``render_backward`` (tiles of this rank, scratch buffer, ONE all-reduce, accumulation)."""

    @staticmethod
    def make():
        import epsm_mitsuba3_amd as epsm
        from _pipeline_oracle import oracle_backward

        class Integ(epsm.ManifoldIntegrator):
            def backward_from_trace(self, trace, params, grad_in, **kw):
                gp, gn, ga, go = oracle_backward(self.variant, [trace], grad_in, params.V, params.B)
                params.pos += gp.float(); params.nrm += gn.float(); params.alpha += ga.float(); params.cam_origin += go.float()
        integ = Integ({"max_depth": 8})
        integ.backward_spp = SPP
        return integ


def _accumulate_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    edist.init_from_env("gloo")
    scene = SyntheticScene(res=RES, n_vertices=K, n_scene_vertices=V, n_bsdfs=B, profile="mixed", device="cpu", tile_paths=TILE)
    integ = _OracleIntegrator.make()
    params = ParamGrads(V, B, device="cpu")
    integ.render_backward(scene, params, _grad_image(), seed=5)
    once = params.flat.clone()
    integ.render_backward(scene, params, _grad_image(), seed=5)      # dr.backward accumulates: NOT zeroed in between
    q.put(_ship(rank, once, params.flat))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_render_backward_accumulates_without_rescaling():
    """a second accumulating call must add ONE more copy of the gradients on every rank -- the values
    already in ``params`` are a sum over the ranks and must not go through the all-reduce again."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_accumulate_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([_unship(q.get(timeout=180)) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single, _ = _rank_work(0, 1)
    m = float(single.flat.abs().max())
    for rank, once, twice in got:
        assert torch.allclose(once, single.flat, rtol=1e-5, atol=1e-7 * m)
        assert torch.allclose(twice, 2 * single.flat, rtol=1e-5, atol=2e-7 * m)


def _render_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _scenes import floor_and_light, on_host
    edist.init_from_env("gloo")
    sc = on_host(floor_and_light(res=16, device="cpu"))
    sc.tile_paths = 256                                  # 16*16*4 = 1024 paths -> 4 tiles, 2 per rank
    img = sc.render_primal(sensor=0, seed=4, spp=4, max_depth=3)       # rank / world from the process group
    q.put(_ship(rank, img))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_primal_render_matches_single_process():
    """The primal pass shards its tiles over the ranks and all-reduces the film accumulator [r,g,b,w]
    before the weight division: every rank ends up with the single-process image (host tracer here)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _scenes import floor_and_light, on_host
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_render_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([_unship(q.get(timeout=180)) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sc = on_host(floor_and_light(res=16, device="cpu"))
    sc.tile_paths = 256
    single = sc.render_primal(sensor=0, seed=4, spp=4, max_depth=3)
    assert float(single.max()) > 0
    assert torch.equal(got[0][1], got[1][1])
    assert torch.allclose(got[0][1], single, rtol=1e-5, atol=1e-6)


def _reparam_single(tile_paths):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import epsm_mitsuba3_amd as epsm
    from _reparam_scenes import build
    sc = build("diffuse_sphere_area_light", 0.0, 12, 8, "cpu")
    sc.tile_paths = tile_paths                            # 16 * 16 * 8 = 2048 paths (12 + 2 * 2 border pixels a side)
    sc.attach("sphere", positions=True, normals=True)
    integ = epsm.load_dict({"type": "prb_reparam", "max_depth": 3, "reparam_rays": 8})
    params = sc.param_grads()
    g = torch.ones((12, 12, 3)) * (0.5 + torch.arange(12, dtype=torch.float32) / 12)[None, :, None]
    integ.render_backward(sc, params, g, sensor=0, seed=3, spp=8)
    once = params.flat.clone()
    integ.render_backward(sc, params, g, sensor=0, seed=3, spp=8)
    return once, params.flat.clone()


def _reparam_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    edist.init_from_env("gloo")
    once, twice = _reparam_single(512)
    q.put(_ship(rank, once, twice))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_prb_reparam_matches_single_process():
    """The reparameterised backward pass shards its tiles like the others: the film of the primal pass is all-reduced
    before the adjoints are formed, each rank replays its own tiles, ONE all-reduce sums the geometry gradients, and a
    second accumulating call adds one more copy."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reparam_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([_unship(q.get(timeout=240)) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single, single2 = _reparam_single(512)
    m = float(single.abs().max())
    assert m > 0
    for rank, once, twice in got:
        assert torch.allclose(once, single, rtol=1e-4, atol=1e-5 * m)
        assert torch.allclose(twice, 2 * single, rtol=1e-4, atol=2e-5 * m)
