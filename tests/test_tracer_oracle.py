"""The tracer against its non-self oracle on the CPU: the host build of the per-path code traces a small scene with
every BSDF kind, and tests/_trace_replay.py re-intersects the logged rays / vertices by brute force in float64
(oracle/epsm_oracle_trace.c) and re-derives the film positions from a numpy restatement of the TEA + PCG32 seeding.
The same replay runs on GPU traces of a 128 k-triangle scene in tests/test_gpu_tracer_oracle.py."""
import pytest
import torch

from _trace_replay import Pcg32Streams, check_film_positions, replay
from test_tracer_wavefront_host import _rich_scene


def test_pcg32_known_answers():
    """pcg32.h demo values: seed(42, 54) -> 0xa15c02b7, 0x7b47f409, 0xba1d3330 (the published PCG32 reference output)."""
    import numpy as np
    r = Pcg32Streams.__new__(Pcg32Streams)
    r.state = np.zeros(1, dtype=np.uint64)
    r.inc = np.array([(54 << 1) | 1], dtype=np.uint64)
    r.next_u32()
    r.state = r.state + np.uint64(42)
    r.next_u32()
    got = [int(r.next_u32()[0]) for _ in range(3)]
    assert got == [0xa15c02b7, 0x7b47f409, 0xba1d3330]


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
@pytest.mark.parametrize("max_depth,point_light,occluder", [(5, False, False), (3, True, True)])
def test_host_trace_replays_against_brute_force(tracer, max_depth, point_light, occluder):
    res, spp, K = 24, 8, min(4, max_depth)
    sc = _rich_scene(res, spp, point_light, occluder)
    sc.tracer = tracer
    n = res * res * spp
    tr = sc._trace(0, seed=11, spp=spp, max_depth=max_depth, K=K, lo=0, hi=n)
    assert check_film_positions(tr, 11, res) == 1.0                       # bit for bit
    rep = replay(sc, tr, K)
    assert rep["primary_rays"] > n // 2
    for name in ["primary"] + [f"bounce{k}" for k in range(1, K)]:
        if name + "_rays" not in rep:
            continue
        assert rep[name + "_hit_found"] == 1.0, (name, rep)
        assert rep[name + "_same_primitive"] >= 0.999, (name, rep)
        assert rep[name + "_t_agrees"] >= 0.999, (name, rep)
        assert rep[name + "_uv_agrees"] >= 0.999, (name, rep)
    assert rep.get("primary_miss_confirmed", 1.0) == 1.0
    assert rep["bounce1_rays"] > 100
    assert rep["shadow_rays"] > 500
    assert rep["occluded_have_zero_weight"] >= 0.998, rep
    if occluder:
        assert rep["occluded_share"] > 0.02, rep                         # the plate above the floor does block samples
    assert rep["visible_have_weight"] > 0.5, rep
    assert rep.get("emitter_point_rebuilt", 1.0) >= 0.999, rep
