"""The GPU tracer against its non-self oracle: both tracer forms trace exp/clutter.py -- floor +
100 tessellated spheres + area light = 128 004 triangles, diffuse and rough-conductor spheres -- and the replay of
tests/_trace_replay.py re-intersects >= 10^5 logged rays per bounce against EVERY triangle in float64
(oracle/epsm_oracle_trace.c, no BVH): primitive index, (t, b0, b1), visibility of the emitter samples, and the film
positions from a numpy restatement of the TEA + PCG32 sampler seeding (common.py:320-335, sampler.cpp:115-134)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tracer", ["mega", "wavefront"])
def test_gpu_trace_of_the_clutter_scene_replays_against_brute_force(tracer):
    from _trace_replay import check_film_positions, replay
    from epsm_mitsuba3_amd.exp import clutter
    dev = torch.device("cuda", 0)
    res, spp, K = 128, 8, 4
    sc = clutter.load_scene(dev, n_spheres=100, res=res, spp=spp)
    assert sc.T == 128004
    sc.tracer = tracer
    n = res * res * spp                                                   # 131 072 primary rays
    tr = sc._trace(2, seed=7, spp=spp, max_depth=clutter.max_depth, K=K, lo=0, hi=n)
    torch.cuda.synchronize()
    assert check_film_positions(tr, 7, res) == 1.0                        # sample positions: bit for bit
    rep = replay(sc, tr, K)
    print(tracer, rep)
    assert rep["primary_rays"] > 100000                                   # >= 10^5 rays against 128 k triangles each
    for name in ["primary", "bounce1", "bounce2", "bounce3"]:
        assert rep[name + "_hit_found"] >= 0.9999, (name, rep)            # (a grazing ray may slip between fp32 and fp64)
        assert rep[name + "_same_primitive"] >= 0.999, (name, rep)
        assert rep[name + "_exact_primitive"] >= 0.995, (name, rep)       # the rest are edge ties (runner-up at the same t)
        assert rep[name + "_t_agrees"] >= 0.999, (name, rep)
        assert rep[name + "_uv_agrees"] >= 0.995, (name, rep)
    assert rep["primary_miss_confirmed"] >= 0.999, rep
    assert rep["bounce1_rays"] > 30000 and rep["bounce3_rays"] > 5000, rep
    assert rep["shadow_rays"] > 100000, rep
    assert rep["occluded_have_zero_weight"] >= 0.998, rep
    assert 0.02 < rep["occluded_share"] < 0.98, rep
    assert rep["emitter_point_rebuilt"] >= 0.999, rep
