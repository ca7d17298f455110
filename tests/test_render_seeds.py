"""The primal and the differential pass of one iteration draw from different seeds (src/python/python/util.py:505-513)."""
import inspect

import numpy as np
import pytest

from epsm_mitsuba3_amd import optim
from epsm_mitsuba3_amd.integrators import render_seeds, sample_tea_32
from tests._trace_replay import _tea32


def test_tea_matches_the_numpy_restatement_of_the_sampler_seeding():
    rng = np.random.default_rng(0)
    v0 = rng.integers(0, 2 ** 32, size=64, dtype=np.uint64).astype(np.uint32)
    v1 = rng.integers(0, 2 ** 32, size=64, dtype=np.uint64).astype(np.uint32)
    a, b = _tea32(v0, v1)
    for i in range(64):
        assert sample_tea_32(int(v0[i]), int(v1[i])) == (int(a[i]), int(b[i]))


def test_differential_seed_is_derived_and_differs():
    seen = set()
    for seed in range(200):
        s, g = render_seeds(seed)
        assert s == seed and g == sample_tea_32(seed, 1)[0] and g != seed and 0 <= g < 2 ** 32
        seen.add(g)
    assert len(seen) == 200
    assert render_seeds(3, 17) == (3, 17)


def test_equal_seeds_are_refused_with_the_reference_message():
    with pytest.raises(Exception, match="primal and differential seed should be different"):
        render_seeds(5, 5)


def test_the_outer_loop_passes_the_derived_seed_to_the_backward_pass():
    src = inspect.getsource(optim.run)
    assert "render_seeds(it)" in src and "seed=seed_grad" in src and "render_backward(scene, params, grad, sensor=sid, seed=it" not in src
