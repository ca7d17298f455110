"""Known answers held by the reference's OWN unit tests (tests/golden/reference_vectors.py, each value with its file:line)
against the oracle and the host builds of the product's per-path code -- the CPU half; tests/test_gpu_reference_vectors.py
drives the device code with the same assertions (tests/_refvec.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import _refvec as R
from tests.golden import reference_vectors as RV


def host_probe(what, rows, cfg=None):
    from tests._scenes import host_tracer
    lib = host_tracer()
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    out = np.zeros((rows.shape[0], R.PROBE_OUT), dtype=np.float32)
    lib.epsm_probe.restype = C.c_int
    rc = lib.epsm_probe(C.c_int(what), C.c_int64(rows.shape[0]), rows.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                        C.byref(cfg) if cfg is not None else None, None)
    assert rc == 0
    return out


@pytest.mark.parametrize("check", R.ALL_PROBE_CHECKS, ids=lambda f: f.__name__)
def test_tracer_functions_on_the_host(check):
    check(host_probe)


def test_sampler_stream_and_host_tea():
    from epsm_mitsuba3_amd.integrators import sample_tea_32
    R.check_tea_python(sample_tea_32)
    R.check_sampler_stream(host_probe, sample_tea_32)
    from tests._trace_replay import _tea32, Pcg32Streams                 # the replay checker's own copies
    v0, v1 = _tea32(np.array([a for a, _ in RV.TEA_INPUTS], dtype=np.uint32), np.array([b for _, b in RV.TEA_INPUTS], dtype=np.uint32))
    assert [R.tea_float32(x) for x in v1] == RV.TEA_FLOAT32
    assert [R.tea_float64(a, b) for a, b in zip(v0, v1)] == RV.TEA_FLOAT64
    st = Pcg32Streams(7, np.array([123456], dtype=np.uint32))
    py = R.Pcg32Py(*sample_tea_32(7, 123456))
    assert [float(st.next_f32()[0]) for _ in range(8)] == [py.next_f32() for _ in range(8)]


def test_probe_refuses_bad_arguments():
    from tests._scenes import host_tracer
    lib = host_tracer()
    lib.epsm_probe.restype = C.c_int
    assert lib.epsm_probe(C.c_int(99), C.c_int64(1), None, None, None, None) != 0
    rows = R.rows_of([[0.0, 0.0, 1.0, 0.0, 0.0, 1.0]])
    out = np.zeros((1, R.PROBE_OUT), dtype=np.float32)
    assert lib.epsm_probe(C.c_int(R.PROBE["MICROFACET"]), C.c_int64(1), rows.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                          None, None) != 0                                # needs an EpsmBsdf


# ------------------------------------------------------------------------------------------------- tangent
def test_oracle_intersection_tangents_match_test_mesh():
    """oracle/epsm_oracle_aux.c's dual-number Moeller-Trumbore against src/render/tests/test_mesh.py:380-455."""
    from oracle.binding import oracle_intersect_tangent
    o, d = RV.TANGENT_RAY
    p0, p1, p2 = R.rect_triangle(RV.RECT_LOWER)
    uv = [RV.RECT_TEXCOORDS[j] for j in RV.RECT_LOWER]
    z = (0.0, 0.0, 0.0)

    def forward(which, seed):
        r = oracle_intersect_tangent(o, d, seed if which == "o" else z, seed if which == "d" else z, p0, p1, p2)
        db1, db2 = r["du"], r["dv"]
        r["duv"] = -(db1 + db2) * uv[0] + db1 * uv[1] + db2 * uv[2]       # si.uv = sum_j b_j uv_j (mesh.cpp:736-745)
        return r
    base = forward("o", z)
    assert np.isclose(base["t"], 10.0) and 0 < base["u"] < 1 and 0 < base["v"] < 1
    for which, seed, what, ref in RV.TANGENT_FORWARD:
        r = forward(which, seed)
        got = {"p": r["dp"], "uv": r["duv"], "t": [r["dt"]]}[what]
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-8), (which, seed, what, got)
    # reverse mode (test_mesh.py:439-455): the adjoint of output component i w.r.t. ray.o is row i of the Jacobian whose
    # columns the three forward passes give
    cols = [forward("o", tuple(float(j == c) for j in range(3))) for c in range(3)]
    for what, comp, ref in RV.TANGENT_BACKWARD:
        row = [(c["dp"][comp] if what == "p" else c["dt"]) for c in cols]
        assert np.allclose(row, ref, rtol=1e-5, atol=1e-8), (what, row)


def _oracle_tangent(o, d, dx, dy, gx, gy, p0, p1, p2):
    from oracle.binding import oracle_first_vertex_tangent
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64).reshape(1, 3), dtype=torch.float32)
    g = torch.zeros((1, 1, 5)); g[0, 0, 3], g[0, 0, 4] = gx, gy
    uv, dp, _ = oracle_first_vertex_tangent(t(o), t(d), t(dx), t(dy), g, 1, 1, t(p0), t(p1), t(p2), torch.ones(1, dtype=torch.bool))
    return float(uv[0, 0, 0]), float(uv[0, 0, 1]), dp[0].numpy()


def _host_core_tangent(o, d, dx, dy, gx, gy, p0, p1, p2):
    from tests.host_core import lib
    l = lib("path")
    inp = np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1) for a in (o, d, dx, dy, [gx, gy], p0, p1, p2)]).astype(np.float32)
    out = np.zeros(8, dtype=np.float32)
    l.epsm_host_tangent_from.restype = C.c_int
    assert l.epsm_host_tangent_from(inp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    return float(out[0]), float(out[1]), out[2:5].astype(np.float64)


def test_first_vertex_tangent_matches_test_mesh():
    R.check_tangent(_oracle_tangent)                    # the oracle's epsm.py:250-272
    R.check_tangent(_host_core_tangent)                 # the product's closed form (csrc/epsm_tangent_core.h), host build


# ------------------------------------------------------------------------------------------------- scatter
def test_oracle_scatter_matches_test_mesh():
    from oracle.binding import oracle_scatter

    def scatter(pi, si, op, ol, od, V):
        gp, gn, ga = oracle_scatter("manifold", pi, si, op, ol, od, V, 0)
        assert float(gn.abs().max()) == 0.0             # a mesh without vertex normals: nothing reaches vertex_normals
        return gp.numpy()
    R.check_scatter(scatter)
