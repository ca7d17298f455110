"""Small analytic scenes + the host-harness backend of the tracer (test infrastructure)."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

from epsm_mitsuba3_amd import scene as S

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_harness")
_SAN = os.environ.get("EPSM_SAN", "0") == "1"      # tools/run_san.sh: the AddressSanitizer / UBSan build (Makefile `san`)
_SO = os.path.join(_DIR, "libtrace_host_san.so" if _SAN else "libtrace_host.so")
_lib = None


def host_tracer():
    global _lib
    if _lib is None:
        from epsm_mitsuba3_amd._lib import build_lock
        with build_lock(_DIR):
            subprocess.run(["make", "-C", _DIR, "-s", _SO], check=True)            # make decides what is stale
        _lib = C.CDLL(_SO)
        for n in ("epsm_trace_paths", "epsm_trace_paths_wavefront", "epsm_film_splat", "epsm_film_develop"):
            getattr(_lib, n).restype = C.c_int
        _lib.epsm_trace_workspace_bytes.restype = C.c_size_t
        _lib.epsm_trace_workspace_bytes.argtypes = [C.c_int64]
    return _lib


def on_host(scene: S.Scene) -> S.Scene:
    scene._backend = host_tracer()
    return scene


def quad(z=0.0, half=1.0, up=True):
    v = np.array([[-half, -half, z], [half, -half, z], [half, half, z], [-half, half, z]], float)
    f = np.array([[0, 1, 2], [0, 2, 3]]) if up else np.array([[0, 2, 1], [0, 3, 2]])
    return v, f


def sensor(origin, target, up=(0, 1, 0), fov=40, res=16, spp=4, rfilter="box", near=0.01, far=100.0, sample_border=False):
    return {"type": "perspective", "fov": fov, "near_clip": near, "far_clip": far,
            "to_world": S.look_at(origin, target, up),
            "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": rfilter}, "sample_border": sample_border},
            "sampler": {"type": "independent", "sample_count": spp}}


def floor_and_light(radiance=50.0, light_half=0.05, height=2.0, reflectance=0.6, res=16, bsdf=None, device="cpu"):
    """Diffuse floor z=0 (normal +z), small square light at z=height facing down, camera looking at the origin."""
    fv, ff = quad(0.0, 3.0, up=True)
    lv, lf = quad(height, light_half, up=False)          # normal -z
    d = {"type": "scene",
         "cam": sensor([0.0, -2.0, 1.5], [0, 0, 0], up=(0, 0, 1), res=res),
         "floor": {"type": "mesh", "vertices": fv, "faces": ff, "face_normals": True,
                   "bsdf": bsdf or {"type": "diffuse", "reflectance": {"type": "rgb", "value": [reflectance] * 3}}},
         "light": {"type": "mesh", "vertices": lv, "faces": lf, "face_normals": True,
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0, 0, 0]}},
                   "emitter": {"type": "area", "radiance": {"type": "rgb", "value": radiance}}}}
    return S.Scene.from_dict(d, device=device)


def furnace(radiance=1.0, reflectance=(0.5, 0.3, 0.8), res=8, spp=16, bsdf=None, device="cpu"):
    return S.Scene.from_dict(furnace_dict(radiance, reflectance, res, spp, bsdf), device=device)


def furnace_dict(radiance=1.0, reflectance=(0.5, 0.3, 0.8), res=8, spp=16, bsdf=None):
    """Closed box [-1,1]^3 seen from inside: every wall emits ``radiance`` and reflects ``reflectance`` (diffuse unless
    ``bsdf`` is given).  The radiance along ANY ray is L (1 + rho + rho^2 + ...) cut after the integrator's depth."""
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], float)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = []
    for a, b, c, d_ in quads:
        f += [[a, b, c], [a, c, d_]]
    f = np.array(f)
    # orient every face towards the inside
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    flip = (n * v[f].mean(1)).sum(1) > 0
    f[flip] = f[flip][:, ::-1]
    d = {"type": "scene", "cam": sensor([0.2, -0.1, 0.3], [1.0, 0.4, 0.2], up=(0, 0, 1), fov=70, res=res, spp=spp),
         "box": {"type": "mesh", "vertices": v, "faces": f, "face_normals": True,
                 "bsdf": bsdf or {"type": "diffuse", "reflectance": {"type": "rgb", "value": list(reflectance)}},
                 "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [radiance] * 3}}}}
    return d


def furnace_with_ball(bsdf, wall_reflectance=0.0, res=8, spp=64, smooth=False, device="cpu"):
    """The furnace with a non-emitting ball (320 triangles, radius 0.25) in the middle of the view."""
    from epsm_mitsuba3_amd.exp.clutter import icosphere
    v, f = icosphere(2)
    centre = np.array([0.62, 0.15, 0.27])
    d = furnace_dict(reflectance=(wall_reflectance,) * 3, res=res, spp=spp)
    d["cam"] = sensor([0.2, -0.1, 0.3], centre, up=(0, 0, 1), fov=70, res=res, spp=spp)
    d["ball"] = {"type": "mesh", "vertices": v * 0.25 + centre, "faces": f, "face_normals": not smooth, "bsdf": bsdf}
    return S.Scene.from_dict(d, device=device)
