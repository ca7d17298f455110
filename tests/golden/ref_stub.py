"""Import the reference's ``epsm.py`` IN PLACE under ``mitsuba``/``drjit`` stubs.

Golden-vector tooling only.  This module reads
``/root/reference`` at import time of the reference file, so it works only in
the build container: nothing under ``tests/test_*.py -m gpu``, ``bench.py`` or
``__graft_entry__.smoke()`` may import it.  No reference source is copied; the
reference file is executed from where it lies.

What is stubbed and why (reference file:line):
  * ``import drjit as dr`` / ``import mitsuba as mi``  (epsm.py:3-4) -- not
    installable here; ``calc_grad`` (epsm.py:745-946, 952-1200) only touches
    ``mi.has_flag``, ``mi.BSDFFlags`` and ``mi.Point3f``.
  * ``.cuda()`` on freshly created tensors (epsm.py:768-773 ...) -- patched to
    the identity so the torch CPU backend is used.
  * ``from .common import ADIntegrator`` (epsm.py:9) -- the real common.py is
    imported too (it only needs ``mi.CppADIntegrator`` / ``mi.Properties``).
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("EPSM_REFERENCE_ROOT", "/root/reference")
_REF_PKG_DIR = os.path.join(REFERENCE_ROOT, "src/python/python/ad/integrators")

# include/mitsuba/render/bsdf.h:40-46,101
BSDF_NULL = 0x1
BSDF_DIFFUSE = 0x2 | 0x4


class _FlagResult:
    def __init__(self, t: torch.Tensor):
        self._t = t

    def torch(self) -> torch.Tensor:
        return self._t


class BsdfFlagsArray:
    """Stands in for the ``mi.UInt32`` returned by ``bsdf.flags()`` (epsm.py:649)."""

    def __init__(self, t: torch.Tensor):
        self.t = t.to(torch.int64)


def _has_flag(a, f):
    t = a.t if isinstance(a, BsdfFlagsArray) else torch.as_tensor(a)
    return _FlagResult((t & int(f)) != 0)


def available() -> bool:
    return os.path.isfile(os.path.join(_REF_PKG_DIR, "epsm.py"))


_cached = None


def load_reference_epsm():
    """Returns the reference module object (``refpkg.epsm``)."""
    global _cached
    if _cached is not None:
        return _cached
    if not available():
        raise RuntimeError(f"reference not present at {_REF_PKG_DIR}")

    mi = types.ModuleType("mitsuba")

    class _BSDFFlags:
        Null = BSDF_NULL
        Diffuse = BSDF_DIFFUSE

    class _Base:
        def __init__(self, props=None):
            pass

    mi.BSDFFlags = _BSDFFlags
    mi.has_flag = _has_flag
    mi.Point3f = lambda x: x
    mi.CppADIntegrator = _Base
    mi.Integrator = _Base
    mi.SamplingIntegrator = _Base
    mi.Properties = lambda: {}
    mi.PCG32 = object
    mi.register_integrator = lambda name, ctor: None
    dr = types.ModuleType("drjit")

    saved = {k: sys.modules.get(k) for k in ("mitsuba", "drjit")}
    sys.modules["mitsuba"] = mi
    sys.modules["drjit"] = dr
    torch.Tensor.cuda = lambda self, *a, **k: self  # epsm.py:768-773
    try:
        pkg = types.ModuleType("refpkg")
        pkg.__path__ = [_REF_PKG_DIR]  # package __init__.py is NOT executed
        sys.modules["refpkg"] = pkg
        mod = importlib.import_module("refpkg.epsm")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _cached = mod
    return mod


def reference_calc_grad(variant: str, path_info, dlduv, dldp):
    """Runs the reference ``calc_grad`` on CPU tensors.

    ``path_info`` follows epsm.py:547,649-654; its ``"bsdf"`` entries may be
    integer tensors (wrapped here).  Inputs are cloned because the reference
    mutates them in place (epsm.py:791, 999).
    Returns three lists of (N,3) tensors.
    """
    mod = load_reference_epsm()
    cls = {"manifold": mod.ManifoldIntegrator,
           "manifold_caustic": mod.ManifoldCausticIntegrator}[variant]
    integ = cls({})
    pi = []
    for rec in path_info:
        r = {}
        for k, v in rec.items():
            if k == "bsdf":
                r[k] = v if isinstance(v, BsdfFlagsArray) else BsdfFlagsArray(v)
            elif isinstance(v, (list, tuple)):
                r[k] = [x.detach().clone() for x in v]
            elif isinstance(v, torch.Tensor):
                r[k] = v.detach().clone()
            else:
                r[k] = v
        pi.append(r)
    dtype = dlduv.dtype
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)  # the reference allocates with torch.zeros(...) (epsm.py:768-773)
    try:
        fp, lg, dg = integ.calc_grad(path_info=pi, dlduv=dlduv.detach().clone(),
                                     dldp=dldp.detach().clone(), Lt=None)
    finally:
        torch.set_default_dtype(prev)
    det = lambda xs: [x.detach() for x in xs]
    return det(fp), det(lg), det(dg)
