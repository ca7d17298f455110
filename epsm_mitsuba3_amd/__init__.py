"""MI355X-native EPSM (Extended Path Space Manifolds) gradient hot path.

Drop-in for the ``manifold`` / ``manifold_caustic`` integrators' gradient path
of jkxing/EPSM_Mitsuba3 (src/python/python/ad/integrators/epsm.py).
"""
from .records import PackedRecords, EpsmVertexRecord, num_param_grads, VARIANTS  # noqa: F401
from .manifold_grad import calc_grad, manifold_grad_packed, OUTLIER_CLIP  # noqa: F401
from ._lib import EpsmError  # noqa: F401

__version__ = "0.1.0"
from .params import ParamGrads  # noqa: F401,E402
from .integrators import (EPSMIntegrator, ManifoldIntegrator, ManifoldCausticIntegrator, PathTrace,  # noqa: F401,E402
                          register_integrator, load_dict)
from .synthetic_scene import SyntheticScene  # noqa: F401,E402
