"""
revrand_amd -- the MI355X-native hot path of NICTA/revrand: random-feature basis expansion
and Gram assembly in hand-written HIP (librevrand_hip.so, reached through ctypes), behind
revrand's own Basis / StandardLinearModel interface.  See DESIGN.md.
"""
from . import basis_functions, btypes, likelihoods  # noqa: F401
from .btypes import Bound, Parameter, Positive  # noqa: F401


def __getattr__(name):
    # estimators import scikit-learn / scipy.optimize: loaded on first use (as `revrand.StandardLinearModel`)
    if name == "StandardLinearModel":
        from .slm import StandardLinearModel
        return StandardLinearModel
    if name in ("GeneralizedLinearModel", "GeneralisedLinearModel"):
        from . import glm
        return getattr(glm, name)
    raise AttributeError(name)

__version__ = "0.1.0"
