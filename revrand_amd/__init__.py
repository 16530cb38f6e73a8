"""
revrand_amd -- the MI355X-native hot path of NICTA/revrand: random-feature basis expansion
and Gram assembly in hand-written HIP (librevrand_hip.so, reached through ctypes), behind
revrand's own Basis / StandardLinearModel interface.  See DESIGN.md.
"""
from . import basis_functions, btypes  # noqa: F401
from .btypes import Bound, Parameter, Positive  # noqa: F401

__version__ = "0.1.0"
