"""
Bound / Positive / Parameter: the parameter types the reference's estimators and
bases exchange (reference: revrand/btypes.py:81-349).  Host-side plumbing, restated
minimally so that the basis and model classes here accept and return the same objects.
"""
from collections import namedtuple

import numpy as np
from sklearn.utils import check_random_state


def _frozen_rvs(dist, shape, rs):
    """`dist.rvs(size=shape, random_state=rs)` for the two frozen scipy.stats distributions the estimators' Parameters use --
    `norm` (glm.py:40, the mixture means) and `gamma` (glm.py:41, basis_functions.py:210: covariances, regularisers, length
    scales, the Gaussian's variance) -- by the calls scipy itself ends in (`random_state.standard_normal(size)`,
    `random_state.standard_gamma(a, size)`, then `* scale + loc`): the same values from the same stream, without the ~25 us
    of argument checking per call that made up half of a default-shaped GLM fit's random starts (4000 calls).  None: not one
    of these (scipy's own `rvs` then)."""
    gen = getattr(dist, "dist", None)
    name = getattr(gen, "name", None)
    if name not in ("norm", "gamma") or type(rs) is not np.random.RandomState:
        return None
    try:
        args, loc, scale = gen._parse_args(*dist.args, **dist.kwds)
    except Exception:
        return None
    if not (np.ndim(loc) == 0 and np.ndim(scale) == 0 and all(np.ndim(a) == 0 for a in args)) or not scale > 0:
        return None
    if name == "norm":
        vals = rs.standard_normal(shape)
    else:
        if len(args) != 1 or not args[0] > 0:
            return None
        vals = rs.standard_gamma(args[0], shape)
    vals = vals * scale + loc
    return vals[()] if shape == () else vals


class _BoundChecks(object):
    """check()/clip() shared by Bound and Positive (btypes.py:12-79)."""

    def check(self, value):
        if self.lower and np.any(value < self.lower):
            return False
        if self.upper and np.any(value > self.upper):
            return False
        return True

    def clip(self, value):
        if not self.lower and not self.upper:
            return value
        return np.clip(value, self.lower, self.upper)


class Bound(namedtuple("Bound", ["lower", "upper"]), _BoundChecks):
    """Closed interval for an optimiser variable; (None, None) = unbounded (btypes.py:81-146)."""

    def __new__(cls, lower=None, upper=None):
        if lower is not None and upper is not None and lower > upper:
            raise ValueError("lower bound cannot be greater than upper bound!")
        return super(Bound, cls).__new__(cls, lower, upper)

    def __getnewargs__(self):
        return (self.lower, self.upper)

    def __repr__(self):
        return "{}(lower={}, upper={})".format(type(self).__name__, self.lower, self.upper)


class Positive(namedtuple("Positive", ["lower", "upper"]), _BoundChecks):
    """Strictly positive bound, lower = 1e-14; triggers the log trick (btypes.py:149-190)."""

    def __new__(cls, upper=None):
        lower = 1e-14
        if upper is not None and lower > upper:
            raise ValueError("Upper bound must be greater than {}".format(lower))
        return super(Positive, cls).__new__(cls, lower, upper)

    def __getnewargs__(self):
        return (self.upper,)

    def __repr__(self):
        return "{}(upper={})".format(type(self).__name__, self.upper)


class Parameter(object):
    """A value (or scipy.stats distribution) with a bound and a shape (btypes.py:193-349).

    With a distribution, ``value`` is the clipped mean (broadcast to ``shape``) and ``rvs``
    draws clipped samples; otherwise ``rvs`` returns the value.  ``Parameter()`` is the null
    parameter (``has_value`` False).
    """

    def __init__(self, value=None, bounds=Bound(), shape=()):
        if value is None:
            value = []
        if hasattr(value, "rvs"):
            self.dist = value
            self.shape = shape
            mean = bounds.clip(value.mean())
            self.value = mean if shape == () else mean * np.ones(shape)
        else:
            if np.any(value) and not bounds.check(value):
                raise ValueError("Value not within bounds!")
            self.dist = None
            self.value = value
            self.shape = np.shape(value)
        self.bounds = bounds

    def rvs(self, random_state=None):
        if self.dist is None:
            return self.value
        rs = check_random_state(random_state)
        fast = _frozen_rvs(self.dist, self.shape, rs)
        return self.bounds.clip(fast if fast is not None else self.dist.rvs(size=self.shape, random_state=rs))

    @property
    def has_value(self):
        return self.shape != (0,)

    @property
    def is_random(self):
        return self.dist is not None

    @property
    def is_scalar(self):
        return self.has_value and self.shape == ()

    def __repr__(self):
        return "{}(value={}, bounds={}, shape={})".format(type(self).__name__, self.value, self.bounds,
                                                          self.shape)
