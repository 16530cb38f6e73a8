"""
Likelihoods of the generalised linear model with the reference's interface (revrand/likelihoods.py):
``loglike / Ey / df / dp / cdf`` on host arrays for the prediction paths, and ``device_spec`` naming the
per-element formulas ``rr_featmat_glm_step`` evaluates on the GPU inside an SVI step (glm.py:296-322).
"""
import numpy as np
from scipy.special import expit, gammaln
from scipy.stats import bernoulli, binom, gamma, norm, poisson

from .btypes import Parameter, Positive

# likelihood ids of the C ABI (include/revrand_hip.h)
RR_LIK_BERNOULLI, RR_LIK_BINOMIAL, RR_LIK_GAUSSIAN, RR_LIK_POISSON_EXP, RR_LIK_POISSON_SOFTPLUS = 0, 1, 2, 3, 4


def softplus(f):
    """log(1 + exp(f)), overflow-safe (mathfun/special.py:91-128)."""
    f = np.asarray(f, dtype=float)
    return np.maximum(f, 0.) + np.log1p(np.exp(-np.abs(f)))


class Bernoulli(object):
    """Bernoulli with a logistic link (likelihoods.py:18-150)."""

    _params = Parameter()

    @property
    def params(self):
        return self._params

    @params.setter
    def params(self, params):
        self._params = params

    def loglike(self, y, f):
        y, f = np.broadcast_arrays(y, f)
        return y * f - softplus(f)

    def Ey(self, f):
        return expit(f)

    def df(self, y, f):
        y, f = np.broadcast_arrays(y, f)
        return y - expit(f)

    def dp(self, y, f, *args):
        return []

    def cdf(self, y, f):
        return bernoulli.cdf(y, expit(f))

    # device_spec does not look at the likelihood's parameters: the SVI loop may evaluate it for a minibatch before the
    # step that uses it (glm.py, on the minibatch worker thread).  Gaussian overrides this.
    spec_is_parameter_free = True

    def device_spec(self, y, lpars, largs):
        """(likelihood id, scalar parameter, per-row argument or None, host constant added to sum(loglike)
        per latent sample) for the fused SVI step."""
        return RR_LIK_BERNOULLI, 0.0, None, 0.0

    def __repr__(self):
        return "{}()".format(type(self).__name__)


class Binomial(Bernoulli):
    """Binomial(n, logistic(f)); n is a (non-learnable) likelihood argument (likelihoods.py:153-258)."""

    def loglike(self, y, f, n):
        return binom.logpmf(y, n=n, p=expit(f))

    def Ey(self, f, n):
        return expit(f) * n

    def df(self, y, f, n):
        y, f, n = np.broadcast_arrays(y, f, n)
        return y - expit(f) * n

    def cdf(self, y, f, n):
        return binom.cdf(y, n=n, p=expit(f))

    def device_spec(self, y, lpars, largs):
        n = np.broadcast_to(np.asarray(largs[0], dtype=float), np.shape(y))
        const = float((gammaln(n + 1) - gammaln(y + 1) - gammaln(n - y + 1)).sum())
        return RR_LIK_BINOMIAL, 0.0, n, const


class Gaussian(Bernoulli):
    """Gaussian with learnable variance (likelihoods.py:261-423)."""

    def __init__(self, var=Parameter(gamma(1., scale=1), Positive())):
        self.params = var

    def _check_param(self, param):
        if param is None:
            return self.params.value
        if not self.params.bounds.check(param):
            raise ValueError("Input parameter is out of bounds!")
        return param

    def loglike(self, y, f, var=None):
        var = self._check_param(var)
        y, f = np.broadcast_arrays(y, f)
        return -0.5 * (np.log(2 * np.pi * var) + (y - f) ** 2 / var)

    def Ey(self, f, var):
        var = self._check_param(var)
        return f

    def df(self, y, f, var):
        var = self._check_param(var)
        y, f = np.broadcast_arrays(y, f)
        return (y - f) / var

    def dp(self, y, f, var):
        var = self._check_param(var)
        y, f = np.broadcast_arrays(y, f)
        ivar = 1. / var
        return 0.5 * (((y - f) * ivar) ** 2 - ivar)

    def cdf(self, y, f, var):
        var = self._check_param(var)
        return norm.cdf(y, loc=f, scale=np.sqrt(var))

    spec_is_parameter_free = False

    def device_spec(self, y, lpars, largs):
        var = float(self._check_param(lpars[0] if len(lpars) else None))
        return RR_LIK_GAUSSIAN, var, None, float(-0.5 * np.log(2 * np.pi * var) * np.size(y))

    def __repr__(self):
        return "{}(var={})".format(type(self).__name__, self.params)


def _sum_gammaln1p(y):
    """sum(gammaln(y + 1)) -- counts (the usual case) through a table of log-factorials instead of 65 536 gammaln calls"""
    y = np.asarray(y, dtype=float)
    if y.size == 0:
        return 0.0
    top = y.max()
    if 0 <= y.min() and top < 4096:
        yi = y.astype(np.int64)
        if np.array_equal(yi, y):
            return float((np.bincount(yi, minlength=1) * gammaln(np.arange(int(top) + 1) + 1.0)).sum())
    return float(gammaln(y + 1).sum())


class Poisson(Bernoulli):
    """Poisson with an exp or softplus link (likelihoods.py:426-545)."""

    def __init__(self, tranfcn='exp'):
        if tranfcn == 'exp' or tranfcn == 'softplus':
            self.tranfcn = tranfcn
        else:
            raise ValueError('Invalid transformation function specified!')

    def loglike(self, y, f):
        y, f = np.broadcast_arrays(y, f)
        if self.tranfcn == 'exp':
            g, logg = np.exp(f), f
        else:
            g = softplus(f)
            logg = np.log(g)
        return y * logg - g - gammaln(y + 1)

    def Ey(self, f):
        return np.exp(f) if self.tranfcn == 'exp' else softplus(f)

    def df(self, y, f):
        y, f = np.broadcast_arrays(y, f)
        if self.tranfcn == 'exp':
            return y - np.exp(f)
        return expit(f) * (y / np.maximum(softplus(f), 1e-100) - 1)

    def cdf(self, y, f):
        mu = np.exp(f) if self.tranfcn == 'exp' else softplus(f)
        return poisson.cdf(y, mu=mu)

    def device_spec(self, y, lpars, largs):
        lid = RR_LIK_POISSON_EXP if self.tranfcn == 'exp' else RR_LIK_POISSON_SOFTPLUS
        return lid, 0.0, None, -_sum_gammaln1p(y)

    def __repr__(self):
        return "{}(tranfcn='{}')".format(type(self).__name__, self.tranfcn)
