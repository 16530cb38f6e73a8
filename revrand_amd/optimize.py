"""
Optimiser front-ends the estimators use: nested ``Parameter`` structures in, scipy's
L-BFGS-B underneath, with the log trick for ``Positive`` bounds.  Host control plane that
runs once per iteration (reference: revrand/optimize/decorators.py:24-130, 255-326,
541-617), restated compactly rather than as a decorator stack.
"""
import logging

import numpy as np

from .btypes import Bound, Positive
from .utils import flatten_values, issequence, shapes_of, unflatten

log = logging.getLogger(__name__)

MINPOS = 1e-100                             # decorators.py:18
MAXPOS = np.sqrt(np.finfo(float).max)       # decorators.py:19
LOGMINPOS, EXPMAX = np.log(MINPOS), np.log(MAXPOS)


def _map(fn, params):
    return [_map(fn, p) for p in params] if issequence(params) else fn(params)


def _flat_bounds(params):
    out = []

    def walk(p):
        if issequence(p):
            for q in p:
                walk(q)
        else:
            out.extend([p.bounds] * int(np.prod(p.shape, dtype=int)))
    walk(params)
    return out


def _random_start(fun, parameters, jac, args, nstarts, random_state):
    """Best of `nstarts` draws from the Parameters' distributions (decorators.py:541-583)."""
    if nstarts < 1:
        raise ValueError("nstarts has to be greater than or equal to 1")
    if not any(flatten_values(_map(lambda p: float(p.is_random), parameters))):
        log.info("No random parameters, not doing any random starts")
        return flatten_values(_map(lambda p: p.value, parameters))
    log.info("Evaluating random starts...")
    best_obj, best = np.inf, None
    for _ in range(nstarts):
        cand = _map(lambda p: p.rvs(random_state), parameters)
        res = fun(*(list(cand) + list(args)))
        obj = res[0] if jac is True else res
        if best is None or obj < best_obj:
            best_obj, best = obj, cand
    log.info("Best start found with objective = {}".format(best_obj))
    return flatten_values(best)


def structured_minimizer(minimizer):
    """Let `minimizer` (scipy.optimize.minimize-like) take nested Parameter lists.

    new_minimizer(fun, parameters, jac=True, args=(), nstarts=0, random_state=None, **kw):
    `fun(*values)` receives one value per top-level Parameter (python float for scalars,
    ndarray otherwise, [] for the null Parameter, nested lists kept) and returns
    ``(objective, nested_gradients)`` when jac is True.  ``result.x`` / ``result.jac`` are
    returned in the same nested structure (decorators.py:24-130).
    """
    def new_minimizer(fun, parameters, jac=True, args=(), nstarts=0, random_state=None, **kwargs):
        shapes = shapes_of(parameters, shape=lambda p: p.shape)
        x0 = flatten_values(_map(lambda p: p.rvs(random_state), parameters))
        bounds = _flat_bounds(parameters)
        if nstarts > 0:
            x0 = _random_start(fun, parameters, jac, args, nstarts, random_state)

        def flat_fun(x, *fargs):
            res = fun(*(unflatten(x, shapes) + list(fargs)))
            if jac is True:
                return res[0], flatten_values(res[1])
            return res

        flat_jac = jac
        if callable(jac):
            def flat_jac(x, *fargs):
                return flatten_values(jac(*(unflatten(x, shapes) + list(fargs))))

        result = minimizer(flat_fun, x0, jac=flat_jac, args=args, bounds=bounds, **kwargs)
        result["x"] = tuple(unflatten(result["x"], shapes))
        if bool(jac) and "jac" in result:
            result["jac"] = tuple(unflatten(result["jac"], shapes))
        return result
    return new_minimizer


def logtrick_minimizer(minimizer):
    """Optimise log(x) for every variable with a ``Positive`` bound (decorators.py:255-326)."""
    def new_minimizer(fun, x0, jac=True, bounds=None, **kwargs):
        if bounds is None:
            return minimizer(fun, x0, jac=jac, bounds=bounds, **kwargs)
        pos = np.array([isinstance(b, Positive) for b in bounds], dtype=bool)

        def to_log(x):
            z = np.array(x, dtype=float)
            z[pos] = np.log(z[pos])
            return z

        def from_log(z):
            x = np.array(z, dtype=float)
            x[pos] = np.exp(x[pos])
            return x

        def chain(g, z):
            g = np.array(g, dtype=float)
            g[pos] *= np.exp(z[pos])
            return g

        new_bounds = [Bound(LOGMINPOS, EXPMAX if b.upper is None else np.log(b.upper)) if p else b
                      for b, p in zip(bounds, pos)]

        if callable(jac):
            def new_jac(z, *a, **k):
                return chain(jac(from_log(z), *a, **k), z)
        else:
            new_jac = jac

        if (not callable(jac)) and bool(jac):
            def new_fun(z, *a, **k):
                o, g = fun(from_log(z), *a, **k)
                return o, chain(g, z)
        else:
            def new_fun(z, *a, **k):
                return fun(from_log(z), *a, **k)

        result = minimizer(new_fun, to_log(x0), jac=new_jac, bounds=new_bounds, **kwargs)
        result["x"] = from_log(result["x"])
        return result
    return new_minimizer
