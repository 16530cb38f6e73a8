"""One `StandardLinearModel._elbo` at the shape of the reference's SARCOS demo (N = 44 484 rows, D = 21, RandomRBF nbases = 512: F = 1024;
/root/reference/demos) -- resident, float32: stage times (statistics, posterior, second pass), the evaluation as the optimiser
calls it, and the host's oracle port of the same evaluation on 4096 rows for parity."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import revrand_amd.basis_functions as bs  # noqa: E402
from revrand_amd import _hip  # noqa: E402
from revrand_amd.btypes import Parameter, Positive  # noqa: E402
from revrand_amd.slm import StandardLinearModel  # noqa: E402

N, d, n = int(os.environ.get("N", 44484)), 21, int(os.environ.get("NBASES", 512))
rs = np.random.RandomState(0)
X = rs.randn(N, d).astype(np.float32)
y = (np.sin(X @ rs.randn(d) / 3) + 0.1 * rs.randn(N)).astype(np.float32)
dev = _hip.get_device()
DT = os.environ.get("DTYPE", "f32")   # DTYPE=f64: the reference's arithmetic (float64 features, Gram, second pass)
if DT == "f64":
    X, y = X.astype(np.float64), y.astype(np.float64)
basis = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()), dtype=DT)
slm = StandardLinearModel(basis)
slm.obj_ = -np.inf
slm._defer_cov = True
st = slm._state = slm._make_state(X, y)
F = 2 * n
ls, var, reg = np.linspace(2.0, 4.0, d), 0.3, 1.0
iL = np.full(F, 1.0 / reg)
slm._elbo(X, y, var, reg, ls)


def med(fn, reps=7):
    ts, out = [], None
    for _ in range(reps):
        dev.sync()
        t0 = time.perf_counter()
        out = fn()
        dev.sync()
        ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.median(ts)), out


t_stats, _ = med(lambda: st.gram_device(ls))
t_post, post = med(lambda: st.posterior(iL, var))
t_p2, _ = med(lambda: st.second_pass(ls, post[0], st.dC, var))
t_eval, _ = med(lambda: slm._elbo(X, y, var, reg, ls * 1.0))
fl = (2.0 * d * n + F * (F + 1.0) + 2.0 * F + 2.0 * F * F + 4.0 * d * n) * N
print("N=%d F=%d: statistics %.3f ms, posterior %.3f ms, second pass %.3f ms; _elbo %.3f ms = %.3f of the %s MFMA peak"
      % (N, F, t_stats, t_post, t_p2, t_eval, fl / (t_eval * 1e-3) / (157.3e12 if DT == "f32" else 78.6e12), DT))
t0 = time.perf_counter()
slm2 = StandardLinearModel(bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d) * 3.0, Positive()), dtype=DT), nstarts=0, maxiter=30)
slm2.fit(X, y)
print("fit(nstarts=0, maxiter=30): %.3f s" % (time.perf_counter() - t0))
