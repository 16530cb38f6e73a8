import numpy as np, logging, sys
sys.path.insert(0, ".")
from revrand_amd import basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel as SLM
rs = np.random.RandomState(0)
N, d = 4000, 4
X = rs.randn(N, d)
f = lambda Z: np.sin(2 * Z[:, 0]) + 0.5 * Z[:, 1] - 0.2 * Z[:, 2]
y = f(X) + 0.05 * rs.randn(N)
cat = bs.RandomMatern52(nbases=150, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())) + bs.LinearBasis(onescol=True)
slm = SLM(cat, nstarts=0, maxiter=60, random_state=3)
orig = SLM._elbo_resident
def wrapped(self, X, y, var, reg, hypers):
    print("IN  var", var, "reg", reg, "hyp", hypers, flush=True)
    st = self._state
    G, b, yty = st.gram(hypers)
    print("   G finite", np.isfinite(G).all(), "b finite", np.isfinite(b).all(), yty, flush=True)
    out = orig(self, X, y, var, reg, hypers)
    print("OUT", out, flush=True)
    return out
SLM._elbo_resident = wrapped
try:
    slm.fit(X, y)
except Exception as e:
    print("EXC", repr(e))
