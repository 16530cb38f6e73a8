import sys, numpy as np
sys.path.insert(0, '.')
import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel as SLM
rs = np.random.RandomState(0)
N, d, n = 20000, 6, 256
X = rs.randn(N, d); y = np.sin(X @ rs.randn(d)) + 0.05 * rs.randn(N)
ls = np.linspace(0.8, 1.5, d)
for var, reg in ((0.3, 1.0), (1e-2, 1e2), (1e-3, 1e4), (1e-4, 1e6)):
    res = {}
    for dt in ("f32", "f64"):
        b = bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive()), dtype=dt)
        s = SLM(b); s.obj_ = -np.inf; s._state = s._make_state(X, y)
        f, (gv, gr, gh) = s._elbo(X, y, var, reg, ls)
        s._state.release(); s._state = None
        res[dt] = (f, gv, gr, np.asarray(gh))
    a, c = res["f32"], res["f64"]
    print("var %g reg %g: obj rel %.1e dvar rel %.1e dreg rel %.1e dhyp normwise %.1e" % (
        var, reg, abs(a[0]-c[0])/abs(c[0]), abs(a[1]-c[1])/abs(c[1]), abs(a[2]-c[2])/abs(c[2]), np.abs(a[3]-c[3]).max()/np.abs(c[3]).max()))
