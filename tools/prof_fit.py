import sys, os, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from revrand_amd.basis_functions import RandomRBF
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel
rs = np.random.RandomState(0)
N, d, n = 10000, 8, 256
X = rs.randn(N, d); y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
def run():
    slm = StandardLinearModel(RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(1.0, Positive())),
                              var=Parameter(1.0, Positive()), nstarts=0, maxiter=20)
    slm.fit(X, y)
run()
cProfile.run("run()", "/tmp/fit.prof")
pstats.Stats("/tmp/fit.prof").sort_stats("cumulative").print_stats(18)
