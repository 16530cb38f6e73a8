import sys, os, logging
sys.path.insert(0, ".")
import numpy as np
logging.basicConfig(level=logging.INFO)
import revrand_amd.basis_functions as bs
from revrand_amd.slm import StandardLinearModel as SLM
rs = np.random.RandomState(1)
X = rs.randn(400, 2); y = np.sin(X[:, 0]) + 0.1 * rs.randn(400)
slm = SLM(bs.RandomRBF(nbases=40, Xdim=2, random_state=3), nstarts=0, maxiter=40).fit(X, y)
