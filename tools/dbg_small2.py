import os, sys, logging, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_golden
import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel as SLM
logging.basicConfig(level=logging.INFO, format="%(message)s")
g = load_golden("fit_c1")
X, y = g["s2_X"], g["s2_y"]
basis = bs.RandomMatern32(nbases=20, Xdim=4, random_state=43, lenscale=Parameter(np.full(4, 1.3), Positive()), regularizer=Parameter(2.0, Positive()))
slm = SLM(basis, var=Parameter(0.4, Positive()), nstarts=0, maxiter=25, random_state=0).fit(X, y)
print("final", slm.obj_)
