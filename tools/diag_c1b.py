import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.slm import StandardLinearModel as SLM
g = np.load("tests/golden/fit_c1.npz")
import importlib.util
spec = importlib.util.spec_from_file_location("t", "tests/test_gpu_parity_r2.py"); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
X, y, Xs = t.c1_data()
def smse(a, b): return ((a - b) ** 2).mean() / a.var()
for it in (20, 40, 80, 200):
    basis = bs.RandomRBF(nbases=256, Xdim=8, random_state=41, lenscale=Parameter(2.0, Positive()), regularizer=Parameter(10.0, Positive()))
    slm = SLM(basis, var=Parameter(0.02, Positive()), nstarts=0, maxiter=it, random_state=0).fit(X, y)
    Ey, Vy = slm.predict_moments(Xs)
    print(os.environ.get("RR_PASS2_NO_FUSE"), it, "obj %.3f" % slm.obj_, "smse_true %.4f ref %.4f" % (smse(g["c1_ys_true"], Ey), smse(g["c1_ys_true"], g["c1_Ey"])), "hyp", slm.hypers_, "var", slm.var_, "reg", slm.regularizer_, flush=True)
