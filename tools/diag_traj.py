"""Every _elbo evaluation of the config-1 fit, stored vs fused second pass: where do they part?"""
import os, sys, subprocess, json, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd import slm as slm_mod
    SLM = slm_mod.StandardLinearModel
    import importlib.util
    spec = importlib.util.spec_from_file_location("t", "tests/test_gpu_parity_r2.py"); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
    X, y, Xs = t.c1_data()
    log = []
    orig = SLM._elbo
    def wrapped(self, X, y, var, reg, hyp):
        r = orig(self, X, y, var, reg, hyp)
        log.append([float(var), float(np.atleast_1d(reg)[0]), float(np.atleast_1d(hyp)[0]), float(r[0]), float(r[1][0]), float(np.atleast_1d(r[1][1])[0]), float(np.atleast_1d(r[1][2])[0])])
        return r
    SLM._elbo = wrapped
    basis = bs.RandomRBF(nbases=256, Xdim=8, random_state=41, lenscale=Parameter(2.0, Positive()), regularizer=Parameter(10.0, Positive()))
    SLM(basis, var=Parameter(0.02, Positive()), nstarts=0, maxiter=20, random_state=0).fit(X, y)
    print("JSON" + json.dumps(log))
else:
    res = {}
    for nf in ("1", "0"):
        env = dict(os.environ, RR_PASS2_NO_FUSE=nf)
        o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        res[nf] = json.loads([l for l in o.stdout.splitlines() if l.startswith("JSON")][0][4:])
    a, b = res["1"], res["0"]
    print(len(a), len(b))
    for i in range(min(len(a), len(b), 40)):
        print(i, "in", ["%.8g" % v for v in a[i][:3]], "|", ["%.8g" % v for v in b[i][:3]])
        print("   out stored", ["%.8g" % v for v in a[i][3:]])
        print("   out fused ", ["%.8g" % v for v in b[i][3:]])
        if max(abs(x - z) / (abs(x) + 1e-300) for x, z in zip(a[i][:3], b[i][:3])) > 1e-3:
            break
