"""Accuracy of the f32 hyper-gradient (stored vs fused second pass) against the float64 pipeline, at several points."""
import os, sys, subprocess, json, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    from revrand_amd.slm import StandardLinearModel as SLM
    import importlib.util
    spec = importlib.util.spec_from_file_location("t", "tests/test_gpu_parity_r2.py"); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
    X, y, Xs = t.c1_data()
    out = {}
    for dtype in ("f32", "f64"):
        for (var, reg, hyp) in [(0.0334, 10.0, 1.4387), (0.02, 10.0, 2.0), (0.27, 10.0, 0.893), (0.1, 3.0, 1.2), (0.05, 10.0, 0.5)]:
            bt = bs.RandomRBF(nbases=256, Xdim=8, random_state=41, lenscale=Parameter(2.0, Positive()), regularizer=Parameter(10.0, Positive()), dtype=dtype)
            one = SLM(bt); one.obj_ = -np.inf; one._state = one._make_state(X, y)
            nelbo, (ndvar, ndreg, ndhyp) = one._elbo(X, y, var, reg, hyp)
            one._state.release()
            out["%s %g %g %g" % (dtype, var, reg, hyp)] = [float(nelbo), float(ndvar), float(np.atleast_1d(ndreg)[0]), float(np.atleast_1d(ndhyp)[0])]
    print("JSON" + json.dumps(out))
else:
    res = {}
    for nf in ("1", "0"):
        env = dict(os.environ, RR_PASS2_NO_FUSE=nf)
        o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
        res[nf] = json.loads([l for l in o.splitlines() if l.startswith("JSON")][0][4:])
    for k in res["1"]:
        if k.startswith("f32"):
            k64 = "f64" + k[3:]
            ref = res["1"][k64]
            a, b = res["1"][k], res["0"][k]
            print(k, "dhyp f64 %.6g | stored err %.2e | fused err %.2e | dvar err %.1e %.1e" % (ref[3], abs(a[3] - ref[3]) / abs(ref[3]), abs(b[3] - ref[3]) / abs(ref[3]), abs(a[1] - ref[1]) / abs(ref[1]), abs(b[1] - ref[1]) / abs(ref[1])))
