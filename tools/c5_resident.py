"""Config 5's `fit` through the resident SVI loop (rr_glm_sgd), for rocprofv3: two fits of different length, the per-step
wall-clock from their difference.  `python tools/c5_resident.py [host|device] [resident|hostloop] [short long]`.
DEVICES=0,0 (or 0,1,...): the same fit with devices=[...] -- the loop over a device group (rr_glm_sgd_group_step)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs  # noqa: E402
from revrand_amd import likelihoods as lk  # noqa: E402
from revrand_amd.btypes import Parameter, Positive  # noqa: E402
from revrand_amd.glm import GeneralizedLinearModel  # noqa: E402

sampler = sys.argv[1] if len(sys.argv) > 1 else "host"
resident = (sys.argv[2] if len(sys.argv) > 2 else "resident") == "resident"
short, long_ = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (8, 72)
N, d, n, K, L, M = 2_000_000, 32, 1024, 10, 50, 65536
rng = np.random.default_rng([20260928, 5])
X = rng.standard_normal((N, d), dtype=np.float32)
y = rng.poisson(np.exp(0.3 * X[:, 0].astype(np.float64))).astype(np.float64)
t = {}
for rep in range(int(os.environ.get("REPS", "2"))):
    if os.environ.get("ALTERNATE"):
        sampler = ("host", "device")[rep % 2]
    for iters in (short, long_):
        g = GeneralizedLinearModel(lk.Poisson(), bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())),
                                   K=K, nsamples=L, batch_size=M, maxiter=iters, nstarts=0, random_state=2, sampler=sampler,
                                   gram_engine=os.environ.get("ENGINE") or None,
                                   devices=[int(v) for v in os.environ["DEVICES"].split(",")] if os.environ.get("DEVICES") else None)
        g._resident_sgd = resident
        marks = {"A": [], "B": []}
        if os.environ.get("STAGES"):   # when did each pipeline stage finish each batch?
            import gc
            gc.callbacks.append(lambda phase, info: marks.setdefault("gc", []).append((phase, info.get("generation"), time.perf_counter())))
            da, ah = g._draw_ahead, g._ahead

            def draw_ahead(batch, upload=False, da=da):
                t0 = time.perf_counter(); r = da(batch, upload); marks["A"].append((t0, time.perf_counter())); return r

            def ahead(batch, draws=True, ah=ah):
                t0 = time.perf_counter(); r = ah(batch, draws); marks["B"].append((t0, time.perf_counter())); return r
            g._draw_ahead, g._ahead = draw_ahead, ahead
        np.random.seed(1)
        t0 = time.perf_counter()
        g.fit(X, y)
        t[iters] = time.perf_counter() - t0
        ck = g.__dict__.get("_resident_clock")
        if ck is not None and len(ck) > 40:
            dt = 1e3 * np.diff(ck[8:])   # (the last interval ends when the queue has drained)
            print("   %d steps: intervals mean %.3f median %.3f p90 %.3f max %.2f ms; over 5 ms: %s"
                  % (iters, dt.mean(), np.median(dt), np.percentile(dt, 90), dt.max(),
                     ["%d:%.1f" % (i + 8, v) for i, v in enumerate(dt) if v > 5.0]))
            for i, v in enumerate(dt):
                if v > 10.0 and os.environ.get("STAGES"):
                    k = i + 8
                    t_a, t_b = ck[k], ck[k + 1]
                    print("      stall before step %d queued [%.1f ms]:" % (k + 1, v))
                    for nm in ("A", "B"):
                        for q, (u0, u1) in enumerate(marks[nm]):
                            if u1 > t_a - 0.005 and u0 < t_b + 0.005:
                                print("        stage %s batch %d: start %+.1f ms, took %.1f ms" % (nm, q, 1e3 * (u0 - t_a), 1e3 * (u1 - u0)))
                    for ph, gen, tt in marks.get("gc", []):
                        if t_a - 0.005 < tt < t_b + 0.005:
                            print("        gc %s gen %s at %+.1f ms" % (ph, gen, 1e3 * (tt - t_a)))
    print("%s sampler, %s: %.3f ms per step (fits of %d and %d steps: %.3f s, %.3f s)"
          % (sampler, "resident loop" if resident else "host loop", 1e3 * (t[long_] - t[short]) / (long_ - short), short, long_, t[short], t[long_]))
