#!/usr/bin/env python3
"""Where does the WORKER thread's time of a config-5 SVI step go (default route: the reference's random stream)?  Wall-clock
per piece of `_ahead` (draws, their upload, the batch's gathers, the likelihood constants) and of the batch generator."""
import os, sys, time, logging, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import revrand_amd.basis_functions as bs
from revrand_amd import likelihoods as lk, optimize, _hip
from revrand_amd.btypes import Parameter, Positive
from revrand_amd.glm import GeneralizedLinearModel
logging.getLogger("revrand_amd").setLevel(logging.ERROR)
N, d, n, K, L, M = 2_000_000, 32, 1024, 10, 50, 65536
rng = np.random.default_rng(5)
X = rng.standard_normal((N, d), dtype=np.float32)
y = rng.poisson(np.exp(0.3 * X[:, 0].astype(np.float64))).astype(np.float64)
T = collections.defaultdict(float)
def timed(obj, name, key=None):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[key or name] += time.perf_counter() - t; return r
    setattr(obj, name, w)
timed(GeneralizedLinearModel, "_reference_draws"); timed(GeneralizedLinearModel, "_draw_ahead"); timed(GeneralizedLinearModel, "_ahead")
timed(GeneralizedLinearModel, "_elbo"); timed(bs.MinibatchFeatures, "prefetch_batch"); timed(lk.Poisson, "device_spec")
timed(bs.MinibatchFeatures, "glm_step_draws"); timed(bs.MinibatchFeatures, "assemble_idx"); timed(bs.MinibatchFeatures, "glm_basis_grads")
orig_gen = optimize.gen_batch
def gen(*a, **k):
    it = orig_gen(*a, **k)
    while True:
        t = time.perf_counter()
        try: b = next(it)
        except StopIteration: return
        T["gen_batch.next"] += time.perf_counter() - t
        yield b
optimize.gen_batch = gen
def run(iters):
    g = GeneralizedLinearModel(lk.Poisson(), bs.RandomRBF(nbases=n, Xdim=d, random_state=1, lenscale=Parameter(np.ones(d), Positive())),
                               K=K, nsamples=L, batch_size=M, maxiter=iters, nstarts=0, random_state=2, sampler=sys.argv[1] if len(sys.argv) > 1 else "host")
    t = time.perf_counter(); g.fit(X, y); return time.perf_counter() - t
if os.environ.get("SWITCH"): sys.setswitchinterval(float(os.environ["SWITCH"]))
run(4); T.clear()
t8 = run(8); T8 = dict(T); T.clear()
t48 = run(48)
print("fit step %.2f ms" % (1e3 * (t48 - t8) / 40))
for k in sorted(T): print("  %-22s %.2f ms per step" % (k, 1e3 * (T[k] - T8.get(k, 0)) / 40))
