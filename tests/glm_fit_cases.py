"""The cases of tests/golden/glm_fit.npz (oracle/make_golden.py: gen_glm_fit -- GeneralizedLinearModel.fit of the REFERENCE,
glm.py:141-203, 20 Adam steps, K = 3, nsamples = 6, N = 300, d = 3, nbases = 8 per Fourier child) as plain data, for the oracle's
CPU test and the GPU tests of both SVI loops.

basis: "ard"      RandomRBF, lenscale Parameter(gamma(4, scale=0.25), Positive(), shape=(d,))
       "cat"      LinearBasis(onescol=True) + RandomRBF + RandomMatern52, all defaults (reference: tests/test_models.py:97-99)
       "bound"    RandomRBF, lenscale Parameter(1.0, Bound(0.996, 1.001))  -- no log trick, truncation + clipping (sgd.py:404-420)
       "posupper" RandomRBF, lenscale Parameter(1.0, Positive(1.03))       -- log trick with an upper limit
forward: the reference's structured_sgd drops batch_size on the way to sgd (decorators.py:244-246: its main loop always runs 10
rows); this implementation forwards it.  True = the reference's own code with sgd's default batch size set to the
estimator's; False with batch 64 = the reference unmodified (main loop at 10 rows, B_ = N / 64), which this implementation
does not reproduce by design (only the oracle is held to it)."""

UPDATERS = {  # tag suffix -> (class name in optimize / revrand.optimize.sgd, constructor arguments, oracle name)
    "adadelta": ("AdaDelta", {}, "adadelta"),
    "adagrad": ("AdaGrad", {"eta": 0.05}, "adagrad"),
    "momentum": ("Momentum", {"rho": 0.5, "eta": 1e-4}, "momentum"),
}

CASES = [  # tag, likelihood, basis, batch_size, nstarts, forward   (a tag ending in _<updater>: that updater instead of Adam)
    ("poisson_ard_bs10_ns0", "poisson_exp", "ard", 10, 0, False),
    ("poisson_ard_bs10_ns5", "poisson_exp", "ard", 10, 5, False),
    ("gaussian_cat_bs10_ns5", "gaussian", "cat", 10, 5, False),
    ("gaussian_cat_bs10_ns0", "gaussian", "cat", 10, 0, False),
    ("binomial_cat_bs10_ns3", "binomial", "cat", 10, 3, False),
    ("poisson_ard_bs64f_ns5", "poisson_exp", "ard", 64, 5, True),
    ("gaussian_cat_bs64f_ns0", "gaussian", "cat", 64, 0, True),
    ("poisson_ard_bs64_ns5", "poisson_exp", "ard", 64, 5, False),
    ("poisson_bound_bs10_ns0", "poisson_exp", "bound", 10, 0, False),
    ("gaussian_posupper_bs64f_ns4", "gaussian", "posupper", 64, 4, True),
    ("bernoulli_cat_bs10_ns2", "bernoulli", "cat", 10, 2, False),
    ("poisson_softplus_ard_bs10_ns0", "poisson_softplus", "ard", 10, 0, False),
    ("poisson_ard_bs10_ns0_adadelta", "poisson_exp", "ard", 10, 0, False),
    ("gaussian_cat_bs10_ns3_adagrad", "gaussian", "cat", 10, 3, False),
    ("binomial_cat_bs64f_ns0_momentum", "binomial", "cat", 64, 0, True),
]


def updater_of(tag):
    """(class name, kwargs, oracle name) of a case's updater, or None for the default Adam."""
    return UPDATERS.get(tag.rsplit("_", 1)[-1])

# the cases an implementation that forwards batch_size reproduces (at the reference's default 10 the two agree)
IMPLEMENTED = [c for c in CASES if c[3] == 10 or c[5]]
