"""GPU parity of FastFoodRBF (transform / _makeVX / grad / gram) and `hadamard` against the
reference's golden vectors and the NumPy oracle."""
import numpy as np
import pytest

import revrand_oracle as orc
from conftest import normwise

pytestmark = pytest.mark.gpu
TOL = {"f32": 1e-3, "f64": 1e-5}


def _ff(d, nb, ard, dtype, seed=3):
    import revrand_amd.basis_functions as bs
    from revrand_amd.btypes import Parameter, Positive
    if ard:
        return bs.FastFoodRBF(nbases=nb, Xdim=d, random_state=seed, dtype=dtype,
                              lenscale=Parameter(np.ones(d), Positive()))
    return bs.FastFoodRBF(nbases=nb, Xdim=d, random_state=seed, dtype=dtype)


def test_hadamard_golden(golden):
    from revrand_amd.linalg import hadamard
    g = golden("hadamard")
    # the reference's own doctest vector (mathfun/linalg.py:202-206) -- exact in binary
    assert np.array_equal(hadamard(g["doctest_in"], ordering=False), g["doctest_nat"])
    assert np.array_equal(hadamard(g["doctest_in"], ordering=True), g["doctest_seq"])
    assert normwise(hadamard(g["Y"], ordering=False), g["nat"]) < 1e-14
    assert normwise(hadamard(g["Y"], ordering=True), g["seq"]) < 1e-14
    for L in (1, 2, 64, 128):
        assert normwise(hadamard(g["Y%d" % L], ordering=False), g["nat%d" % L]) < 1e-14
    Y = np.random.RandomState(0).randn(300, 1024).astype(np.float32)
    assert normwise(hadamard(Y, ordering=False), orc.hadamard(Y, False)) < 1e-5
    # involution: H(H(y)) = y / n
    assert normwise(hadamard(hadamard(Y, False), False) * 1024, Y) < 1e-5
    with pytest.raises(AssertionError):
        hadamard(np.zeros((2, 12)))


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("case", [(1, 10), (2, 10), (5, 16), (16, 64), (128, 256)])
def test_fastfood_golden(golden, case, dtype):
    d, nb = case
    g = golden("fastfood")
    k = "d%d_nb%d" % (d, nb)
    b = _ff(d, nb, False, dtype)
    # host sampling reproduces the reference's B, G, PI, S bit-for-bit (draw order B->G->PI->S)
    assert np.array_equal(b.B, g[k + "_B"]) and np.array_equal(b.PI, g[k + "_PI"])
    assert np.array_equal(b.G, g[k + "_G"]) and normwise(b.S, g[k + "_S"]) < 1e-14
    X = g[k + "_X"]
    assert normwise(b._makeVX(X), g[k + "_VX"]) < TOL[dtype] * 1e-2
    for tag, ls in [("iso0.7", 0.7), ("iso2.0", 2.0)]:
        P, dP = b.transform(X, ls), b.grad(X, ls)
        assert P.dtype == np.float64 and P.shape == g["%s_%s_Phi" % (k, tag)].shape
        assert normwise(P, g["%s_%s_Phi" % (k, tag)]) < TOL[dtype]
        assert normwise(dP, g["%s_%s_dPhi" % (k, tag)]) < TOL[dtype]
    if k + "_ard_Phi" in g:
        ba = _ff(d, nb, True, dtype)
        ls = np.linspace(0.5, 2.0, d)
        assert normwise(ba.transform(X, ls), g[k + "_ard_Phi"]) < TOL[dtype]
        dP = ba.grad(X, ls)
        assert dP.shape == g[k + "_ard_dPhi"].shape
        assert normwise(dP, g[k + "_ard_dPhi"]) < TOL[dtype]


@pytest.mark.parametrize("shape", [(1000, 3, 7), (513, 20, 100), (300, 64, 200), (257, 100, 256), (200, 128, 1024)])
def test_fastfood_vs_oracle(shape):
    N, d, nb = shape
    rs = np.random.RandomState(N)
    X = rs.randn(N, d).astype(np.float32)
    b = _ff(d, nb, True, "f32", seed=11)
    ls = np.linspace(0.6, 1.7, d)
    B, G, PI, S = orc.fastfood_matrices(nb, d, 11)
    ref = orc.fastfood_transform(X.astype(np.float64), B, G, PI, S, ls)
    assert normwise(b.transform(X, ls), ref) < 1e-3
    y = rs.randn(N).astype(np.float32)
    Gm, bv, yty = b.gram(X, y, ls)
    Gr, br, tr = orc.gram_stats(ref, y.astype(np.float64))
    assert normwise(Gm, Gr) < 1e-3 and normwise(bv, br) < 1e-3 and abs(yty - tr) < 1e-5 * tr


def test_fastfood_in_concat_and_slm():
    import revrand_amd.basis_functions as bs
    from revrand_amd.slm import StandardLinearModel
    rs = np.random.RandomState(4)
    X = rs.randn(300, 3)
    y = np.sin(X[:, 0]) + 0.05 * rs.randn(300)
    base = bs.FastFoodRBF(nbases=20, Xdim=3, random_state=1) + bs.LinearBasis(onescol=True)
    P = base.transform(X, 1.0)
    assert P.shape == (300, 2 * 20 + 4) and base.get_dim(X) == P.shape[1]
    slm = StandardLinearModel(base, nstarts=0, maxiter=100).fit(X, y)
    assert ((slm.predict(X) - y) ** 2).mean() < 0.25 * y.var()
