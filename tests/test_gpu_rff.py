"""
GPU parity of the random Fourier feature path (Phi, dPhi, fused Gram) through the C ABI,
against (a) the golden vectors produced by the reference and (b) the NumPy oracle on seeded
inputs.  Tolerances follow BASELINE.json: 1e-3 relative for f32 arithmetic, 1e-5 for f64;
Phi crosses zero so "relative" is normwise: max|err| <= tol * max|ref|.
"""
import numpy as np
import pytest

import revrand_oracle as orc
from conftest import normwise

pytestmark = pytest.mark.gpu

TOL = {"f32": 1e-3, "f64": 1e-5}
CLASSES = ["RandomRBF", "RandomLaplace", "RandomCauchy", "RandomMatern32", "RandomMatern52", "OrthogonalRBF"]


def _bs():
    import revrand_amd.basis_functions as bs
    return bs


def _make(cname, d, n, seed, ard, dtype):
    from revrand_amd.btypes import Parameter, Positive
    cls = getattr(_bs(), cname)
    if ard:
        return cls(nbases=n, Xdim=d, random_state=seed, dtype=dtype,
                   lenscale=Parameter(np.ones(d), Positive()))
    return cls(nbases=n, Xdim=d, random_state=seed, dtype=dtype)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("cname", CLASSES)
def test_golden_transform_grad(golden, cname, dtype):
    """Same seeds as the reference -> same W (host sampling) -> Phi / dPhi within tolerance."""
    g = golden("rff")
    for d in ((1, 5, 8) if cname == "RandomRBF" else (5,)):
        X = g["X_d%d" % d]
        seed = int(g["%s_d%d_seed" % (cname, d)])
        for tag, ls in [("iso0.7", 0.7), ("iso2.0", 2.0), ("ard", np.linspace(0.5, 2.0, d))]:
            b = _make(cname, d, 12, seed, tag == "ard", dtype)
            assert np.array_equal(b.W, g["%s_d%d_W" % (cname, d)])  # bit-exact host sampling
            P = b.transform(X, ls)
            dP = b.grad(X, ls)
            Pref = g["%s_d%d_%s_Phi" % (cname, d, tag)]
            dPref = g["%s_d%d_%s_dPhi" % (cname, d, tag)]
            assert P.dtype == np.float64 and P.shape == Pref.shape  # always float64, cos|sin
            assert dP.shape == dPref.shape
            assert normwise(P, Pref) < TOL[dtype], (cname, d, tag)
            assert normwise(dP, dPref) < TOL[dtype], (cname, d, tag)


def test_float32_input_gives_float64_output(golden):
    g = golden("rff")
    X32 = np.random.RandomState(0).randn(24, 5).astype(np.float32)
    b = _make("RandomRBF", 5, 12, 16, False, "f32")
    P = b.transform(X32, 1.3)
    assert P.dtype == np.float64
    assert normwise(P, g["RandomRBF_d5_f32in_iso1.3_Phi"]) < 1e-3


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("shape", [(1, 3, 4), (7, 8, 256), (513, 32, 130), (2000, 21, 300), (64, 64, 64),
                                   (300, 100, 40)])
def test_transform_shapes_vs_oracle(shape, dtype):
    """Ragged N, n not a multiple of 128/256, d not a power of two, d up to 128."""
    N, d, n = shape
    rs = np.random.RandomState(N + d + n)
    X = rs.randn(N, d)
    b = _make("RandomRBF", d, n, 5, True, dtype)
    ls = np.linspace(0.7, 1.9, d)
    assert normwise(b.transform(X, ls), orc.rff_transform(X, b.W, ls)) < TOL[dtype]
    # non-contiguous row view and float32 input take the same path
    Xw = np.zeros((N, d + 3), dtype=np.float32)
    Xw[:, :d] = X
    assert normwise(b.transform(Xw[:, :d], ls), orc.rff_transform(Xw[:, :d], b.W, ls)) < TOL["f32"]


def test_transform_empty_and_errors():
    b = _make("RandomRBF", 4, 8, 0, False, "f32")
    assert b.transform(np.zeros((0, 4))).shape == (0, 16)
    with pytest.raises(ValueError, match="Dimensions of data inconsistent!"):
        b.transform(np.zeros((3, 5)))
    with pytest.raises(ValueError, match="Dimension of input parameter is inconsistent!"):
        b.transform(np.zeros((3, 4)), np.ones(3))


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_grad_ard_large(dtype):
    N, d, n = 130, 9, 70
    rs = np.random.RandomState(3)
    X = rs.randn(N, d)
    b = _make("RandomMatern52", d, n, 5, True, dtype)
    ls = np.linspace(0.7, 1.9, d)
    dP = b.grad(X, ls)
    assert dP.shape == (N, 2 * n, d)
    assert normwise(dP, orc.rff_grad(X, b.W, ls)) < TOL[dtype]
    # iso quirk: dimension 0 only
    bi = _make("RandomMatern52", d, n, 5, False, dtype)
    assert normwise(bi.grad(X, 1.4), orc.rff_grad(X, bi.W, 1.4)) < TOL[dtype]


@pytest.mark.parametrize("shape", [(1, 3, 4), (500, 4, 16), (4099, 8, 128), (3000, 32, 200), (1500, 21, 260),
                                   (70000, 8, 128), (1000, 64, 128)])
def test_gram_vs_oracle(shape):
    """Fused Phi -> (Phi^T Phi, Phi^T y, yty): f32 MFMA, f64 across K-splits."""
    N, d, n = shape
    rs = np.random.RandomState(N + n)
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)).astype(np.float32)
    b = _make("RandomRBF", d, n, 2, True, "f32")
    ls = np.linspace(0.8, 1.6, d)
    G, bv, yty = b.gram(X, y, ls)
    Gr, br, ytyr = orc.rff_gram_chunked(X.astype(np.float64), y.astype(np.float64), b.W, ls)
    assert G.shape == (2 * n, 2 * n)
    assert np.array_equal(G, G.T)  # exactly symmetric, like the reference's syrk
    assert normwise(G, Gr) < 1e-3
    assert normwise(bv, br) < 1e-3
    assert abs(yty - ytyr) < 1e-5 * ytyr
    # tighter, informational bound actually achieved by f32 MFMA + f64 flush
    assert normwise(G, Gr) < 2e-5, normwise(G, Gr)
    # G only (no y)
    G2, b2, t2 = b.gram(X, None, ls)
    assert b2 is None and t2 is None and normwise(G2, Gr) < 1e-3


def test_gram_posterior_weights():
    """1e-3 on posterior weights (BASELINE north star) through the host Cholesky."""
    N, d, n = 20000, 8, 128
    rs = np.random.RandomState(11)
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)).astype(np.float32)
    b = _make("RandomRBF", d, n, 2, False, "f32")
    G, bv, _ = b.gram(X, y, 1.1)
    Gr, br, _ = orc.rff_gram_chunked(X.astype(np.float64), y.astype(np.float64), b.W, 1.1)
    m, C, _ = orc.slm_posterior_from_stats(G, bv, 0.5, np.full(2 * n, 1.0))
    mr, Cr, _ = orc.slm_posterior_from_stats(Gr, br, 0.5, np.full(2 * n, 1.0))
    assert normwise(m, mr) < 1e-3 and normwise(C, Cr) < 1e-3


def test_gram_linearity_and_sharding():
    """Size-independent properties: G(X1 ++ X2) = G(X1) + G(X2); device accumulation over shards."""
    from revrand_amd import _hip
    N, d, n = 6000, 16, 128
    rs = np.random.RandomState(5)
    X = rs.randn(N, d).astype(np.float32)
    y = rs.randn(N).astype(np.float32)
    b = _make("RandomRBF", d, n, 2, False, "f32")
    G, bv, yty = b.gram(X, y, 1.0)
    Ga, ba, ta = b.gram(X[:2500], y[:2500], 1.0)
    Gb, bb, tb = b.gram(X[2500:], y[2500:], 1.0)
    assert normwise(Ga + Gb, G) < 1e-6 and normwise(ba + bb, bv) < 1e-6
    assert abs(ta + tb - yty) < 1e-9 * yty
    # trace identity: sum_j cos^2 + sin^2 = 1  =>  trace(G) = N exactly (up to rounding)
    assert abs(np.trace(G) - N) < 1e-4 * N
    # device-resident accumulation across two shards equals the one-shot result
    h = b._handle()
    dev = h.dev
    F = 2 * n
    dG = dev.zeros(F * F * 8)
    db = dev.zeros((F + 1) * 8)
    for sl in (slice(0, 2500), slice(2500, N)):
        dX = h.upload(X[sl])
        dy = dev.upload_vector(y[sl])
        h.gram_dev(dX, dy, 1.0, dG, db, _hip.ctypes.c_void_p(db.ptr.value + F * 8))
        dev.sync()
    h.symmetrize_dev(dG)
    G2 = dev.download(dG, (F, F), np.float64)
    b2 = dev.download(db, (F + 1,), np.float64)
    assert normwise(G2, G) < 1e-6 and normwise(b2[:F], bv) < 1e-6 and abs(b2[F] - yty) < 1e-9 * yty


@pytest.mark.parametrize("shape", [(1, 3, 4), (500, 4, 16), (4099, 8, 130), (3000, 32, 200), (1500, 21, 260),
                                   (20000, 8, 128), (1000, 64, 64), (700, 100, 70)])
def test_gram_f64_vs_oracle(shape):
    """f64 mode (v_mfma_f64_16x16x4_f64): 1e-5 relative by BASELINE, ~1e-12 in practice."""
    N, d, n = shape
    rs = np.random.RandomState(N + n)
    X = rs.randn(N, d)
    y = np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)
    b = _make("RandomMatern32", d, n, 2, True, "f64")
    ls = np.linspace(0.8, 1.6, d)
    G, bv, yty = b.gram(X, y, ls)
    Gr, br, ytyr = orc.rff_gram_chunked(X, y, b.W, ls)
    assert np.array_equal(G, G.T)
    assert normwise(G, Gr) < 1e-10 and normwise(bv, br) < 1e-10 and abs(yty - ytyr) < 1e-12 * ytyr
    m, C, _ = orc.slm_posterior_from_stats(G, bv, 0.5, np.full(2 * n, 1.0))
    mr, Cr, _ = orc.slm_posterior_from_stats(Gr, br, 0.5, np.full(2 * n, 1.0))
    assert normwise(m, mr) < 1e-5 and normwise(C, Cr) < 1e-5
    # float32 inputs with f64 arithmetic promote exactly like the reference's np.dot
    X32 = X.astype(np.float32)
    G32, _, _ = b.gram(X32, None, ls)
    assert normwise(G32, orc.rff_gram_chunked(X32.astype(np.float64), y, b.W, ls)[0]) < 1e-10


def test_heavy_tailed_weights_at_scale():
    """Cauchy-distributed frequencies (RandomLaplace) at the headline width: phases of ~1e5 revolutions.
    The class defaults to the f32 pipeline with float64 phases; plain f32 stays within tolerance for the lighter-tailed bases."""
    bs = _bs()
    rs = np.random.RandomState(0)
    X = rs.randn(500, 32).astype(np.float32)
    b = bs.RandomLaplace(nbases=2048, Xdim=32, random_state=1)
    assert b.dtype == "f32" and b.phase64 and np.abs(b.W).max() > 1e3
    ref = orc.rff_transform(X, b.W, 1.0)
    assert normwise(b.transform(X, 1.0), ref) < 1e-5
    # the f32 pipeline behind float64 phases holds 1e-3 at nbases = 2048, D = 32 -- checked on the f32
    # feature matrix itself (rr_featmat_put_rff -> rr_features_rowmajor_f32 -> the RR_F32P64 kernel) and on the fused Gram,
    # with a float64 X that float32 cannot represent
    X64 = rs.randn(500, 32)
    ref64 = orc.rff_transform(X64, b.W, 1.0)
    from revrand_amd.basis_functions import MinibatchFeatures
    mf = MinibatchFeatures(b)
    Wp = np.eye(4096)[:, ::16]
    assert normwise(mf.project(X64, [1.0], Wp), ref64[:, ::16]) < 1e-3
    mf.release()
    G, _, _ = b.gram(X64, None, 1.0)
    assert normwise(G, ref64.T @ ref64) < 1e-3
    # what the plain f32 phase would give (the reason for the variant): an order of magnitude outside the tolerance
    plain = bs.RandomRBF(nbases=2048, Xdim=32, random_state=1)
    plain.W = b.W
    assert normwise(plain.transform(X64, 1.0), ref64) > 1e-2
    for cname in ("RandomRBF", "RandomCauchy", "RandomMatern32", "RandomMatern52", "OrthogonalRBF"):
        b = getattr(bs, cname)(nbases=2048, Xdim=32, random_state=1)
        assert b.dtype == "f32"
        assert normwise(b.transform(X, 1.0), orc.rff_transform(X, b.W, 1.0)) < 1e-3, cname


def test_config2_full_size_properties():
    """BASELINE config 2 at full size (RandomRBF F=4096, D=32, N=1M, f32), checked through
    size-independent properties instead of the (hours-long) CPU oracle:
      * cos^2 + sin^2 = 1  =>  G[f,f] + G[n+f,n+f] = N/n for EVERY frequency, trace(G) = N;
      * exact symmetry; linearity over row shards (two halves sum to the whole);
      * a 4096-row slice agrees with the Gram of the GPU `transform` output of the same rows,
        and that slice of Phi agrees with the oracle."""
    from revrand_amd import _hip
    N, d, n = 1_000_000, 32, 2048
    F = 2 * n
    rng = np.random.default_rng(7)
    X = rng.standard_normal((N, d), dtype=np.float32)
    y = np.sin(X @ rng.standard_normal(d, dtype=np.float32)).astype(np.float32)
    b = _make("RandomRBF", d, n, 42, False, "f32")
    h = b._handle()
    dev = h.dev
    dX = h.upload(X)
    dy = dev.upload_vector(y)
    G, bv, yty = h.gram_host(dX, dy, 1.0)
    assert np.array_equal(G, G.T)
    dg = np.diag(G)
    assert np.abs(dg[:n] + dg[n:] - N / n).max() < 2e-5 * (N / n)
    assert abs(np.trace(G) - N) < 1e-6 * N
    assert abs(yty - float(y.astype(np.float64) @ y.astype(np.float64))) < 1e-9 * yty
    # linearity over shards, through the device-resident accumulate-into API
    acc = dev.zeros((F * F + F + 1) * 8)
    base = acc.ptr.value
    half = N // 2
    for (r0, r1) in ((0, half), (half, N)):
        dXs = _hip.DeviceMatrix(dev, _hip.ctypes.c_void_p(dX.ptr.value + r0 * dX.ld * 4), (r1 - r0, d), dX.ld, np.float32)
        dys = _hip.DeviceBuffer(dev, _hip.ctypes.c_void_p(dy.ptr.value + r0 * 4), (r1 - r0) * 4)
        h.gram_dev(dXs, dys, 1.0, _hip.ctypes.c_void_p(base), _hip.ctypes.c_void_p(base + F * F * 8),
                   _hip.ctypes.c_void_p(base + (F * F + F) * 8))
        dev.sync()
        dXs.ptr = None  # views: not owned
        dys.ptr = None
    h.symmetrize_dev(_hip.ctypes.c_void_p(base))
    out = dev.download(acc, (F * F + F + 1,), np.float64)
    assert normwise(out[:F * F].reshape(F, F), G) < 1e-5 and normwise(out[F * F:F * F + F], bv) < 1e-5  # f32 K-splits differ
    # a slice against transform + NumPy, and transform against the oracle
    sl = slice(123_456, 123_456 + 4096)
    P = b.transform(X[sl], 1.0)
    Gs, bs_, _ = b.gram(X[sl], y[sl], 1.0)
    assert normwise(Gs, P.T @ P) < 2e-5 and normwise(bs_, P.T @ y[sl].astype(np.float64)) < 2e-5
    assert normwise(P[:256], orc.rff_transform(X[sl][:256], b.W, 1.0)) < 1e-3


_TORCH_POINTERS = r'''
import sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import revrand_oracle as orc
from revrand_amd import _hip
import revrand_amd.basis_functions as bs
_hip.load_library()            # before torch: RR_HIP_RUNTIME=torch makes it the wheel's runtime, one per process
import torch
assert torch.cuda.is_available(), "torch sees no GPU although the library runs on torch's HIP runtime"
assert "/torch/lib/" in _hip.hip_runtime_path()
N, d, n = 5000, 32, 256
F = 2 * n
rs = np.random.RandomState(0)
X = rs.randn(N, d).astype(np.float32)
y = rs.randn(N).astype(np.float32)
b = bs.RandomRBF(nbases=n, Xdim=d, random_state=3)
h = b._handle()
tX = torch.from_numpy(X).cuda()
ty = torch.from_numpy(y).cuda()
acc = torch.zeros(F * F + F + 1, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
dX = _hip.DeviceMatrix(h.dev, _hip.ctypes.c_void_p(tX.data_ptr()), (N, d), d, np.float32)
dy = _hip.DeviceBuffer(h.dev, _hip.ctypes.c_void_p(ty.data_ptr()), N * 4)
nw = lambda a, r: np.abs(a - r).max() / np.abs(r).max()
try:
    p0 = acc.data_ptr()
    h.gram_dev(dX, dy, 1.0, p0, p0 + F * F * 8, p0 + (F * F + F) * 8)
    h.symmetrize_dev(p0)
    h.dev.sync()
    out = acc.cpu().numpy()
    Gr, br, tr = orc.rff_gram_chunked(X.astype(np.float64), y.astype(np.float64), b.W, 1.0)
    assert nw(out[:F * F].reshape(F, F), Gr) < 2e-5 and nw(out[F * F:F * F + F], br) < 2e-5
    assert abs(out[-1] - tr) < 1e-6 * tr
    # a basis with d = 21 needs 32 padded columns: an unpadded device X is refused, not misread
    b21 = bs.RandomRBF(nbases=64, Xdim=21, random_state=3)
    t21 = torch.zeros(100, 21, dtype=torch.float32, device="cuda")
    bad = _hip.DeviceMatrix(h.dev, _hip.ctypes.c_void_p(t21.data_ptr()), (100, 21), 21, np.float32)
    try:
        b21._handle().gram_dev(bad, None, 1.0, p0)
        raise SystemExit("an unpadded device X was accepted")
    except _hip.HipError as e:
        assert "padded" in str(e), str(e)
    bad.ptr = None
finally:
    dX.ptr = None   # torch owns the memory
    dy.ptr = None
print("TORCH_POINTERS_OK")
'''


def test_external_device_pointers_torch():
    """The *_dev entry points take ANY device memory of the GPU: here torch CUDA tensors.  Also: the padded-layout contract
    is enforced.  In a process of its own with RR_HIP_RUNTIME=torch: since round 3 the library runs on the HIP runtime it was
    built against by default, and a process that imports torch AFTER loading the library must say so (one runtime per
    process)."""
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    from conftest import ROOT
    code = _TORCH_POINTERS % (ROOT, os.path.join(ROOT, "oracle"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RR_HIP_RUNTIME="torch"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "TORCH_POINTERS_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("ard", [False, True])
def test_grad_contract_vs_materialised_gradient(ard):
    """sum(E o dPhi_i) through rr_rff_grad_contract == the same contraction of the oracle's dPhi tensor
    (the GLM's basis-gradient consumer, glm.py:274-275), incl. the isotropic dimension-0 quirk."""
    N, d, n = 3000, 7, 150
    rs = np.random.RandomState(2)
    X = rs.randn(N, d)
    E = rs.randn(N, 2 * n)
    b = _make("RandomCauchy", d, n, 4, ard, "f32")
    ls = np.linspace(0.7, 1.6, d) if ard else 1.3
    got = b.grad_contract(X, E, ls)
    dP = orc.rff_grad(X, b.W, ls)
    want = np.array([(E * dP[:, :, i]).sum() for i in range(d)]) if ard else (E * dP).sum()
    assert np.shape(got) == np.shape(want)
    assert normwise(np.atleast_1d(got), np.atleast_1d(want)) < 1e-3


def test_extreme_lenscale_stays_finite():
    """The optimiser's log-space bounds reach lenscale = 1e-100 (optimize/decorators.py:18); the reference
    returns finite (meaningless) features there and so must the f32 path -- no inf - inf."""
    rs = np.random.RandomState(0)
    X = rs.randn(500, 4)
    y = rs.randn(500)
    b = _make("RandomRBF", 4, 64, 1, True, "f32")
    ls = np.array([1e-100, 3e68, 1e78, 3e58])
    Phi = b.transform(X, ls)
    assert np.isfinite(Phi).all() and np.abs(Phi).max() <= 1 / np.sqrt(64) + 1e-6
    G, bb, _ = b.gram(X, y, ls)
    assert np.isfinite(G).all() and np.isfinite(bb).all()


@pytest.mark.parametrize("N,d,n", [(1, 1, 1), (3, 2, 1), (31, 1, 5), (33, 9, 3), (257, 17, 129)])
def test_tiny_and_ragged_shapes_end_to_end(N, d, n):
    """One row, one frequency, one input dimension, sizes straddling every tile edge: transform, grad, Gram, the
    second pass and predict_moments against the oracle."""
    from revrand_amd.btypes import Parameter, Positive
    rs = np.random.RandomState(N * 1000 + d * 10 + n)
    X = rs.randn(N, d)
    y = rs.randn(N)
    b = _bs().RandomRBF(nbases=n, Xdim=d, random_state=2, lenscale=Parameter(np.ones(d), Positive()))
    ls = np.linspace(0.7, 1.3, d)
    Phi = orc.rff_transform(X, b.W, ls)
    dP = orc.rff_grad(X, b.W, ls)
    assert normwise(b.transform(X, ls), Phi) < 1e-3
    got = b.grad(X, ls)
    assert got.shape == dP.shape and normwise(got, dP) < 1e-3
    G, bv, yty = b.gram(X, y, ls)
    assert normwise(G, Phi.T @ Phi) < 1e-4 and normwise(bv, Phi.T @ y) < 1e-4 and abs(yty - y @ y) < 1e-6 * max(y @ y, 1e-12)
    F = 2 * n
    C = np.linalg.inv(np.eye(F) / 1.3 + Phi.T @ Phi / 0.4)
    m = C @ (Phi.T @ y) / 0.4
    st = b.device_fit_state(X, y)
    sq, dh = st.second_pass(ls, m, C, 0.4)
    st.release()
    err = y - Phi @ m
    slabs = [dP[:, :, i] for i in range(d)] if dP.ndim == 3 else [dP]
    want = np.array([-(m @ (err @ g) - ((g.T @ Phi) * C).sum()) / 0.4 for g in slabs])
    scale = np.array([(abs(m @ (err @ g)) + abs(((g.T @ Phi) * C).sum())) / 0.4 for g in slabs])  # the two terms cancel
    assert abs(sq - err @ err) < 1e-3 * max(err @ err, 1e-9)
    assert np.all(np.abs(np.atleast_1d(dh) - want) < 5e-3 * scale + 1e-6)
    Ey, Vf = b.predict_moments(X, ls, m, C)
    assert normwise(Ey, Phi @ m) < 1e-3 and normwise(Vf, ((Phi @ C) * Phi).sum(axis=1)) < 1e-3


def test_random_shape_sweep_against_oracle():
    """Thirty pseudo-random (N, d, nbases, class, iso/ARD, f32/f64) combinations -- every padded width of the
    kernels (d in 1..128), frequencies not a multiple of any tile, ragged row counts: transform, Gram and Phi^T y
    against the oracle."""
    from revrand_amd.btypes import Parameter, Positive
    rs = np.random.RandomState(2024)
    classes = ["RandomRBF", "RandomCauchy", "RandomMatern32", "RandomMatern52", "OrthogonalRBF"]
    for trial in range(30):
        d = int(rs.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 128]))
        n = int(rs.randint(1, 300))
        N = int(rs.randint(1, 1500))
        ard = bool(rs.randint(2))
        dtype = "f64" if trial % 5 == 4 else "f32"
        cname = classes[trial % len(classes)]
        X = rs.randn(N, d)
        y = rs.randn(N)
        b = _make(cname, d, n, trial, ard, dtype)
        ls = rs.uniform(0.6, 1.8, d) if ard else float(rs.uniform(0.6, 1.8))
        Phi = orc.rff_transform(X, b.W, ls)
        tol = 1e-3 if dtype == "f32" else 1e-9
        assert normwise(b.transform(X, ls), Phi) < tol, (trial, cname, N, d, n)
        G, bv, yty = b.gram(X, y, ls)
        gt = 2e-5 if dtype == "f32" else 1e-10
        assert normwise(G, Phi.T @ Phi) < gt * 10 and normwise(bv, Phi.T @ y) < gt * 10, (trial, cname, N, d, n, dtype)
        assert np.array_equal(G, G.T) and abs(yty - y @ y) <= 1e-6 * (y @ y) + 1e-12
