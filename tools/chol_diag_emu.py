"""NumPy model of rr_chol_diag_pipe_kernel's schedule (revrand_amd/csrc/rr_posdef.hip): the 128 x 128 diagonal block in the
registers of a 16 x 16 thread grid (8 x 8 elements per thread, slot (i, j) = element (16 i + ty, 16 j + tx)), the pivot row
through a double-buffered LDS row, step q's rank-1 update split into the slot that holds row q + 1 (done first, so that
row q + 1 can be published) and the rest (done in step q + 1, under its LDS reads and pivot chain); the strict lower slots
carry the forward substitution U^T T = I.  Prints the error of U and of U^-1 against numpy.linalg; the kernel was written
from this model, and tests/test_host_logic.py runs it."""
import numpy as np
rs = np.random.RandomState(0)
M = rs.randn(128, 300); A = M @ M.T / 300 + np.eye(128) * 0.5
ty = np.arange(16)[:, None] * np.ones((1, 16), int); tx = np.ones((16, 1), int) * np.arange(16)[None, :]
a = np.zeros((8, 8, 16, 16))
for i in range(8):
    for j in range(8):
        r = 16 * i + ty; c = 16 * j + tx
        a[i, j] = np.where(c >= r, A[r, c], 0.0)
rowbuf = np.zeros((2, 128))
def owner_write(q):
    qi, qk = q >> 4, q & 15
    for j in range(8):
        rowbuf[q & 1, 16 * j + np.arange(16)] = a[qi, j, qk, :]
def chain(q):
    rb = rowbuf[q & 1]
    app = rb[q]; dp = np.sqrt(app); inv = 1.0 / dp
    ur = [rb[16 * i + ty] * inv for i in range(8)]
    vc = [np.where(16 * j + tx == q, inv, rb[16 * j + tx] * inv) for j in range(8)]
    return dp, inv, ur, vc
def finalize(q, dp, vc):
    qi, qk = q >> 4, q & 15
    for j in range(8):
        c = 16 * j + tx
        new = np.where(c == q, dp, vc[j])
        a[qi, j] = np.where(ty == qk, new, a[qi, j])
def early(q, ur, vc):
    qi, qk = q >> 4, q & 15
    if qk < 15:
        e = qi
        ue = np.where(ty > qk, ur[e], 0.0)
        for j in range(8):
            if j < qi: v = vc[j]
            elif j == qi: v = np.where((tx <= qk) | (tx >= ty), vc[j], 0.0)
            else: v = vc[j]
            a[e, j] -= ue * v
    else:
        e = qi + 1
        if e > 7: return
        for j in range(8):
            if j <= qi: v = vc[j]
            elif j == e: v = np.where(tx >= ty, vc[j], 0.0)
            else: v = vc[j]
            a[e, j] -= ur[e] * v
def bulk(q, ur, vc):  # step p = q - 1 on slots > q >> 4
    qi, qk = q >> 4, q & 15
    vq = np.where(tx < qk, vc[qi], 0.0)
    for i in range(qi + 1, 8):
        for j in range(8):
            if j < qi: v = vc[j]
            elif j == qi: v = vq
            elif j < i: continue
            elif j == i: v = np.where(tx >= ty, vc[j], 0.0)
            else: v = vc[j]
            a[i, j] -= ur[i] * v
owner_write(0)
dp, inv, ur, vc = chain(0); finalize(0, dp, vc); early(0, ur, vc); owner_write(1)
for q in range(1, 128):
    dpn, invn, urn, vcn = chain(q)      # reads
    bulk(q, ur, vc)
    finalize(q, dpn, vcn)
    if q < 127:
        early(q, urn, vcn); owner_write(q + 1)
    ur, vc = urn, vcn
U = np.zeros((128, 128)); Ui = np.zeros((128, 128))
for i in range(8):
    for j in range(8):
        r = 16 * i + ty; c = 16 * j + tx
        U[r, c] = np.where(c >= r, a[i, j], 0.0)
        Ui[c, r] = np.where(c < r, a[i, j], np.where(c == r, 1.0 / a[i, j], 0.0))
Ur = np.linalg.cholesky(A).T
ERR_U, ERR_UINV = np.abs(U - Ur).max(), np.abs(Ui - np.linalg.inv(Ur)).max()
if __name__ == "__main__":
    print("U err", ERR_U, "Uinv err", ERR_UINV, np.abs(Ui).max())
