"""
Host-side dense solves that stay on the CPU by design (BASELINE north star: "host-side
Cholesky"): reference revrand/mathfun/linalg.py:84-179.
"""
import numpy as np
from scipy.linalg import LinAlgError, cho_solve, cholesky, svd

CHOLTHRESH = 1e-5  # linalg.py:31


def solve_posdef(A, b):
    """``(A^-1 b, log|A|)`` for positive semi-definite A (linalg.py:84-125).

    Upper Cholesky first; if it fails or any diagonal of the factor is below CHOLTHRESH,
    fall back to an SVD whose singular values are clamped at 1e-15 (linalg.py:128-179).
    """
    try:
        U = cholesky(A, lower=False)
        if np.any(U.diagonal() < CHOLTHRESH):
            raise LinAlgError("Unstable cholesky factor detected")
        return cho_solve((U, False), b), 2.0 * np.log(U.diagonal()).sum()
    except LinAlgError:
        Us, s, Vt = svd(A)
        half = 1.0 / np.sqrt(np.maximum(s, 1e-15))
        left, right = Us * half, half[:, None] * Vt
        n = A.shape[0]
        m = b.shape[1] if np.ndim(b) > 1 else 1
        X = left.dot(right.dot(b)) if m < n else left.dot(right).dot(b)
        return X, np.log(s).sum()


def hadamard(Y, ordering=True):
    """Fast Walsh-Hadamard transform of each row of Y (rows, 2^p), normalised by 1/n
    (mathfun/linalg.py:182-220); ``ordering=True`` reorders from natural to sequency order.
    Runs on the GPU (``rr_hadamard``)."""
    from . import _hip
    Y = np.asarray(Y)
    n = Y.shape[1]
    assert n & (n - 1) == 0  # required, as in the reference (:211)
    return _hip.hadamard(Y, ordering=ordering)
