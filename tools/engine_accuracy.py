#!/usr/bin/env python3
"""Accuracy of the Gram engines: max |G - G_f64| / max |G| and trace(G)/N - 1 against the float64
NumPy oracle, per engine, on a few shapes; and the headline-shape difference between engines (no oracle at that size)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import revrand_oracle as orc  # checker only
import revrand_amd.basis_functions as bs
from revrand_amd.btypes import Parameter, Positive
from revrand_amd import _hip

dev = _hip.get_device()
ENGINES = ("f32", "fp16x3", "bf16x4", "bf16x3")
print("%-22s" % "N, d, n" + "".join("%-26s" % e for e in ENGINES))
for (N, d, n) in [(64, 4, 16), (4099, 8, 128), (3000, 32, 200), (70000, 8, 128), (20000, 32, 1000)]:
    rs = np.random.RandomState(N + n)
    X = rs.randn(N, d).astype(np.float32)
    y = (np.sin(X @ rs.randn(d)) + 0.1 * rs.randn(N)).astype(np.float32)
    b = bs.RandomRBF(nbases=n, Xdim=d, random_state=2, lenscale=Parameter(np.ones(d), Positive()))
    ls = np.linspace(0.8, 1.6, d)
    Gr, _, _ = orc.rff_gram_chunked(X.astype(np.float64), y.astype(np.float64), b.W, ls)
    row = "%-22s" % ("%d, %d, %d" % (N, d, n))
    for eng in ENGINES:
        dev.set_gram_engine(eng)
        G, _, _ = b.gram(X, y, ls)
        row += "%-26s" % ("%.2e (trace %.1e)" % (np.abs(G - Gr).max() / np.abs(Gr).max(), abs(np.trace(G) - N) / N))
    print(row)
N, d, n = 2_000_000, 32, 2048
rng = np.random.default_rng(0)
X = rng.standard_normal((N, d), dtype=np.float32)
b = bs.RandomRBF(nbases=n, Xdim=d, random_state=42)
dev.set_gram_engine("f32")
Gf, _, _ = b.gram(X, None, 1.0)
row = "%-22s%-26s" % ("%d, %d, %d vs f32" % (N, d, n), "trace %.1e" % (abs(np.trace(Gf) - N) / N))
for eng in ENGINES[1:]:
    dev.set_gram_engine(eng)
    G, _, _ = b.gram(X, None, 1.0)
    row += "%-26s" % ("%.2e (trace %.1e)" % (np.abs(G - Gf).max() / np.abs(Gf).max(), abs(np.trace(G) - N) / N))
print(row)
dev.set_gram_engine("f32")
