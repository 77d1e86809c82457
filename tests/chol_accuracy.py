"""Accuracy of the device Cholesky factor (svgp_posterior's Lk) against numpy's on the oracle's Kuu: backward error
||L L' - Kuu||_F / ||Kuu||_F and the largest elementwise deviation from the fp64 LAPACK factor, for the library named by
SVGP_MI355X_LIB (A/B builds of the block factorisation must not move these)."""
import os, sys
R = os.path.dirname(os.path.abspath(__file__))
for p in ("approximategps.jl_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(R, "..", p))
import numpy as np, svgp_oracle as o
from approxgp import _ffi
from helpers import device_model
ctx = _ffi.Context(0)
tag = os.environ.get("SVGP_MI355X_LIB", "default").split("/")[-1]
for dt in (np.float64, np.float32):
    for M, d in ((128, 2), (1024, 8), (1990, 4)):
        x, y, sva, s2 = o.synth_problem(510 + M, 2000, M, d, dtype=dt)
        model = device_model(ctx, sva, dtype=dt, sigma2=s2)
        Lk, alpha, B = model.posterior()
        K = o.kuu(sva)
        Lref = np.linalg.cholesky(K)
        L = np.tril(np.asarray(Lk, dtype=np.float64))
        back = np.linalg.norm(L @ L.T - K) / np.linalg.norm(K)
        print(f"{tag} {np.dtype(dt).name} M={M}: backward error {back:.2e}  max|L - Lref| / max|Lref| {np.abs(L - Lref).max() / np.abs(Lref).max():.2e}  cond(Kuu) {np.linalg.cond(K):.1e}", flush=True)
        model.free()
