"""The far corners of the size range against the oracle: inducing sets of 8 192 (f64), 12 288 and 16 384 (fp32) points - 64 to 128 panels of the
blocked factorisation - and 4 096 inducing points at the largest input dimension (d = 64) with gradient.  A script for the GPU box (the
oracle is the checker; a case takes the oracle tens of seconds): python tests/extreme_sizes.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "approximategps.jl_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import svgp_oracle as o  # noqa: E402
from approxgp import _ffi  # noqa: E402
from helpers import device_model, rel  # noqa: E402

ctx = _ffi.Context(0)
bad = 0
for (M, N, d, dtype, grad) in ((8192, 3000, 8, np.float64, False), (12288, 3000, 8, np.float32, False), (16384, 2000, 3, np.float32, False),
                              (4096, 20000, 64, np.float64, True), (4096, 20000, 64, np.float32, True), (6144, 70000, 16, np.float32, True)):
    t0 = time.time()
    x, y, sva, s2 = o.synth_problem(7000 + M, N, M, d, dtype=dtype)
    model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    vals = {model.elbo(data, 0, N, 2.0 * N)[0] for _ in range(2)}
    v = vals.pop()
    f64 = dtype == np.float64
    line = f"M={M} N={N} d={d} {dtype.__name__}: run-to-run identical {not vals}"
    if grad:
        vg, _, g = model.elbo_grad(data, 0, N, 2.0 * N)
        ref, g_ref = o.elbo_grad(sva, x, y, sigma2=s2, num_data=2.0 * N)
        eL = float(np.abs(np.asarray(g["Lq"], dtype=np.float64) - g_ref["Lq"]).max() / np.abs(g_ref["Lq"]).max())
        gz = np.asarray(g["z"], dtype=np.float64).reshape(g_ref["z"].shape, order="F")
        ez = float(np.abs(gz - g_ref["z"]).max() / np.abs(g_ref["z"]).max())
        eil = float(np.abs(g["inv_lengthscale"] - g_ref["inv_lengthscale"]).max() / np.abs(g_ref["inv_lengthscale"]).max())
        ok = rel(v, ref) < (1e-8 if f64 else 1e-4) and rel(vg, ref) < (1e-8 if f64 else 1e-4) and max(eL, ez, eil) < (1e-6 if f64 else 5e-3)
        line += f" value {rel(v, ref):.1e} grad-value {rel(vg, ref):.1e} Lq {eL:.1e} z {ez:.1e} inv_lengthscale {eil:.1e}"
    else:
        ref = o.elbo(sva, x, y, sigma2=s2, num_data=2.0 * N)
        Lk, _, _ = model.posterior()
        K = o.kuu(sva)
        L = np.tril(np.asarray(Lk, dtype=np.float64))
        back = float(np.linalg.norm(L @ L.T - K) / np.linalg.norm(K))
        ok = rel(v, ref) < (1e-8 if f64 else 1e-4) and back < (1e-14 if f64 else 2e-6)
        line += f" value {rel(v, ref):.1e} |L L' - Kuu| / |Kuu| {back:.1e}"
    ok = ok and not vals
    bad += not ok
    print(line, "ok" if ok else "FAIL", f"({time.time() - t0:.0f} s)", flush=True)
    model.free()
    data.free()
ctx.close()
print("EXTREMES", "OK" if not bad else f"{bad} FAILED")
sys.exit(1 if bad else 0)
