"""Measurement behind the fp32 gradient tolerances (not a test): device fp32 error per gradient block against the fp64 oracle,
beside the sensitivity of the SAME gradient to the smallest perturbation an fp32 evaluation cannot avoid - every kernel-matrix
entry rounded once to fp32 (oracle, fp64 arithmetic otherwise).  usage: python tests/f32_grad_accuracy.py [big]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("approximategps.jl_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(R, p))
import torch, numpy as np, svgp_oracle as o
from approxgp import _ffi
from helpers import device_model

def rounded_kernel_gradient(sva, x, y, **kw):
    orig = o._kappa
    o._kappa = lambda k, r2: orig(k, r2).astype(np.float32).astype(np.float64)
    try:
        return o.elbo_grad(sva, x, y, **kw)
    finally:
        o._kappa = orig

if __name__ == "__main__":
    ctx = _ffi.Context(0)
    cases = [(2100, 3000, 4, o.KERNEL_SE, 1e-3), (2048, 20000, 16, o.KERNEL_MATERN52, 1e-3), (1024, 2500, 8, o.KERNEL_SE, None), (512, 4000, 8, o.KERNEL_SE, None)]
    if len(sys.argv) > 1:
        cases.append((8192, 800, 8, o.KERNEL_SE, None))
    e = lambda a, b: float(np.abs(np.asarray(a, dtype=float).reshape(np.shape(b), order="F") - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-12))
    for M, N, d, fam, jit in cases:
        kw = {} if jit is None else {"jitter": jit}
        x, y, sva, s2 = o.synth_problem(88, N, M, d, dtype=np.float32, family=fam, **kw)
        model = device_model(ctx, sva, dtype=np.float32, sigma2=s2); data = _ffi.DeviceData(ctx, x, y, np.float32)
        vr, gr = o.elbo_grad(sva, x, y, sigma2=s2, num_data=2.0 * N)
        vp, gp = rounded_kernel_gradient(sva, x, y, sigma2=s2, num_data=2.0 * N)
        v, _, g = model.elbo_grad(data, 0, N, 2.0 * N)
        print(f"M={M} N={N} d={d} jitter={sva.jitter}: value dev {abs(v - vr) / abs(vr):.2e} model {abs(vp - vr) / abs(vr):.2e}")
        for k in ("m", "Lq", "inv_lengthscale", "z"):
            print(f"   {k:16s} device {e(g[k], gr[k]):.3e}   one-rounding model {e(gp[k], gr[k]):.3e}   ratio {e(g[k], gr[k]) / max(e(gp[k], gr[k]), 1e-30):.2f}")
        print(f"   {'variance':16s} device {abs(g['variance'] - gr['variance']) / abs(gr['variance']):.3e}   model {abs(gp['variance'] - gr['variance']) / abs(gr['variance']):.3e}", flush=True)
        model.free(); data.free()
