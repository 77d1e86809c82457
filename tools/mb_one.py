"""One minibatch shape, a few value-and-gradient calls (for kernel traces): tools/mb_one.py n M [f64|f32]"""
import os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd"))
import numpy as np, bench
from approxgp import _ffi
n, M = int(sys.argv[1]), int(sys.argv[2]); dtype = sys.argv[3] if len(sys.argv) > 3 else "f64"
ctx = _ffi.Context(0)
p = bench.synth(7, n, M, 8, bench.SE, bench.GAUSS, dtype)
desc, keep = _ffi.make_desc(p["np_dt"], bench.SE, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], lik_sigma2=p["sigma2"])
model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
ts = []
for _ in range(12):
    t0 = time.perf_counter(); model.elbo_grad(data, 0, n, float(n)); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.2f}" for t in ts))
