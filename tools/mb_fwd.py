"""Median forward wall clock for small batches: tools/mb_fwd.py [f64|f32]"""
import os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd"))
import numpy as np, bench
from approxgp import _ffi
dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
shapes = [(1024, 1024), (4096, 1024), (8192, 1024), (16384, 1024), (32768, 1024), (2048, 768), (4096, 2048)]
ctx = _ffi.Context(0)
out = []
for n, M in shapes:
    p = bench.synth(7, n, M, 8, bench.SE, bench.GAUSS, dtype)
    desc, keep = _ffi.make_desc(p["np_dt"], bench.SE, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    for _ in range(5): model.elbo(data, 0, n, float(n))
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); v = model.elbo(data, 0, n, float(n))[0]; ts.append(time.perf_counter() - t0)
    out.append(f"{n}/{M}: {np.median(ts)*1e3:.3f}")
    model.free(); data.free()
print(dtype, " | ".join(out))
