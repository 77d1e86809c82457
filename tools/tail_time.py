#!/usr/bin/env python
"""Times svgp_elbo over several batch lengths (same resident data) — run once with SVGP_TAIL=0 and once with 1 to see
what the half-width tail launch buys.  usage: python tools/tail_time.py [f64|f32] [M]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd")); sys.path.insert(0, ROOT)
import bench
from approxgp import _ffi

dt = sys.argv[1] if len(sys.argv) > 1 else "f64"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = 300_000
p = bench.synth(3, n, M, 8, 0, 0, dt)
ctx = _ffi.Context(0)
desc, keep = _ffi.make_desc(p["np_dt"], 0, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=0, lik_sigma2=p["sigma2"])
model = _ffi.DeviceModel(ctx, desc, keep)
data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
out = []
for ln in (4096, 8192, 16384, 24576, 32768, 40000, 65536, 100000, 131072, 200000, 262144):
    model.elbo(data, 0, ln, float(n))
    ts, val = [], None
    for _ in range(5):
        val, _ = model.elbo(data, 0, ln, float(n))
        ts.append(ctx.timing().ms_strip)
    out.append((ln, float(np.median(ts)), val))
print(os.environ.get("SVGP_TAIL", "1"), dt, M, " ".join(f"{ln}:{t:.3f}ms" for ln, t, _ in out))
print("vals", " ".join(repr(v) for _, _, v in out))
