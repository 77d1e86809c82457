#!/usr/bin/env python
"""ms_prep (Kuu, Cholesky, T panels, KL: HIP events of the library) for several M — the M-sized critical path."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd")); sys.path.insert(0, ROOT)
import bench
from approxgp import _ffi

ctx = _ffi.Context(0)
for dt in ("f64", "f32"):
    row = []
    for M in (128, 512, 1024, 2048, 4096):
        p = bench.synth(3, 4096, M, 8, 0, 0, dt)
        desc, keep = _ffi.make_desc(p["np_dt"], 0, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=0, lik_sigma2=p["sigma2"])
        model = _ffi.DeviceModel(ctx, desc, keep)
        data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
        ts = []
        for _ in range(6):
            val, _ = model.elbo(data, 0, 4096, 4096.0)
            ts.append(ctx.timing().ms_prep)
        row.append(f"M={M}: {np.median(ts[1:]):.3f} ms (elbo {val:.6f})")
        model.free(); data.free()
    print(dt, " | ".join(row))
