#!/usr/bin/env python
"""ms_prep at M = 128 (one potf2 launch + Kuu + KL) for the library named by SVGP_MI355X_LIB (ablation builds)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd")); sys.path.insert(0, ROOT)
import bench
from approxgp import _ffi
ctx = _ffi.Context(0)
out = []
for dt in ("f64", "f32"):
    p = bench.synth(3, 4096, 128, 8, 0, 0, dt)
    desc, keep = _ffi.make_desc(p["np_dt"], 0, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=0, lik_sigma2=p["sigma2"], neg_var_policy=_ffi.NEGVAR_CLAMP)
    model = _ffi.DeviceModel(ctx, desc, keep)
    data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    ts = []
    for _ in range(8):
        try:
            model.elbo(data, 0, 4096, 4096.0)
        except Exception:
            pass
        ts.append(ctx.timing().ms_prep)
    out.append(f"{dt} {np.median(ts[2:])*1e3:.1f} us")
print(os.environ.get("SVGP_MI355X_LIB", "default").split("/")[-1], " ".join(out))
