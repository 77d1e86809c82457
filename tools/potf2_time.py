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
for dt in os.environ.get("POTF2_DTYPES", "f64,f32").split(","):   # the stamps printed below are those of the LAST dtype
    p = bench.synth(3, 4096, int(os.environ.get("POTF2_M", "128")), 8, 0, 0, dt)   # POTF2_M=256: the stamps are those of the FUSED launch's block (the last one factored)
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

import ctypes
L = ctypes.CDLL(os.environ.get("SVGP_MI355X_LIB", _ffi.LIB_PATH))
if hasattr(L, "svgp_debug_potf2_stamps"):
    buf = (ctypes.c_ulonglong * 128)()
    L.svgp_debug_potf2_stamps(buf)
    s = np.array(list(buf), dtype=np.float64)
    tot = s[41] - s[0]
    print(f"clock64 ticks: total {tot:.0f}; load {s[1]-s[0]:.0f}; first factor {s[2]-s[1]:.0f}; store {s[41]-s[40]:.0f}")
    print(f"  first 16 x 16 factor alone (wave 0): {s[42]-s[1]:.0f}")
    for p in range(8):
        b = 2 + 4 * p
        extra = f"  [wave 0: tile update {s[44+2*p]-s[b+1]:.0f}, factor16 {s[45+2*p]-s[44+2*p]:.0f}, wait at the barrier {s[b+2]-s[45+2*p]:.0f}]" if p < 7 else ""
        w = s[64 + 4 * p: 64 + 4 * p + 4]
        extra += f"  [wave 1: start +{w[0]-s[b+1]:.0f}, items {w[2]-w[0]:.0f}, wait {s[b+2]-w[2]:.0f}]"
        print(f"  block {p}: panel {s[b+1]-s[b]:.0f}  lookahead phase (next factor | trailing + inverse row) {s[b+2]-s[b+1]:.0f}" + extra)
