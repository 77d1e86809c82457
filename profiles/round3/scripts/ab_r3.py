"""Same-box A/B against the round-3 library (tools/build_r3.sh): wall clock of svgp_elbo / svgp_elbo_grad for a list of shapes.
usage: [SVGP_MI355X_LIB=.../libsvgp_r3.so] python tools/ab_r3.py"""
import os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd"))
import numpy as np, bench
from approxgp import _ffi
if "r3" in os.environ.get("SVGP_MI355X_LIB", ""):
    _ffi.SYMBOLS.pop("svgp_last_timing_sized", None)   # ABI v4 library
tag = "r3 " if "r3" in os.environ.get("SVGP_MI355X_LIB", "") else "new"
ctx = _ffi.Context(0)
for dtype, n, M in (("f64", 16384, 1024), ("f64", 8192, 1024), ("f64", 32768, 1024), ("f64", 4096, 512), ("f32", 16384, 1024), ("f32", 65536, 1024),
                    ("f64", 100000, 512), ("f32", 262144, 1024)):
    p = bench.synth(7, n, M, 8, bench.SE, bench.GAUSS, dtype)
    desc, keep = _ffi.make_desc(p["np_dt"], bench.SE, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    out = []
    for name, fn in (("elbo", lambda: model.elbo(data, 0, n, float(n))[0]), ("grad", lambda: model.elbo_grad(data, 0, n, float(n))[0])):
        for _ in range(4): fn()
        ts = []
        for _ in range(25):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        out.append(f"{name} {np.median(ts)*1e3:.3f} ms")
    print(f"{tag} {dtype} n={n:6d} M={M:5d}: " + "  ".join(out), flush=True)
    model.free(); data.free()
