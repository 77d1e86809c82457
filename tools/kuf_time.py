"""Kuf assembly alone for a bench config: median / p95 over 32 launches after a warm-up (HIP events of the library)."""
import os, sys
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd"))
import numpy as np, bench
from approxgp import _ffi
ctx = _ffi.Context(0)
print("lib:", os.environ.get("SVGP_MI355X_LIB", "default").split("/")[-1])
for cfg in sys.argv[1:] or ["H"]:
    n, M, d, family, lik, dtype, cid = bench.CONFIGS[cfg]
    p = bench.synth(cid, n, M, d, family, lik, dtype)
    desc, keep = _ffi.make_desc(p["np_dt"], family, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=lik, lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    for _ in range(3): model.elbo(data, 0, n, float(n))          # clocks up
    ts = []
    for _ in range(34):
        model.kuf(data, 0, n, fetch=False); ts.append(ctx.timing().ms_kuf)
    ts = np.array(ts[2:]); es = 8 if dtype == "f64" else 4
    b = es * (M * n + n * d + M * d)
    print(f"{cfg}: kuf median {np.median(ts):.3f} ms = {b/np.median(ts)/1e6:.0f} GB/s ({b/np.median(ts)/1e6/8000:.3f} of 8 TB/s), p95 {np.percentile(ts,95):.3f} ms, min {ts.min():.3f} ms")
    model.free(); data.free()
