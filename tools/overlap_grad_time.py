"""Training-step wall clock (svgp_elbo_grad) with the strips beside the factorisation (SVGP_OVERLAP=1) and behind it (=0)."""
import os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd"))
import numpy as np, bench
from approxgp import _ffi
dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
shapes = [(4096, 512), (16384, 1024), (8192, 1024), (32768, 1024), (4096, 2048), (16384, 2048)]
ctx = _ffi.Context(0)
for n, M in shapes:
    p = bench.synth(7, n, M, 8, bench.SE, bench.GAUSS, dtype)
    desc, keep = _ffi.make_desc(p["np_dt"], bench.SE, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    for k in ("0", "1"):
        os.environ["SVGP_OVERLAP"] = k
        for _ in range(4):
            model.elbo_grad(data, 0, n, float(n))
    ts = {"0": [], "1": []}; vals = {}
    for rep in range(30):
        for k in ("0", "1"):
            os.environ["SVGP_OVERLAP"] = k
            t0 = time.perf_counter(); v = model.elbo_grad(data, 0, n, float(n))[0]; ts[k].append(time.perf_counter() - t0); vals[k] = v
    m0, m1 = np.median(ts["0"]) * 1e3, np.median(ts["1"]) * 1e3
    print(f"{dtype} grad n={n:6d} M={M:5d}: serial {m0:.3f} ms  beside {m1:.3f} ms  {100 * (m1 / m0 - 1):+.1f} %  bitwise {'same' if vals['0'] == vals['1'] else 'DIFFERENT'}", flush=True)
    model.free(); data.free()
