"""Times svgp_elbo_grad vs svgp_elbo for a bench config."""
import os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd"))
import numpy as np, bench
from approxgp import _ffi
cfg = sys.argv[1] if len(sys.argv) > 1 else "H"
n, M, d, family, lik, dtype, cid = bench.CONFIGS[cfg]
p = bench.synth(cid, n, M, d, family, lik, dtype)
ctx = _ffi.Context(0)
desc, keep = _ffi.make_desc(p["np_dt"], family, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=lik, lik_sigma2=p["sigma2"], neg_var_policy=_ffi.NEGVAR_CLAMP)
model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
for name, fn in (("elbo", lambda: model.elbo(data, 0, n, float(n))[0]), ("elbo_grad", lambda: model.elbo_grad(data, 0, n, float(n))[0])):
    fn(); ts = []
    for _ in range(3):
        t0 = time.perf_counter(); v = fn(); ts.append(time.perf_counter() - t0)
    print(f"{cfg} {name}: {min(ts)*1e3:.3f} ms  value {v:.6f}")
