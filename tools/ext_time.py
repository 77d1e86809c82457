"""Times svgp_elbo_grad (built-in likelihood) against svgp_elbo_grad_ext (host-supplied point gradients) for a bench config."""
import os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd"))
import numpy as np, bench
from approxgp import _ffi
cfg = sys.argv[1] if len(sys.argv) > 1 else "H"
n, M, d, family, lik, dtype, cid = bench.CONFIGS[cfg]
p = bench.synth(cid, n, M, d, family, lik, dtype)
ctx = _ffi.Context(0)
desc, keep = _ffi.make_desc(p["np_dt"], family, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=lik, lik_sigma2=p["sigma2"])
model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
t0 = time.perf_counter(); mu, var = model.marginals(data, 0, n); t_m = time.perf_counter() - t0
y = np.asarray(p["y"], dtype=np.float64); s2 = p["sigma2"]
gmu, gv = (y - mu) / s2, np.full(n, -0.5 / s2)
sum_e = float(np.sum(-0.5 * (np.log(2 * np.pi * s2) + ((y - mu) ** 2 + var) / s2)))
for name, fn in (("builtin", lambda: model.elbo_grad(data, 0, n, float(n))), ("ext", lambda: model.elbo_grad(data, 0, n, float(n), ext=(sum_e, gmu, gv)))):
    fn(); ts = []; tk = []
    for _ in range(3):
        t0 = time.perf_counter(); v = fn()[0]; ts.append(time.perf_counter() - t0); tk.append(ctx.timing().ms_strip)
    print(f"{cfg} {name}: wall {min(ts)*1e3:.1f} ms  device (strips..tail) {min(tk):.1f} ms  value {v:.6f}")
print(f"marginals: {t_m*1e3:.1f} ms")
