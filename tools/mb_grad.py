"""Value-and-gradient of one minibatch (N points, M inducing) 10 times: the workload to profile for the M-sized tail."""
import os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd"))
import numpy as np
from approxgp import _ffi
from approxgp.synthetic import synth_arrays
N, M, d = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16384, 1024, 8)))
dt = np.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else np.float64
ctx = _ffi.Context(0)
a = synth_arrays(1, N, M, d, dtype=dt)
desc, keep = _ffi.make_desc(dt, 0, a["variance"], a["inv_lengthscale"], a["z"], a["m"], a["Lq"], a["jitter"], likelihood=0, lik_sigma2=a["sigma2"])
model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, a["x"], a["y"], dt)
ts = []
for _ in range(11):
    t0 = time.perf_counter(); model.update(desc, keep); v = model.elbo_grad(data, 0, N, float(10 * N))[0]; ts.append(time.perf_counter() - t0)
t = ctx.timing()
print(f"N={N} M={M} d={d} {np.dtype(dt).name}: update + value-and-gradient median {np.median(ts[1:])*1e3:.3f} ms (device prep {t.ms_prep:.3f}, rest {t.ms_strip:.3f}) value {v:.6f}")
