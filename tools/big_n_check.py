import os, sys, time
sys.path.insert(0, "/root/repo/approximategps.jl_amd")
import numpy as np
from approxgp import _ffi
from approxgp.synthetic import synth_arrays
ctx = _ffi.Context(0)
N, M, d = 40_000_000, 128, 8
a = synth_arrays(1, 1000, M, d, dtype=np.float32)
rng = np.random.default_rng(0)
x = rng.standard_normal((d, N), dtype=np.float32); y = np.sin(x.sum(0) / np.sqrt(d)).astype(np.float32)
desc, keep = _ffi.make_desc(np.float32, 0, a["variance"], a["inv_lengthscale"], a["z"], a["m"], a["Lq"], a["jitter"], likelihood=0, lik_sigma2=0.3)
model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, x, y, np.float32)
t0 = time.time(); full = model.elbo_partial(data); t1 = time.time()
h = N // 2 + 12345
p1, p2 = model.elbo_partial(data, 0, h), model.elbo_partial(data, h, N - h)
print("full", full, "time", t1 - t0, "halves rel", abs(p1[0] + p2[0] - full[0]) / abs(full[0]))
# host check of a window far beyond 2^31 bytes into the arrays
off = N - 5000
sub = _ffi.DeviceData(ctx, x[:, off:], y[off:], np.float32)
print("tail window", model.elbo_partial(data, off, 5000)[0], model.elbo_partial(sub)[0])
v, t, g = model.elbo_grad(data, 0, N, float(N))
print("grad ok", v, np.abs(g["m"]).max())
