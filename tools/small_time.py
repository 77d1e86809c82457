"""Wall time of one svgp_model_update + svgp_elbo (and value-and-gradient) at small sizes: the launch-latency floor."""
import os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd"))
import numpy as np
from approxgp import _ffi
from approxgp.synthetic import synth_arrays
ctx = _ffi.Context(0)
for (N, M, d) in ((1000, 32, 1), (10000, 20, 1), (4096, 128, 8), (4096, 512, 8), (16384, 1024, 8)):
    a = synth_arrays(1, N, M, d)
    desc, keep = _ffi.make_desc(np.float64, 0, a["variance"], a["inv_lengthscale"], a["z"], a["m"], a["Lq"], a["jitter"], likelihood=0, lik_sigma2=a["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, a["x"], a["y"], np.float64)
    for name, fn in (("elbo", lambda: model.elbo(data, 0, N, float(N))[0]), ("update+elbo", lambda: (model.update(desc, keep), model.elbo(data, 0, N, float(N)))[1][0]),
                     ("elbo_grad", lambda: model.elbo_grad(data, 0, N, float(N))[0])):
        fn(); ts = []
        for _ in range(30):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        t = ctx.timing()
        print(f"N={N} M={M} d={d} {name}: median {np.median(ts)*1e6:.0f} us  min {min(ts)*1e6:.0f} us  (device: prep {t.ms_prep*1e3:.0f} us, strip {t.ms_strip*1e3:.0f} us)")
    model.free(); data.free()
