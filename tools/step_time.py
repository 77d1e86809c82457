"""Wall clock of one TRAINING STEP as a host sees it: svgp_model_update (new z, m, Lq from host memory) + svgp_elbo_grad (gradients back
to host memory), against the device time of the same step (HIP events): tools/step_time.py [f64|f32]"""
import os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd"))
import numpy as np, bench
from approxgp import _ffi
dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
shapes = [(16384, 512), (16384, 1024), (4096, 2048), (16384, 2048)]
ctx = _ffi.Context(0)
for n, M in shapes:
    p = bench.synth(7, n, M, 8, bench.SE, bench.GAUSS, dtype)
    desc, keep = _ffi.make_desc(p["np_dt"], bench.SE, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    tu, tg, td = [], [], []
    reuse = os.environ.get("STEP_REUSE_OUT", "1") == "1"
    g = None
    for it in range(24):
        t0 = time.perf_counter(); model.update(desc, keep); t1 = time.perf_counter()
        g = model.elbo_grad(data, 0, n, float(n), **({"out": g} if (reuse and g is not None) else {}))[2]; t2 = time.perf_counter()
        if it >= 4:
            tu.append(t1 - t0); tg.append(t2 - t1); td.append(ctx.timing().ms_total)
    print(f"{dtype} n={n} M={M}: update {np.median(tu)*1e3:.3f} ms  elbo_grad {np.median(tg)*1e3:.3f} ms (device {np.median(td):.3f})  step {np.median(np.add(tu, tg))*1e3:.3f} ms", flush=True)
    model.free(); data.free()
