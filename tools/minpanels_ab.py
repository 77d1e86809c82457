"""SVGP_OVERLAP_MIN_PANELS 5 (k=0) / 2 (k=1), interleaved in one process (the knob is read per call): forward and value-and-gradient wall clock.
tools/split_ab.py [f64|f32]"""
import os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..")); sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd"))
import numpy as np, bench
from approxgp import _ffi
dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
shapes = [(512, 512), (1024, 512), (2048, 512), (4096, 512), (8192, 512), (16384, 512), (1024, 384), (4096, 384), (1024, 256), (4096, 256)]
ctx = _ffi.Context(0)
for n, M in shapes:
    p = bench.synth(7, n, M, 8, bench.SE, bench.GAUSS, dtype)
    desc, keep = _ffi.make_desc(p["np_dt"], bench.SE, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    g = {"0": None, "1": None}
    tf = {"0": [], "1": []}; tg = {"0": [], "1": []}; dv = {"0": [], "1": []}
    for rep in range(34):
        for k in ("0", "1"):
            os.environ["SVGP_OVERLAP_MIN_PANELS"] = "5" if k == "0" else "2"
            t0 = time.perf_counter(); model.elbo(data, 0, n, float(n)); t1 = time.perf_counter()
            g[k] = model.elbo_grad(data, 0, n, float(n), **({"out": g[k]} if g[k] is not None else {}))[2]; t2 = time.perf_counter()
            if rep >= 4:
                tf[k].append(t1 - t0); tg[k].append(t2 - t1); dv[k].append(ctx.timing().ms_total)
    f0, f1, g0, g1 = (np.median(v) * 1e3 for v in (tf["0"], tf["1"], tg["0"], tg["1"]))
    print(f"{dtype} n={n:6d} M={M:5d}: forward {f0:.3f} -> {f1:.3f} ms ({100*(f1/f0-1):+.1f} %)   value+gradient {g0:.3f} -> {g1:.3f} ms ({100*(g1/g0-1):+.1f} %)"
          f"   [device, gradient: {np.median(dv['0']):.3f} -> {np.median(dv['1']):.3f}]", flush=True)
    model.free(); data.free()
