import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd")); sys.path.insert(0, ROOT)
import bench
from approxgp import _ffi
ctx = _ffi.Context(0)
for M in (1152, 1280, 1408, 1536, 1792, 2048):
    p = bench.synth(3, 4096, M, 8, 0, 0, "f64")
    desc, keep = _ffi.make_desc(p["np_dt"], 0, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=0, lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    try:
        print(M, model.elbo(data, 0, 4096, 4096.0)[0], flush=True)
    except Exception as e:
        print(M, "FAIL", str(e)[:80], flush=True)
    model.free(); data.free()
