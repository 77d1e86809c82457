"""Repeated prep at several M: catches races in the factorisation (PosDef failures / run-to-run differences)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd")); sys.path.insert(0, ROOT)
import bench
from approxgp import _ffi
ctx = _ffi.Context(0)
bad = 0
for dt in ("f64", "f32"):
    for M in (128, 640, 1024, 2048, 3200):
        p = bench.synth(3, 4096, M, 8, 0, 0, dt)
        desc, keep = _ffi.make_desc(p["np_dt"], 0, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=0, lik_sigma2=p["sigma2"])
        model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
        vals, errs = set(), 0
        for _ in range(12):
            try:
                vals.add(model.elbo(data, 0, 4096, 4096.0)[0])
            except Exception as e:
                errs += 1
        print(dt, M, "errors", errs, "distinct values", len(vals), sorted(vals)[:2], flush=True)
        bad += errs + (len(vals) > 1)
        model.free(); data.free()
print(os.environ.get("SVGP_MI355X_LIB", "default").split("/")[-1], "BAD" if bad else "OK")
