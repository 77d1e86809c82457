"""A few forward evaluations of a small batch on a large Kuu (for kernel traces of the factorisation): python chol_once.py f32 8192"""
import os, sys
R = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(R, "..", ".."), os.path.join(R, "..", "..", "approximategps.jl_amd")]
import numpy as np, bench
from approxgp import _ffi
dt, M = sys.argv[1], int(sys.argv[2])
ctx = _ffi.Context(0)
p = bench.synth(4, 4096, M, 8, 0, 0, dt)
desc, keep = _ffi.make_desc(p["np_dt"], 0, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=0, lik_sigma2=p["sigma2"])
model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
for _ in range(4):
    v = model.elbo(data, 0, 4096, 4096.0)[0]; t = ctx.timing()
print(f"{dt} M={M}: cholesky {t.ms_chol:.3f} ms prep {t.ms_prep:.3f} elbo {v!r}")
