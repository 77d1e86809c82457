#!/usr/bin/env python
"""PCIe-inclusive timing of the one-shot entry point svgp_elbo_host at a bench config (host x, y uploaded on every call,
model created on every call) against the resident-data path.  Never part of bench.py's `value`."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd")); sys.path.insert(0, ROOT)
import torch  # noqa: F401  (HIP runtime load order)
import bench
from approxgp import _ffi
cfg = sys.argv[1] if len(sys.argv) > 1 else "H"
n, M, d, family, lik, dtype, cid = bench.CONFIGS[cfg]
p = bench.synth(cid, n, M, d, family, lik, dtype)
ctx = _ffi.Context(0)
desc, keep = _ffi.make_desc(p["np_dt"], family, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=lik, lik_sigma2=p["sigma2"])
xb = np.asfortranarray(p["x"]); yb = np.ascontiguousarray(p["y"])
out, terms = C.c_double(), _ffi.Terms()
def one():
    rc = ctx.lib.svgp_elbo_host(ctx.h, C.byref(desc), _ffi.COLVECS, n, xb.ctypes.data_as(C.c_void_p), yb.ctypes.data_as(C.c_void_p),
                                float(n), C.byref(out), C.byref(terms))
    assert rc == 0, rc
one()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); one(); ts.append(time.perf_counter() - t0)
model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
model.elbo(data, 0, n, float(n)); tr = []
for _ in range(5):
    t0 = time.perf_counter(); v = model.elbo(data, 0, n, float(n))[0]; tr.append(time.perf_counter() - t0)
print(f"{cfg}: svgp_elbo_host (host x, y: {xb.nbytes + yb.nbytes >> 20} MiB pageable, ColVecs transposed on the device) {min(ts)*1e3:.1f} ms/eval = {1/min(ts):.1f} evals/s;"
      f" resident data {min(tr)*1e3:.1f} ms/eval; same value: {out.value == v}")
