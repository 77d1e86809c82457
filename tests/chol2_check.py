"""Two-level blocked Cholesky (large Kuu, fp32) against the one-level loop: same process (separate model builds per knob via
subprocess is not needed: the knob is read once per process, so this script is run once per setting), prints ms_chol and the ELBO /
posterior checks.  usage: SVGP_CHOL_TWO_LEVEL=0|1 python tests/chol2_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd")); sys.path.insert(0, ROOT)
import bench
from approxgp import _ffi
ctx = _ffi.Context(0)
tag = "two_level=" + os.environ.get("SVGP_CHOL_TWO_LEVEL", "1")
for M in (2176, 2304, 3200, 4224, 8192):
    p = bench.synth(4, 4096, M, 8, 0, 0, "f32")
    desc, keep = _ffi.make_desc(p["np_dt"], 0, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=0, lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    vals, ts, errs = [], [], 0
    for _ in range(8):
        try:
            vals.append(model.elbo(data, 0, 4096, 4096.0)[0]); ts.append(ctx.timing().ms_chol)
        except Exception as e:
            errs += 1; last = repr(e)
    Lk, alpha, B = model.posterior() if not errs else (None, None, None)
    kl, ld = model.prior_kl() if not errs else (float("nan"), float("nan"))
    chk = ""
    if Lk is not None and M <= 4224:   # reconstruct Kuu from the factor: max |L L' - K| / max |K| against the host kernel matrix
        sys.path.insert(0, os.path.join(ROOT, "oracle")); import svgp_oracle as o
        K = o.kernelmatrix(o.Kernel(0, p["variance"], p["inv_l"]), p["z"], p["z"]) + p["jitter"] * np.eye(M)
        L64 = np.tril(Lk.astype(np.float64))
        chk = f" |LL'-K|/|K| {np.abs(L64 @ L64.T - K).max() / np.abs(K).max():.2e}"
    print(f"{tag} M={M}: ms_chol {np.median(ts[1:]) if ts else float('nan'):.3f} errors {errs} distinct {len(set(vals))} elbo {vals[0] if vals else None} logdet {ld:.6f}{chk}", flush=True)
    model.free(); data.free()
