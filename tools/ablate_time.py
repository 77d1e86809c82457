"""Times the strip kernel of one (possibly ablated, timing-only) library build: prints ms_strip for config H/H32."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "approximategps.jl_amd"))
import numpy as np
import bench
from approxgp import _ffi

cfg = sys.argv[1] if len(sys.argv) > 1 else "H"
n, M, d, family, lik, dtype, cid = bench.CONFIGS[cfg]
p = bench.synth(cid, n, M, d, family, lik, dtype)
ctx = _ffi.Context(0)
desc, keep = _ffi.make_desc(p["np_dt"], family, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"],
                            likelihood=lik, lik_sigma2=p["sigma2"], neg_var_policy=_ffi.NEGVAR_CLAMP)
model = _ffi.DeviceModel(ctx, desc, keep)
data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
ts = []
for i in range(4):
    try:
        model.elbo_partial(data, 0, n)
    except Exception as e:  # ablated builds produce garbage
        pass
    ts.append(ctx.timing().ms_strip)
print(f"{os.environ.get('SVGP_MI355X_LIB','default').split('/')[-1]:28s} NT={os.environ.get('SVGP_STRIP_NT','64'):4s} {cfg}: strip ms {min(ts[1:]):.2f}  prep {ctx.timing().ms_prep:.2f}")
