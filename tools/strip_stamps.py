#!/usr/bin/env python
"""Reads the s_memtime stamps of one steady-state strip (library built by `tools/build_ablate.sh stripstamps x`)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd")); sys.path.insert(0, ROOT)
import bench
from approxgp import _ffi
cfg = sys.argv[1] if len(sys.argv) > 1 else "H"
n, M, d, family, lik, dtype, cid = bench.CONFIGS[cfg]
n = min(n, 400_000)
p = bench.synth(cid, n, M, d, family, lik, dtype)
ctx = _ffi.Context(0)
desc, keep = _ffi.make_desc(p["np_dt"], family, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=lik, lik_sigma2=p["sigma2"])
model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
model.elbo(data, 0, n, float(n))     # the stamps are sums over every strip (but the first) of one workgroup, one launch
L = ctypes.CDLL(os.environ.get("SVGP_MI355X_LIB", _ffi.LIB_PATH))
buf = (ctypes.c_ulonglong * 128)()
L.svgp_debug_strip_stamps(buf)
raw = np.array([int(v) for v in buf], dtype=object)
cnt = int(raw[127])
s = np.array([float((int(v) - int(raw[0])) % (1 << 64)) / max(cnt, 1) for v in raw[:127]] + [0.0])   # mean offsets from the strip start
print(f"averaged over {cnt} strips")
nP = (M + 127) // 128
print(f"{cfg}: strip total {s[101]-s[0]:.0f} ticks (s_memtime counts shader clocks, ~2.3-2.4 GHz: tools/ubench/clock_rate.hip); x staging {s[1]-s[0]:.0f}; final moments {s[101]-s[100]:.0f}")
l1 = [s[3 + 3 * I] - s[2 + 3 * I] for I in range(nP)]; e1 = [s[4 + 3 * I] - s[3 + 3 * I] for I in range(nP)]
l2 = [s[61 + 2 * J] - s[60 + 2 * J] for J in range(nP)]
e2 = [(s[60 + 2 * (J + 1)] if J + 1 < nP else s[100]) - s[61 + 2 * J] for J in range(nP)]
print(" phase 1 loops    ", [int(v) for v in l1], "sum", int(sum(l1)))
print(" phase 1 epilogues", [int(v) for v in e1], "sum", int(sum(e1)))
print(" phase 2 loops    ", [int(v) for v in l2], "sum", int(sum(l2)))
print(" phase 2 epilogues", [int(v) for v in e2], "sum", int(sum(e2)))
steps1 = [(I + 1) * 8 for I in range(nP)]
print(" ticks per k-step, phase 1:", [round(a / b, 1) for a, b in zip(l1, steps1)], " phase 2:", [round(a / b, 1) for a, b in zip(l2, steps1[::-1])])
