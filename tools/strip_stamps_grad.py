#!/usr/bin/env python
"""s_memtime stamps of the value-and-gradient strips (library built by `tools/build_ablate.sh stripstamps x`): where a strip's time goes -
phase-1 loops / epilogues (incl. the point-major A), phase-3 loops / K-dot / point-major R A.  usage: SVGP_MI355X_LIB=.../libsvgp_stripstamps.so
python tools/strip_stamps_grad.py [H|H32]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd")); sys.path.insert(0, ROOT)
import bench
from approxgp import _ffi
cfg = sys.argv[1] if len(sys.argv) > 1 else "H"
n, M, d, family, lik, dtype, cid = bench.CONFIGS[cfg]
n = 262_144
p = bench.synth(cid, n, M, d, family, lik, dtype)
ctx = _ffi.Context(0)
desc, keep = _ffi.make_desc(p["np_dt"], family, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=lik, lik_sigma2=p["sigma2"])
model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
model.elbo_grad(data, 0, n, float(n))     # the stamps are sums over every strip (but the first) of one workgroup, last chunk's launch
L = ctypes.CDLL(os.environ.get("SVGP_MI355X_LIB", _ffi.LIB_PATH))
buf = (ctypes.c_ulonglong * 128)()
L.svgp_debug_strip_stamps(buf)
raw = [int(v) for v in buf]
cnt = max(raw[127], 1)
s = [float((v - raw[0]) % (1 << 64)) / cnt for v in raw]
nP = (M + 127) // 128
print(f"{cfg} value-and-gradient strip, averaged over {cnt} strips of one workgroup: total {s[101]-s[0]:.0f} ticks; x staging + pre-generation {s[1]-s[0]:.0f}")
l1 = [s[3 + 3 * I] - s[2 + 3 * I] for I in range(nP)]; e1 = [s[4 + 3 * I] - s[3 + 3 * I] for I in range(nP)]
l3 = [s[27 + 4 * I] - s[26 + 4 * I] for I in range(nP)]; kd = [s[28 + 4 * I] - s[27 + 4 * I] for I in range(nP)]; st = [s[29 + 4 * I] - s[28 + 4 * I] for I in range(nP)]
print(" phase 1 loops             ", [int(v) for v in l1], "sum", int(sum(l1)))
print(" phase 1 epilogues (A, At) ", [int(v) for v in e1], "sum", int(sum(e1)))
print(" phase 3 loops             ", [int(v) for v in l3], "sum", int(sum(l3)))
print(" phase 3 K-dot             ", [int(v) for v in kd], "sum", int(sum(kd)))
print(" phase 3 point-major store ", [int(v) for v in st], "sum", int(sum(st)))
print(" moments + tail            ", int(s[101] - s[100]))
print(" ticks per k-step: phase 1", [round(a / ((I + 1) * 8), 1) for I, a in enumerate(l1)], " phase 3", [round(a / (nP * 8), 1) for a in l3])
# per-workgroup timeline of the last (chunk) launch
wt = (ctypes.c_ulonglong * (16 * 12))()
if hasattr(L, "svgp_debug_wg_times"):
    L.svgp_debug_wg_times(wt)
    t = np.array([int(v) for v in wt], dtype=np.int64).reshape(16, 12)
    t0 = min(int(r[0]) for r in t if r[0] > 0)
    print(" workgroup timelines (ticks from the first stamped start; per strip durations; XCC):")
    for w in range(16):
        r = t[w]
        if r[0] == 0:
            continue
        ns = int(r[11] >> 8); starts = [int(v) for v in r[:min(ns, 10)]] + [int(r[10])]
        durs = [starts[i + 1] - starts[i] for i in range(len(starts) - 1)]
        print(f"  wg {37 * w:4d} xcc {int(r[11] & 0xf)} start {int(r[0]) - t0:8d} strips {ns} durations {durs} end {int(r[10]) - t0}")
