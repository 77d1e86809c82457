"""cholesky(Kuu) of a large Kuu with / without the two-stream look-ahead (EXPERIMENTS library: SVGP_CHOL_LOOKAHEAD is read once per
process, so each setting runs in a child process): ms of the factorisation (HIP events of the library), bitwise equality of the
ELBO, for f32 / f64 and M = 2304 ... 8192.  usage: SVGP_MI355X_LIB=.../libsvgp_experiments.so python tools/round5/chol_lookahead_ab.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.abspath(__file__))
CODE = r'''
import os, sys
sys.path[:0] = [%r, %r]
import numpy as np, bench
from approxgp import _ffi
ctx = _ffi.Context(0)
for dt in ("f32", "f64"):
    for M in (2304, 4096, 8192):
        if dt == "f64" and M > 4096: continue
        p = bench.synth(4, 4096, M, 8, 0, 0, dt)
        desc, keep = _ffi.make_desc(p["np_dt"], 0, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=0, lik_sigma2=p["sigma2"])
        model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
        ts = []
        for _ in range(8):
            v = model.elbo(data, 0, 4096, 4096.0)[0]; t = ctx.timing(); ts.append((t.ms_chol, t.ms_prep))
        ts = np.array(ts[2:])
        print(f"{dt} M={M}: cholesky {np.median(ts[:,0]):.3f} ms (min {ts[:,0].min():.3f}), prep {np.median(ts[:,1]):.3f} ms, elbo {v!r}", flush=True)
        model.free(); data.free()
''' % (os.path.join(R, "..", ".."), os.path.join(R, "..", "..", "approximategps.jl_amd"))
outs = {}
for la in ("0", "1", "0", "1"):
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, SVGP_CHOL_LOOKAHEAD=la), capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if "cholesky" in l]
    print(f"---- SVGP_CHOL_LOOKAHEAD={la}"); print("\n".join(lines)); sys.stdout.flush()
    if r.returncode: print(r.stderr[-2000:])
    outs.setdefault(la, []).append([l.split("elbo ")[1] for l in lines])
print("bitwise equal ELBOs with the look-ahead on / off:", outs["0"][0] == outs["1"][0] == outs["0"][1] == outs["1"][1])
