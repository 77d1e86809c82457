"""cholesky(Kuu) of a large Kuu under settings of ONE experiments-library knob (read once per process: each setting runs in a child
process, twice, interleaved): ms of the factorisation (HIP events of the library) and the ELBO.
usage: SVGP_MI355X_LIB=.../libsvgp_experiments.so python tools/round5/chol_env_ab.py SVGP_SOME_KNOB 0 1 [2 ...] [-- f32:8192 f32:4096 f64:4096]"""
import os, subprocess, sys
R = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1:]
cases = ["f32:4096", "f32:8192", "f64:4096"]
if "--" in args:
    cases = args[args.index("--") + 1:]; args = args[:args.index("--")]
knob, values = args[0], args[1:]
CODE = r'''
import os, sys
sys.path[:0] = [%r, %r]
import numpy as np, bench
from approxgp import _ffi
ctx = _ffi.Context(0)
for case in %r:
    dt, M = case.split(":"); M = int(M)
    p = bench.synth(4, 4096, M, 8, 0, 0, dt)
    desc, keep = _ffi.make_desc(p["np_dt"], 0, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=0, lik_sigma2=p["sigma2"])
    model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
    ts = []
    for _ in range(8):
        try:
            v = model.elbo(data, 0, 4096, 4096.0)[0]
        except Exception as e:
            v = repr(e)[:60]
        t = ctx.timing(); ts.append((t.ms_chol, t.ms_prep))
    ts = np.array(ts[2:])
    print(f"{dt} M={M}: cholesky {np.median(ts[:,0]):.3f} ms (min {ts[:,0].min():.3f}), prep {np.median(ts[:,1]):.3f} ms, elbo {v!r}", flush=True)
    model.free(); data.free()
''' % (os.path.join(R, "..", ".."), os.path.join(R, "..", "..", "approximategps.jl_amd"), cases)
for v in values + values:
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **{knob: v}), capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if "cholesky" in l]
    print(f"---- {knob}={v}"); print("\n".join(lines)); sys.stdout.flush()
    if r.returncode: print(r.stderr[-1500:])
