"""Same-process A/B of the value-and-gradient chunk pipeline (round 5): one context per setting of SVGP_GRAD_PIPELINE /
SVGP_GRAD_PIPE_STREAMS / SVGP_GRAD_PIPE_PRIO (read at context creation), the same model and data uploaded to each, calls interleaved
over the settings, median and minimum wall time per call; every setting's value and gradient blocks must be bitwise the serial ones.
usage: pipe_ab.py CONFIG[,CONFIG..] [rounds] [settings "lanes:streams:prio,..."]"""
import os, sys, time
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "..")); sys.path.insert(0, os.path.join(R, "..", "..", "approximategps.jl_amd"))
import numpy as np, bench
from approxgp import _ffi

cfgs = (sys.argv[1] if len(sys.argv) > 1 else "H").split(",")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
settings = (sys.argv[3] if len(sys.argv) > 3 else "1:1:0,2:1:0,3:1:0,3:2:0,4:2:0").split(",")

def make_ctx(setting):
    lanes, streams, prio, mode = (setting.split(":") + ["1"])[:4]
    os.environ["SVGP_GRAD_PIPELINE"] = lanes; os.environ["SVGP_GRAD_PIPE_STREAMS"] = streams; os.environ["SVGP_GRAD_PIPE_PRIO"] = prio
    os.environ["SVGP_GRAD_PIPE_MODE"] = mode
    c = _ffi.Context(0)
    for k in ("SVGP_GRAD_PIPELINE", "SVGP_GRAD_PIPE_STREAMS", "SVGP_GRAD_PIPE_PRIO", "SVGP_GRAD_PIPE_MODE"): os.environ.pop(k)
    return c

for cfg in cfgs:
    n, M, d, family, lik, dtype, cid = bench.CONFIGS[cfg]
    p = bench.synth(cid, n, M, d, family, lik, dtype)
    runs = []
    for st in settings:
        ctx = make_ctx(st)
        desc, keep = _ffi.make_desc(p["np_dt"], family, p["variance"], p["inv_l"], p["z"], p["m"], p["Lq"], p["jitter"], likelihood=lik, lik_sigma2=p["sigma2"], neg_var_policy=_ffi.NEGVAR_CLAMP)
        model = _ffi.DeviceModel(ctx, desc, keep); data = _ffi.DeviceData(ctx, p["x"], p["y"], p["np_dt"])
        v, _, g = model.elbo_grad(data, 0, n, float(n))   # warm: allocations
        runs.append(dict(st=st, ctx=ctx, model=model, data=data, v=v, g={k: np.array(g[k], copy=True) for k in ("z", "m", "Lq", "inv_lengthscale")}, sc=(g["variance"], g["lik_sigma2"], g["mean_const"]), ts=[]))
    ref = runs[0]
    for r in runs[1:]:
        same = r["v"] == ref["v"] and r["sc"] == ref["sc"] and all(np.array_equal(r["g"][k], ref["g"][k]) for k in ref["g"])
        print(f"{cfg} {r['st']}: bitwise equal to {ref['st']}: {same}", flush=True)
    for _ in range(rounds):
        for r in runs:
            t0 = time.perf_counter(); r["model"].elbo_grad(data=r["data"], off=0, length=n, num_data=float(n)) if False else r["model"].elbo_grad(r["data"], 0, n, float(n)); r["ts"].append(time.perf_counter() - t0)
    for r in runs:
        ts = np.array(r["ts"]) * 1e3
        print(f"{cfg} lanes:streams:prio[:mode] {r['st']}: median {np.median(ts):.3f} ms  min {ts.min():.3f} ms  (x{np.median(ts) / np.median(np.array(ref['ts']) * 1e3):.3f} of {ref['st']})", flush=True)
    for r in runs:
        r["model"].free(); r["data"].free(); r["ctx"].close()
