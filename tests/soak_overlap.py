"""Soak of the strips-beside-the-factorisation path (a script like tests/soak.py, not a pytest file): 300 evaluations alternating
forward / value-and-gradient, model updates in between, every result compared bit for bit with the same call behind the prep
(SVGP_OVERLAP=0).  A race between the two streams (scratch reuse, events of consecutive evaluations) would show as a difference.
usage (GPU box): python tests/soak_overlap.py [reps]"""
import os, sys
R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "approximategps.jl_amd")); sys.path.insert(0, os.path.join(R, "..", "oracle")); sys.path.insert(0, R)
import numpy as np
import svgp_oracle as o
from approxgp import _ffi
from helpers import desc_from_oracle


def make_ctx(**env):   # the overlap settings are read once, at context creation (csrc/knobs.hpp); an ambient value of the same variable is
    old = {k: os.environ.get(k) for k in env}   # restored afterwards, as helpers.context_with_env does
    os.environ.update(env)
    try:
        return _ffi.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctxs = {"1": make_ctx(SVGP_OVERLAP="1", SVGP_SEG_SPLIT="0"), "0": make_ctx(SVGP_OVERLAP="0")}
rng = np.random.default_rng(0)
bad = 0
for dtype, N, M, d in ((np.float64, 9000, 1024, 4), (np.float32, 20000, 700, 8)):
    x, y, sva, s2 = o.synth_problem(9100, N, M, d, dtype=dtype)
    desc, keep = desc_from_oracle(sva, dtype=dtype, sigma2=s2)
    models = {k: _ffi.DeviceModel(c, desc, keep) for k, c in ctxs.items()}
    datas = {k: _ffi.DeviceData(c, x, y, dtype) for k, c in ctxs.items()}
    for rep in range(reps // 2):
        # new parameter values every few steps (a training loop): m and the kernel variance move
        if rep % 3 == 0:
            sva.m = (sva.m + 0.01 * rng.standard_normal(M)).astype(np.float64)
            sva.kernel = o.Kernel(sva.kernel.family, sva.kernel.variance * (1.0 + 0.01 * rng.standard_normal()), sva.kernel.inv_lengthscale)
            desc, keep = desc_from_oracle(sva, dtype=dtype, sigma2=s2)
            for mdl in models.values():
                mdl.update(desc, keep)
        off = int(rng.integers(0, 64)); n = N - off - int(rng.integers(0, 64))
        res = {}
        for k in ("1", "0"):
            v = models[k].elbo(datas[k], off, n, float(N))[0]
            vg, _, g = models[k].elbo_grad(datas[k], off, n, float(N))
            res[k] = (v, vg, g)
        same = res["1"][0] == res["0"][0] and res["1"][1] == res["0"][1] and all(
            np.array_equal(np.asarray(res["1"][2][q]), np.asarray(res["0"][2][q])) for q in ("z", "m", "Lq", "inv_lengthscale"))
        bad += 0 if same else 1
    print(f"{np.dtype(dtype).name} N={N} M={M}: {reps // 2} rounds of (forward, gradient) x (beside, behind): mismatches so far {bad}", flush=True)
    for k in models:
        models[k].free(); datas[k].free()
# Small batches close with SEVERAL workgroups per strip (split closing launch: the last workgroup of a strip to arrive adds the parts'
# column sums in part order).  Which workgroup arrives last varies from call to call; the result must not: every call on the same
# inputs returns identical bits (forward, and every gradient block), and agrees with the unsplit launch to rounding.  Windows of
# varying length and offset, reused host arrays (`out=`), the pinned read-back in pieces.
split_ctx = make_ctx(SVGP_OVERLAP="1", SVGP_SEG_SPLIT="1")
for dtype, N, M, d in ((np.float64, 1500, 1024, 4), (np.float32, 3000, 1536, 8), (np.float64, 700, 2048, 2)):
    x, y, sva, s2 = o.synth_problem(9200 + M, N, M, d, dtype=dtype)
    desc, keep = desc_from_oracle(sva, dtype=dtype, sigma2=s2)
    model = _ffi.DeviceModel(split_ctx, desc, keep)
    data = _ffi.DeviceData(split_ctx, x, y, dtype)
    model0 = _ffi.DeviceModel(ctxs["1"], desc, keep)      # the same path with the unsplit closing launch
    data0 = _ffi.DeviceData(ctxs["1"], x, y, dtype)
    gout = None
    for rep in range(reps // 6):
        off = int(rng.integers(0, 64)); n = N - off - int(rng.integers(0, 64))
        ref = None
        for again in range(3):
            v = model.elbo(data, off, n, float(N))[0]
            vg, _, gout = model.elbo_grad(data, off, n, float(N), **({"out": gout} if gout is not None else {}))
            cur = (v, vg, {q: np.array(gout[q], copy=True) for q in ("z", "m", "Lq", "inv_lengthscale")})
            if ref is None:
                ref = cur
            elif not (cur[0] == ref[0] and cur[1] == ref[1] and all(np.array_equal(cur[2][q], ref[2][q]) for q in cur[2])):
                bad += 1
        v0 = model0.elbo(data0, off, n, float(N))[0]
        vg0 = model0.elbo_grad(data0, off, n, float(N))[0]
        tol = 1e-12 if dtype == np.float64 else 1e-5
        if abs(ref[0] - v0) > tol * abs(v0) or abs(ref[1] - vg0) > tol * abs(vg0):
            bad += 1
    print(f"{np.dtype(dtype).name} N={N} M={M}: {reps // 6} windows x 3 repeated calls with the split closing launch: mismatches so far {bad}", flush=True)
    model.free(); data.free(); model0.free(); data0.free()
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
