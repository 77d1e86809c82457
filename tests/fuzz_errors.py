"""Randomised sweep of the ERROR paths: one argument of a call is made invalid, or the numbers are made hostile (a non-positive-definite
Kuu, a zero on the diagonal of Lq, NaN / Inf in x, y or the parameters, a batch window outside the data), and the call must come back with
a status - never crash, never hang - after which the SAME context must evaluate a healthy problem to the oracle's value (nothing of the
failed call may be left in the workspaces, on the second stream or in the pinned staging).

    python tests/fuzz_errors.py [--seconds 240] [--seed 36]

A script for the GPU box, not a pytest file; the oracle is the checker.  Exit code 1 when a hostile call is accepted with a finite
value, or when the healthy evaluation after it is off.  (The first run found one: a NaN coordinate in x gave a FINITE value - the clamp
of the distance tile, min(v, c0), returned c0 for a NaN; the clamps now let a NaN through, profiles/round6/fuzz_grad.md.)"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "approximategps.jl_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import svgp_oracle as o  # noqa: E402
from approxgp import _ffi  # noqa: E402
from helpers import rel  # noqa: E402


def healthy(ctx, rng, dtype):
    N, M, d = int(rng.choice([200, 1500, 9000])), int(rng.choice([20, 130, 640])), int(rng.choice([1, 3, 8, 20]))
    x, y, sva, s2 = o.synth_problem(int(rng.integers(1, 1 << 30)), N, M, d, dtype=dtype)
    return dict(N=N, M=M, d=d, x=x, y=y, sva=sva, s2=s2, dtype=dtype)


def desc_of(p, **over):
    s = p["sva"]
    kw = dict(dtype=p["dtype"], kernel=s.kernel.family, variance=s.kernel.variance, inv_lengthscale=s.kernel.inv_lengthscale, z=s.z, m=s.m,
              Lq=s.Lq, jitter=s.jitter)
    kw.update({k: v for k, v in over.items() if k in kw})
    desc, keep = _ffi.make_desc(kw["dtype"], kw["kernel"], kw["variance"], kw["inv_lengthscale"], kw["z"], kw["m"], kw["Lq"], kw["jitter"],
                                lik_sigma2=over.get("lik_sigma2", p["s2"]), likelihood=over.get("likelihood", o.LIK_GAUSSIAN),
                                quadrature_n=over.get("quadrature_n", 0), mean_const=over.get("mean_const", 0.0))
    for k in ("d", "M"):
        if k in over:
            setattr(desc, k, over[k])
    if "raw_dtype" in over:
        desc.dtype = over["raw_dtype"]
    if "raw_kernel" in over:
        desc.kernel = over["raw_kernel"]
    if "raw_param" in over:
        desc.parametrization = over["raw_param"]
    if "null" in over:
        setattr(desc, over["null"], None)
    return desc, keep


HOSTILE = ["bad_dtype", "bad_kernel", "bad_lik", "bad_param", "d_zero", "d_65", "M_zero", "neg_variance", "nan_variance", "nan_lengthscale",
           "zero_sigma2", "null_z", "null_Lq", "quad_huge", "not_posdef", "zero_diag_Lq", "nan_in_z", "nan_in_m", "nan_in_x", "inf_in_y",
           "window_neg", "window_past_end", "window_empty", "null_out"]


def hostile_call(ctx, rng, p, kind):
    """-> (status text, value or None).  Statuses arrive as exceptions of the ctypes wrapper."""
    s = p["sva"]
    M, N = p["M"], p["N"]
    over, data_x, data_y, off, nb = {}, p["x"], p["y"], 0, N
    if kind == "bad_dtype": over["raw_dtype"] = 7
    elif kind == "bad_kernel": over["raw_kernel"] = int(rng.choice([-1, 3, 99]))
    elif kind == "bad_lik": over["likelihood"] = int(rng.choice([-1, 6, 1000]))
    elif kind == "bad_param": over["raw_param"] = 5
    elif kind == "d_zero": over["d"] = 0
    elif kind == "d_65": over["d"] = 65
    elif kind == "M_zero": over["M"] = 0
    elif kind == "neg_variance": over["variance"] = -1.0
    elif kind == "nan_variance": over["variance"] = float("nan")
    elif kind == "nan_lengthscale":
        il = np.array(s.kernel.inv_lengthscale, dtype=np.float64).copy()
        il[int(rng.integers(il.size))] = float("nan")
        over["inv_lengthscale"] = il
    elif kind == "zero_sigma2": over["lik_sigma2"] = 0.0
    elif kind == "null_z": over["null"] = "z"
    elif kind == "null_Lq": over["null"] = "Lq"
    elif kind == "quad_huge": over["quadrature_n"] = 100000
    elif kind == "not_posdef": over["jitter"] = -10.0
    elif kind == "zero_diag_Lq":
        Lq = np.array(s.Lq).copy()
        Lq[int(rng.integers(M)), :] = 0.0
        over["Lq"] = Lq
    elif kind == "nan_in_z":
        z = np.array(s.z).copy()
        z.flat[int(rng.integers(z.size))] = float("nan")
        over["z"] = z
    elif kind == "nan_in_m":
        m = np.array(s.m).copy()
        m[int(rng.integers(M))] = float("nan")
        over["m"] = m
    elif kind == "nan_in_x":
        data_x = np.array(p["x"]).copy()
        data_x.flat[int(rng.integers(data_x.size))] = float("nan")
    elif kind == "inf_in_y":
        data_y = np.array(p["y"]).copy()
        data_y[int(rng.integers(N))] = float("inf")
    elif kind == "window_neg": off = -1
    elif kind == "window_past_end": off, nb = N // 2, N
    elif kind == "window_empty": nb = 0
    model = data = None
    try:
        desc, keep = desc_of(p, **over)
        model = _ffi.DeviceModel(ctx, desc, keep)
        data = _ffi.DeviceData(ctx, data_x, data_y, p["dtype"])
        if kind == "null_out":
            terms = _ffi.Terms()
            rc = ctx.lib.svgp_elbo(ctx.h, model.h, data.h, 0, N, 0.0, None, C.byref(terms))
            return f"rc={rc}", None
        if rng.random() < 0.5:
            return "OK", model.elbo(data, off, nb, 0.0)[0]
        return "OK", model.elbo_grad(data, off, nb, 0.0)[0]
    except _ffi.SvgpError as e:
        return type(e).__name__ + ": " + str(e)[:90], None
    except ValueError as e:   # SVGP_INVALID_ARG (the wrapper raises Julia's ArgumentError as ValueError)
        return "ValueError: " + str(e)[:90], None
    finally:
        if model is not None:
            model.free()
        if data is not None:
            data.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=36)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = _ffi.Context(0)
    t0, n, bad, seen = time.time(), 0, [], {}
    while time.time() - t0 < args.seconds:
        dtype = np.float32 if rng.random() < 0.3 else np.float64
        p = healthy(ctx, rng, dtype)
        kind = str(rng.choice(HOSTILE))
        status, val = hostile_call(ctx, rng, p, kind)
        note = ""
        if status == "OK":
            # accepted: the numbers decide.  A finite value must be the oracle's; NaN / Inf inputs may give NaN / Inf
            if val is not None and np.isfinite(val):   # NaN / Inf inputs must give NaN / Inf, as the reference's arithmetic does
                note = "ACCEPTED with a finite value"
                bad.append((n, kind, status, val))
        seen.setdefault(kind, set()).add(status.split(":")[0] if status != "OK" else ("OK finite" if val is not None and np.isfinite(val) else "OK non-finite"))
        # the context must be fully usable: a healthy value and gradient against the oracle
        model = _ffi.DeviceModel(ctx, *desc_of(p))
        data = _ffi.DeviceData(ctx, p["x"], p["y"], dtype)
        v, _, g = model.elbo_grad(data, 0, p["N"], 0.0)
        v_ref, g_ref = o.elbo_grad(p["sva"], p["x"], p["y"], sigma2=p["s2"])
        ev = rel(v, v_ref)
        eL = float(np.abs(np.asarray(g["Lq"], dtype=np.float64) - g_ref["Lq"]).max() / np.abs(g_ref["Lq"]).max())
        model.free()
        data.free()
        ok = ev < (1e-8 if dtype == np.float64 else 2e-4) and eL < (1e-6 if dtype == np.float64 else 2e-2)
        if not ok:
            bad.append((n, kind, "healthy evaluation after it", ev, eL))
        print("CASE", n, kind, dtype.__name__, f"[N={p['N']} M={p['M']} d={p['d']}]", "->", status, note, "| after:", f"{ev:.1e} {eL:.1e}",
              "ok" if ok else "FAIL", flush=True)
        n += 1
    ctx.close()
    print(f"SUMMARY {n} hostile calls in {time.time() - t0:.0f} s, {len(bad)} findings")
    for k in HOSTILE:
        print("  ", k, "->", sorted(seen.get(k, [])))
    for b in bad:
        print("  BAD", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
