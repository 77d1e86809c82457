"""Randomised parity sweep of the forward entry points against the oracle: svgp_marginals, svgp_predict (mean / var / cov),
svgp_predict_cross_cov, svgp_kuf, svgp_posterior, svgp_prior_kl - shapes, kernel family, parametrisation, dtype and batch window drawn
at random (ragged and degenerate ones included), as tests/fuzz_grad.py does for value and gradient.

    python tests/fuzz_forward.py [--seconds 300] [--seed 16] [--large]

Errors are max |device - oracle| over an array divided by max(|oracle|, floor); tolerances: fp64 1e-7 (alpha and the Centered B carry
cond(Lk)); fp32 3e-3, and 2e-2 for alpha = Lk^-T m and the means that are k' alpha (eps * cond(Lk): with 500 inducing points in one or
two dimensions and the fp32 jitter of the recipe that is 3e-3 ... 8e-3 - tests/test_gpu_round5.py pins that regime).  Exit code 1 when a
case is outside its tolerance."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "approximategps.jl_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import svgp_oracle as o  # noqa: E402
from approxgp import _ffi  # noqa: E402
from helpers import device_model  # noqa: E402

FAMS = [o.KERNEL_SE, o.KERNEL_MATERN32, o.KERNEL_MATERN52]


def draw(rng):
    return dict(d=int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 31, 32, 33, 48, 64])),
                M=int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 300, 511, 512, 640])),
                N=int(rng.choice([1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 500, 1000, 1023, 1025, 2049])),
                family=int(rng.choice(FAMS)), centered=bool(rng.random() < 0.3),
                dtype=np.float32 if rng.random() < 0.4 else np.float64, seed=int(rng.integers(1, 1 << 30)))


def draw_large(rng):
    c = draw(rng)
    c.update(M=int(rng.choice([1000, 1024, 1536, 2048, 3000])), N=int(rng.choice([5000, 20000, 65537])), d=int(rng.choice([1, 3, 8, 16, 33, 64])))
    return c


def err(a, b, floor=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor)) if b.size else 0.0


def run_case(ctx, c):
    N, M, d, dtype = c["N"], c["M"], c["d"], c["dtype"]
    x, y, sva, s2 = o.synth_problem(c["seed"], N, M, d, family=c["family"], dtype=dtype)
    if c["centered"]:
        sva = o.SVA(sva.kernel, sva.z, sva.m + 0.3, 0.7 * sva.Lq, jitter=1e-4 if dtype == np.float64 else 1e-2, mean_const=0.15,
                    centered=True)
    else:
        sva.mean_const = -0.2
    post = o.posterior(sva)
    model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    out = {}
    try:
        off = N // 4
        nb = max(1, N - off - N // 5)
        xw = x[:, off:off + nb] if x.ndim == 2 else x[off:off + nb]
        mu, var = model.marginals(data, off, nb)
        mu_ref, v_ref = o.mean_and_var(post, xw)
        out["marg_mean"], out["marg_var"] = err(mu, mu_ref), err(var, v_ref + 1e-18)
        npred = min(nb, 97)
        xs = xw[:, :npred] if xw.ndim == 2 else xw[:npred]
        xt = xw[:, npred // 2:] if xw.ndim == 2 else xw[npred // 2:]
        xt = xt[:, :61] if xt.ndim == 2 else xt[:61]
        mean, v, cov = model.predict(xs, True, True, True)
        mr, vr = o.mean_and_var(post, xs)
        out["pred_mean"], out["pred_var"], out["pred_cov"] = err(mean, mr), err(v, vr), err(cov, o.cov(post, xs))
        out["cross_cov"] = err(model.cross_cov(xs, xt), o.cov(post, xs, xt))
        out["kuf"] = err(model.kuf(data, off, nb), o.kernelmatrix(sva.kernel, sva.z, xw))
        Lk, alpha, B = model.posterior()
        out["Lk"], out["alpha"], out["B"] = err(Lk, post.Lk), err(alpha, post.alpha), err(B, post.B)
        kl, _ = model.prior_kl()
        out["kl"] = abs(kl - o.prior_kl(sva)) / max(abs(o.prior_kl(sva)), 1e-6)
    finally:
        model.free()
        data.free()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=16)
    ap.add_argument("--large", action="store_true", help="M 1000 ... 3000, N 5000 ... 65 537")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = _ffi.Context(0)
    t0, n, bad, worst = time.time(), 0, [], {}
    while time.time() - t0 < args.seconds:
        c = draw_large(rng) if args.large else draw(rng)
        f64 = c["dtype"] == np.float64
        tol = 1e-7 if f64 else 3e-3
        tag = {k: (v.__name__ if k == "dtype" else v) for k, v in c.items()}
        try:
            errs = run_case(ctx, c)
        except Exception as e:
            bad.append((tag, repr(e)))
            print("CASE", n, tag, "EXCEPTION", repr(e), flush=True)
            n += 1
            continue
        fails = {k: v for k, v in errs.items()
                 if v > (2e-2 if (not f64 and k in ("alpha", "marg_mean", "pred_mean")) else tol) or not np.isfinite(v)}
        for k, v in errs.items():
            key = (k, "f64" if f64 else "f32")
            worst[key] = max(worst.get(key, 0.0), v)
        print("CASE", n, tag, "FAIL" if fails else "ok", {k: f"{v:.1e}" for k, v in (fails or errs).items()}, flush=True)
        if fails:
            bad.append((tag, fails))
        n += 1
    ctx.close()
    print(f"SUMMARY {n} cases in {time.time() - t0:.0f} s, {len(bad)} outside tolerance")
    for k in sorted(worst):
        print("  worst", k, f"{worst[k]:.2e}")
    for b in bad:
        print("  BAD", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
