"""Randomised parity sweep of svgp_elbo / svgp_elbo_grad against the oracle (round 6: the kernel-gradient reductions are new).

    python tests/fuzz_grad.py [--seconds 420] [--seed 6] [--large]

Draws (points, inducing points, input dimension, kernel family, likelihood, quadrature, parametrisation, dtype, window) at random -
ragged and degenerate shapes included - and compares value and every gradient block with oracle/svgp_oracle.py.  Tolerances are the
test suite's (tests/test_gpu_grad.py): fp64 1e-8 on the value, 1e-6 of the block's max-norm on gradients; fp32 1e-4 / 5e-3; the value of
the gradient entry point against the forward one (two algebraically equal forms of the variance) 1e-10 / 1e-4.
Prints one line per case and a summary; exit code 1 when a case is outside its tolerance.  A script for the GPU box, not a pytest file; it uses the oracle as the checker, so it lives
with the tests (as tests/soak_overlap.py does)."""
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
from helpers import device_model, rel  # noqa: E402

LIKS = [o.LIK_GAUSSIAN, o.LIK_GAUSSIAN, o.LIK_BERNOULLI_LOGISTIC, o.LIK_POISSON_EXP, o.LIK_EXPONENTIAL_EXP, o.LIK_GAMMA_EXP,
        o.LIK_BERNOULLI_NORMCDF]
FAMS = [o.KERNEL_SE, o.KERNEL_MATERN32, o.KERNEL_MATERN52]


def draw_large(rng):
    """Shapes of the multi-launch plans: several gradient chunks (N > 65 536 / its fp32 and M-dependent equivalents), ragged tails of
    half-width strips, six and more factorisation panels (strips beside the factorisation), windows that start inside a chunk."""
    M = int(rng.choice([640, 1000, 1024, 1100, 1536, 2048]))
    N = int(rng.choice([20000, 40000, 65536, 65537, 70001, 100000]))   # (the oracle holds a dozen M x N fp64 arrays: <= 1.6 GB each)
    off = int(rng.integers(0, N // 2))
    return dict(N=N, M=M, d=int(rng.choice([1, 3, 8, 9, 16, 20, 33])), family=int(rng.choice(FAMS)), lik=int(rng.choice(LIKS)),
                qn=int(rng.choice([0, 0, 7])), centered=bool(rng.random() < 0.2), dtype=np.float32 if rng.random() < 0.4 else np.float64,
                window=bool(rng.random() < 0.4), off=off, nb=int(rng.integers(1, N - off + 1)), seed=int(rng.integers(1, 1 << 30)))


def draw(rng):
    d = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 12, 16, 17, 23, 31, 32, 33, 47, 48, 63, 64]))
    M = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 300, 511, 512, 640]))
    N = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 500, 1000, 1023, 1025, 2049, 4097, 9001]))
    if M > 300 and N > 4097:
        N = 4097   # keeps the oracle's M x N temporaries and its M^3 adjoint within seconds
    return dict(N=N, M=M, d=d, family=int(rng.choice(FAMS)), lik=int(rng.choice(LIKS)), qn=int(rng.choice([0, 0, 0, 5, 12])),
                centered=bool(rng.random() < 0.25), dtype=np.float32 if rng.random() < 0.4 else np.float64,
                window=bool(rng.random() < 0.3), seed=int(rng.integers(1, 1 << 30)))


def block_err(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def run_case(ctx, c):
    N, M, d, dtype, lik = c["N"], c["M"], c["d"], c["dtype"], c["lik"]
    x, y, sva, s2 = o.synth_problem(c["seed"], N, M, d, family=c["family"], lik=lik, dtype=dtype)
    sva.mean_const = 0.1
    if c["centered"]:
        jit = 1e-4 if dtype == np.float64 else 1e-2
        tame = 0.1 if lik == o.LIK_POISSON_EXP else 1.0
        sva = o.SVA(sva.kernel, sva.z, tame * (sva.m + 0.3), 0.7 * tame * sva.Lq, jitter=jit, mean_const=0.15, centered=True)
    off, nb = 0, N
    if c["window"] and N > 4:
        off = c.get("off", N // 3)
        nb = c.get("nb", max(1, N // 2))
    xs, ys = (x[:, off:off + nb] if x.ndim == 2 else x[off:off + nb]), y[off:off + nb]
    val_ref, g_ref = o.elbo_grad(sva, xs, ys, lik=lik, sigma2=s2, num_data=2.5 * N, quadrature_n=c["qn"])
    model = device_model(ctx, sva, dtype=dtype, lik=lik, sigma2=s2, quadrature_n=c["qn"])
    data = _ffi.DeviceData(ctx, x, y, dtype)
    try:
        val, _, g = model.elbo_grad(data, off, nb, 2.5 * N)
        fwd = model.elbo(data, off, nb, 2.5 * N)[0]
    finally:
        model.free()
        data.free()
    errs = {"value": rel(val, val_ref), "fwd_vs_grad_value": rel(val, fwd)}
    gz = g["z"].reshape(g_ref["z"].shape, order="F") if d > 1 else g["z"]
    errs["z"] = block_err(gz, g_ref["z"] if d > 1 else g_ref["z"][0])
    for k in ("m", "Lq", "inv_lengthscale"):
        errs[k] = block_err(g[k], g_ref[k])
    for k in ("variance", "mean_const"):
        errs[k] = block_err([g[k]], [g_ref[k]])
    if lik in (o.LIK_GAUSSIAN, o.LIK_GAMMA_EXP) and not (c["centered"] and lik == o.LIK_GAMMA_EXP):
        errs["lik_sigma2"] = block_err([g["lik_sigma2"]], [g_ref["lik_sigma2"]])
    return errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=420.0)
    ap.add_argument("--seed", type=int, default=6)
    ap.add_argument("--max-cases", type=int, default=100000)
    ap.add_argument("--large", action="store_true", help="M 640 ... 2500, N 20 000 ... 150 000 (seconds per case: the oracle)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = _ffi.Context(0)
    t0, n, bad, worst = time.time(), 0, [], {}
    while time.time() - t0 < args.seconds and n < args.max_cases:
        c = draw_large(rng) if args.large else draw(rng)
        f64 = c["dtype"] == np.float64
        vtol, gtol = (1e-8, 1e-6) if f64 else (1e-4, 5e-3)
        tag = {k: (v.__name__ if k == "dtype" else v) for k, v in c.items()}
        try:
            errs = run_case(ctx, c)
        except Exception as e:   # a status from the library is a finding too
            bad.append((tag, repr(e)))
            print("CASE", n, tag, "EXCEPTION", repr(e), flush=True)
            n += 1
            continue
        fails = {k: v for k, v in errs.items()
                 if v > (vtol if k == "value" else (1e-10 if f64 else 1e-4) if k == "fwd_vs_grad_value" else gtol) or not np.isfinite(v)}
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
