#!/usr/bin/env python
"""A likelihood the C-ABI does not enumerate, trained through the device path: count data with a SOFTPLUS link,
y_i ~ Poisson(log(1 + exp f_i)) (in the reference: `PoissonLikelihood(softplus)`, any GPLikelihoods likelihood / link works the
same way).  Only SVA:355 - expected_loglikelihood(quadrature, lik, q_f, y) - depends on the likelihood and it is O(n) scalar
work, so the caller evaluates it (here: 20-point Gauss-Hermite in numpy; in Julia: the reference's own GPLikelihoods method,
under Zygote through `rrule_via_ad`) on marginals the device computed, and the device runs the whole O(M^2 n) backward pass:

    mu, var   = model.marginals(data)                                   # svgp_marginals      (SVA:354 on the MI355X)
    E, gmu, gv = likelihood.expectation(mu, var, y, 20, True)           # the caller          (SVA:355 on the host)
    elbo, _, g = model.elbo_grad(data, 0, N, N, ext=(E, gmu, gv))       # svgp_elbo_grad_ext  (the backward pass on the MI355X)

N = 20 000 points in 2-D, M = 64 inducing points, Adam on (log variance, log inverse lengthscales, z, m, A).

    python examples/c_caller_likelihood.py      # needs an MI355X
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd"))
import approxgp as ag  # noqa: E402
from approxgp import _ffi  # noqa: E402


class PoissonSoftplus(ag.CallerLikelihood):
    """log p(y | f) = y log(lambda) - lambda - log y!,  lambda = softplus(f); Gauss-Hermite expectation and its (mu, v) gradients."""

    def expectation(self, mu, var, y, n_points, want_grad):
        xs, ws = np.polynomial.hermite.hermgauss(int(n_points))
        ws = ws / np.sqrt(np.pi)
        sd = np.sqrt(var)
        f = mu[None, :] + np.sqrt(2.0) * sd[None, :] * xs[:, None]
        lam = np.logaddexp(0.0, f)
        e = float((ws[:, None] * (y[None, :] * np.log(lam) - lam)).sum())     # (the constant -log y! does not move anything)
        if not want_grad:
            return e, None, None
        dl = (y[None, :] / lam - 1.0) / (1.0 + np.exp(-f))
        return e, (ws[:, None] * dl).sum(axis=0), (ws[:, None] * dl * xs[:, None]).sum(axis=0) / (np.sqrt(2.0) * sd)


def main(seed=0, steps=200):
    rng = np.random.default_rng(seed)
    N, M, d = 20_000, 64, 2
    x = rng.uniform(-2, 2, (d, N))
    f_true = 1.5 * np.sin(2.0 * x[0]) * np.cos(1.5 * x[1]) + 0.5
    y = rng.poisson(np.logaddexp(0.0, f_true)).astype(np.float64)
    lik = PoissonSoftplus()
    ctx = _ffi.default_context()
    data = _ffi.DeviceData(ctx, x, None, np.float64)          # no observations on the device: the likelihood is the caller's

    def unpack(t):
        return np.exp(t[0]), np.exp(t[1:1 + d]), t[1 + d:1 + d + d * M].reshape(d, M, order="F"), t[1 + d + d * M:1 + d + d * M + M], \
            np.tril(t[1 + d + d * M + M:].reshape(M, M))

    def desc(t):
        var, il, z, m, A = unpack(t)
        return _ffi.make_desc(np.float64, _ffi.KERNEL_SE, var, il, z, m, A, 1e-5)

    z0 = x[:, rng.choice(N, M, replace=False)]
    t = np.concatenate([[0.0], np.zeros(d), z0.ravel(order="F"), np.zeros(M), np.eye(M).ravel()])
    d0, keep = desc(t)
    model = _ffi.DeviceModel(ctx, d0, keep)

    def value_and_grad(t):
        var, il, z, m, A = unpack(t)
        dd, keep = desc(t)
        model.update(dd, keep)
        mu, v = model.marginals(data)
        e, gmu, gv = lik.expectation(mu, v, y, 20, True)
        val, _, g = model.elbo_grad(data, 0, N, float(N), ext=(e, gmu, gv))
        grad = np.concatenate([[g["variance"] * var], np.asarray(g["inv_lengthscale"]) * il, np.asarray(g["z"]).ravel(order="F"),
                               np.asarray(g["m"]), np.tril(np.asarray(g["Lq"])).ravel()])
        return val, grad

    # Adam on the negative ELBO
    m1, m2, lr = np.zeros_like(t), np.zeros_like(t), 0.02
    t0 = time.perf_counter()
    for it in range(1, steps + 1):
        val, g = value_and_grad(t)
        m1 = 0.9 * m1 + 0.1 * (-g)
        m2 = 0.999 * m2 + 0.001 * g * g
        t = t - lr * (m1 / (1 - 0.9 ** it)) / (np.sqrt(m2 / (1 - 0.999 ** it)) + 1e-8)
        if it == 1 or it % 50 == 0:
            print(f"step {it:4d}  ELBO {val:12.2f}")
    dt = time.perf_counter() - t0
    var, il, z, m, A = unpack(t)
    print(f"{steps} steps in {dt:.2f} s ({dt / steps * 1e3:.1f} ms per step: marginals + host likelihood + device backward pass)")
    dd, keep = desc(t)
    post = ag.posterior(ag.SparseVariationalApproximation(ag.GP(var * ag.with_lengthscale(ag.SqExponentialKernel(), 1.0 / il))(z, 1e-5),
                                                          ag.MvNormal.from_cholesky(m, A)))
    xs = np.array([[-1.5, -0.5, 0.0, 0.8, 1.6], [0.3, -1.0, 0.0, 1.2, -0.4]])
    mu, v = post.mean_and_var(xs)
    truth = 1.5 * np.sin(2.0 * xs[0]) * np.cos(1.5 * xs[1]) + 0.5
    print("latent truth   :", np.round(truth, 2))
    print("posterior mean :", np.round(mu, 2), " sd:", np.round(np.sqrt(v), 2))


if __name__ == "__main__":
    main()
