#!/usr/bin/env python
"""The reference's examples/b-classification/script.jl through the MI355X library: binary labels drawn from a latent GP
(SE kernel, Bernoulli-logistic likelihood), N = 30 training points, M = 15 inducing points (script.jl:57-86), jitter 1e-3
(script.jl:114), all parameters (kernel variance and precision, z, m, A) optimised with L-BFGS on the negative ELBO
(script.jl:130-142) using the library's value-and-gradient (20-point Gauss-Hermite quadrature), then posterior samples
`rand(post(x, 1e-6), 20)` pushed through the logistic link (script.jl:151-156).

    python examples/b_classification.py      # needs an MI355X
"""
import os
import sys

import numpy as np
from scipy.optimize import minimize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd"))
import approxgp as ag  # noqa: E402
from approxgp import _ffi  # noqa: E402


def main(seed=1):
    rng = np.random.default_rng(seed)
    x_true = np.arange(0.0, 6.0 + 1e-9, 0.02)
    # a draw from the true latent GP: variance 10, ScaleTransform 0.1... the script samples it with AbstractGPs; numpy here
    k_true = lambda a, b: 10.0 * np.exp(-0.5 * (0.9 * (a[:, None] - b[None, :])) ** 2)
    f_true = np.linalg.cholesky(k_true(x_true, x_true) + 1e-6 * np.eye(x_true.size)) @ rng.standard_normal(x_true.size)
    y_true = (rng.random(x_true.size) < 1.0 / (1.0 + np.exp(-f_true))).astype(np.float64)
    N, M, jitter = 30, 15, 1e-3
    mask = np.sort(rng.choice(x_true.size, N, replace=False))
    x, y = x_true[mask], y_true[mask]
    ctx = _ffi.default_context()
    data = _ffi.DeviceData(ctx, x, y, np.float64)

    def unpack(t):   # positive parameters through exp, as ParameterHandling.positive does
        return np.exp(t[0]), np.exp(t[1]), t[2:2 + M], t[2 + M:2 + 2 * M], np.tril(t[2 + 2 * M:].reshape(M, M))

    def desc(t):
        var, prec, z, m, A = unpack(t)
        return _ffi.make_desc(np.float64, _ffi.KERNEL_SE, var, [prec], z, m, A, jitter, likelihood=_ffi.LIK_BERNOULLI_LOGISTIC)

    t0 = np.concatenate([np.log([rng.random() + 0.1, rng.random() + 0.1]), rng.uniform(0, 6, M), np.zeros(M), np.eye(M).ravel()])
    d0, keep = desc(t0)
    model = _ffi.DeviceModel(ctx, d0, keep)

    def loss_and_grad(t):
        var, prec, z, m, A = unpack(t)
        d, keep = desc(t)
        try:
            model.update(d, keep)
            val, _, g = model.elbo_grad(data, 0, N, float(N), z_shape=(M,))
        except (_ffi.PosDefException, _ffi.DomainError):
            return 1e10, np.zeros_like(t)     # reject the step, as Optim's line search would
        gA = np.tril(np.asarray(g["Lq"]))
        grad = np.concatenate([[g["variance"] * var, g["inv_lengthscale"][0] * prec], np.asarray(g["z"]), np.asarray(g["m"]), gA.ravel()])
        return -val, -grad

    res = minimize(loss_and_grad, t0, jac=True, method="L-BFGS-B", options={"maxiter": 4000})
    var, prec, z, m, A = unpack(res.x)
    print(f"L-BFGS: {res.nit} iterations, -ELBO {loss_and_grad(t0)[0]:.3f} -> {res.fun:.3f}; variance {var:.3f}, precision {prec:.3f}")
    f = ag.GP(var * ag.TransformedKernel(ag.SqExponentialKernel(), ag.ScaleTransform(prec)))
    A = A + 1e-9 * np.eye(M) * (np.abs(np.diag(A)) < 1e-9)   # guard a zero diagonal of an untouched entry
    post = ag.posterior(ag.SparseVariationalApproximation(f(z, jitter), ag.MvNormal.from_cholesky(m, A)), ctx=ctx)
    xs = x_true[::30]
    samples = post.rand(xs, 20, jitter=1e-6, rng=rng)                 # rand(post(x, 1e-6), 20)
    p_mean = (1.0 / (1.0 + np.exp(-samples))).mean(axis=1)
    truth = 1.0 / (1.0 + np.exp(-f_true[::30]))
    print("x          :", np.round(xs, 2))
    print("true p(y=1):", np.round(truth, 2))
    print("posterior  :", np.round(p_mean, 2))
    model.free()
    data.free()
    return res.fun


if __name__ == "__main__":
    main()
