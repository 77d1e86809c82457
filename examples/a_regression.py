#!/usr/bin/env python
"""The reference's examples/a-regression/script.jl driven through the MI355X library: N = 10 000 noisy samples of
g(x) = sin(3 pi x) + 0.3 cos(9 pi x) + 0.5 sin(7 pi x), M = 20 inducing points (the first M inputs), NonCentered SVGP with
an SE kernel (softplus-constrained variance and lengthscale, script.jl:55-63), Gaussian noise 0.3 and jitter 1e-5
(script.jl:89-90), minibatches of 100 points and 300 Adam steps (script.jl:176-195).  Parameters, z, m and the factor A
are updated with the gradients the library returns (what Zygote returns in the reference); the softplus chain rule is
host-side, as it stays Julia-side under the binding.

    python examples/a_regression.py          # needs an MI355X; prints the minibatch loss and the final full-data ELBO
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "approximategps.jl_amd"))
import approxgp as ag  # noqa: E402
from approxgp import _ffi  # noqa: E402


def softplus(v):
    return np.log1p(np.exp(-abs(v))) + max(v, 0.0)


def invsoftplus(v):
    return v + np.log(-np.expm1(-v))


def main(steps=300, batch=100, lr=0.01, seed=1234):
    rng = np.random.default_rng(seed)
    N, M = 10_000, 20
    x = rng.uniform(-1, 1, N)
    y = np.sin(3 * np.pi * x) + 0.3 * np.cos(9 * np.pi * x) + 0.5 * np.sin(7 * np.pi * x) + 0.3 * rng.standard_normal(N)
    lik_noise, jitter = 0.3, 1e-5
    theta = {"k": np.array([invsoftplus(1.3), invsoftplus(0.3)]), "z": x[:M].copy(), "m": np.zeros(M), "A": np.eye(M)}
    ctx = _ffi.default_context()
    data = _ffi.DeviceData(ctx, x, y, np.float64)          # uploaded once; every step evaluates a window of it
    adam = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in theta.items()}

    def desc():
        var, ell = softplus(theta["k"][0]), softplus(theta["k"][1])
        return _ffi.make_desc(np.float64, _ffi.KERNEL_SE, var, [1.0 / ell], theta["z"], theta["m"], np.tril(theta["A"]), jitter,
                              likelihood=_ffi.LIK_GAUSSIAN, lik_sigma2=lik_noise), var, ell

    (d0, keep), _, _ = desc()
    model = _ffi.DeviceModel(ctx, d0, keep)
    for step in range(1, steps + 1):
        (d, keep), var, ell = desc()
        model.update(d, keep)
        off = int(rng.integers(0, N - batch))
        val, _, g = model.elbo_grad(data, off, batch, float(N), z_shape=(M,))
        sig = lambda t: 1.0 / (1.0 + np.exp(-t))            # d softplus / dt
        grads = {  # of the LOSS = -elbo, w.r.t. the unconstrained parameters
            "k": -np.array([g["variance"] * sig(theta["k"][0]),
                            g["inv_lengthscale"][0] * (-1.0 / ell**2) * sig(theta["k"][1])]),
            "z": -np.asarray(g["z"]), "m": -np.asarray(g["m"]), "A": -np.tril(np.asarray(g["Lq"])),
        }
        for k in theta:                                      # Adam (script.jl:176 `ADAM(0.01)`)
            m1, m2 = adam[k]
            m1[...] = 0.9 * m1 + 0.1 * grads[k]
            m2[...] = 0.999 * m2 + 0.001 * grads[k] ** 2
            theta[k] = theta[k] - lr * (m1 / (1 - 0.9**step)) / (np.sqrt(m2 / (1 - 0.999**step)) + 1e-8)
        if step % 50 == 0 or step == 1:
            print(f"step {step:4d}  minibatch loss {-val:12.3f}")
    (d, keep), var, ell = desc()
    model.update(d, keep)
    full, _ = model.elbo(data, 0, N, float(N))
    print(f"final: variance {var:.3f}, lengthscale {ell:.3f}, full-data ELBO {full:.2f}")
    # prediction through the reference-shaped API
    f = ag.GP(var * ag.with_lengthscale(ag.SqExponentialKernel(), ell))
    sva = ag.SparseVariationalApproximation(f(theta["z"], jitter), ag.MvNormal.from_cholesky(theta["m"], np.tril(theta["A"])))
    mu, v = ag.posterior(sva, ctx=ctx).mean_and_var(np.linspace(-1, 1, 5))
    print("posterior mean at -1, -0.5, 0, 0.5, 1:", np.round(mu, 3), " var:", np.round(v, 4))
    model.free()
    data.free()
    return full


if __name__ == "__main__":
    main()
