"""The synthetic workloads of SURVEY §8d, in ONE place: bench.py, the examples, the oracle's `synth_problem`
(tests, golden fixtures, smoke) all draw their inputs here, so the benchmarked problem is the tested problem.
Pure numpy; nothing here evaluates a GP."""
from __future__ import annotations

import math

import numpy as np

# codes of include/svgp_mi355x.h
LIK_GAUSSIAN, LIK_BERNOULLI_LOGISTIC, LIK_POISSON_EXP, LIK_EXPONENTIAL_EXP, LIK_GAMMA_EXP, LIK_BERNOULLI_NORMCDF = 0, 1, 2, 3, 4, 5


def _observations(rng, x, lik):
    d, N = x.shape
    s = x.sum(axis=0) / math.sqrt(d)
    sigma2 = 0.3
    if lik == LIK_GAUSSIAN:
        y = np.sin(s) + math.sqrt(sigma2) * rng.standard_normal(N)
    elif lik in (LIK_BERNOULLI_LOGISTIC, LIK_BERNOULLI_NORMCDF):
        p = 1.0 / (1.0 + np.exp(-2.0 * np.sin(s)))
        y = (rng.random(N) < p).astype(np.float64)
    elif lik == LIK_POISSON_EXP:
        y = rng.poisson(np.exp(np.sin(s))).astype(np.float64)
    elif lik == LIK_EXPONENTIAL_EXP:
        y = rng.exponential(np.exp(np.sin(s)))   # numpy's argument is the scale, as Distributions.Exponential's
    else:
        sigma2 = 2.5  # the Gamma shape alpha travels in the likelihood-parameter slot
        y = rng.gamma(sigma2, np.exp(np.sin(s)))
    return y, sigma2


def synth_arrays(config_id: int, N: int, M: int, d: int, lik: int = LIK_GAUSSIAN, dtype=np.float64, jitter=None, shard: int = 0):
    """x ~ N(0, I_d) (d x N); z = the first M points of x + 1e-3 N(0, 1) (examples/a-regression/script.jl:69-70 picks
    inducing inputs from the data); lengthscales sqrt(d) (0.75 + 0.5 k / d), variance 1.3; m ~ 0.1 N(0, I);
    Lq = I + 0.05 tril(N(0,1)) / sqrt(M) with a positive diagonal; y from sin(sum x / sqrt(d)) through the likelihood;
    jitter 1e-5 (fp64) / 1e-3 (fp32).  Everything is rounded through `dtype` and returned as float64, so fp32 runs and
    the fp64 oracle see identical inputs.

    `shard` > 0 (data-parallel ranks): the model (z, m, Lq, hyper-parameters) is shard 0's, identical on every rank;
    only (x, y) are this shard's own draws."""
    rng = np.random.default_rng(20260313 + config_id)
    x = rng.standard_normal((d, N))
    zbase = x[:, :M] if M <= N else rng.standard_normal((d, M))  # more inducing points than data: fresh draws
    z = zbase + 1e-3 * rng.standard_normal((d, M))
    ell = math.sqrt(d) * (0.75 + 0.5 * np.arange(d) / d)
    m = 0.1 * rng.standard_normal(M)
    Lq = np.eye(M) + 0.05 * np.tril(rng.standard_normal((M, M))) / math.sqrt(M)
    Lq[np.diag_indices(M)] = np.abs(np.diag(Lq))
    if shard == 0:
        y, sigma2 = _observations(rng, x, lik)
    else:
        srng = np.random.default_rng([20260313 + config_id, shard])
        x = srng.standard_normal((d, N))
        y, sigma2 = _observations(srng, x, lik)
    if jitter is None:
        jitter = 1e-5 if np.dtype(dtype) == np.float64 else 1e-3
    rt = lambda a: np.asarray(a, dtype=dtype).astype(np.float64)
    return dict(x=rt(x), y=rt(y), z=rt(z), m=rt(m), Lq=rt(Lq), inv_lengthscale=1.0 / ell, variance=1.3, sigma2=sigma2,
                jitter=jitter)
