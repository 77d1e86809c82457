"""50-digit mpmath restatement of the NonCentered SVGP ELBO for tiny cases (known answer K6).

TEST INFRASTRUCTURE ONLY (see oracle/svgp_oracle.py header; PARITY UNPINNED applies here too).
Independent of numpy/LAPACK: its own Cholesky, substitutions and Gauss–Hermite rule, so it
pins the fp64 numpy oracle itself.  Follows /root/reference/src/SparseVariationalApproximationModule.jl
:160-187 (posterior), :215-219/:246-253 (Kuf, A, μ, v), :340-360 (ELBO), :364-373 (KL).
Pure-Python loops: keep N ≤ ~16, M ≤ ~8.
"""
from __future__ import annotations

import mpmath as mp

mp.mp.dps = 50

SE, MATERN32, MATERN52 = 0, 1, 2
GAUSSIAN, BERNOULLI_LOGISTIC, POISSON_EXP, BERNOULLI_NORMCDF = 0, 1, 2, 5


def _k(family, variance, inv_l, a, b):
    r2 = mp.mpf(0)
    for k in range(len(a)):
        t = (mp.mpf(a[k]) - mp.mpf(b[k])) * mp.mpf(inv_l[k])
        r2 += t * t
    if family == SE:
        return variance * mp.exp(-r2 / 2)
    r = mp.sqrt(r2)
    if family == MATERN32:
        s = mp.sqrt(3) * r
        return variance * (1 + s) * mp.exp(-s)
    s = mp.sqrt(5) * r
    return variance * (1 + s + mp.mpf(5) / 3 * r2) * mp.exp(-s)


def _chol(K):
    n = len(K)
    L = [[mp.mpf(0)] * n for _ in range(n)]
    for j in range(n):
        s = K[j][j] - sum(L[j][k] ** 2 for k in range(j))
        L[j][j] = mp.sqrt(s)
        for i in range(j + 1, n):
            L[i][j] = (K[i][j] - sum(L[i][k] * L[j][k] for k in range(j))) / L[j][j]
    return L


def _fwd(L, b):
    n = len(b)
    x = [mp.mpf(0)] * n
    for i in range(n):
        x[i] = (b[i] - sum(L[i][k] * x[k] for k in range(i))) / L[i][i]
    return x


def gausshermite(n):
    """Nodes/weights for weight exp(-x^2): roots of H_n by polishing, w = 2^{n-1} n! √π / (n² H_{n-1}(x)²)."""
    import numpy as np

    guess, _ = np.polynomial.hermite.hermgauss(n)
    xs, ws = [], []
    for g in guess:
        x = mp.findroot(lambda t: mp.hermite(n, t), mp.mpf(float(g)))
        xs.append(x)
        ws.append(mp.mpf(2) ** (n - 1) * mp.factorial(n) * mp.sqrt(mp.pi) / (n * n * mp.hermite(n - 1, x) ** 2))
    return xs, ws


def _loglik(lik, f, y, sigma2):
    if lik == GAUSSIAN:
        return -(mp.log(2 * mp.pi) + mp.log(sigma2) + (y - f) ** 2 / sigma2) / 2
    if lik == BERNOULLI_LOGISTIC:
        p = 1 / (1 + mp.exp(-f))
        return mp.log(p) if y > 0.5 else mp.log(1 - p)
    if lik == BERNOULLI_NORMCDF:   # the reference's own form: logpdf(Bernoulli(normcdf(f)), y), at 50 digits
        p = mp.ncdf(f)
        return mp.log(p) if y > 0.5 else mp.log(1 - p)
    return y * f - mp.exp(f) - mp.loggamma(y + 1)


def elbo(family, variance, inv_l, z, m, Lq, jitter, x, y, lik=GAUSSIAN, sigma2=1.0,
         num_data=None, quadrature_n=0, mean_const=0.0):
    """z: list of M points (each a list of d), x: list of n points.  Returns dict of mp values."""
    M, n = len(z), len(x)
    variance, sigma2 = mp.mpf(variance), mp.mpf(sigma2)
    Kuu = [[_k(family, variance, inv_l, z[i], z[j]) + (mp.mpf(jitter) if i == j else 0) for j in range(M)]
           for i in range(M)]
    Lk = _chol(Kuu)
    mm = [mp.mpf(v) for v in m]
    L = [[mp.mpf(Lq[i][j]) if j <= i else mp.mpf(0) for j in range(M)] for i in range(M)]
    if quadrature_n == 0 and lik in (BERNOULLI_LOGISTIC, BERNOULLI_NORMCDF):
        quadrature_n = 20
    if quadrature_n:
        xs, ws = gausshermite(quadrature_n)
    E = mp.mpf(0)
    mus, vs = [], []
    for p in range(n):
        kcol = [_k(family, variance, inv_l, z[i], x[p]) for i in range(M)]
        a = _fwd(Lk, kcol)
        mu = mp.mpf(mean_const) + sum(a[i] * mm[i] for i in range(M))
        c = [sum(L[i][j] * a[i] for i in range(j, M)) for j in range(M)]
        v = variance - sum(t * t for t in a) + sum(t * t for t in c) + mp.mpf("1e-18")
        mus.append(mu)
        vs.append(v)
        yp = mp.mpf(y[p])
        if quadrature_n == 0 and lik == GAUSSIAN:
            E += -(mp.log(2 * mp.pi) + mp.log(sigma2) + ((yp - mu) ** 2 + v) / sigma2) / 2
        elif quadrature_n == 0 and lik == POISSON_EXP:
            E += yp * mu - mp.exp(mu + v / 2) - mp.loggamma(yp + 1)
        else:
            sd = mp.sqrt(v)
            E += sum(w * _loglik(lik, mp.sqrt(2) * sd * t + mu, yp, sigma2) for t, w in zip(xs, ws)) / mp.sqrt(mp.pi)
    kl = (sum(L[i][j] ** 2 for i in range(M) for j in range(i + 1)) + sum(t * t for t in mm) - M
          - 2 * sum(mp.log(L[i][i]) for i in range(M))) / 2
    scale = (mp.mpf(num_data) if num_data is not None else mp.mpf(n)) / n
    return {"elbo": E * scale - kl, "E": E, "kl": kl, "mu": mus, "v": vs}
