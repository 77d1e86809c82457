"""Second source for the parts of the oracle that scikit-learn cannot reach: TEST INFRASTRUCTURE, CPU only.

The oracle (oracle/svgp_oracle.py) is parity-UNPINNED against the reference itself (no Julia here).  tests/test_oracle_sklearn.py
pins kernels / exact GP / z = x through scikit-learn; this file pins, against scipy only (no oracle function inside any
reference value):
  (a) the Gauss-Hermite / analytic expectations E_{N(mu, v)}[log p(y | f)] of EVERY enumerated likelihood (SVA:355,
      GPLikelihoods conventions of oracle/CONVENTIONS.md) against adaptive quadrature of scipy.stats log-densities - shape /
      scale / rate / link misreadings (Gamma scale-vs-rate, Exponential scale, probit-vs-logit) change these numbers at the
      first digit;
  (b) the Centered `_prior_kl` (SVA:362, Distributions.kldivergence(q, fz)) against the dense textbook formula with
      numpy.linalg.slogdet / inv;
  (c) oracle.elbo_grad by central finite differences for the Centered parametrisation x every likelihood (the NonCentered x every
      likelihood x every kernel grid is tests/test_oracle_grad.py).
"""
import math

import numpy as np
import pytest
from scipy import stats
from scipy.integrate import quad
from scipy.special import expit

import svgp_oracle as o

pytestmark = pytest.mark.filterwarnings("ignore::scipy.integrate.IntegrationWarning")   # the requested 1e-14 is below quad's own round-off
ALPHA = 2.5   # Gamma shape
S2 = 0.37     # Gaussian noise variance


def _scipy_logp(lik, f, y):
    """log p(y | f) from scipy.stats distributions only."""
    if lik == o.LIK_GAUSSIAN:
        return stats.norm.logpdf(y, loc=f, scale=math.sqrt(S2))
    if lik == o.LIK_BERNOULLI_LOGISTIC:
        return stats.bernoulli.logpmf(int(y), expit(f))
    if lik == o.LIK_BERNOULLI_NORMCDF:
        return stats.bernoulli.logpmf(int(y), stats.norm.cdf(f))
    if lik == o.LIK_POISSON_EXP:
        return stats.poisson.logpmf(int(y), math.exp(f))
    if lik == o.LIK_EXPONENTIAL_EXP:
        return stats.expon.logpdf(y, scale=math.exp(f))
    if lik == o.LIK_GAMMA_EXP:
        return stats.gamma.logpdf(y, a=ALPHA, scale=math.exp(f))
    raise ValueError(lik)


def _param(lik):
    return {o.LIK_GAUSSIAN: S2, o.LIK_GAMMA_EXP: ALPHA}.get(lik, 1.0)


def _problem(lik, seed):
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal(5) * 0.6
    sd = 0.15 + 0.6 * rng.random(5)
    if lik in (o.LIK_BERNOULLI_LOGISTIC, o.LIK_BERNOULLI_NORMCDF):
        y = np.array([0.0, 1.0, 1.0, 0.0, 1.0])
    elif lik == o.LIK_POISSON_EXP:
        y = np.array([0.0, 2.0, 1.0, 4.0, 3.0])
    elif lik == o.LIK_GAUSSIAN:
        y = rng.standard_normal(5)
    else:
        y = 0.2 + 2.5 * rng.random(5)
    return mu, sd, y


def _quad_expectation(lik, mu, sd, y):
    tot = 0.0
    for m_, s_, y_ in zip(mu, sd, y):
        g = lambda t: _scipy_logp(lik, t, y_) * stats.norm.pdf(t, loc=m_, scale=s_)
        # |f - mu| <= 8.2 sd: beyond it the Gaussian weight is < 1e-15 and the naive scipy Bernoulli log-pmf reaches log(0)
        val, _ = quad(g, m_ - 8.2 * s_, m_ + 8.2 * s_, epsabs=1e-14, epsrel=1e-14, limit=500, points=[m_])
        tot += val
    return tot


ALL_LIKS = [o.LIK_GAUSSIAN, o.LIK_BERNOULLI_LOGISTIC, o.LIK_BERNOULLI_NORMCDF, o.LIK_POISSON_EXP, o.LIK_EXPONENTIAL_EXP, o.LIK_GAMMA_EXP]


@pytest.mark.parametrize("lik", ALL_LIKS)
def test_expected_loglik_against_scipy_quadrature(lik):
    mu, sd, y = _problem(lik, 100 + lik)
    ref = _quad_expectation(lik, mu, sd, y)
    p = _param(lik)
    # DefaultExpectationMethod (analytic where GPLikelihoods has a closed form, else GH-20)
    dflt = o.expected_loglik(lik, mu, sd, y, sigma2=p)
    closed_form = lik in (o.LIK_GAUSSIAN, o.LIK_POISSON_EXP, o.LIK_EXPONENTIAL_EXP, o.LIK_GAMMA_EXP)
    assert dflt == pytest.approx(ref, rel=1e-10 if closed_form else 2e-6)   # GH-20 of a Bernoulli log-pmf: its own quadrature error
    # Gauss-Hermite with many nodes converges to the integral for every likelihood: pins the rule (nodes, weights, sqrt(2) sigma x + mu,
    # the 1 / sqrt(pi)) together with the integrand
    gh = o.expected_loglik(lik, mu, sd, y, sigma2=p, quadrature_n=96)
    assert gh == pytest.approx(ref, rel=1e-10 if lik != o.LIK_BERNOULLI_NORMCDF else 1e-9)
    # and the rule itself against a polynomial it must integrate exactly: E[f^2] = mu^2 + v
    xs, ws = o.gausshermite(7)
    e2 = sum(w * (math.sqrt(2.0) * sd[0] * x_ + mu[0]) ** 2 for x_, w in zip(xs, ws)) / math.sqrt(math.pi)
    assert e2 == pytest.approx(mu[0] ** 2 + sd[0] ** 2, rel=1e-13)


@pytest.mark.parametrize("lik", ALL_LIKS)
def test_point_gradients_against_scipy_quadrature(lik):
    """dE/dmu and dE/dv (the adjoint seeds of the whole backward pass) by central differences of the scipy-only quadrature."""
    mu, sd, y = _problem(lik, 200 + lik)
    gmu, gv, _ = o.expected_loglik_grads(lik, mu, sd * sd, y, _param(lik), 96 if lik in (o.LIK_BERNOULLI_LOGISTIC, o.LIK_BERNOULLI_NORMCDF) else 0)
    h = 1e-5
    for i in range(len(mu)):
        e = lambda m_, v_: _quad_expectation(lik, np.array([m_]), np.array([math.sqrt(v_)]), y[i:i + 1])
        v = sd[i] ** 2
        assert gmu[i] == pytest.approx((e(mu[i] + h, v) - e(mu[i] - h, v)) / (2 * h), rel=2e-6, abs=2e-8)
        assert gv[i] == pytest.approx((e(mu[i], v + h) - e(mu[i], v - h)) / (2 * h), rel=2e-6, abs=2e-8)


def test_centered_prior_kl_against_dense_formula():
    x, y, nc, s2 = o.synth_problem(23, 30, 11, 3, family=o.KERNEL_MATERN52)
    c = o.SVA(nc.kernel, nc.z, 0.4 + nc.m, 0.6 * nc.Lq, jitter=nc.jitter, mean_const=0.25, centered=True)
    K = np.asarray(o.kernelmatrix(c.kernel, c.z), dtype=np.float64) + c.jitter * np.eye(c.M)   # cov(fz) = k(z, z) + jitter I (utils.jl:17)
    S = c.Lq @ c.Lq.T
    dm = np.full(c.M, 0.25) - c.m
    Ki = np.linalg.inv(K)
    kl = 0.5 * (np.trace(Ki @ S) + dm @ Ki @ dm - c.M + np.linalg.slogdet(K)[1] - np.linalg.slogdet(S)[1])
    assert o.prior_kl(c) == pytest.approx(kl, rel=1e-10)
    # the NonCentered closed form is the same formula at K = I, mean 0
    Sn = nc.Lq @ nc.Lq.T
    kln = 0.5 * (np.trace(Sn) + nc.m @ nc.m - nc.M - np.linalg.slogdet(Sn)[1])
    assert o.prior_kl(nc) == pytest.approx(kln, rel=1e-11)


def _fd(fun, x0, h=1e-6):
    g = np.zeros_like(x0, dtype=np.float64)
    it = np.nditer(x0, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        xp, xm = x0.copy(), x0.copy()
        xp[i] += h
        xm[i] -= h
        g[i] = (fun(xp) - fun(xm)) / (2 * h)
    return g


@pytest.mark.parametrize("lik,qn", [(o.LIK_GAUSSIAN, 0), (o.LIK_GAUSSIAN, 7), (o.LIK_BERNOULLI_LOGISTIC, 0), (o.LIK_BERNOULLI_NORMCDF, 0),
                                    (o.LIK_POISSON_EXP, 0), (o.LIK_EXPONENTIAL_EXP, 0), (o.LIK_GAMMA_EXP, 0), (o.LIK_GAMMA_EXP, 9)])
def test_centered_gradient_every_likelihood(lik, qn):
    family = [o.KERNEL_SE, o.KERNEL_MATERN32, o.KERNEL_MATERN52][lik % 3]
    x, y, nc, s2 = o.synth_problem(51 + lik, 30, 6, 2, family=family, lik=lik)
    sva = o.SVA(nc.kernel, nc.z, nc.m + 0.2, 0.8 * nc.Lq, jitter=1e-4, mean_const=0.1, centered=True)
    kw = dict(lik=lik, num_data=70.0, quadrature_n=qn)
    val, g = o.elbo_grad(sva, x, y, sigma2=s2, **kw)
    assert val == pytest.approx(o.elbo(sva, x, y, sigma2=s2, **kw), rel=1e-12)

    def with_(**ch):
        k = o.Kernel(family, ch.get("variance", sva.kernel.variance), ch.get("il", sva.kernel.inv_lengthscale))
        s = o.SVA(k, ch.get("z", sva.z), ch.get("m", sva.m), ch.get("Lq", sva.Lq), jitter=sva.jitter, mean_const=ch.get("c", sva.mean_const),
                  centered=True)
        return o.elbo(s, x, y, sigma2=ch.get("s2", s2), **kw)

    tol = dict(rtol=5e-6, atol=5e-6)
    np.testing.assert_allclose(g["m"], _fd(lambda t: with_(m=t), sva.m.copy()), **tol)
    np.testing.assert_allclose(g["z"], _fd(lambda t: with_(z=t), sva.z.copy()), **tol)
    np.testing.assert_allclose(g["inv_lengthscale"], _fd(lambda t: with_(il=t), sva.kernel.inv_lengthscale.copy()), **tol)
    np.testing.assert_allclose(g["Lq"], np.tril(_fd(lambda t: with_(Lq=np.tril(t)), sva.Lq.copy())), **tol)
    assert g["variance"] == pytest.approx(float(_fd(lambda t: with_(variance=float(t[0])), np.array([sva.kernel.variance]))[0]), rel=5e-6, abs=5e-6)
    assert g["mean_const"] == pytest.approx(float(_fd(lambda t: with_(c=float(t[0])), np.array([0.1]))[0]), rel=5e-6, abs=5e-6)
    if lik in (o.LIK_GAUSSIAN, o.LIK_GAMMA_EXP):
        assert g["lik_sigma2"] == pytest.approx(float(_fd(lambda t: with_(s2=float(t[0])), np.array([s2]))[0]), rel=5e-6, abs=5e-6)
