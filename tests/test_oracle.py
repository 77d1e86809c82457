"""Pins the CPU oracle: the reference's own input-independent assertions
(/root/reference/test/SparseVariationalApproximationModule.jl) re-stated over seeded inputs,
plus the derived known answers K1–K7 of SURVEY §8c.  CPU only."""
import math

import numpy as np
import pytest

import svgp_oracle as o
import svgp_oracle_mp as omp


def _elbo_case(seed=654321, N=20, M=5, jitter=1e-18):
    # test/SparseVariationalApproximationModule.jl:74-85 with numpy's RNG (the MersenneTwister
    # stream cannot be reproduced without Julia; every assertion is input-independent).
    rng = np.random.default_rng(seed)
    x = rng.random(N) * 10
    y = np.sin(x) + 0.9 * np.cos(x * 1.6) + 0.4 * rng.random(N)
    z = x[:M].copy()
    kernel = o.make_kernel([0.2, 0.6])
    return kernel, x, y, z, 0.1, jitter


def _optimal_sva(kernel, z, jitter, x, s2, y, centered):
    m, S = o.optimal_variational_posterior(kernel, z, jitter, x, s2, y)
    if centered:
        return o.SVA(kernel, z, m, np.linalg.cholesky(S), jitter=jitter, centered=True)
    me, Se = o.whiten(kernel, z, jitter, m, S)
    return o.SVA(kernel, z, me, np.linalg.cholesky(Se), jitter=jitter, centered=False)


def test_noncentered_equals_centered():
    # ref test :10-70 (Matérn-3/2, x = range(-1,1,5), z = range(-1,1,4), σ² = 1e-3, jitter 1e-6)
    kernel = o.Kernel(o.KERNEL_MATERN32, 1.0, [1.0])
    x = np.linspace(-1, 1, 5)
    z = np.linspace(-1, 1, 4)
    rng = np.random.default_rng(123456)
    Kxx = o.kernelmatrix(kernel, x) + 1e-3 * np.eye(5)
    y = np.linalg.cholesky(Kxx) @ rng.standard_normal(5)
    c = _optimal_sva(kernel, z, 1e-6, x, 1e-3, y, True)
    nc = _optimal_sva(kernel, z, 1e-6, x, 1e-3, y, False)
    assert o.prior_kl(nc) == pytest.approx(o.prior_kl(c), rel=1e-5)  # :61-65
    a = np.linspace(-1, 1, 6)
    b = rng.standard_normal(7)
    pc, pn = o.posterior(c), o.posterior(nc)
    np.testing.assert_allclose(o.mean(pn, a), o.mean(pc, a), rtol=1e-7, atol=1e-9)  # :66
    np.testing.assert_allclose(o.cov(pn, a, b), o.cov(pc, a, b), rtol=1e-6, atol=1e-9)  # :67-68
    assert o.elbo(nc, x, y, sigma2=1e-3) == pytest.approx(o.elbo(c, x, y, sigma2=1e-3), rel=1e-7)  # :69
    # interface conformance (:30-34, :54-58): var == diag(cov), mean_and_cov consistent, PSD
    for p in (pc, pn):
        C = o.cov(p, a)
        np.testing.assert_allclose(o.var(p, a), np.diag(C), atol=1e-10)
        mu, C2 = o.mean_and_cov(p, a)
        np.testing.assert_allclose(mu, o.mean(p, a), atol=1e-12)
        np.testing.assert_allclose(C2, C, atol=1e-12)
        np.testing.assert_allclose(C, C.T, atol=1e-12)
        assert np.linalg.eigvalsh(C).min() > -1e-9
        np.testing.assert_allclose(o.cov(p, a, a), C, atol=1e-10)


def test_elbo_basics():
    kernel, x, y, z, s2, jitter = _elbo_case(jitter=1e-10)
    sva = _optimal_sva(kernel, z, jitter, x, s2, y, False)
    e = o.elbo_finite_gp(sva, x, y, s2)
    assert isinstance(e, float)  # :87
    assert e <= o.exact_gp_logpdf(kernel, x, s2, y)  # :88
    with pytest.raises(RuntimeError, match="homoscedastic"):  # :90-91
        o.elbo_finite_gp(sva, x, y, np.full(len(y), 0.1))
    assert o.elbo(sva, x, y, lik=o.LIK_GAUSSIAN, sigma2=s2) == pytest.approx(e, abs=1e-10)  # :93-96


def test_K1_titsias_bound():
    kernel, x, y, z, s2, jitter = _elbo_case(jitter=1e-10)
    sva = _optimal_sva(kernel, z, jitter, x, s2, y, False)
    assert o.elbo(sva, x, y, sigma2=s2) == pytest.approx(o.titsias_bound(kernel, z, jitter, x, s2, y), rel=1e-9)


def test_gpr_equivalence_z_equals_x():
    # ref :99-134: z = x + optimal q (Centered) == exact GPR, atol 1e-10; elbo <= logpdf + 1e-5
    kernel, x, y, _, s2, _ = _elbo_case()
    jitter = 1e-12  # the reference uses the 1e-18 default; LAPACK in numpy needs a little more for N=20
    sva = _optimal_sva(kernel, x.copy(), jitter, x, s2, y, True)
    p = o.posterior(sva)
    mu, C = o.exact_gp_posterior(kernel, x, s2, y, x)
    np.testing.assert_allclose(o.mean(p, x), mu, atol=1e-6)
    np.testing.assert_allclose(o.cov(p, x), C, atol=1e-6)
    assert o.elbo(sva, x, y, sigma2=s2) <= o.exact_gp_logpdf(kernel, x, s2, y) + 1e-5
    # K5: gap shrinks with jitter (bound + trend, not equality)
    gaps = []
    for j in (1e-4, 1e-6, 1e-8):
        s = _optimal_sva(kernel, x.copy(), j, x, s2, y, False)
        gaps.append(o.exact_gp_logpdf(kernel, x, s2, y) - o.elbo(s, x, y, sigma2=s2))
    assert gaps[0] > gaps[1] > gaps[2] > -1e-6


def test_K2_gh_equals_analytic_for_gaussian():
    x, y, sva, s2 = o.synth_problem(7, 64, 8, 3)
    ea = o.elbo(sva, x, y, sigma2=s2)
    for n in (2, 5, 20):
        assert o.elbo(sva, x, y, sigma2=s2, quadrature_n=n) == pytest.approx(ea, rel=1e-12)


def test_K3_kl_closed_forms_agree():
    kernel, x, y, z, s2, _ = _elbo_case()
    c = _optimal_sva(kernel, z, 1e-8, x, s2, y, True)
    nc = _optimal_sva(kernel, z, 1e-8, x, s2, y, False)
    assert o.prior_kl(c) == pytest.approx(o.prior_kl(nc), rel=1e-7)


def test_K4_fusion_identity():
    x, y, sva, _ = o.synth_problem(3, 50, 12, 4)
    p = o.posterior(sva)
    A, Kuf = o.A_and_Kuf(p, x)
    np.testing.assert_allclose(Kuf.T @ p.alpha, A.T @ sva.m, atol=1e-11)


@pytest.mark.parametrize("family", [o.KERNEL_SE, o.KERNEL_MATERN32, o.KERNEL_MATERN52])
@pytest.mark.parametrize("lik", [o.LIK_GAUSSIAN, o.LIK_BERNOULLI_LOGISTIC, o.LIK_POISSON_EXP, o.LIK_BERNOULLI_NORMCDF])
def test_K6_mpmath(family, lik):
    x, y, sva, s2 = o.synth_problem(11 + family, 6, 4, 2, family=family, lik=lik)
    ref = omp.elbo(family, sva.kernel.variance, list(sva.kernel.inv_lengthscale), sva.z.T.tolist(),
                   sva.m.tolist(), sva.Lq.tolist(), sva.jitter, x.T.tolist(), y.tolist(), lik=lik,
                   sigma2=s2, num_data=60)
    t = o.elbo_terms(sva, x, y, lik=lik, sigma2=s2, num_data=60)
    assert t.elbo == pytest.approx(float(ref["elbo"]), rel=1e-11)
    assert t.kl == pytest.approx(float(ref["kl"]), rel=1e-12)
    np.testing.assert_allclose(t.mu, [float(v) for v in ref["mu"]], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(t.v, [float(v) for v in ref["v"]], rtol=1e-9, atol=1e-12)


def test_K7_gauss_hermite_rule():
    xs, ws = o.gausshermite(20)
    assert ws.sum() == pytest.approx(math.sqrt(math.pi), rel=1e-14)
    mx, mw = omp.gausshermite(20)
    np.testing.assert_allclose(xs, [float(t) for t in mx], atol=1e-13)
    np.testing.assert_allclose(ws, [float(t) for t in mw], rtol=1e-11)
    # SURVEY Appendix D constants
    assert sorted(abs(xs))[0] == pytest.approx(0.2453407083009, abs=1e-12)
    assert max(abs(xs)) == pytest.approx(5.3874808900112, abs=1e-12)


def test_minibatch_scale_and_negative_variance():
    x, y, sva, s2 = o.synth_problem(5, 40, 6, 2)
    t1 = o.elbo_terms(sva, x, y, sigma2=s2)
    t2 = o.elbo_terms(sva, x, y, sigma2=s2, num_data=100.5)  # SVA:357-358: float division
    assert t2.elbo == pytest.approx(t1.expectation * 100.5 / 40 - t1.kl, rel=1e-14)
    bad = o.SVA(sva.kernel, sva.z, sva.m, 1e-3 * np.eye(6), jitter=-2.0 * sva.kernel.variance * 0 + 1e-5)
    bad.kernel = o.Kernel(o.KERNEL_SE, -1.0, sva.kernel.inv_lengthscale)  # forces Kuu indefinite
    with pytest.raises(o.PosDefException):
        o.posterior(bad)


@pytest.mark.parametrize("lik", [o.LIK_POISSON_EXP, o.LIK_EXPONENTIAL_EXP, o.LIK_GAMMA_EXP])
def test_exp_link_closed_forms_match_numerical_integration(lik):
    """AnalyticExpectation of the exp-link likelihoods [GPLikelihoods]: E[exp(±f)] = exp(±mu + v/2).  Checked against
    adaptive quadrature of log p(y|f) N(f; mu, v) and against Gauss-Hermite with 40 points."""
    from scipy.integrate import quad

    rng = np.random.default_rng(5)
    mu, sd = rng.standard_normal(6) * 0.7, 0.2 + rng.random(6)
    alpha = 2.5
    y = np.array([0.0, 1.0, 3.0, 2.0, 5.0, 1.0]) if lik == o.LIK_POISSON_EXP else 0.1 + rng.random(6) * 3
    closed = o.expected_loglik(lik, mu, sd, y, sigma2=alpha)
    ref = 0.0
    for m_, s_, y_ in zip(mu, sd, y):
        f = lambda t: float(o.loglik(lik, np.array([t]), np.array([y_]), alpha)[0]) * np.exp(-0.5 * ((t - m_) / s_) ** 2) / (s_ * np.sqrt(2 * np.pi))
        ref += quad(f, m_ - 12 * s_, m_ + 12 * s_, epsabs=1e-13, epsrel=1e-13, limit=400)[0]
    assert closed == pytest.approx(ref, rel=1e-10)
    assert closed == pytest.approx(o.expected_loglik(lik, mu, sd, y, sigma2=alpha, quadrature_n=40), rel=1e-12)


def test_exponential_likelihood_is_scale_parametrised():
    """GPLikelihoods: (l::ExponentialLikelihood)(f) = Exponential(l.invlink(f)) and Distributions.Exponential(θ) has SCALE θ,
    pdf (1/θ) exp(-y/θ): log p(y|f) = -f - y exp(-f), the Gamma likelihood (shape α, scale exp f) at α = 1, and the
    AnalyticExpectation is -μ - y exp(v/2 - μ).  Checked against scipy.stats (an independent implementation of both
    distributions) and between likelihood codes 3 and 4."""
    from scipy import stats

    rng = np.random.default_rng(11)
    f = rng.standard_normal(50)
    y = rng.exponential(np.exp(f))
    np.testing.assert_allclose(o.loglik(o.LIK_EXPONENTIAL_EXP, f, y), stats.expon(scale=np.exp(f)).logpdf(y), rtol=1e-13)
    np.testing.assert_allclose(o.loglik(o.LIK_GAMMA_EXP, f, y, 1.0), o.loglik(o.LIK_EXPONENTIAL_EXP, f, y), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(o.loglik(o.LIK_GAMMA_EXP, f, y, 2.5), stats.gamma(a=2.5, scale=np.exp(f)).logpdf(y), rtol=1e-12)
    k = rng.poisson(np.exp(f)).astype(float)
    np.testing.assert_allclose(o.loglik(o.LIK_POISSON_EXP, f, k), stats.poisson(np.exp(f)).logpmf(k), rtol=1e-12)
    b = (rng.random(50) < 0.5).astype(float)
    np.testing.assert_allclose(o.loglik(o.LIK_BERNOULLI_LOGISTIC, f, b), stats.bernoulli(1 / (1 + np.exp(-f))).logpmf(b), rtol=1e-12)
    np.testing.assert_allclose(o.loglik(o.LIK_GAUSSIAN, f, y, 0.3), stats.norm(f, np.sqrt(0.3)).logpdf(y), rtol=1e-12)
    # BernoulliLikelihood(NormalCDFLink()): logpdf(Bernoulli(normcdf(f)), y), the reference's naive form where it is finite
    np.testing.assert_allclose(o.loglik(o.LIK_BERNOULLI_NORMCDF, f, b), stats.bernoulli(stats.norm.cdf(f)).logpmf(b), rtol=1e-12)
    mu, sd = rng.standard_normal(50) * 0.5, 0.3 + rng.random(50)
    for qn in (0, 30):
        assert o.expected_loglik(o.LIK_EXPONENTIAL_EXP, mu, sd, y, 1.0, qn) == pytest.approx(
            o.expected_loglik(o.LIK_GAMMA_EXP, mu, sd, y, 1.0, qn), rel=1e-13)
    ge, gg = (o.expected_loglik_grads(l, mu, sd**2, y, 1.0, 0) for l in (o.LIK_EXPONENTIAL_EXP, o.LIK_GAMMA_EXP))
    np.testing.assert_allclose(ge[0], gg[0], rtol=1e-13)
    np.testing.assert_allclose(ge[1], gg[1], rtol=1e-13)


def test_golden_fixtures_are_the_oracle():
    """tests/golden/*.npz (generated by tests/golden/make_golden.py) still equal the oracle: a regression guard for
    the oracle itself (the HIP library is compared with the same files on the GPU)."""
    import glob
    import os

    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
    assert len(paths) >= 10
    for path in paths:
        g = np.load(path)
        kernel = o.Kernel(int(g["family"]), float(g["variance"]), g["inv_lengthscale"])
        sva = o.SVA(kernel, g["z"], g["m"], g["Lq"], jitter=float(g["jitter"]), mean_const=float(g["mean_const"]),
                    centered=bool(int(g["centered"])))
        nd = float(g["num_data"])
        t = o.elbo_terms(sva, g["x"], g["y"], lik=int(g["lik"]), sigma2=float(g["sigma2"]), num_data=nd if nd > 0 else None,
                         quadrature_n=int(g["quadrature_n"]))
        assert t.elbo == pytest.approx(float(g["elbo"]), rel=1e-12), path
        assert t.kl == pytest.approx(float(g["kl"]), rel=1e-12)
        _, gr = o.elbo_grad(sva, g["x"], g["y"], lik=int(g["lik"]), sigma2=float(g["sigma2"]), num_data=nd if nd > 0 else None,
                            quadrature_n=int(g["quadrature_n"]))
        np.testing.assert_allclose(gr["Lq"], g["g_Lq"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(gr["z"], g["g_z"], rtol=1e-9, atol=1e-10)


def test_kl_matches_scipy():
    """Distributions.kldivergence(q, fz) (SVA:362) and the NonCentered closed form (SVA:364-373) against scipy.stats:
    KL(q || p) = -H(q) - E_q[log p] with H(q) from multivariate_normal.entropy() and
    E_q[log p] = log p(mean(q)) - tr(inv(cov p) cov q) / 2."""
    from scipy import stats

    x, y, nc, s2 = o.synth_problem(21, 30, 9, 2)
    M = nc.M
    # Centered: p = N(mean_const, Kuu + jitter I), q = N(m, Lq Lq')
    c = o.SVA(nc.kernel, nc.z, 0.3 + nc.m, 0.7 * nc.Lq, jitter=nc.jitter, mean_const=0.3, centered=True)
    Kuu = o.kuu(c)
    S = c.Lq @ c.Lq.T
    kl_ref = (-stats.multivariate_normal(c.m, S).entropy() - stats.multivariate_normal(np.full(M, 0.3), Kuu).logpdf(c.m)
              + 0.5 * np.trace(np.linalg.solve(Kuu, S)))
    assert o.prior_kl(c) == pytest.approx(kl_ref, rel=1e-10)
    # NonCentered: p = N(0, I)
    Sn = nc.Lq @ nc.Lq.T
    kl_nc = -stats.multivariate_normal(nc.m, Sn).entropy() - stats.multivariate_normal(np.zeros(M), np.eye(M)).logpdf(nc.m) + 0.5 * np.trace(Sn)
    assert o.prior_kl(nc) == pytest.approx(kl_nc, rel=1e-10)
