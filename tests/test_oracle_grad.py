"""Pins the oracle's analytic ELBO gradient (oracle/svgp_oracle.py: elbo_grad) by central finite differences of
the oracle's own elbo() — the check the reference applies to its Laplace gradients
(test/LaplaceApproximationModule.jl:51-53, FiniteDifferences vs Zygote)."""
import numpy as np
import pytest

import svgp_oracle as o


def _fd(fun, x0, h):
    g = np.zeros_like(x0, dtype=np.float64)
    it = np.nditer(x0, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        xp, xm = x0.copy(), x0.copy()
        xp[i] += h
        xm[i] -= h
        g[i] = (fun(xp) - fun(xm)) / (2 * h)
    return g


@pytest.mark.parametrize("family", [o.KERNEL_SE, o.KERNEL_MATERN32, o.KERNEL_MATERN52])
@pytest.mark.parametrize("lik,qn", [(o.LIK_GAUSSIAN, 0), (o.LIK_BERNOULLI_LOGISTIC, 0), (o.LIK_POISSON_EXP, 0), (o.LIK_GAUSSIAN, 7),
                                    (o.LIK_EXPONENTIAL_EXP, 0), (o.LIK_GAMMA_EXP, 0), (o.LIK_GAMMA_EXP, 9), (o.LIK_BERNOULLI_NORMCDF, 0)])
def test_gradient_matches_finite_differences(family, lik, qn):
    x, y, sva, s2 = o.synth_problem(31 + family, 40, 7, 3, family=family, lik=lik)
    sva.mean_const = 0.2
    kw = dict(lik=lik, num_data=100.0, quadrature_n=qn)
    val, g = o.elbo_grad(sva, x, y, sigma2=s2, **kw)
    assert val == pytest.approx(o.elbo(sva, x, y, sigma2=s2, **kw), rel=1e-13)

    def with_(**ch):
        k = o.Kernel(family, ch.get("variance", sva.kernel.variance), ch.get("il", sva.kernel.inv_lengthscale))
        s = o.SVA(k, ch.get("z", sva.z), ch.get("m", sva.m), ch.get("Lq", sva.Lq), jitter=sva.jitter,
                  mean_const=ch.get("c", sva.mean_const))
        return o.elbo(s, x, y, sigma2=ch.get("s2", s2), **kw)

    tol = dict(rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(g["m"], _fd(lambda t: with_(m=t), sva.m.copy(), 1e-6), **tol)
    np.testing.assert_allclose(g["z"], _fd(lambda t: with_(z=t), sva.z.copy(), 1e-6), **tol)
    np.testing.assert_allclose(g["inv_lengthscale"], _fd(lambda t: with_(il=t), sva.kernel.inv_lengthscale.copy(), 1e-6), **tol)
    fdL = _fd(lambda t: with_(Lq=np.tril(t)), sva.Lq.copy(), 1e-6)
    np.testing.assert_allclose(g["Lq"], np.tril(fdL), **tol)
    assert g["variance"] == pytest.approx(float(_fd(lambda t: with_(variance=float(t[0])), np.array([sva.kernel.variance]), 1e-6)[0]), rel=2e-6, abs=2e-6)
    assert g["mean_const"] == pytest.approx(float(_fd(lambda t: with_(c=float(t[0])), np.array([0.2]), 1e-6)[0]), rel=2e-6, abs=2e-6)
    if lik in (o.LIK_GAUSSIAN, o.LIK_GAMMA_EXP):   # the likelihood parameter: sigma^2 / Gamma shape alpha
        assert g["lik_sigma2"] == pytest.approx(float(_fd(lambda t: with_(s2=float(t[0])), np.array([s2]), 1e-6)[0]), rel=2e-6, abs=2e-6)


@pytest.mark.parametrize("family,lik", [(o.KERNEL_SE, o.LIK_GAUSSIAN), (o.KERNEL_MATERN52, o.LIK_BERNOULLI_LOGISTIC)])
def test_centered_gradient_matches_finite_differences(family, lik):
    x, y, nc, s2 = o.synth_problem(41 + family, 35, 6, 2, family=family, lik=lik)
    sva = o.SVA(nc.kernel, nc.z, nc.m + 0.3, 0.7 * nc.Lq, jitter=1e-4, mean_const=0.15, centered=True)
    kw = dict(lik=lik, num_data=80.0)
    val, g = o.elbo_grad(sva, x, y, sigma2=s2, **kw)
    assert val == pytest.approx(o.elbo(sva, x, y, sigma2=s2, **kw), rel=1e-12)

    def with_(**ch):
        k = o.Kernel(family, ch.get("variance", sva.kernel.variance), ch.get("il", sva.kernel.inv_lengthscale))
        s = o.SVA(k, ch.get("z", sva.z), ch.get("m", sva.m), ch.get("Lq", sva.Lq), jitter=sva.jitter,
                  mean_const=ch.get("c", sva.mean_const), centered=True)
        return o.elbo(s, x, y, sigma2=s2, **kw)

    tol = dict(rtol=5e-6, atol=5e-6)
    np.testing.assert_allclose(g["m"], _fd(lambda t: with_(m=t), sva.m.copy(), 1e-6), **tol)
    np.testing.assert_allclose(g["z"], _fd(lambda t: with_(z=t), sva.z.copy(), 1e-6), **tol)
    np.testing.assert_allclose(g["inv_lengthscale"], _fd(lambda t: with_(il=t), sva.kernel.inv_lengthscale.copy(), 1e-6), **tol)
    np.testing.assert_allclose(g["Lq"], np.tril(_fd(lambda t: with_(Lq=np.tril(t)), sva.Lq.copy(), 1e-6)), **tol)
    assert g["variance"] == pytest.approx(float(_fd(lambda t: with_(variance=float(t[0])), np.array([sva.kernel.variance]), 1e-6)[0]), rel=5e-6, abs=5e-6)
    assert g["mean_const"] == pytest.approx(float(_fd(lambda t: with_(c=float(t[0])), np.array([0.15]), 1e-6)[0]), rel=5e-6, abs=5e-6)


def test_point_gradient_form_equals_the_fused_form():
    """elbo_grad_from_point_grads (the oracle side of svgp_elbo_grad_ext) fed with the oracle's own point gradients."""
    x, y, sva, s2 = o.synth_problem(77, 60, 9, 2, family=o.KERNEL_MATERN32, lik=o.LIK_BERNOULLI_LOGISTIC)
    val, g = o.elbo_grad(sva, x, y, lik=o.LIK_BERNOULLI_LOGISTIC, num_data=200.0)
    mu, sd = o.marginals(o.posterior(sva), x)
    sum_e = o.expected_loglik(o.LIK_BERNOULLI_LOGISTIC, mu, sd, y)
    gmu, gv, _ = o.expected_loglik_grads(o.LIK_BERNOULLI_LOGISTIC, mu, sd * sd, y)
    val2, g2 = o.elbo_grad_from_point_grads(sva, x, sum_e, gmu, gv, num_data=200.0)
    assert val2 == pytest.approx(val, rel=1e-13)
    for k in ("z", "m", "Lq", "inv_lengthscale"):
        np.testing.assert_allclose(g2[k], g[k], rtol=1e-10, atol=1e-12)
