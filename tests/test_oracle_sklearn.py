"""Second source for the oracle: scikit-learn's Gaussian-process kernels and exact GP regression share
no code with this repository and none with its author.  Checked here:

* ``oracle.kernelmatrix`` (SE / Matern-3/2 / Matern-5/2, isotropic and ARD, with a variance) against
  ``sklearn.gaussian_process.kernels`` (``ConstantKernel * RBF`` / ``Matern(nu=1.5 / 2.5)``, ``length_scale`` = 1 / inverse
  lengthscale: the KernelFunctions convention ``k o ARDTransform(1 ./ l)``);
* ``exact_gp_logpdf`` / ``exact_gp_posterior`` against ``GaussianProcessRegressor(alpha=sigma2, optimizer=None)``;
* through the known answer K1 (z = x with the optimal q makes the SVGP posterior the exact GP posterior and the ELBO the
  log marginal likelihood up to the jitter) the oracle's ``elbo`` / ``posterior`` themselves against that third-party number.

This does not pin the oracle to the reference (only a Julia run can: oracle/reference_julia.jl); it does catch a wrong
dependency convention of the kind round 1 had ("Oracle").  CPU only."""
import numpy as np
import pytest

import svgp_oracle as o

sk = pytest.importorskip("sklearn.gaussian_process")
from sklearn.gaussian_process import GaussianProcessRegressor  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern  # noqa: E402

FAMILIES = [(o.KERNEL_SE, None), (o.KERNEL_MATERN32, 1.5), (o.KERNEL_MATERN52, 2.5)]


def _sk_kernel(nu, variance, inv_l):
    ls = 1.0 / np.asarray(inv_l, dtype=np.float64)
    ls = float(ls[0]) if ls.size == 1 else ls
    base = RBF(length_scale=ls) if nu is None else Matern(length_scale=ls, nu=nu)
    return ConstantKernel(constant_value=variance) * base


def _inputs(seed, d, na, nb):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((d, na)), rng.standard_normal((d, nb))


@pytest.mark.parametrize("family,nu", FAMILIES)
@pytest.mark.parametrize("d,ard", [(1, False), (3, False), (3, True), (8, True), (16, True)])
def test_kernelmatrix_matches_sklearn(family, nu, d, ard):
    rng = np.random.default_rng(100 * d + family)
    inv_l = (0.4 + rng.random(d)) if ard else np.full(d, 0.7)
    variance = 1.3
    a, b = _inputs(7 + d, d, 23, 17)
    k = o.Kernel(family, variance, inv_l)
    ks = _sk_kernel(nu, variance, inv_l if ard else inv_l[:1])
    # sklearn takes points as rows (n, d); the oracle takes ColVecs (d, n)
    np.testing.assert_allclose(o.kernelmatrix(k, a, b), ks(a.T, b.T), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(o.kernelmatrix(k, a), ks(a.T), rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(o.kernelmatrix_diag(k, a), ks.diag(a.T), rtol=1e-15)


def test_vector_inputs_are_one_dimensional_points():
    # `Vector{<:Real}` inputs (d = 1): the regression example's x (examples/a-regression/script.jl:34)
    x = np.linspace(-1.0, 1.0, 11)
    k = o.Kernel(o.KERNEL_MATERN52, 0.8, [1.0 / 0.3])
    np.testing.assert_allclose(o.kernelmatrix(k, x), _sk_kernel(2.5, 0.8, [1.0 / 0.3])(x[:, None]), rtol=1e-13, atol=1e-14)


def _gpr(nu, variance, inv_l, sigma2, x, y):
    g = GaussianProcessRegressor(kernel=_sk_kernel(nu, variance, inv_l), alpha=sigma2, optimizer=None, normalize_y=False)
    g.fit(x.T, y)
    return g


@pytest.mark.parametrize("family,nu", FAMILIES)
def test_exact_gp_matches_sklearn_regressor(family, nu):
    rng = np.random.default_rng(20 + family)
    d, n, ns = 2, 40, 9
    x = rng.standard_normal((d, n))
    xs = rng.standard_normal((d, ns))
    y = np.sin(x.sum(0)) + 0.3 * rng.standard_normal(n)
    inv_l, variance, sigma2 = np.array([0.9, 0.6]), 1.1, 0.2
    k = o.Kernel(family, variance, inv_l)
    g = _gpr(nu, variance, inv_l, sigma2, x, y)
    assert o.exact_gp_logpdf(k, x, sigma2, y) == pytest.approx(g.log_marginal_likelihood(), rel=1e-12)
    mu, C = o.exact_gp_posterior(k, x, sigma2, y, xs)
    mu_s, C_s = g.predict(xs.T, return_cov=True)
    np.testing.assert_allclose(mu, mu_s, rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(C, C_s, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("family,nu", FAMILIES)
@pytest.mark.parametrize("centered", [False, True])
def test_elbo_and_posterior_at_z_equals_x_match_sklearn(family, nu, centered):
    # ref test/SparseVariationalApproximationModule.jl:99-130: z = x and the optimal q give the exact posterior; the ELBO then is the
    # log marginal likelihood up to the jitter on Kuu (a bound + a gap that shrinks with the jitter)
    rng = np.random.default_rng(300 + family)
    n = 24
    x = np.sort(rng.random(n) * 6.0)
    y = np.sin(x) + 0.9 * np.cos(1.6 * x) + 0.4 * rng.random(n)
    inv_l, variance, sigma2, jitter = np.array([0.8]), 1.2, 0.1, 1e-9
    k = o.Kernel(family, variance, inv_l)
    m, S = o.optimal_variational_posterior(k, x, jitter, x, sigma2, y)
    if centered:
        sva = o.SVA(k, x, m, np.linalg.cholesky(S), jitter=jitter, centered=True)
    else:
        me, Se = o.whiten(k, x, jitter, m, S)
        sva = o.SVA(k, x, me, np.linalg.cholesky(Se), jitter=jitter, centered=False)
    g = _gpr(nu, variance, inv_l, sigma2, x[None, :], y)
    lml = g.log_marginal_likelihood()
    val = o.elbo(sva, x, y, sigma2=sigma2)
    assert val <= lml + 1e-9                     # ref :88, :132-133
    assert abs(val - lml) < 5e-5 * abs(lml)      # the jitter's gap
    xs = np.linspace(0.0, 6.0, 13)
    post = o.posterior(sva)
    mu_s, C_s = g.predict(xs[:, None], return_cov=True)
    np.testing.assert_allclose(o.mean(post, xs), mu_s, atol=2e-6)
    np.testing.assert_allclose(o.cov(post, xs), C_s, atol=2e-6)
