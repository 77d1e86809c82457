"""GPU parity of svgp_elbo_grad against the oracle's analytic gradient (itself pinned by finite differences in
tests/test_oracle_grad.py).  Tolerances: fp64 relative 1e-6 of the gradient's max-norm per block, fp32 2e-3."""
import numpy as np
import pytest

import svgp_oracle as o
from approxgp import _ffi
from helpers import device_model, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _ffi.Context(0)
    yield c
    c.close()


def _close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-12)
    assert np.abs(a - b).max() <= tol * scale, (np.abs(a - b).max(), scale)


CASES = [
    (300, 20, 1, o.KERNEL_SE, o.LIK_GAUSSIAN, 0),
    (777, 200, 3, o.KERNEL_MATERN32, o.LIK_GAUSSIAN, 0),
    (1500, 256, 8, o.KERNEL_SE, o.LIK_GAUSSIAN, 0),
    (900, 300, 16, o.KERNEL_MATERN52, o.LIK_BERNOULLI_LOGISTIC, 0),
    (640, 129, 2, o.KERNEL_SE, o.LIK_POISSON_EXP, 0),
    (513, 64, 5, o.KERNEL_MATERN52, o.LIK_GAUSSIAN, 9),
    (450, 70, 3, o.KERNEL_MATERN32, o.LIK_EXPONENTIAL_EXP, 0),
    (500, 60, 2, o.KERNEL_SE, o.LIK_GAMMA_EXP, 0),
    (400, 33, 2, o.KERNEL_MATERN52, o.LIK_GAMMA_EXP, 8),
    (520, 90, 3, o.KERNEL_SE, o.LIK_BERNOULLI_NORMCDF, 0),
]


@pytest.mark.parametrize("N,M,d,family,lik,qn", CASES)
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 2e-3)])
def test_gradient_matches_oracle(ctx, N, M, d, family, lik, qn, dtype, tol):
    x, y, sva, s2 = o.synth_problem(300 + N, N, M, d, family=family, lik=lik, dtype=dtype)
    sva.mean_const = 0.1
    val_ref, g_ref = o.elbo_grad(sva, x, y, lik=lik, sigma2=s2, num_data=2.5 * N, quadrature_n=qn)
    model = device_model(ctx, sva, dtype=dtype, lik=lik, sigma2=s2, quadrature_n=qn)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    val, terms, g = model.elbo_grad(data, 0, N, 2.5 * N)
    assert rel(val, val_ref) < (1e-8 if dtype == np.float64 else 1e-4)
    _close(g["m"], g_ref["m"], tol)
    _close(g["Lq"], g_ref["Lq"], tol)
    _close(g["z"].reshape(g_ref["z"].shape, order="F") if d > 1 else g["z"], g_ref["z"] if d > 1 else g_ref["z"][0], tol)
    _close(g["inv_lengthscale"], g_ref["inv_lengthscale"], tol)
    _close([g["variance"]], [g_ref["variance"]], tol)
    _close([g["mean_const"]], [g_ref["mean_const"]], tol)
    if lik in (o.LIK_GAUSSIAN, o.LIK_GAMMA_EXP):   # the likelihood parameter: sigma^2 / Gamma shape
        _close([g["lik_sigma2"]], [g_ref["lik_sigma2"]], tol)
    # same value as the forward-only entry point (the gradient build takes the variance from its dense product, v - k(x,x) =
    # k_j' (R A)_j, not from sum C^2 - sum A^2: equal up to rounding in the compute dtype)
    assert rel(val, model.elbo(data, 0, N, 2.5 * N)[0]) < (1e-12 if dtype == np.float64 else 1e-5)
    model.free()
    data.free()


def test_gradient_chunked_and_minibatch(ctx):
    """A batch larger than one workspace chunk (M = 512 -> chunks of 65536 columns) and a minibatch window."""
    N, M, d = 70_000, 512, 4
    x, y, sva, s2 = o.synth_problem(44, N, M, d)
    model = device_model(ctx, sva, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, np.float64)
    val, _, g = model.elbo_grad(data, 0, N, float(N))
    val_ref, g_ref = o.elbo_grad(sva, x, y, sigma2=s2)
    assert rel(val, val_ref) < 1e-8
    for k in ("m", "Lq", "inv_lengthscale"):
        _close(g[k], g_ref[k], 1e-6)
    _close(g["z"], g_ref["z"], 1e-6)
    off, nb = 12_345, 3000
    val, _, g = model.elbo_grad(data, off, nb, float(N))
    val_ref, g_ref = o.elbo_grad(sva, x[:, off:off + nb], y[off:off + nb], sigma2=s2, num_data=N)
    assert rel(val, val_ref) < 1e-8
    _close(g["z"], g_ref["z"], 1e-6)
    _close(g["Lq"], g_ref["Lq"], 1e-6)
    model.free()
    data.free()


CENTERED_CASES = [
    (300, 20, 1, o.KERNEL_SE, o.LIK_GAUSSIAN, 0),
    (777, 200, 3, o.KERNEL_MATERN32, o.LIK_GAUSSIAN, 0),
    (900, 300, 16, o.KERNEL_MATERN52, o.LIK_BERNOULLI_LOGISTIC, 0),
    (513, 64, 5, o.KERNEL_SE, o.LIK_GAUSSIAN, 7),
    (640, 129, 2, o.KERNEL_SE, o.LIK_POISSON_EXP, 0),
]


@pytest.mark.parametrize("N,M,d,family,lik,qn", CENTERED_CASES)
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 5e-3)])
def test_centered_gradient_matches_oracle(ctx, N, M, d, family, lik, qn, dtype, tol):
    """Centered parametrisation (q(u) = N(m, Lq Lq')): the device whitens, runs the NonCentered adjoint and chains back
    through Lk (src/SparseVariationalApproximationModule.jl:115-136 under Zygote)."""
    x, y, nc, s2 = o.synth_problem(500 + N, N, M, d, family=family, lik=lik, dtype=dtype)
    jit = 1e-4 if dtype == np.float64 else 1e-2   # Lk \\ Lq amplifies rounding by cond(Lk): keep fp32 inside 1e-4
    tame = 0.1 if lik == o.LIK_POISSON_EXP else 1.0   # keep exp(mu + v/2) in a range fp32 can resolve to 1e-4
    sva = o.SVA(nc.kernel, nc.z, tame * (nc.m + 0.3), 0.7 * tame * nc.Lq, jitter=jit, mean_const=0.15, centered=True)
    val_ref, g_ref = o.elbo_grad(sva, x, y, lik=lik, sigma2=s2, num_data=1.5 * N, quadrature_n=qn)
    model = device_model(ctx, sva, dtype=dtype, lik=lik, sigma2=s2, quadrature_n=qn)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    val, terms, g = model.elbo_grad(data, 0, N, 1.5 * N)
    assert rel(val, val_ref) < (1e-8 if dtype == np.float64 else 1e-4)
    _close(g["m"], g_ref["m"], tol)
    _close(g["Lq"], g_ref["Lq"], tol)
    _close(g["z"].reshape(g_ref["z"].shape, order="F") if d > 1 else g["z"], g_ref["z"] if d > 1 else g_ref["z"][0], tol)
    _close(g["inv_lengthscale"], g_ref["inv_lengthscale"], tol)
    _close([g["variance"]], [g_ref["variance"]], tol)
    _close([g["mean_const"]], [g_ref["mean_const"]], tol)
    if lik == o.LIK_GAUSSIAN:
        _close([g["lik_sigma2"]], [g_ref["lik_sigma2"]], tol)
    assert rel(val, model.elbo(data, 0, N, 1.5 * N)[0]) < (1e-12 if dtype == np.float64 else 1e-5)
    model.free()
    data.free()


def test_optimised_posterior_matches_exact_gpr(ctx):
    """The reference's only end-to-end test (test/SparseVariationalApproximationModule.jl:136-186): N = M = 20,
    z = x, NonCentered, jitter 1e-5; optimise (m, A) on -elbo and compare the posterior with exact GP regression
    (atol 1e-4).  The reference runs 20 000 Adam steps through Zygote; here L-BFGS drives svgp_elbo_grad."""
    from scipy.optimize import minimize

    rng = np.random.default_rng(654321)
    N = 20
    x = rng.random(N) * 10
    y = np.sin(x) + 0.9 * np.cos(x * 1.6) + 0.4 * rng.random(N)
    kernel = o.make_kernel([0.2, 0.6])
    s2, jitter = 0.1, 1e-5
    tril = np.tril_indices(N)
    data = _ffi.DeviceData(ctx, x, y, np.float64)
    sva0 = o.SVA(kernel, x.copy(), np.zeros(N), np.eye(N), jitter=jitter)
    model = device_model(ctx, sva0, sigma2=s2)

    def unpack(theta):
        m = theta[:N]
        L = np.zeros((N, N))
        L[tril] = theta[N:]
        return m, L

    def fun(theta):
        m, L = unpack(theta)
        d = np.diag(L).copy()
        L[np.diag_indices(N)] = np.exp(d)          # positive diagonal through an exp reparametrisation
        sva = o.SVA(kernel, x.copy(), m, L, jitter=jitter)
        from helpers import desc_from_oracle
        desc, keep = desc_from_oracle(sva, sigma2=s2)
        model.update(desc, keep)
        val, _, g = model.elbo_grad(data, 0, N, float(N))
        gL = np.asarray(g["Lq"]).copy()
        gL[np.diag_indices(N)] *= np.exp(d)
        return -val, -np.concatenate([g["m"], gL[tril]])

    theta0 = np.concatenate([np.zeros(N), np.zeros(len(tril[0]))])
    res = minimize(fun, theta0, jac=True, method="L-BFGS-B", options=dict(maxiter=3000, maxfun=6000, ftol=1e-15, gtol=1e-9))
    m, L = unpack(res.x)
    L[np.diag_indices(N)] = np.exp(np.diag(L))
    sva = o.SVA(kernel, x.copy(), m, L, jitter=jitter)
    from helpers import desc_from_oracle
    desc, keep = desc_from_oracle(sva, sigma2=s2)
    model.update(desc, keep)
    mean, var, cov = model.predict(x, True, True, True)
    mu_ref, cov_ref = o.exact_gp_posterior(kernel, x, s2, y, x)
    np.testing.assert_allclose(mean, mu_ref, atol=1e-4)   # ref :184
    np.testing.assert_allclose(cov, cov_ref, atol=1e-4)   # ref :185
    assert -res.fun <= o.exact_gp_logpdf(kernel, x, s2, y) + 1e-6
    model.free()
    data.free()


@pytest.mark.parametrize("centered", [False, True])
def test_shard_gradients_sum_to_the_global_gradient(ctx, centered):
    """svgp_elbo_grad_shard (scale = num_data / n_global, kl_weight = 1 / world) on two shards of the batch: the plain
    sum over the 'ranks' is the full-batch value and gradient, for both parametrisations (what the RCCL all-reduce of
    approxgp/distributed.py adds up)."""
    N, M, d, world = 1001, 150, 3, 2
    x, y, nc, s2 = o.synth_problem(61, N, M, d, family=o.KERNEL_MATERN32)
    sva = o.SVA(nc.kernel, nc.z, nc.m + 0.3, 0.7 * nc.Lq, jitter=1e-4, mean_const=0.15, centered=True) if centered else nc
    model = device_model(ctx, sva, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, np.float64)
    full_val, _, full = model.elbo_grad(data, 0, N, 7.0 * N)
    from approxgp.distributed import shard_range
    tot_val, tot = 0.0, None
    for r in range(world):
        lo, hi = shard_range(N, r, world)
        v, t, g = model.elbo_grad(data, lo, hi - lo, shard=(7.0, 1.0 / world))
        assert t.scale == 7.0
        tot_val += v
        tot = g if tot is None else {k: np.asarray(tot[k]) + np.asarray(g[k]) for k in g}
    assert rel(tot_val, full_val) < 1e-12
    for k in full:
        _close(tot[k], full[k], 1e-10)
    # and against the oracle's shard form
    lo, hi = shard_range(N, 1, world)
    v, _, g = model.elbo_grad(data, lo, hi - lo, shard=(7.0, 0.5))
    v_ref, g_ref = o.elbo_grad(sva, x[:, lo:hi], y[lo:hi], sigma2=s2, num_data=7.0 * (hi - lo), kl_weight=0.5)
    assert rel(v, v_ref) < 1e-8
    for k in ("m", "Lq", "inv_lengthscale"):
        _close(g[k], g_ref[k], 1e-6)
    model.free()
    data.free()


def test_randomized_gradient_sweep(ctx):
    """Seeded sweep of svgp_elbo_grad over ragged shapes, families, likelihoods and both parametrisations (fp64)."""
    rng = np.random.default_rng(77)
    liks = [o.LIK_GAUSSIAN, o.LIK_BERNOULLI_LOGISTIC, o.LIK_POISSON_EXP, o.LIK_EXPONENTIAL_EXP, o.LIK_GAMMA_EXP, o.LIK_BERNOULLI_NORMCDF]
    for case in range(14):
        N = int(rng.integers(2, 900))
        M = int(rng.choice([1, 5, 16, 127, 129, 190]))
        d = int(rng.choice([1, 3, 8, 9, 16, 19]))
        fam = int(rng.integers(0, 3))
        lik = liks[case % len(liks)]
        centered = bool(case % 3 == 1)
        x, y, nc, s2 = o.synth_problem(5000 + case, N, M, d, family=fam, lik=lik)
        tame = 0.1 if lik in (o.LIK_POISSON_EXP, o.LIK_EXPONENTIAL_EXP, o.LIK_GAMMA_EXP) else 1.0
        sva = o.SVA(nc.kernel, nc.z, tame * nc.m, nc.Lq if not centered else 0.7 * nc.Lq, jitter=1e-4, mean_const=0.05,
                    centered=centered)
        qn = int(rng.choice([0, 0, 7]))
        val_ref, g_ref = o.elbo_grad(sva, x, y, lik=lik, sigma2=s2, num_data=3.0 * N, quadrature_n=qn)
        model = device_model(ctx, sva, lik=lik, sigma2=s2, quadrature_n=qn)
        data = _ffi.DeviceData(ctx, x, y, np.float64)
        val, _, g = model.elbo_grad(data, 0, N, 3.0 * N)
        info = (case, N, M, d, fam, lik, centered, qn)
        assert rel(val, val_ref) < 1e-8, info
        for k in ("m", "Lq", "inv_lengthscale"):
            a, b = np.asarray(g[k], dtype=np.float64), np.asarray(g_ref[k])
            assert np.abs(a - b).max() <= 1e-6 * max(np.abs(b).max(), 1e-12), (k,) + info
        zb = np.asarray(g["z"], dtype=np.float64).reshape(g_ref["z"].shape, order="F") if d > 1 else np.asarray(g["z"])[None, :]
        assert np.abs(zb - g_ref["z"]).max() <= 1e-6 * max(np.abs(g_ref["z"]).max(), 1e-12), ("z",) + info
        for k in ("variance", "mean_const"):
            assert abs(g[k] - g_ref[k]) <= 1e-6 * max(abs(g_ref[k]), 1e-9) + 1e-9, (k,) + info
        model.free()
        data.free()


@pytest.mark.parametrize("N", [3000, 40_000])
def test_gradient_large_m_float32_strips(ctx, N):
    """Mp > 2048 in fp32 selects the 64-point strips (C4's regime): N = 3000 runs entirely as 32-point half-width strips
    (forward AND value-and-gradient kernels: the fp32 32-point instantiation), N = 40 000 as 64-point strips with a short
    last chunk; value and gradient against the oracle at the full M."""
    M, d = 2100, 4
    # jitter 0.05 keeps cond(Kuu) small enough for fp32 kernel-parameter gradients (the Cholesky backward squares it)
    x, y, sva, s2 = o.synth_problem(88, N, M, d, dtype=np.float32, jitter=0.05)
    model = device_model(ctx, sva, dtype=np.float32, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, np.float32)
    val_ref, g_ref = o.elbo_grad(sva, x, y, sigma2=s2, num_data=2.0 * N)
    val, _, g = model.elbo_grad(data, 0, N, 2.0 * N)
    assert rel(val, val_ref) < 1e-4
    assert rel(model.elbo(data, 0, N, 2.0 * N)[0], val_ref) < 1e-4
    for k in ("m", "Lq", "inv_lengthscale"):
        _close(g[k], g_ref[k], 3e-3)
    _close(g["z"].reshape(g_ref["z"].shape, order="F"), g_ref["z"], 3e-3)
    _close([g["variance"]], [g_ref["variance"]], 3e-3)
    model.free()
    data.free()


def test_gradient_more_than_1024_inducing_points_f64(ctx):
    """M > 1024 in fp64: every block of the Kuu part of the kernel-parameter / inducing-input gradients (the slices of the
    uu reduction have to cover all M columns)."""
    N, M, d = 2500, 1300, 3
    x, y, sva, s2 = o.synth_problem(89, N, M, d, family=o.KERNEL_MATERN52)
    model = device_model(ctx, sva, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, np.float64)
    val_ref, g_ref = o.elbo_grad(sva, x, y, sigma2=s2, num_data=3.0 * N)
    val, _, g = model.elbo_grad(data, 0, N, 3.0 * N)
    assert rel(val, val_ref) < 1e-8
    for k in ("m", "Lq", "inv_lengthscale"):
        _close(g[k], g_ref[k], 1e-6)
    _close(g["z"].reshape(g_ref["z"].shape, order="F"), g_ref["z"], 1e-6)
    _close([g["variance"]], [g_ref["variance"]], 1e-6)
    model.free()
    data.free()


# ---- likelihoods the ABI does not enumerate: host-evaluated on the device marginals (svgp_marginals / svgp_elbo_grad_ext) ----
# fp32: the built-in gradient build takes v from k_j'(R A)_j, svgp_marginals from sum C^2 - sum A^2; and either fp32 gradient is itself 1e-4 ... 3e-4
# of the block's scale away from the fp64 evaluation of the same fp32 inputs (profiles/round6/kgrad_ab.log: both the VALU and the MFMA form of the
# kernel-gradient reductions), so two fp32 evaluations whose point gradients differ in the last bits are compared at 2e-4 (5e-5 until round 6: passed
# by the round-5 reductions with 3.9e-5, not by the MFMA form with 6.2e-5 on the Centered case)
@pytest.mark.parametrize("dtype,vtol,gtol", [(np.float64, 1e-12, 1e-10), (np.float32, 1e-5, 2e-4)])
@pytest.mark.parametrize("centered", [False, True])
def test_host_evaluated_likelihood_equals_builtin(ctx, dtype, vtol, gtol, centered):
    """The split path must reproduce the fused one when the host evaluates a likelihood the library also has: marginals ->
    expected log-likelihood and its point gradients on the host (here the oracle's GH-20 Bernoulli, standing in for the
    reference's GPLikelihoods call) -> svgp_elbo_grad_ext, against svgp_elbo / svgp_elbo_grad with SVGP_LIK_BERNOULLI_LOGISTIC.
    Two strip-kernel chunks (N > 65536 would be slow on the host side: the chunking is forced by SVGP_GRAD_CHUNK in the soak)."""
    N, M, d, lik = 3000, 150, 4, o.LIK_BERNOULLI_LOGISTIC
    x, y, nc, s2 = o.synth_problem(8100, N, M, d, family=o.KERNEL_MATERN52, lik=lik, dtype=dtype)
    sva = o.SVA(nc.kernel, nc.z, nc.m + 0.1, 0.8 * nc.Lq, jitter=nc.jitter, mean_const=0.05, centered=True) if centered else nc
    model = device_model(ctx, sva, dtype=dtype, lik=lik, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    nodata = _ffi.DeviceData(ctx, x, None, dtype)            # the split path needs no observations on the device
    off, n, num_data = 137, 2500, 9000.0
    mu, var = model.marginals(nodata, off, n)
    ref = o.elbo_terms(sva, x[:, off:off + n], y[off:off + n], lik=lik, sigma2=s2, num_data=num_data)
    np.testing.assert_allclose(mu, ref.mu, rtol=0, atol=(1e-10 if dtype == np.float64 else 2e-3))
    np.testing.assert_allclose(var, ref.v + 1e-18, rtol=0, atol=(1e-10 if dtype == np.float64 else 2e-3))
    yb = y[off:off + n]
    sum_e = o.expected_loglik(lik, mu, np.sqrt(var), yb, s2)
    gmu, gv, _ = o.expected_loglik_grads(lik, mu, var, yb, s2)
    val_b, _, g_b = model.elbo_grad(data, off, n, num_data)
    val_e, t_e, g_e = model.elbo_grad(nodata, off, n, num_data, ext=(sum_e, gmu, gv))
    assert abs(val_e - val_b) <= vtol * abs(val_b)
    assert t_e.n_points == n and t_e.scale == pytest.approx(num_data / n)
    for k in ("variance", "mean_const"):
        assert abs(g_e[k] - g_b[k]) <= gtol * max(abs(g_b[k]), 1e-6), k
    assert g_e["lik_sigma2"] == 0.0
    for k in ("inv_lengthscale", "z", "m", "Lq"):
        _close(g_e[k], g_b[k], gtol)
    with pytest.raises(ValueError):
        model.elbo_grad(nodata, off, n, num_data, ext=(sum_e, gmu[:-1], gv[:-1]))
    for h in (model, data, nodata):
        h.free()


def test_host_evaluated_custom_likelihood_against_the_oracle(ctx):
    """A likelihood outside the enumeration (Poisson with a softplus link, GH-20 on the host): the split path against the
    oracle's analytic backward pass fed with the same point gradients."""
    N, M, d = 900, 70, 3
    x, _, sva, _ = o.synth_problem(8200, N, M, d, family=o.KERNEL_SE)
    rng = np.random.default_rng(5)
    y = rng.poisson(np.log1p(np.exp(np.sin(x.sum(axis=0))))).astype(np.float64)
    model = device_model(ctx, sva)
    data = _ffi.DeviceData(ctx, x, None, np.float64)
    mu, var = model.marginals(data)
    xs, ws = o.gausshermite(20)
    ws = ws / np.sqrt(np.pi)
    sd = np.sqrt(var)
    f = mu[None, :] + np.sqrt(2.0) * sd[None, :] * xs[:, None]
    lam = np.logaddexp(0.0, f)                                   # softplus link
    logp = y * np.log(lam) - lam
    dlogp = (y / lam - 1.0) / (1.0 + np.exp(-f))
    sum_e = float((ws[:, None] * logp).sum())
    gmu = (ws[:, None] * dlogp).sum(axis=0)
    gv = (ws[:, None] * dlogp * xs[:, None]).sum(axis=0) / (np.sqrt(2.0) * sd)
    val, _, g = model.elbo_grad(data, 0, N, 4.0 * N, ext=(sum_e, gmu, gv))
    val_ref, g_ref = o.elbo_grad_from_point_grads(sva, x, sum_e, gmu, gv, num_data=4.0 * N)
    assert rel(val, val_ref) < 1e-10
    for k in ("m", "Lq", "inv_lengthscale"):
        _close(g[k], g_ref[k], 1e-6)
    _close(np.asarray(g["z"]).reshape(g_ref["z"].shape, order="F"), g_ref["z"], 1e-6)
    assert abs(g["variance"] - g_ref["variance"]) <= 1e-6 * abs(g_ref["variance"])
    model.free()
    data.free()


def test_host_evaluated_likelihood_across_gradient_chunks(ctx):
    """More points than one gradient chunk (65 536 columns): the point gradients of the second chunk are read at its offset."""
    N, M, d = 70_001, 40, 2
    x, y, sva, s2 = o.synth_problem(8300, N, M, d)
    model = device_model(ctx, sva, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, np.float64)
    mu, var = model.marginals(data)
    gmu, gv = (y - mu) / s2, np.full(N, -0.5 / s2)
    sum_e = float(np.sum(-0.5 * (np.log(2 * np.pi * s2) + ((y - mu) ** 2 + var) / s2)))
    vb, _, gb = model.elbo_grad(data, 0, N, 2.0 * N)
    ve, _, ge = model.elbo_grad(data, 0, N, 2.0 * N, ext=(sum_e, gmu, gv))
    assert abs(ve - vb) <= 1e-12 * abs(vb)
    for k in ("inv_lengthscale", "z", "m", "Lq"):
        _close(ge[k], gb[k], 1e-10)
    model.free()
    data.free()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gradient_layouts_rowvecs_and_vector(ctx, dtype):
    """The inducing-input gradient comes back in the layout z was given in (ColVecs d x M, RowVecs M x d, Vector M): a RowVecs
    model on RowVecs data returns the transpose of the ColVecs gradient, bit for bit; d = 1 as a plain vector likewise."""
    x, y, sva, s2 = o.synth_problem(8400, 500, 45, 3, family=o.KERNEL_MATERN52, dtype=dtype)
    model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
    a = _ffi.DeviceData(ctx, x, y, dtype, _ffi.COLVECS)
    v0, _, g0 = model.elbo_grad(a, 0, 500, 1500.0)
    desc, keep = _ffi.make_desc(dtype, sva.kernel.family, sva.kernel.variance, sva.kernel.inv_lengthscale, np.ascontiguousarray(sva.z.T),
                                sva.m, sva.Lq, sva.jitter, lik_sigma2=s2, layout_z=_ffi.ROWVECS)
    m2 = _ffi.DeviceModel(ctx, desc, keep)
    b = _ffi.DeviceData(ctx, np.ascontiguousarray(x.T), y, dtype, _ffi.ROWVECS)
    v1, _, g1 = m2.elbo_grad(b, 0, 500, 1500.0, z_shape=(45, 3))
    assert v1 == v0
    assert np.array_equal(np.asarray(g1["z"]), np.asarray(g0["z"]).T)
    for k in ("m", "Lq", "inv_lengthscale"):
        assert np.array_equal(np.asarray(g1[k]), np.asarray(g0[k])), k
    # d = 1: Vector layout for z and x against the 1 x M / 1 x N ColVecs form
    x1, y1, s1, s21 = o.synth_problem(8401, 400, 30, 1, dtype=dtype)
    mc = device_model(ctx, s1, dtype=dtype, sigma2=s21)
    dc = _ffi.DeviceData(ctx, x1, y1, dtype, _ffi.COLVECS)
    vc, _, gc = mc.elbo_grad(dc, 0, 400, 400.0)
    desc, keep = _ffi.make_desc(dtype, s1.kernel.family, s1.kernel.variance, s1.kernel.inv_lengthscale, s1.z[0], s1.m, s1.Lq, s1.jitter, lik_sigma2=s21)
    mv = _ffi.DeviceModel(ctx, desc, keep)
    dv = _ffi.DeviceData(ctx, x1[0], y1, dtype)
    vv, _, gv = mv.elbo_grad(dv, 0, 400, 400.0)
    assert vv == vc and np.array_equal(np.asarray(gv["z"]).ravel(), np.asarray(gc["z"]).ravel())
    for h in (model, m2, a, b, mc, dc, mv, dv):
        h.free()


@pytest.mark.parametrize("N,M,d", [(1, 1, 1), (1, 5, 2), (3, 130, 1), (2, 257, 4)])
def test_gradient_degenerate_shapes(ctx, N, M, d):
    """One point, one inducing point, more inducing points than data, M just over a panel boundary: value and every gradient
    block against the oracle (fp64), plus the fp32 run within its tolerance."""
    for dtype, vt, gt in ((np.float64, 1e-9, 1e-6), (np.float32, 1e-4, 5e-3)):
        x, y, sva, s2 = o.synth_problem(8500 + M, N, M, d, dtype=dtype)   # per dtype: the fp32 recipe carries its own jitter (1e-3)
        val_ref, g_ref = o.elbo_grad(sva, x, y, sigma2=s2, num_data=10.0)
        model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
        data = _ffi.DeviceData(ctx, x, y, dtype)
        val, t, g = model.elbo_grad(data, 0, N, 10.0)
        assert rel(val, val_ref) < vt and t.n_points == N
        for k in ("m", "Lq", "inv_lengthscale"):
            _close(g[k], g_ref[k], gt)
        _close(np.asarray(g["z"]).reshape(g_ref["z"].shape, order="F") if d > 1 else np.asarray(g["z"])[None, :], g_ref["z"], gt)
        assert abs(g["variance"] - g_ref["variance"]) <= gt * max(abs(g_ref["variance"]), 1e-6)
        model.free()
        data.free()


def test_gradient_workspace_reused_across_models_of_different_m(ctx):
    """the gradient workspace is cached by (dtype, Mp, d); a model with the same Mp but a larger M (40 then 120, both
    Mp = 128) reused buffers sized for the smaller one.  A fresh context evaluates the small model FIRST, then the large one -
    its gradient must equal the oracle's and the small model's must be unchanged when re-evaluated afterwards."""
    c = _ffi.Context(0)
    try:
        out = {}
        for M in (40, 120, 40):
            x, y, sva, s2 = o.synth_problem(900 + M, 700, M, 3, family=o.KERNEL_MATERN52)
            val_ref, g_ref = o.elbo_grad(sva, x, y, sigma2=s2, num_data=2000.0)
            model = device_model(c, sva, sigma2=s2)
            data = _ffi.DeviceData(c, x, y, np.float64)
            val, _, g = model.elbo_grad(data, 0, 700, 2000.0)
            assert rel(val, val_ref) < 1e-8
            for k in ("m", "Lq", "inv_lengthscale"):
                _close(g[k], g_ref[k], 1e-6)
            _close(g["z"].reshape(g_ref["z"].shape, order="F"), g_ref["z"], 1e-6)
            if M in out:   # the second visit of M = 40 reproduces the first bit for bit
                assert val == out[M][0] and np.array_equal(np.asarray(g["Lq"]), out[M][1])
            out[M] = (val, np.asarray(g["Lq"]).copy())
            model.free()
            data.free()
    finally:
        c.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gradient_does_not_depend_on_the_workspace_history(dtype):
    """The chunking of a value-and-gradient evaluation and the SYRK's split-K slice count are functions of the call (batch length,
    M, dtype), not of what the context's cached workspace was first sized for: a 1 300-point batch evaluated after a 9 000-point
    one (workspace sized for the long batch) returns bit for bit what a fresh context returns."""
    x, y, sva, s2 = o.synth_problem(515, 9000, 150, 4, dtype=dtype, family=o.KERNEL_SE)
    res = []
    for first in (9000, None):
        c = _ffi.Context(0)
        try:
            model = device_model(c, sva, dtype=dtype, sigma2=s2)
            data = _ffi.DeviceData(c, x, y, dtype)
            if first:
                model.elbo_grad(data, 0, first, 9000.0)
            val, _, g = model.elbo_grad(data, 2000, 1300, 9000.0)
            res.append((val, {k: np.asarray(v).copy() for k, v in g.items() if isinstance(v, np.ndarray) or np.ndim(v)}))
            model.free()
            data.free()
        finally:
            c.close()
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        assert np.array_equal(res[0][1][k], res[1][1][k]), k
