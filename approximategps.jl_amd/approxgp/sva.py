"""Python host mirror of ApproximateGPs.jl's SparseVariationalApproximation API — same names, argument
meaning and error behaviour as /root/reference/src/SparseVariationalApproximationModule.jl (SVA) —
over the C-ABI of libsvgp_mi355x.so.  This file only packs parameters and maps status codes to
exceptions; the numbers come from the HIP library (there is no CPU path)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _ffi
from .gp import (DEFAULT_SIGMA2, BernoulliLikelihood, DefaultExpectationMethod, FiniteGP, GaussHermiteExpectation,
                 GaussianLikelihood, CallerLikelihood, LatentFiniteGP, MvNormal, NormalCDFLink, LogisticLink, PoissonLikelihood, ExponentialLikelihood,
                 GammaLikelihood, _as_dn)
from .kernels import unpack_kernel


class Centered:  # SVA:41
    pass


class NonCentered:  # SVA:57
    pass


@dataclass(eq=False)
class SparseVariationalApproximation:
    """SparseVariationalApproximation([Centered()|NonCentered(),] fz::FiniteGP, q::MvNormal)  (SVA:59-95).
    The two-argument form is NonCentered (SVA:93-95)."""

    parametrization: object
    fz: FiniteGP
    q: MvNormal

    def __init__(self, *args):
        if len(args) == 2:
            self.parametrization, (self.fz, self.q) = NonCentered(), args
        elif len(args) == 3:
            self.parametrization, self.fz, self.q = args
            if isinstance(self.parametrization, type):
                self.parametrization = self.parametrization()
        else:
            raise TypeError("SparseVariationalApproximation([parametrization,] fz, q)")
        if not isinstance(self.fz, FiniteGP) or not isinstance(self.q, MvNormal):
            raise TypeError("fz must be a FiniteGP and q an MvNormal")

    @property
    def is_centered(self):
        return isinstance(self.parametrization, Centered)


def SVGP(*args):
    """src/deprecations.jl:1 — SVGP(args...) = SparseVariationalApproximation(Centered(), args...)."""
    return SparseVariationalApproximation(Centered(), *args)


_LIK = {GaussianLikelihood: _ffi.LIK_GAUSSIAN, BernoulliLikelihood: _ffi.LIK_BERNOULLI_LOGISTIC,
        PoissonLikelihood: _ffi.LIK_POISSON_EXP, ExponentialLikelihood: _ffi.LIK_EXPONENTIAL_EXP,
        GammaLikelihood: _ffi.LIK_GAMMA_EXP}


def _desc(sva: SparseVariationalApproximation, lik=None, quadrature=None, dtype=None, neg_var_policy=_ffi.NEGVAR_ERROR):
    fz = sva.fz
    z = np.asarray(fz.x)
    if not fz.is_isotropic():
        raise _ffi.UnsupportedError("fz.Σy must be isotropic jitter")
    dtype = np.dtype(dtype if dtype is not None else (np.float32 if z.dtype == np.float32 else np.float64))
    d = 1 if z.ndim == 1 else z.shape[0]
    family, variance, il = unpack_kernel(fz.f.kernel, d)
    lik_code, s2 = _ffi.LIK_GAUSSIAN, 1.0
    if lik is not None:
        if type(lik) not in _LIK:
            raise _ffi.UnsupportedError(f"unsupported likelihood {lik!r}")
        lik_code = _LIK[type(lik)]
        if isinstance(lik, BernoulliLikelihood):
            if isinstance(lik.invlink, NormalCDFLink):
                lik_code = _ffi.LIK_BERNOULLI_NORMCDF
            elif not isinstance(lik.invlink, LogisticLink):
                raise _ffi.UnsupportedError(f"unsupported link {lik.invlink!r}")
        if isinstance(lik, GaussianLikelihood):
            s2 = float(lik.sigma2)
        elif isinstance(lik, GammaLikelihood):
            s2 = float(lik.alpha)   # the shape travels in the likelihood-parameter slot of the ABI
    qn = 0
    if isinstance(quadrature, GaussHermiteExpectation):
        qn = int(quadrature.n)
    elif quadrature is not None and not isinstance(quadrature, DefaultExpectationMethod):
        raise _ffi.UnsupportedError(f"unsupported quadrature {quadrature!r}")
    return _ffi.make_desc(dtype, family, variance, il, z, sva.q.m, sva.q.chol_lower, float(fz.Sigma_y),
                          parametrization=_ffi.CENTERED if sva.is_centered else _ffi.NONCENTERED,
                          likelihood=lik_code, lik_sigma2=s2, quadrature_n=qn, mean_const=fz.f.mean_const,
                          neg_var_policy=neg_var_policy)


# ------------------------------------------------------------------------------------------------
# elbo / approx_lml
# ------------------------------------------------------------------------------------------------
def _decline_if_small(sva, lfx, y, small_problems, want_grad, ctx=None):
    """The rule of the Julia hooks (try_elbo / the rrule return `nothing` and the reference's own body runs): below the
    library's offload threshold (svgp_offload_advice; measured crossover, include/svgp_mi355x.h) the device is slower than the
    host.  The mirror has no host path, so it only applies the rule on request (`small_problems="decline"`) and then raises.
    Never under a library communicator: the calls are collective there, and a rule decided on one rank's own shard
    length would let that rank leave while its peers wait in the all-reduce - every rank of a data-parallel job offloads."""
    if small_problems == "run":
        return
    if small_problems != "decline":
        raise ValueError('small_problems must be "run" or "decline"')
    # ctx=None means the default context (elbo / elbo_and_gradient resolve it after this check): a communicator initialised on THAT
    # context counts too (with it skipped, one rank could decline while its peers sat in the all-reduce).  Peeked at, never
    # created here: the rule itself must work without a GPU.
    c = ctx if ctx is not None else _ffi._default_ctx
    if c is not None and c.comm_info()[0] > 1:
        return
    z = np.asarray(sva.fz.x)
    d, M = (1, z.shape[0]) if z.ndim == 1 else z.shape
    n = np.asarray(y).shape[0]
    if not _ffi.offload_advice(n, M, d, want_gradient=want_grad):
        raise _ffi.DeclinedError(f"n = {n}, M = {M}, d = {d}: work {_ffi.offload_work(n, M, d):.3g} is below the offload "
                                 "threshold (SVGP_OFFLOAD_MIN_WORK, default 3e6); the Julia binding runs the reference's own "
                                 "method here")


def elbo(sva: SparseVariationalApproximation, fx, y, *, num_data=None, quadrature=None, ctx=None, dtype=None,
         return_terms=False, small_problems="run"):
    """elbo(sva, fx::FiniteGP | lfx::LatentFiniteGP, y; num_data=length(y), quadrature=DefaultExpectationMethod())

    FiniteGP method (SVA:307-317): Gaussian likelihood with σ² = fx.Σy[1]; non-isotropic noise raises the
    reference's ErrorException text (SVA:319-327).  LatentFiniteGP method: SVA:340-360."""
    if isinstance(fx, FiniteGP):
        if not fx.is_isotropic():
            raise RuntimeError(
                "The observation noise fx.Σy must be homoscedastic.\n"
                "To avoid this error, construct fx using: f = GP(kernel); fx = f(x, σ²), where σ² is a positive Real."
            )
        lfx = LatentFiniteGP(fx, GaussianLikelihood(float(fx.Sigma_y)))
    elif isinstance(fx, LatentFiniteGP):
        lfx = fx
    else:
        raise TypeError("elbo expects a FiniteGP or a LatentFiniteGP")
    if sva.fz.f is not lfx.fx.f:  # SVA:347-351
        raise ValueError("(Latent)FiniteGP prior is not consistent with SparseVariationalApproximation's")
    _decline_if_small(sva, lfx, y, small_problems, False, ctx)
    ctx = ctx or _ffi.default_context()
    if isinstance(lfx.lik, CallerLikelihood):
        return _elbo_host_likelihood(sva, lfx, y, num_data, quadrature, ctx, dtype, False)[0]
    desc, keep = _desc(sva, lfx.lik, quadrature, dtype)
    y = np.asarray(y)
    n = y.shape[0]
    data = _ffi.DeviceData(ctx, lfx.fx.x, y, _ffi.np_dtype(desc.dtype))
    model = _ffi.DeviceModel(ctx, desc, keep)
    try:
        val, terms = model.elbo(data, 0, n, float(num_data) if num_data is not None else 0.0)
    finally:
        model.free()
        data.free()
    return (val, terms) if return_terms else val


def _elbo_host_likelihood(sva, lfx, y, num_data, quadrature, ctx, dtype, want_grad):
    """The route of a likelihood outside the ABI's enumeration: marginals(f_post(x)) (SVA:354) from the device, SVA:355 on the
    host, KL / backward pass on the device.  Value: E num_data / n - KL (SVA:357-359)."""
    if quadrature is not None and not isinstance(quadrature, (GaussHermiteExpectation, DefaultExpectationMethod)):
        raise _ffi.UnsupportedError(f"unsupported quadrature {quadrature!r}")
    qn = int(quadrature.n) if isinstance(quadrature, GaussHermiteExpectation) else 20
    desc, keep = _desc(sva, None, None, dtype)
    y = np.asarray(y, dtype=np.float64)
    n = y.shape[0]
    data = _ffi.DeviceData(ctx, lfx.fx.x, None, _ffi.np_dtype(desc.dtype))
    model = _ffi.DeviceModel(ctx, desc, keep)
    try:
        mu, var = model.marginals(data, 0, n)
        sum_e, gmu, gv = lfx.lik.expectation(mu, var, y, qn, want_grad)   # the caller's code (SVA:355)
        nd = float(num_data) if num_data is not None else float(n)
        if not want_grad:
            return sum_e * nd / n - model.prior_kl()[0], None
        val, _, grads = model.elbo_grad(data, 0, n, nd, z_shape=np.asarray(sva.fz.x).shape, ext=(sum_e, gmu, gv))
        return val, grads
    finally:
        model.free()
        data.free()


def elbo_and_gradient(sva: SparseVariationalApproximation, fx, y, *, num_data=None, quadrature=None, ctx=None, dtype=None,
                      small_problems="run"):
    """ELBO and its gradient w.r.t. (kernel variance, inverse lengthscales, inducing inputs z, mean(q) m, the lower factor
    Lq of cov(q), Gaussian noise σ², ConstMean) — what `Zygote.gradient(-elbo, ...)` yields for the reference's training
    loops (examples/a-regression/script.jl:188-194); the Julia shim wraps it as a ChainRulesCore.rrule."""
    if isinstance(fx, FiniteGP):
        if not fx.is_isotropic():
            raise RuntimeError("The observation noise fx.Σy must be homoscedastic.")
        lfx = LatentFiniteGP(fx, GaussianLikelihood(float(fx.Sigma_y)))
    else:
        lfx = fx
    if sva.fz.f is not lfx.fx.f:
        raise ValueError("(Latent)FiniteGP prior is not consistent with SparseVariationalApproximation's")
    _decline_if_small(sva, lfx, y, small_problems, True, ctx)
    ctx = ctx or _ffi.default_context()
    if isinstance(lfx.lik, CallerLikelihood):
        return _elbo_host_likelihood(sva, lfx, y, num_data, quadrature, ctx, dtype, True)
    desc, keep = _desc(sva, lfx.lik, quadrature, dtype)
    y = np.asarray(y)
    data = _ffi.DeviceData(ctx, lfx.fx.x, y, _ffi.np_dtype(desc.dtype))
    model = _ffi.DeviceModel(ctx, desc, keep)
    try:
        val, _, grads = model.elbo_grad(data, 0, y.shape[0], float(num_data) if num_data is not None else 0.0,
                                        z_shape=np.asarray(sva.fz.x).shape)
    finally:
        model.free()
        data.free()
    return val, grads


def approx_lml(sva, l_fx, ys, **kwargs):
    """API.approx_lml (SVA:276-280): forwards to elbo."""
    return elbo(sva, l_fx, ys, **kwargs)


# ------------------------------------------------------------------------------------------------
# posterior and the prediction API on ApproxPosteriorGP{<:SparseVariationalApproximation}
# ------------------------------------------------------------------------------------------------
class ApproxPosteriorGP:
    """ApproxPosteriorGP(sva, prior, (Kuu = Cholesky(Lk), B, α))  (SVA:134-135, :185-186), with the model
    kept resident on the GPU for the prediction methods."""

    def __init__(self, approx: SparseVariationalApproximation, ctx=None, dtype=None):
        self.approx = approx
        self.prior = approx.fz.f
        self.ctx = ctx or _ffi.default_context()
        desc, keep = _desc(approx, None, None, dtype)
        self._model = _ffi.DeviceModel(self.ctx, desc, keep)
        Lk, alpha, B = self._model.posterior()
        self.data = {"Kuu": Lk, "B": B, "α": alpha, "alpha": alpha}

    def inducing_points(self):  # SVA:270
        return self.approx.fz.x

    def mean(self, x):  # SVA:208-212
        return self._model.predict(x, True, False, False)[0]

    def var(self, x):  # SVA:230-235
        return self._model.predict(x, False, True, False)[1]

    def mean_and_var(self, x):  # SVA:246-253
        m, v, _ = self._model.predict(x, True, True, False)
        return m, v

    def cov(self, x, y=None):  # SVA:223-228 and :255-264
        if y is None:
            return self._model.predict(x, False, False, True)[2]
        return self._model.cross_cov(x, y)

    def mean_and_cov(self, x):  # SVA:237-244
        m, _, c = self._model.predict(x, True, False, True)
        return m, c

    def rand(self, x, n_samples=1, jitter=DEFAULT_SIGMA2, rng=None):
        """rand(f_post(x, jitter), n_samples) (examples/b-classification/script.jl:153) [dep AbstractGPs: mean +
        cholesky(cov + jitter I).L ξ].  The moments come from the device (svgp_predict); the n x n factorisation of
        the test covariance is host-side, as in the reference.  -> (n, n_samples)."""
        m, c = self.mean_and_cov(x)
        m, c = np.asarray(m, dtype=np.float64), np.asarray(c, dtype=np.float64)
        L = np.linalg.cholesky(c + float(jitter) * np.eye(c.shape[0]))
        rng = rng if rng is not None else np.random.default_rng()
        return m[:, None] + L @ rng.standard_normal((c.shape[0], int(n_samples)))

    def marginals(self, x):
        """marginals(f_post(x)) (SVA:354): (μ, σ) of Normal.(μ, sqrt.(v + 1e-18))."""
        m, v = self.mean_and_var(x)
        v = v + 1e-18
        if np.any(v < 0):
            raise _ffi.DomainError("sqrt of a negative variance")
        return m, np.sqrt(v)


def posterior(sva: SparseVariationalApproximation, fx=None, y=None, *, ctx=None, dtype=None):
    """posterior(sva) (SVA:115-136 / :160-187); the 3-argument forms assert the same prior and ignore the
    data (SVA:189-201)."""
    if fx is not None:
        prior = fx.f if isinstance(fx, FiniteGP) else fx.fx.f
        assert sva.fz.f is prior  # SVA:192,199
    return ApproxPosteriorGP(sva, ctx=ctx, dtype=dtype)


def inducing_points(f: ApproxPosteriorGP):
    return f.inducing_points()


def prior_kl(sva: SparseVariationalApproximation, *, ctx=None, dtype=None):
    """_prior_kl(sva) (SVA:362-373)."""
    ctx = ctx or _ffi.default_context()
    desc, keep = _desc(sva, None, None, dtype)
    model = _ffi.DeviceModel(ctx, desc, keep)
    try:
        return model.prior_kl()[0]
    finally:
        model.free()
