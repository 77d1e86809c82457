"""The AbstractGPs / GPLikelihoods / Distributions objects that appear in the reference's SVGP call
signatures (SVA = /root/reference/src/SparseVariationalApproximationModule.jl), as thin parameter
holders for the Python host mirror.  No arithmetic on M×n objects happens here."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

DEFAULT_SIGMA2 = 1e-18  # AbstractGPs.default_σ²


def _as_dn(x):
    x = np.asarray(x)
    return x[None, :] if x.ndim == 1 else x


@dataclass(eq=False)
class GP:
    """GP(kernel) with ZeroMean, or GP(c, kernel) with ConstMean(c).  Identity (`is`) matters:
    elbo checks `sva.fz.f === lfx.fx.f` (SVA:347-351)."""

    kernel: object
    mean_const: float = 0.0

    def __init__(self, *args):
        if len(args) == 1:
            self.kernel, self.mean_const = args[0], 0.0
        elif len(args) == 2:
            self.mean_const, self.kernel = float(args[0]), args[1]
        else:
            raise TypeError("GP(kernel) or GP(mean_const, kernel)")

    def __call__(self, x, Sigma_y=DEFAULT_SIGMA2):
        return FiniteGP(self, np.asarray(x), Sigma_y)


@dataclass(eq=False)
class FiniteGP:
    """f(x, Σy).  Σy: a Real is Diagonal(Fill(σ², n)) (homoscedastic); a vector is heteroscedastic."""

    f: GP
    x: np.ndarray
    Sigma_y: object = DEFAULT_SIGMA2

    @property
    def n(self):
        return _as_dn(self.x).shape[1]

    def is_isotropic(self):
        return np.ndim(self.Sigma_y) == 0


@dataclass(frozen=True)
class GaussianLikelihood:
    sigma2: float = 1e-6


@dataclass(frozen=True)
class LogisticLink:
    pass


@dataclass(frozen=True)
class NormalCDFLink:
    """probit: invlink = normcdf  [GPLikelihoods]"""


ProbitLink = NormalCDFLink


@dataclass(frozen=True)
class BernoulliLikelihood:
    """BernoulliLikelihood(l = logistic): y ~ Bernoulli(l(f)); l is LogisticLink() (default) or NormalCDFLink()"""
    invlink: object = LogisticLink()


@dataclass(frozen=True)
class PoissonLikelihood:
    """exp link"""


@dataclass(frozen=True)
class ExponentialLikelihood:
    """exp link: y ~ Distributions.Exponential(exp f), i.e. SCALE exp f: log p = -f - y exp(-f)  [GPLikelihoods]"""


@dataclass(frozen=True)
class GammaLikelihood:
    """exp link: y ~ Gamma(shape alpha, scale exp f)  [GPLikelihoods]"""
    alpha: float = 1.0


class CallerLikelihood:
    """Base class for a single-latent likelihood the C-ABI does not enumerate (in the reference: any other GPLikelihoods
    likelihood or link).  It is CALLER code, like the likelihood object a Julia user passes to `elbo`: only SVA:355,
    expected_loglikelihood(quadrature, lik, q_f, y), depends on it, and that is O(n) scalar work on the marginals.  The
    library computes marginals(f_post(x)) on the device (svgp_marginals), calls `expectation` below on the host - exactly
    where the Julia binding calls the reference's own GPLikelihoods method - and runs the backward pass on the device again
    with the point gradients it returns (svgp_elbo_grad_ext).  This package contains no implementation of it and no host
    numerics: the enumerated likelihoods never take this route, and there is nothing it falls back from.

    Subclasses implement
        expectation(mu, var, y, n_points, want_grad) -> (sum_i E_i, dE_i/dmu_i or None, dE_i/dv_i or None)
    with E_i = E_{N(mu_i, var_i)}[log p(y_i | f)] and `n_points` the Gauss-Hermite order the caller asked for."""

    def expectation(self, mu, var, y, n_points, want_grad):
        raise NotImplementedError


@dataclass(eq=False)
class LatentGP:
    f: GP
    lik: object
    Sigma_y: float = DEFAULT_SIGMA2

    def __call__(self, x):
        return LatentFiniteGP(self.f(x, self.Sigma_y), self.lik)


@dataclass(eq=False)
class LatentFiniteGP:
    fx: FiniteGP
    lik: object


@dataclass(frozen=True)
class GaussHermiteExpectation:
    n: int = 20


@dataclass(frozen=True)
class DefaultExpectationMethod:
    pass


@dataclass(eq=False)
class MvNormal:
    """MvNormal(m, S).  `chol_lower` is _chol_lower(_chol_cov(q)) (src/utils.jl:15-18): free when built
    from a factor (`MvNormal.from_cholesky`, the PDMat(Cholesky(LowerTriangular(A))) pattern of
    examples/a-regression/script.jl:110-111), one host LAPACK potrf of the user's dense S otherwise —
    parameter packaging, as the Julia shim would do it, not part of the device path."""

    m: np.ndarray
    S: np.ndarray = None
    _L: np.ndarray = field(default=None, repr=False)

    def __init__(self, m, S=None, _L=None):
        self.m = np.asarray(m)
        self.S = None if S is None else np.asarray(S)
        self._L = _L
        if self.S is None and _L is None:
            raise TypeError("MvNormal needs a covariance or a Cholesky factor")

    @classmethod
    def from_cholesky(cls, m, L):
        return cls(m, None, np.tril(np.asarray(L)))

    @property
    def chol_lower(self):
        if self._L is None:
            self._L = np.linalg.cholesky(0.5 * (self.S + self.S.T))
        return self._L

    @property
    def cov(self):
        return self.S if self.S is not None else self._L @ self._L.T
