"""approxgp — Python host mirror of ApproximateGPs.jl's sparse-variational API over the MI355X HIP library.

    from approxgp import *
    f = GP(1.3 * with_lengthscale(SqExponentialKernel(), 0.3))
    sva = SparseVariationalApproximation(f(z, 1e-5), MvNormal.from_cholesky(m, A))
    elbo(sva, f(x, 0.3), y, num_data=N); posterior(sva).mean_and_var(xs)
"""
from ._ffi import (Context, DeclinedError, DeviceData, DeviceModel, DomainError, PosDefException, SvgpError, UnsupportedError,
                   default_context, gausshermite, load_library, offload_advice, offload_work)
from .gp import (GP, BernoulliLikelihood, DefaultExpectationMethod, FiniteGP, GaussHermiteExpectation,
                 GaussianLikelihood, LatentFiniteGP, LatentGP, MvNormal, PoissonLikelihood, ExponentialLikelihood,
                 GammaLikelihood, CallerLikelihood, LogisticLink, NormalCDFLink, ProbitLink)
from .kernels import (ARDTransform, Matern32Kernel, Matern52Kernel, ScaledKernel, ScaleTransform, SEKernel,
                      SqExponentialKernel, TransformedKernel, with_lengthscale)
from .sva import (SVGP, ApproxPosteriorGP, Centered, NonCentered, SparseVariationalApproximation, approx_lml, elbo,
                  elbo_and_gradient,
                  inducing_points, posterior, prior_kl)

__all__ = [n for n in dir() if not n.startswith("_")]
