"""Shared helpers for the parity tests: turn an oracle SVA (oracle/svgp_oracle.py) into the C-ABI objects."""
import contextlib
import os

import numpy as np

import svgp_oracle as o
from approxgp import _ffi


def desc_from_oracle(sva: o.SVA, dtype=np.float64, lik=o.LIK_GAUSSIAN, sigma2=1.0, quadrature_n=0,
                     neg_var_policy=_ffi.NEGVAR_ERROR):
    return _ffi.make_desc(
        dtype, sva.kernel.family, sva.kernel.variance, sva.kernel.inv_lengthscale, sva.z, sva.m, sva.Lq, sva.jitter,
        parametrization=_ffi.CENTERED if sva.centered else _ffi.NONCENTERED, likelihood=lik, lik_sigma2=sigma2,
        quadrature_n=quadrature_n, mean_const=sva.mean_const, neg_var_policy=neg_var_policy)


def device_model(ctx, sva, **kw):
    desc, keep = desc_from_oracle(sva, **kw)
    return _ffi.DeviceModel(ctx, desc, keep)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


@contextlib.contextmanager
def context_with_env(**env):
    """A fresh context created under the given environment: the library reads its settings ONCE, at svgp_ctx_create
    (csrc/knobs.hpp) - SVGP_TIMING, SVGP_OVERLAP, SVGP_SEG_SPLIT in every build, the tuning / A-B knobs in the experiments build."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        c = _ffi.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        yield c
    finally:
        c.close()


def ambient_on(name: str) -> bool:
    """The ambient value of an operational 0 / 1 setting (default 1): what a context created WITHOUT context_with_env carries.  The
    suite also runs under SVGP_OVERLAP=0, SVGP_SEG_SPLIT=0, SVGP_TIMING=0 (profiles/round6/gputest_settings.log); assertions about the
    timing record or about the overlapped path being taken on such a context ask this first."""
    return os.environ.get(name, "1") != "0"


def experiments_build() -> bool:
    """True when the loaded library is the experiments build (tools/build_experiments.sh, SVGP_MI355X_LIB=...): it alone exports
    svgp_debug_experiments and honours the tuning / A-B environment knobs."""
    return hasattr(_ffi.load_library(), "svgp_debug_experiments")


class GaussHermiteLikelihood:
    """Test-side CALLER likelihood for the host-evaluated route (approxgp.CallerLikelihood): log p(y | f) and its f-derivative
    as vectorised callables, expectation by Gauss-Hermite in numpy - the stand-in for the reference's own
    GPLikelihoods.expected_loglikelihood that the Julia binding calls at this point."""

    def __new__(cls, logp, dlogp=None):
        import approxgp as ag

        class _L(ag.CallerLikelihood):
            def expectation(self, mu, var, y, n_points, want_grad):
                xs, ws = np.polynomial.hermite.hermgauss(int(n_points))
                ws = ws / np.sqrt(np.pi)
                sd = np.sqrt(var)
                f = mu[None, :] + np.sqrt(2.0) * sd[None, :] * xs[:, None]
                sum_e = float((ws[:, None] * logp(f, y[None, :])).sum())
                if not want_grad:
                    return sum_e, None, None
                dl = dlogp(f, y[None, :])
                return sum_e, (ws[:, None] * dl).sum(axis=0), (ws[:, None] * dl * xs[:, None]).sum(axis=0) / (np.sqrt(2.0) * sd)

        return _L()
