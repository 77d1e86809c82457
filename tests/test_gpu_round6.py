"""Round-6 GPU tests: a 48-panel fp32 Kuu, and (experiments build) the 256 x 256-tile rank-256 trailing update of the large fp32
Cholesky (prep.hip: syrk256_kernel, two-level schedule of potrf_t: built, measured, not adopted).  The kernel-gradient reductions on the MFMA (grad.hip: kgrad_mfma_kernel) are covered by every gradient test
of tests/test_gpu_grad.py / test_gpu_round4.py / test_gpu_round5.py (d = 1 ... 64, both dtypes, every kernel family), which now run on it."""
import numpy as np
import pytest

import svgp_oracle as o
from approxgp import _ffi
from helpers import device_model, experiments_build, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _ffi.Context(0)
    yield c
    c.close()


def _factor_against_lapack(c, M, N, d, seed):
    x, y, sva, s2 = o.synth_problem(seed, N, M, d, dtype=np.float32)
    model = device_model(c, sva, dtype=np.float32, sigma2=s2)
    data = _ffi.DeviceData(c, x, y, np.float32)
    vals = {model.elbo(data, 0, N, float(N))[0] for _ in range(3)}
    assert len(vals) == 1, vals                      # run-to-run identical bits: no missing dependency between the launches
    Lk, _, _ = model.posterior()
    K = o.kuu(sva)
    L = np.tril(np.asarray(Lk, dtype=np.float64))
    back = np.linalg.norm(L @ L.T - K) / np.linalg.norm(K)
    Lref = np.linalg.cholesky(K)
    err = np.abs(L - Lref).max() / np.abs(Lref).max()
    val = vals.pop()
    ref = o.elbo(sva, x, y, sigma2=s2, num_data=float(N))
    model.free()
    data.free()
    return back, err, rel(val, ref)


def test_large_fp32_kuu_at_48_panels(ctx):
    """M = 6144 fp32 (48 panels): the factor must be LAPACK's to fp32 rounding and the ELBO the oracle's.  (Experiments build with
    SVGP_CHOL_TWO_LEVEL=1: the cost rule of potrf_t then mixes two-level pairs on 256 x 256 tiles with one-level steps.)"""
    back, err, erel = _factor_against_lapack(ctx, 6144, 6200, 3, 8800)
    assert back < 2e-6, back
    assert err < 2e-3, err
    assert erel < 1e-4, erel


def test_rank256_trailing_update_forced_at_every_pair():
    """Experiments build: SVGP_CHOL_T256_US=0.01 makes every pair with an even number of trailing block rows take the 256 x 256 tiles
    (M = 2304: 18 panels, tiles down to a single one; M = 2432: 19 panels - an odd count, the pairs start after a one-level step)."""
    if not experiments_build():
        pytest.skip("the cost-rule knobs exist in the experiments build only (tools/build_experiments.sh)")
    import os
    prev = os.environ.get("SVGP_CHOL_T256_US")
    os.environ["SVGP_CHOL_T256_US"] = "0.01"       # (read at every factorisation in the experiments build)
    prev2 = os.environ.get("SVGP_CHOL_TWO_LEVEL")
    os.environ["SVGP_CHOL_TWO_LEVEL"] = "1"
    try:
        c = _ffi.Context(0)
        for M in (2304, 2432):
            back, err, erel = _factor_against_lapack(c, M, M + 100, 3, 8810 + M)
            assert back < 2e-6 and err < 2e-3 and erel < 1e-4, (M, back, err, erel)
        c.close()
    finally:
        for k, v in (("SVGP_CHOL_T256_US", prev), ("SVGP_CHOL_TWO_LEVEL", prev2)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
