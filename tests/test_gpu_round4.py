"""GPU tests of what round 4 added: inputs of 17..64 dimensions on the MFMA-distance pre-generation (every kernel family, value
and gradient), the post-strip point-gradient kernel against the round-3 in-kernel forms, and the device-side halves of the
ADVICE r3 fixes.  Everything goes through the C-ABI (ctypes)."""
import ctypes as C
import os

import numpy as np
import pytest

import svgp_oracle as o
from approxgp import _ffi
from helpers import ambient_on, GaussHermiteLikelihood, context_with_env, device_model, experiments_build, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _ffi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("d", [17, 24, 32, 40, 64])
@pytest.mark.parametrize("family", [o.KERNEL_SE, o.KERNEL_MATERN32, o.KERNEL_MATERN52])
@pytest.mark.parametrize("dtype,tol,mtol,gtol", [(np.float64, 1e-8, 1e-10, 1e-6), (np.float32, 1e-4, 2e-4, 3e-3)])
def test_wide_inputs_on_the_mfma_pregeneration(ctx, dtype, tol, mtol, gtol, family, d):
    """cov(f.prior, z, x) is dimension-agnostic in the reference (SVA:216); the device's strips used to leave the
    MFMA-distance path at d = 17.  16 < d <= 64 now run the 32- / 64-feature bodies of pregen_mfma (strip.hip): ELBO, the
    per-point marginals (a stricter check than one scalar) and the gradient, M not a multiple of 128, ragged batch."""
    N, M = 1237, 150
    x, y, sva, s2 = o.synth_problem(4100 + d, N, M, d, family=family, dtype=dtype)
    model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    val_ref, g_ref = o.elbo_grad(sva, x, y, sigma2=s2, num_data=3.0 * N)
    assert rel(model.elbo(data, 0, N, 3.0 * N)[0], val_ref) < tol
    mu, var = model.marginals(data, 0, N)
    mu_ref, var_ref = o.mean_and_var(o.posterior(sva), x)
    assert np.abs(mu - mu_ref).max() < mtol * max(1.0, np.abs(mu_ref).max())
    assert np.abs(var - (var_ref + 1e-18)).max() < mtol * sva.kernel.variance
    val, _, g = model.elbo_grad(data, 0, N, 3.0 * N)
    assert rel(val, val_ref) < tol
    for k in ("m", "Lq", "inv_lengthscale"):
        a, b = np.asarray(g[k], dtype=np.float64), np.asarray(g_ref[k])
        assert np.abs(a - b).max() <= gtol * np.abs(b).max(), k
    zb = np.asarray(g["z"], dtype=np.float64).reshape(g_ref["z"].shape, order="F")
    assert np.abs(zb - g_ref["z"]).max() <= gtol * np.abs(g_ref["z"]).max()
    model.free()
    data.free()


# (test_point_gradient_kernel_equals_the_in_kernel_forms - point_grad_kernel against the round-3 in-kernel likelihood gradients, selected
#  by SVGP_GRAD_POST=0 in the experiments build - was retired in round 6 together with those forms: profiles/round6/removed_variants.patch;
#  point_grad_kernel is checked against the oracle for every likelihood by tests/test_gpu_grad.py)


def test_host_evaluated_route_through_the_point_gradient_kernel(ctx):
    """svgp_elbo_grad_ext: the host's (dE/dmu, dE/dv) now enter through point_grad_kernel too (one strip instantiation for both
    likelihood routes): a Gaussian evaluated on the host reproduces the built-in likelihood, across two gradient chunks' worth
    of strips and with negative-variance clamping off."""
    N, M, d = 3001, 96, 3
    x, y, sva, s2 = o.synth_problem(5300, N, M, d)
    model = device_model(ctx, sva, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, np.float64)
    vb, _, gb = model.elbo_grad(data, 0, N, 4.0 * N)
    mu, var = model.marginals(data, 0, N)
    sum_e = float(np.sum(-0.5 * (np.log(2 * np.pi) + np.log(s2) + ((y - mu) ** 2 + var) / s2)))
    gmu, gv = (y - mu) / s2, np.full(N, -0.5 / s2)
    ve, _, ge = model.elbo_grad(data, 0, N, 4.0 * N, ext=(sum_e, gmu, gv))
    assert rel(ve, vb) < 1e-12
    for k in ("z", "m", "Lq", "inv_lengthscale"):
        assert np.abs(np.asarray(ge[k]) - np.asarray(gb[k])).max() <= 1e-9 * np.abs(np.asarray(gb[k])).max(), k
    model.free()
    data.free()


def test_timing_calls_respect_the_callers_buffer(ctx):
    """svgp_last_timing writes the 48-byte v2 / v3 layout and not a byte more; svgp_last_timing_sized writes
    what the caller says it has (and carries ms_chol, the v4 field)."""
    x, y, sva, s2 = o.synth_problem(5400, 600, 130, 2)
    model = device_model(ctx, sva, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, np.float64)
    model.elbo(data, 0, 600, 600.0)
    lib = ctx.lib
    buf = (C.c_ubyte * 80)(*([0xAB] * 80))
    assert lib.svgp_last_timing(ctx.h, C.cast(buf, C.POINTER(_ffi.Timing))) == _ffi.OK
    assert bytes(buf[48:]) == b"\xab" * 32, "svgp_last_timing wrote past the v3 layout"
    t3 = _ffi.Timing.from_buffer_copy(bytes(buf[:64]))
    timed = ambient_on("SVGP_TIMING")
    assert (t3.ms_total > 0 and t3.ms_prep > 0 and t3.strip_launches >= 1) if timed else t3.ms_total == 0
    for nbytes in (0, 8, 48, 56, 64, 80):
        buf = (C.c_ubyte * 80)(*([0xCD] * 80))
        assert lib.svgp_last_timing_sized(ctx.h, buf, nbytes) == _ffi.OK
        wrote = min(nbytes, C.sizeof(_ffi.Timing))
        assert bytes(buf[wrote:]) == b"\xcd" * (80 - wrote), nbytes
    t = ctx.timing()
    assert (t.ms_chol > 0 if timed else t.ms_chol == 0) and t.ms_chol <= t.ms_prep and t.ms_total == t3.ms_total
    assert lib.svgp_last_timing_sized(ctx.h, buf, -1) == _ffi.INVALID_ARG
    model.free()
    data.free()


def test_dimension_beyond_the_maximum_is_unsupported_everywhere(ctx):
    """ADVICE r3 (low, 3): the header promises SVGP_UNSUPPORTED for d > SVGP_MAX_D (the host falls back); the data entry points
    answered SVGP_INVALID_ARG (an ArgumentError in the binding)."""
    lib = ctx.lib
    x65 = np.zeros((65, 10))
    h = C.c_void_p()
    rc = lib.svgp_data_upload(ctx.h, _ffi.F64, _ffi.COLVECS, 65, 10, x65.ctypes.data_as(C.c_void_p), None, C.byref(h))
    assert rc == _ffi.UNSUPPORTED
    rc = lib.svgp_data_wrap_device(ctx.h, _ffi.F64, 65, 10, 10, C.c_void_p(0x1000), None, C.byref(h))
    assert rc == _ffi.UNSUPPORTED
    with pytest.raises(_ffi.UnsupportedError):
        _ffi.DeviceData(ctx, x65, np.zeros(10), np.float64)
    rc = lib.svgp_data_upload(ctx.h, _ffi.F64, _ffi.COLVECS, 0, 10, x65.ctypes.data_as(C.c_void_p), None, C.byref(h))
    assert rc == _ffi.INVALID_ARG


# The overlap settings are read once per context (csrc/knobs.hpp): every variant below is its own context on the same inputs.
# SVGP_OVERLAP_MIN_PANELS / _HEAD / _HEAD_MIN_PANELS take effect in the experiments build only (the product build overlaps from
# five panels on and never runs a segmented head); under the product library the small models simply evaluate serially in all
# contexts and the comparison is trivially true for them - the larger ones still cross the two paths.
# (SVGP_TIMING=1: these tests read the timing record to see that the overlapped path was taken - also when the suite runs under an ambient
#  SVGP_TIMING=0, as profiles/round6/gputest_settings.log does)
_OV = dict(SVGP_OVERLAP_MIN_PANELS="2", SVGP_OVERLAP_HEAD="1", SVGP_OVERLAP_HEAD_MIN_PANELS="2", SVGP_TIMING="1")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("N,M,d,lik", [(4096, 512, 8, o.LIK_GAUSSIAN), (16384, 1024, 8, o.LIK_GAUSSIAN), (3000, 300, 3, o.LIK_BERNOULLI_LOGISTIC),
                                      (777, 1500, 2, o.LIK_POISSON_EXP), (20000, 256, 16, o.LIK_GAUSSIAN), (64, 2048, 1, o.LIK_GAUSSIAN),
                                      # more than one round of strips: a segmented head beside the factorisation + the rest behind it
                                      (100000, 512, 8, o.LIK_GAUSSIAN), (70001, 650, 4, o.LIK_BERNOULLI_LOGISTIC), (150000, 300, 2, o.LIK_GAUSSIAN)])
def test_strips_beside_the_factorisation_are_bitwise_the_serial_result(dtype, N, M, d, lik):
    """a batch of at most one round of strips runs as segmented strips on a second stream, panel I behind the
    event of block row I of T, beside the Cholesky of Kuu (api.hip: SegRun).  Per strip the arithmetic is the
    one-launch kernel's, register for register: ELBO, expectation and the per-point marginals must be IDENTICAL bits with the
    overlap on and off, repeatedly (a race between the streams would show as a flaky difference), at several M / widths."""
    x, y, sva, s2 = o.synth_problem(6000 + M, N, M, d, lik=lik, dtype=dtype)
    off = min(17, N - 1)                     # an offset window, ragged end
    with context_with_env(SVGP_OVERLAP="0") as c0:
        model = device_model(c0, sva, dtype=dtype, lik=lik, sigma2=s2)
        data = _ffi.DeviceData(c0, x, y, dtype)
        v0, t0 = model.elbo(data, 0, N, 3.0 * N)
        w0 = model.elbo(data, off, N - off, 0.0)[0]
        assert c0.timing().ms_overlap == 0.0
        model.free(), data.free()
    # small batches: the split closing launch sums the variance in another order (below): off here
    with context_with_env(SVGP_OVERLAP="1", SVGP_SEG_SPLIT="0", **_OV) as c1:
        model = device_model(c1, sva, dtype=dtype, lik=lik, sigma2=s2)
        data = _ffi.DeviceData(c1, x, y, dtype)
        for rep in range(4):
            v1, t1 = model.elbo(data, 0, N, 3.0 * N)
            assert v1 == v0 and t1.expectation == t0.expectation and t1.kl == t0.kl, (rep, v1, v0)
        tm = c1.timing()
        assert model.elbo(data, off, N - off, 0.0)[0] == w0
        model.free(), data.free()
    # the product default: phase 2 of a batch of fewer strips than workgroup slots in a closing launch of its own, several
    # workgroups per strip - identical bits run to run, rounding-level agreement with the unsplit launch
    with context_with_env(SVGP_OVERLAP="1", SVGP_SEG_SPLIT="1", **_OV) as c2:
        model = device_model(c2, sva, dtype=dtype, lik=lik, sigma2=s2)
        data = _ffi.DeviceData(c2, x, y, dtype)
        vs = model.elbo(data, 0, N, 3.0 * N)[0]
        assert model.elbo(data, 0, N, 3.0 * N)[0] == vs
        assert rel(vs, v0) < (1e-13 if dtype == np.float64 else 1e-6), (vs, v0)
        model.free(), data.free()
    ref = o.elbo(sva, x, y, lik=lik, sigma2=s2, num_data=3.0 * N)
    assert rel(v0, ref) < (1e-8 if dtype == np.float64 else 1e-4)
    nP = (M + 127) // 128
    took_it = experiments_build() and M >= 256 or (N, M) in ((16384, 1024), (777, 1500), (64, 2048))   # product: >= 5 panels, one round
    if took_it:   # the path was really taken (one launch per panel + the pre-generation [+ the rest of the batch])
        assert nP + 1 <= tm.strip_launches <= nP + 3, tm.strip_launches
        assert tm.ms_overlap > 0.0


def test_overlapped_strips_report_a_non_positive_definite_kuu(ctx):
    """The strips beside the factorisation wait for events, not for values: a failed Cholesky (a negative jitter beyond the
    smallest eigenvalue) still records every row event, the strips run on garbage and the call returns SVGP_NOT_POSDEF with the
    failing order - no waiter is left behind, and the context works afterwards.  (Six panels: overlapped in the product build too.)"""
    N, M, d = 5000, 768, 2
    x, y, sva, s2 = o.synth_problem(6100, N, M, d)
    bad = o.SVA(sva.kernel, sva.z, sva.m, sva.Lq, jitter=-0.5)
    model = device_model(ctx, bad, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, np.float64)
    with pytest.raises(_ffi.PosDefException) as ei:
        model.elbo(data, 0, N, float(N))
    with pytest.raises(o.PosDefException) as ref:
        o.posterior(bad)
    assert ei.value.info == ref.value.info and 0 < ei.value.info <= M
    model.free()
    good = device_model(ctx, sva, sigma2=s2)
    v = good.elbo(data, 0, N, float(N))[0]
    if ambient_on("SVGP_OVERLAP") and ambient_on("SVGP_TIMING"):   # (the module's context carries the ambient settings)
        assert ctx.timing().ms_overlap > 0.0
    assert rel(v, o.elbo(sva, x, y, sigma2=s2, num_data=float(N))) < 1e-8
    good.free()
    data.free()


def test_batches_of_many_rounds_keep_the_one_launch_path(ctx):
    """More than eight rounds of strips (or fewer than four panels): the dynamic-queue kernel behind the prep alone - a head of one round
    would be noise there."""
    N, M, d = 70000, 256, 4
    x, y, sva, s2 = o.synth_problem(6200, N, M, d)
    model = device_model(ctx, sva, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, np.float64)
    v = model.elbo(data, 0, N, float(N))[0]
    assert ctx.timing().strip_launches <= 2 and ctx.timing().ms_overlap == 0.0
    assert rel(v, o.elbo(sva, x, y, sigma2=s2, num_data=float(N))) < 1e-8
    model.free()
    data.free()


@pytest.mark.parametrize("dtype,gtol", [(np.float64, 1e-11), (np.float32, 3e-4)])   # fp32: the two SYRKs round differently and the Cholesky adjoint amplifies it
@pytest.mark.parametrize("N,M,clamp", [(3001, 200, False), (70001, 130, False), (1000, 64, True)])
def test_uniform_weight_syrk_equals_the_weighted_one(ctx, dtype, gtol, N, M, clamp):
    """Gaussian likelihood: d E_i / d v_i = -1 / (2 sigma^2) for every point, so W = A diag(2 g_v) A' is w A A' and the SYRK runs its
    unweighted loop (grad.hip: UW).  Against the per-point weighted SYRK: same gradient to rounding, on batches whose length is not a
    multiple of the 16-point k-step (the replicated columns of the last strip must not count), over more than one gradient chunk, and
    with the clamping policy on a posterior with negative variances.  The weighted SYRK is what the host-evaluated route runs
    (svgp_elbo_grad_ext: the weights are the host's), so the comparison needs no knob: the host hands back exactly the Gaussian's
    point gradients; in the experiments build SVGP_SYRK_UNIFORM=0 selects the weighted loop for the built-in likelihood as well."""
    x, y, sva, s2 = o.synth_problem(7000 + M, N, M, 3, dtype=dtype)
    if clamp:   # a small NEGATIVE jitter and a tiny cov(q): at the points next to an inducing input k - sum A^2 dips below zero and is
        # clamped (tests/test_gpu_parity.py::test_error_statuses builds its case the same way); their g_v is still -scale / (2 sigma^2)
        x = np.concatenate([np.asarray(sva.z), x[:, : N - M]], axis=1).astype(dtype)
        sva = o.SVA(o.Kernel(o.KERNEL_SE, 1.0, [6.0, 6.0, 6.0]), sva.z, np.zeros(M), 1e-3 * np.eye(M), jitter=-1e-3)
    pol = _ffi.NEGVAR_CLAMP if clamp else _ffi.NEGVAR_ERROR

    def host_weights(model, data, off, n):
        mu, var = model.marginals(data, off, n)
        yb = np.asarray(y, dtype=np.float64)[off:off + n]
        sum_e = float(np.sum(-0.5 * (np.log(2 * np.pi * s2) + ((yb - mu) ** 2 + var) / s2)))
        return sum_e, (yb - mu) / s2, np.full(n, -0.5 / s2)

    model = device_model(ctx, sva, dtype=dtype, sigma2=s2, neg_var_policy=pol)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    v1, t1, g1 = model.elbo_grad(data, 0, N, 2.0 * N)
    w1, _, h1 = model.elbo_grad(data, 5, N - 9, 0.0)
    v0, _, g0 = model.elbo_grad(data, 0, N, 2.0 * N, ext=host_weights(model, data, 0, N))
    w0, _, h0 = model.elbo_grad(data, 5, N - 9, 0.0, ext=host_weights(model, data, 5, N - 9))
    vt = 1e-12 if dtype == np.float64 else 2e-5   # (the host's sum E against the device's: another summation order)
    assert rel(v1, v0) < vt and rel(w1, w0) < vt
    if clamp:
        assert t1.n_neg_var > 0
    pairs = [(g1, g0), (h1, h0)]
    model.free(), data.free()
    if experiments_build():
        with context_with_env(SVGP_SYRK_UNIFORM="0") as c:
            model = device_model(c, sva, dtype=dtype, sigma2=s2, neg_var_policy=pol)
            data = _ffi.DeviceData(c, x, y, dtype)
            vk, _, gk = model.elbo_grad(data, 0, N, 2.0 * N)
            assert vk == v1          # the value does not depend on the SYRK at all
            pairs.append((g1, gk))
            model.free(), data.free()
    for a, b in pairs:
        for k in ("z", "m", "Lq", "inv_lengthscale"):
            p, q = np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64)
            assert np.abs(p - q).max() <= gtol * max(np.abs(q).max(), 1e-30), k
        assert abs(a["variance"] - b["variance"]) <= gtol * max(abs(b["variance"]), 1e-12) * 10


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("N,M,d,lik", [(8192, 1024, 8, o.LIK_GAUSSIAN), (3000, 700, 3, o.LIK_BERNOULLI_LOGISTIC), (16384, 640, 8, o.LIK_POISSON_EXP),
                                      (500, 1300, 2, o.LIK_GAUSSIAN)])
def test_gradient_strips_beside_the_factorisation_are_bitwise_the_serial_result(dtype, N, M, d, lik):
    """The training step's strips run their phase 1 beside the factorisation too (segmented, panel I behind the event of block row
    I of T) and their phase 3 behind the M-sized gradient prep (Linv, alpha, R) that the main stream computes meanwhile: value and
    EVERY gradient block must be identical bits with SVGP_OVERLAP on and off, repeatedly.  (Five panels or more: overlapped in the
    product build too.)"""
    x, y, sva, s2 = o.synth_problem(8000 + M, N, M, d, lik=lik, dtype=dtype)
    with context_with_env(SVGP_OVERLAP="0") as c0:
        model = device_model(c0, sva, dtype=dtype, lik=lik, sigma2=s2)
        data = _ffi.DeviceData(c0, x, y, dtype)
        v0, t0, g0 = model.elbo_grad(data, 0, N, 2.0 * N)
        f0 = model.elbo(data, 0, N, 2.0 * N)[0]
        model.free(), data.free()
    # (the split closing launch of small batches sums the variance in another order: off here, on below)
    with context_with_env(SVGP_OVERLAP="1", SVGP_SEG_SPLIT="0", **_OV) as c1:
        model = device_model(c1, sva, dtype=dtype, lik=lik, sigma2=s2)
        data = _ffi.DeviceData(c1, x, y, dtype)
        for rep in range(3):
            v1, t1, g1 = model.elbo_grad(data, 0, N, 2.0 * N)
            assert v1 == v0, (rep, v1, v0)
            for k in ("z", "m", "Lq", "inv_lengthscale"):
                assert np.array_equal(np.asarray(g1[k]), np.asarray(g0[k])), (rep, k)
            for k in ("variance", "lik_sigma2", "mean_const"):
                assert g1[k] == g0[k], (rep, k)
        # the forward entry point between two gradient calls (shared scratch, shared events)
        assert model.elbo(data, 0, N, 2.0 * N)[0] == f0
        assert c1.timing().ms_overlap > 0.0
        assert model.elbo_grad(data, 0, N, 2.0 * N)[0] == v0
        model.free(), data.free()
    # Split closing launch (the default for batches of fewer strips than workgroup slots): S workgroups per strip share phase 3's
    # output panels, the variance is summed per part and then over the parts - same operations, another association.  Run-to-run
    # identical bits; against the unsplit launch: rounding-level agreement.
    with context_with_env(SVGP_OVERLAP="1", SVGP_SEG_SPLIT="1", **_OV) as c2:
        model = device_model(c2, sva, dtype=dtype, lik=lik, sigma2=s2)
        data = _ffi.DeviceData(c2, x, y, dtype)
        vs, _, gs = model.elbo_grad(data, 0, N, 2.0 * N)
        vs2, _, gs2 = model.elbo_grad(data, 0, N, 2.0 * N)
        assert vs2 == vs
        rt, gt = (1e-13, 1e-10) if dtype == np.float64 else (1e-6, 2e-4)
        assert rel(vs, v0) < rt, (vs, v0)
        for k in ("z", "m", "Lq", "inv_lengthscale"):
            a, b, c = np.asarray(gs[k], dtype=np.float64), np.asarray(g0[k], dtype=np.float64), np.asarray(gs2[k], dtype=np.float64)
            assert np.array_equal(a, c), k
            assert np.abs(a - b).max() <= gt * max(np.abs(b).max(), 1e-300), (k, np.abs(a - b).max(), np.abs(b).max())
        model.free(), data.free()
    val_ref, g_ref = o.elbo_grad(sva, x, y, lik=lik, sigma2=s2, num_data=2.0 * N)
    assert rel(v0, val_ref) < (1e-8 if dtype == np.float64 else 1e-4)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("M,d", [(130, 3), (1100, 8)])   # gradient blocks below one piece of the read-back, and of several pieces
def test_gradient_read_back_in_pieces_and_into_reused_buffers(ctx, dtype, M, d):
    """svgp_elbo_grad returns M^2 elements of Lq_bar to host memory every step: the read-back goes through a pinned staging buffer in up
    to 8 pieces whose host copies overlap the bus (api.hip: grad_finish).  Piece boundaries fall anywhere inside z_bar | m_bar | Lq_bar:
    every block must equal a one-piece reference (models of different sizes sharing the context's staging buffer), and a second call
    into the SAME host arrays (`out=`) must overwrite them with identical bits."""
    N = 2000
    x, y, sva, s2 = o.synth_problem(4100 + M, N, M, d, dtype=dtype)
    model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    v0, _, g0 = model.elbo_grad(data, 0, N, float(N))
    val_ref, g_ref = o.elbo_grad(sva, x, y, sigma2=s2, num_data=float(N))
    assert rel(v0, val_ref) < (1e-8 if dtype == np.float64 else 1e-4)
    tol = 1e-7 if dtype == np.float64 else 5e-3
    for k in ("z", "m", "Lq"):
        a = np.asarray(g0[k], dtype=np.float64).reshape(np.shape(g_ref[k]), order="F")
        assert np.abs(a - g_ref[k]).max() <= tol * max(np.abs(g_ref[k]).max(), 1e-12), k
    assert not np.any(np.triu(np.asarray(g0["Lq"]), 1))          # Lq_bar is lower triangular in the caller's buffer
    keep = {k: np.array(g0[k], copy=True) for k in ("z", "m", "Lq", "inv_lengthscale")}
    for k in ("z", "m", "Lq", "inv_lengthscale"):
        g0[k][...] = 7                                             # stale contents must not survive
    v1, _, g1 = model.elbo_grad(data, 0, N, float(N), out=g0)
    assert v1 == v0 and g1["Lq"] is g0["Lq"] and g1["z"] is g0["z"]
    for k in keep:
        assert np.array_equal(np.asarray(g1[k]), keep[k]), k
    with pytest.raises(ValueError):
        model.elbo_grad(data, 0, N, float(N), out=dict(g0, Lq=np.zeros((M, M + 1), dtype=dtype, order="F")))
    model.free()
    data.free()


def test_context_without_timing_events():
    """SVGP_TIMING=0 at context creation: no timing events on the stream (each record costs the stream ~5 us: 3 % of a 1024-point
    forward call).  Same results bit for bit, svgp_last_timing reports zeros, and nothing is left behind in the HIP error state
    (querying an event that was never recorded would be a sticky error)."""
    x, y, sva, s2 = o.synth_problem(5150, 3000, 700, 3, dtype=np.float64)
    prev = os.environ.get("SVGP_TIMING")   # (restored: a suite run under SVGP_TIMING=0 must keep it for the tests that follow)
    os.environ["SVGP_TIMING"] = "1"
    try:
        ref_ctx = _ffi.Context(0)
        os.environ["SVGP_TIMING"] = "0"
        quiet = _ffi.Context(0)
    finally:
        if prev is None:
            os.environ.pop("SVGP_TIMING", None)
        else:
            os.environ["SVGP_TIMING"] = prev
    out = []
    for c in (ref_ctx, quiet):
        model = device_model(c, sva, dtype=np.float64, sigma2=s2)
        data = _ffi.DeviceData(c, x, y, np.float64)
        v = model.elbo(data, 0, 3000, 3000.0)[0]
        tf = c.timing()
        vg, _, g = model.elbo_grad(data, 0, 3000, 3000.0)
        tg = c.timing()
        mu, var = model.marginals(data, 0, 3000)
        out.append((v, vg, g, mu, var, tf, tg))
        model.free()
        data.free()
    a, b = out
    assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    for k in ("z", "m", "Lq", "inv_lengthscale"):
        assert np.array_equal(np.asarray(a[2][k]), np.asarray(b[2][k])), k
    assert a[5].ms_total > 0 and a[6].ms_total > 0
    assert b[5].ms_total == 0 and b[5].ms_prep == 0 and b[5].ms_chol == 0 and b[6].ms_total == 0
    quiet.close()
    ref_ctx.close()
