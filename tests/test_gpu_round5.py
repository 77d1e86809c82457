"""GPU tests of round 5: the fp32 value-and-gradient call on ill-conditioned posteriors (VERDICT r4 item 9 / ADVICE r3), the packed
lower-triangle read-back of Lq_bar, batches of several gradient chunks without a knob, the operational environment settings."""
import numpy as np
import pytest

import svgp_oracle as o
from approxgp import _ffi
from helpers import context_with_env, device_model, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _ffi.Context(0)
    yield c
    c.close()


def _blk(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(np.shape(b), order="F")
    return float(np.abs(a - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.mark.parametrize("shrink", [0.1, 0.003])
def test_fp32_value_and_gradient_on_an_ill_conditioned_posterior(ctx, shrink):
    """A posterior whose marginal variance is 1e-2 ... 4.5e-5 of the prior's (cov(q) scaled down: q(u) close to the exact posterior of
    a low-noise problem) is where the variance's cancellation k - sum A^2 + sum (B'A)^2 bites in fp32.  The
    value-and-gradient strips form the variance as k_j'(R A)_j instead (strip.hip, phase 3): the value they return, svgp_elbo's and
    the fp64 oracle's must agree to the bounds include/svgp_mi355x.h states at svgp_elbo_grad, and every gradient block must stay
    within the fp32 tolerances of the well-conditioned tests (measured: profiles/round5/f32_value_gap.log)."""
    N, M, d = 6000, 640, 4
    x, y, sva, s2 = o.synth_problem(940 + M, N, M, d, dtype=np.float32)
    sva = o.SVA(sva.kernel, sva.z, sva.m, (shrink * sva.Lq).astype(np.float32).astype(np.float64), jitter=sva.jitter)
    model = device_model(ctx, sva, dtype=np.float32, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, np.float32)
    vf = model.elbo(data, 0, N, float(N))[0]
    vg, _, g = model.elbo_grad(data, 0, N, float(N))
    mu, var = model.marginals(data, 0, N)
    assert var.min() / sva.kernel.variance < (2e-2 if shrink == 0.1 else 1e-4)      # the posterior IS ill-conditioned
    vr, gr = o.elbo_grad(sva, x, y, sigma2=s2, num_data=float(N))
    assert rel(vg, vf) < 2e-6, (vg, vf)
    assert rel(vf, vr) < 1e-5 and rel(vg, vr) < 1e-5, (vf, vg, vr)
    assert _blk(g["m"], gr["m"]) < 2e-4 and _blk(g["Lq"], gr["Lq"]) < 2e-4
    assert _blk(g["z"], gr["z"]) < 2e-3 and _blk(g["inv_lengthscale"], gr["inv_lengthscale"]) < 2e-3
    assert abs(g["variance"] - gr["variance"]) <= 2e-3 * abs(gr["variance"])
    model.free()
    data.free()


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 2e-3)])
def test_three_gradient_chunks_without_a_knob(ctx, dtype, tol):
    """150 000 points at M = 100: three chunks of the value-and-gradient evaluation at the product chunk size (65 536 points; a batch of
    at most two chunks' worth runs as one) - the per-chunk SYRK / kernel-gradient accumulation, the host-evaluated point gradients read
    at each chunk's offset, and the oracle over all points, with no environment knob (the product library reads none)."""
    N, M, d = 150_000, 100, 2
    x, y, sva, s2 = o.synth_problem(8400, N, M, d, dtype=dtype)
    model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    vb, _, gb = model.elbo_grad(data, 0, N, 2.0 * N)
    vr, gr = o.elbo_grad(sva, x, y, sigma2=s2, num_data=2.0 * N)
    assert rel(vb, vr) < (1e-8 if dtype == np.float64 else 1e-4)
    for k in ("m", "Lq", "z", "inv_lengthscale"):
        assert _blk(gb[k], gr[k]) < tol, k
    if dtype == np.float64:
        mu, var = model.marginals(data)
        gmu, gv = (y - mu) / s2, np.full(N, -0.5 / s2)
        sum_e = float(np.sum(-0.5 * (np.log(2 * np.pi * s2) + ((y - mu) ** 2 + var) / s2)))
        ve, _, ge = model.elbo_grad(data, 0, N, 2.0 * N, ext=(sum_e, gmu, gv))
        assert rel(ve, vb) < 1e-12
        for k in ("m", "Lq", "z", "inv_lengthscale"):
            assert _blk(ge[k], np.asarray(gb[k], dtype=np.float64).reshape(np.shape(gr[k]), order="F")) < 1e-9, k
    model.free()
    data.free()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("M,d", [(1, 1), (2, 3), (129, 2), (1030, 8)])
def test_packed_read_back_of_the_lower_triangle(ctx, dtype, M, d):
    """Lq_bar crosses the bus as its M (M + 1) / 2 lower-triangle entries (pack_tril_kernel -> pinned staging in pieces -> the host copy
    scatters each column into the caller's dense column-major array and zeroes the part above the diagonal).  Against the oracle for
    sizes whose packed stream ends inside a piece, spans several pieces (M = 1030 f64: 8 pieces) or is a single entry; stale contents
    of a reused output array must not survive anywhere, the strict upper triangle is exactly zero."""
    N = 700
    x, y, sva, s2 = o.synth_problem(8500 + M, N, M, d, dtype=dtype)
    model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    v0, _, g0 = model.elbo_grad(data, 0, N, float(N))
    vr, gr = o.elbo_grad(sva, x, y, sigma2=s2, num_data=float(N))
    assert rel(v0, vr) < (1e-8 if dtype == np.float64 else 1e-4)
    tol = 1e-7 if dtype == np.float64 else 5e-3
    for k in ("z", "m", "Lq"):
        assert _blk(g0[k], gr[k]) <= tol, k
    Lb = np.asarray(g0["Lq"])
    assert Lb.shape == (M, M) and not np.any(np.triu(Lb, 1))
    keep = {k: np.array(g0[k], copy=True) for k in ("z", "m", "Lq")}
    for k in keep:
        g0[k][...] = 7
    v1, _, g1 = model.elbo_grad(data, 0, N, float(N), out=g0)
    assert v1 == v0
    for k in keep:
        assert np.array_equal(np.asarray(g1[k]), keep[k]), k
    model.free()
    data.free()


def test_operational_settings_are_read_at_context_creation():
    """SVGP_OVERLAP / SVGP_SEG_SPLIT / SVGP_TIMING are read ONCE, in svgp_ctx_create (csrc/knobs.hpp): changing the environment
    afterwards changes nothing for an existing context - no getenv on the evaluation path."""
    import os
    N, M, d = 8192, 1024, 4
    x, y, sva, s2 = o.synth_problem(8600, N, M, d)
    with context_with_env(SVGP_OVERLAP="0", SVGP_TIMING="1") as c0, context_with_env(SVGP_OVERLAP="1", SVGP_TIMING="1") as c1:
        res = {}
        for name, c in (("off", c0), ("on", c1)):
            model = device_model(c, sva, sigma2=s2)
            data = _ffi.DeviceData(c, x, y, np.float64)
            ambient = {k: os.environ.get(k) for k in ("SVGP_OVERLAP", "SVGP_TIMING")}   # (the suite itself may run under a setting)
            os.environ["SVGP_OVERLAP"] = "1" if name == "off" else "0"      # the opposite of what the context was created with
            os.environ["SVGP_TIMING"] = "0"
            try:
                v = model.elbo(data, 0, N, float(N))[0]
                t = c.timing()
            finally:
                for k, v0 in ambient.items():
                    if v0 is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v0
            res[name] = (v, t.ms_overlap, t.ms_total)
            model.free(), data.free()
    assert res["off"][0] == res["on"][0]
    assert res["off"][1] == 0.0 and res["on"][1] > 0.0          # each context kept the setting it was created with
    assert res["off"][2] > 0.0 and res["on"][2] > 0.0           # ... and its timing events


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 2e-3)])
def test_large_kuu_factorisation(ctx, dtype, tol):
    """M = 2300 (18 panels: more than the 16 row events, so cholesky(Kuu) takes the large-Kuu path: 512-thread trailing updates with
    the next block factorisation fused in, one T-panel launch at the end).  The factor must be LAPACK's to rounding, the ELBO the oracle's, and
    repeated evaluations identical bits (a missing dependency would show as a run-to-run difference or a wrong tile)."""
    N, M, d = 3000, 2300, 3
    x, y, sva, s2 = o.synth_problem(8700, N, M, d, dtype=dtype)
    model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    vals = {model.elbo(data, 0, N, float(N))[0] for _ in range(6)}
    assert len(vals) == 1, vals
    Lk, alpha, B = model.posterior()
    K = o.kuu(sva)
    L = np.tril(np.asarray(Lk, dtype=np.float64))
    back = np.linalg.norm(L @ L.T - K) / np.linalg.norm(K)
    assert back < (5e-15 if dtype == np.float64 else 2e-6), back
    Lref = np.linalg.cholesky(K)
    assert np.abs(L - Lref).max() <= tol * np.abs(Lref).max()
    assert rel(vals.pop(), o.elbo(sva, x, y, sigma2=s2, num_data=float(N))) < (1e-8 if dtype == np.float64 else 1e-4)
    # a non-positive-definite Kuu is reported with LAPACK's info through these launches too
    bad = device_model(ctx, o.SVA(sva.kernel, sva.z, sva.m, sva.Lq, jitter=-0.5), dtype=dtype, sigma2=s2)
    with pytest.raises(_ffi.PosDefException) as ei:
        bad.elbo(data, 0, N, float(N))
    assert 0 < ei.value.info <= M
    assert model.elbo(data, 0, N, float(N))[0] == model.elbo(data, 0, N, float(N))[0]
    bad.free()
    model.free()
    data.free()


@pytest.mark.parametrize("dtype,atol", [(np.float64, 1e-12), (np.float32, 3e-6)])
@pytest.mark.parametrize("family", [o.KERNEL_SE, o.KERNEL_MATERN32, o.KERNEL_MATERN52])
def test_kuf_assembly_at_every_padded_feature_count(ctx, dtype, atol, family):
    """svgp_kuf = cov(f.prior, z, x) (SVA:216) for every padded feature count of the standalone assembly kernels (4, 8, 16, 20, 24, 32,
    48, 64 feature rows: strip.hip, launch_kuf_f), each at its upper edge and one past the edge below it, on both kernel forms: the
    column-owning kernel (a column of Kuf is at most 8 KiB: M = 150, one row chunk, and M = 600, three chunks with a ragged last one)
    and the block kernel (M = 1100 f64 / 2100 fp32).  Ragged batch, window offset 3."""
    rng = np.random.default_rng(77)
    for d in (3, 5, 9, 16, 17, 20, 21, 24, 25, 32, 33, 48, 49, 64):
        for M in (150, 600, 1100 if dtype == np.float64 else 2100):
            if M > 600 and d not in (9, 24, 33, 64):
                continue
            N = 333
            x, y, sva, s2 = o.synth_problem(5000 + 7 * d + M, N + 3, M, d, family=family, dtype=dtype)
            model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
            data = _ffi.DeviceData(ctx, x, y, dtype)
            K = model.kuf(data, 3, N)
            ref = o.kernelmatrix(sva.kernel, sva.z, x[:, 3:])
            assert K.shape == ref.shape, (d, M, K.shape, ref.shape)
            err = np.abs(K - ref).max()
            assert err <= atol * sva.kernel.variance, (d, M, err)
            model.free()
            data.free()


@pytest.mark.parametrize("d", [9, 16])
@pytest.mark.parametrize("family", [o.KERNEL_SE, o.KERNEL_MATERN52])
@pytest.mark.parametrize("dtype,tol,gtol", [(np.float64, 1e-8, 1e-6), (np.float32, 1e-4, 3e-3)])
def test_gradient_on_the_sixteen_feature_reduction_kernel(ctx, dtype, tol, gtol, family, d):
    """8 < d <= 16 takes the 16-feature form of kgrad_kernel (grad.hip), since round 5 with ONE row per lane in f64 (two rows spilled
    111-143 VGPRs) and two in fp32: value and every gradient block against the oracle's, M spanning several 64- / 128-row blocks with a
    ragged last one, a ragged batch of several slices."""
    N, M = 2311, 300
    x, y, sva, s2 = o.synth_problem(5200 + d, N, M, d, family=family, dtype=dtype)
    model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, dtype)
    val_ref, g_ref = o.elbo_grad(sva, x, y, sigma2=s2, num_data=2.0 * N)
    val, _, g = model.elbo_grad(data, 0, N, 2.0 * N)
    assert rel(val, val_ref) < tol
    for k in ("m", "Lq", "z", "inv_lengthscale"):
        assert _blk(g[k], g_ref[k]) <= gtol, (k, _blk(g[k], g_ref[k]))
    model.free()
    data.free()


@pytest.mark.parametrize("dtype,tol,gtol", [(np.float64, 1e-8, 1e-6), (np.float32, 1e-4, 3e-3)])
def test_wide_input_gradient_on_tiny_and_ragged_batches(ctx, dtype, tol, gtol):
    """The wide-input kernel-gradient reductions (grad.hip: lanes per row in f64, a wave per 16-feature group in fp32) walk a batch in
    groups of 4 points per wave and blocks of 64 / 128 points: batches of 1, 3, 5 and 131 points (shorter than a group, than a block,
    one point past a block), d = 20 and 40, against the oracle."""
    for N, d in ((1, 20), (3, 40), (5, 20), (131, 40)):
        M = 70
        x, y, sva, s2 = o.synth_problem(6100 + 10 * N + d, max(N, 2), M, d, family=o.KERNEL_MATERN32, dtype=dtype)
        x, y = x[:, :N], y[:N]
        model = device_model(ctx, sva, dtype=dtype, sigma2=s2)
        data = _ffi.DeviceData(ctx, x, y, dtype)
        val_ref, g_ref = o.elbo_grad(sva, x, y, sigma2=s2, num_data=9.0 * N)
        val, _, g = model.elbo_grad(data, 0, N, 9.0 * N)
        assert rel(val, val_ref) < tol, (N, d)
        for k in ("m", "Lq", "z", "inv_lengthscale"):
            assert _blk(g[k], g_ref[k]) <= gtol, (N, d, k, _blk(g[k], g_ref[k]))
        model.free()
        data.free()
