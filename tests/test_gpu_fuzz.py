"""A short, seeded slice of the randomised sweeps (tests/fuzz_grad.py, fuzz_forward.py, fuzz_stateful.py, fuzz_errors.py; the long runs
and what they found: profiles/round6/fuzz_grad.md) so that every run of the GPU suite crosses random shapes, dtypes, kernel families,
likelihoods, parametrisations, layouts and operation orders as well as the hand-picked ones.  fp64 keeps the suite's tolerances; fp32
one-element blocks (scalars that are the small remainder of sums over the points) get 0.1 instead of 5e-3 - the long sweeps flag a handful
of those per ten thousand cases, with the VALU kernels of earlier rounds as with today's."""
import numpy as np
import pytest

import fuzz_errors
import fuzz_forward
import fuzz_grad
import fuzz_stateful
from approxgp import _ffi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _ffi.Context(0)
    yield c
    c.close()


def test_random_value_and_gradient_cases(ctx):
    rng = np.random.default_rng(606)
    bad = []
    for n in range(160):
        c = fuzz_grad.draw(rng)
        if c["M"] > 300 and c["N"] > 1025:
            c["N"] = 1025   # (keeps the oracle's share of this test to seconds)
        f64 = c["dtype"] == np.float64
        errs = fuzz_grad.run_case(ctx, c)
        for k, v in errs.items():
            scalar = k in ("variance", "mean_const", "lik_sigma2") or (k == "inv_lengthscale" and c["d"] == 1)
            tol = (1e-8 if f64 else 1e-4) if k == "value" else (1e-10 if f64 else 1e-4) if k == "fwd_vs_grad_value" else \
                  (1e-6 if f64 else (0.1 if scalar else 5e-3))
            if not v <= tol:
                bad.append((n, {a: (b.__name__ if a == "dtype" else b) for a, b in c.items()}, k, v))
    assert not bad, bad


def test_random_forward_cases(ctx):
    rng = np.random.default_rng(616)
    bad = []
    for n in range(160):
        c = fuzz_forward.draw(rng)
        f64 = c["dtype"] == np.float64
        for k, v in fuzz_forward.run_case(ctx, c).items():
            tol = 1e-7 if f64 else (2e-2 if k in ("alpha", "marg_mean", "pred_mean") else 3e-3)
            if not v <= tol:
                bad.append((n, {a: (b.__name__ if a == "dtype" else b) for a, b in c.items()}, k, v))
    assert not bad, bad


def test_random_operation_sequence_over_several_live_models(ctx):
    rng = np.random.default_rng(626)
    slots, bad = [], []
    for n in range(220):
        r = rng.random()
        if len(slots) < 4 or r < 0.08:
            if len(slots) >= 4:
                slots.pop(int(rng.integers(len(slots)))).free()
            s = fuzz_stateful.Slot(ctx, rng)
            while s.M >= 512 and s.N > 3000:   # (the oracle again)
                s.free()
                s = fuzz_stateful.Slot(ctx, rng)
            slots.append(s)
            continue
        s = slots[int(rng.integers(len(slots)))]
        if r < 0.3:
            s.update(rng)
            continue
        f64 = s.dtype == np.float64
        op, errs = fuzz_stateful.step(rng, s)
        for k, v in errs.items():
            tol = (1e-8 if k == "value" else 1e-6) if f64 else (2e-4 if k == "value" else (0.1 if k == "variance" else 2e-2))
            if not v <= tol:
                bad.append((n, op, s.tag(), s.last_window, k, v))
    for s in slots:
        s.free()
    assert not bad, bad


def test_hostile_calls_then_a_healthy_one(ctx):
    rng = np.random.default_rng(636)
    for kind in fuzz_errors.HOSTILE:
        dtype = np.float32 if rng.random() < 0.3 else np.float64
        p = fuzz_errors.healthy(ctx, rng, dtype)
        status, val = fuzz_errors.hostile_call(ctx, rng, p, kind)
        assert not (status == "OK" and val is not None and np.isfinite(val)), (kind, status, val)
        model = _ffi.DeviceModel(ctx, *fuzz_errors.desc_of(p))
        data = _ffi.DeviceData(ctx, p["x"], p["y"], dtype)
        v = model.elbo(data, 0, p["N"], 0.0)[0]
        ref = fuzz_errors.o.elbo(p["sva"], p["x"], p["y"], sigma2=p["s2"])
        model.free()
        data.free()
        assert abs(v - ref) <= (1e-8 if dtype == np.float64 else 2e-4) * abs(ref), (kind, v, ref)
