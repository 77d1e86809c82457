"""Randomised STATEFUL sweep against the oracle: several models and data sets of different shapes and dtypes live on ONE context and a
random sequence of operations runs over them - create, svgp_model_update with new parameter values, svgp_elbo, svgp_elbo_grad (plain,
shard form, host-evaluated likelihood), svgp_marginals, svgp_predict, free - every result checked; a quarter of the models Centered, a third with RowVecs storage of z or x.  What it is after is state: workspaces
grown for one shape and reused for another, the prepared flag of a model across updates, the second stream's scratch between a large and
a small batch, pinned staging reused across models.  (tests/fuzz_grad.py / fuzz_forward.py create and free one model per case.)

    python tests/fuzz_stateful.py [--seconds 300] [--seed 26] [--slots 4] [--comm]

A script for the GPU box, not a pytest file; the oracle is the checker.  Exit code 1 when a result is outside its tolerance."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "approximategps.jl_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import svgp_oracle as o  # noqa: E402
from approxgp import _ffi  # noqa: E402
from helpers import rel  # noqa: E402

FAMS = [o.KERNEL_SE, o.KERNEL_MATERN32, o.KERNEL_MATERN52]
LIKS = [o.LIK_GAUSSIAN, o.LIK_GAUSSIAN, o.LIK_BERNOULLI_LOGISTIC, o.LIK_POISSON_EXP]


class Slot:
    """A live (model, data) pair with the oracle-side description of what the device holds."""

    def __init__(self, ctx, rng):
        self.d = int(rng.choice([1, 2, 3, 8, 9, 16, 17, 33, 64]))
        self.M = int(rng.choice([1, 17, 64, 100, 128, 129, 300, 512, 640, 1024]))
        self.N = int(rng.choice([1, 33, 64, 127, 500, 1025, 3000, 9001, 20000]))
        if self.M >= 512 and self.N > 9001:
            self.N = 9001
        self.family, self.lik = int(rng.choice(FAMS)), int(rng.choice(LIKS))
        self.dtype = np.float32 if rng.random() < 0.35 else np.float64
        self.x, self.y, self.sva, self.s2 = o.synth_problem(int(rng.integers(1, 1 << 30)), self.N, self.M, self.d, family=self.family,
                                                             lik=self.lik, dtype=self.dtype)
        self.sva.mean_const = 0.05
        if rng.random() < 0.25:   # Centered: q(u) = N(m, Lq Lq') itself
            nc = self.sva
            tame = 0.1 if self.lik == o.LIK_POISSON_EXP else 1.0
            self.sva = o.SVA(nc.kernel, nc.z, tame * (nc.m + 0.3), 0.7 * tame * nc.Lq, jitter=1e-4 if self.dtype == np.float64 else 1e-2,
                             mean_const=0.15, centered=True)
        # RowVecs storage of the inducing inputs (M x d) and / or of the data (n x d): the same numbers, the other layout code
        self.z_rows = self.d > 1 and rng.random() < 0.3
        self.x_rows = self.d > 1 and rng.random() < 0.3
        self.model = _ffi.DeviceModel(ctx, *self.desc())
        self.data = (_ffi.DeviceData(ctx, self.x.T, self.y, self.dtype, layout=_ffi.ROWVECS) if self.x_rows
                     else _ffi.DeviceData(ctx, self.x, self.y, self.dtype))

    def desc(self):
        s = self.sva
        return _ffi.make_desc(self.dtype, s.kernel.family, s.kernel.variance, s.kernel.inv_lengthscale, s.z.T if self.z_rows else s.z, s.m, s.Lq,
                              s.jitter, parametrization=_ffi.CENTERED if s.centered else _ffi.NONCENTERED, likelihood=self.lik,
                              lik_sigma2=self.s2, mean_const=s.mean_const, layout_z=_ffi.ROWVECS if self.z_rows else _ffi.COLVECS)

    def tag(self):
        return (f"[N={self.N} M={self.M} d={self.d} fam={self.family} lik={self.lik} {self.dtype.__name__}"
                f"{' centered' if self.sva.centered else ''}{' z-rows' if self.z_rows else ''}{' x-rows' if self.x_rows else ''}]")

    def update(self, rng):
        """New parameter values of the same shape (a training step)."""
        k, s = self.sva.kernel, self.sva
        f = lambda a, e: np.asarray(a * (1.0 + e * rng.standard_normal(np.shape(a))), dtype=self.dtype).astype(np.float64)  # noqa: E731
        kern = o.Kernel(k.family, float(k.variance * (1.0 + 0.1 * rng.random())), f(k.inv_lengthscale, 0.05))
        Lq = np.tril(f(s.Lq, 0.02))
        np.fill_diagonal(Lq, np.abs(np.diag(Lq)) + 1e-3)
        z = s.z + np.asarray(0.01 * rng.standard_normal(s.z.shape), dtype=self.dtype).astype(np.float64)
        z = np.asarray(z, dtype=self.dtype).astype(np.float64)
        self.sva = o.SVA(kern, z, f(s.m, 0.05) + 0.01, Lq, jitter=s.jitter, mean_const=float(s.mean_const + 0.01), centered=s.centered)
        self.model.update(*self.desc())

    def zshape(self):
        return (self.M, self.d) if self.z_rows else None

    def window(self, rng):
        if rng.random() < 0.5 or self.N < 4:
            return 0, self.N
        off = int(rng.integers(0, self.N // 2))
        return off, int(rng.integers(1, self.N - off + 1))

    def free(self):
        self.model.free()
        self.data.free()


def block_err(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def check_grads(s, g, g_ref):
    d = s.d
    gz = (g["z"].T if s.z_rows else g["z"].reshape(g_ref["z"].shape, order="F")) if d > 1 else g["z"]
    e = {"z": block_err(gz, g_ref["z"] if d > 1 else g_ref["z"][0])}
    for k in ("m", "Lq", "inv_lengthscale"):
        e[k] = block_err(g[k], g_ref[k])
    e["variance"] = block_err([g["variance"]], [g_ref["variance"]])
    return e


def step(rng, s):
    """One checked operation on a slot -> (name, dict of errors)."""
    off, nb = s.window(rng)
    xs = s.x[:, off:off + nb] if s.x.ndim == 2 else s.x[off:off + nb]
    ys = s.y[off:off + nb]
    op = str(rng.choice(["elbo", "grad", "grad", "shard", "ext", "marginals", "predict"]))
    s.last_window = (off, nb)
    nd = float(rng.choice([0.0, 3.0 * s.N]))
    kw = dict(lik=s.lik, sigma2=s.s2, num_data=nd if nd else None)
    if op == "elbo":
        v = s.model.elbo(s.data, off, nb, nd)[0]
        return op, {"value": rel(v, o.elbo(s.sva, xs, ys, **kw))}
    if op == "grad":
        v, _, g = s.model.elbo_grad(s.data, off, nb, nd, z_shape=s.zshape())
        v_ref, g_ref = o.elbo_grad(s.sva, xs, ys, **kw)
        return op, {"value": rel(v, v_ref), **check_grads(s, g, g_ref)}
    if op == "shard":   # scale and KL weight given by the caller (data-parallel shard form)
        scale, klw = 1.7, 0.25
        v, _, g = s.model.elbo_grad(s.data, off, nb, shard=(scale, klw), z_shape=s.zshape())
        v_ref, g_ref = o.elbo_grad(s.sva, xs, ys, lik=s.lik, sigma2=s.s2, num_data=scale * nb, kl_weight=klw)
        return op, {"value": rel(v, v_ref), **check_grads(s, g, g_ref)}
    if op == "ext":     # the host evaluates the likelihood on the device's marginals
        mu, var = s.model.marginals(s.data, off, nb)
        gmu, gv, _ = o.expected_loglik_grads(s.lik, mu, var, ys, sigma2=s.s2)
        sum_e = o.expected_loglik(s.lik, mu, np.sqrt(var), ys, sigma2=s.s2)
        v, _, g = s.model.elbo_grad(s.data, off, nb, nd, ext=(sum_e, gmu, gv), z_shape=s.zshape())
        v_ref, g_ref = o.elbo_grad(s.sva, xs, ys, **kw)
        return op, {"value": rel(v, v_ref), **check_grads(s, g, g_ref)}
    post = o.posterior(s.sva)
    if op == "marginals":
        mu, var = s.model.marginals(s.data, off, nb)
        mr, vr = o.mean_and_var(post, xs)
        return op, {"mean": block_err(mu, mr), "var": block_err(var, vr + 1e-18)}
    npred = min(nb, 80)
    xp = xs[:, :npred] if xs.ndim == 2 else xs[:npred]
    mean, var, cov = s.model.predict(xp, True, True, True)
    mr, vr = o.mean_and_var(post, xp)
    return op, {"mean": block_err(mean, mr), "var": block_err(var, vr), "cov": block_err(cov, o.cov(post, xp))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=26)
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--comm", action="store_true", help="attach an RCCL communicator of ONE rank: every evaluation takes the collective forms "
                    "(opening all-reduce with the failure flag, device-resident batch size, grouped gradient all-reduce)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = _ffi.Context(0)
    if args.comm:
        ctx.attach_comm(_ffi.comm_unique_id(), 1, 0)
    slots, t0, n, bad, worst = [], time.time(), 0, [], {}
    while time.time() - t0 < args.seconds:
        r = rng.random()
        if len(slots) < args.slots or r < 0.08:
            if len(slots) >= args.slots:
                slots.pop(int(rng.integers(len(slots)))).free()
            slots.append(Slot(ctx, rng))
            print("OP", n, "create", slots[-1].tag(), flush=True)
            n += 1
            continue
        s = slots[int(rng.integers(len(slots)))]
        if r < 0.3:
            s.update(rng)
            print("OP", n, "update", s.tag(), flush=True)
            n += 1
            continue
        f64 = s.dtype == np.float64
        try:
            op, errs = step(rng, s)
        except Exception as e:
            bad.append((n, s.tag(), repr(e)))
            print("OP", n, "EXCEPTION", s.tag(), repr(e), flush=True)
            n += 1
            continue
        tol = {k: ((1e-8 if k == "value" else 1e-6) if f64 else (2e-4 if k == "value" else 2e-2)) for k in errs}
        fails = {k: v for k, v in errs.items() if not (v <= tol[k])}
        for k, v in errs.items():
            key = (op, k, "f64" if f64 else "f32")
            worst[key] = max(worst.get(key, 0.0), v)
        print("OP", n, op, s.tag(), "window", s.last_window, "FAIL" if fails else "ok", {k: f"{v:.1e}" for k, v in (fails or errs).items()}, flush=True)
        if fails:
            bad.append((n, op, s.tag(), fails))
        n += 1
    for s in slots:
        s.free()
    ctx.close()
    print(f"SUMMARY {n} operations in {time.time() - t0:.0f} s, {len(bad)} outside tolerance")
    for k in sorted(worst):
        print("  worst", k, f"{worst[k]:.2e}")
    for b in bad:
        print("  BAD", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
