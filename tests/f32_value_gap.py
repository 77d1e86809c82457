"""fp32: the value svgp_elbo_grad returns against svgp_elbo's on the same batch.  The forward strips form the
variance as k(x,x) - sum A^2 + sum (B'A)^2 with fp64 column sums; the value-and-gradient strips have no phase 2 and take it from
k_j' (R A)_j (strip.hip, phase 3) - the same cancellation, other roundings.  Prints |v_grad - v_fwd| / |v_fwd| and both against the
fp64 oracle where it is affordable, for the bench configurations (reduced N) and for ill-conditioned posteriors (variance down to
1e-6 of the prior's: q(u) close to the exact posterior of a low-noise problem)."""
import os, sys
R = os.path.dirname(os.path.abspath(__file__))
for p in ("approximategps.jl_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(R, "..", p))
import numpy as np, svgp_oracle as o
from approxgp import _ffi
from helpers import device_model

ctx = _ffi.Context(0)
rows = []
def run(tag, N, M, d, fam, lik, shrink=None, oracle=True):
    x, y, sva, s2 = o.synth_problem(940 + M, N, M, d, family=fam, lik=lik, dtype=np.float32)
    if shrink is not None:   # an ill-conditioned posterior: S = Lq Lq' scaled down, the mean moved towards the data
        sva = o.SVA(sva.kernel, sva.z, sva.m, (shrink * sva.Lq).astype(np.float32).astype(np.float64), jitter=sva.jitter)
    model = device_model(ctx, sva, dtype=np.float32, lik=lik, sigma2=s2)
    data = _ffi.DeviceData(ctx, x, y, np.float32)
    vf = model.elbo(data, 0, N, float(N))[0]
    vg, _, g = model.elbo_grad(data, 0, N, float(N))
    mu, var = model.marginals(data, 0, N)
    line = f"{tag}: N={N} M={M} d={d} shrink={shrink}: |v_grad - v_fwd|/|v_fwd| = {abs(vg - vf) / abs(vf):.2e}; min var / prior var = {var.min() / sva.kernel.variance:.2e}"
    if oracle:
        vr, gr = o.elbo_grad(sva, x, y, lik=lik, sigma2=s2, num_data=float(N))
        e = lambda a, b: float(np.abs(np.asarray(a, dtype=float).reshape(np.shape(b), order="F") - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-12))
        line += f"; vs fp64 oracle: fwd {abs(vf - vr) / abs(vr):.2e} grad-call {abs(vg - vr) / abs(vr):.2e}; gradient blocks " + " ".join(f"{k} {e(g[k], gr[k]):.1e}" for k in ("m", "Lq", "z", "inv_lengthscale"))
    print(line, flush=True)
    model.free(); data.free()

run("H32-shape", 20000, 1024, 8, o.KERNEL_SE, o.LIK_GAUSSIAN)
run("C5-shape", 16384, 1024, 8, o.KERNEL_SE, o.LIK_GAUSSIAN)
run("C3-shape", 8000, 2048, 16, o.KERNEL_MATERN52, o.LIK_BERNOULLI_LOGISTIC)
run("C2-shape-f32", 20000, 512, 8, o.KERNEL_SE, o.LIK_GAUSSIAN)
for sh in (0.3, 0.1, 0.03, 0.01, 0.003):
    run("ill-conditioned", 6000, 640, 4, o.KERNEL_SE, o.LIK_GAUSSIAN, shrink=sh)
run("H32 full", 1000000, 1024, 8, o.KERNEL_SE, o.LIK_GAUSSIAN, oracle=False)
run("C3 full", 1000000, 2048, 16, o.KERNEL_MATERN52, o.LIK_BERNOULLI_LOGISTIC, oracle=False)
run("C5 full", 262144, 1024, 8, o.KERNEL_SE, o.LIK_GAUSSIAN, oracle=False)
