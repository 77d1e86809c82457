"""Generates tests/golden/*.npz from the CPU oracle (oracle/svgp_oracle.py) on seeded inputs.

The reference (pure Julia) cannot run in the build container and holds no golden numbers of its own
for this path, so these fixtures pin the HIP library to the *oracle*, which is itself pinned by
tests/test_oracle.py (reference assertions re-stated + mpmath) and oracle/CONVENTIONS.md.  They are a REGRESSION
fixture, not a pin to the reference: `julia oracle/reference_julia.jl` evaluates the real ApproximateGPs.jl on exactly
these inputs (NonCentered and Centered, elbo / KL / mean / var / cov / cov(x,y) / Zygote gradients, Float64 and Float32)
and is the run that would turn "parity unpinned" into "pinned".  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import svgp_oracle as o  # noqa: E402

CASES = [
    # name, config_id, N, M, d, family, lik, quadrature_n, num_data, centered, dtype
    ("c1_se_gauss", 1, 1000, 32, 1, o.KERNEL_SE, o.LIK_GAUSSIAN, 0, None, False, np.float64),
    ("m52_bern_gh20", 3, 300, 40, 4, o.KERNEL_MATERN52, o.LIK_BERNOULLI_LOGISTIC, 0, 1234.5, False, np.float64),
    ("m32_poisson", 4, 257, 17, 2, o.KERNEL_MATERN32, o.LIK_POISSON_EXP, 0, None, False, np.float64),
    ("se_gauss_gh7", 5, 129, 130, 3, o.KERNEL_SE, o.LIK_GAUSSIAN, 7, 5000.0, False, np.float64),
    ("m52_gamma", 6, 211, 23, 3, o.KERNEL_MATERN52, o.LIK_GAMMA_EXP, 0, 999.0, False, np.float64),
    ("se_exponential_gh11", 7, 150, 31, 2, o.KERNEL_SE, o.LIK_EXPONENTIAL_EXP, 11, None, False, np.float64),
    ("se_exponential_analytic", 8, 180, 19, 3, o.KERNEL_SE, o.LIK_EXPONENTIAL_EXP, 0, 720.0, False, np.float64),
    # Centered parametrisation (SVA:115-136, :362) and a Float32 case (inputs rounded through fp32; expected values are
    # the fp64 oracle's on those inputs, tolerance 1e-4)
    ("centered_m32_gauss", 9, 160, 24, 2, o.KERNEL_MATERN32, o.LIK_GAUSSIAN, 0, 640.0, True, np.float64),
    ("centered_se_bern", 10, 140, 21, 3, o.KERNEL_SE, o.LIK_BERNOULLI_LOGISTIC, 0, None, True, np.float64),
    ("f32_m52_gauss", 11, 400, 48, 5, o.KERNEL_MATERN52, o.LIK_GAUSSIAN, 0, None, False, np.float32),
    # BernoulliLikelihood(NormalCDFLink()) - the "other invlink" of examples/c-comparisons/script.jl:33-34 (code 5)
    ("m52_bern_normcdf", 12, 220, 28, 3, o.KERNEL_MATERN52, o.LIK_BERNOULLI_NORMCDF, 0, 880.0, False, np.float64),
]


def main():
    only = sys.argv[1:]   # optional: names of the cases to (re)generate
    for name, cid, N, M, d, fam, lik, qn, nd, centered, dtype in CASES:
        if only and name not in only:
            continue
        x, y, sva, s2 = o.synth_problem(cid, N, M, d, family=fam, lik=lik, dtype=dtype)
        if centered:   # q(u) itself: a mean near the prior mean 0.25 and a factor of Kuu-like scale
            sva = o.SVA(sva.kernel, sva.z, 0.25 + sva.m, 0.8 * sva.Lq, jitter=sva.jitter, mean_const=0.25, centered=True)
        if name.startswith("c1"):
            rng = np.random.default_rng(99)
            x = rng.uniform(-1, 1, (1, N))          # examples/a-regression/script.jl:34
            sva.z = x[:, :M].copy()
            sva.kernel = o.Kernel(fam, 1.3, [1 / 0.3])  # :62-63
            y = np.sin(3 * x[0]) + np.sqrt(s2) * rng.standard_normal(N)
        t = o.elbo_terms(sva, x, y, lik=lik, sigma2=s2, num_data=nd, quadrature_n=qn)
        post = o.posterior(sva)
        xs, xt = x[:, :9], x[:, 9:16]
        # reverse-mode gradients (the reference: Zygote.gradient of elbo, test/SparseVariationalApproximationModule.jl:163-175)
        _, g = o.elbo_grad(sva, x, y, lik=lik, sigma2=s2, num_data=nd, quadrature_n=qn)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            x=x, y=y, z=sva.z, m=sva.m, Lq=sva.Lq, inv_lengthscale=sva.kernel.inv_lengthscale,
            variance=sva.kernel.variance, family=fam, lik=lik, sigma2=s2, jitter=sva.jitter, quadrature_n=qn,
            num_data=-1.0 if nd is None else nd,
            elbo=t.elbo, expectation=t.expectation, kl=t.kl, mu=t.mu, v=t.v,
            Lk=post.Lk, alpha=post.alpha, cov9=o.cov(post, xs), kuf9=o.kernelmatrix(sva.kernel, sva.z, xs),
            cov_cross=o.cov(post, xs, xt), B=post.B, centered=int(centered), mean_const=sva.mean_const,
            f32=int(np.dtype(dtype) == np.float32),
            g_variance=g["variance"], g_inv_lengthscale=g["inv_lengthscale"], g_z=g["z"], g_m=g["m"], g_Lq=g["Lq"],
            g_lik_sigma2=g["lik_sigma2"], g_mean_const=g["mean_const"],
        )
        print(name, t.elbo)


if __name__ == "__main__":
    main()
