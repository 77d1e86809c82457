"""Generates tests/golden/*.npz from the CPU oracle (oracle/svgp_oracle.py) on seeded inputs.

The reference (pure Julia) cannot run in the build container and holds no golden numbers of its own
for this path, so these fixtures pin the HIP library to the *oracle*, which is itself pinned by
tests/test_oracle.py (reference assertions re-stated + mpmath).  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import svgp_oracle as o  # noqa: E402

CASES = [
    # name, config_id, N, M, d, family, lik, quadrature_n, num_data
    ("c1_se_gauss", 1, 1000, 32, 1, o.KERNEL_SE, o.LIK_GAUSSIAN, 0, None),
    ("m52_bern_gh20", 3, 300, 40, 4, o.KERNEL_MATERN52, o.LIK_BERNOULLI_LOGISTIC, 0, 1234.5),
    ("m32_poisson", 4, 257, 17, 2, o.KERNEL_MATERN32, o.LIK_POISSON_EXP, 0, None),
    ("se_gauss_gh7", 5, 129, 130, 3, o.KERNEL_SE, o.LIK_GAUSSIAN, 7, 5000.0),
    ("m52_gamma", 6, 211, 23, 3, o.KERNEL_MATERN52, o.LIK_GAMMA_EXP, 0, 999.0),
    ("se_exponential_gh11", 7, 150, 31, 2, o.KERNEL_SE, o.LIK_EXPONENTIAL_EXP, 11, None),
]


def main():
    only = sys.argv[1:]   # optional: names of the cases to (re)generate
    for name, cid, N, M, d, fam, lik, qn, nd in CASES:
        if only and name not in only:
            continue
        x, y, sva, s2 = o.synth_problem(cid, N, M, d, family=fam, lik=lik)
        if name.startswith("c1"):
            rng = np.random.default_rng(99)
            x = rng.uniform(-1, 1, (1, N))          # examples/a-regression/script.jl:34
            sva.z = x[:, :M].copy()
            sva.kernel = o.Kernel(fam, 1.3, [1 / 0.3])  # :62-63
            y = np.sin(3 * x[0]) + np.sqrt(s2) * rng.standard_normal(N)
        t = o.elbo_terms(sva, x, y, lik=lik, sigma2=s2, num_data=nd, quadrature_n=qn)
        post = o.posterior(sva)
        xs = x[:, :9]
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            x=x, y=y, z=sva.z, m=sva.m, Lq=sva.Lq, inv_lengthscale=sva.kernel.inv_lengthscale,
            variance=sva.kernel.variance, family=fam, lik=lik, sigma2=s2, jitter=sva.jitter, quadrature_n=qn,
            num_data=-1.0 if nd is None else nd,
            elbo=t.elbo, expectation=t.expectation, kl=t.kl, mu=t.mu, v=t.v,
            Lk=post.Lk, alpha=post.alpha, cov9=o.cov(post, xs), kuf9=o.kernelmatrix(sva.kernel, sva.z, xs),
        )
        print(name, t.elbo)


if __name__ == "__main__":
    main()
