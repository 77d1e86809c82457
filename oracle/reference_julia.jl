# oracle/reference_julia.jl — pins the CPU oracle (and with it the HIP library) to the REAL reference.
#
# For anyone who has Julia: evaluates ApproximateGPs.jl itself on the inputs stored in tests/golden/*.npz and
# compares with the oracle values stored beside them (elbo, KL, posterior mean/variance at every point, Lk).
#
#     julia --project=<env with ApproximateGPs, AbstractGPs, KernelFunctions, GPLikelihoods, Distributions,
#                      PDMats, NPZ> oracle/reference_julia.jl [tests/golden]
#
# NOT run in this repository (the build image has no Julia; the reference pins no package versions, so the
# result also depends on the resolved AbstractGPs / GPLikelihoods versions).  Until somebody runs it the oracle
# stays "parity unpinned" (see oracle/svgp_oracle.py header and DESIGN.md §4); this script is the way out.
# It is test infrastructure, never imported by the product.
using ApproximateGPs, AbstractGPs, KernelFunctions, GPLikelihoods, Distributions, LinearAlgebra
using PDMats: PDMat
using NPZ

base_kernel(fam) = fam == 0 ? SqExponentialKernel() : fam == 1 ? Matern32Kernel() : Matern52Kernel()
scalar(a) = a isa AbstractArray ? only(a) : a
relerr(a, b) = abs(a - b) / max(abs(b), 1e-300)

function check(path)
    g = npzread(path)
    fam, lk, qn = Int(scalar(g["family"])), Int(scalar(g["lik"])), Int(scalar(g["quadrature_n"]))
    k = scalar(g["variance"]) * (base_kernel(fam) ∘ ARDTransform(vec(g["inv_lengthscale"])))
    f = GP(k)
    x, z, y = ColVecs(g["x"]), ColVecs(g["z"]), vec(g["y"])
    # q = MvNormal(m, PDMat(Cholesky(LowerTriangular(A))))  as in examples/a-regression/script.jl:110-111
    q = MvNormal(vec(g["m"]), PDMat(Cholesky(LowerTriangular(g["Lq"]))))
    sva = SparseVariationalApproximation(f(z, scalar(g["jitter"])), q)          # NonCentered (SVA:93-95)
    lik = lk == 0 ? GaussianLikelihood(scalar(g["sigma2"])) : lk == 1 ? BernoulliLikelihood() :
          lk == 2 ? PoissonLikelihood() : lk == 3 ? ExponentialLikelihood() : GammaLikelihood(scalar(g["sigma2"]))  # shape α stored in "sigma2"
    quad = qn == 0 ? GPLikelihoods.DefaultExpectationMethod() : GaussHermiteExpectation(qn)
    nd = scalar(g["num_data"]) < 0 ? length(y) : scalar(g["num_data"])
    lfx = LatentGP(f, lik, 1e-18)(x)
    val = elbo(sva, lfx, y; num_data=nd, quadrature=quad)                        # SVA:340-360
    post = posterior(sva)                                                         # SVA:160-187
    μ, v = mean_and_var(post, x)                                                  # SVA:246-253
    Lk = post.data.Kuu.L
    kl = ApproximateGPs.SparseVariationalApproximationModule._prior_kl(sva)       # SVA:364-373
    println(rpad(basename(path), 24),
            " elbo ", relerr(val, scalar(g["elbo"])),
            "  kl ", relerr(kl, scalar(g["kl"])),
            "  max|μ-μ_oracle| ", maximum(abs.(μ .- vec(g["mu"]))),
            "  max|v-v_oracle| ", maximum(abs.(v .- vec(g["v"]))),
            "  max|Lk-Lk_oracle| ", maximum(abs.(Matrix(Lk) .- g["Lk"])))
    return relerr(val, scalar(g["elbo"]))
end

dir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")
worst = maximum(check(joinpath(dir, f)) for f in sort(readdir(dir)) if endswith(f, ".npz"))
println("worst relative ELBO difference reference vs oracle: ", worst, worst <= 1e-8 ? "  (inside the 1e-8 contract)" : "  (OUTSIDE 1e-8)")
