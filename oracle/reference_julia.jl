# oracle/reference_julia.jl — pins the CPU oracle (and with it the HIP library) to the REAL reference.
#
# For anyone who has Julia: evaluates ApproximateGPs.jl itself on the inputs stored in tests/golden/*.npz and compares
# with the oracle values stored beside them.  ONE run covers everything the parity tests rely on:
#   * both parametrisations: NonCentered (SVA:160-187, :364-373) and Centered (SVA:115-136, :362);
#   * elbo (SVA:340-360), _prior_kl, posterior data (Lk, α, B), mean_and_var (SVA:246-253), cov (SVA:223-228),
#     cov(f, x, y) (SVA:255-264);
#   * reverse-mode gradients of the elbo w.r.t. kernel variance, inverse lengthscales, z, m and the lower factor of cov(q),
#     taken with Zygote the way test/SparseVariationalApproximationModule.jl:163-175 and the examples do;
#   * Float64 fixtures (tolerance 1e-8 on the elbo) and the Float32 fixture (inputs converted to Float32; tolerance 1e-4).
#
#     julia --project=<env with ApproximateGPs, AbstractGPs, KernelFunctions, GPLikelihoods, Distributions,
#                      PDMats, Zygote, NPZ> oracle/reference_julia.jl [tests/golden]
#
# NOT run in this repository (the build image has no Julia; the reference pins no package versions, so the result also
# depends on the resolved AbstractGPs / GPLikelihoods versions).  Until somebody runs it the oracle stays "parity
# unpinned" (see oracle/svgp_oracle.py header, oracle/CONVENTIONS.md and DESIGN.md); this script is the way out.
# It is test infrastructure, never imported by the product.
using ApproximateGPs, AbstractGPs, KernelFunctions, GPLikelihoods, Distributions, LinearAlgebra
using PDMats: PDMat
using StatsFuns: normcdf
using Zygote
using NPZ

const SVAM = ApproximateGPs.SparseVariationalApproximationModule

base_kernel(fam) = fam == 0 ? SqExponentialKernel() : fam == 1 ? Matern32Kernel() : Matern52Kernel()
scalar(a) = a isa AbstractArray ? only(a) : a
relerr(a, b) = abs(a - b) / max(abs(b), 1e-300)
maxrel(a, b) = maximum(abs.(a .- b)) / max(maximum(abs.(b)), 1e-12)

make_lik(lk, p) = lk == 0 ? GaussianLikelihood(p) : lk == 1 ? BernoulliLikelihood() :
                  lk == 2 ? PoissonLikelihood() : lk == 3 ? ExponentialLikelihood() :
                  lk == 5 ? BernoulliLikelihood(normcdf) : GammaLikelihood(p)   # Gamma shape α travels in "sigma2"; 5 = NormalCDFLink

# the model as a function of its differentiable parameters (so Zygote sees every dependency)
function build(T, fam, variance, invl, z, m, A, jitter, c, centered)
    k = variance * (base_kernel(fam) ∘ ARDTransform(invl))
    f = c == 0 ? GP(k) : GP(T(c), k)
    q = MvNormal(m, PDMat(Cholesky(LowerTriangular(A))))     # examples/a-regression/script.jl:110-111
    fz = f(ColVecs(z), T(jitter))
    sva = centered ? SparseVariationalApproximation(Centered(), fz, q) : SparseVariationalApproximation(NonCentered(), fz, q)
    return f, sva
end

function check(path)
    g = npzread(path)
    T = Int(scalar(g["f32"])) == 1 ? Float32 : Float64
    tol_elbo = T == Float32 ? 1e-4 : 1e-8
    cv(a) = T.(a)
    fam, lk, qn = Int(scalar(g["family"])), Int(scalar(g["lik"])), Int(scalar(g["quadrature_n"]))
    centered = Int(scalar(g["centered"])) == 1
    variance, invl = T(scalar(g["variance"])), cv(vec(g["inv_lengthscale"]))
    z, m, A = cv(g["z"]), cv(vec(g["m"])), cv(g["Lq"])
    x, y = ColVecs(cv(g["x"])), cv(vec(g["y"]))
    jitter, c = scalar(g["jitter"]), scalar(g["mean_const"])
    lik = make_lik(lk, T(scalar(g["sigma2"])))
    quad = qn == 0 ? GPLikelihoods.DefaultExpectationMethod() : GaussHermiteExpectation(qn)
    nd = scalar(g["num_data"]) < 0 ? length(y) : scalar(g["num_data"])

    f, sva = build(T, fam, variance, invl, z, m, A, jitter, c, centered)
    lfx = LatentGP(f, lik, 1e-18)(x)
    val = elbo(sva, lfx, y; num_data=nd, quadrature=quad)                        # SVA:340-360
    kl = SVAM._prior_kl(sva)                                                      # SVA:362 / :364-373
    post = posterior(sva)                                                         # SVA:115-136 / :160-187
    μ, v = mean_and_var(post, x)                                                  # SVA:246-253
    xs, xt = ColVecs(cv(g["x"][:, 1:9])), ColVecs(cv(g["x"][:, 10:16]))
    C9 = cov(post, xs)                                                            # SVA:223-228
    Cx = cov(post, xs, xt)                                                        # SVA:255-264
    Lk, α, B = Matrix(post.data.Kuu.L), post.data.α, Matrix(post.data.B)

    # Zygote gradient of the elbo w.r.t. (variance, inverse lengthscales, z, m, A)
    loss(variance, invl, z, m, A) = begin
        f_, sva_ = build(T, fam, variance, invl, z, m, A, jitter, c, centered)
        elbo(sva_, LatentGP(f_, lik, 1e-18)(x), y; num_data=nd, quadrature=quad)
    end
    gvar, ginvl, gz, gm, gA = Zygote.gradient(loss, variance, invl, z, m, A)
    gA = LowerTriangular(gA)                                                      # only the lower triangle of A is read

    rows = [
        ("elbo", relerr(val, scalar(g["elbo"])), tol_elbo),
        ("kl", relerr(kl, scalar(g["kl"])), T == Float32 ? 1e-5 : 1e-10),
        ("mean", maximum(abs.(μ .- vec(g["mu"]))), T == Float32 ? 1e-4 : 1e-9),
        ("var", maximum(abs.(v .- vec(g["v"]))), T == Float32 ? 1e-4 : 1e-9),
        ("cov", maximum(abs.(C9 .- g["cov9"])), T == Float32 ? 1e-4 : 1e-9),
        ("cov(x,y)", maximum(abs.(Cx .- g["cov_cross"])), T == Float32 ? 1e-4 : 1e-9),
        ("Lk", maxrel(Lk, g["Lk"]), T == Float32 ? 1e-3 : 1e-7),
        ("alpha", maxrel(α, vec(g["alpha"])), T == Float32 ? 1e-2 : 1e-5),
        ("B", maxrel(LowerTriangular(B), g["B"]), T == Float32 ? 1e-3 : 1e-7),
        ("d/dvariance", relerr(gvar, scalar(g["g_variance"])), T == Float32 ? 3e-3 : 1e-6),
        ("d/dinvl", maxrel(ginvl, vec(g["g_inv_lengthscale"])), T == Float32 ? 3e-3 : 1e-6),
        ("d/dz", maxrel(gz, g["g_z"]), T == Float32 ? 3e-3 : 1e-6),
        ("d/dm", maxrel(gm, vec(g["g_m"])), T == Float32 ? 3e-3 : 1e-6),
        ("d/dLq", maxrel(Matrix(gA), g["g_Lq"]), T == Float32 ? 3e-3 : 1e-6),
    ]
    ok = true
    print(rpad(basename(path), 30), centered ? "Centered    " : "NonCentered ", T, "  ")
    for (name, err, tol) in rows
        bad = !(err <= tol)
        ok &= !bad
        print(name, " ", round(err; sigdigits=2), bad ? " (OUTSIDE $(tol))  " : "  ")
    end
    println()
    return ok
end

dir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")
results = [check(joinpath(dir, f)) for f in sort(readdir(dir)) if endswith(f, ".npz")]
println(all(results) ? "ALL FIXTURES AGREE WITH THE REFERENCE: the oracle is pinned." :
        "DISAGREEMENT: $(count(!, results)) of $(length(results)) fixtures differ from the reference (see rows marked OUTSIDE).")
exit(all(results) ? 0 : 1)
