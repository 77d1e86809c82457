# integration/julia/test/runtests.jl — for a maintainer with Julia, a patched ApproximateGPs.jl (ApproximateGPs_hooks.patch)
# and an MI355X:   SVGP_MI355X_LIB=/path/to/libsvgp_mi355x.so julia --project=<env> integration/julia/test/runtests.jl
#
# Every comparison is device path (hooks enabled) against the reference's own pure-Julia body (hooks disabled) ON THE SAME
# CALL, so nothing here depends on this repository's oracle.  NOT run in this repository (no Julia in the build image);
# the C structs and status codes it relies on are the ones tests/test_abi_cpu.py pins for the ctypes mirror.
using Test, Random, LinearAlgebra
using ApproximateGPs, AbstractGPs, KernelFunctions, GPLikelihoods, Distributions, Zygote
using PDMats: PDMat
using StatsFuns: normcdf
const MI = ApproximateGPs.SVGPMI355X
# the problems below are small on purpose (the reference's own test sizes): without this the hooks would decline them all
# (svgp_offload_advice, the small-problem rule) and every comparison would be Julia against Julia
ENV["SVGP_OFFLOAD_MIN_WORK"] = "0"

@testset "struct layouts = include/svgp_mi355x.h = approxgp/_ffi.py" begin
    @test sizeof(MI.ModelDesc) == 104 && fieldoffset(MI.ModelDesc, 9) == 32 && fieldoffset(MI.ModelDesc, 17) == 96
    @test sizeof(MI.Terms) == 64 && fieldoffset(MI.Terms, 8) == 56
    @test sizeof(MI.Grads) == 56 && fieldoffset(MI.Grads, 4) == 24
    @test ccall((:svgp_version, MI.lib), Int32, ()) == 5
end

on(f) = (MI.enable!(true); f())
off(f) = try MI.enable!(false); f() finally MI.enable!(true) end

function problem(rng, T; N=400, M=24, d=3, base=SqExponentialKernel(), centered=false, lik=GaussianLikelihood(T(0.3)), c=0.0)
    X = randn(rng, T, d, N); x = ColVecs(X)
    z = ColVecs(X[:, 1:M] .+ T(1e-3) .* randn(rng, T, d, M))
    θ = (var=T(1.3), invl=T.(1 ./ (sqrt(d) .* (0.75 .+ 0.5 .* (0:d-1) ./ d))), Z=z.X, m=T(0.1) .* randn(rng, T, M),
         A=Matrix{T}(I, M, M) .+ T(0.05 / sqrt(M)) .* LowerTriangular(randn(rng, T, M, M)))
    y = lik isa GaussianLikelihood ? sin.(vec(sum(X; dims=1)) ./ T(sqrt(d))) .+ T(sqrt(0.3)) .* randn(rng, T, N) :
        lik isa BernoulliLikelihood ? T.(rand(rng, N) .< 0.5) : T.(rand(rng, N) .+ 0.1)
    build(θ) = begin
        k = θ.var * (base ∘ ARDTransform(θ.invl))
        f = c == 0 ? GP(k) : GP(T(c), k)
        q = MvNormal(θ.m, PDMat(Cholesky(LowerTriangular(θ.A))))       # examples/a-regression/script.jl:110-111
        fz = f(ColVecs(θ.Z), T(1e-5))
        sva = centered ? SparseVariationalApproximation(Centered(), fz, q) : SparseVariationalApproximation(fz, q)
        return f, sva
    end
    return x, y, θ, build, lik
end

@testset "small problems are declined below the measured crossover" begin
    withenv("SVGP_OFFLOAD_MIN_WORK" => nothing) do
        @test !MI.worth_offloading(100, 20, 1) && !MI.worth_offloading(100, 20, 1; grad=true)   # examples/a-regression minibatch
        @test MI.worth_offloading(10_000, 20, 1) && MI.worth_offloading(1000, 32, 1)
        rng = MersenneTwister(1)
        x, y, θ, build, lik = problem(rng, Float64; N=100, M=20, d=1)
        f, sva = build(θ)
        @test ApproximateGPs.MI355XHooks.try_elbo(sva, LatentGP(f, lik, 1e-18)(x), y, 100, GPLikelihoods.DefaultExpectationMethod()) === nothing
    end
end

@testset "elbo / approx_lml: device == reference body ($T, centered = $cen, $(nameof(typeof(lik))))" for
        T in (Float64, Float32), cen in (false, true),
        lik in (GaussianLikelihood(T(0.3)), BernoulliLikelihood(), PoissonLikelihood(), ExponentialLikelihood(), GammaLikelihood(T(2.5)))
    rng = MersenneTwister(1)
    x, y, θ, build, lik = problem(rng, T; centered=cen, lik=lik, base=Matern52Kernel(), c=0.2)
    lik isa PoissonLikelihood && (y = T.(floor.(3 .* y)))
    f, sva = build(θ)
    lfx = LatentGP(f, lik, 1e-18)(x)
    rtol = T === Float64 ? 1e-8 : 1e-4
    @test on(() -> elbo(sva, lfx, y; num_data=1234)) ≈ off(() -> elbo(sva, lfx, y; num_data=1234)) rtol = rtol
    @test on(() -> approx_lml(sva, lfx, y)) ≈ off(() -> approx_lml(sva, lfx, y)) rtol = rtol
    lik isa GaussianLikelihood &&
        @test on(() -> elbo(sva, f(x, T(0.3)), y)) ≈ off(() -> elbo(sva, f(x, T(0.3)), y)) rtol = rtol   # SVA:307-317 wrapper
    @test on(() -> elbo(sva, lfx, y; quadrature=GaussHermiteExpectation(13))) ≈
          off(() -> elbo(sva, lfx, y; quadrature=GaussHermiteExpectation(13))) rtol = rtol
end

@testset "other links and host-evaluated likelihoods ($T)" for T in (Float64, Float32)
    rng = MersenneTwister(5)
    rtol = T === Float64 ? 1e-8 : 1e-4
    x, y, θ, build, _ = problem(rng, T; lik=BernoulliLikelihood(), base=Matern32Kernel())
    f, sva = build(θ)
    # Bool labels (examples/b-classification/script.jl:58) and the normcdf link (examples/c-comparisons/script.jl:33-34): code 5
    for lik in (BernoulliLikelihood(), BernoulliLikelihood(normcdf)), yy in (y, y .> 0.5)
        lfx = LatentGP(f, lik, 1e-18)(x)
        @test on(() -> elbo(sva, lfx, yy; num_data=999)) ≈ off(() -> elbo(sva, lfx, yy; num_data=999)) rtol = rtol
    end
    # a likelihood the ABI does not enumerate (Poisson with a non-exp link): svgp_marginals -> GPLikelihoods here -> value;
    # under Zygote: rrule_via_ad of expected_loglikelihood -> svgp_elbo_grad_ext
    lik = PoissonLikelihood(x -> log1p(exp(x)))
    yc = floor.(3 .* rand(rng, T, length(y)))
    @test MI.unpack_lik(lik) === nothing && MI.ext_ok(lik)
    lfx = LatentGP(f, lik, 1e-18)(x)
    @test on(() -> elbo(sva, lfx, yc; num_data=999)) ≈ off(() -> elbo(sva, lfx, yc; num_data=999)) rtol = rtol
    loss(θ) = (fs = build(θ); -elbo(fs[2], LatentGP(fs[1], lik, 1e-18)(x), yc; num_data=2000))
    gd = on(() -> Zygote.gradient(loss, θ)[1]); gr = off(() -> Zygote.gradient(loss, θ)[1])
    tol = T === Float64 ? 1e-6 : 3e-3
    for k in (:var, :invl, :Z, :m)
        @test maximum(abs.(getfield(gd, k) .- getfield(gr, k))) <= tol * max(maximum(abs.(getfield(gr, k))), 1e-9)
    end
end

@testset "Zygote through elbo: device rrule == Zygote on the reference body ($T, centered = $cen)" for T in (Float64, Float32), cen in (false, true)
    rng = MersenneTwister(2)
    for lik in (GaussianLikelihood(T(0.3)), BernoulliLikelihood())
        x, y, θ, build, lik = problem(rng, T; centered=cen, lik=lik)
        loss(θ) = begin
            f, sva = build(θ)
            -elbo(sva, LatentGP(f, lik, 1e-18)(x), y; num_data=2000)       # the shape of examples/b-classification/script.jl:132-142
        end
        gd = on(() -> Zygote.gradient(loss, θ)[1])
        gr = off(() -> Zygote.gradient(loss, θ)[1])
        tol = T === Float64 ? 1e-6 : 3e-3
        for k in (:var, :invl, :Z, :m)
            @test maximum(abs.(getfield(gd, k) .- getfield(gr, k))) <= tol * max(maximum(abs.(getfield(gr, k))), 1e-9)
        end
        @test maximum(abs.(LowerTriangular(gd.A) .- LowerTriangular(gr.A))) <= tol * maximum(abs.(LowerTriangular(gr.A)))
    end
    # the FiniteGP method with a trainable noise (examples/a-regression/script.jl:136-141): gradient w.r.t. σ² through lik.σ²
    x, y, θ, build, _ = problem(rng, T; centered=cen)
    lossn(σ²) = (fs = build(θ); -elbo(fs[2], fs[1](x, σ²), y))
    @test on(() -> Zygote.gradient(lossn, T(0.3))[1]) ≈ off(() -> Zygote.gradient(lossn, T(0.3))[1]) rtol = (T === Float64 ? 1e-6 : 3e-3)
end

@testset "posterior and the predictive API (SVA:208-264), unchanged call sites ($T, centered = $cen)" for T in (Float64, Float32), cen in (false, true)
    rng = MersenneTwister(3)
    x, y, θ, build, _ = problem(rng, T; centered=cen, base=Matern32Kernel())
    f, sva = build(θ)
    xs, xt = ColVecs(x.X[:, 1:40]), ColVecs(x.X[:, 41:70])
    pd, pr = on(() -> posterior(sva)), off(() -> posterior(sva))
    tol = T === Float64 ? 1e-8 : 2e-4
    @test pd.data.Kuu.L ≈ pr.data.Kuu.L rtol = tol
    @test pd.data.α ≈ pr.data.α rtol = 100tol
    @test Matrix(pd.data.B) ≈ Matrix(pr.data.B) rtol = 10tol
    for fn in (mean, var, cov)
        @test on(() -> fn(pr, xs)) ≈ off(() -> fn(pr, xs)) rtol = tol atol = tol
    end
    @test all(on(() -> mean_and_var(pr, xs)) .≈ off(() -> mean_and_var(pr, xs)))
    @test all(isapprox.(on(() -> mean_and_cov(pr, xs)), off(() -> mean_and_cov(pr, xs)); rtol=tol, atol=tol))
    @test on(() -> cov(pr, xs, xt)) ≈ off(() -> cov(pr, xs, xt)) rtol = tol atol = tol
    @test on(() -> posterior(sva, f(x, T(0.3)), y)).data.α ≈ pr.data.α rtol = 100tol      # 3-argument form SVA:189-201
end

@testset "reference error behaviour is preserved" begin
    rng = MersenneTwister(4)
    x, y, θ, build, lik = problem(rng, Float64)
    f, sva = build(θ)
    g = GP(SqExponentialKernel())
    @test_throws ArgumentError elbo(sva, LatentGP(g, lik, 1e-18)(x), y)                       # SVA:347-351, raised before the hook
    @test_throws ErrorException elbo(sva, f(x, Diagonal(rand(length(y)))), y)                 # SVA:319-327
    bad = SparseVariationalApproximation(f(ColVecs(θ.Z), -1.0), sva.q)                        # Kuu - I: not positive definite
    @test_throws PosDefException elbo(bad, LatentGP(f, lik, 1e-18)(x), y)
    @test_throws MethodError elbo(sva, LatentGP(f, BernoulliLikelihood(), 1e-18)(x), Float64.(y .> 0);
                                  quadrature=GPLikelihoods.AnalyticExpectation())           # no closed form: the hook declines
    # unsupported pieces fall through to the reference body, silently and correctly
    k2 = θ.var * (SqExponentialKernel() ∘ LinearTransform(Matrix(1.0I, 3, 3)))
    f2 = GP(k2); sva2 = SparseVariationalApproximation(f2(ColVecs(θ.Z), 1e-5), sva.q)
    @test on(() -> elbo(sva2, LatentGP(f2, lik, 1e-18)(x), y)) == off(() -> elbo(sva2, LatentGP(f2, lik, 1e-18)(x), y))
end
