# SVGPMI355X.jl — the reference-side binding of libsvgp_mi355x.so (include/svgp_mi355x.h).
#
# What a maintainer of ApproximateGPs.jl adds so that `elbo` / `approx_lml` / `posterior` and the predictive API of a
# `SparseVariationalApproximation` run on an MI355X WITHOUT any change at the call sites — including under Zygote
# (examples/a-regression/script.jl:136-141,188-194, examples/b-classification/script.jl:132-142,
# test/SparseVariationalApproximationModule.jl:163-175) and for `mean / var / cov / mean_and_var / mean_and_cov /
# cov(x, y)` (SVA:208-264):
#
#   1. src/mi355x_hooks.jl (the hook functions with their do-nothing fallbacks), included before the SVA module,
#   2. this file, included after it (it needs the SVA types), and
#   3. the one-line hooks of integration/julia/ApproximateGPs_hooks.patch at the top of the reference methods
#      (`r = MI355XHooks.try_elbo(...); r === nothing || return r`).
#
# A hook returns `nothing` whenever the library is absent, switched off (`SVGPMI355X.enable!(false)`) or the model is
# outside what the library supports (status SVGP_UNSUPPORTED, exotic kernels / transforms / likelihoods / element types),
# and the reference's own body then runs unchanged.  No method is shadowed, nothing is `invoke`d.
#
# AD: `try_elbo` carries a `ChainRulesCore.rrule` whose pullback returns STRUCTURAL tangents for `sva` (kernel variance,
# inverse lengthscales, inducing inputs, constant mean, mean(q), the Cholesky factor of cov(q)) and for the likelihood
# parameter, from ONE svgp_elbo_grad call.  When the hook declines, its rrule returns `nothing` with zero tangents and
# Zygote differentiates the reference body as before.  Gradients with respect to the DATA inputs `x`, `y` and the jitter
# `fz.Σy` are not computed by the library (zero tangents): the reference's callers never ask for them.
#
# NOT EXECUTED IN THIS REPOSITORY: the build image has no Julia.  The identical C symbols, struct layouts and status
# conventions are exercised by the Python ctypes mirror (approximategps.jl_amd/approxgp/_ffi.py) that tests/ call;
# tests/test_abi_cpu.py pins the struct sizes / offsets asserted in integration/julia/test/runtests.jl.
#
# File:line references are to the reference repository (SVA = src/SparseVariationalApproximationModule.jl).
module SVGPMI355X

using AbstractGPs, KernelFunctions, GPLikelihoods, LinearAlgebra, Distributions
using ChainRulesCore
using FillArrays: Fill
using PDMats: PDMat, ScalMat
import Libdl

# `..` = ApproximateGPs: this file is included from src/ApproximateGPs.jl after mi355x_hooks.jl and the SVA module
using ..MI355XHooks
using ..SparseVariationalApproximationModule: SparseVariationalApproximation, Centered, NonCentered
using ..ApproximateGPs: _chol_lower, _chol_cov, ApproxPosteriorGP

const lib = get(ENV, "SVGP_MI355X_LIB", "libsvgp_mi355x.so")
const FT = Union{Float32,Float64}
const ENABLED = Ref(true)
"Switch every hook off (they return `nothing`) or back on; the reference's pure-Julia bodies then run."
enable!(on::Bool) = (ENABLED[] = on)

# ---------------------------------------------------------------------------------------------------------
# C structs, field for field (include/svgp_mi355x.h; sizes 104 / 64 / 56 bytes, checked in test/runtests.jl)
# ---------------------------------------------------------------------------------------------------------
struct ModelDesc                      # svgp_model_desc
    dtype::Int32; kernel::Int32; parametrization::Int32; likelihood::Int32
    quadrature_n::Int32; layout_z::Int32; neg_var_policy::Int32; d::Int32
    M::Int64; variance::Float64; inv_lengthscale::Ptr{Float64}
    mean_const::Float64; jitter::Float64; lik_sigma2::Float64
    z::Ptr{Cvoid}; m::Ptr{Cvoid}; Lq::Ptr{Cvoid}
end

mutable struct Terms                  # svgp_terms
    elbo::Float64; expectation::Float64; kl::Float64; scale::Float64; logdet_kuu::Float64
    n_points::Int64; n_neg_var::Int64; chol_info::Int32; reserved::Int32
    Terms() = new(0, 0, 0, 0, 0, 0, 0, 0, 0)
end

mutable struct Grads                  # svgp_grads
    variance::Float64; lik_sigma2::Float64; mean_const::Float64
    inv_lengthscale::Ptr{Float64}; z::Ptr{Cvoid}; m::Ptr{Cvoid}; Lq::Ptr{Cvoid}
end

# ---------------------------------------------------------------------------------------------------------
# context (one per process and GPU) and status -> exception (SURVEY §8b)
# ---------------------------------------------------------------------------------------------------------
const CTX = Ref{Ptr{Cvoid}}(C_NULL)
const AVAILABLE = Ref{Union{Nothing,Bool}}(nothing)

"true when the library can be loaded and sees a GPU (probed once); otherwise every hook declines."
function available()
    if AVAILABLE[] === nothing
        ok = Libdl.dlopen(lib; throw_error=false) !== nothing
        ok = ok && ccall((:svgp_device_count, lib), Int32, ()) > 0
        AVAILABLE[] = ok
    end
    return AVAILABLE[]::Bool
end

function ctx()
    if CTX[] == C_NULL
        dev = parse(Int32, get(ENV, "SVGP_MI355X_DEVICE", "0"))
        st = ccall((:svgp_ctx_create, lib), Int32, (Int32, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), dev, C_NULL, CTX)
        st == 0 || error("svgp_ctx_create failed with status $st")
        # finalizers of DeviceData / DeviceModel run AFTER atexit hooks: the hook clears CTX[] and they skip a dead context
        atexit() do
            c = CTX[]
            CTX[] = C_NULL
            c == C_NULL || ccall((:svgp_ctx_destroy, lib), Int32, (Ptr{Cvoid},), c)
        end
    end
    return CTX[]
end
"free a handle on the live context; after the atexit hook destroyed the context the device memory is gone with it"
free_on_ctx(sym::Symbol, h::Ptr{Cvoid}) = (CTX[] == C_NULL || h == C_NULL) ? Int32(0) :
    (sym === :data ? ccall((:svgp_data_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), CTX[], h) :
                     ccall((:svgp_model_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), CTX[], h))
"""
`worth_offloading(n, M, d; grad=false)`: the library's own small-problem rule (`svgp_offload_advice`, include/svgp_mi355x.h).
A call has a floor of 160-250 us (forward) / ~600 us (value and gradient) whatever the size (profiles/round3/small_problems.md);
the reference's own examples (N = 10 000, M = 20, minibatch 100) sit below the crossover, so every hook declines there and the
pure-Julia method runs.  `ENV["SVGP_OFFLOAD_MIN_WORK"] = "0"` offloads everything.
"""
worth_offloading(n::Integer, M::Integer, d::Integer; grad::Bool=false) =
    ccall((:svgp_offload_advice, lib), Int32, (Int64, Int64, Int32, Int32, Int32), n, M, d, 0, grad ? 1 : 0) == 1

"size of the library communicator on the context (1 without one)"
function comm_world()
    w = Ref{Int32}(1)
    ccall((:svgp_ctx_comm_info, lib), Int32, (Ptr{Cvoid}, Ref{Int32}, Ptr{Int32}), ctx(), w, C_NULL)
    return Int(w[])
end
last_error() = unsafe_string(ccall((:svgp_last_error, lib), Cstring, (Ptr{Cvoid},), ctx()))

struct Unsupported <: Exception end   # internal: "decline, the pure-Julia body runs"

function check(st::Integer, terms::Union{Terms,Nothing}=nothing)
    st == 0 && return nothing
    st == 4 && throw(Unsupported())
    st == 1 && throw(ArgumentError(last_error()))
    st == 2 && throw(PosDefException(terms === nothing ? 1 : Int(terms.chol_info)))   # cholesky(Kuu), src/utils.jl:17
    st == 3 && throw(DomainError(-1.0, "sqrt of a negative predictive variance (marginals, SVA:354)"))
    st == 7 && throw(OutOfMemoryError())
    return error("libsvgp_mi355x status $st: " * last_error())
end

# ---------------------------------------------------------------------------------------------------------
# unpacking the reference's objects into the POD description; anything unknown -> Unsupported -> the hook declines
# ---------------------------------------------------------------------------------------------------------
kfamily(::SqExponentialKernel) = Int32(0)
kfamily(::Matern32Kernel) = Int32(1)
kfamily(::Matern52Kernel) = Int32(2)
kfamily(::Any) = nothing

# variance * (Base ∘ ScaleTransform(1/l) | ARDTransform(1 ./ l))  [KernelFunctions: ScaledKernel(kernel, σ²::Vector),
# TransformedKernel(kernel, transform), ScaleTransform(s::Vector), ARDTransform(v::Vector)]
function unpack_kernel(k, d)
    σ² = 1.0
    if k isa ScaledKernel
        σ², k = Float64(only(k.σ²)), k.kernel
    end
    invl = ones(Float64, d)
    if k isa TransformedKernel
        t = k.transform
        if t isa ScaleTransform
            invl = fill(Float64(only(t.s)), d)
        elseif t isa ARDTransform
            length(t.v) == d || return nothing
            invl = collect(Float64, t.v)
        else
            return nothing
        end
        k = k.kernel
    end
    fam = kfamily(k)
    fam === nothing && return nothing
    return fam, σ², invl
end

unpack_mean(::AbstractGPs.ZeroMean) = 0.0
unpack_mean(m::AbstractGPs.ConstMean) = Float64(only(m.c))
unpack_mean(::Any) = nothing

# (code, parameter, has a closed-form expectation [GPLikelihoods AnalyticExpectation])
unpack_lik(l::GaussianLikelihood) = (Int32(0), Float64(only(l.σ²)), true)
unpack_lik(::BernoulliLikelihood{<:LogisticLink}) = (Int32(1), 1.0, false)
unpack_lik(::BernoulliLikelihood{<:NormalCDFLink}) = (Int32(5), 1.0, false)   # probit: y ~ Bernoulli(normcdf(f))
unpack_lik(::PoissonLikelihood{<:ExpLink}) = (Int32(2), 1.0, true)
unpack_lik(::ExponentialLikelihood{<:ExpLink}) = (Int32(3), 1.0, true)      # Exponential(scale = exp f), oracle/CONVENTIONS.md
unpack_lik(l::GammaLikelihood{<:Any,<:ExpLink}) = (Int32(4), Float64(only(l.α)), true)   # shape in the parameter slot
unpack_lik(::Any) = nothing

layout(x::ColVecs) = (Int32(0), x.X, size(x.X, 1))
layout(x::RowVecs) = (Int32(1), x.X, size(x.X, 2))
layout(x::AbstractVector{<:Real}) = (Int32(2), x, 1)
layout(::Any) = nothing

# quadrature -> quadrature_n.  AnalyticExpectation on a likelihood WITHOUT a closed form is a MethodError in the reference
# (GPLikelihoods defines no such method): decline, so that the reference body raises exactly that.
function quad_n(q, closed_form::Bool)
    q isa GPLikelihoods.DefaultExpectationMethod && return Int32(0)
    q isa GPLikelihoods.GaussHermiteExpectation && return Int32(length(q.xs))
    q isa GPLikelihoods.AnalyticExpectation && return closed_form ? Int32(0) : nothing
    return nothing
end

isotropic_jitter(Σ::Diagonal{<:Real,<:Fill}) = Float64(Σ[1])          # the types SVA:309 accepts; SVA:314 reads Σy[1]
isotropic_jitter(Σ::ScalMat) = Float64(Σ[1])
isotropic_jitter(::Any) = nothing

"Host arrays a ModelDesc points into; must outlive every ccall that receives the desc (GC.@preserve)."
struct Packed{T}
    invl::Vector{Float64}; Z::Array{T}; m::Vector{T}; Lq::Matrix{T}
    desc::ModelDesc
    ext::Bool        # likelihood not enumerated by the ABI: evaluated HERE on the device marginals (svgp_marginals / svgp_elbo_grad_ext)
end

# Any single-latent GPLikelihoods likelihood can take the host-evaluated route; the two multi-latent ones cannot.
ext_ok(l) = l isa GPLikelihoods.AbstractLikelihood && !(l isa CategoricalLikelihood) && !(l isa HeteroscedasticGaussianLikelihood)

compute_type(::Type{Float32}) = Float32
compute_type(::Type{Float64}) = Float64
compute_type(::Type) = nothing

function pack(sva::SparseVariationalApproximation{P}, lik, quadrature, ::Type{T}) where {P,T<:FT}
    lay = layout(sva.fz.x)
    lay === nothing && throw(Unsupported())
    lz, Z, d = lay
    (1 <= d <= 64) || throw(Unsupported())       # SVGP_MAX_D
    sva.fz.f isa AbstractGPs.GP || throw(Unsupported())
    sva.q isa MvNormal || throw(Unsupported())
    ku = unpack_kernel(sva.fz.f.kernel, d)
    c = unpack_mean(sva.fz.f.mean)
    lk = unpack_lik(lik)
    ext = lk === nothing
    ext && !ext_ok(lik) && throw(Unsupported())
    ext && (lk = (Int32(0), 1.0, true))        # the descriptor's likelihood slot is unused on the host-evaluated route
    jit = isotropic_jitter(sva.fz.Σy)
    (ku === nothing || c === nothing || jit === nothing) && throw(Unsupported())
    qn = ext ? Int32(0) : quad_n(quadrature, lk[3])
    qn === nothing && throw(Unsupported())
    fam, σ², invl = ku
    m = Vector{T}(mean(sva.q))
    Lq = Matrix{T}(_chol_lower(_chol_cov(sva.q)))          # src/utils.jl:15-18
    Zd = Array{T}(Z)
    desc = ModelDesc(T === Float64 ? 0 : 1, fam, P === Centered ? 1 : 0, lk[1], qn, lz, 0, d, length(m), σ²,
                     pointer(invl), c, jit, lk[2], pointer(Zd), pointer(m), pointer(Lq))
    return Packed{T}(invl, Zd, m, Lq, desc, ext)
end

# ---------------------------------------------------------------------------------------------------------
# elbo(sva, lfx, y; num_data, quadrature)   hook for SVA:340-360   (the FiniteGP method SVA:307-317 and approx_lml
# SVA:276-280 forward to that method unchanged, so they need no hook of their own)
# ---------------------------------------------------------------------------------------------------------
"""
    try_elbo(sva, lfx, y, num_data, quadrature) -> Union{Nothing,Float64}

The ELBO of SVA:340-360 on the device, or `nothing` to let the reference body run.  Called AFTER the prior-identity
check (SVA:347-351), which stays in Julia.  (Method of `MI355XHooks.try_elbo`.)
"""
function MI355XHooks.try_elbo(sva::SparseVariationalApproximation, lfx, y::AbstractVector, num_data, quadrature)
    r = elbo_and_grads(sva, lfx, y, num_data, quadrature, false)
    return r === nothing ? nothing : r[1]
end

# The likelihood-dependent step of SVA:355 kept in Julia: E = expected_loglikelihood(quadrature, lik, marginals, y) on the
# device's marginals and, under AD (`config` = the caller's RuleConfig), its pullback w.r.t. (lik, mu, v).
function host_expectation(config, quadrature, lik, μ::Vector{Float64}, v::Vector{Float64}, y)
    E(l, a, b) = expected_loglikelihood(quadrature, l, Normal.(a, sqrt.(b)), y)      # SVA:354-355
    config === nothing && return sum(E(lik, μ, v)), nothing
    val, back = rrule_via_ad(config, E, lik, μ, v)
    _, Δlik, gμ, gv = back(one(val))
    return val, (Δlik, collect(Float64, unthunk(gμ)), collect(Float64, unthunk(gv)))
end

# value (and, if `want`, the raw gradient blocks) or `nothing`; `config`: the AD rule configuration (host-evaluated route only)
function elbo_and_grads(sva, lfx, y, num_data, quadrature, want::Bool, config=nothing)
    (ENABLED[] && available()) || return nothing
    lay = layout(lfx.fx.x)
    lay === nothing && return nothing
    lx, X, dx = lay
    # the compute type is the inputs' / parameters' float type; the observations only have to convert to it (Bool labels
    # of a Bernoulli likelihood and Int counts of a Poisson one - examples/b-classification/script.jl:58 - are the normal case)
    eltype(y) <: Real || return nothing
    T = compute_type(promote_type(eltype(X), eltype(mean(sva.q))))
    T === nothing && return nothing
    local p
    try
        p = pack(sva, lfx.lik, quadrature, T)
    catch e
        e isa Unsupported || rethrow()
        return nothing
    end
    dx == p.desc.d || return nothing
    # small problems: the Julia body is faster.  NOT under a library communicator: svgp_elbo_host / svgp_elbo_grad are collective
    # there, and a rule decided on this rank's own shard length (or on a rank-local ENV value) would let one rank run the Julia
    # body - and return a shard-local ELBO - while its peers wait in ncclAllReduce (uneven shards, a short last shard).  Every
    # rank of a data-parallel job therefore offloads, whatever its shard's size.
    (comm_world() > 1 || worth_offloading(length(y), length(p.m), Int(p.desc.d); grad=want)) || return nothing
    Xd, yd = Array{T}(X), Vector{T}(y)
    n = length(yd)
    out, terms = Ref{Float64}(), Terms()
    d, M = Int(p.desc.d), length(p.m)
    gl, gz, gm, gLq = zeros(Float64, d), similar(p.Z), similar(p.m), similar(p.Lq)
    g = Grads(0, 0, 0, pointer(gl), pointer(gz), pointer(gm), pointer(gLq))
    st = Int32(0)
    Δlik = nothing
    GC.@preserve p Xd yd gl gz gm gLq begin
        if p.ext
            # host-evaluated likelihood: marginals from the device, SVA:355 here, the backward pass on the device again
            want && config === nothing && return nothing
            # data-parallel contexts: the forward value of this route (svgp_marginals + svgp_prior_kl) is this rank's shard only,
            # while svgp_elbo_grad_ext is collective - value and gradient would disagree; decline (the Julia body runs)
            comm_world() > 1 && return nothing
            hm, hd = Ref{Ptr{Cvoid}}(C_NULL), Ref{Ptr{Cvoid}}(C_NULL)
            μ, v = zeros(Float64, n), zeros(Float64, n)
            try   # the user's likelihood / AD may throw: the handles are freed on every path
            st = ccall((:svgp_model_create, lib), Int32, (Ptr{Cvoid}, Ref{ModelDesc}, Ptr{Ptr{Cvoid}}), ctx(), p.desc, hm)
            if st == 0
                st = ccall((:svgp_data_upload, lib), Int32,
                           (Ptr{Cvoid}, Int32, Int32, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
                           ctx(), p.desc.dtype, lx, p.desc.d, n, Xd, C_NULL, hd)
            end
            if st == 0
                st = ccall((:svgp_marginals, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}),
                           ctx(), hm[], hd[], 0, n, μ, v)
            end
            if st == 0
                sumE, pb = host_expectation(want ? config : nothing, quadrature, lfx.lik, μ, v, y)
                if want
                    Δlik, gμ, gv = pb
                    st = ccall((:svgp_elbo_grad_ext, lib), Int32,
                               (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float64, Float64, Ptr{Float64}, Ptr{Float64},
                                Ref{Float64}, Ref{Terms}, Ref{Grads}),
                               ctx(), hm[], hd[], 0, n, Float64(num_data), Float64(sumE), gμ, gv, out, terms, g)
                else
                    kl = Ref{Float64}()
                    st = ccall((:svgp_prior_kl, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}), ctx(), hm[], kl, C_NULL)
                    out[] = sumE * Float64(num_data) / n - kl[]                       # SVA:357-359
                end
            end
            finally
                free_on_ctx(:data, hd[])
                free_on_ctx(:model, hm[])
            end
        elseif !want
            # one-shot entry point: uploads x, y, evaluates, frees (resident handles below avoid the upload in loops)
            st = ccall((:svgp_elbo_host, lib), Int32,
                       (Ptr{Cvoid}, Ref{ModelDesc}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ref{Float64}, Ref{Terms}),
                       ctx(), p.desc, lx, n, Xd, yd, Float64(num_data), out, terms)
        else
            hm, hd = Ref{Ptr{Cvoid}}(C_NULL), Ref{Ptr{Cvoid}}(C_NULL)
            st = ccall((:svgp_model_create, lib), Int32, (Ptr{Cvoid}, Ref{ModelDesc}, Ptr{Ptr{Cvoid}}), ctx(), p.desc, hm)
            if st == 0
                st = ccall((:svgp_data_upload, lib), Int32,
                           (Ptr{Cvoid}, Int32, Int32, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
                           ctx(), p.desc.dtype, lx, p.desc.d, n, Xd, yd, hd)
            end
            if st == 0
                st = ccall((:svgp_elbo_grad, lib), Int32,
                           (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float64, Ref{Float64}, Ref{Terms}, Ref{Grads}),
                           ctx(), hm[], hd[], 0, n, Float64(num_data), out, terms, g)
            end
            ccall((:svgp_data_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), hd[])
            ccall((:svgp_model_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), hm[])
        end
    end
    st == 4 && return nothing            # SVGP_UNSUPPORTED: decline
    check(st, terms)
    # `Δlik`: the likelihood's own tangent from the host pullback (host-evaluated route), to be scaled by num_data / n
    return out[], (σ²=g.variance, invl=gl, z=gz, m=gm, Lq=gLq, lik=g.lik_sigma2, c=g.mean_const, Δlik=Δlik, scale=Float64(num_data) / n)
end

# ---- structural tangents -------------------------------------------------------------------------------------------
# Mirror images of unpack_kernel / unpack_mean / layout / _chol_cov: each builds the Tangent of exactly the object it unpacked.
function kernel_tangent(k, gσ², ginvl)
    if k isa ScaledKernel
        return Tangent{typeof(k)}(; kernel=inner_kernel_tangent(k.kernel, ginvl), σ²=[oftype(only(k.σ²), gσ²)])
    end
    return inner_kernel_tangent(k, ginvl)
end
function inner_kernel_tangent(k, ginvl)
    k isa TransformedKernel || return NoTangent()           # bare base kernel: no parameters
    t = k.transform
    tt = t isa ScaleTransform ? Tangent{typeof(t)}(; s=[oftype(only(t.s), sum(ginvl))]) :   # invl = fill(s, d)
                                Tangent{typeof(t)}(; v=convert(typeof(t.v), ginvl))
    return Tangent{typeof(k)}(; kernel=NoTangent(), transform=tt)
end
mean_tangent(m::AbstractGPs.ConstMean, gc) = Tangent{typeof(m)}(; c=m.c isa AbstractArray ? [oftype(only(m.c), gc)] : oftype(m.c, gc))
mean_tangent(::Any, gc) = NoTangent()
inputs_tangent(x::Union{ColVecs,RowVecs}, gz) = Tangent{typeof(x)}(; X=convert(typeof(x.X), gz))
inputs_tangent(x::AbstractVector{<:Real}, gz) = convert(typeof(x), vec(gz))

# q = MvNormal(μ, Σ::PDMat): _chol_cov(q) = cholesky(q.Σ) = q.Σ.chol (src/utils.jl:18), so the factor's gradient belongs to
# `Σ.chol.factors` (lower storage: as is; upper storage: transposed); Σ.mat is not read by the NonCentered path.  Other Σ
# types (dense matrix, ScalMat, ...) -> the hook's rrule declines and Zygote differentiates the reference body.
function q_tangent(q::MvNormal, gm, gLq)
    Σ = q.Σ
    Σ isa PDMat || return nothing
    C = Σ.chol
    gf = C.uplo == 'L' ? LowerTriangular(gLq) : UpperTriangular(permutedims(gLq))
    ΔC = Tangent{typeof(C)}(; factors=convert(typeof(C.factors), Matrix(gf)))
    return Tangent{typeof(q)}(; μ=convert(typeof(q.μ), gm), Σ=Tangent{typeof(Σ)}(; chol=ΔC))
end

function lik_tangent(l, glik)
    l isa GaussianLikelihood && return Tangent{typeof(l)}(; σ²=[oftype(only(l.σ²), glik)])
    l isa GammaLikelihood && return Tangent{typeof(l)}(; α=l.α isa AbstractArray ? [oftype(only(l.α), glik)] : oftype(l.α, glik))
    return NoTangent()
end

function ChainRulesCore.rrule(config::RuleConfig{>:HasReverseMode}, ::typeof(MI355XHooks.try_elbo), sva::SparseVariationalApproximation, lfx,
                              y::AbstractVector, num_data, quadrature)
    zero5 = (NoTangent(), NoTangent(), NoTangent(), NoTangent(), NoTangent(), NoTangent())
    decline = (nothing, _ -> zero5)
    sva.q isa MvNormal && sva.q.Σ isa PDMat || return decline       # the tangent of cov(q) is only defined for a stored factor
    r = elbo_and_grads(sva, lfx, y, num_data, quadrature, true, config)
    r === nothing && return decline
    val, g = r
    function try_elbo_pullback(Δ)
        Δ = unthunk(Δ)
        s(a) = Δ .* a
        f = sva.fz.f
        Δf = Tangent{typeof(f)}(; mean=mean_tangent(f.mean, Δ * g.c), kernel=kernel_tangent(f.kernel, Δ * g.σ², s(g.invl)))
        Δfz = Tangent{typeof(sva.fz)}(; f=Δf, x=inputs_tangent(sva.fz.x, s(g.z)))      # Σy (jitter): not differentiated
        Δsva = Tangent{typeof(sva)}(; fz=Δfz, q=q_tangent(sva.q, s(g.m), s(g.Lq)))
        # the prior of lfx is the SAME object (SVA:347-351), its tangent is already on sva.fz.f; data inputs: none
        Δlfx = Tangent{typeof(lfx)}(; lik=g.Δlik === nothing ? lik_tangent(lfx.lik, Δ * g.lik) : (Δ * g.scale) * g.Δlik)
        return (NoTangent(), Δsva, Δlfx, NoTangent(), NoTangent(), NoTangent())
    end
    return val, try_elbo_pullback
end

# ---------------------------------------------------------------------------------------------------------
# posterior(sva)   hook for SVA:115-136 (Centered) and SVA:160-187 (NonCentered): the same
# ApproxPosteriorGP(sva, fz.f, (Kuu, B, α)) the reference builds, with the factors computed on the device.
# ---------------------------------------------------------------------------------------------------------
function MI355XHooks.try_posterior(sva::SparseVariationalApproximation{P}) where {P}
    (ENABLED[] && available()) || return nothing
    T = compute_type(eltype(mean(sva.q)))
    T === nothing && return nothing
    local p
    try
        p = pack(sva, GaussianLikelihood(1.0), GPLikelihoods.DefaultExpectationMethod(), T)
    catch e
        e isa Unsupported || rethrow()
        return nothing
    end
    M = length(p.m)
    worth_offloading(0, M, Int(p.desc.d)) || return nothing    # posterior(sva) is M-sized work only: M^3 / 3 against the call's floor
    Lk, α, B = Matrix{T}(undef, M, M), Vector{T}(undef, M), Matrix{T}(undef, M, M)
    hm = Ref{Ptr{Cvoid}}(C_NULL)
    st = Int32(0)
    GC.@preserve p Lk α B begin
        st = ccall((:svgp_model_create, lib), Int32, (Ptr{Cvoid}, Ref{ModelDesc}, Ptr{Ptr{Cvoid}}), ctx(), p.desc, hm)
        if st == 0
            st = ccall((:svgp_posterior, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                       ctx(), hm[], Lk, α, B)
        end
        ccall((:svgp_model_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), hm[])
    end
    st == 4 && return nothing
    check(st)        # SVGP_NOT_POSDEF -> PosDefException, as cholesky(Kuu) in the reference (the order is not returned here: 1)
    data = (Kuu=Cholesky(Lk, 'L', 0), B=LowerTriangular(B), α=α)
    return ApproxPosteriorGP(sva, sva.fz.f, data)
end
ChainRulesCore.@non_differentiable available()
# differentiating THROUGH posterior(sva) (nobody in the reference does: elbo has its own rule) uses the Julia body
ChainRulesCore.rrule(::typeof(MI355XHooks.try_posterior), sva::SparseVariationalApproximation) = (nothing, _ -> (NoTangent(), NoTangent()))

# ---------------------------------------------------------------------------------------------------------
# mean / var / cov / mean_and_var / mean_and_cov / cov(x, y)   hooks for SVA:208-264
# ---------------------------------------------------------------------------------------------------------
"`(μ, v, C)` of the approximate posterior at `x` (any of the three may be skipped), or `nothing` to decline."
function MI355XHooks.try_predict(f::ApproxPosteriorGP, x::AbstractVector; want_mean::Bool=true, want_var::Bool=true, want_cov::Bool=false)
    (ENABLED[] && available()) || return nothing
    sva = f.approx
    sva isa SparseVariationalApproximation || return nothing
    lay = layout(x)
    lay === nothing && return nothing
    lx, X, dx = lay
    T = compute_type(eltype(X))
    T === nothing && return nothing
    local p
    try
        p = pack(sva, GaussianLikelihood(1.0), GPLikelihoods.DefaultExpectationMethod(), T)
    catch e
        e isa Unsupported || rethrow()
        return nothing
    end
    dx == p.desc.d || return nothing
    n = length(x)
    worth_offloading(n, length(p.m), Int(p.desc.d)) || return nothing
    Xd = Array{T}(X)
    μ = want_mean ? Vector{T}(undef, n) : nothing
    v = want_var ? Vector{T}(undef, n) : nothing
    C = want_cov ? Matrix{T}(undef, n, n) : nothing
    ptr(a) = a === nothing ? C_NULL : pointer(a)
    hm = Ref{Ptr{Cvoid}}(C_NULL)
    st = Int32(0)
    GC.@preserve p Xd μ v C begin
        st = ccall((:svgp_model_create, lib), Int32, (Ptr{Cvoid}, Ref{ModelDesc}, Ptr{Ptr{Cvoid}}), ctx(), p.desc, hm)
        if st == 0
            st = ccall((:svgp_predict, lib), Int32,
                       (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                       ctx(), hm[], lx, n, Xd, ptr(μ), ptr(v), ptr(C))
        end
        ccall((:svgp_model_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), hm[])
    end
    st == 4 && return nothing
    check(st)
    return (μ, v, C)
end

"`cov(f, x, y)` (SVA:255-264) on the device, or `nothing`."
function MI355XHooks.try_cross_cov(f::ApproxPosteriorGP, x::AbstractVector, y::AbstractVector)
    (ENABLED[] && available()) || return nothing
    sva = f.approx
    sva isa SparseVariationalApproximation || return nothing
    lax, lay = layout(x), layout(y)
    (lax === nothing || lay === nothing || lax[1] != lay[1]) && return nothing
    lx, X, dx = lax
    _, Y, dy = lay
    T = compute_type(eltype(X))
    (T === nothing || eltype(Y) !== eltype(X) || dx != dy) && return nothing
    local p
    try
        p = pack(sva, GaussianLikelihood(1.0), GPLikelihoods.DefaultExpectationMethod(), T)
    catch e
        e isa Unsupported || rethrow()
        return nothing
    end
    dx == p.desc.d || return nothing
    nx, ny = length(x), length(y)
    worth_offloading(nx + ny, length(p.m), Int(p.desc.d)) || return nothing
    Xd, Yd = Array{T}(X), Array{T}(Y)
    C = Matrix{T}(undef, nx, ny)
    hm = Ref{Ptr{Cvoid}}(C_NULL)
    st = Int32(0)
    GC.@preserve p Xd Yd C begin
        st = ccall((:svgp_model_create, lib), Int32, (Ptr{Cvoid}, Ref{ModelDesc}, Ptr{Ptr{Cvoid}}), ctx(), p.desc, hm)
        if st == 0
            st = ccall((:svgp_predict_cross_cov, lib), Int32,
                       (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                       ctx(), hm[], lx, nx, Xd, ny, Yd, C)
        end
        ccall((:svgp_model_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), hm[])
    end
    st == 4 && return nothing
    check(st)
    return C
end
# predictions inside a differentiated loss: decline, Zygote differentiates the reference bodies (SVA:208-264)
ChainRulesCore.rrule(::typeof(MI355XHooks.try_predict), f::ApproxPosteriorGP, x::AbstractVector; kw...) = (nothing, _ -> (NoTangent(), NoTangent(), NoTangent()))
ChainRulesCore.rrule(::typeof(MI355XHooks.try_cross_cov), f::ApproxPosteriorGP, x::AbstractVector, y::AbstractVector) = (nothing, _ -> (NoTangent(), NoTangent(), NoTangent(), NoTangent()))

# ---------------------------------------------------------------------------------------------------------
# resident handles for training loops (optional, explicit): upload x, y once, update the model every step
# ---------------------------------------------------------------------------------------------------------
mutable struct DeviceData
    h::Ptr{Cvoid}; n::Int
end
function DeviceData(x, y::AbstractVector{T}) where {T<:FT}
    lx, X, d = layout(x)
    Xd, yd = Array{T}(X), Vector{T}(y)
    h = Ref{Ptr{Cvoid}}()
    GC.@preserve Xd yd check(ccall((:svgp_data_upload, lib), Int32,
        (Ptr{Cvoid}, Int32, Int32, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
        ctx(), T === Float64 ? 0 : 1, lx, d, length(yd), Xd, yd, h))
    D = DeviceData(h[], length(yd))
    finalizer(D -> free_on_ctx(:data, D.h), D)
    return D
end

mutable struct DeviceModel
    h::Ptr{Cvoid}
end
function DeviceModel(p::Packed)
    h = Ref{Ptr{Cvoid}}()
    GC.@preserve p check(ccall((:svgp_model_create, lib), Int32, (Ptr{Cvoid}, Ref{ModelDesc}, Ptr{Ptr{Cvoid}}), ctx(), p.desc, h))
    M = DeviceModel(h[])
    finalizer(M -> free_on_ctx(:model, M.h), M)
    return M
end
update!(M::DeviceModel, p::Packed) =
    GC.@preserve p check(ccall((:svgp_model_update, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{ModelDesc}), ctx(), M.h, p.desc))

"ELBO of points off+1 : off+len of resident data (a minibatch window), SVA:340-360."
function elbo_resident(M::DeviceModel, D::DeviceData, off::Integer, len::Integer, num_data::Real)
    out, terms = Ref{Float64}(), Terms()
    check(ccall((:svgp_elbo, lib), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float64, Ref{Float64}, Ref{Terms}),
                ctx(), M.h, D.h, off, len, Float64(num_data), out, terms), terms)
    return out[]
end

# ---------------------------------------------------------------------------------------------------------
# multi-GPU from ONE Julia process (svgp_group_*): data sharded over the devices, model replicated, the partial sums of
# SVA:355-359 combined by ONE ncclAllReduce inside the library.  (One process per GPU instead: svgp_comm_unique_id on
# rank 0, the 128 bytes sent with MPI.jl / Distributed.jl, svgp_ctx_attach_comm on every rank; `try_elbo` is then
# collective for the enumerated likelihoods and needs no further change; the host-evaluated-likelihood route declines under a
# communicator - its forward value would be this rank's shard only.)
# ---------------------------------------------------------------------------------------------------------
mutable struct Group
    h::Ptr{Cvoid}; n::Int
    data::Vector{Ptr{Cvoid}}; models::Vector{Ptr{Cvoid}}; shard::Vector{Int64}
end
function Group(devices::Vector{<:Integer})
    h = Ref{Ptr{Cvoid}}()
    ids = Int32.(devices)
    st = ccall((:svgp_group_create, lib), Int32, (Int32, Ptr{Int32}, Ptr{Ptr{Cvoid}}), length(ids), ids, h)
    st == 0 || error("svgp_group_create failed with status $st")
    G = Group(h[], length(ids), Ptr{Cvoid}[], Ptr{Cvoid}[], Int64[])
    finalizer(G) do g
        for i in 1:g.n
            c = ccall((:svgp_group_ctx, lib), Ptr{Cvoid}, (Ptr{Cvoid}, Int32), g.h, i - 1)
            isempty(g.models) || ccall((:svgp_model_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), c, g.models[i])
            isempty(g.data) || ccall((:svgp_data_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), c, g.data[i])
        end
        ccall((:svgp_group_destroy, lib), Int32, (Ptr{Cvoid},), g.h)
    end
    return G
end
group_error(G::Group) = unsafe_string(ccall((:svgp_group_last_error, lib), Cstring, (Ptr{Cvoid},), G.h))
function upload!(G::Group, x, y::AbstractVector{T}) where {T<:FT}
    lx, X, d = layout(x)
    Xd, yd = Array{T}(X), Vector{T}(y)
    G.data = fill(C_NULL, G.n)
    GC.@preserve Xd yd begin
        st = ccall((:svgp_group_data_upload, lib), Int32,
                   (Ptr{Cvoid}, Int32, Int32, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
                   G.h, T === Float64 ? 0 : 1, lx, d, length(yd), Xd, yd, G.data)
        st == 0 || error("svgp_group_data_upload: " * group_error(G))
    end
    base, rem = divrem(length(yd), G.n)
    G.shard = [base + (i <= rem ? 1 : 0) for i in 1:G.n]
    return G
end
function set_model!(G::Group, p::Packed)
    GC.@preserve p begin
        if isempty(G.models)
            G.models = fill(C_NULL, G.n)
            st = ccall((:svgp_group_model_create, lib), Int32, (Ptr{Cvoid}, Ref{ModelDesc}, Ptr{Ptr{Cvoid}}), G.h, p.desc, G.models)
        else
            st = ccall((:svgp_group_model_update, lib), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ref{ModelDesc}), G.h, G.models, p.desc)
        end
        st == 0 || error("svgp_group_model_*: " * group_error(G))
    end
    return G
end
"Global ELBO over every member's window `offs[i]+1 : offs[i]+lens[i]` of its shard."
function elbo(G::Group, num_data::Real; offs=zeros(Int64, G.n), lens=G.shard)
    out, terms = Ref{Float64}(), Terms()
    st = ccall((:svgp_group_elbo, lib), Int32,
               (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}, Ptr{Int64}, Ptr{Int64}, Float64, Ref{Float64}, Ref{Terms}),
               G.h, G.models, G.data, Int64.(offs), Int64.(lens), Float64(num_data), out, terms)
    st == 0 || (st == 2 ? throw(PosDefException(Int(terms.chol_info))) : error("svgp_group_elbo status $st: " * group_error(G)))
    return out[]
end

end # module
