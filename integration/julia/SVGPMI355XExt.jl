# SVGPMI355XExt.jl — the reference-side binding of libsvgp_mi355x.so (include/svgp_mi355x.h).
#
# What a maintainer of ApproximateGPs.jl would add (as a package extension or a `src/` file) so that
# `elbo` / `approx_lml` / `posterior` and the predictive API of a `SparseVariationalApproximation` run on an
# MI355X.  It only ADDS methods; every unsupported case falls through to the existing pure-Julia method.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no Julia.  The identical C symbols, struct layouts
# and status conventions are exercised by the Python ctypes mirror (approximategps.jl_amd/approxgp/_ffi.py),
# which tests/ call; field order and types below are copied from include/svgp_mi355x.h.
#
# File:line references are to the reference repository (SVA = src/SparseVariationalApproximationModule.jl).
module SVGPMI355XExt

using ApproximateGPs, AbstractGPs, KernelFunctions, GPLikelihoods, LinearAlgebra, Distributions
using ChainRulesCore
using FillArrays: Fill
using PDMats: ScalMat
using ApproximateGPs.SparseVariationalApproximationModule:
    SparseVariationalApproximation, Centered, NonCentered
using ApproximateGPs: _chol_lower, _chol_cov

const lib = get(ENV, "SVGP_MI355X_LIB", "libsvgp_mi355x.so")

# ---------------------------------------------------------------------------------------------------------
# C structs (field for field)
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
# context (one per process and GPU) and status -> exception
# ---------------------------------------------------------------------------------------------------------
const CTX = Ref{Ptr{Cvoid}}(C_NULL)
function ctx()
    if CTX[] == C_NULL
        dev = parse(Int32, get(ENV, "SVGP_MI355X_DEVICE", "0"))
        st = ccall((:svgp_ctx_create, lib), Int32, (Int32, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), dev, C_NULL, CTX)
        st == 0 || error("svgp_ctx_create failed with status $st")
        atexit(() -> ccall((:svgp_ctx_destroy, lib), Int32, (Ptr{Cvoid},), CTX[]))
    end
    return CTX[]
end
last_error() = unsafe_string(ccall((:svgp_last_error, lib), Cstring, (Ptr{Cvoid},), ctx()))

struct Unsupported <: Exception end   # internal: "use the pure-Julia method"

function check(st::Integer, terms::Union{Terms,Nothing}=nothing)
    st == 0 && return nothing
    st == 4 && throw(Unsupported())
    st == 1 && throw(ArgumentError(last_error()))
    st == 2 && throw(PosDefException(terms === nothing ? 1 : Int(terms.chol_info)))   # cholesky(Kuu), src/utils.jl:17
    st == 3 && throw(DomainError(-1.0, "sqrt of a negative predictive variance (SVA:354)"))
    return error("libsvgp_mi355x status $st: " * last_error())
end

# ---------------------------------------------------------------------------------------------------------
# unpacking the reference's objects into the POD description
# ---------------------------------------------------------------------------------------------------------
kfamily(::SqExponentialKernel) = Int32(0)
kfamily(::Matern32Kernel) = Int32(1)
kfamily(::Matern52Kernel) = Int32(2)
kfamily(::Any) = nothing

# variance * (Base ∘ ScaleTransform(1/l) | ARDTransform(1 ./ l));  anything else -> nothing -> Julia fallback
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
unpack_mean(m::AbstractGPs.ConstMean) = Float64(m.c)
unpack_mean(::Any) = nothing

unpack_lik(l::GaussianLikelihood) = (Int32(0), Float64(only(l.σ²)))
unpack_lik(::BernoulliLikelihood{<:LogisticLink}) = (Int32(1), 1.0)
unpack_lik(::PoissonLikelihood{<:ExpLink}) = (Int32(2), 1.0)
unpack_lik(::ExponentialLikelihood{<:ExpLink}) = (Int32(3), 1.0)
unpack_lik(l::GammaLikelihood{<:Any,<:ExpLink}) = (Int32(4), Float64(only(l.α)))   # shape in the parameter slot
unpack_lik(::Any) = nothing

layout(x::ColVecs) = (Int32(0), x.X, size(x.X, 1))
layout(x::RowVecs) = (Int32(1), x.X, size(x.X, 2))
layout(x::AbstractVector{<:Real}) = (Int32(2), x, 1)

quad_n(::GPLikelihoods.DefaultExpectationMethod) = Int32(0)
quad_n(q::GPLikelihoods.GaussHermiteExpectation) = Int32(length(q.xs))
quad_n(::GPLikelihoods.AnalyticExpectation) = Int32(0)
quad_n(::Any) = nothing

isotropic_jitter(Σ::Diagonal{<:Real,<:Fill}) = Float64(Σ[1])          # the types SVA:309 accepts; SVA:314 reads Σy[1]
isotropic_jitter(Σ::ScalMat) = Float64(Σ[1])
isotropic_jitter(::Any) = nothing

"Host arrays a ModelDesc points into; must outlive every ccall that receives the desc (GC.@preserve)."
struct Packed{T}
    invl::Vector{Float64}; Z::Array{T}; m::Vector{T}; Lq::Matrix{T}
    desc::ModelDesc
end

function pack(sva::SparseVariationalApproximation{P}, lik, quadrature, ::Type{T}) where {P,T}
    lz, Z, d = layout(sva.fz.x)
    ku = unpack_kernel(sva.fz.f.kernel, d)
    c = unpack_mean(sva.fz.f.mean)
    lk = unpack_lik(lik)
    qn = quad_n(quadrature)
    jit = isotropic_jitter(sva.fz.Σy)
    (ku === nothing || c === nothing || lk === nothing || qn === nothing || jit === nothing) && throw(Unsupported())
    fam, σ², invl = ku
    m = Vector{T}(mean(sva.q))
    Lq = Matrix{T}(_chol_lower(_chol_cov(sva.q)))          # src/utils.jl:15-18
    Zd = Array{T}(Z)
    desc = ModelDesc(T === Float64 ? 0 : 1, fam, P === Centered ? 1 : 0, lk[1], qn, lz, 0, d, length(m), σ²,
                     pointer(invl), c, jit, lk[2], pointer(Zd), pointer(m), pointer(Lq))
    return Packed{T}(invl, Zd, m, Lq, desc)
end

# ---------------------------------------------------------------------------------------------------------
# elbo(sva, lfx, y; num_data, quadrature)   SVA:340-360   (the FiniteGP method SVA:307-317 and approx_lml
# SVA:276-280 forward here unchanged)
# ---------------------------------------------------------------------------------------------------------
const FT = Union{Float32,Float64}

function AbstractGPs.elbo(
    sva::SparseVariationalApproximation, lfx::AbstractGPs.LatentFiniteGP, y::AbstractVector{T};
    num_data=length(y), quadrature=GPLikelihoods.DefaultExpectationMethod(),
) where {T<:FT}
    sva.fz.f === lfx.fx.f ||
        throw(ArgumentError("(Latent)FiniteGP prior is not consistent with SparseVariationalApproximation's"))  # SVA:347-351
    try
        return elbo_device(sva, lfx.lik, lfx.fx.x, y, Float64(num_data), quadrature)
    catch e
        e isa Unsupported || rethrow()
        return invoke(AbstractGPs.elbo,
                      Tuple{SparseVariationalApproximation,AbstractGPs.LatentFiniteGP,AbstractVector},
                      sva, lfx, y; num_data, quadrature)
    end
end

function elbo_device(sva, lik, x, y::AbstractVector{T}, num_data, quadrature) where {T}
    p = pack(sva, lik, quadrature, T)
    lx, X, _ = layout(x)
    Xd, yd = Array{T}(X), Vector{T}(y)
    out, terms = Ref{Float64}(), Terms()
    GC.@preserve p Xd yd begin
        st = ccall((:svgp_elbo_host, lib), Int32,
                   (Ptr{Cvoid}, Ref{ModelDesc}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ref{Float64}, Ref{Terms}),
                   ctx(), p.desc, lx, length(yd), Xd, yd, num_data, out, terms)
    end
    check(st, terms)
    return out[]
end

# ---------------------------------------------------------------------------------------------------------
# resident handles for training loops: upload x, y once, update the model every step
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
    finalizer(D -> ccall((:svgp_data_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), D.h), D)
    return D
end

mutable struct DeviceModel
    h::Ptr{Cvoid}
end
function DeviceModel(p::Packed)
    h = Ref{Ptr{Cvoid}}()
    GC.@preserve p check(ccall((:svgp_model_create, lib), Int32, (Ptr{Cvoid}, Ref{ModelDesc}, Ptr{Ptr{Cvoid}}), ctx(), p.desc, h))
    M = DeviceModel(h[])
    finalizer(M -> ccall((:svgp_model_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), M.h), M)
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
# gradient: one rrule on a flat primitive; Zygote differentiates the (cheap) parameter packing around it
# (examples/a-regression/script.jl:188-194, test/SparseVariationalApproximationModule.jl:170-175)
# ---------------------------------------------------------------------------------------------------------
"""
    svgp_elbo_flat(σ², invl, Z, m, Lq, σ²lik, c, meta, D, off, len, num_data)

`meta = (family, parametrization, likelihood, quadrature_n, layout_z, jitter)`; `Z`, `m`, `Lq` arrays of the
compute eltype; `D::DeviceData`.
"""
function svgp_elbo_flat(σ², invl, Z::Array{T}, m::Vector{T}, Lq::Matrix{T}, σ²lik, c, meta, D, off, len, num_data) where {T}
    return first(value_and_grads(σ², invl, Z, m, Lq, σ²lik, c, meta, D, off, len, num_data, false))
end

function value_and_grads(σ², invl, Z::Array{T}, m::Vector{T}, Lq::Matrix{T}, σ²lik, c, meta, D, off, len, num_data, want) where {T}
    fam, par, lik, qn, lz, jit = meta
    d = length(invl)
    invl64 = collect(Float64, invl)
    desc = ModelDesc(T === Float64 ? 0 : 1, fam, par, lik, qn, lz, 0, d, length(m), Float64(σ²), pointer(invl64),
                     Float64(c), Float64(jit), Float64(σ²lik), pointer(Z), pointer(m), pointer(Lq))
    out, terms = Ref{Float64}(), Terms()
    gl, gz, gm, gLq = zeros(Float64, d), similar(Z), similar(m), similar(Lq)
    g = Grads(0, 0, 0, pointer(gl), pointer(gz), pointer(gm), pointer(gLq))
    GC.@preserve invl64 Z m Lq gl gz gm gLq begin
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:svgp_model_create, lib), Int32, (Ptr{Cvoid}, Ref{ModelDesc}, Ptr{Ptr{Cvoid}}), ctx(), desc, h))
        try
            if want
                check(ccall((:svgp_elbo_grad, lib), Int32,
                            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float64, Ref{Float64}, Ref{Terms}, Ref{Grads}),
                            ctx(), h[], D.h, off, len, Float64(num_data), out, terms, g), terms)
            else
                check(ccall((:svgp_elbo, lib), Int32,
                            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float64, Ref{Float64}, Ref{Terms}),
                            ctx(), h[], D.h, off, len, Float64(num_data), out, terms), terms)
            end
        finally
            ccall((:svgp_model_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), h[])
        end
    end
    return out[], (g.variance, gl, gz, gm, gLq, g.lik_sigma2, g.mean_const)
end

"""
    shard_value_and_grads(σ², invl, Z, m, Lq, σ²lik, c, meta, D, off, len, scale, kl_weight)

Data-parallel shard (one Julia process per GPU, MPI.jl / NCCL.jl for the collective): `scale·ΣE − kl_weight·KL` and its
gradient over this rank's window.  With `scale = num_data / n_global`, `kl_weight = 1 / world_size` on every rank, ONE
sum all-reduce of `(value, gradients...)` is the global ELBO and gradient (SVA:355-359), for both parametrisations.
"""
function shard_value_and_grads(σ², invl, Z::Array{T}, m::Vector{T}, Lq::Matrix{T}, σ²lik, c, meta, D, off, len, scale, kl_weight) where {T}
    fam, par, lik, qn, lz, jit = meta
    d = length(invl)
    invl64 = collect(Float64, invl)
    desc = ModelDesc(T === Float64 ? 0 : 1, fam, par, lik, qn, lz, 0, d, length(m), Float64(σ²), pointer(invl64),
                     Float64(c), Float64(jit), Float64(σ²lik), pointer(Z), pointer(m), pointer(Lq))
    out, terms = Ref{Float64}(), Terms()
    gl, gz, gm, gLq = zeros(Float64, d), similar(Z), similar(m), similar(Lq)
    g = Grads(0, 0, 0, pointer(gl), pointer(gz), pointer(gm), pointer(gLq))
    GC.@preserve invl64 Z m Lq gl gz gm gLq begin
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:svgp_model_create, lib), Int32, (Ptr{Cvoid}, Ref{ModelDesc}, Ptr{Ptr{Cvoid}}), ctx(), desc, h))
        try
            check(ccall((:svgp_elbo_grad_shard, lib), Int32,
                        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Float64, Float64, Ref{Float64}, Ref{Terms}, Ref{Grads}),
                        ctx(), h[], D.h, off, len, Float64(scale), Float64(kl_weight), out, terms, g), terms)
        finally
            ccall((:svgp_model_free, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), h[])
        end
    end
    return out[], (g.variance, gl, gz, gm, gLq, g.lik_sigma2, g.mean_const)
end

function ChainRulesCore.rrule(::typeof(svgp_elbo_flat), σ², invl, Z, m, Lq, σ²lik, c, meta, D, off, len, num_data)
    val, (gσ², gl, gz, gm, gLq, gσ²lik, gc) = value_and_grads(σ², invl, Z, m, Lq, σ²lik, c, meta, D, off, len, num_data, true)
    function pullback(Δ)
        Δ = unthunk(Δ)
        return (NoTangent(), Δ * gσ², Δ .* gl, Δ .* gz, Δ .* gm, Δ .* LowerTriangular(gLq), Δ * gσ²lik, Δ * gc,
                NoTangent(), NoTangent(), NoTangent(), NoTangent(), NoTangent())
    end
    return val, pullback
end

# ---------------------------------------------------------------------------------------------------------
# posterior(sva)   SVA:115-136 (Centered), SVA:160-187 (NonCentered): fills data = (Kuu, B, α)
# ---------------------------------------------------------------------------------------------------------
function posterior_device(sva::SparseVariationalApproximation{P}, ::Type{T}) where {P,T<:FT}
    p = pack(sva, GaussianLikelihood(1.0), GPLikelihoods.DefaultExpectationMethod(), T)
    M = length(p.m)
    Lk, α, B = Matrix{T}(undef, M, M), Vector{T}(undef, M), Matrix{T}(undef, M, M)
    mdl = DeviceModel(p)
    terms = Terms()
    st = ccall((:svgp_posterior, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
               ctx(), mdl.h, Lk, α, B)
    check(st, terms)
    data = (Kuu=Cholesky(LowerTriangular(Lk)), B=LowerTriangular(B), α=α)
    return ApproximateGPs.ApproxPosteriorGP(sva, sva.fz.f, data)
end

# opt-in (keeps `posterior(sva)` itself on the Julia path unless the eltype is a hardware float):
function AbstractGPs.posterior(sva::SparseVariationalApproximation{P,<:AbstractGPs.FiniteGP,<:AbstractMvNormal}, ::Val{:mi355x}) where {P}
    T = eltype(mean(sva.q)) === Float32 ? Float32 : Float64
    try
        return posterior_device(sva, T)
    catch e
        e isa Unsupported || rethrow()
        return posterior(sva)
    end
end

# ---------------------------------------------------------------------------------------------------------
# mean_and_var / mean_and_cov at test inputs   SVA:208-253   (device-side moments for large test sets)
# ---------------------------------------------------------------------------------------------------------
function mean_and_var_device(sva::SparseVariationalApproximation, x, ::Type{T}=Float64; cov::Bool=false) where {T<:FT}
    p = pack(sva, GaussianLikelihood(1.0), GPLikelihoods.DefaultExpectationMethod(), T)
    mdl = DeviceModel(p)
    lx, X, _ = layout(x)
    Xd = Array{T}(X)
    n = length(x)
    μ, v = Vector{T}(undef, n), Vector{T}(undef, n)
    C = cov ? Matrix{T}(undef, n, n) : nothing
    GC.@preserve Xd check(ccall((:svgp_predict, lib), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        ctx(), mdl.h, lx, n, Xd, μ, v, cov ? C : C_NULL))
    return cov ? (μ, C) : (μ, v)
end

end # module
