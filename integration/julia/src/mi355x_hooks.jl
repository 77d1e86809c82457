# mi355x_hooks.jl — included from src/ApproximateGPs.jl BEFORE SparseVariationalApproximationModule.jl.
#
# The hook points the reference methods call first (integration/julia/ApproximateGPs_hooks.patch).  Their only methods
# here are the do-nothing fallbacks: with no SVGPMI355X.jl loaded — or for any argument types it does not cover — every
# hook returns `nothing` and the reference's pure-Julia body runs, under Zygote too (differentiating a function that
# returns `nothing` contributes nothing).  src/SVGPMI355X.jl adds the device methods and the ChainRules rule.
module MI355XHooks

"`elbo(sva, lfx, y; num_data, quadrature)` (SVA:340-360) on the device, or `nothing`."
try_elbo(args...) = nothing
"`posterior(sva)` (SVA:115-136, :160-187) with the factors computed on the device, or `nothing`."
try_posterior(args...) = nothing
"`(mean, var, cov)` of the approximate posterior at `x` (SVA:208-253), or `nothing`."
try_predict(args...; kwargs...) = nothing
"`cov(f, x, y)` (SVA:255-264), or `nothing`."
try_cross_cov(args...) = nothing

end # module
