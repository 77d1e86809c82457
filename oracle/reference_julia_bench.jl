# oracle/reference_julia_bench.jl — times the REAL reference (ApproximateGPs.jl's own elbo, SVA:340-360) on the synthetic
# problem bench.py hands over as an .npz (BASELINE.md §3.1).  Only run when bench.py finds `julia` on the host; prints one
# JSON line {"elbo":…, "seconds_per_eval":…, "threads":…}.  Test infrastructure: never used by the product path.
# NOT run in the build image (no Julia there); same model construction as oracle/reference_julia.jl.
using ApproximateGPs, AbstractGPs, KernelFunctions, GPLikelihoods, Distributions, LinearAlgebra
using PDMats: PDMat
using NPZ

scalar(a) = a isa AbstractArray ? only(a) : a
g = npzread(ARGS[1])
BLAS.set_num_threads(Sys.CPU_THREADS)
fam, lk = Int(scalar(g["family"])), Int(scalar(g["lik"]))
base = fam == 0 ? SqExponentialKernel() : fam == 1 ? Matern32Kernel() : Matern52Kernel()
k = scalar(g["variance"]) * (base ∘ ARDTransform(vec(g["inv_lengthscale"])))
f = GP(k)
q = MvNormal(vec(g["m"]), PDMat(Cholesky(LowerTriangular(g["Lq"]))))
sva = SparseVariationalApproximation(NonCentered(), f(ColVecs(g["z"]), scalar(g["jitter"])), q)
lik = lk == 0 ? GaussianLikelihood(scalar(g["sigma2"])) : lk == 1 ? BernoulliLikelihood() : PoissonLikelihood()
lfx = LatentGP(f, lik, 1e-18)(ColVecs(g["x"]))
y = vec(g["y"])
elbo(sva, lfx, y)                     # compile
ts = Float64[]
val = 0.0
for _ in 1:3
    t0 = time()
    global val = elbo(sva, lfx, y)
    push!(ts, time() - t0)
end
sort!(ts)
println("{\"elbo\": $(val), \"seconds_per_eval\": $(ts[2]), \"threads\": $(BLAS.get_num_threads()), \"unit\": \"s per eval on the sample\"}")
