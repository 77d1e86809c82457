/*
 * svgp_mi355x.h — C-ABI of libsvgp_mi355x.so: the MI355X-native (gfx950) sparse-variational-GP
 * ELBO / posterior path that drops in under ApproximateGPs.jl's
 * SparseVariationalApproximation / elbo / approx_lml / posterior API.
 *
 * The reference has NO FFI boundary today (pure Julia, multiple dispatch on AbstractGPs generics).
 * Each entry point below therefore cites the reference METHOD BODY it replaces
 * (paths relative to the reference repo; SVA = src/SparseVariationalApproximationModule.jl),
 * and INTEGRATION.md shows the Julia `ccall` binding a maintainer would add.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns an int32 status (SVGP_OK == 0) and
 *     never throws, aborts, prints or calls back into the host language.
 *   - all matrices are column-major (Julia layout).  Host pointers are borrowed for the call only.
 *   - array elements are in the compute dtype of the object (SVGP_F64 -> double, SVGP_F32 -> float);
 *     scalars are always double.
 *   - one svgp_ctx is bound to one GPU and one HIP stream; distinct contexts may be used
 *     concurrently from different threads, one context is not re-entrant.  Calls block until
 *     their host-visible results are written.
 *   - multi-GPU: a context may carry one rank of an RCCL communicator (svgp_ctx_attach_comm for one process per
 *     GPU, svgp_group_create for one process driving several GPUs).  svgp_elbo and svgp_elbo_grad are then
 *     COLLECTIVE: every rank evaluates its own shard and the library sums the partial results with one
 *     ncclAllReduce over xGMI on the device (no host hop).  Everything else stays local to the context.
 */
#ifndef SVGP_MI355X_H
#define SVGP_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVGP_MAX_D 64 /* largest input dimension the device path takes (round 3: 32 -> 64); beyond it SVGP_UNSUPPORTED from every entry
                         point that takes a dimension (model, data upload / wrap): the host falls back */
#define SVGP_ABI_VERSION 5 /* 3 = 2 + svgp_marginals, svgp_elbo_grad_ext, SVGP_LIK_BERNOULLI_NORMCDF; 4 = 3 + svgp_offload_advice /
                              svgp_offload_work and svgp_timing.ms_chol; 5 = 4 + svgp_last_timing_sized: svgp_last_timing writes the
                              48-byte v2 / v3 layout again (v4 let it write 56 bytes into a buffer a v3 host sized at 48), the
                              fields appended since are read through the sized call.  Additions only: v2 / v3 callers keep working */

/* status codes -> Julia exceptions raised by the shim */
enum {
  SVGP_OK = 0,
  SVGP_INVALID_ARG = 1,  /* ArgumentError        (SVA:347-351 prior mismatch is checked Julia-side) */
  SVGP_NOT_POSDEF = 2,   /* PosDefException(info) from cholesky(Kuu)  [LinearAlgebra]; info in svgp_terms */
  SVGP_NEG_VARIANCE = 3, /* DomainError from sqrt(v < 0) inside marginals (SVA:354) */
  SVGP_UNSUPPORTED = 4,  /* shim falls back to the pure-Julia method */
  SVGP_HIP_ERROR = 5,
  SVGP_RCCL_ERROR = 6,
  SVGP_OOM = 7
};

enum { SVGP_F64 = 0, SVGP_F32 = 1 };

/* input layouts [KernelFunctions]: ColVecs(X) X is d×n (a point is contiguous); RowVecs(X) X is n×d
 * (a feature is contiguous); a plain Vector is d = 1. */
enum { SVGP_COLVECS = 0, SVGP_ROWVECS = 1, SVGP_VEC = 2 };

/* kernel families [KernelFunctions]: variance * (Base ∘ ARDTransform(inv_lengthscale)) */
enum { SVGP_KERNEL_SE = 0, SVGP_KERNEL_MATERN32 = 1, SVGP_KERNEL_MATERN52 = 2 };

/* likelihoods [GPLikelihoods]; SVA:307-317 wraps FiniteGP noise as GAUSSIAN(σ² = fx.Σy[1]) */
enum {
  SVGP_LIK_GAUSSIAN = 0,
  SVGP_LIK_BERNOULLI_LOGISTIC = 1,
  SVGP_LIK_POISSON_EXP = 2,      /* PoissonLikelihood(exp):      y ~ Poisson(exp f) */
  SVGP_LIK_EXPONENTIAL_EXP = 3,  /* ExponentialLikelihood(exp):  y ~ Exponential(scale exp f) = Gamma(1, scale exp f) */
  SVGP_LIK_GAMMA_EXP = 4,        /* GammaLikelihood(alpha, exp): y ~ Gamma(shape alpha, scale exp f); alpha in lik_sigma2 */
  SVGP_LIK_BERNOULLI_NORMCDF = 5 /* BernoulliLikelihood(NormalCDFLink()) (probit): y ~ Bernoulli(Phi(f)); Gauss-Hermite */
};

/* SVA:41 Centered, SVA:57 NonCentered (the 2-arg constructor's default, SVA:93-95) */
enum { SVGP_NONCENTERED = 0, SVGP_CENTERED = 1 };

/* what to do when a predictive variance (+1e-18) is negative: the reference throws DomainError */
enum { SVGP_NEGVAR_ERROR = 0, SVGP_NEGVAR_CLAMP = 1 };

typedef struct svgp_ctx svgp_ctx;
typedef struct svgp_data svgp_data;
typedef struct svgp_model svgp_model;
struct svgp_model_desc;

/* Everything a SparseVariationalApproximation + likelihood holds (SVA:59-62):
 *   fz = GP(mean_const, variance * (Base ∘ ARD(inv_lengthscale)))(z, jitter),  q = MvNormal(m, Lq Lq'). */
typedef struct svgp_model_desc {
  int32_t dtype;           /* SVGP_F64 | SVGP_F32: compute dtype AND element type of z, m, Lq */
  int32_t kernel;          /* SVGP_KERNEL_* */
  int32_t parametrization; /* SVGP_NONCENTERED | SVGP_CENTERED */
  int32_t likelihood;      /* SVGP_LIK_* */
  int32_t quadrature_n;    /* 0 = DefaultExpectationMethod (analytic where closed form exists, else GH-20);
                              n > 0 = GaussHermiteExpectation(n) */
  int32_t layout_z;        /* layout of z: SVGP_COLVECS (d×M) | SVGP_ROWVECS (M×d) | SVGP_VEC */
  int32_t neg_var_policy;  /* SVGP_NEGVAR_* */
  int32_t d;               /* input dimension */
  int64_t M;               /* number of inducing points */
  double variance;         /* kernel variance σ_k² */
  const double* inv_lengthscale; /* d entries; isotropic = all equal */
  double mean_const;       /* ConstMean value, 0 for ZeroMean */
  double jitter;           /* fz.Σy (isotropic), part of Kuu (src/utils.jl:17) */
  double lik_sigma2;       /* likelihood parameter: GaussianLikelihood σ², GammaLikelihood shape α; unused otherwise */
  const void* z;           /* inducing inputs */
  const void* m;           /* mean(q), M */
  const void* Lq;          /* _chol_lower(_chol_cov(q)) (src/utils.jl:15,18), M×M, upper triangle ignored */
} svgp_model_desc;

/* Optional breakdown of one ELBO evaluation (SVA:355-359). */
typedef struct svgp_terms {
  double elbo;        /* expectation * scale - kl */
  double expectation; /* Σ_i E_{q(f_i)}[log p(y_i | f_i)], unscaled */
  double kl;          /* _prior_kl(sva)  SVA:362-373 */
  double scale;       /* num_data / n_batch  SVA:357-358 */
  double logdet_kuu;  /* 2 Σ log diag chol(Kuu) */
  int64_t n_points;
  int64_t n_neg_var;  /* points whose variance + 1e-18 was negative */
  int32_t chol_info;  /* 0, or the order of the first non-positive leading minor of Kuu (LAPACK info) */
  int32_t reserved;
} svgp_terms;

/* Device timings of the last call on the context, measured with HIP events on the context's stream. */
typedef struct svgp_timing {
  double ms_total;
  double ms_prep;  /* Kuu, cholesky, diagonal-block inverses, panel products, KL */
  double ms_strip; /* the fused Kuf -> trsm -> trmm strip kernel alone (one launch) */
  double ms_expect;/* marginals + expected log-likelihood + final reduce */
  double ms_kuf;   /* standalone Kuf assembly kernel (svgp_kuf only) */
  int64_t strip_launches;
  /* ---- end of the v2 / v3 layout (SVGP_TIMING_V3_BYTES = 48): svgp_last_timing writes exactly the fields above ---- */
  double ms_chol;  /* (v4) cholesky(Kuu) alone - the blocked factorisation with its T panels - inside ms_prep */
  double ms_overlap; /* (v5) part of ms_prep that ran BESIDE the strips (flag-gated strips on the second stream); 0 when serial */
} svgp_timing;
#define SVGP_TIMING_V3_BYTES 48

/* ---- library / context ------------------------------------------------------------------- */
int32_t svgp_version(void);
int32_t svgp_device_count(void);
/* `stream` is a hipStream_t to run on (e.g. torch's current stream) or NULL to create a private one. */
int32_t svgp_ctx_create(int32_t device_id, void* stream, svgp_ctx** out);
int32_t svgp_ctx_destroy(svgp_ctx* ctx);
/* text of the last error on ctx (owned by the library, valid until the next call on ctx) */
const char* svgp_last_error(const svgp_ctx* ctx);
/* writes the first SVGP_TIMING_V3_BYTES of the timing struct: the layout every ABI version shares.
 * NOTE for hosts compiled against the ABI-v4 header (56-byte struct): v4's svgp_last_timing also wrote ms_chol; since v5 it does NOT
 * (a v3-compiled host had sized its buffer at 48 bytes), so a v4 host reading ms_chol after this call sees whatever its buffer held.
 * The library cannot know the caller's struct size here: v4 hosts must move to svgp_last_timing_sized. */
int32_t svgp_last_timing(const svgp_ctx* ctx, svgp_timing* out);
/* (v5) writes the first min(out_bytes, size of svgp_timing) bytes: a host passes the size of the struct IT was compiled against, so the
 * struct can grow without the library ever writing past the caller's buffer */
int32_t svgp_last_timing_sized(const svgp_ctx* ctx, void* out, int64_t out_bytes);

/* ---- multi-GPU: data-parallel shards of the sum over points (SVA:355-359), SURVEY §8e ----------
 * The expectation term is a plain sum over data points, so each rank holds a shard of (x, y) in its own HBM,
 * replicates the M-sized work (Kuu, cholesky, KL) and the ranks' {sum E, n_points, n_neg_var, status flags}
 * are summed by ONE ncclAllReduce of 8 doubles issued by the library on the context's stream, straight from
 * the device buffer the reduction kernel wrote.  RCCL is loaded with dlopen on first use (no link-time
 * dependency; SVGP_RCCL_LIB overrides the library name); failures map to SVGP_RCCL_ERROR.
 *
 * One process per GPU: rank 0 calls svgp_comm_unique_id, the host transports the 128 bytes to every rank
 * (MPI / torch.distributed / Distributed.jl), every rank calls svgp_ctx_attach_comm on its own context
 * (collective: ncclCommInitRank).  No rank can leave its peers inside a collective:
 *   svgp_elbo / svgp_elbo_host: ONE fixed-size all-reduce (8 doubles); a rank that fails locally - bad arguments, NULL model,
 *     allocation failure, HIP error - enters it with zero contributions and the failure flag; its peers return SVGP_RCCL_ERROR.
 *   svgp_elbo_grad / svgp_elbo_grad_ext: an opening fixed-size all-reduce {batch size, failure flag} every rank enters, whatever
 *     state it is in - a rank that cannot evaluate (also one handed a NULL model) sends the flag, returns its own error and joins
 *     nothing else; every other rank reads the reduced flag before the closing gradient all-reduce (whose size depends on M and d),
 *     skips it and returns SVGP_RCCL_ERROR.  A rank that fails later, inside its backward pass, enters the closing all-reduce with
 *     its failure flag.  The communicator stays usable after either kind of failure.
 * ncclCommAbort is left for the case that a collective itself cannot be issued (an RCCL error).  The one-process group calls
 * (svgp_group_elbo / svgp_group_elbo_grad) validate every member before anything is enqueued and return the first member's
 * error without touching the communicator. */
#define SVGP_COMM_ID_BYTES 128
int32_t svgp_comm_unique_id(void* id_out /* SVGP_COMM_ID_BYTES */);
int32_t svgp_ctx_attach_comm(svgp_ctx* ctx, const void* id, int32_t world_size, int32_t rank);
int32_t svgp_ctx_detach_comm(svgp_ctx* ctx);
int32_t svgp_ctx_comm_info(const svgp_ctx* ctx, int32_t* world_size_out, int32_t* rank_out);

/* One process, several GPUs (the natural shape of a Julia host): a group owns one context per device and their
 * communicator (ncclCommInitAll).  Arrays of handles are indexed by member; svgp_group_ctx(g, i) gives member i's
 * context for the local calls (svgp_predict, svgp_posterior, svgp_*_free ...). */
typedef struct svgp_group svgp_group;
int32_t svgp_group_create(int32_t n_devices, const int32_t* device_ids, svgp_group** out);
int32_t svgp_group_destroy(svgp_group* group);
int32_t svgp_group_size(const svgp_group* group);
svgp_ctx* svgp_group_ctx(svgp_group* group, int32_t member);
const char* svgp_group_last_error(const svgp_group* group);
/* uploads x, y once, sharded contiguously by point index over the members (shards_out: n_devices handles) */
int32_t svgp_group_data_upload(svgp_group* group, int32_t dtype, int32_t layout, int32_t d, int64_t n,
                               const void* x_host, const void* y_host, svgp_data** shards_out);
/* one replica of the model per member (models_out: n_devices handles) */
int32_t svgp_group_model_create(svgp_group* group, const struct svgp_model_desc* desc, svgp_model** models_out);
int32_t svgp_group_model_update(svgp_group* group, svgp_model* const* models, const struct svgp_model_desc* desc);

/* ---- data: x = lfx.fx.x, y (SVA:340-343) kept resident in HBM ------------------------------ */
int32_t svgp_data_upload(svgp_ctx* ctx, int32_t dtype, int32_t layout, int32_t d, int64_t n,
                         const void* x_host, const void* y_host, svgp_data** out);
/* wrap device memory without copying: x_dev is feature-major [d][ldx] (RowVecs storage), y_dev may be NULL */
int32_t svgp_data_wrap_device(svgp_ctx* ctx, int32_t dtype, int32_t d, int64_t n, int64_t ldx,
                              const void* x_dev, const void* y_dev, svgp_data** out);
int32_t svgp_data_free(svgp_ctx* ctx, svgp_data* data);

/* ---- model: SparseVariationalApproximation(fz, q) + likelihood, resident in HBM -------------- */
int32_t svgp_model_create(svgp_ctx* ctx, const svgp_model_desc* desc, svgp_model** out);
/* new parameter values, same M / d / dtype (a training step) */
int32_t svgp_model_update(svgp_ctx* ctx, svgp_model* model, const svgp_model_desc* desc);
int32_t svgp_model_free(svgp_ctx* ctx, svgp_model* model);

/* ---- elbo(sva, lfx, y; num_data, quadrature)  replaces SVA:340-360 (and :307-317, :276-280) --- */
/* evaluates points [batch_off, batch_off + batch_len) of `data`; num_data <= 0 means batch_len.
 * On a context with a communicator: COLLECTIVE — every rank passes its own shard's batch and gets the global
 * ELBO = (sum over all ranks' points) * num_data / n_global - KL; terms_out->n_points = n_global.
 * Non-finite numbers are NOT an error, as in the reference: a NaN coordinate of x gives NaN marginals for that point (and only that
 * point) and a NaN ELBO, an infinite y an infinite ELBO, a zero on the diagonal of Lq KL = +Inf - the status is SVGP_OK and the numbers say
 * it; a NaN in z or in a kernel parameter makes Kuu NaN and cholesky(Kuu) reports SVGP_NOT_POSDEF as LAPACK does (tests/fuzz_errors.py). */
int32_t svgp_elbo(svgp_ctx* ctx, svgp_model* model, const svgp_data* data, int64_t batch_off,
                  int64_t batch_len, double num_data, double* elbo_out, svgp_terms* terms_out);
/* data-parallel shard: only Σ_i E[log p(y_i|f_i)] over the shard's points (no scale, no KL), so that
 * ranks sum partials with ONE all-reduce and subtract the KL once (SVA:355-359).
 * partial_out[4] = {sum_expectation, n_points, n_neg_var, chol_info}.  Always LOCAL (never a collective): the building
 * block for a host that runs its own all-reduce. */
int32_t svgp_elbo_partial(svgp_ctx* ctx, svgp_model* model, const svgp_data* data, int64_t batch_off,
                          int64_t batch_len, double partial_out[4]);
/* KL(q || p) and logdet(Kuu) of the model alone: _prior_kl  SVA:362-373 */
int32_t svgp_prior_kl(svgp_ctx* ctx, svgp_model* model, double* kl_out, double* logdet_kuu_out);
/* one-shot variant taking host x, y (the literal drop-in for an ad-hoc elbo(sva, fx, y) call) */
int32_t svgp_elbo_host(svgp_ctx* ctx, const svgp_model_desc* desc, int32_t layout_x, int64_t n,
                       const void* x_host, const void* y_host, double num_data, double* elbo_out,
                       svgp_terms* terms_out);

/* ---- gradient of the ELBO (what Zygote produces for the reference's training loops:
 * examples/a-regression/script.jl:188-194, test/SparseVariationalApproximationModule.jl:170-175).  Both parametrisations
 * (Centered: the adjoint runs on the whitened (Lk \\ (m - c), Lk \\ Lq) and is chained back through Lk).  Every output array is caller-allocated host memory; z, m, Lq gradients have the
 * dtype / layout of the corresponding svgp_model_desc arrays (Lq: lower triangle, upper zeroed); NULL skips one. */
typedef struct svgp_grads {
  double variance;          /* d elbo / d kernel variance */
  double lik_sigma2;        /* d elbo / d likelihood parameter (Gaussian sigma^2, Gamma alpha; 0 otherwise) */
  double mean_const;        /* d elbo / d ConstMean value */
  double* inv_lengthscale;  /* d entries */
  void* z;
  void* m;
  void* Lq;
} svgp_grads;
/* On a context with a communicator: COLLECTIVE — value and gradient of the global ELBO on every rank (the batch size
 * is all-reduced on the device before the backward pass, the gradient by one grouped ncclAllReduce after it).
 * The VALUE this call returns and svgp_elbo's value of the same batch are the same quantity computed along two paths: svgp_elbo takes
 * the posterior variance as k(x,x) - sum A^2 + sum (B'A)^2 (SVA:251, two column sums), the value-and-gradient strips have no B'A product
 * and take it from the algebraically equal k_j' (R A)_j, R = Lk^-T (Lq Lq' - I).  In SVGP_F64 the two agree to ~1e-15 relative.  In
 * SVGP_F32 they differ by rounding: measured |value_grad - value_elbo| / |value_elbo| = 2e-9 ... 3e-7 on the benchmark shapes (H32, C3, C5
 * at full size) and on ill-conditioned posteriors with a marginal variance down to 4.5e-5 of the prior's (profiles/round5/f32_value_gap.log;
 * both are within 1e-6 of the fp64 value there) - a host that logs both should expect agreement to ~1e-6 relative in fp32, not to the last
 * bit (tests/test_gpu_round5.py asserts 2e-6).  The fp32 gradient blocks on those posteriors stay within 5e-5 (m, Lq) / 5e-4 (z, inverse
 * lengthscales) of the fp64 gradient, relative to each block's largest entry. */
int32_t svgp_elbo_grad(svgp_ctx* ctx, svgp_model* model, const svgp_data* data, int64_t batch_off, int64_t batch_len,
                       double num_data, double* elbo_out, svgp_terms* terms_out, svgp_grads* grads_out);
/* the same two evaluations driven from one process over the members of a group: member i evaluates points
 * [offs[i], offs[i] + lens[i]) of shards[i]; results are the global ones (read from member 0). */
int32_t svgp_group_elbo(svgp_group* group, svgp_model* const* models, const svgp_data* const* shards,
                        const int64_t* offs, const int64_t* lens, double num_data, double* elbo_out,
                        svgp_terms* terms_out);
int32_t svgp_group_elbo_grad(svgp_group* group, svgp_model* const* models, const svgp_data* const* shards,
                             const int64_t* offs, const int64_t* lens, double num_data, double* elbo_out,
                             svgp_terms* terms_out, svgp_grads* grads_out);
/* data-parallel shard of the value-and-gradient (the gradient counterpart of svgp_elbo_partial):
 *   value = scale * Σ_{i in shard} E_{q(f_i)}[log p(y_i|f_i)] - kl_weight * KL   and its gradient.  Always LOCAL.
 * With scale = num_data / n_global and kl_weight = 1 / world_size on every rank, ONE sum all-reduce of
 * (value, gradients) is the global ELBO (SVA:355-359) and its gradient, for both parametrisations
 * (terms_out->elbo holds the same value; terms_out->scale = scale). */
int32_t svgp_elbo_grad_shard(svgp_ctx* ctx, svgp_model* model, const svgp_data* data, int64_t batch_off, int64_t batch_len,
                             double scale, double kl_weight, double* value_out, svgp_terms* terms_out,
                             svgp_grads* grads_out);

/* ---- any other single-latent likelihood: the host evaluates it on the device-computed marginals ----------------
 * (SURVEY 8 f4 "generic GH for user link functions".)  Only SVA:355, expected_loglikelihood(quadrature, lik, q_f, y),
 * depends on the likelihood, and it is O(n) scalar work; everything O(M^2 n) is likelihood-free.  So for a likelihood
 * the SVGP_LIK_* codes do not enumerate the binding keeps the reference's own GPLikelihoods call and gives it the
 * device's marginals:
 *   svgp_marginals        replaces marginals(f_post(x)) of SVA:354: mean_out[i] = mu_i, var_out[i] = v_i + 1e-18
 *                         (fp64, batch_len each; host); SVGP_NEG_VARIANCE / clamping per neg_var_policy.  Needs no y.
 *   svgp_elbo_grad_ext    value and gradient of  (num_data / batch_len) * sum_e - KL  where the host passes
 *                         sum_e = sum_i E_i and g_mu[i] = dE_i/dmu_i, g_v[i] = dE_i/dv_i (fp64, unscaled, host): the
 *                         same backward pass as svgp_elbo_grad with these in place of the built-in likelihood's;
 *                         grads_out->lik_sigma2 = 0 (the likelihood's own parameters are the host's to differentiate).
 *                         On a context with a communicator it is collective exactly like svgp_elbo_grad (sum_e and the
 *                         point gradients are then this rank's shard's). */
int32_t svgp_marginals(svgp_ctx* ctx, svgp_model* model, const svgp_data* data, int64_t batch_off, int64_t batch_len,
                       double* mean_out, double* var_out);
int32_t svgp_elbo_grad_ext(svgp_ctx* ctx, svgp_model* model, const svgp_data* data, int64_t batch_off, int64_t batch_len,
                           double num_data, double sum_e, const double* g_mu, const double* g_v, double* elbo_out,
                           svgp_terms* terms_out, svgp_grads* grads_out);

/* ---- posterior(sva)  replaces SVA:115-136 (Centered) / SVA:160-187 (NonCentered) -------------
 * fills ApproxPosteriorGP.data = (Kuu = Cholesky(Lk), B, α): Lk_out M×M lower (upper zeroed),
 * alpha_out M, B_out M×M (NULL to skip; NonCentered B is the caller's Lq). */
int32_t svgp_posterior(svgp_ctx* ctx, svgp_model* model, void* Lk_out, void* alpha_out, void* B_out);

/* ---- mean / var / mean_and_var / cov / mean_and_cov  replaces SVA:208-253 ---------------------
 * mean_out, var_out: n (either may be NULL); cov_out: n×n or NULL (SVA:223-228). */
int32_t svgp_predict(svgp_ctx* ctx, svgp_model* model, int32_t layout_x, int64_t n, const void* x_host,
                     void* mean_out, void* var_out, void* cov_out);
/* cross-covariance cov(f, x, y)  replaces SVA:255-264; cov_out is nx×ny */
int32_t svgp_predict_cross_cov(svgp_ctx* ctx, svgp_model* model, int32_t layout, int64_t nx,
                               const void* x_host, int64_t ny, const void* y_host, void* cov_out);

/* ---- Kuf = cov(prior, z, x) alone  (SVA:216), for the "Kuf-assembly HBM GB/s" metric ----------
 * Kuf_out_host: M×batch_len column-major or NULL (result stays in HBM; timing only). */
int32_t svgp_kuf(svgp_ctx* ctx, svgp_model* model, const svgp_data* data, int64_t batch_off,
                 int64_t batch_len, void* Kuf_out_host);

/* ---- Gauss–Hermite rule used by GH-n (FastGaussQuadrature.gausshermite) ---------------------- */
int32_t svgp_gausshermite(int32_t n, double* nodes_out, double* weights_out);

/* ---- small problems: is the device worth calling? -----------------------------------------------
 * The reference's own workloads are tiny (examples/a-regression/script.jl:33,69,176: N = 10 000, M = 20, minibatch 100;
 * test/SparseVariationalApproximationModule.jl: N <= 100).  A call into this library has a floor that does not depend on the
 * problem - measured on MI355X (profiles/round3/small_problems.md): 160-180 us for svgp_elbo on resident data, ~250 us for the
 * one-shot svgp_elbo_host the un-modified `elbo(sva, lfx, y)` reaches through the hook, ~600 us for value-and-gradient - while
 * the host path costs about 1e-10 s per unit of  work(n, M, d) = n M (2 M + 3 d + 30) + M^3 / 3.  Returns 1 when
 * work >= SVGP_OFFLOAD_MIN_WORK (default 3e6: the measured crossover), else 0; the hooks of the Julia binding (try_elbo,
 * try_posterior, try_predict, the rrule) and the Python mirror decline below it and the pure-Julia method body runs.
 * The environment variable SVGP_OFFLOAD_MIN_WORK overrides the threshold (0: always offload).  Advice only: the explicit
 * resident-handle calls (svgp_elbo, svgp_elbo_grad, ...) never refuse a problem for being small.  No context, no GPU needed. */
int32_t svgp_offload_advice(int64_t n_points, int64_t M, int32_t d, int32_t dtype, int32_t want_gradient);
double svgp_offload_work(int64_t n_points, int64_t M, int32_t d);

#ifdef __cplusplus
}
#endif
#endif /* SVGP_MI355X_H */
