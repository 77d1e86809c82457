// api.hip — the C-ABI of libsvgp_mi355x.so (include/svgp_mi355x.h).  Host orchestration only: every
// number is produced by the HIP kernels in prep.hip / strip.hip; there is no CPU compute path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/svgp_mi355x.h"
#include "ctx.hpp"
#include "kernels.hpp"
#include "knobs.hpp"
#include "lik.hpp"

using namespace svgp;

// Events between the context's own streams and for device-side timing need no system-scope fence (the cache write-back and invalidation
// that makes device memory visible to the HOST): a record then costs the stream ~1 instead of ~5 us (profiles/round4/minibatch_step.md
// section 7).  Everything the host reads comes through hipMemcpy + a stream / event synchronisation of DEFAULT events (ev_piece).
// SVGP_EVENT_FENCE=1 (experiments build, process-wide, A/B): default events everywhere.
inline unsigned event_flags(bool timing) {
  static const bool fence = exp_int("SVGP_EVENT_FENCE", 0) == 1;   // experiments build: A/B
  return (timing ? 0u : unsigned(hipEventDisableTiming)) | (fence ? 0u : unsigned(hipEventDisableSystemFence));
}
#define kSyncEvent event_flags(false)
#define kTimingEvent event_flags(true)
// (no query of events that were never recorded: it would leave a sticky "invalid resource handle" on the context's device)
inline hipError_t elapsed_ms(const svgp_ctx* ctx, float* ms, hipEvent_t a, hipEvent_t b) {
  if (!ctx->timing_on) { *ms = 0.0f; return hipErrorNotReady; }
  return hipEventElapsedTime(ms, a, b);
}
// timing events (ms_prep / ms_strip / ms_chol / ms_overlap of svgp_last_timing): recorded unless the context was created with SVGP_TIMING=0
#define TREC(ctx, ev, stream)                            \
  do {                                                   \
    if ((ctx)->timing_on) HIPC(ctx, hipEventRecord(ev, stream)); \
  } while (0)

namespace {

size_t esize(int dtype) { return dtype == SVGP_F64 ? 8 : 4; }

// Golub–Welsch: nodes/weights of the n-point Gauss–Hermite rule (weight exp(-x²)), as
// FastGaussQuadrature.gausshermite(n).  Symmetric tridiagonal QL with implicit shifts on the Jacobi matrix.
int gauss_hermite(int n, double* xs, double* ws) {
  if (n < 1 || n > 512) return SVGP_INVALID_ARG;
  std::vector<double> d(n, 0.0), e(n, 0.0), z(n, 0.0);
  for (int i = 1; i < n; ++i) e[i - 1] = std::sqrt(0.5 * i);
  z[0] = 1.0;  // first row of the eigenvector matrix
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
      }
      if (m != l) {
        if (++iter > 200) return SVGP_INVALID_ARG;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i], b = c * e[i];
          r = std::hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
          f = z[i + 1];
          z[i + 1] = s * z[i] + c * f;
          z[i] = c * z[i] - s * f;
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  // sort ascending
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  for (int i = 1; i < n; ++i) {
    int k = idx[i], j = i - 1;
    while (j >= 0 && d[idx[j]] > d[k]) {
      idx[j + 1] = idx[j];
      --j;
    }
    idx[j + 1] = k;
  }
  const double mu0 = 1.7724538509055160273;  // sqrt(pi)
  for (int i = 0; i < n; ++i) {
    xs[i] = d[idx[i]];
    ws[i] = mu0 * z[idx[i]] * z[idx[i]];
  }
  // symmetrise (the rule is exactly symmetric)
  for (int i = 0; i < n / 2; ++i) {
    const double a = 0.5 * (xs[n - 1 - i] - xs[i]);
    xs[i] = -a;
    xs[n - 1 - i] = a;
    const double w = 0.5 * (ws[i] + ws[n - 1 - i]);
    ws[i] = ws[n - 1 - i] = w;
  }
  if (n % 2) xs[n / 2] = 0.0;
  return SVGP_OK;
}

int validate_desc(svgp_ctx* ctx, const svgp_model_desc* d) {
  if (!d) return fail(ctx, SVGP_INVALID_ARG, "null model descriptor");
  if (d->dtype != SVGP_F64 && d->dtype != SVGP_F32) return fail(ctx, SVGP_INVALID_ARG, "dtype must be SVGP_F64 or SVGP_F32");
  if (d->kernel < 0 || d->kernel > SVGP_KERNEL_MATERN52) return fail(ctx, SVGP_UNSUPPORTED, "unsupported kernel family");
  if (d->likelihood < 0 || d->likelihood > SVGP_LIK_BERNOULLI_NORMCDF) return fail(ctx, SVGP_UNSUPPORTED, "unsupported likelihood");
  if (d->parametrization != SVGP_NONCENTERED && d->parametrization != SVGP_CENTERED)
    return fail(ctx, SVGP_INVALID_ARG, "parametrization must be SVGP_NONCENTERED or SVGP_CENTERED");
  if (d->d < 1 || d->d > SVGP_MAX_D) return fail(ctx, SVGP_UNSUPPORTED, "input dimension must be in 1..64");
  if (d->M < 1) return fail(ctx, SVGP_INVALID_ARG, "M must be >= 1");
  if (d->layout_z < 0 || d->layout_z > SVGP_VEC) return fail(ctx, SVGP_INVALID_ARG, "bad layout_z");
  if (d->layout_z == SVGP_VEC && d->d != 1) return fail(ctx, SVGP_INVALID_ARG, "SVGP_VEC layout requires d == 1");
  if (d->quadrature_n < 0 || d->quadrature_n > 512) return fail(ctx, SVGP_INVALID_ARG, "quadrature_n must be in 0..512");
  if (!d->inv_lengthscale || !d->z || !d->m || !d->Lq) return fail(ctx, SVGP_INVALID_ARG, "null parameter array");
  if (!(d->variance > 0)) return fail(ctx, SVGP_INVALID_ARG, "kernel variance must be positive");
  if (d->likelihood == SVGP_LIK_GAUSSIAN && !(d->lik_sigma2 > 0)) return fail(ctx, SVGP_INVALID_ARG, "Gaussian likelihood needs sigma2 > 0");
  if (d->likelihood == SVGP_LIK_GAMMA_EXP && !(d->lik_sigma2 > 0)) return fail(ctx, SVGP_INVALID_ARG, "Gamma likelihood needs shape alpha > 0");
  return SVGP_OK;
}

// Gaussian sigma^2 / Gamma shape alpha; 1 for the parameter-free likelihoods
double lik_param(const svgp_model_desc& d) {
  return (d.likelihood == SVGP_LIK_GAUSSIAN || d.likelihood == SVGP_LIK_GAMMA_EXP) ? d.lik_sigma2 : 1.0;
}

int effective_gh(const svgp_model_desc& d) {
  if (d.quadrature_n > 0) return d.quadrature_n;
  if (d.likelihood != SVGP_LIK_BERNOULLI_LOGISTIC && d.likelihood != SVGP_LIK_BERNOULLI_NORMCDF) return 0;  // closed forms [GPLikelihoods AnalyticExpectation]
  return 20;  // DefaultExpectationMethod -> GaussHermiteExpectation(20)  [GPLikelihoods]
}

int upload_params(svgp_ctx* ctx, svgp_model* m, const svgp_model_desc* d) {
  const size_t es = m->es;
  m->desc = *d;
  m->invl_host.assign(d->inv_lengthscale, d->inv_lengthscale + d->d);
  m->desc.inv_lengthscale = m->invl_host.data();
  m->desc.z = m->desc.m = m->desc.Lq = nullptr;  // host pointers are borrowed for the call only
  HIPC(ctx, hipMemcpyAsync(m->z_raw, d->z, size_t(m->M) * m->d * es, hipMemcpyHostToDevice, ctx->stream));
  HIPC(ctx, hipMemcpyAsync(m->m_raw, d->m, size_t(m->M) * es, hipMemcpyHostToDevice, ctx->stream));
  HIPC(ctx, hipMemcpyAsync(m->Lq_raw, d->Lq, size_t(m->M) * m->M * es, hipMemcpyHostToDevice, ctx->stream));
  if (m->dtype == SVGP_F64) {
    HIPC(ctx, hipMemcpyAsync(m->invl, m->invl_host.data(), m->d * 8, hipMemcpyHostToDevice, ctx->stream));
  } else {
    std::vector<float> f(m->invl_host.begin(), m->invl_host.end());
    HIPC(ctx, hipMemcpyAsync(m->invl, f.data(), m->d * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPC(ctx, hipStreamSynchronize(ctx->stream));
  }
  const int gh = effective_gh(*d);
  if (gh != m->gh_n) {
    if (m->gh_x) (void)hipFree(m->gh_x);
    if (m->gh_w) (void)hipFree(m->gh_w);
    m->gh_x = m->gh_w = nullptr;
    m->gh_n = gh;
    if (gh > 0) {
      std::vector<double> xs(gh), ws(gh);
      if (gauss_hermite(gh, xs.data(), ws.data()) != SVGP_OK) return fail(ctx, SVGP_INVALID_ARG, "Gauss-Hermite rule failed to converge");
      for (auto& w : ws) w /= 1.7724538509055160273;
      HIPC(ctx, hipMalloc(&m->gh_x, gh * 8));
      HIPC(ctx, hipMalloc(&m->gh_w, gh * 8));
      HIPC(ctx, hipMemcpy(m->gh_x, xs.data(), gh * 8, hipMemcpyHostToDevice));
      HIPC(ctx, hipMemcpy(m->gh_w, ws.data(), gh * 8, hipMemcpyHostToDevice));
    }
  }
  HIPC(ctx, hipStreamSynchronize(ctx->stream));  // host buffers may be released by the caller
  m->prepared = false;
  return SVGP_OK;
}

KernelParams kparams(const svgp_model* m) {
  KernelParams kp;
  kp.family = m->desc.kernel;
  kp.d = m->d;
  kp.variance = m->desc.variance;
  kp.invl = m->invl;
  return kp;
}

// The M-sized work of posterior(sva): enqueue only, no sync.
// overlap (NonCentered only): everything the strips need besides T - the scaled inducing inputs, U = Lq', the padded mean - is
// enqueued FIRST and ctx->ev_fork recorded behind it; the factorisation then records ctx->ev_row[p] as block row p of T becomes
// final, so that strips on a second stream can run beside it (SegRun: seg_enqueue_row).
// info + the factorisation's hand-over counters and flags: 1 + 2 nP ints, rounded up to 256 bytes - a memset whose size is not a
// multiple of 16 bytes becomes TWO fill kernels (aligned body + tail), ~5 us of every call's prologue
inline size_t info_bytes(int64_t Mp) { return (sizeof(int) * size_t(1 + 2 * (Mp / 128)) + 255) / 256 * 256; }
int ensure_overlap(svgp_ctx* ctx, size_t state_doubles); int ensure_overlap(svgp_ctx* ctx, size_t state_doubles);   // (below) second stream + the row events
int enqueue_prep(svgp_ctx* ctx, svgp_model* m, bool overlap = false, const RowHook* hook = nullptr) {
  hipStream_t s = ctx->stream;
  const KernelParams kp = kparams(m);
  HIPC(ctx, hipMemsetAsync(m->info, 0, info_bytes(m->Mp), s));   // info + the factorisation's hand-over counters and flags
  launch_scale_inputs(m->dtype, s, m->z_raw, m->desc.layout_z, m->d, m->M, m->Mp, m->invl, m->zs);
  KCHECK(ctx, "scale_inputs");
  if (overlap) {
    launch_pack_q(m->dtype, s, m->Lq_raw, m->m_raw, m->M, m->Mp, m->U, m->mp);          // B = Lq      SVA:183-184
    KCHECK(ctx, "pack_q");
    HIPC(ctx, hipEventRecord(ctx->ev_fork, s));
    TREC(ctx, ctx->ev_ov[0], s);
    if (hook && hook->fn) hook->fn(hook->user, -1);   // the strips' pre-generation: behind the fork, before the chain is enqueued
  }
  launch_kuu(m->dtype, s, kp, m->zs, m->M, m->Mp, m->desc.jitter, m->L);
  KCHECK(ctx, "kuu");
  TREC(ctx, ctx->ev_chol[0], s);
  launch_potrf(m->dtype, s, m->L, m->T, m->Mp, m->info, reinterpret_cast<unsigned*>(m->info + 1), ctx->num_cus, overlap ? ctx->ev_row : nullptr,
               overlap ? hook : nullptr);   // T panels included
  KCHECK(ctx, "potrf");
  TREC(ctx, ctx->ev_chol[1], s);
  if (overlap) {
    launch_kl_terms(m->dtype, s, m->Lq_raw, m->m_raw, m->L, m->M, m->Mp, m->scal);        // SVA:364-373
  } else if (m->desc.parametrization == SVGP_NONCENTERED) {
    launch_pack_q(m->dtype, s, m->Lq_raw, m->m_raw, m->M, m->Mp, m->U, m->mp);          // B = Lq      SVA:183-184
    launch_kl_terms(m->dtype, s, m->Lq_raw, m->m_raw, m->L, m->M, m->Mp, m->scal);        // SVA:364-373
  } else {
    // Centered (SVA:115-136): B = Lk \ Lq, whitened mean m~ = Lk \ (m - mean(fz)); then α = Lk' \ m~ = Kuu \ (m - μ)
    // and KL(q || p(u)) = ½(ΣB² + m~'m~ − M − logdet(BB')) — the NonCentered expression evaluated at (m~, B).
    launch_pad_lower(m->dtype, s, m->Lq_raw, m->M, m->Mp, m->B);
    launch_trsm_mat(m->dtype, s, m->T, m->Mp, m->B);
    launch_shift_vec(m->dtype, s, m->m_raw, -m->desc.mean_const, m->M, m->Mp, m->mp);
    launch_trsv2(m->dtype, s, m->L, m->T, m->Mp, 0, m->mp);
    launch_pack_q_ld(m->dtype, s, m->B, m->Mp, nullptr, m->Mp, m->Mp, m->U, nullptr);
    launch_kl_terms_ld(m->dtype, s, m->B, m->Mp, m->mp, m->L, m->M, m->Mp, m->scal);
  }
  KCHECK(ctx, "pack_q/kl");
  return SVGP_OK;
}

// scalars of the last prep -> host (requires a stream sync by the caller before use)
struct PrepScalars { double scal[4]; int info; };

void finish_prep(svgp_model* m, const PrepScalars& ps) {
  m->chol_info = ps.info;
  m->kl = 0.5 * (ps.scal[0] + ps.scal[1] - double(m->M) - 2.0 * ps.scal[2]);
  m->logdet_kuu = 2.0 * ps.scal[3];
  m->prepared = (ps.info == 0);
}

int ensure_scratch(svgp_ctx* ctx, size_t work_bytes, size_t npoints) {
  if (work_bytes > ctx->work_bytes) {
    if (ctx->work) (void)hipFree(ctx->work);
    ctx->work = nullptr;
    ctx->work_bytes = 0;
    HIPC(ctx, hipMalloc(&ctx->work, work_bytes));
    ctx->work_bytes = work_bytes;
  }
  if (npoints > ctx->mom_cap) {
    if (ctx->mom) (void)hipFree(ctx->mom);
    ctx->mom = nullptr;
    ctx->mom_cap = 0;
    HIPC(ctx, hipMalloc(&ctx->mom, 2 * npoints * sizeof(double)));
    ctx->mom_cap = npoints;
  }
  return SVGP_OK;
}

// second stream (+ fork / join events, its own strip queue head) for launches that run BESIDE the main stream's
int ensure_stream2(svgp_ctx* ctx) {
  if (ctx->stream2) return SVGP_OK;
  // the second stream carries work that runs BESIDE the main stream's (the ragged tail of a large batch, the segmented strips beside
  // the factorisation): lowest priority, so that where both have workgroups to dispatch the main stream's serial chain goes first
  static const int prio_knob = exp_int("SVGP_STREAM2_LOW_PRIO", 1);   // A/B knob (experiments build)
  int least = 0, greatest = 0;
  // A/B knob (round 4): keep SVGP_STREAM2_RESERVE CUs out of the second stream's reach (hipExtStreamCreateWithCUMask), so that the
  // factorisation's launches always find free CUs beside the segmented strips
  static const int reserve = exp_int("SVGP_STREAM2_RESERVE", 0);
  if (reserve > 0 && reserve < ctx->num_cus) {
    std::vector<uint32_t> mask(size_t((ctx->num_cus + 31) / 32), 0xffffffffu);
    for (int c = 0; c < reserve; ++c) {   // spread the reserved CUs: one every num_cus / reserve
      const int cu = int((int64_t(c) * ctx->num_cus) / reserve);
      mask[size_t(cu / 32)] &= ~(1u << (cu % 32));
    }
    HIPC(ctx, hipExtStreamCreateWithCUMask(&ctx->stream2, uint32_t(mask.size()), mask.data()));
  } else if (prio_knob && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
    HIPC(ctx, hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, least));
  } else {
    HIPC(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  }
  HIPC(ctx, hipEventCreateWithFlags(&ctx->ev_fork, kSyncEvent));
  HIPC(ctx, hipEventCreateWithFlags(&ctx->ev_join, kSyncEvent));
  HIPC(ctx, hipMalloc(&ctx->counter2, 64));
  return SVGP_OK;
}

// split closing launch of small batches (kernels.hpp: seg_split): at most kSegSplitSlots workgroups, each leaving 3 x NT <= 384 doubles;
// then one counter per strip
constexpr size_t kSegSplitSlots = 512, kSegSplitPart = 384, kSegSplitDoubles = kSegSplitSlots * kSegSplitPart + kSegSplitSlots;
// S = the largest power of two with S <= nP and nstrips S <= kSegSplitSlots - if that is at least 4: two workgroups per strip do not
// pay for the extra launch (measured, profiles/round4/minibatch_step.md: 8192 points / M = 1024 f64 forward 0.84 -> 0.87 ms, gradient
// 2.29 -> 2.28; 4096 points 0.81 -> 0.78 / 2.13 -> 2.00; 1024 points 0.80 -> 0.71 / 2.05 -> 1.78).  SVGP_SEG_SPLIT=0 (at context creation): never split
int seg_split_factor(const svgp_ctx* ctx, int nP, int64_t nstrips) {
  int S = 1;
  if (ctx->kn.seg_split)
    while (2 * S <= nP && int64_t(2 * S) * nstrips <= int64_t(kSegSplitSlots)) S *= 2;
  return S >= 4 ? S : 1;
}
// points the split closing launch at its partials and counters (behind the saved sums) and zeroes the counters on the second stream
int seg_split_setup(svgp_ctx* ctx, StripArgs& a, int S, int64_t nstrips) {
  double* extra = ctx->seg_state + (ctx->seg_state_doubles - kSegSplitDoubles);
  a.seg_split = S;
  a.seg_part = extra;
  a.seg_cnt = reinterpret_cast<unsigned*>(extra + kSegSplitSlots * kSegSplitPart);
  HIPC(ctx, hipMemsetAsync(a.seg_cnt, 0, size_t(nstrips) * sizeof(unsigned), ctx->stream2));
  return SVGP_OK;
}
int ensure_overlap(svgp_ctx* ctx, size_t state_doubles) {
  int rc = ensure_stream2(ctx);
  if (rc) return rc;
  if (!ctx->ev_row_ready) {
    for (auto& e : ctx->ev_row)
      if (!e) HIPC(ctx, hipEventCreateWithFlags(&e, kSyncEvent));
    for (auto& e : ctx->ev_ov)
      if (!e) HIPC(ctx, hipEventCreateWithFlags(&e, kTimingEvent));
    if (!ctx->ev_R) HIPC(ctx, hipEventCreateWithFlags(&ctx->ev_R, kSyncEvent));
    if (!ctx->ev_S) HIPC(ctx, hipEventCreateWithFlags(&ctx->ev_S, kSyncEvent));
    ctx->ev_row_ready = true;
  }
  state_doubles += kSegSplitDoubles;   // behind the saved sums: the split closing launch's partials and its per-strip counters
  if (state_doubles > ctx->seg_state_doubles) {
    if (ctx->seg_state) (void)hipFree(ctx->seg_state);
    ctx->seg_state = nullptr;
    ctx->seg_state_doubles = 0;
    HIPC(ctx, hipMalloc(&ctx->seg_state, state_doubles * sizeof(double)));
    ctx->seg_state_doubles = state_doubles;
  }
  return SVGP_OK;
}

struct StripOuts {
  void* mu = nullptr;
  void* var = nullptr;
  void* A = nullptr;
  void* C = nullptr;
  void* At = nullptr;
  void* Ct = nullptr;
  int64_t lda = 0;
  bool skip_expect = false;   // svgp_marginals: the caller wants the moments themselves
  int64_t mom_shift = 0;      // the batch's moments start at this index of the context's moment arrays (a batch evaluated in two parts)
  bool no_ctail = false;      // never a concurrent tail launch (the second stream is taken: segmented head)
};

// enqueue the fused strip kernel + final reduce over points [off, off+len) of (x, y)
int enqueue_strips(svgp_ctx* ctx, svgp_model* m, const void* x, int64_t ldx, const void* y, int64_t off, int64_t len,
                   const StripOuts& o) {
  StripPlan plan = strip_plan(m->dtype, m->Mp, len, ctx->num_cus);
  if (plan.concurrent_tail && (o.A || o.C || o.At || o.Ct || o.no_ctail)) plan = strip_plan_single(m->dtype, m->Mp, len, ctx->num_cus);
  const size_t wb_main = plan.grid ? strip_work_bytes(m->dtype, m->Mp, plan.nt, plan.grid) : 0;
  const size_t wb_tail = plan.nt_tail ? strip_work_bytes(m->dtype, m->Mp, plan.nt_tail, plan.grid_tail) : 0;
  int rc = ensure_scratch(ctx, plan.concurrent_tail ? wb_main : (wb_main > wb_tail ? wb_main : wb_tail), size_t(len + o.mom_shift));
  if (rc) return rc;
  if (plan.concurrent_tail) {
    rc = ensure_stream2(ctx);
    if (rc) return rc;
    if (wb_tail > ctx->work2_bytes) {
      if (ctx->work2) (void)hipFree(ctx->work2);
      ctx->work2 = nullptr;
      ctx->work2_bytes = 0;
      HIPC(ctx, hipMalloc(&ctx->work2, wb_tail));
      ctx->work2_bytes = wb_tail;
    }
  }
  StripArgs a{};
  a.T = m->T;
  a.U = m->U;
  a.zs = m->zs;
  a.mp = m->mp;
  a.x = x;
  a.work = ctx->work;
  a.counter = ctx->counter;
  a.mom_mu = ctx->mom + o.mom_shift;
  a.mom_var = ctx->mom + ctx->mom_cap + o.mom_shift;
  a.A_out = o.A;
  a.C_out = o.C;
  a.At_out = o.At;
  a.Ct_out = o.Ct;
  a.lda = o.lda;
  a.ldx = ldx;
  a.off = off;
  a.len = len;
  a.Mp = m->Mp;
  a.M = m->M;
  a.kp = kparams(m);
  a.mean_const = m->desc.mean_const;
  LikParams lp{};
  lp.lik = m->desc.likelihood;
  lp.gh_n = m->gh_n;
  lp.sigma2 = lik_param(m->desc);
  lp.digamma_alpha = m->desc.likelihood == SVGP_LIK_GAMMA_EXP ? digamma_d(m->desc.lik_sigma2) : 0.0;
  lp.gh_x = m->gh_x;
  lp.gh_w = m->gh_w;
  lp.clamp_neg_var = (m->desc.neg_var_policy == SVGP_NEGVAR_CLAMP);
  lp.mean_const = m->desc.mean_const;
  if (plan.concurrent_tail) HIPC(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  if (plan.grid) {
    StripArgs am = a;
    am.len = plan.points;
    HIPC(ctx, hipMemsetAsync(ctx->counter, 0, sizeof(unsigned), ctx->stream));
    launch_strip(m->dtype, ctx->stream, am, plan.nt, plan.grid, plan.nstrips);
    KCHECK(ctx, "strip");
  }
  if (plan.nt_tail) {   // remaining points as half-width strips: same arithmetic per point, outputs shifted by plan.points
    StripArgs at = a;
    const int64_t sh = plan.points;
    const size_t es = m->es;
    at.off = off + sh;
    at.len = len - sh;
    at.mom_mu = a.mom_mu + sh;
    at.mom_var = a.mom_var + sh;
    if (a.A_out) at.A_out = static_cast<char*>(a.A_out) + size_t(sh) * es;
    if (a.C_out) at.C_out = static_cast<char*>(a.C_out) + size_t(sh) * es;
    if (a.At_out) at.At_out = static_cast<char*>(a.At_out) + size_t(sh) * size_t(m->Mp) * es;
    if (a.Ct_out) at.Ct_out = static_cast<char*>(a.Ct_out) + size_t(sh) * size_t(m->Mp) * es;
    if (plan.concurrent_tail) {
      // fork: the tail may start as soon as everything before this point on the main stream is done (the prep kernels,
      // recorded BEFORE the main launch); join: the main stream continues after both launches
      at.work = ctx->work2;
      at.counter = ctx->counter2;
      HIPC(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
      HIPC(ctx, hipMemsetAsync(ctx->counter2, 0, sizeof(unsigned), ctx->stream2));
      launch_strip(m->dtype, ctx->stream2, at, plan.nt_tail, plan.grid_tail, plan.nstrips_tail);
      KCHECK(ctx, "strip tail (concurrent)");
      HIPC(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
      HIPC(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    } else {
      HIPC(ctx, hipMemsetAsync(ctx->counter, 0, sizeof(unsigned), ctx->stream));
      launch_strip(m->dtype, ctx->stream, at, plan.nt_tail, plan.grid_tail, plan.nstrips_tail);
      KCHECK(ctx, "strip tail");
    }
  }
  ctx->timing.strip_launches = (plan.grid ? 1 : 0) + (plan.nt_tail ? 1 : 0);
  if (o.mom_shift) return SVGP_OK;   // the second part of a batch: the caller joins the parts and runs the expectation over both
  TREC(ctx, ctx->ev[2], ctx->stream);
  if (o.skip_expect) return SVGP_OK;
  launch_expect(m->dtype, ctx->stream, lp, a.mom_mu, a.mom_var, y, off, len, ctx->partial, ctx->negcnt, o.mu, o.var);
  KCHECK(ctx, "expect");
  launch_final_reduce(ctx->stream, ctx->partial, ctx->negcnt, expect_blocks(len), m->info, double(len), ctx->d_res, m->scal);
  KCHECK(ctx, "final_reduce");
  HIPC(ctx, hipGetLastError());
  return SVGP_OK;
}

// ---- prep beside the strips (round 4) --------------------------------------------------------------------
// The M-sized prep is a serial chain of ~2 nP launches (0.43 ms at M = 1024 whatever the batch) and minibatches are where real
// callers live.  Phase 1 of panel I needs only block row I of T, final after panel step I of the factorisation.  For a batch of
// at most one round of strips the evaluation therefore runs as SEGMENTED strips (strip.hip: SEG) on the second stream: the Kuf
// pre-generation right away, phase-1 panel I behind ev_row[I], phase 2 with the last panel; the main stream joins before the
// expectation.  Nothing spins and no launch waits while resident, so the factorisation's launches always find free CUs.
struct OverlapPlan { bool on = false; int nt = 0, grid = 0; int64_t nstrips = 0, head_points = 0; };

OverlapPlan overlap_plan(const svgp_ctx* ctx, const svgp_model* m, int64_t len, const StripOuts& o) {
  OverlapPlan p;
  if (!ctx->kn.overlap) return p;   // SVGP_OVERLAP=0 at context creation
  const int nP = int(m->Mp / 128);
  // measured (profiles/round4/overlap.md, f64, one round of strips): M = 256 +5 %, 512 0..-4 %, 1024 -9..-11 %, 2048 -19 %: the
  // 2 nP extra launches and the chain's slowdown beside the strips are paid back from about six panels on
  const int min_panels = ctx->kn.overlap_min_panels;   // 5 (SVGP_OVERLAP_MIN_PANELS in the experiments build)
  if (m->desc.parametrization != SVGP_NONCENTERED || m->d > 16 || nP < 2 || nP > potrf_max_row_events()) return p;
  if (o.A || o.C || o.At || o.Ct) return p;
  const StripPlan sp = strip_plan_single(m->dtype, m->Mp, len, ctx->num_cus);
  if (sp.grid && sp.nt_tail) return p;
  p.nt = sp.grid ? sp.nt : sp.nt_tail;
  p.nstrips = sp.grid ? sp.nstrips : sp.nstrips_tail;
  p.grid = sp.grid ? sp.grid : sp.grid_tail;
  // the segmented kernels exist for 32- / 64-point strips (f64) and 32 / 64 / 128 (f32) on 256 threads (launch_strip_seg); a wider plan -
  // only an experiments build can ask for one (SVGP_STRIP_NT=128) - keeps the one-launch path
  if (p.nt > (m->dtype == SVGP_F64 ? 64 : 128)) return p;
  if (p.nstrips <= p.grid && nP < min_panels) return p;   // one round: pays from about five panels on (multi-round heads: below)
  p.head_points = len;
  if (p.nstrips > p.grid) {
    // More than one round of strips (C5: 4, C2: 3): a SEGMENTED HEAD - the first round's worth of strips runs beside the
    // factorisation like a one-round batch, the rest as the one-launch kernel behind the prep on the main stream (its dynamic queue
    // intact), the two joined before the expectation.  The head's phase 1 fills the chip while the chain would have had it alone.
    // MEASURED, default OFF (profiles/round4/overlap.md): C5 (4 rounds of fp32 strips) 5.02 -> 4.91 ms wall, C2 (3 rounds, M = 512) 1.339
    // -> 1.340, H32 / C3 unchanged - the head's panel launches are long (a full round of full-width strips) and the factorisation
    // waits for their CUs (C5 prep 0.48 -> 0.84 ms), which gives back most of what the head gains.  SVGP_OVERLAP_HEAD=1 enables it
    // in the experiments build; the product build never takes this branch.
    const int64_t rounds = (p.nstrips + p.grid - 1) / p.grid;
    if (!ctx->kn.overlap_head || !sp.grid || nP < ctx->kn.overlap_head_min_panels || rounds > 8) return OverlapPlan{};
    p.nstrips = p.grid;
    p.head_points = int64_t(p.grid) * p.nt;
  }
  p.on = true;
  return p;
}

// The segmented strips of ONE evaluation.  Their launches wait for events the factorisation records panel by panel, and a wait can
// only be enqueued BEHIND its record - so they are enqueued from inside the factorisation's launch loop (RowHook), each right behind
// the record it waits for.  (Until late in round 4 they were enqueued after the whole prep: the host needs 5-10 us per launch, the ~25
// launches of the prep took it longer than the device needed to start the chain, and the strips' first panel reached the second stream
// when the chain was already a third - in the gradient, where the M-sized adjoint prep is enqueued first as well, completely - done.)
struct SegRun {
  svgp_ctx* ctx = nullptr;
  svgp_model* m = nullptr;
  StripArgs a{};
  OverlapPlan op;
  bool grad = false;
  int split = 1;   // forward: phase 2 in a closing launch of its own, `split` workgroups per strip (small batches; kernels.hpp: seg_split)
  int nP = 0, rc = SVGP_OK;
  size_t wb1 = 0;
  int64_t head = 0;
  std::function<int()>* pre = nullptr;   // work for the second stream ahead of the pre-generation (the gradient's chain-independent prep)
};

// row -1: the Kuf pre-generation (behind ev_fork); row I >= 0: phase-1 panel I (behind ev_row[I]); the forward's last panel carries
// phase 2, the gradient's closing launch (phase 2 + 3, behind ev_R) is enqueued by grad_enqueue_impl
int seg_enqueue_row(SegRun& r, int row) {
  svgp_ctx* ctx = r.ctx;
  hipStream_t s2 = ctx->stream2;
  StripArgs& a = r.a;
  const int dt = r.m->dtype;
  if (row < 0) {
    HIPC(ctx, hipStreamWaitEvent(s2, ctx->ev_fork, 0));
    a.seg_flags = kSegPregen; a.seg_lo = 0; a.seg_hi = 0;
    launch_strip_seg(dt, s2, a, r.op.nt, r.op.grid, r.op.nstrips, r.grad);
    KCHECK(ctx, "strip (segmented: pre-generation)");
    return SVGP_OK;
  }
  const int I = row, nP = r.nP;
  HIPC(ctx, hipStreamWaitEvent(s2, ctx->ev_row[I], 0));
  a.seg_lo = I; a.seg_hi = I + 1;
  a.seg_flags = (I > 0 ? kSegLoad : 0) | ((r.grad || r.split > 1 || I + 1 < nP) ? kSegStore : kSegPhase2);
  launch_strip_seg(dt, s2, a, r.op.nt, r.op.grid, r.op.nstrips, r.grad);
  KCHECK(ctx, "strip (segmented: panel)");
  if (I == 0) TREC(ctx, ctx->ev_ov[1], s2);
  // The gradient's chain-independent prep goes behind the strips of panel 1: its ~10 launches are ~55 us of HOST time, and enqueued
  // any earlier (it sat in front of the pre-generation at first) they kept the host from enqueueing the factorisation's first launches -
  // the main stream idled for 45 us waiting for its Kuu kernel (kernel trace, profiles/round4/minibatch_step.md section 7)
  if (I == 1 && r.pre) {
    const int rcp = (*r.pre)();
    if (rcp) return rcp;
  }
  return SVGP_OK;
}

void seg_row_hook(void* user, int row) {
  SegRun& r = *static_cast<SegRun*>(user);
  if (r.rc == SVGP_OK) r.rc = seg_enqueue_row(r, row);
}

// forward evaluation, stage 1 (BEFORE the prep is enqueued): every allocation and the strips' arguments
int seg_prepare_forward(svgp_ctx* ctx, svgp_model* m, const void* x, int64_t ldx, int64_t off, int64_t len, const OverlapPlan& op, SegRun& r) {
  const int nP = int(m->Mp / 128);
  const int64_t head = op.head_points < len ? op.head_points : len;
  // the segmented strips' scratch is per STRIP and lives beside the main launch's per-workgroup scratch (a segmented head runs
  // concurrently with the rest of its batch): its own buffer
  const size_t wb1 = strip_work_bytes(m->dtype, m->Mp, op.nt, int(op.nstrips)), wb = wb1;
  // (A checkpointed phase 2 beside the factorisation - bitwise-tested, measured in round 4, profiles/round4/overlap.md: the factorisation
  // slows by what the strips' extra work beside it occupies, 16 384 / 1024 f64 1.046 -> 1.06-1.07 ms, M = 2048 1.94 -> 2.11 - left the
  // tree in round 6.)
  if (wb > ctx->work_seg_bytes) {
    if (ctx->work_seg) (void)hipFree(ctx->work_seg);
    ctx->work_seg = nullptr;
    ctx->work_seg_bytes = 0;
    HIPC(ctx, hipMalloc(&ctx->work_seg, wb));
    ctx->work_seg_bytes = wb;
  }
  int rc = ensure_scratch(ctx, 0, size_t(len));
  if (rc) return rc;
  if (head < len) {   // the part behind the head: sized and allocated BEFORE anything is enqueued (ensure_scratch may reallocate)
    const StripPlan rp = strip_plan_single(m->dtype, m->Mp, len - head, ctx->num_cus);
    const size_t w1 = rp.grid ? strip_work_bytes(m->dtype, m->Mp, rp.nt, rp.grid) : 0;
    const size_t w2 = rp.nt_tail ? strip_work_bytes(m->dtype, m->Mp, rp.nt_tail, rp.grid_tail) : 0;
    rc = ensure_scratch(ctx, w1 > w2 ? w1 : w2, size_t(len));
    if (rc) return rc;
  }
  r.ctx = ctx; r.m = m; r.op = op; r.grad = false; r.nP = nP; r.wb1 = wb1; r.head = head;
  // A small batch (fewer strips than workgroup slots; not a segmented head): phase 2 - one panel C_J after the other inside a strip's
  // workgroup - is latency-bound on the few CUs it reaches, and its panels are independent: a closing launch with S workgroups per strip
  r.split = (head < len) ? 1 : seg_split_factor(ctx, nP, op.nstrips);
  StripArgs& a = r.a;
  a.T = m->T; a.U = m->U; a.zs = m->zs; a.mp = m->mp; a.x = x; a.work = ctx->work_seg; a.counter = ctx->counter2;
  a.mom_mu = ctx->mom; a.mom_var = ctx->mom + ctx->mom_cap;
  a.ldx = ldx; a.off = off; a.len = head; a.Mp = m->Mp; a.M = m->M; a.kp = kparams(m); a.mean_const = m->desc.mean_const;
  a.seg_state = ctx->seg_state;
  return SVGP_OK;
}

// forward evaluation, stage 2 (AFTER the prep, whose launch loop enqueued the segments): the rest of the batch, the join, the expectation
int seg_finish_forward(svgp_ctx* ctx, svgp_model* m, const void* x, int64_t ldx, const void* y, int64_t off, int64_t len,
                       const StripOuts& o, SegRun& r) {
  hipStream_t s = ctx->stream, s2 = ctx->stream2;
  if (r.rc) return r.rc;
  const int64_t head = r.head;
  int launches = r.nP + 1;
  if (r.split > 1) {   // phase 2 + moments, r.split workgroups per strip
    StripArgs& a = r.a;
    const int rcs = seg_split_setup(ctx, a, r.split, r.op.nstrips);
    if (rcs) return rcs;
    a.seg_lo = a.seg_hi = r.nP;
    a.seg_flags = kSegLoad | kSegPhase2;
    launch_strip_seg(m->dtype, s2, a, r.op.nt, int(r.op.nstrips) * r.split, r.op.nstrips, false);
    KCHECK(ctx, "strip (segmented: split phase 2)");
    ++launches;
  }
  HIPC(ctx, hipEventRecord(ctx->ev_join, s2));
  if (head < len) {   // the rest of the batch: the one-launch kernel behind the prep, on the main stream, beside the head's tail
    StripOuts rest = o;
    rest.mom_shift = head;
    rest.no_ctail = true;
    rest.skip_expect = true;
    const int rc = enqueue_strips(ctx, m, x, ldx, y, off + head, len - head, rest);
    if (rc) return rc;
    launches += int(ctx->timing.strip_launches);
  }
  HIPC(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
  TREC(ctx, ctx->ev[2], s);
  ctx->timing.strip_launches = launches;
  if (o.skip_expect) return SVGP_OK;
  LikParams lp{};
  lp.lik = m->desc.likelihood;
  lp.gh_n = m->gh_n;
  lp.sigma2 = lik_param(m->desc);
  lp.digamma_alpha = m->desc.likelihood == SVGP_LIK_GAMMA_EXP ? digamma_d(m->desc.lik_sigma2) : 0.0;
  lp.gh_x = m->gh_x;
  lp.gh_w = m->gh_w;
  lp.clamp_neg_var = (m->desc.neg_var_policy == SVGP_NEGVAR_CLAMP);
  lp.mean_const = m->desc.mean_const;
  launch_expect(m->dtype, s, lp, r.a.mom_mu, r.a.mom_var, y, off, len, ctx->partial, ctx->negcnt, o.mu, o.var);
  KCHECK(ctx, "expect");
  launch_final_reduce(s, ctx->partial, ctx->negcnt, expect_blocks(len), m->info, double(len), ctx->d_res, m->scal);
  KCHECK(ctx, "final_reduce");
  HIPC(ctx, hipGetLastError());
  return SVGP_OK;
}

int check_batch(svgp_ctx* ctx, const svgp_model* m, const svgp_data* data, int64_t off, int64_t len, bool need_y) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (!m || !data) return fail(ctx, SVGP_INVALID_ARG, "null model or data");
  if (data->dtype != m->dtype) return fail(ctx, SVGP_INVALID_ARG, "data and model dtypes differ");
  if (data->d != m->d) return fail(ctx, SVGP_INVALID_ARG, "data and model input dimensions differ");
  if (off < 0 || len < 1 || off + len > data->n) return fail(ctx, SVGP_INVALID_ARG, "batch range outside the data");
  if (need_y && !data->y) return fail(ctx, SVGP_INVALID_ARG, "data has no observations y");
  return SVGP_OK;
}

// One evaluation = three stages, so that a process driving several GPUs (svgp_group_*) can enqueue every device before
// it waits for any:  elbo_enqueue (prep + strips + reduce, asynchronous)  ->  elbo_collective (ONE ncclAllReduce of the
// device-resident 8-vector d_res on the context's stream; nothing without a communicator)  ->  elbo_finish (read back).
struct ElboRead {
  double E = 0, n_points = 0, n_neg = 0, bad_chol = 0, failed = 0;
};

int elbo_enqueue(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len) {
  hipStream_t s = ctx->stream;
  HIPC(ctx, hipSetDevice(ctx->device));
  const OverlapPlan op = overlap_plan(ctx, m, len, StripOuts{});
  int rc = SVGP_OK;
  if (op.on) {
    rc = ensure_overlap(ctx, size_t(op.nstrips) * strip_seg_state_doubles(m->dtype, op.nt));
    if (rc) return rc;
  }
  ctx->overlapped = op.on;
  // (experiments build) diagnostic: the overlap's prep - reordered, with its row events recorded - but the strips behind it as usual:
  // what the events alone cost the chain
  const bool dry = op.on && ctx->kn.overlap_dry == 1;
  SegRun seg;
  RowHook hook{seg_row_hook, &seg};
  if (op.on && !dry) {
    rc = seg_prepare_forward(ctx, m, data->x, data->ldx, off, len, op, seg);
    if (rc) return rc;
  }
  TREC(ctx, ctx->ev[0], s);
  rc = enqueue_prep(ctx, m, op.on, (op.on && !dry) ? &hook : nullptr);
  if (rc == SVGP_OK && op.on && !dry) rc = seg.rc;
  if (rc) {   // a failure between the fork and the join leaves work on the second stream that the main stream never waited for: drain it,
              // so that the next call on this context cannot meet it in the shared scratch
    if (op.on && ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    return rc;
  }
  TREC(ctx, ctx->ev[1], s);
  if (dry) {
    ctx->overlapped = false;   // ev_ov[1] is not recorded on this path
    rc = enqueue_strips(ctx, m, data->x, data->ldx, data->y, off, len, StripOuts{});
    if (rc) return rc;
    TREC(ctx, ctx->ev[3], s);
    return SVGP_OK;
  }
  rc = op.on ? seg_finish_forward(ctx, m, data->x, data->ldx, data->y, off, len, StripOuts{}, seg)
             : enqueue_strips(ctx, m, data->x, data->ldx, data->y, off, len, StripOuts{});
  if (rc) {
    if (op.on && ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    return rc;
  }
  TREC(ctx, ctx->ev[3], s);
  return SVGP_OK;
}

// `local_rc` is the status of this rank's enqueue: a rank that failed still takes part in the collective, with its
// failure flag set, so that its peers return an error instead of waiting for it forever (they all see failed > 0).
int elbo_collective(svgp_ctx* ctx, int local_rc) {
  if (!ctx->comm) return local_rc;
  if (local_rc != SVGP_OK) {
    const std::string keep = ctx->err;
    static const double poison[8] = {0, 0, 0, 0, 1, 0, 0, 0};
    if (hipSetDevice(ctx->device) != hipSuccess ||
        hipMemcpyAsync(ctx->d_res, poison, sizeof poison, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
      comm_abort(ctx);
      ctx->err = keep;
      return local_rc;
    }
    ctx->err = keep;
  }
  const int rc = comm_allreduce(ctx, ctx->d_res, 8, SVGP_F64);
  if (rc != SVGP_OK) {
    comm_abort(ctx);
    return local_rc != SVGP_OK ? local_rc : rc;
  }
  return local_rc;
}

int elbo_finish(svgp_ctx* ctx, svgp_model* m, ElboRead* out) {
  hipStream_t s = ctx->stream;
  HIPC(ctx, hipSetDevice(ctx->device));
  double res[13];   // {sum E, n, n_neg, chol flag, failure flag, 0, 0, 0 | prep scalars (4), chol_info}: one copy
  PrepScalars ps;
  HIPC(ctx, hipMemcpyAsync(res, ctx->d_res, sizeof(res), hipMemcpyDeviceToHost, s));
  HIPC(ctx, hipStreamSynchronize(s));
  for (int q = 0; q < 4; ++q) ps.scal[q] = res[8 + q];
  ps.info = int(res[12]);
  finish_prep(m, ps);
  float t01 = 0, t12 = 0, t23 = 0;
  (void)elapsed_ms(ctx, &t01, ctx->ev[0], ctx->ev[1]);
  (void)elapsed_ms(ctx, &t12, ctx->ev[1], ctx->ev[2]);
  (void)elapsed_ms(ctx, &t23, ctx->ev[2], ctx->ev[3]);
  ctx->timing.ms_prep = t01;
  ctx->timing.ms_strip = t12;
  ctx->timing.ms_expect = t23;
  ctx->timing.ms_total = t01 + t12 + t23;
  ctx->timing.ms_kuf = 0;
  float tch = 0;
  (void)elapsed_ms(ctx, &tch, ctx->ev_chol[0], ctx->ev_chol[1]);
  ctx->timing.ms_chol = tch;
  ctx->timing.ms_overlap = 0;
  if (ctx->overlapped) {   // the part of the prep the strips ran beside: from their first launch's completion to the prep's end
    float tov = 0;
    if (elapsed_ms(ctx, &tov, ctx->ev_ov[1], ctx->ev[1]) == hipSuccess && tov > 0) ctx->timing.ms_overlap = tov;
  }
  out->E = res[0];
  out->n_points = res[1];
  out->n_neg = res[2];
  out->bad_chol = res[3];
  out->failed = res[4];
  return SVGP_OK;
}

// prep + strips (+ collective) + readback; refreshes the model's prep scalars
int run_elbo(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len, bool collective, ElboRead* out) {
  int rc = elbo_enqueue(ctx, m, data, off, len);
  if (collective) rc = elbo_collective(ctx, rc);
  if (rc) return rc;
  return elbo_finish(ctx, m, out);
}

int status_of(svgp_ctx* ctx, const svgp_model* m, double nneg, double bad_chol = 0, double failed = 0) {
  if (failed > 0)
    return fail(ctx, SVGP_RCCL_ERROR, "a peer rank failed before the collective; the data-parallel evaluation was abandoned on every rank");
  if (m->chol_info != 0 || bad_chol > 0) {
    char buf[160];
    snprintf(buf, sizeof buf, "Kuu is not positive definite: leading minor of order %d (PosDefException)", m->chol_info);
    return fail(ctx, SVGP_NOT_POSDEF, buf);
  }
  if (nneg > 0 && m->desc.neg_var_policy == SVGP_NEGVAR_ERROR) {
    char buf[160];
    snprintf(buf, sizeof buf, "%lld predictive variances were negative (DomainError in sqrt)", (long long)nneg);
    return fail(ctx, SVGP_NEG_VARIANCE, buf);
  }
  return SVGP_OK;
}

int ensure_prepared(svgp_ctx* ctx, svgp_model* m) {
  if (m->prepared) return SVGP_OK;
  int rc = enqueue_prep(ctx, m);
  if (rc) return rc;
  PrepScalars ps;
  HIPC(ctx, hipMemcpyAsync(ps.scal, m->scal, sizeof(ps.scal), hipMemcpyDeviceToHost, ctx->stream));
  HIPC(ctx, hipMemcpyAsync(&ps.info, m->info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIPC(ctx, hipStreamSynchronize(ctx->stream));
  finish_prep(m, ps);
  return status_of(ctx, m, 0);
}

int make_data(svgp_ctx* ctx, int dtype, int layout, int d, int64_t n, const void* x_host, const void* y_host,
              svgp_data** out) {
  if (!out) return fail(ctx, SVGP_INVALID_ARG, "null out pointer");
  if (dtype != SVGP_F64 && dtype != SVGP_F32) return fail(ctx, SVGP_INVALID_ARG, "bad dtype");
  if (d > SVGP_MAX_D) return fail(ctx, SVGP_UNSUPPORTED, "input dimension beyond SVGP_MAX_D");   // as the model: the host falls back
  if (d < 1 || n < 1 || !x_host) return fail(ctx, SVGP_INVALID_ARG, "bad data shape");
  if (layout < 0 || layout > SVGP_VEC || (layout == SVGP_VEC && d != 1)) return fail(ctx, SVGP_INVALID_ARG, "bad layout");
  const size_t es = esize(dtype);
  DataGuard guard;
  guard.D = new (std::nothrow) svgp_data();
  svgp_data* D = guard.D;
  if (!D) return SVGP_OOM;
  D->dtype = dtype;
  D->d = d;
  D->n = n;
  D->ldx = n;
  hipStream_t s = ctx->stream;
  hipError_t e = hipMalloc(&D->x, size_t(n) * d * es);
  if (e == hipSuccess && y_host) e = hipMalloc(&D->y, size_t(n) * es);
  if (e != hipSuccess) return fail(ctx, SVGP_OOM, "hipMalloc failed for data");
  if (layout == SVGP_COLVECS && d > 1) {
    DevBuf tmp;
    HIPC(ctx, tmp.alloc(size_t(n) * d * es));
    HIPC(ctx, hipMemcpyAsync(tmp.p, x_host, size_t(n) * d * es, hipMemcpyHostToDevice, s));
    launch_transpose_colvecs(dtype, s, tmp.p, d, n, D->ldx, D->x);
    HIPC(ctx, hipStreamSynchronize(s));
  } else {
    HIPC(ctx, hipMemcpyAsync(D->x, x_host, size_t(n) * d * es, hipMemcpyHostToDevice, s));
  }
  if (y_host) HIPC(ctx, hipMemcpyAsync(D->y, y_host, size_t(n) * es, hipMemcpyHostToDevice, s));
  HIPC(ctx, hipStreamSynchronize(s));
  *out = guard.release();
  return SVGP_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

int32_t svgp_version(void) { return SVGP_ABI_VERSION; }

int32_t svgp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int32_t svgp_gausshermite(int32_t n, double* nodes_out, double* weights_out) {
  if (!nodes_out || !weights_out) return SVGP_INVALID_ARG;
  return gauss_hermite(n, nodes_out, weights_out);
}

// work(n, M, d) of one evaluation in host flops (2 M^2 n for trsm + trmm, M^3 / 3 for the Cholesky, plus
// the Kuf assembly and the elementwise reductions the reference materialises, ~ (3 d + 30) per element of Kuf)
double svgp_offload_work(int64_t n, int64_t M, int32_t d) {
  if (n < 0 || M < 0 || d < 0) return 0.0;
  const double nd = double(n), Md = double(M);
  return nd * Md * (2.0 * Md + 3.0 * double(d) + 30.0) + Md * Md * Md / 3.0;
}

int32_t svgp_offload_advice(int64_t n, int64_t M, int32_t d, int32_t /*dtype*/, int32_t /*want_gradient*/) {
  // One threshold for both forms: the device floor of a value-and-gradient call is ~2.5x the forward floor (600 vs 250 us
  // through the one-shot route), and so is the host's reverse-mode cost (profiles/round3/small_problems.md)
  const char* e = getenv("SVGP_OFFLOAD_MIN_WORK");   // read on every call: hosts (and the tests) change it at run time
  double min_work = 3.0e6;
  if (e && *e) {   // a value that does not parse as a non-negative number is ignored (atof gave 0 = "always offload")
    char* end = nullptr;
    const double v = strtod(e, &end);
    while (end && (*end == ' ' || *end == '\t')) ++end;
    if (end && end != e && *end == '\0' && v >= 0.0 && v == v) min_work = v;
  }
  return svgp_offload_work(n, M, d) >= min_work ? 1 : 0;
}

#ifdef SVGP_EXPERIMENTS
// only the experiments build (tools/build_experiments.sh) exports this: tools and tests tell the two libraries apart by it
int svgp_debug_experiments(void) { return 1; }
#endif

int32_t svgp_ctx_create(int32_t device_id, void* stream, svgp_ctx** out) {
  if (!out) return SVGP_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return SVGP_HIP_ERROR;  // no GPU: the library has no CPU path
  if (device_id < 0 || device_id >= n) return SVGP_INVALID_ARG;
  svgp_ctx* c = new (std::nothrow) svgp_ctx();
  if (!c) return SVGP_OOM;
  c->device = device_id;
  read_knobs(c->kn);   // the operational variables (knobs.hpp), once per context
  c->timing_on = c->kn.timing != 0;
  if (hipSetDevice(device_id) != hipSuccess) { delete c; return SVGP_HIP_ERROR; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cus = prop.multiProcessorCount;
  if (stream) {
    c->stream = static_cast<hipStream_t>(stream);
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SVGP_HIP_ERROR; }
    c->own_stream = true;
  }
  for (auto& e : c->ev)
    if (hipEventCreateWithFlags(&e, kTimingEvent) != hipSuccess) { delete c; return SVGP_HIP_ERROR; }
  for (auto& e : c->ev_chol)
    if (hipEventCreateWithFlags(&e, kTimingEvent) != hipSuccess) { delete c; return SVGP_HIP_ERROR; }
  if (hipMalloc(&c->d_res, 16 * sizeof(double)) != hipSuccess || hipMalloc(&c->d_coll, 8 * sizeof(double)) != hipSuccess || hipMalloc(&c->counter, 64) != hipSuccess || hipMalloc(&c->partial, 1024 * sizeof(double)) != hipSuccess ||
      hipMalloc(&c->negcnt, 1024 * sizeof(unsigned)) != hipSuccess) { delete c; return SVGP_OOM; }
  *out = c;
  return SVGP_OK;
}

int32_t svgp_ctx_destroy(svgp_ctx* c) {
  if (!c) return SVGP_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->work) (void)hipFree(c->work);
  if (c->partial) (void)hipFree(c->partial);
  if (c->negcnt) (void)hipFree(c->negcnt);
  if (c->mom) (void)hipFree(c->mom);
  if (c->comm) (void)svgp_ctx_detach_comm(c);
  if (c->d_res) (void)hipFree(c->d_res);
  if (c->d_coll) (void)hipFree(c->d_coll);
  if (c->h_open) (void)hipHostFree(c->h_open);
  if (c->ev_open) (void)hipEventDestroy(c->ev_open);
  if (c->counter) (void)hipFree(c->counter);
  if (c->counter2) (void)hipFree(c->counter2);
  if (c->work2) (void)hipFree(c->work2);
  for (auto& e : c->ev_row)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : c->ev_ov)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_R) (void)hipEventDestroy(c->ev_R);
  if (c->ev_S) (void)hipEventDestroy(c->ev_S);
  if (c->seg_state) (void)hipFree(c->seg_state);
  if (c->work_seg) (void)hipFree(c->work_seg);
  if (c->hstage) (void)hipHostFree(c->hstage);
  for (auto& e : c->ev_piece)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
  if (c->kuf_buf) (void)hipFree(c->kuf_buf);
  if (c->ext_g) (void)hipFree(c->ext_g);
  for (auto& st : c->pst)
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  if (c->ev_pipe_prep) (void)hipEventDestroy(c->ev_pipe_prep);
  if (c->pwork) (void)hipFree(c->pwork);
  if (c->pcounter) (void)hipFree(c->pcounter);
  if (c->pmom) (void)hipFree(c->pmom);
  if (c->gws) { c->gws->release(); delete c->gws; }
  for (auto& e : c->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : c->ev_chol)
    if (e) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return SVGP_OK;
}

const char* svgp_last_error(const svgp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// the v2 / v3 layout only (48 bytes): a host compiled against an older header sized its buffer for that
int32_t svgp_last_timing(const svgp_ctx* ctx, svgp_timing* out) {
  if (!ctx || !out) return SVGP_INVALID_ARG;
  static_assert(offsetof(svgp_timing, ms_chol) == SVGP_TIMING_V3_BYTES, "the fields appended since v3 start at byte 48");
  memcpy(out, &ctx->timing, SVGP_TIMING_V3_BYTES);
  return SVGP_OK;
}

int32_t svgp_last_timing_sized(const svgp_ctx* ctx, void* out, int64_t out_bytes) {
  if (!ctx || !out || out_bytes < 0) return SVGP_INVALID_ARG;
  const size_t n = size_t(out_bytes) < sizeof(svgp_timing) ? size_t(out_bytes) : sizeof(svgp_timing);
  memcpy(out, &ctx->timing, n);
  return SVGP_OK;
}

int32_t svgp_data_upload(svgp_ctx* ctx, int32_t dtype, int32_t layout, int32_t d, int64_t n, const void* x_host,
                         const void* y_host, svgp_data** out) {
  if (!ctx) return SVGP_INVALID_ARG;
  HIPC(ctx, hipSetDevice(ctx->device));
  return make_data(ctx, dtype, layout, d, n, x_host, y_host, out);
}

int32_t svgp_data_wrap_device(svgp_ctx* ctx, int32_t dtype, int32_t d, int64_t n, int64_t ldx, const void* x_dev,
                              const void* y_dev, svgp_data** out) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (d > SVGP_MAX_D) return fail(ctx, SVGP_UNSUPPORTED, "input dimension beyond SVGP_MAX_D");
  if (!out || !x_dev || d < 1 || n < 1 || ldx < n) return fail(ctx, SVGP_INVALID_ARG, "bad wrap arguments");
  if (dtype != SVGP_F64 && dtype != SVGP_F32) return fail(ctx, SVGP_INVALID_ARG, "bad dtype");
  svgp_data* D = new (std::nothrow) svgp_data();
  if (!D) return SVGP_OOM;
  D->dtype = dtype;
  D->d = d;
  D->n = n;
  D->ldx = ldx;
  D->x = const_cast<void*>(x_dev);
  D->y = const_cast<void*>(y_dev);
  D->own = false;
  *out = D;
  return SVGP_OK;
}

int32_t svgp_data_free(svgp_ctx* ctx, svgp_data* D) {
  if (!D) return SVGP_OK;
  if (ctx) (void)hipSetDevice(ctx->device);
  if (D->own) {
    if (D->x) (void)hipFree(D->x);
    if (D->y) (void)hipFree(D->y);
  }
  delete D;
  return SVGP_OK;
}

int32_t svgp_model_free(svgp_ctx* ctx, svgp_model* m) {
  if (!m) return SVGP_OK;
  if (ctx) (void)hipSetDevice(ctx->device);
  void* bufs[] = {m->z_raw, m->m_raw, m->Lq_raw, m->invl, m->zs, m->L, m->T, m->U, m->mp, m->B, m->scal, m->info, m->gh_x, m->gh_w};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  delete m;
  return SVGP_OK;
}

int32_t svgp_model_create(svgp_ctx* ctx, const svgp_model_desc* desc, svgp_model** out) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (!out) return fail(ctx, SVGP_INVALID_ARG, "null out pointer");
  *out = nullptr;
  int rc = validate_desc(ctx, desc);
  if (rc) return rc;
  HIPC(ctx, hipSetDevice(ctx->device));
  svgp_model* m = new (std::nothrow) svgp_model();
  if (!m) return SVGP_OOM;
  m->M = desc->M;
  m->Mp = (desc->M + 127) / 128 * 128;
  m->dtype = desc->dtype;
  m->d = desc->d;
  m->es = esize(desc->dtype);
  const size_t es = m->es, M = size_t(m->M), Mp = size_t(m->Mp);
  struct { void** p; size_t bytes; } allocs[] = {
      {&m->z_raw, M * m->d * es}, {&m->m_raw, M * es},     {&m->Lq_raw, M * M * es}, {&m->invl, size_t(m->d) * es},
      {&m->zs, Mp * m->d * es},   {&m->L, Mp * Mp * es},   {&m->T, Mp * Mp * es},    {&m->U, Mp * Mp * es},
      {&m->mp, Mp * es},          {(void**)&m->scal, (8 + Mp) * sizeof(double)},     {(void**)&m->info, info_bytes(Mp)},
  };
  for (auto& a : allocs)
    if (hipMalloc(a.p, a.bytes) != hipSuccess) {
      svgp_model_free(ctx, m);
      return fail(ctx, SVGP_OOM, "hipMalloc failed for model buffers");
    }
  if (desc->parametrization == SVGP_CENTERED && hipMalloc(&m->B, Mp * Mp * es) != hipSuccess) {
    svgp_model_free(ctx, m);
    return fail(ctx, SVGP_OOM, "hipMalloc failed for model buffers");
  }
  if (hipMemsetAsync(m->T, 0, Mp * Mp * es, ctx->stream) != hipSuccess) {   // launch_potrf's contract: T zero above the diagonal
    svgp_model_free(ctx, m);
    return fail(ctx, SVGP_HIP_ERROR, "hipMemsetAsync failed for the model's T buffer");
  }
  rc = upload_params(ctx, m, desc);
  if (rc) {
    svgp_model_free(ctx, m);
    return rc;
  }
  *out = m;
  return SVGP_OK;
}

int32_t svgp_model_update(svgp_ctx* ctx, svgp_model* m, const svgp_model_desc* desc) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (!m) return fail(ctx, SVGP_INVALID_ARG, "null model");
  int rc = validate_desc(ctx, desc);
  if (rc) return rc;
  if (desc->M != m->M || desc->d != m->d || desc->dtype != m->dtype || desc->parametrization != m->desc.parametrization)
    return fail(ctx, SVGP_INVALID_ARG, "svgp_model_update: M, d, dtype and parametrization must not change");
  HIPC(ctx, hipSetDevice(ctx->device));
  return upload_params(ctx, m, desc);
}

int32_t svgp_elbo_partial(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len,
                          double partial_out[4]) {
  int rc = check_batch(ctx, m, data, off, len, true);
  if (rc) return rc;
  if (!partial_out) return fail(ctx, SVGP_INVALID_ARG, "null output");
  ElboRead r;
  rc = run_elbo(ctx, m, data, off, len, false, &r);   // always local: no collective
  if (rc) return rc;
  partial_out[0] = r.E;
  partial_out[1] = double(len);
  partial_out[2] = r.n_neg;
  partial_out[3] = double(m->chol_info);
  return status_of(ctx, m, r.n_neg);
}

namespace {
void fill_terms(svgp_terms* t, const svgp_model* m, double elbo, const ElboRead& r, double scale) {
  if (!t) return;
  t->elbo = elbo;
  t->expectation = r.E;
  t->kl = m->kl;
  t->scale = scale;
  t->logdet_kuu = m->logdet_kuu;
  t->n_points = int64_t(r.n_points);
  t->n_neg_var = int64_t(r.n_neg);
  t->chol_info = m->chol_info;
  t->reserved = 0;
}
}  // namespace

// With a communicator attached (svgp_ctx_attach_comm / svgp_group_create) this call is COLLECTIVE: every rank passes
// its own shard's batch, the ranks' {sum E, n, n_neg, flags} are summed by one ncclAllReduce on the device, and every
// rank returns the same global ELBO = sum E * num_data / n_global - KL (SVA:355-359; the KL is replicated, not summed).
int32_t svgp_elbo(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len, double num_data,
                  double* elbo_out, svgp_terms* terms_out) {
  int rc = check_batch(ctx, m, data, off, len, true);
  if (rc) return (ctx && ctx->comm) ? elbo_collective(ctx, rc) : rc;   // keep the peers' collective matched
  ElboRead r;
  rc = run_elbo(ctx, m, data, off, len, true, &r);
  if (rc) return rc;
  const double scale = (num_data > 0 ? num_data : r.n_points) / r.n_points;   // SVA:357-358
  const double elbo = r.E * scale - m->kl;                                     // SVA:359
  fill_terms(terms_out, m, elbo, r, scale);
  rc = status_of(ctx, m, r.n_neg, r.bad_chol, r.failed);
  if (elbo_out) *elbo_out = (rc == SVGP_OK) ? elbo : NAN;
  return rc;
}

int32_t svgp_prior_kl(svgp_ctx* ctx, svgp_model* m, double* kl_out, double* logdet_out) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (!m) return fail(ctx, SVGP_INVALID_ARG, "null model");
  HIPC(ctx, hipSetDevice(ctx->device));
  int rc = ensure_prepared(ctx, m);
  if (rc) return rc;
  if (kl_out) *kl_out = m->kl;
  if (logdet_out) *logdet_out = m->logdet_kuu;
  return SVGP_OK;
}

int32_t svgp_elbo_host(svgp_ctx* ctx, const svgp_model_desc* desc, int32_t layout_x, int64_t n, const void* x_host,
                       const void* y_host, double num_data, double* elbo_out, svgp_terms* terms_out) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (!desc || !y_host) {
    const int rc0 = fail(ctx, SVGP_INVALID_ARG, "null descriptor or observations");
    return ctx->comm ? elbo_collective(ctx, rc0) : rc0;
  }
  svgp_model* m = nullptr;
  svgp_data* D = nullptr;
  int rc = svgp_model_create(ctx, desc, &m);
  if (rc == SVGP_OK) rc = svgp_data_upload(ctx, desc->dtype, layout_x, desc->d, n, x_host, y_host, &D);
  if (rc == SVGP_OK) rc = svgp_elbo(ctx, m, D, 0, n, num_data, elbo_out, terms_out);
  else if (ctx->comm) rc = elbo_collective(ctx, rc);   // this rank never reached svgp_elbo: join its peers' all-reduce with the failure flag
  svgp_data_free(ctx, D);
  svgp_model_free(ctx, m);
  return rc;
}

int32_t svgp_kuf(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len, void* Kuf_out_host) {
  int rc = check_batch(ctx, m, data, off, len, false);
  if (rc) return rc;
  HIPC(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t bytes = size_t(m->M) * size_t(len) * m->es;
  if (bytes > ctx->kuf_bytes) {
    if (ctx->kuf_buf) (void)hipFree(ctx->kuf_buf);
    ctx->kuf_buf = nullptr;
    ctx->kuf_bytes = 0;
    HIPC(ctx, hipMalloc(&ctx->kuf_buf, bytes));
    ctx->kuf_bytes = bytes;
  }
  launch_scale_inputs(m->dtype, s, m->z_raw, m->desc.layout_z, m->d, m->M, m->Mp, m->invl, m->zs);
  TREC(ctx, ctx->ev[0], s);
  launch_kuf(m->dtype, s, kparams(m), m->zs, m->M, m->Mp, data->x, data->ldx, off, len, ctx->kuf_buf);
  TREC(ctx, ctx->ev[1], s);
  HIPC(ctx, hipGetLastError());
  if (Kuf_out_host) HIPC(ctx, hipMemcpyAsync(Kuf_out_host, ctx->kuf_buf, bytes, hipMemcpyDeviceToHost, s));
  HIPC(ctx, hipStreamSynchronize(s));
  float t = 0;
  (void)elapsed_ms(ctx, &t, ctx->ev[0], ctx->ev[1]);
  ctx->timing = svgp_timing{};
  ctx->timing.ms_kuf = t;
  ctx->timing.ms_total = t;
  return SVGP_OK;
}

int32_t svgp_posterior(svgp_ctx* ctx, svgp_model* m, void* Lk_out, void* alpha_out, void* B_out) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (!m) return fail(ctx, SVGP_INVALID_ARG, "null model");
  HIPC(ctx, hipSetDevice(ctx->device));
  int rc = ensure_prepared(ctx, m);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const size_t es = m->es, M = size_t(m->M), Mp = size_t(m->Mp);
  DevBuf tmpbuf;
  HIPC(ctx, tmpbuf.alloc((M * M + Mp) * es));
  void* tmp = tmpbuf.p;
  void* vec = static_cast<char*>(tmp) + M * M * es;
  if (Lk_out) {
    launch_extract_lower(m->dtype, s, m->L, m->Mp, m->M, tmp);
    HIPC(ctx, hipMemcpyAsync(Lk_out, tmp, M * M * es, hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipStreamSynchronize(s));
  }
  if (alpha_out) {
    // α = Lk' \ m~ (SVA:182; for Centered m~ = Lk \ (m − μ) so α = Kuu \ (m − μ), SVA:134)
    HIPC(ctx, hipMemcpyAsync(vec, m->mp, Mp * es, hipMemcpyDeviceToDevice, s));
    launch_trsv2(m->dtype, s, m->L, m->T, m->Mp, 1, vec);
    HIPC(ctx, hipMemcpyAsync(alpha_out, vec, M * es, hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipStreamSynchronize(s));
  }
  if (B_out) {
    if (m->desc.parametrization == SVGP_CENTERED) {
      launch_extract_lower(m->dtype, s, m->B, m->Mp, m->M, tmp);
    } else {
      launch_extract_lower(m->dtype, s, m->Lq_raw, m->M, m->M, tmp);
    }
    HIPC(ctx, hipMemcpyAsync(B_out, tmp, M * M * es, hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipStreamSynchronize(s));
  }
  HIPC(ctx, hipGetLastError());
  return SVGP_OK;
}

static int32_t predict_impl(svgp_ctx* ctx, svgp_model* m, int32_t layout, int64_t nx, const void* x_host, int64_t ny,
                            const void* y_host, void* mean_out, void* var_out, void* cov_out, bool cross) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (!m || !x_host || nx < 1) return fail(ctx, SVGP_INVALID_ARG, "bad predict arguments");
  HIPC(ctx, hipSetDevice(ctx->device));
  int rc = ensure_prepared(ctx, m);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const size_t es = m->es;
  svgp_data* X = nullptr;
  svgp_data* Y = nullptr;
  rc = make_data(ctx, m->dtype, layout, m->d, nx, x_host, nullptr, &X);
  if (rc) return rc;
  if (cross) {
    rc = make_data(ctx, m->dtype, layout, m->d, ny, y_host, nullptr, &Y);
    if (rc) { svgp_data_free(ctx, X); return rc; }
  }
  auto pad = [](int64_t n) { return (n + 127) / 128 * 128; };
  const int64_t ldax = pad(nx), lday = cross ? pad(ny) : 0;
  void *mu = nullptr, *var = nullptr, *Ax = nullptr, *Cx = nullptr, *Ay = nullptr, *Cy = nullptr, *cov = nullptr;
  const bool want_cov = cov_out != nullptr;
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = hipMalloc(&mu, size_t(nx) * es);
  if (e == hipSuccess) e = hipMalloc(&var, size_t(nx) * es);
  if (want_cov) {
    if (e == hipSuccess) e = hipMalloc(&Ax, size_t(m->Mp) * ldax * es);
    if (e == hipSuccess) e = hipMalloc(&Cx, size_t(m->Mp) * ldax * es);
    if (cross) {
      if (e == hipSuccess) e = hipMalloc(&Ay, size_t(m->Mp) * lday * es);
      if (e == hipSuccess) e = hipMalloc(&Cy, size_t(m->Mp) * lday * es);
    }
    if (e == hipSuccess) e = hipMalloc(&cov, size_t(nx) * size_t(cross ? ny : nx) * es);
  }
  auto cleanup = [&]() {
    for (void* p : {mu, var, Ax, Cx, Ay, Cy, cov})
      if (p) (void)hipFree(p);
    svgp_data_free(ctx, X);
    svgp_data_free(ctx, Y);
  };
  if (e != hipSuccess) { cleanup(); return fail(ctx, SVGP_OOM, "hipMalloc failed in predict"); }
  StripOuts o;
  o.mu = mu; o.var = var; o.A = Ax; o.C = Cx; o.lda = ldax;
  rc = enqueue_strips(ctx, m, X->x, X->ldx, nullptr, 0, nx, o);
  if (rc == SVGP_OK && cross && want_cov) {
    StripOuts oy;
    oy.A = Ay; oy.C = Cy; oy.lda = lday;
    rc = enqueue_strips(ctx, m, Y->x, Y->ldx, nullptr, 0, ny, oy);
  }
  if (rc == SVGP_OK && want_cov) {
    if (cross)
      launch_cov_assemble(m->dtype, s, kparams(m), X->x, X->ldx, nx, Y->x, Y->ldx, ny, Ax, Cx, ldax, Ay, Cy, lday, m->Mp, cov);
    else
      launch_cov_assemble(m->dtype, s, kparams(m), X->x, X->ldx, nx, X->x, X->ldx, nx, Ax, Cx, ldax, Ax, Cx, ldax, m->Mp, cov);
  }
  hipError_t le = hipGetLastError();
  if (rc == SVGP_OK && le != hipSuccess) rc = fail(ctx, SVGP_HIP_ERROR, hipGetErrorString(le));
  if (rc == SVGP_OK) {
    if (mean_out) (void)hipMemcpyAsync(mean_out, mu, size_t(nx) * es, hipMemcpyDeviceToHost, s);
    if (var_out) (void)hipMemcpyAsync(var_out, var, size_t(nx) * es, hipMemcpyDeviceToHost, s);
    if (want_cov) (void)hipMemcpyAsync(cov_out, cov, size_t(nx) * size_t(cross ? ny : nx) * es, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) rc = fail(ctx, SVGP_HIP_ERROR, "stream sync failed in predict");
  } else {
    (void)hipStreamSynchronize(s);
  }
  cleanup();
  return rc;
}

int32_t svgp_predict(svgp_ctx* ctx, svgp_model* m, int32_t layout_x, int64_t n, const void* x_host, void* mean_out,
                     void* var_out, void* cov_out) {
  return predict_impl(ctx, m, layout_x, n, x_host, 0, nullptr, mean_out, var_out, cov_out, false);
}

int32_t svgp_predict_cross_cov(svgp_ctx* ctx, svgp_model* m, int32_t layout, int64_t nx, const void* x_host, int64_t ny,
                               const void* y_host, void* cov_out) {
  if (ctx && (!y_host || ny < 1 || !cov_out)) return fail(ctx, SVGP_INVALID_ARG, "bad cross-cov arguments");
  return predict_impl(ctx, m, layout, nx, x_host, ny, y_host, nullptr, nullptr, cov_out, true);
}

}  // extern "C"

namespace {

// split-K slice count of the SYRK for a chunk of nc points (the rule: grad_workspace)
int syrk_slices(const svgp_ctx* ctx, const svgp_model* m, int64_t nc) {
  static const int ns_forced = exp_int("SVGP_GEMM_PM_SLICES", 0);   // tuning knob (experiments build)
  if (ns_forced > 0) return ns_forced;
  const int64_t Mp = m->Mp;
  const int nP = int(Mp / 128), ntiles = nP * (nP + 1) / 2;
  const int items = ntiles * (m->dtype == SVGP_F64 ? 2 : 1), slots = 2 * ctx->num_cus;
  int ns = 1;
  double best = 0.0;
  for (int c = 1; c <= 64; ++c) {
    if (c > 1 && (size_t(c) * size_t(Mp) * size_t(Mp) * m->es > (size_t(2) << 30) || nc / c < 512)) break;
    const double wg = double(items) * c, eff = wg / (std::ceil(wg / slots) * slots);
    if (eff > best + 1e-9) { best = eff; ns = c; }
    if (eff >= 0.97) break;
  }
  return ns;
}

// points per chunk of a value-and-gradient evaluation over `len` points (a multiple of 128): the At / Pt buffers hold one chunk
int64_t grad_chunk_points(int64_t Mp, size_t es, int64_t len) {
  static const int64_t cap_cols = exp_ll("SVGP_GRAD_CHUNK", 65536ll);   // tuning knobs (experiments build)
  static const double cap_bytes = exp_double("SVGP_GRAD_CHUNK_BYTES", 1.0e9);
  int64_t cap = int64_t(cap_bytes / double(Mp * int64_t(es))) / 128 * 128;
  cap = cap < 128 ? 128 : (cap > cap_cols ? cap_cols : cap);
  int64_t nc = (len + 127) / 128 * 128;
  // a batch of at most two chunks' worth goes as ONE chunk (C2, 1e5 points: 3.7 -> 3.5 ms - one SYRK / kgrad launch, no drain between
  // the strips of the two chunks); longer batches keep the 65 536-point chunks (H, H32 flat from 32 768 to 262 144; C5 best at 65 536)
  if (nc > cap && nc <= 2 * cap && double(nc) * double(Mp * int64_t(es)) <= 2.0 * cap_bytes) cap = nc;
  return nc < cap ? nc : cap;
}

int grad_workspace(svgp_ctx* ctx, svgp_model* m, int64_t len, GradWs** out) {
  const size_t es = m->es;
  const int64_t Mp = m->Mp;
  const int64_t nc = grad_chunk_points(Mp, es, len);
  GradWs* w = ctx->gws;
  if (w && w->dtype == m->dtype && w->Mp == Mp && w->d == m->d && w->nc >= nc) { *out = w; return SVGP_OK; }
  if (w) { w->release(); delete w; ctx->gws = nullptr; }
  w = new (std::nothrow) GradWs();
  if (!w) return SVGP_OOM;
  w->dtype = m->dtype; w->Mp = Mp; w->d = m->d; w->nc = nc;
  // Split-K slices of the SYRK: work items = tiles (f64: two 128 x 64 halves each) x slices over 2 workgroup slots per CU.  The
  // smallest count whose last round of slots is >= 97 % full, else the fullest (H: 72 items, 7 slices = 504 of 512; C3: 136 items,
  // 11 slices = 1496 of 1536 - the round-2 rule floor(512 / tiles) left C3 with 408 of 512: 175 -> 164 ms), within 2 GiB of slice
  // buffer and >= 512 points per slice.  Placing the items of a slice on one XCD (they read the same rows of A) was measured
  // and rejected: H 81.9 -> 85-87 ms - the operand re-reads come out of the Infinity Cache at no cost to the MFMA pipe.
  w->nslices = syrk_slices(ctx, m, int64_t(1) << 40);   // the buffer holds the count an unbounded chunk would take: a call's count never exceeds it
  w->rb = grad_rowblocks(m->dtype, m->d, Mp);
  // workgroups per CU of the kernel-gradient reductions (grad.hip: kgrad_mfma_kernel): as many as its registers and LDS admit - the kernel
  // is a chain global load -> MFMA -> kernel function -> MFMA per 16-point tile and lives off the waves it can interleave (rocprofv3, us
  // per 65 536-point chunk at M = 1024, 2 / 3 / 4 / 6 per CU: d = 8 f64 185 / 171 / 162 / 177, fp32 115 / 99 / 88 / 104; d = 64 f64, whose
  // 72 KiB of LDS admit two, 512 / 671 / 549 / 598; profiles/round6/kgrad_wg_per_cu.log)
  const int kg_dflt = grad_dreg(m->d) <= 16 ? 4 : (grad_dreg(m->d) <= 32 ? 3 : 2);
  static const int kg_knob = exp_int("SVGP_KGRAD_WG_PER_CU", 0);   // tuning knob (experiments build)
  const int kg_wg = kg_knob > 0 ? kg_knob : kg_dflt;
  int nu = (kg_wg * ctx->num_cus + w->rb - 1) / w->rb;
  w->ns_uf = nu < 1 ? 1 : (nu > 256 ? 256 : nu);
  w->ns_uu = 8;
  const int dreg = grad_dreg(m->d);
  const size_t mn = size_t(Mp) * size_t(nc) * es, mm = size_t(Mp) * size_t(Mp) * es;
  w->g_b = size_t(w->nslices) * mm;
  w->rp_uf_b = size_t(w->ns_uf) * (2 + dreg) * Mp * 8; w->sp_uf_b = size_t(w->ns_uf) * w->rb * (1 + dreg) * 8;
  w->rp_uu_b = size_t(w->ns_uu) * (2 + dreg) * Mp * 8; w->sp_uu_b = size_t(w->ns_uu) * w->rb * (1 + dreg) * 8;
  w->part5_strips = nc / 32 + 2;   // narrowest strips: 32 points
  // [rp_uf | sp_uf | rp_uu | sp_uu | sums (8) | scal_out (1 + dreg) | prep scalars (4) + info (1)]: one memset, and the tail
  // [sums .. info] is the ONE fp64 read-back of an evaluation
  w->zero_b = w->rp_uf_b + w->sp_uf_b + w->rp_uu_b + w->sp_uu_b + size_t(8 + 1 + dreg + 5) * 8;
  struct { void** p; size_t b; } req[] = {
      {&w->At, mn}, {&w->Pt, mn}, {&w->gmu, 2 * size_t(nc) * es + 256},   // g_mu | g_v contiguous: one memset per chunk (+ the SYRK's weight DMA reads 256 B at a time)
      {&w->Lqp, mm}, {&w->G1, w->g_b}, {&w->G2, mm}, {&w->LkRM, mm},
      {&w->LbarRM, mm}, {&w->Phi, mm}, {&w->tmp, mm}, {&w->H, mm}, {&w->LinvRM, mm}, {&w->LinvCM, mm},
      // the user-layout blocks hold M d + M + M^2 elements; sized by Mp because the workspace is reused for every model of the
      // same (dtype, Mp, d), whatever its M (M = 45 then M = 96 on one context overran the smaller buffers)
      {&w->gblk, (size_t(Mp) * m->d + size_t(Mp)) * es + mm}, {&w->cblk, (size_t(Mp) * m->d + size_t(Mp)) * es + mm / 2 + size_t(Mp) * es},
      {&w->BbarRM, mm}, {&w->rbar, size_t(Mp) * es},
      {&w->W2, mm}, {&w->Rcm, mm}, {&w->G1p, mm}, {&w->alpha, size_t(Mp) * es}, {(void**)&w->avec, size_t(Mp) * 8},
      {(void**)&w->gemv_part, size_t(Mp / 128) * size_t(Mp) * 8},
      {&w->zero_blk, w->zero_b},
      {(void**)&w->partial5, size_t(w->part5_strips) * 5 * 8}, {(void**)&w->invl_d, size_t(m->d) * 8},
      {(void**)&w->kred, (size_t(2 + dreg) * size_t(Mp) + size_t(1 + dreg)) * 8}};
  for (auto& r : req) {
    if (hipMalloc(r.p, r.b) != hipSuccess) {
      w->release();
      delete w;
      return fail(ctx, SVGP_OOM, "hipMalloc failed for the gradient workspace");
    }
    w->all.push_back(*r.p);
  }
  w->gv = static_cast<char*>(w->gmu) + size_t(nc) * es;
  {
    char* z = static_cast<char*>(w->zero_blk);
    w->rp_uf = reinterpret_cast<double*>(z); z += w->rp_uf_b;
    w->sp_uf = reinterpret_cast<double*>(z); z += w->sp_uf_b;
    w->rp_uu = reinterpret_cast<double*>(z); z += w->rp_uu_b;
    w->sp_uu = reinterpret_cast<double*>(z); z += w->sp_uu_b;
    w->sums = reinterpret_cast<double*>(z);
  }
  w->scal_out = w->sums + 8;   // contiguous with sums: the data-parallel path all-reduces [sums | scal_out] in one piece
  // Linv is written block-lower only and the products skip the tiles above the diagonal, but the stale upper part of a reused
  // buffer must at least be finite (0 x NaN): start from zeros once
  for (void* p : {w->LinvRM, w->LinvCM})
    if (hipMemsetAsync(p, 0, mm, ctx->stream) != hipSuccess) {
      w->release();
      delete w;
      return fail(ctx, SVGP_HIP_ERROR, "memset failed");
    }
  // the point-major chunk buffers are read beyond the written points of a short last chunk (against g_v = 0): start from zeros
  for (void* p : {w->At, w->Pt})
    if (hipMemsetAsync(p, 0, mn, ctx->stream) != hipSuccess) {
      w->release();
      delete w;
      return fail(ctx, SVGP_HIP_ERROR, "memset failed");
    }
  ctx->gws = w;
  *out = w;
  return SVGP_OK;
}

// ---- chunk pipeline of a value-and-gradient evaluation (round 5) -----------------------------------------------
// A batch of more than one chunk used to run strictly serially on one stream: strips(k) [MFMA, ragged end: the last 13 % of a launch
// run one workgroup per CU] -> point gradients -> kernel-gradient reductions(k) [f64 VALU, no MFMA] -> SYRK(k) [MFMA] -> strips(k + 1)
// [starts in its MFMA-free pre-generation] ...  Now the per-chunk arrays (A, P point-major, g_mu | g_v, the point-gradient partials)
// exist in `lanes` sets; the strips + point gradients of chunk k run on a pipeline stream (chunk k on stream k mod `streams`), the
// consumers of chunk k (reductions, sum5, SYRK) on the main stream behind ev_strips[lane], and the strips of chunk k + lanes wait for
// ev_done[lane].  Every accumulation (slice buffer, row partials, sums) stays on the main stream in chunk order, so the result is
// bitwise the serial one.
struct PipeCfg { int lanes = 1, streams = 1, prio = 0; };
PipeCfg pipe_cfg(const svgp_ctx* ctx) { return PipeCfg{ctx->kn.pipe_lanes, ctx->kn.pipe_streams, ctx->kn.pipe_prio}; }

int ensure_pipe(svgp_ctx* ctx, GradWs* w, const svgp_model* m, const PipeCfg& pc, size_t work_bytes, size_t nc) {
  const size_t es = m->es;
  for (int q = 0; q < pc.streams; ++q) {
    if (ctx->pst[q]) continue;
    int least = 0, greatest = 0;
    if (pc.prio != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
      HIPC(ctx, hipStreamCreateWithPriority(&ctx->pst[q], hipStreamNonBlocking, pc.prio < 0 ? least : greatest));
    else
      HIPC(ctx, hipStreamCreateWithFlags(&ctx->pst[q], hipStreamNonBlocking));
  }
  if (!ctx->ev_pipe_prep) HIPC(ctx, hipEventCreateWithFlags(&ctx->ev_pipe_prep, kSyncEvent));
  if (pc.streams > 1) {   // the second pipeline stream's strips: their own scratch strips, queue head and moments
    if (work_bytes > ctx->pwork_bytes) {
      if (ctx->pwork) (void)hipFree(ctx->pwork);
      ctx->pwork = nullptr; ctx->pwork_bytes = 0;
      HIPC(ctx, hipMalloc(&ctx->pwork, work_bytes));
      ctx->pwork_bytes = work_bytes;
    }
    if (nc > ctx->pmom_cap) {
      if (ctx->pmom) (void)hipFree(ctx->pmom);
      ctx->pmom = nullptr; ctx->pmom_cap = 0;
      HIPC(ctx, hipMalloc(&ctx->pmom, 2 * nc * sizeof(double)));
      ctx->pmom_cap = nc;
    }
    if (!ctx->pcounter) HIPC(ctx, hipMalloc(&ctx->pcounter, 64));
  }
  if (w->lanes.empty()) {
    GradWs::Lane l0;
    l0.At = w->At; l0.Pt = w->Pt; l0.gmu = w->gmu; l0.gv = w->gv; l0.partial5 = w->partial5;
    w->lanes.push_back(l0);
  }
  const size_t mn = size_t(w->Mp) * size_t(w->nc) * es;
  while (int(w->lanes.size()) < pc.lanes) {
    GradWs::Lane l;
    struct { void** p; size_t b; bool zero; } req[] = {
        {&l.At, mn, true}, {&l.Pt, mn, true}, {&l.gmu, 2 * size_t(w->nc) * es + 256, false}, {(void**)&l.partial5, size_t(w->part5_strips) * 5 * 8, false}};
    for (auto& r : req) {
      if (hipMalloc(r.p, r.b) != hipSuccess) return fail(ctx, SVGP_OOM, "hipMalloc failed for the gradient pipeline");
      w->all.push_back(*r.p);
      if (r.zero) HIPC(ctx, hipMemsetAsync(*r.p, 0, r.b, ctx->stream));   // read beyond the written points of a short last chunk (against g_v = 0)
    }
    l.gv = static_cast<char*>(l.gmu) + size_t(w->nc) * es;
    w->lanes.push_back(l);
  }
  for (GradWs::Lane& l : w->lanes) {
    if (!l.ev_strips) HIPC(ctx, hipEventCreateWithFlags(&l.ev_strips, kSyncEvent));
    if (!l.ev_done) HIPC(ctx, hipEventCreateWithFlags(&l.ev_done, kSyncEvent));
  }
  return SVGP_OK;
}

}  // namespace

namespace {
// value = scale * sum_i E_i - klw * KL and its gradient over points [off, off + len): three stages like the forward
// evaluation.  Data-parallel (a communicator on the context, `collective`): every rank uses scale = num_data / n_global
// with n_global all-reduced on the device BEFORE the backward pass (the strips' phase 3 reads it there; no host hop) and
// klw = 1 / world, so the plain sum over ranks of (value, gradient) is the global ELBO and its gradient; that sum is ONE
// grouped ncclAllReduce of {z_bar, m_bar, Lq_bar, [sums | scal_out]} at the end.
struct GradCall {
  GradWs* w = nullptr;
  const double* n_global_dev = nullptr;
  double scale = 1.0, klw = 1.0, num_data = 0.0;
  bool collective = false, centered = false;
  bool packed = false;   // w->cblk already holds {z_bar | m_bar | packed tril(Lq_bar)} (the collective's all-reduced block)
  int64_t len = 0;
  // host-evaluated likelihood (svgp_elbo_grad_ext): per-point dE/dmu, dE/dv (host, fp64, [len] each) and the host's sum E
  const double *ext_gmu = nullptr, *ext_gv = nullptr;
  double ext_sum_e = 0.0;
};

// out = Xt' Yt for M x M operands ("k-major": Xt[k][r] at Xt[k Mp + r]; a row-major matrix Z is the operand Z, a column-major
// one is Z').  Lower 128-tiles of the row-major result, or all tiles with kMmFull; kMm?Low / kMm?Up name triangular operands
// (half the k-steps).  One workgroup pair per tile is 72 (128) workgroups at M = 1024 for 256 CUs, so the product is split
// along K into slices whose partials land in `scratch` ([ns][Mp][Mp], the SYRK's slice buffer when it is idle) and are summed
// in a fixed order.  The result OVERWRITES `out` (no pre-zeroing: round 2 spent a memset per product on it).
void gemm_mm(svgp_ctx* ctx, GradWs* w, int dt, hipStream_t s, const void* Xt, const void* Yt, int64_t Mp, void* out, int flags = 0,
             void* scratch = nullptr) {   // scratch: [min(nP, nslices)][Mp][Mp] instead of the SYRK's slice buffer
  const int nP = int(Mp / 128), ntiles = (flags & kMmFull) ? nP * nP : nP * (nP + 1) / 2;
  int ns = (2 * ctx->num_cus) / (ntiles * (dt == SVGP_F64 ? 2 : 1));   // fill the workgroup slots once (f64: two 128 x 64 halves per tile)
  if (ns > nP) ns = nP;                         // at least 8 k-steps of 16 per slice
  if (ns > w->nslices) ns = w->nslices;         // the scratch is the SYRK's [nslices][Mp][Mp]
  static const int knob = exp_int("SVGP_GEMM_MM_SPLITK", 1);   // A/B knob (experiments build)
  if (ns < 2 || !knob) {
    launch_gemm_pm(dt, s, Xt, Yt, nullptr, 1.0, Mp, Mp, Mp, 1, out, 1, flags);
    return;
  }
  const int64_t sl = ((Mp + ns - 1) / ns + 15) / 16 * 16;
  void* sc = scratch ? scratch : w->G1;
  launch_gemm_pm(dt, s, Xt, Yt, nullptr, 1.0, Mp, Mp, sl, ns, sc, 1, flags);
  launch_sum_slices_lower(dt, s, sc, ns, Mp, out, (flags & kMmFull) ? 1 : 0, 1);
}

int grad_enqueue_impl(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len, GradCall& gc);
int grad_enqueue(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len, GradCall& gc) {
  const int rc = grad_enqueue_impl(ctx, m, data, off, len, gc);
  // a failure between the fork and the join of the segmented strips leaves work on the second stream that the main stream never
  // waited for: drain it, so that the next call on this context cannot meet it in the shared scratch
  if (rc != SVGP_OK && ctx->overlapped && ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
  if (rc != SVGP_OK && ctx->pipelined)   // likewise the chunk pipeline's streams
    for (hipStream_t st : ctx->pst)
      if (st) (void)hipStreamSynchronize(st);
  return rc;
}

int grad_enqueue_impl(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len, GradCall& gc) {
  const bool centered = gc.centered = (m->desc.parametrization == SVGP_CENTERED);
  HIPC(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  GradWs* w = gc.w;
  if (!w) {
    int rcw = grad_workspace(ctx, m, len, &w);
    if (rcw) return rcw;
    gc.w = w;
  }
  int rc = SVGP_OK;
  gc.len = len;
  const double scale = gc.scale, klw = gc.klw;
  const double* n_global_dev = gc.n_global_dev;   // data-parallel: the all-reduced batch size, on the device (grad_handshake)
  const int dt = m->dtype;
  const int64_t Mp = m->Mp, M = m->M;
  // this call's chunk (a reused workspace may hold more: the chunking - and with it the summation order - depends on the call only)
  const int64_t nc = std::min<int64_t>(w->nc, grad_chunk_points(Mp, m->es, len));
  const int ns_syrk = std::min(w->nslices, syrk_slices(ctx, m, nc));   // the SYRK's slices: a function of the call, like the chunk
  const size_t es = m->es;
  const int dreg = grad_dreg(m->d);
  // the user-layout gradient blocks of THIS model, contiguous: {z_bar | m_bar | Lq_bar}
  w->zbar = w->gblk;
  w->mbar = static_cast<char*>(w->gblk) + size_t(M) * m->d * es;
  w->Lqbar = static_cast<char*>(w->mbar) + size_t(M) * es;
  // Strips beside the factorisation (round 4): a one-chunk, one-round batch runs its phase 1 as segmented strips on the second
  // stream - panel I behind the event of block row I of T - and its phase 3 behind the M-sized gradient prep (Linv, alpha, R) that
  // the main stream computes meanwhile.  Same conditions as the forward path (overlap_plan), plus: the whole batch is one chunk.
  OverlapPlan gop;
  if (len <= nc && centered == false) {
    gop = overlap_plan(ctx, m, len, StripOuts{});
    if (gop.on) {   // the gradient's strips: the single-launch plan's geometry (no concurrent tail there)
      const StripPlan sp = strip_plan_single(dt, Mp, len, ctx->num_cus);
      gop.nt = sp.grid ? sp.nt : sp.nt_tail;
      gop.grid = sp.grid ? sp.grid : sp.grid_tail;
      gop.nstrips = sp.grid ? sp.nstrips : sp.nstrips_tail;
      if (gop.head_points < len || gop.nstrips > gop.grid || (sp.grid && sp.nt_tail)) gop.on = false;   // one chunk, one round only
    }
    if (gop.on) {
      rc = ensure_overlap(ctx, size_t(gop.nstrips) * strip_seg_state_doubles(dt, gop.nt));
      // the size the chunk loop below asks for (per workgroup >= per strip here): nothing may reallocate once segments are enqueued
      if (rc == SVGP_OK) rc = ensure_scratch(ctx, 2 * strip_work_bytes(dt, Mp, gop.nt, gop.grid > int(gop.nstrips) ? gop.grid : int(gop.nstrips)), size_t(nc));
      if (rc) return rc;
    }
  }
  LikParams lp{};
  lp.lik = m->desc.likelihood;
  lp.gh_n = m->gh_n;
  lp.sigma2 = lik_param(m->desc);
  lp.digamma_alpha = m->desc.likelihood == SVGP_LIK_GAMMA_EXP ? digamma_d(m->desc.lik_sigma2) : 0.0;
  lp.gh_x = m->gh_x;
  lp.gh_w = m->gh_w;
  lp.clamp_neg_var = (m->desc.neg_var_policy == SVGP_NEGVAR_CLAMP);
  lp.mean_const = m->desc.mean_const;
  const KernelParams kp = kparams(m);
  if (gc.ext_gmu) {   // the host's point gradients: one H2D copy of 2 x len doubles
    if (size_t(len) > ctx->ext_cap) {
      if (ctx->ext_g) (void)hipFree(ctx->ext_g);
      ctx->ext_g = nullptr;
      ctx->ext_cap = 0;
      if (hipMalloc(&ctx->ext_g, 2 * size_t(len) * 8) != hipSuccess) return fail(ctx, SVGP_OOM, "hipMalloc failed for the point gradients");
      ctx->ext_cap = size_t(len);
    }
    HIPC(ctx, hipMemcpyAsync(ctx->ext_g, gc.ext_gmu, size_t(len) * 8, hipMemcpyHostToDevice, s));
    HIPC(ctx, hipMemcpyAsync(ctx->ext_g + ctx->ext_cap, gc.ext_gv, size_t(len) * 8, hipMemcpyHostToDevice, s));
  }
  // The likelihood gradients come from point_grad_kernel behind the strips (both likelihood routes); A g_mu (the data part of m_bar): the
  // kernel-gradient reductions, which evaluate the kernel anyway, sum Kuf g_mu and the tail applies Lk^-1 (one gemv with the explicit
  // inverse), so A is read by the SYRK only (round 3, f64: 0.32 -> 0.2 ms per 65 536-point chunk at M = 1024).
  // the SYRK's weights 2 g_v are uniform over the points for the built-in Gaussian likelihood (grad.hip: UW) unless a variance was
  // negative and clamped (then that point's g_v differs... it does not: dE/dv = -1 / (2 sigma^2) whatever v) - so: Gaussian, built in
  const bool uniform_w = m->desc.likelihood == SVGP_LIK_GAUSSIAN && !gc.ext_gmu && m->gh_n == 0 && ctx->kn.syrk_uniform;   // (knob: experiments build)
  // Round 6: strips beside the factorisation + uniform weights = the SYRK needs nothing but A, which phase 1 has
  // written by the time the chain ends - so it goes on the second stream right behind the last phase-1 segment, BESIDE the M-sized prep
  // of phase 3 on the main stream (Lk^-1 by recursive doubling, R, alpha: ~230 us of small launches at M = 1024), instead of behind
  // phase 3 and the kernel-gradient reductions.  Same kernel, same slices, same single overwrite of the slice buffer: bitwise the
  // serial result.  The one M x M product that runs meanwhile (R) takes a split-K scratch of its own.  Measured (same box, update +
  // value and gradient, ms): 16 384 points M = 1024 f64 2.72 -> 2.67, fp32 1.94 -> 1.88; 4 096 points 1.87 -> 1.81 - the SYRK's 504
  // workgroups and the prep's small launches slow each other, so only a quarter of the 230 us comes back (profiles/round6/minibatch_trace.md).
  static const int early_knob = exp_int("SVGP_SYRK_EARLY", 1);   // (experiments build: 0 = the SYRK behind phase 3, as rounds 3-5)
  const bool syrk_early = gop.on && uniform_w && early_knob != 0;
  if (syrk_early && !w->Gmm) {
    const size_t gmm_b = size_t(std::min<int64_t>(Mp / 128, w->nslices)) * size_t(Mp) * size_t(Mp) * m->es;
    if (hipMalloc(&w->Gmm, gmm_b) != hipSuccess) { w->Gmm = nullptr; return fail(ctx, SVGP_OOM, "hipMalloc failed for the gradient workspace"); }
    w->all.push_back(w->Gmm);
  }
  // the strips' arguments for the chunk [c0, c0 + clen) (scratch / moment pointers: read after the ensure_scratch of the caller)
  auto strip_args = [&](int64_t c0, int64_t clen, LikParams& lpc) -> StripArgs {
    StripArgs a{};
    a.T = m->T; a.U = m->U; a.zs = m->zs; a.mp = m->mp; a.x = data->x; a.work = ctx->work; a.counter = ctx->counter;
    a.At_out = w->At; a.ldx = data->ldx; a.off = off + c0; a.len = clen; a.Mp = Mp; a.M = M; a.kp = kp;
    a.mean_const = m->desc.mean_const;
    a.mom_mu = ctx->mom; a.mom_var = ctx->mom + ctx->mom_cap;   // the strips' (mu, v), read by launch_point_grads
    a.R = w->Rcm; a.alpha = w->alpha; a.Pt_out = w->Pt;
    lpc = lp;
    if (gc.ext_gmu) { lpc.lik = kLikExternal; lpc.gh_x = ctx->ext_g + c0; lpc.gh_w = ctx->ext_g + ctx->ext_cap + c0; }   // (launch_point_grads' arguments)
    return a;
  };
  // The M-sized work of the adjoint that does NOT depend on the factorisation: the cleared accumulators, the inverse lengthscales for
  // the kernel-gradient kernels, S = B B' - I (B = the padded Lq: NonCentered) - ~60 us of small launches.  Serial path: behind the
  // prep on the main stream.  Strips beside the factorisation: on the second stream AHEAD of the pre-generation (that stream has
  // nothing to do until block row 0 of T exists, ~130 us into the chain), the main stream waits for ev_S before it forms R.
  // the adjoint runs on the whitened problem: (m~, B) are (m, Lq) for NonCentered and (Lk^-1 (m - c), Lk^-1 Lq) for Centered
  const void* Bq = centered ? m->B : w->Lqp;
  auto grad_pre_chain = [&](hipStream_t st) -> int {
    HIPC(ctx, hipMemsetAsync(w->zero_blk, 0, w->zero_b, st));   // every accumulator of the evaluation, one fill
    // (as kernel arguments: a hipMemcpyAsync from pageable memory blocks the host until the stream gets there - on the second stream
    // that was ~45 us during which the main stream's Kuu launch was not even enqueued)
    launch_setvec_f64(st, w->invl_d, m->invl_host.data(), m->d);
    if (!centered) launch_pad_lower(dt, st, m->Lq_raw, M, Mp, w->Lqp);
    gemm_mm(ctx, w, dt, st, Bq, Bq, Mp, w->G2, kMmXUp | kMmYUp);   // lower tiles of B B' (row-major); B[r][k] = 0 for k > r
    launch_sym_from_lower(dt, st, w->G2, 1, Mp, 1.0, w->tmp);     // S = B B' - I, full
    KCHECK(ctx, "grad prep (S)");
    return SVGP_OK;
  };
  std::function<int()> seg_pre = [&]() -> int {
    const int rcp = grad_pre_chain(ctx->stream2);
    if (rcp) return rcp;
    HIPC(ctx, hipEventRecord(ctx->ev_S, ctx->stream2));
    return SVGP_OK;
  };
  // segmented strips (gop.on): enqueued panel by panel from inside the factorisation's launch loop (SegRun), so built before the prep
  SegRun gseg;
  RowHook ghook{seg_row_hook, &gseg};
  LikParams lpc_seg{};
  if (gop.on) {
    gseg.ctx = ctx; gseg.m = m; gseg.op = gop; gseg.grad = true; gseg.nP = int(Mp / 128);
    gseg.a = strip_args(0, len, lpc_seg);
    gseg.a.seg_state = ctx->seg_state;
    gseg.a.counter = ctx->counter2;
    gseg.pre = &seg_pre;
  }
  ctx->overlapped = gop.on;
  TREC(ctx, ctx->ev[0], s);
  rc = enqueue_prep(ctx, m, gop.on, gop.on ? &ghook : nullptr);
  if (rc == SVGP_OK && gop.on) rc = gseg.rc;
  if (rc) {
    if (gop.on && ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);   // segments already enqueued: drain before the scratch is reused
    return rc;
  }
  TREC(ctx, ctx->ev[1], s);
  if (!gop.on) {
    rc = grad_pre_chain(s);
    if (rc) return rc;
  } else {
    HIPC(ctx, hipStreamWaitEvent(s, ctx->ev_S, 0));   // S = B B' - I and the cleared accumulators: done on the second stream beside the chain
  }
  // Linv = Lk^-1 (both storage orders): every Lk^-T . below is a GEMM with it (round 2: four blocked substitutions, 0.18 ms each
  // at M = 1024 whatever the batch size)
  launch_linv(dt, s, m->L, m->T, Mp, w->LinvRM, w->LinvCM, w->H);
  // M-sized operands of the strips' phase 3:  R = Lk^-T (B B' - I)  (column-major: the P operand of the GEMM); then, for the
  // kernel-gradient reductions behind the strips, alpha = Lk^-T m~ (no strip reads it: it follows ev_R)
  gemm_mm(ctx, w, dt, s, w->tmp, w->LinvRM, Mp, w->Rcm, kMmFull | kMmYLow, syrk_early ? w->Gmm : nullptr); // out[c][r] = sum_k S[k][c] Linv[k][r] = R[r][c]: R column-major
  KCHECK(ctx, "grad prep");
  if (gop.on) HIPC(ctx, hipEventRecord(ctx->ev_R, s));   // R is final: the segmented strips' closing launch (phase 3) may run
  launch_linv_t_gemv(dt, s, w->LinvRM, m->mp, Mp, w->alpha, w->gemv_part);
  // the strips' scratch (the A strip and, beside it, the Kuf strip) and the chunk's moments, sized for every chunk of the call BEFORE
  // anything of the loop is enqueued (ensure_scratch may reallocate): the full chunks and the shorter last one may plan differently
  auto chunk_plan = [&](int64_t clen, int& nt, int& grid, int64_t& nstrips) {
    const StripPlan plan = strip_plan_single(dt, Mp, clen, ctx->num_cus);
    nt = plan.grid ? plan.nt : plan.nt_tail;
    grid = plan.grid ? plan.grid : plan.grid_tail;
    nstrips = plan.grid ? plan.nstrips : plan.nstrips_tail;
  };
  size_t wb_max = 0;
  const int64_t nchunks = (len + nc - 1) / nc;
  for (int64_t clen : {std::min(len, nc), len - (nchunks - 1) * nc}) {
    int nt, grid; int64_t nstrips;
    chunk_plan(clen, nt, grid, nstrips);
    wb_max = std::max(wb_max, 2 * strip_work_bytes(dt, Mp, nt, grid));
    if (nstrips > w->part5_strips) return fail(ctx, SVGP_HIP_ERROR, "internal: strip partial buffer too small");
  }
  rc = ensure_scratch(ctx, wb_max, size_t(nc));
  if (rc) return rc;
  // chunk pipeline (ensure_pipe; experiments build)
  static const int kg_overlap = exp_int("SVGP_KGRAD_OVERLAP", 0);   // experiments build
  const PipeCfg pc = pipe_cfg(ctx);
  const bool pipe_any = !gop.on && !kg_overlap && nchunks >= 2 && pc.lanes >= 2;
  const bool pipe = pipe_any && ctx->kn.pipe_mode != 2;   // mode 1: strips(k + 1) on a pipeline stream beside the consumers of chunk k
  const bool trail = pipe_any && ctx->kn.pipe_mode == 2;  // mode 2: everything on the main stream but kgrad(k), which trails on a pipeline stream beside SYRK(k) and strips(k + 1)
  ctx->pipelined = pipe_any;
  if (pipe_any) {
    PipeCfg pcc = pc;
    if (trail) pcc.streams = 1;
    rc = ensure_pipe(ctx, w, m, pcc, wb_max, size_t(nc));
    if (rc) return rc;
    if (pipe) HIPC(ctx, hipEventRecord(ctx->ev_pipe_prep, s));   // R, alpha, the cleared accumulators: everything the strips read
  }
  for (int64_t c0 = 0, kc = 0; c0 < len; c0 += nc, ++kc) {
    const int64_t clen = (len - c0 < nc) ? len - c0 : nc;
    const int64_t ncp = (clen + 127) / 128 * 128;
    // forward strips + likelihood gradients + phase 3 in ONE launch: leaves A, P point-major and g_mu, g_v of the chunk
    int nt, grid; int64_t nstrips;
    chunk_plan(clen, nt, grid, nstrips);
    // this chunk's buffer set and the stream of its strips (serial path: lane 0 = the workspace's own arrays, the main stream)
    const int lane = pipe_any ? int(kc % pc.lanes) : 0, pq = pipe ? int(kc % pc.streams) : 0;
    GradWs::Lane L;
    if (pipe_any) L = w->lanes[size_t(lane)];
    else { L.At = w->At; L.Pt = w->Pt; L.gmu = w->gmu; L.gv = w->gv; L.partial5 = w->partial5; }
    hipStream_t ss = pipe ? ctx->pst[pq] : s;
    if (pipe) {
      if (kc < pc.streams) HIPC(ctx, hipStreamWaitEvent(ss, ctx->ev_pipe_prep, 0));
      if (kc >= pc.lanes) HIPC(ctx, hipStreamWaitEvent(ss, L.ev_done, 0));   // the lane's previous chunk has been consumed
    }
    if (trail && kc >= pc.lanes) HIPC(ctx, hipStreamWaitEvent(s, L.ev_done, 0));   // the reductions of the lane's previous chunk have read P / g
    // (g_mu | g_v, w->nc apart: the weighted SYRK reads g_v over the chunk padded to 128 points - launch_point_grads writes that padding)
    LikParams lpc{};
    StripArgs a = gop.on ? gseg.a : strip_args(c0, clen, lpc);
    if (gop.on) lpc = lpc_seg;
    if (!gop.on) {
      a.At_out = L.At; a.Pt_out = L.Pt;
      if (pipe && pq == 1) { a.work = ctx->pwork; a.counter = ctx->pcounter; a.mom_mu = ctx->pmom; a.mom_var = ctx->pmom + ctx->pmom_cap; }
    }
    if (gop.on) {   // (single chunk) the segments are on the second stream already; the closing launch (phase 2 + 3) waits for R and alpha,
                    // and the main stream joins before the point gradients
      hipStream_t s2 = ctx->stream2;
      const int nPn = int(Mp / 128);
      if (a.work != ctx->work || a.mom_mu != ctx->mom) return fail(ctx, SVGP_HIP_ERROR, "internal: scratch moved under the segmented strips");
      // A small batch has fewer strips than the chip has workgroup slots and phase 3 - 2 M^2 flops per point, one panel after the other
      // inside a strip's workgroup - is latency-bound on the few CUs it reaches (1024 points, M = 1024: 32 workgroups, 340 us for 27 us
      // of MFMA work).  Its output panels are independent: the closing launch runs S workgroups per strip (strip.hip: seg_split).
      // SVGP_SEG_SPLIT=0 (per call): the unsplit launch, whose result is bitwise the serial kernel's (the split one differs in the
      // variance's summation order).
      const int S = seg_split_factor(ctx, nPn, nstrips);
      int cgrid = grid;
      if (S > 1) {
        rc = seg_split_setup(ctx, a, S, nstrips);
        if (rc) return rc;
        cgrid = int(nstrips) * S;
      }
      if (syrk_early) {   // W = w A A' behind the last phase-1 segment, beside the main stream's Lk^-1 / R / alpha
        const int64_t n16 = (clen + 15) / 16 * 16, sl_e = ((ncp + ns_syrk - 1) / ns_syrk + 15) / 16 * 16;
        if (n16 > clen) HIPC(ctx, hipMemsetAsync(static_cast<char*>(L.At) + size_t(clen) * size_t(Mp) * es, 0, size_t(n16 - clen) * size_t(Mp) * es, s2));
        launch_syrk_uniform(dt, s2, L.At, -0.5 / lp.sigma2, scale, n_global_dev, gc.num_data, 2.0, Mp, n16, sl_e, ns_syrk, w->G1, 1);
        KCHECK(ctx, "syrk (beside the prep of phase 3)");
      }
      HIPC(ctx, hipStreamWaitEvent(s2, ctx->ev_R, 0));
      a.seg_lo = a.seg_hi = nPn;
      a.seg_flags = kSegLoad | kSegPhase2;
      launch_strip_seg(dt, s2, a, nt, cgrid, nstrips, true);
      KCHECK(ctx, "strip (value and gradient, segmented)");
      HIPC(ctx, hipEventRecord(ctx->ev_join, s2));
      HIPC(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
    } else {
      // the head of the strips' queue: zeroed here for the first chunk, by the previous chunk's launch_point_grads for the others
      if (kc == 0 || pipe_any) HIPC(ctx, hipMemsetAsync(a.counter, 0, sizeof(unsigned), ss));
      launch_strip_grad(dt, ss, a, nt, grid, nstrips);
      KCHECK(ctx, "strip (value and gradient)");
    }
    launch_point_grads(dt, ss, lpc, a.mom_mu, a.mom_var, gc.ext_gmu ? nullptr : data->y, off + c0, clen, scale, n_global_dev, gc.num_data, L.gmu, L.gv, L.partial5,
                       (!gop.on && !pipe_any) ? a.counter : nullptr, ncp);
    KCHECK(ctx, "point gradients");
    const int n5 = point_grad_blocks(clen);   // rows of partial5: one per 256-point block
    if (pipe) {   // the consumers of this chunk: on the main stream, behind the chunk's strips
      HIPC(ctx, hipEventRecord(L.ev_strips, ss));
      HIPC(ctx, hipStreamWaitEvent(s, L.ev_strips, 0));
    }
    if (trail) {
      HIPC(ctx, hipEventRecord(L.ev_strips, s));
      HIPC(ctx, hipStreamWaitEvent(ctx->pst[0], L.ev_strips, 0));
    }
    // Knob (off): the kernel-gradient reductions (f64 VALU, latency-bound, no MFMA) on the second stream BESIDE the SYRK
    // (MFMA-bound); both only read this chunk's A / P / g, the join comes before the next chunk's strips overwrite them.
    // Measured and not adopted: H 97.8-98.3 vs 98.2-98.4 ms, C5 16.5-16.6 vs 16.6-16.7 ms (same box) - the SYRK's 504
    // workgroups leave kgrad no room to run beside them.
    hipStream_t sk = trail ? ctx->pst[0] : s;
    if (kg_overlap) {
      rc = ensure_stream2(ctx);
      if (rc) return rc;
      sk = ctx->stream2;
      HIPC(ctx, hipEventRecord(ctx->ev_fork, s));
      HIPC(ctx, hipStreamWaitEvent(sk, ctx->ev_fork, 0));
    }
    int64_t ksl = ((clen + w->ns_uf - 1) / w->ns_uf + 127) / 128 * 128;
    launch_kgrad(dt, sk, kp, m->zs, Mp, M, data->x, data->ldx, off + c0, 0, clen, clen, L.Pt, L.gmu, L.gv, w->alpha, ksl, w->ns_uf,
                 w->rp_uf, w->sp_uf, 1);
    KCHECK(ctx, "kgrad uf");
    if (kg_overlap) HIPC(ctx, hipEventRecord(ctx->ev_join, sk));
    if (trail) HIPC(ctx, hipEventRecord(L.ev_done, sk));
    launch_sum5(s, L.partial5, n5, w->sums);
    int64_t sl = ((ncp + ns_syrk - 1) / ns_syrk + 15) / 16 * 16;   // as even as the 16-point k-step allows
    // W (+)= A diag(2 g_v) A' (lower tiles, split-K slices): the first chunk overwrites, so the slice buffer needs no zeroing
    if (syrk_early) {
      // (already on the second stream, joined above)
    } else if (uniform_w) {
      // g_v is the same for every point (Gaussian: -scale / (2 sigma^2)): the unweighted loop, the weight applied to the accumulators.
      // Columns of the chunk's last strip beyond its last point hold the replicated last point: zero them up to the k-step boundary
      const int64_t n16 = (clen + 15) / 16 * 16;
      if (n16 > clen) HIPC(ctx, hipMemsetAsync(static_cast<char*>(L.At) + size_t(clen) * size_t(Mp) * es, 0, size_t(n16 - clen) * size_t(Mp) * es, s));
      launch_syrk_uniform(dt, s, L.At, -0.5 / lp.sigma2, scale, n_global_dev, gc.num_data, 2.0, Mp, n16, sl, ns_syrk, w->G1, c0 == 0 ? 1 : 0);
    } else {
      launch_gemm_pm(dt, s, L.At, L.At, L.gv, 2.0, Mp, ncp, sl, ns_syrk, w->G1, c0 == 0 ? 1 : 0);
    }
    KCHECK(ctx, "syrk");
    if (kg_overlap) HIPC(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
    if (pipe) HIPC(ctx, hipEventRecord(L.ev_done, s));
  }
  if (trail)   // join: the trailing reductions of the last chunks
    for (int64_t q = 0; q < std::min<int64_t>(nchunks, pc.lanes); ++q) HIPC(ctx, hipStreamWaitEvent(s, w->lanes[size_t(q)].ev_done, 0));
  if (gc.ext_gmu) launch_add_f64(s, w->sums, gc.ext_sum_e);   // sums[0] = sum E: the host's, before any collective
  TREC(ctx, ctx->ev[2], s);
  // M-sized tail.  With W = A diag(2 g_v) A' and a = A g_mu:
  //   Lq_bar = tril(W B) - klw dKL/dB,   Lk_bar = -tril(alpha a' + R W)     (B = Lq whitened; W, R carry the factors 2)
  launch_sym_from_lower(dt, s, w->G1, ns_syrk, Mp, 0.0, w->W2);
  gemm_mm(ctx, w, dt, s, w->W2, m->U, Mp, w->G1p, kMmYLow);   // (W B)[r][c] = sum_i W[i][r] B[i][c]; B[i][c] = 0 for i < c
  gemm_mm(ctx, w, dt, s, w->Rcm, w->W2, Mp, w->G2);           // (R W)[r][c] = sum_i R[r][i] W[i][c]
  launch_avec(s, w->rp_uf, w->ns_uf, int64_t(2 + dreg) * Mp, Mp, w->avec);
  // avec holds Kuf g_mu: A g_mu = Lk^-1 (Kuf g_mu)
  launch_linv_t_gemv(dt, s, w->LinvCM, w->avec, Mp, w->avec, w->gemv_part, 1, 1);   // avec is fp64 in both builds
  launch_finish_mm2(dt, s, w->G1p, w->G2, w->alpha, w->avec, Mp, M, centered ? m->B : m->Lq_raw, centered ? Mp : M, klw, w->Lqbar,
                    centered ? w->BbarRM : nullptr, w->LbarRM);
  KCHECK(ctx, "Lq_bar / Lk_bar");
  if (centered) {
    // chain through m~ = Lk \ (m - c) and B = Lk \ Lq:  m_bar = Lk^-T m~_bar,  Rb = Lk^-T B_bar,  Lq_bar = tril(Rb),
    // Lk_bar -= tril(m_bar m~') + tril(Rb B')
    launch_mbar(dt, s, w->avec, m->mp, klw, M, Mp, w->rbar);
    launch_linv_t_gemv(dt, s, w->LinvRM, w->rbar, Mp, w->alpha, w->gemv_part);   // alpha is free after finish_mm2: holds m_bar (padded)
    HIPC(ctx, hipMemcpyAsync(w->mbar, w->alpha, size_t(M) * es, hipMemcpyDeviceToDevice, s));
    // out[c][r] = sum_i B_bar[i][c] Linv[i][r] = Rb[r][c]: Rb column-major = the k-major operand of Rb B' below
    gemm_mm(ctx, w, dt, s, w->BbarRM, w->LinvRM, Mp, w->tmp, kMmFull | kMmXLow | kMmYLow);
    launch_cm_tril_to_user(dt, s, w->tmp, Mp, M, w->Lqbar);
    gemm_mm(ctx, w, dt, s, w->tmp, m->B, Mp, w->Phi, kMmYUp);   // (Rb B')[r][c] = sum_i Rb[r][i] B[c][i]; B[c][i] = 0 for i > c
    launch_lbar_adjust(dt, s, w->LbarRM, w->Phi, w->alpha, m->mp, Mp);
    KCHECK(ctx, "centered chain");
  }
  // Cholesky backward: Kuu_bar = sym(Lk^-T Phi(Lk' Lk_bar) Lk^-1)
  launch_lower_to_rowmajor(dt, s, m->L, Mp, w->LkRM);
  gemm_mm(ctx, w, dt, s, w->LkRM, w->LbarRM, Mp, w->Phi, kMmXLow | kMmYLow);   // (Lk' Lk_bar)[r][c], lower tiles
  launch_phi(dt, s, w->Phi, Mp);                                               // tril, diagonal halved; zero above
  gemm_mm(ctx, w, dt, s, w->Phi, w->LinvRM, Mp, w->tmp, kMmFull | kMmXLow | kMmYLow);   // out[c][r] = sum_i Phi[i][c] Linv[i][r] = (Linv' Phi)[r][c]
  gemm_mm(ctx, w, dt, s, w->tmp, w->LinvRM, Mp, w->G1p, kMmFull | kMmYLow);             // out[r][c] = sum_j (Linv' Phi)[r][j] Linv[j][c]
  launch_symmetrize(dt, s, w->G1p, Mp, w->H);
  KCHECK(ctx, "chol backward");
  // the Kuu part: the ns_uu slices must cover all M columns (a fixed slice of 128 covered only 1024 of them: the kernel-
  // parameter and z gradients were wrong for M > 1024 until tests/test_gpu_grad.py::test_gradient_large_m_float32_strips)
  const int64_t uu_sl = ((M + w->ns_uu - 1) / w->ns_uu + 127) / 128 * 128;
  launch_kgrad(dt, s, kp, m->zs, Mp, M, m->zs, Mp, 0, 1, M, M, w->H, nullptr, nullptr, nullptr, uu_sl, w->ns_uu, w->rp_uu, w->sp_uu);
  launch_finish_kgrad(dt, s, m->d, M, Mp, m->zs, w->invl_d, w->rp_uf, w->ns_uf, w->rp_uu, w->ns_uu, w->sp_uf, w->ns_uf * w->rb,
                      w->sp_uu, w->ns_uu * w->rb, m->mp, klw, m->desc.layout_z, m->desc.variance, w->zbar, centered ? nullptr : w->mbar,
                      w->scal_out, w->kred, w->avec);
  // status slots of the all-reduced scalars + this rank's prep scalars / info behind them: ONE fp64 read-back per evaluation
  launch_grad_status(s, w->sums, m->info, double(len), m->scal, w->scal_out + 1 + dreg);
  KCHECK(ctx, "kgrad uu / finish");
  TREC(ctx, ctx->ev[3], s);
  return SVGP_OK;
}

// Opening all-reduce of a collective value-and-gradient call: {global batch size, failure flag}, summed over the ranks ON THE DEVICE
// (the strips' phase 3 reads num_data / n_global there).  Asynchronous since round 3: the two values travel as kernel arguments, the
// 16-byte ncclAllReduce is enqueued behind them.  Round 6: it is the ONE collective of the call whose size does not
// depend on the model, so EVERY rank enters it - also one that cannot evaluate at all (NULL model, bad arguments, no memory for the
// gradient workspace): that rank sends {0, 1}, returns its own error and joins nothing else.  The reduced flag comes back through a
// pinned host word behind an event (grad_peers_failed); a healthy rank reads it after it has enqueued its backward pass - by then the
// 16 bytes have long arrived, so the wait costs nothing - and, if any peer failed, SKIPS the closing gradient all-reduce and returns
// SVGP_RCCL_ERROR.  No rank is left inside a collective a peer will never enter, and nothing is aborted.  (Rounds 2-5: the failing rank
// had to match the closing all-reduce with M^2-sized zeros, and without a model - the element counts unknown - it aborted its
// communicator, which does not wake the peers: a documented hang.)
int grad_handshake(svgp_ctx* ctx, GradCall& gc, int64_t len, bool failed) {
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, SVGP_HIP_ERROR, "hipSetDevice failed");
  if (!ctx->h_open && hipHostMalloc(reinterpret_cast<void**>(&ctx->h_open), 64, hipHostMallocDefault) != hipSuccess) {
    ctx->h_open = nullptr;
    return fail(ctx, SVGP_OOM, "hipHostMalloc failed for the opening all-reduce's flag");
  }
  if (!ctx->ev_open && hipEventCreateWithFlags(&ctx->ev_open, hipEventDisableTiming) != hipSuccess) {
    ctx->ev_open = nullptr;
    return fail(ctx, SVGP_HIP_ERROR, "hipEventCreate failed");
  }
  launch_set2_f64(ctx->stream, ctx->d_coll, failed ? 0.0 : double(len), failed ? 1.0 : 0.0);
  if (hipGetLastError() != hipSuccess) return fail(ctx, SVGP_HIP_ERROR, "launch failed in the opening all-reduce");
  const int rc = comm_allreduce(ctx, ctx->d_coll, 2, SVGP_F64);
  if (rc != SVGP_OK) return rc;
  *ctx->h_open = -1.0;
  if (hipMemcpyAsync(ctx->h_open, ctx->d_coll + 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipEventRecord(ctx->ev_open, ctx->stream) != hipSuccess)
    return fail(ctx, SVGP_HIP_ERROR, "read-back of the opening all-reduce's flag failed");
  gc.n_global_dev = ctx->d_coll;
  return SVGP_OK;
}
// number of ranks that entered the opening all-reduce with the failure flag (blocks until that all-reduce has completed); < 0: unknown
double grad_peers_failed(svgp_ctx* ctx) {
  if (!ctx->ev_open || hipEventSynchronize(ctx->ev_open) != hipSuccess) return -1.0;
  return *ctx->h_open;
}

// A rank of a collective svgp_elbo_grad / svgp_elbo_grad_ext that failed BEFORE anything could be enqueued (argument checks, NULL model,
// workspace allocation): it enters the opening all-reduce with the failure flag - its peers then skip the closing one and return
// SVGP_RCCL_ERROR - and returns its own error.  Only if even that 16-byte collective cannot be issued is the communicator aborted.
int grad_fail_collective(svgp_ctx* ctx, int pre_rc) {
  const std::string keep = ctx->err;
  GradCall gc;
  if (grad_handshake(ctx, gc, 0, true) != SVGP_OK) comm_abort(ctx);
  else (void)hipStreamSynchronize(ctx->stream);
  ctx->err = keep;
  return pre_rc;
}

// the gradient's ONE (grouped) all-reduce; a rank whose enqueue failed still takes part with its failure flag set
int grad_collective(svgp_ctx* ctx, svgp_model* m, GradCall& gc, int local_rc) {
  if (!ctx->comm || !gc.collective) return local_rc;
  GradWs* w = gc.w;
  if (!w) {   // failed before the workspace existed: nothing to reduce with; make the peers fail instead of hang
    comm_abort(ctx);
    return local_rc;
  }
  const std::string keep = ctx->err;
  const int dreg = grad_dreg(m->d);
  if (local_rc != SVGP_OK) {
    std::vector<double> poison(size_t(8 + 1 + dreg), 0.0);
    poison[7] = 1.0;
    if (hipSetDevice(ctx->device) != hipSuccess ||
        hipMemcpyAsync(w->sums, poison.data(), poison.size() * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
      comm_abort(ctx);
      ctx->err = keep;
      return local_rc;
    }
  }
  // {z_bar | m_bar | tril(Lq_bar)}: the lower triangle packed (M (M + 1) / 2 entries instead of M^2: SURVEY 8 f1), one all-reduce in
  // the compute dtype, one in fp64 for the scalars, then unpacked in place
  const int64_t M = m->M, head = M * m->d + M;
  const size_t ncoll = size_t(head) + size_t(M) * size_t(M + 1) / 2;
  if (hipSetDevice(ctx->device) == hipSuccess) launch_pack_tril(m->dtype, ctx->stream, w->gblk, w->cblk, head, M, 0);
  int rc = comm_group_start(ctx);
  if (rc == SVGP_OK) rc = comm_allreduce(ctx, w->cblk, ncoll, m->dtype);
  if (rc == SVGP_OK) rc = comm_allreduce(ctx, w->sums, size_t(8 + 1 + dreg), SVGP_F64);
  const int rce = comm_group_end(ctx);
  if (rc == SVGP_OK) rc = rce;
  // (the reduced block stays packed: grad_finish reads {z_bar | m_bar | packed tril(Lq_bar)} back as it is)
  if (rc == SVGP_OK) gc.packed = true;
  if (rc != SVGP_OK) {
    comm_abort(ctx);
    if (local_rc != SVGP_OK) ctx->err = keep;
    return local_rc != SVGP_OK ? local_rc : rc;
  }
  if (local_rc != SVGP_OK) ctx->err = keep;
  return local_rc;
}

int grad_finish(svgp_ctx* ctx, svgp_model* m, GradCall& gc, double* elbo_out, svgp_terms* terms_out, svgp_grads* g) {
  hipStream_t s = ctx->stream;
  HIPC(ctx, hipSetDevice(ctx->device));
  GradWs* w = gc.w;
  const int dt = m->dtype;
  const int64_t M = m->M;
  const size_t es = m->es;
  const int dreg = grad_dreg(m->d);
  const bool centered = gc.centered;
  // Read back: the fp64 block [sums (8) | scal_out (1 + dreg) | prep scalars (4) | chol_info] and the gradient blocks
  // {z_bar | m_bar | Lq_bar}, contiguous on the device (round 2: seven copies, ~20 us of host latency each).  Lq_bar is M^2 elements -
  // 8.4 MB at M = 1024 f64, 33.5 MB at M = 2048 - and a minibatch step is 1-3 ms of device time: until late in round 4 this went through a
  // fresh std::vector per call (page faults + zero fill), a pageable device-to-host copy and a second host copy into the caller's
  // buffers (M = 2048: 6.6 ms of an 11.7 ms call).  Round 4: a pinned staging buffer kept by the context, the copy issued in up to 8
  // pieces with an event behind each, and the host copy of piece i into the caller's buffers running while piece i + 1 is on the bus.
  // Round 5: Lq_bar is lower triangular, so only its M (M + 1) / 2 entries cross the bus - packed by columns on the
  // device (pack_tril_kernel, the all-reduce's layout) and scattered into the caller's dense column-major M x M array by the host
  // copy, which also writes the zeros above the diagonal.  The C-ABI keeps its dense Lq_bar.
  const size_t nz = size_t(M) * m->d, head = nz + size_t(M), tri = size_t(M) * size_t(M + 1) / 2, nblk = head + tri;
  const size_t f64n = size_t(8 + 1 + dreg + 5), f64b = (f64n * 8 + 255) / 256 * 256, gbytes = nblk * es;
  if (!gc.packed) launch_pack_tril(dt, s, w->gblk, w->cblk, int64_t(head), M, 0);
  KCHECK(ctx, "pack Lq_bar");
  if (f64b + gbytes > ctx->hstage_bytes) {
    if (ctx->hstage) (void)hipHostFree(ctx->hstage);
    ctx->hstage = nullptr;
    ctx->hstage_bytes = 0;
    HIPC(ctx, hipHostMalloc(&ctx->hstage, f64b + gbytes, hipHostMallocDefault));
    ctx->hstage_bytes = f64b + gbytes;
  }
  for (auto& e : ctx->ev_piece)
    if (!e) HIPC(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  double* f64blk = static_cast<double*>(ctx->hstage);
  char* gh = static_cast<char*>(ctx->hstage) + f64b;
  HIPC(ctx, hipMemcpyAsync(f64blk, w->sums, f64n * 8, hipMemcpyDeviceToHost, s));
  const int npiece = int(std::min<size_t>(8, std::max<size_t>(1, gbytes >> 20)));   // >= 1 MiB per piece
  const size_t piece = ((gbytes + npiece - 1) / npiece + 255) / 256 * 256;
  for (int q = 0; q < npiece; ++q) {
    const size_t lo = std::min(gbytes, size_t(q) * piece), hi = std::min(gbytes, lo + piece);
    if (hi > lo) HIPC(ctx, hipMemcpyAsync(gh + lo, static_cast<const char*>(w->cblk) + lo, hi - lo, hipMemcpyDeviceToHost, s));
    HIPC(ctx, hipEventRecord(ctx->ev_piece[q], s));
  }
  struct Dst { size_t lo, hi; void* p; };
  const Dst dst[2] = {{0, nz * es, g->z}, {nz * es, head * es, g->m}};
  int64_t col = 0;   // column of Lq_bar the packed stream has reached (pieces arrive in order)
  for (int q = 0; q < npiece; ++q) {
    HIPC(ctx, hipEventSynchronize(ctx->ev_piece[q]));
    const size_t lo = std::min(gbytes, size_t(q) * piece), hi = std::min(gbytes, lo + piece);
    for (const Dst& d : dst) {
      const size_t a = std::max(lo, d.lo), b = std::min(hi, d.hi);
      if (d.p && b > a) memcpy(static_cast<char*>(d.p) + (a - d.lo), gh + a, b - a);
    }
    // packed entries [e_lo, e_hi) of the triangle (piece boundaries are multiples of 256 bytes, hence of the element size):
    // column c holds the rows c .. M - 1 at st(c) = c M - c (c - 1) / 2; its part above the diagonal is zeroed as the column begins
    if (g->Lq && hi > head * es && hi > lo) {
      const int64_t e_lo = int64_t((std::max(lo, head * es) - head * es) / es), e_hi = int64_t((hi - head * es) / es);
      auto st = [M](int64_t c) { return c * M - c * (c - 1) / 2; };
      char* Lq = static_cast<char*>(g->Lq);
      const char* pk = gh + head * es;
      while (col + 1 < M && st(col + 1) <= e_lo) ++col;
      for (int64_t c = col; c < M && st(c) < e_hi; ++c) {
        const int64_t a = std::max(e_lo, st(c)), b = std::min(e_hi, st(c + 1));
        if (a == st(c) && c > 0) memset(Lq + size_t(c) * size_t(M) * es, 0, size_t(c) * es);
        if (b > a) memcpy(Lq + (size_t(c) * size_t(M) + size_t(c + (a - st(c)))) * es, pk + size_t(a) * es, size_t(b - a) * es);
        col = c;
      }
    }
  }
  HIPC(ctx, hipStreamSynchronize(s));
  const double* sums = f64blk;
  const double* sc = sums + 8;
  PrepScalars ps;
  for (int q = 0; q < 4; ++q) ps.scal[q] = sums[8 + 1 + dreg + q];
  ps.info = int(sums[8 + 1 + dreg + 4]);
  const char* mhost = gh + nz * es;
  finish_prep(m, ps);
  float t01 = 0, t13 = 0;
  (void)elapsed_ms(ctx, &t01, ctx->ev[0], ctx->ev[1]);
  (void)elapsed_ms(ctx, &t13, ctx->ev[1], ctx->ev[3]);
  ctx->timing = svgp_timing{};
  ctx->timing.ms_prep = t01;
  ctx->timing.ms_strip = t13;
  ctx->timing.ms_total = t01 + t13;
  float tch = 0;
  (void)elapsed_ms(ctx, &tch, ctx->ev_chol[0], ctx->ev_chol[1]);
  ctx->timing.ms_chol = tch;
  ElboRead r;
  r.E = sums[0]; r.n_neg = sums[4]; r.n_points = sums[5]; r.bad_chol = gc.collective ? sums[6] : 0.0; r.failed = gc.collective ? sums[7] : 0.0;
  // collective: the sums are global; scale = num_data / n_global (as on the device), and the KL counts once
  const double scale = gc.collective ? (gc.num_data > 0 ? gc.num_data / r.n_points : 1.0) : gc.scale;
  const double klw = gc.collective ? 1.0 : gc.klw;
  const double elbo = r.E * scale - klw * m->kl;
  g->variance = sc[0] + sums[2];
  g->lik_sigma2 = sums[3];
  double msum = 0.0;   // Centered: mean_const also enters through m~ = Lk \\ (m - c)
  if (centered)
    for (int64_t i = 0; i < M; ++i) msum += (dt == SVGP_F64) ? reinterpret_cast<const double*>(mhost)[i] : double(reinterpret_cast<const float*>(mhost)[i]);
  g->mean_const = sums[1] - msum;
  if (g->inv_lengthscale)
    for (int f = 0; f < m->d; ++f) g->inv_lengthscale[f] = sc[1 + f];
  fill_terms(terms_out, m, elbo, r, scale);
  const int rc = status_of(ctx, m, r.n_neg, r.bad_chol, r.failed);
  if (elbo_out) *elbo_out = (rc == SVGP_OK) ? elbo : NAN;
  return rc;
}

int elbo_grad_impl(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len, double scale, double klw,
                   double num_data, bool collective, double* elbo_out, svgp_terms* terms_out, svgp_grads* g,
                   const double* ext_gmu = nullptr, const double* ext_gv = nullptr, double ext_sum_e = 0.0) {
  GradCall gc;
  gc.ext_gmu = ext_gmu; gc.ext_gv = ext_gv; gc.ext_sum_e = ext_sum_e;
  gc.collective = collective && ctx && ctx->comm;
  gc.scale = scale;
  gc.klw = gc.collective ? 1.0 / double(ctx->world) : klw;
  gc.num_data = num_data;
  int rc = check_batch(ctx, m, data, off, len, ext_gmu == nullptr);   // a host-evaluated likelihood needs no y on the device
  if (rc == SVGP_OK && !g) rc = fail(ctx, SVGP_INVALID_ARG, "null gradient output");
  if (rc == SVGP_OK && (!(scale > 0.0) || !(klw >= 0.0))) rc = fail(ctx, SVGP_INVALID_ARG, "scale must be positive and kl_weight non-negative");
  if (rc == SVGP_OK && hipSetDevice(ctx->device) != hipSuccess) rc = fail(ctx, SVGP_HIP_ERROR, "hipSetDevice failed");
  if (rc == SVGP_OK) rc = grad_workspace(ctx, m, len, &gc.w);
  if (rc != SVGP_OK) return (ctx && gc.collective) ? grad_fail_collective(ctx, rc) : rc;
  if (gc.collective) {
    rc = grad_handshake(ctx, gc, len, false);
    if (rc != SVGP_OK) {   // the opening all-reduce itself could not be issued: nothing sane can follow on this communicator
      comm_abort(ctx);
      return rc;
    }
  }
  rc = grad_enqueue(ctx, m, data, off, len, gc);
  if (gc.collective) {
    const double nfail = grad_peers_failed(ctx);
    if (nfail != 0.0) {   // a peer failed before its backward pass (or the flag could not be read): it will not enter the closing all-reduce
      (void)hipStreamSynchronize(ctx->stream);
      if (ctx->overlapped && ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
      if (nfail < 0.0) { comm_abort(ctx); return rc != SVGP_OK ? rc : fail(ctx, SVGP_RCCL_ERROR, "the opening all-reduce of svgp_elbo_grad did not complete"); }
      return rc != SVGP_OK ? rc : fail(ctx, SVGP_RCCL_ERROR, "a peer rank failed before its backward pass (opening all-reduce of svgp_elbo_grad): the gradient all-reduce was skipped on every rank");
    }
  }
  rc = grad_collective(ctx, m, gc, rc);
  if (rc) return rc;
  return grad_finish(ctx, m, gc, elbo_out, terms_out, g);
}
}  // namespace

// With a communicator attached this call is COLLECTIVE like svgp_elbo: value and gradient of the GLOBAL minibatch ELBO on
// every rank (one 8-byte all-reduce of the batch size up front, one grouped all-reduce of the gradient at the end).
extern "C" int32_t svgp_elbo_grad(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len,
                                  double num_data, double* elbo_out, svgp_terms* terms_out, svgp_grads* g) {
  if (!ctx) return SVGP_INVALID_ARG;
  const double scale = len >= 1 ? (num_data > 0 ? num_data : double(len)) / double(len) : 1.0;   // len < 1 fails check_batch
  return elbo_grad_impl(ctx, m, data, off, len, scale, 1.0, num_data, true, elbo_out, terms_out, g);
}

// always local (no collective): the building block for hosts that run their own all-reduce
extern "C" int32_t svgp_elbo_grad_shard(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len,
                                        double scale, double kl_weight, double* value_out, svgp_terms* terms_out,
                                        svgp_grads* g) {
  return elbo_grad_impl(ctx, m, data, off, len, scale, kl_weight, 0.0, false, value_out, terms_out, g);
}

// ---- likelihoods the ABI does not enumerate ("generic GH for user link functions") -------------------
// The M^2 N work does not depend on the likelihood: only (mu_i, v_i) -> E_i does, and that is O(N) scalar work the host can do
// with ANY single-latent GPLikelihoods likelihood and quadrature.  svgp_marginals hands the host marginals(f_post(x)) of
// SVA:354; the host evaluates expected_loglikelihood (SVA:355) and, for training, its derivatives w.r.t. (mu_i, v_i);
// svgp_elbo_grad_ext runs the same backward pass as svgp_elbo_grad with those point gradients in place of lik.hpp's.
extern "C" int32_t svgp_marginals(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len,
                                  double* mean_out, double* var_out) {
  int rc = check_batch(ctx, m, data, off, len, false);
  if (rc) return rc;
  if (!mean_out || !var_out) return fail(ctx, SVGP_INVALID_ARG, "null output");
  hipStream_t s = ctx->stream;
  HIPC(ctx, hipSetDevice(ctx->device));
  TREC(ctx, ctx->ev[0], s);
  rc = enqueue_prep(ctx, m);
  if (rc) return rc;
  TREC(ctx, ctx->ev[1], s);
  StripOuts so;
  so.skip_expect = true;
  rc = enqueue_strips(ctx, m, data->x, data->ldx, nullptr, off, len, so);
  if (rc) return rc;
  PrepScalars ps;
  HIPC(ctx, hipMemcpyAsync(mean_out, ctx->mom, size_t(len) * 8, hipMemcpyDeviceToHost, s));
  HIPC(ctx, hipMemcpyAsync(var_out, ctx->mom + ctx->mom_cap, size_t(len) * 8, hipMemcpyDeviceToHost, s));
  HIPC(ctx, hipMemcpyAsync(ps.scal, m->scal, sizeof(ps.scal), hipMemcpyDeviceToHost, s));
  HIPC(ctx, hipMemcpyAsync(&ps.info, m->info, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPC(ctx, hipStreamSynchronize(s));
  finish_prep(m, ps);
  float t01 = 0, t12 = 0;
  (void)elapsed_ms(ctx, &t01, ctx->ev[0], ctx->ev[1]);
  (void)elapsed_ms(ctx, &t12, ctx->ev[1], ctx->ev[2]);
  ctx->timing = svgp_timing{};
  ctx->timing.ms_prep = t01;
  ctx->timing.ms_strip = t12;
  ctx->timing.ms_total = t01 + t12;
  double nneg = 0;
  const bool clamp = m->desc.neg_var_policy == SVGP_NEGVAR_CLAMP;
  for (int64_t i = 0; i < len; ++i) {   // FiniteGP(f_post, x, 1e-18): the variance the reference's marginals() hold
    double v = var_out[i] + kDefaultSigma2;
    if (v < 0.0) { nneg += 1; if (clamp) v = 0.0; }
    var_out[i] = v;
  }
  return status_of(ctx, m, nneg);
}

// Value and gradient of  scale sum_e - kl_weight KL  with (dE_i/dmu_i, dE_i/dv_i) = (g_mu[i], g_v[i]) supplied by the host
// (fp64, unscaled, one per point of the batch), scale = num_data / len.  Everything else - outputs, statuses, the collective
// form on a context with a communicator (sum_e and the gradients are then this rank's shard's) - is svgp_elbo_grad's;
// grads->lik_sigma2 is 0 (the likelihood's own parameters are the host's).
extern "C" int32_t svgp_elbo_grad_ext(svgp_ctx* ctx, svgp_model* m, const svgp_data* data, int64_t off, int64_t len,
                                      double num_data, double sum_e, const double* g_mu, const double* g_v,
                                      double* elbo_out, svgp_terms* terms_out, svgp_grads* g) {
  if (!ctx) return SVGP_INVALID_ARG;
  if (!g_mu || !g_v) {
    const int rc = fail(ctx, SVGP_INVALID_ARG, "null point gradients");
    return ctx->comm ? grad_fail_collective(ctx, rc) : rc;   // the peers learn it from the opening all-reduce: they return SVGP_RCCL_ERROR
  }
  const double scale = len >= 1 ? (num_data > 0 ? num_data : double(len)) / double(len) : 1.0;
  return elbo_grad_impl(ctx, m, data, off, len, scale, 1.0, num_data, true, elbo_out, terms_out, g, g_mu, g_v, sum_e);
}

// ================================================================================================
// One process, several GPUs (the shape of a Julia host): the group's member contexts each hold one shard of the data and
// a replica of the model; an evaluation enqueues every device, then issues the members' all-reduces as ONE ncclGroup,
// then reads device 0.  Same kernels, same collective, same result as the process-per-GPU path.
extern "C" {

int32_t svgp_group_data_upload(svgp_group* g, int32_t dtype, int32_t layout, int32_t d, int64_t n, const void* x_host,
                               const void* y_host, svgp_data** shards_out) {
  if (!g || !shards_out || !x_host || n < int64_t(g->ctxs.size())) return SVGP_INVALID_ARG;
  if (dtype != SVGP_F64 && dtype != SVGP_F32) return SVGP_INVALID_ARG;
  const int W = int(g->ctxs.size());
  const size_t es = esize(dtype);
  for (int i = 0; i < W; ++i) shards_out[i] = nullptr;
  int rc = SVGP_OK;
  for (int i = 0; i < W && rc == SVGP_OK; ++i) {
    // contiguous shard [lo, hi): the first n % W members get one more point
    const int64_t base = n / W, rem = n % W, lo = i * base + (i < rem ? i : rem), cnt = base + (i < rem ? 1 : 0);
    const char* xp = static_cast<const char*>(x_host);
    const char* yp = static_cast<const char*>(y_host);
    std::vector<char> gather;
    const void* xs = nullptr;
    if (layout == SVGP_ROWVECS && d > 1) {   // n x d column-major: a shard's rows are strided; gather them
      gather.resize(size_t(cnt) * d * es);
      for (int f = 0; f < d; ++f) memcpy(gather.data() + size_t(f) * cnt * es, xp + (size_t(f) * n + lo) * es, size_t(cnt) * es);
      xs = gather.data();
    } else {
      xs = xp + size_t(lo) * (layout == SVGP_COLVECS ? d : 1) * es;
    }
    rc = svgp_data_upload(g->ctxs[size_t(i)], dtype, layout, d, cnt, xs, yp ? yp + size_t(lo) * es : nullptr, &shards_out[i]);
    if (rc) g->err = svgp_last_error(g->ctxs[size_t(i)]);
  }
  if (rc != SVGP_OK)
    for (int i = 0; i < W; ++i) {
      svgp_data_free(g->ctxs[size_t(i)], shards_out[i]);
      shards_out[i] = nullptr;
    }
  return rc;
}

int32_t svgp_group_model_create(svgp_group* g, const svgp_model_desc* desc, svgp_model** models_out) {
  if (!g || !models_out) return SVGP_INVALID_ARG;
  const int W = int(g->ctxs.size());
  for (int i = 0; i < W; ++i) models_out[i] = nullptr;
  int rc = SVGP_OK;
  for (int i = 0; i < W && rc == SVGP_OK; ++i) {
    rc = svgp_model_create(g->ctxs[size_t(i)], desc, &models_out[i]);
    if (rc) g->err = svgp_last_error(g->ctxs[size_t(i)]);
  }
  if (rc != SVGP_OK)
    for (int i = 0; i < W; ++i) {
      svgp_model_free(g->ctxs[size_t(i)], models_out[i]);
      models_out[i] = nullptr;
    }
  return rc;
}

int32_t svgp_group_model_update(svgp_group* g, svgp_model* const* models, const svgp_model_desc* desc) {
  if (!g || !models) return SVGP_INVALID_ARG;
  for (size_t i = 0; i < g->ctxs.size(); ++i) {
    const int rc = svgp_model_update(g->ctxs[i], models[i], desc);
    if (rc) { g->err = svgp_last_error(g->ctxs[i]); return rc; }
  }
  return SVGP_OK;
}

const char* svgp_group_last_error(const svgp_group* g) { return g ? g->err.c_str() : "null group"; }

// elbo over the members' batches: member i evaluates points [offs[i], offs[i] + lens[i]) of shards[i]
int32_t svgp_group_elbo(svgp_group* g, svgp_model* const* models, const svgp_data* const* shards, const int64_t* offs,
                        const int64_t* lens, double num_data, double* elbo_out, svgp_terms* terms_out) {
  if (!g || !models || !shards || !offs || !lens) return SVGP_INVALID_ARG;
  const size_t W = g->ctxs.size();
  std::vector<int> rcs(W, SVGP_OK);
  // one process knows every member: a member that cannot take part is found BEFORE anything is enqueued and nothing touches
  // the communicator (a failing member used to leave the others' all-reduces waiting for it inside one ncclGroup)
  for (size_t i = 0; i < W; ++i) {
    const int rci = check_batch(g->ctxs[i], models[i], shards[i], offs[i], lens[i], true);
    if (rci != SVGP_OK) { g->err = svgp_last_error(g->ctxs[i]); return rci; }
  }
  for (size_t i = 0; i < W; ++i) rcs[i] = elbo_enqueue(g->ctxs[i], models[i], shards[i], offs[i], lens[i]);
  int rc = comm_group_start(g->ctxs[0]);
  if (rc == SVGP_OK) {
    for (size_t i = 0; i < W; ++i) rcs[i] = elbo_collective(g->ctxs[i], rcs[i]);
    rc = comm_group_end(g->ctxs[0]);
  }
  ElboRead r0;
  for (size_t i = 0; i < W; ++i) {
    ElboRead r;
    if (rcs[i] == SVGP_OK) rcs[i] = elbo_finish(g->ctxs[i], models[i], &r);
    if (i == 0) r0 = r;
    if (rcs[i] != SVGP_OK && rc == SVGP_OK) { rc = rcs[i]; g->err = svgp_last_error(g->ctxs[i]); }
  }
  if (rc != SVGP_OK) return rc;
  svgp_model* m = models[0];
  const double scale = (num_data > 0 ? num_data : r0.n_points) / r0.n_points;
  const double elbo = r0.E * scale - m->kl;
  fill_terms(terms_out, m, elbo, r0, scale);
  rc = status_of(g->ctxs[0], m, r0.n_neg, r0.bad_chol, r0.failed);
  if (rc) g->err = svgp_last_error(g->ctxs[0]);
  if (elbo_out) *elbo_out = (rc == SVGP_OK) ? elbo : NAN;
  return rc;
}

// value and gradient of the same global ELBO; the host knows the global batch size, so only the final grouped all-reduce runs
int32_t svgp_group_elbo_grad(svgp_group* g, svgp_model* const* models, const svgp_data* const* shards, const int64_t* offs,
                             const int64_t* lens, double num_data, double* elbo_out, svgp_terms* terms_out,
                             svgp_grads* grads_out) {
  if (!g || !models || !shards || !offs || !lens || !grads_out) return SVGP_INVALID_ARG;
  const size_t W = g->ctxs.size();
  int64_t n_global = 0;
  for (size_t i = 0; i < W; ++i) n_global += lens[i];
  if (n_global < 1) return SVGP_INVALID_ARG;
  std::vector<int> rcs(W, SVGP_OK);
  std::vector<GradCall> gcs(W);
  // validate EVERY member (arguments and gradient workspace) before anything is enqueued: if one cannot join, no member
  // issues a collective and the error is returned with the communicator untouched (the others used to hang)
  for (size_t i = 0; i < W; ++i) {
    int rci = check_batch(g->ctxs[i], models[i], shards[i], offs[i], lens[i], true);
    if (rci == SVGP_OK && hipSetDevice(g->ctxs[i]->device) != hipSuccess) rci = fail(g->ctxs[i], SVGP_HIP_ERROR, "hipSetDevice failed");
    if (rci == SVGP_OK) rci = grad_workspace(g->ctxs[i], models[i], lens[i], &gcs[i].w);
    if (rci != SVGP_OK) { g->err = svgp_last_error(g->ctxs[i]); return rci; }
  }
  for (size_t i = 0; i < W; ++i) {
    GradCall& gc = gcs[i];
    gc.scale = (num_data > 0 ? num_data : double(n_global)) / double(n_global);
    gc.klw = 1.0 / double(W);
    gc.num_data = num_data;
    gc.collective = false;   // scale known on the host: no batch-size all-reduce
    rcs[i] = grad_enqueue(g->ctxs[i], models[i], shards[i], offs[i], lens[i], gc);
    gc.collective = true;    // ... but the final reduction and the global bookkeeping of grad_finish apply
  }
  int rc = comm_group_start(g->ctxs[0]);
  if (rc == SVGP_OK) {
    for (size_t i = 0; i < W; ++i) rcs[i] = grad_collective(g->ctxs[i], models[i], gcs[i], rcs[i]);
    rc = comm_group_end(g->ctxs[0]);
  }
  svgp_grads scratch{};   // members other than 0 only need their status
  for (size_t i = 0; i < W; ++i) {
    if (rcs[i] == SVGP_OK) {
      svgp_grads* out = (i == 0) ? grads_out : &scratch;
      rcs[i] = grad_finish(g->ctxs[i], models[i], gcs[i], i == 0 ? elbo_out : nullptr, i == 0 ? terms_out : nullptr, out);
    }
    if (rcs[i] != SVGP_OK && rc == SVGP_OK) { rc = rcs[i]; g->err = svgp_last_error(g->ctxs[i]); }
  }
  return rc;
}

}  // extern "C"
