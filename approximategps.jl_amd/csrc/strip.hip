// strip.hip — the data-parallel hot path of elbo(sva, lfx, y) (reference
// src/SparseVariationalApproximationModule.jl:340-360 -> :246-253 -> :215-219) as ONE fused kernel.
//
// One 256-thread workgroup owns a strip of NT data points (columns) at a time and never lets the
// M x NT blocks Kuf, A = Lk \ Kuf and B'A reach HBM as whole matrices:
//   phase 1 (trsm, SVA:217):  for each 128-row panel I
//        A_I = inv(L_II) K_I - (inv(L_II) L_I,<I) A_<I  =  T[I, 0:(I+1)128] * [A_<I ; K_I]
//      one MFMA GEMM whose last 128 k-rows are GENERATED from x and z (Kuf assembly, SVA:216) straight
//      into LDS; A_I goes to a per-workgroup scratch strip (L2/MALL resident), and
//      colsumsq(A), A'm (= Kuf'α, SVA:250) accumulate in registers.
//   phase 2 (trmm, SVA:251):  C_J = U[J, J*128:Mp] * A_>=J with U = B' ; colsumsq(C) in registers.
//   epilogue: v = k(x,x) - Σ A² + Σ C² + 1e-18 (SVA:251, :354), expected log-likelihood per point
//      (GPLikelihoods), deterministic per-strip sum.
// Roofline: MFMA-bound (2 Mp² flops per point); algorithmic HBM bytes are only x, y.
#define SVGP_DIAG_TU_STRIP
#include "device_common.hpp"
#include "kernels.hpp"
#include "knobs.hpp"
#include "lik.hpp"

#ifndef SVGP_ASYNC
#define SVGP_ASYNC 1   // f64 64-point strips: both operand tiles by LDS-DMA through three LDS buffers (TileGemm::loop_tri_async)
#endif
#ifndef SVGP_TRI
#define SVGP_TRI 3   // bit 1: skip zero tiles of T diagonal blocks (phase 1), bit 2: of U diagonal blocks (phase 2)
#endif

#include <cstdlib>
#include <type_traits>

namespace svgp {

namespace {

// Kuf block of one strip (Mp inducing rows x NT points) written to the workgroup's scratch strip, k-major [Mp][NT].
// Pairwise distances on the MFMA as in kuf_kernel (r2 = |x|^2 + |z|^2 - 2 x.z, accumulator preloaded with the norms),
// here with z as the A operand so that a lane's 16-lane group writes 16 consecutive points of one row.  xs: the strip's
// scaled inputs in LDS, [DL][NT], zero padded to DL feature rows.  A wave owns every (NTHR/64)-th block of 16 rows; the
// x fragments and column norms are fetched once per strip, the row norms travel by two xor-shuffles and one index shuffle.
#ifndef SVGP_PREGEN_EXPTAB
#define SVGP_PREGEN_EXPTAB 1   // the f64 SE pre-generation takes exp from a 64-entry table of 2^(j/64) in LDS + a degree-5 polynomial (0: kexp; A/B builds)
#endif
// (kexp_tab: device_common.hpp)
template <typename T, int NT, int NTHR, int F, int DL>
__device__ __forceinline__ void pregen_mfma(const T* __restrict__ xs, const T* __restrict__ zs, int d, int64_t Mp, int64_t M,
                                            double variance_d, T* __restrict__ work, const double* __restrict__ exptab = nullptr) {
  // DL <= 16: the strip's x fragments stay in registers for the whole pass (JT x KS values).  DL = 32 / 64 (round 4: d in (16, 64]
  // used to fall off the MFMA path onto a scalar per-feature loop): the distance chain runs over the 16-feature chunks with the x
  // fragment of each MFMA read from the LDS image as it is needed (one conflict-free ds_read per MFMA: 16 lanes x consecutive
  // points of one feature row) - JT x KS fragments would be 128 VGPRs at d = 64; the z fragments of a 16-row block (KS values)
  // are fetched once per block as before.
  constexpr int KS = DL / 4, JT = NT / 16, NW = NTHR / 64;
  constexpr bool XREG = (DL <= 16);
  using M16 = Mfma16<T>;
  using acc_t = typename M16::acc_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  const T variance = T(variance_d);
  const T c1 = (F == KSE) ? T(-0.5) : T(1);
  const T c0 = (F == KSE) ? T(log(variance_d)) : T(0);
  const T ascale = (F == KSE) ? T(1) : T(-2);
  T xb[XREG ? JT : 1][XREG ? KS : 1], xn[JT];
  const T* __restrict__ xl = xs + g * NT + l15;   // this lane's element of slab 0, column tile 0
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) {
    T s = T(0);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const T v = xl[(4 * q) * NT + jt * 16];
      if constexpr (XREG) xb[jt][q] = v;
      s = fma(v, v, s);
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    xn[jt] = fma(c1, s, c0);
  }
  for (int kb = wave; kb < int(Mp / 16); kb += NW) {
    const int64_t k0 = int64_t(kb) * 16;
    T za[KS];
    T s = T(0);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int f = 4 * q + g;
      const T v = (f < d) ? zs[int64_t(f) * Mp + k0 + l15] : T(0);
      za[q] = ascale * v;
      s = fma(v, v, s);
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    s *= c1;                                   // c1 |z|^2 of row l15, in every lane
    T zn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) zn[r] = __shfl(s, M16::row(lane, r));
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      acc_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = xn[jt] + zn[r];
      if constexpr (XREG) {
#pragma unroll
        for (int q = 0; q < KS; ++q) acc = M16::mma(za[q], xb[jt][q], acc);
      } else {
        // two independent chains (even / odd feature slabs), summed at the end: halves the dependent-MFMA latency of the 8 / 16 steps
        acc_t acc2 = {0, 0, 0, 0};
        __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads of a tile with its MFMAs: hoisted across tiles they are 128 VGPRs
#pragma unroll
        for (int q = 0; q < KS; q += 2) {
          acc = M16::mma(za[q], xl[(4 * q) * NT + jt * 16], acc);
          acc2 = M16::mma(za[q + 1], xl[(4 * q + 4) * NT + jt * 16], acc2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t k = k0 + M16::row(lane, r);
        const T v = acc[r];
        T out;
        if constexpr (SVGP_PREGEN_EXPTAB && F == KSE && sizeof(T) == 8) out = T(kexp_tab(double(v > c0 ? c0 : v), exptab));
        else out = (F == KSE) ? kexp(v > c0 ? c0 : v) : kappa<T>(F, v < T(0) ? T(0) : v, variance);
        work[k * NT + jt * 16 + l15] = (k < M) ? out : T(0);
      }
    }
  }
}

// (s_memtime stamps of the diagnostic build: SVGP_SSTAMP*, diag.hpp; timing-only ablations: diag::ablate<BIT>)

// d E[log p] / d (mu, v) of one point for the value-and-gradient strips.  Not inlined: lgamma / exp / log1p and the
// Gauss-Hermite loop must not take part in the register allocation of the MFMA loops around the call (146 spilled VGPRs
// when they did).
// Value-and-gradient strips: a 128 x NT accumulator tile -> point-major rows out[(c0 + col) * Mp + row0 + 0..127] through
// LDS, so that the 128 values of a point leave as one contiguous KiB (f64) / half KiB (f32) per wave instruction instead
// of the 32- / 16-byte pieces a direct store from the MFMA layout gives (L2 merges those, but the fp32 pieces in
// particular cost: the round-1 fp32 path preferred two extra transposition passes over them).  The staging buffers are
// idle in an epilogue; tiles wider than 64 columns go in two passes.  Ends with a barrier (LDS free again).
template <typename G, typename T, int NT, int NTHR>
__device__ __forceinline__ void store_tile_point_major(const typename G::Acc& acc, T* __restrict__ smem, T* __restrict__ out,
                                                       int64_t c0, int64_t Mp, int row0) {
  constexpr int HC = NT < 64 ? NT : 64, RS = G::NB + (sizeof(T) == 8 ? 4 : 8), VEC = G::VEC, CPR = G::NB / VEC;
  using V = typename G::V;
  static_assert(size_t(HC) * RS * sizeof(T) <= ((SVGP_ASYNC && G::kAsync) ? G::ASYNC_LDS_BYTES : G::LDS_BYTES),
                "the transposition tile must fit the staging buffers");
#pragma unroll
  for (int p = 0; p < NT / HC; ++p) {
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
      const int col = G::acc_col(j);
      if (col / HC == p) {
#pragma unroll
        for (int i = 0; i < G::MI; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) smem[(col % HC) * RS + G::acc_row(i, r)] = acc.v[i][j][r];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < HC * CPR; e += NTHR) {
      const int col = e / CPR, rv = (e % CPR) * VEC;
      *reinterpret_cast<V*>(out + (c0 + p * HC + col) * Mp + row0 + rv) = *reinterpret_cast<const V*>(smem + col * RS + rv);
    }
    __syncthreads();
  }
}

// Wave priority of the value-and-gradient strips, alternated strip by strip (round 6).  The two workgroups of a CU share its SIMDs
// wave for wave, and at equal priority the issue arbiter prefers the OLDER wave: the workgroup of the first dispatch round
// (blockIdx < grid / 2) wins every contended MFMA slot, finishes its strips 10-15 % earlier than its partner (s_memtime, H: 3.47 + 3.35 M
// ticks for its two strips against 4.06 + 3.58 M), and the partner then runs alone - at the half MFMA rate one workgroup can sustain -
// to the end of the launch.  A gradient chunk is exactly two rounds of strips, so that ragged end comes with EVERY launch.  Here the
// older workgroup takes the higher priority for its first strip, the younger one for its second, and so on: both reach the end of the
// launch together.  Same box, value and gradient ms, f64: H 78.9 -> 77.0, C2 3.30 -> 3.14; alternating per panel 78.0, by time slices
// of the shared clock (2^18 / 2^20 ticks) 77.6 / 77.4; fp32 (H32, C5), whose VALU work co-executes with the partner's MFMAs,
// unchanged by any of them (profiles/round6/strip_prio_ab.log).  SVGP_STRIP_PRIO=0: no priorities (A/B builds).
#ifndef SVGP_STRIP_PRIO
#define SVGP_STRIP_PRIO 1
#endif
template <bool ON>
__device__ __forceinline__ void strip_prio(int strips_done) {
  if constexpr (ON && SVGP_STRIP_PRIO != 0) {
    if ((strips_done + (blockIdx.x >= gridDim.x / 2 ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(0);
    else __builtin_amdgcn_s_setprio(1);
  }
}

struct PointGrads { double e, gmu, gv, gs2; };
__device__ __noinline__ PointGrads strip_point_grads(LikParams lp, double mu, double v, double yv, double scale) {
  // everything by value: taking the address of the kernel argument block would move it (and with it the wave-uniform
  // operand base pointers of the LDS-DMA instructions) from SGPRs to scratch memory
  double a = 0.0, b = 0.0, gs2 = 0.0;
  expected_loglik_grad_point(lp, mu, v, yv, a, b, gs2);
  PointGrads r;
  r.e = expected_loglik_point(lp, mu, v, yv, log(lp.sigma2));
  r.gmu = a * scale;
  r.gv = b * scale;
  r.gs2 = gs2 * scale;
  return r;
}

// GRAD: the value-and-gradient form (svgp_elbo_grad).  Phase 1 as in the forward build, then - INSTEAD of phase 2 -
//   phase 3:  R A,   R = Lk^-T (Lq Lq' - I)   (ONE dense Mp x Mp GEMM on the A strip that is still in its scratch strip: the
//             adjoint's two triangular products Lq (Lq'A) - A and Lk' \ . folded into a precomputed M x M matrix)
// whose epilogue also yields the variance, v_j - k(x_j, x_j) = k_j' (R A)_j (the Kuf block keeps its own scratch strip for
// it): 3 GEMM units per point in this kernel instead of 4.  Then the likelihood gradients (g_mu, g_v) = scale dE/d(mu, v).
// Outputs: A and R A point-major (for the products contracted over points: the SYRK W = A diag(2 g_v) A' and the
// kernel-gradient reductions, which form P = Kuf_bar = alpha g_mu' + 2 (R A) diag(g_v) themselves), g_mu, g_v, and five
// per-strip sums.
// The likelihood gradients are NOT evaluated in here (round 4): a value-and-gradient strip leaves its moments in mom_mu / mom_var exactly
// like a forward strip and point_grad_kernel evaluates SVA:354-355 and their adjoint afterwards (as expect_kernel does for the forward
// path) - for the built-in likelihoods and for the host-evaluated ones alike.  Since phase 3 moved in FRONT of the likelihood gradients
// nothing in the strip consumes g_mu / g_v, and the non-inlined call was all that separated the f64 kernel (256 VGPRs + 72 spilled) from a
// spill-free shape.  The round-3 in-kernel forms (and the fp32 per-strip row sums A g_mu that went with them) left the tree in round 6:
// profiles/round6/removed_variants.patch.
// BIGD (round 4): the instantiation for 16 < d <= 64.  A separate kernel, not a branch: with the 32- / 64-feature pre-generation
// bodies inside, the register allocation of the WHOLE kernel changed (the headline f64 kernel went from 0 to 335 spilled VGPRs) -
// so the d <= 16 kernels stay bit for bit what they were and the wide-input kernels hold only the wide bodies.
// SEG (round 4): the SEGMENTED form for strips that run BESIDE the factorisation of Kuu (api.hip:
// enqueue_strips_overlapped).  Phase 1 of panel I needs nothing of the prep but block row I of T, which is final after panel
// step I of the Cholesky; so a batch of at most one round of strips is evaluated as a sequence of short launches on a second
// stream - pre-generation, then one launch per panel (each behind the event of its T row), then phase 2 + moments - instead of
// one launch behind the whole prep.  A launch covers the panels [seg_lo, seg_hi); between launches a strip's state is its scratch
// strip (per STRIP here, not per workgroup) and the fp64 column sums its threads carry, saved and restored register for register,
// so that the arithmetic - and the result, bit for bit - is that of the one-launch kernel.  No spinning, no flags: a launch that
// is waiting occupies nothing, so the factorisation's own launches always find the chip (VERDICT r3 item 2 asked for flag-gated
// persistent strips; those hold every CU while they wait - DESIGN section 3 "prep beside the strips").
template <typename T, int NT, int BK, int NTHR, int MINW = 2, int PAD = 16, bool GRAD = false, bool BIGD = false, bool SEG = false>
__global__ void __launch_bounds__(NTHR, MINW) strip_kernel(StripArgs a, int64_t nstrips) {
  using G = TileGemm<T, NT, BK, NTHR, PAD>;
  using Acc = typename G::Acc;
  using QRegs = typename G::QRegs;
  constexpr int NB = G::NB, MI = G::MI, NJ = G::NJ, VEC = G::VEC;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);                 // staging (2 buffers), reused as reduction scratch
  T* xs = smem;   // [dl][NT] inputs of the strip, scaled by 1/l: only the pre-generation pass reads them, so they alias the
                  // (then idle) staging buffers
  // The strip's whole Kuf block (SVA:216) is generated into the scratch strip BEFORE phase 1 (pregen_mfma; the epilogue of
  // panel I later overwrites rows I with A_I), so every k-step of both phases streams its Q tile.  History, measured with
  // s_memtime stamps inside one strip (tools/strip_stamps.py): generating the panel's Kuf rows inside the k-loop made a
  // generated step cost 11k cycles against 3k for a streamed one (64 of them = 27 % of a strip): z fetched from global
  // memory and waited for on the spot, a run-time loop over the features with a per-element family switch, and all of it
  // under the MFMA loop's register pressure (up to 117 spilled VGPRs).  Staging z in LDS + an unrolled body brought H from
  // 37.9 to 34.3 ms; moving the generation out of the loop altogether, onto MFMA distances, gave 34.0 ms (f64) and
  // H32 18.9 -> 17.45, C3 73.0 -> 68.4, C5 5.20 -> 4.83 ms (fp32, whose VALU work co-executes with the partner
  // workgroup's MFMAs).
  // MFMA pre-generation: xs zero padded to 8 / 16 feature rows (BIGD: 32 / 64 - SVGP_MAX_D = 64, every dimension the library takes)
  const int pre_dl = BIGD ? (a.kp.d <= 32 ? 32 : 64) : (a.kp.d <= 8 ? 8 : a.kp.d <= 16 ? 16 : 0);
  const int dl = pre_dl ? pre_dl : a.kp.d;                      // feature rows of xs

  const T* __restrict__ Tm = static_cast<const T*>(a.T);
  const T* __restrict__ U = static_cast<const T*>(a.U);
  const T* __restrict__ zs = static_cast<const T*>(a.zs);
  const T* __restrict__ mp = static_cast<const T*>(a.mp);
  const T* __restrict__ x = static_cast<const T*>(a.x);
  const T* __restrict__ invl = static_cast<const T*>(a.kp.invl);
  const int64_t Mp = a.Mp, M = a.M;
  const int d = a.kp.d, family = a.kp.family;
  const T variance = T(a.kp.variance);
  const int nP = int(Mp / NB);
  T* __restrict__ work = static_cast<T*>(a.work) + int64_t(blockIdx.x) * Mp * NT;   // SEG: re-pointed per strip below
  // GRAD: the generated Kuf block keeps a scratch strip of its own (the A strip goes beside it instead of overwriting it panel by
  // panel): phase 3's epilogue needs K again for the variance (see below).  Forward builds: one strip, K overwritten in place.
  T* __restrict__ workK = (GRAD && !diag::ablate<256>) ? static_cast<T*>(a.work) + (int64_t(gridDim.x) + blockIdx.x) * Mp * NT : work;   // not const: SEG re-points it
  const int tid = threadIdx.x, lane = tid & 63;
  const typename G::QOff qoff = G::q_offsets(NT);           // per-thread byte offsets inside a scratch-strip tile

  // Strips are handed out dynamically (one atomic per strip): workgroups that run alone on their CU near the end
  // of the launch take more strips, which removes most of the ragged last round (C2: 3.05 rounds of strips).
  // Which workgroup evaluates a strip does not change its arithmetic, so results stay bitwise reproducible.
  __shared__ unsigned next_strip;
#if SVGP_PREGEN_EXPTAB
  __shared__ double s_exptab[64];
  if (sizeof(T) == 8 && threadIdx.x < 64) s_exptab[threadIdx.x] = exp2(double(threadIdx.x) * 0.015625);
  const double* exptab = s_exptab;
#else
  const double* exptab = nullptr;
#endif
  SVGP_SSTAMP_KERNEL_BEGIN();
  [[maybe_unused]] int prio_strips = 0;
  [[maybe_unused]] int part = 0, nsplit = 1;   // split closing launch (kernels.hpp: seg_split)
  if constexpr (SEG) {
    if (a.seg_split > 1) { nsplit = a.seg_split; part = int(blockIdx.x / nstrips); }
  }
  for (int64_t strip = (nsplit > 1) ? int64_t(blockIdx.x % nstrips) : int64_t(blockIdx.x); strip < nstrips;) {
    if constexpr (SEG) {   // a strip's scratch must outlive the launch: indexed by strip, not by workgroup
      work = static_cast<T*>(a.work) + strip * Mp * NT;
      workK = GRAD ? static_cast<T*>(a.work) + (nstrips + strip) * Mp * NT : work;   // GRAD: the Kuf strips behind the A strips
    }
    SVGP_SSTAMP_STRIP_BEGIN();
    SVGP_SSTAMP(0);
    strip_prio<GRAD && !SEG>(prio_strips++);
    const int64_t c0 = strip * NT;                          // first column of the strip inside the batch
    if constexpr (SEG) { if (tid == 0) next_strip = unsigned(strip + gridDim.x); }   // static schedule: launches are at most one round
    else if (tid == 0) next_strip = gridDim.x + atomicAdd(a.counter, 1u);
    const int64_t last = a.off + a.len - 1;
    const bool do_pregen = !SEG || (a.seg_flags & kSegPregen);
    // scaled inputs of the strip -> LDS (columns past the batch end replicate the last point; masked later)
    if (do_pregen)
    for (int e = tid; e < dl * NT; e += NTHR) {
      const int f = e / NT, c = e % NT;
      int64_t g = a.off + c0 + c;
      g = g > last ? last : g;
      xs[e] = (f < d) ? x[int64_t(f) * a.ldx + g] * invl[f] : T(0);
    }
    __syncthreads();
    if (do_pregen) {
      auto pregen = [&](auto fam) {
        constexpr int F = decltype(fam)::value;
        using V = typename G::V;
        constexpr int CPR = NT / VEC;                       // vectors per k-row of the scratch strip
        for (int e = tid; e < int(Mp) * CPR; e += NTHR) {
          const int k = e / CPR, c = (e % CPR) * VEC;
          T r2[VEC];
#pragma unroll
          for (int q = 0; q < VEC; ++q) r2[q] = T(0);
#pragma unroll 8
          for (int f = 0; f < d; ++f) {
            const T zf = zs[int64_t(f) * Mp + k];
            const V xf = *reinterpret_cast<const V*>(xs + f * NT + c);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
              const T df = xf[q] - zf;
              r2[q] = fma(df, df, r2[q]);
            }
          }
          V out;
#pragma unroll
          for (int q = 0; q < VEC; ++q) out[q] = (k < M) ? kappa<T>(F, r2[q], variance) : T(0);
          *reinterpret_cast<V*>(workK + int64_t(k) * NT + c) = out;
        }
      };
      if constexpr (BIGD) {
        if (pre_dl == 32) {
          if (family == KSE) pregen_mfma<T, NT, NTHR, KSE, 32>(xs, zs, d, Mp, M, a.kp.variance, workK, exptab);
          else if (family == KM32) pregen_mfma<T, NT, NTHR, KM32, 32>(xs, zs, d, Mp, M, a.kp.variance, workK);
          else pregen_mfma<T, NT, NTHR, KM52, 32>(xs, zs, d, Mp, M, a.kp.variance, workK);
        } else {
          if (family == KSE) pregen_mfma<T, NT, NTHR, KSE, 64>(xs, zs, d, Mp, M, a.kp.variance, workK, exptab);
          else if (family == KM32) pregen_mfma<T, NT, NTHR, KM32, 64>(xs, zs, d, Mp, M, a.kp.variance, workK);
          else pregen_mfma<T, NT, NTHR, KM52, 64>(xs, zs, d, Mp, M, a.kp.variance, workK);
        }
      } else if (pre_dl == 8) {
        if (family == KSE) pregen_mfma<T, NT, NTHR, KSE, 8>(xs, zs, d, Mp, M, a.kp.variance, workK, exptab);
        else if (family == KM32) pregen_mfma<T, NT, NTHR, KM32, 8>(xs, zs, d, Mp, M, a.kp.variance, workK);
        else pregen_mfma<T, NT, NTHR, KM52, 8>(xs, zs, d, Mp, M, a.kp.variance, workK);
      } else if (pre_dl == 16) {
        if (family == KSE) pregen_mfma<T, NT, NTHR, KSE, 16>(xs, zs, d, Mp, M, a.kp.variance, workK, exptab);
        else if (family == KM32) pregen_mfma<T, NT, NTHR, KM32, 16>(xs, zs, d, Mp, M, a.kp.variance, workK);
        else pregen_mfma<T, NT, NTHR, KM52, 16>(xs, zs, d, Mp, M, a.kp.variance, workK);
      } else if (family == KSE) pregen(std::integral_constant<int, KSE>{});   // d > 16 in a kernel without the wide bodies (A/B builds)
      else if (family == KM32) pregen(std::integral_constant<int, KM32>{});
      else pregen(std::integral_constant<int, KM52>{});
      __syncthreads();
    }

    double sA[NJ], sM[NJ], sC[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) sA[j] = sM[j] = sC[j] = 0.0;
    int I_lo = 0, I_hi = nP;
    if constexpr (SEG) {
      I_lo = a.seg_lo;
      I_hi = a.seg_hi;
      if ((a.seg_flags & kSegLoad) && part == 0) {   // the column sums this thread carried out of the previous launch of the strip
        const double* __restrict__ st = a.seg_state + (strip * NTHR + tid) * (2 * NJ);
#pragma unroll
        for (int j = 0; j < NJ; ++j) { sA[j] = st[j]; sM[j] = st[NJ + j]; }
      }
    }

    // ---------------- phase 1: A = Lk \ Kuf, panel by panel ----------------
    SVGP_SSTAMP(1);
    for (int I = I_lo; I < I_hi; ++I) {
      SVGP_SSTAMP(2 + 3 * I);
      Acc acc;
      acc.zero();
      // the last NB/BK steps multiply the lower-triangular inv(L_II): their zero 16-row tiles are skipped
      const int tK = I * (NB / BK);   // first k-step of the panel's own (still Kuf) rows
      if constexpr (SVGP_ASYNC && G::kAsync) {
        auto qsrc = [&](int t) { return (t < tK ? work : workK) + int64_t(t) * BK * NT; };
        G::template loop_tri_async<(SVGP_TRI & 1) ? 1 : 0>(acc, Tm + int64_t(I) * NB, Mp, (I + 1) * (NB / BK), qsrc, smem);
      } else {
        auto qload = [&](int t, QRegs& r) { G::load_q(r, (t < tK ? work : workK) + int64_t(t) * BK * NT, qoff); };
        G::template loop_tri<(BK == 16 && (SVGP_TRI & 1)) ? 1 : 0>(acc, Tm + int64_t(I) * NB, Mp, (I + 1) * (NB / BK), qload, smem);
      }

      SVGP_SSTAMP(3 + 3 * I);
      // epilogue: A_I -> scratch strip, column sums in fp64
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = I * NB + G::acc_row(i, r);
          const double mr = double(mp[row]);
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const T val = acc.v[i][j][r];
            const int col = G::acc_col(j);
            if constexpr (!diag::ablate<4>) work[int64_t(row) * NT + col] = val;
            if (a.A_out) static_cast<T*>(a.A_out)[int64_t(row) * a.lda + c0 + col] = val;
            if constexpr (!GRAD) {
              if (a.At_out) static_cast<T*>(a.At_out)[(c0 + col) * Mp + row] = val;
            }
            if constexpr (!diag::ablate<16>) {
              const double dv = double(val);
              if constexpr (!GRAD) sA[j] = fma(dv, dv, sA[j]);
              sM[j] = fma(dv, mr, sM[j]);
            } else {
              sA[j] += double(val) + mr;
            }
          }
        }
      }
      if constexpr (GRAD && !diag::ablate<32>) store_tile_point_major<G, T, NT, NTHR>(acc, smem, static_cast<T*>(a.At_out), c0, Mp, I * NB);
      __syncthreads();  // scratch rows of panel I visible to the whole workgroup
      SVGP_SSTAMP(4 + 3 * I);
    }

    // ---------------- phase 2: C = B' A  (forward builds; the value-and-gradient build gets the variance from phase 3) --------
    if constexpr (SEG) {
      if (a.seg_flags & kSegStore) {
        double* __restrict__ st = a.seg_state + (strip * NTHR + tid) * (2 * NJ);
#pragma unroll
        for (int j = 0; j < NJ; ++j) { st[j] = sA[j]; st[NJ + j] = sM[j]; }
      }
      if (!(a.seg_flags & kSegPhase2)) {   // this launch ends here for the strip; phase 2 (GRAD: phase 3) and the moments come with a later one
        strip = next_strip;
        __syncthreads();
        continue;
      }
    }
    if constexpr (!GRAD)
    // (segmented builds: a split closing launch deals the panels round-robin - C_J costs nP - J block products - to the strip's nsplit
    // workgroups.  A checkpointed phase 2 beside the factorisation was built in round 4, measured slower and left the tree in round 6:
    // profiles/round6/removed_variants.patch)
    for (int J = SEG ? part : 0; J < nP; J += (SEG ? nsplit : 1)) {
      SVGP_SSTAMP(60 + 2 * J);
      Acc acc;
      acc.zero();
      const T* wq = work + int64_t(J) * NB * NT;
      // the first NB/BK steps multiply the upper-triangular diagonal block of B': zero tiles skipped likewise
      if constexpr (SVGP_ASYNC && G::kAsync) {
        auto qsrc = [&](int t) { return wq + int64_t(t) * BK * NT; };
        G::template loop_tri_async<(SVGP_TRI & 2) ? -1 : 0>(acc, U + int64_t(J) * NB + int64_t(J) * NB * Mp, Mp, (nP - J) * (NB / BK), qsrc,
                                                            smem);
      } else {
        auto qload = [&](int t, QRegs& r) { G::load_q(r, wq + int64_t(t) * BK * NT, qoff); };
        G::template loop_tri<(BK == 16 && (SVGP_TRI & 2)) ? -1 : 0>(acc, U + int64_t(J) * NB + int64_t(J) * NB * Mp, Mp, (nP - J) * (NB / BK), qload,
                                                   smem);
      }
      SVGP_SSTAMP(61 + 2 * J);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const T val = acc.v[i][j][r];
            if (a.C_out) static_cast<T*>(a.C_out)[int64_t(J * NB + G::acc_row(i, r)) * a.lda + c0 + G::acc_col(j)] = val;
            if (a.Ct_out) static_cast<T*>(a.Ct_out)[(c0 + G::acc_col(j)) * Mp + J * NB + G::acc_row(i, r)] = val;
            const double dv = double(val);
            sC[j] = fma(dv, dv, sC[j]);
          }
    }

    // ---------------- phase 3 (GRAD): R A, panel by panel (dense: every k-step is a full tile) ----------------
    // Round 3: the value-and-gradient build has NO phase 2.  With S = B B', R = Lk^-T (S - I) and Lk a_j = k_j (column j of Kuf):
    //   v_j - k(x_j, x_j) = sum C_.j^2 - sum A_.j^2 = a_j' (S - I) a_j = (Lk a_j)' (R a_j) = k_j' (R A)_.j
    // so the dense product this build needs anyway gives the variance too: the epilogue dots each R A tile with the Kuf tile
    // (kept in its own scratch strip) - 4 GEMM units per point (trsm 1 + dense 2 + SYRK 1) instead of 5.  The tile leaves
    // UNSCALED (g_v is only known once all panels are done): P = alpha g_mu' + 2 (R A) diag(g_v) is formed by its one consumer,
    // kgrad_kernel.  Numerics: same cancellation as the forward formula; emulated in fp32 on posteriors with v down to 4e-4 of
    // the prior variance the two formulas err alike (3e-6 absolute; 7e-15 in f64).
    if constexpr (GRAD) {
      const T* __restrict__ Rm = static_cast<const T*>(a.R);
      T* __restrict__ Pt = static_cast<T*>(a.Pt_out);
      int p3_lo = 0, p3_hi = nP;
      if constexpr (SEG) { p3_lo = (part * nP) / nsplit; p3_hi = ((part + 1) * nP) / nsplit; }
      for (int I = p3_lo; I < p3_hi; ++I) {
        SVGP_SSTAMP(26 + 4 * I);   // (diagnostic builds, nP <= 8) phase 3: loop start / loop end / after the K-dot / after the point-major store
        Acc acc;
        acc.zero();
        if constexpr (SVGP_ASYNC && G::kAsync) {
          auto qsrc = [&](int t) { return work + int64_t(t) * BK * NT; };
          G::template loop_tri_async<0>(acc, Rm + int64_t(I) * NB, Mp, nP * (NB / BK), qsrc, smem);
        } else {
          auto qload = [&](int t, QRegs& r) { G::load_q(r, work + int64_t(t) * BK * NT, qoff); };
          G::template loop_tri<0>(acc, Rm + int64_t(I) * NB, Mp, nP * (NB / BK), qload, smem);
        }
        SVGP_SSTAMP(27 + 4 * I);
        if constexpr (!diag::ablate<128>) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int64_t row = int64_t(I) * NB + G::acc_row(i, r);
#pragma unroll
              for (int j = 0; j < NJ; ++j) sC[j] = fma(double(acc.v[i][j][r]), double(workK[row * NT + G::acc_col(j)]), sC[j]);
            }
        } else {
#pragma unroll
          for (int j = 0; j < NJ; ++j) sC[j] += double(acc.v[0][j][0]);
        }
        SVGP_SSTAMP(28 + 4 * I);
        if constexpr (!diag::ablate<64>) store_tile_point_major<G, T, NT, NTHR>(acc, smem, Pt, c0, Mp, I * NB);
        else if (acc.v[0][0][0] == T(12345.678)) Pt[c0] = acc.v[0][0][0];   // keep the accumulators live
        SVGP_SSTAMP(29 + 4 * I);
      }
    }

    SVGP_SSTAMP(100);
    // ---------------- per-point moments: mu = mean + A'm (SVA:250), v = k(x,x) - ΣA² + ΣC² (SVA:251) ----------------
    double* red = reinterpret_cast<double*>(smem_raw);       // [3][WR][NT]; staging is idle here
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      double va = sA[j], vm = sM[j], vc = sC[j];
      va += __shfl_xor(va, 16); va += __shfl_xor(va, 32);
      vm += __shfl_xor(vm, 16); vm += __shfl_xor(vm, 32);
      vc += __shfl_xor(vc, 16); vc += __shfl_xor(vc, 32);
      if ((lane >> 4) == 0) {
        const int wr = (tid >> 6) / G::WC;
        const int col = G::acc_col(j);
        red[(0 * G::WR + wr) * NT + col] = va;
        red[(1 * G::WR + wr) * NT + col] = vm;
        red[(2 * G::WR + wr) * NT + col] = vc;
      }
    }
    __syncthreads();
    {
      bool split_done = false;
      if constexpr (SEG) {
        if (nsplit > 1) {   // this part's column sums -> seg_part; the last part of the strip to arrive adds them in part order
          __shared__ int s_last_part;
          if (tid < NT) {
            double qa = 0, qm = 0, qc = 0;
#pragma unroll
            for (int w = 0; w < G::WR; ++w) {
              qa += red[(0 * G::WR + w) * NT + tid];
              qm += red[(1 * G::WR + w) * NT + tid];
              qc += red[(2 * G::WR + w) * NT + tid];
            }
            double* __restrict__ pp = a.seg_part + ((int64_t(part) * nstrips + strip) * 3) * NT;
            pp[tid] = qa;
            pp[NT + tid] = qm;
            pp[2 * NT + tid] = qc;
          }
          __threadfence();   // release: the partials device-wide before the count goes up
          __syncthreads();
          if (tid == 0) s_last_part = (__hip_atomic_fetch_add(&a.seg_cnt[strip], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == unsigned(nsplit - 1));
          __syncthreads();
          if (s_last_part) {
            __threadfence();   // acquire: the other parts' partials (other CUs, possibly other XCDs)
            if (tid < NT && c0 + tid < a.len) {
              double qa = 0, qm = 0, qc = 0;   // (qa, qm: part 0 alone carries the sums of phase 1; the others add zeros)
              for (int q = 0; q < nsplit; ++q) {
                const double* __restrict__ pp = a.seg_part + ((int64_t(q) * nstrips + strip) * 3) * NT;
                qa += __builtin_nontemporal_load(pp + tid);
                qm += __builtin_nontemporal_load(pp + NT + tid);
                qc += __builtin_nontemporal_load(pp + 2 * NT + tid);
              }
              a.mom_mu[c0 + tid] = a.mean_const + qm;
              a.mom_var[c0 + tid] = a.kp.variance - qa + qc;
            }
          }
          __syncthreads();
          split_done = true;
        }
      }
      if (!split_done) {
      if (tid < NT && c0 + tid < a.len) {
        double qa = 0, qm = 0, qc = 0;
#pragma unroll
        for (int w = 0; w < G::WR; ++w) {
          qa += red[(0 * G::WR + w) * NT + tid];
          qm += red[(1 * G::WR + w) * NT + tid];
          qc += red[(2 * G::WR + w) * NT + tid];
        }
        a.mom_mu[c0 + tid] = a.mean_const + qm;
        // GRAD: qc = k_j' (R A)_.j = sum C^2 - sum A^2 (qa is not accumulated in that build, it stays 0)
        a.mom_var[c0 + tid] = a.kp.variance - qa + qc;
      }
      __syncthreads();
      }
    }
    strip = next_strip;
    __syncthreads();
    SVGP_SSTAMP(101);
  }
  SVGP_SSTAMP_KERNEL_END();
}

// ---------------------------------------------------------------------------------------------
// marginals + expected_loglikelihood (SVA:354-355): one point per thread, fixed-order block sums.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(k256) expect_kernel(LikParams lp, const double* __restrict__ mom_mu,
                                                      const double* __restrict__ mom_var, const T* __restrict__ y,
                                                      int64_t off, int64_t len, double* __restrict__ partial,
                                                      unsigned* __restrict__ negcnt, T* __restrict__ mu_out,
                                                      T* __restrict__ var_out) {
  __shared__ double sh[k256];
  __shared__ unsigned sn[k256];
  const double log_sigma2 = log(lp.sigma2);
  double e = 0.0;
  unsigned neg = 0;
  for (int64_t i = int64_t(blockIdx.x) * k256 + threadIdx.x; i < len; i += int64_t(gridDim.x) * k256) {
    const double mu = mom_mu[i];
    const double vraw = mom_var[i];
    double v = vraw + kDefaultSigma2;                     // FiniteGP(f_post, x, 1e-18) -> marginals
    bool bad = v < 0.0;
    if (bad) {
      ++neg;
      if (lp.clamp_neg_var) { v = 0.0; bad = false; }
    }
    if (mu_out) mu_out[i] = T(mu);
    if (var_out) var_out[i] = T(vraw);
    if (y && !bad) e += expected_loglik_point(lp, mu, v, double(y[off + i]), log_sigma2);
  }
  sh[threadIdx.x] = e;
  sn[threadIdx.x] = neg;
  __syncthreads();
  for (int w = k256 / 2; w > 0; w >>= 1) {
    if (int(threadIdx.x) < w) {
      sh[threadIdx.x] += sh[threadIdx.x + w];
      sn[threadIdx.x] += sn[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = sh[0];
    negcnt[blockIdx.x] = sn[0];
  }
}

// ---------------------------------------------------------------------------------------------
// value-and-gradient path (round 4): marginals + expected log-likelihood AND its adjoint d E / d (mu, v) of every point of a
// chunk, from the moments the strips left (SVA:354-355; the likelihood's own parameter gradient too).  One point per thread;
// per-block sums {E, sum g_mu, sum g_v, dE/dsigma2, n_neg} in a fixed order (LDS tree), summed over blocks by sum5_kernel.
// lp.lik == kLikExternal: the host evaluated the likelihood on svgp_marginals; lp.gh_x / lp.gh_w hold its (unscaled) point
// gradients of this chunk and E is the host's.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(k256) point_grad_kernel(LikParams lp, const double* __restrict__ mom_mu,
                                                          const double* __restrict__ mom_var, const T* __restrict__ y, int64_t off,
                                                          int64_t len, double scale_host, const double* __restrict__ n_global_dev,
                                                          double num_data, T* __restrict__ gmu_out, T* __restrict__ gv_out,
                                                          double* __restrict__ part5, unsigned* __restrict__ strip_queue, int64_t pad_to) {
  __shared__ double sh[5][k256];
  const int64_t i = int64_t(blockIdx.x) * k256 + threadIdx.x;
  // (round 6: two fills per gradient chunk - the head of the strips' queue and the padding of g_mu | g_v - ride in this launch, which
  // follows the chunk's strips on their stream: 32 launches less per H-sized evaluation)
  if (strip_queue && blockIdx.x == 0 && threadIdx.x == 0) *strip_queue = 0u;
  if (i >= len && i < pad_to) { gmu_out[i] = T(0); gv_out[i] = T(0); }
  double e5[5] = {0, 0, 0, 0, 0};
  if (i < len) {
    const double scale = n_global_dev ? (num_data > 0.0 ? num_data / *n_global_dev : 1.0) : scale_host;
    const double mu = mom_mu[i];
    double v = mom_var[i] + kDefaultSigma2;                 // FiniteGP(f_post, x, 1e-18) -> marginals
    bool bad = v < 0.0;
    if (bad) {
      e5[4] = 1.0;
      if (lp.clamp_neg_var) { v = 0.0; bad = false; }
    }
    double gm = 0.0, gvv = 0.0;
    if (!bad) {
      if (lp.lik == kLikExternal) {
        e5[1] = gm = lp.gh_x[i] * scale;
        e5[2] = gvv = lp.gh_w[i] * scale;
      } else {
        const PointGrads pg = strip_point_grads(lp, mu, v, double(y[off + i]), scale);
        e5[0] = pg.e; e5[1] = gm = pg.gmu; e5[2] = gvv = pg.gv; e5[3] = pg.gs2;
      }
    }
    gmu_out[i] = T(gm);
    gv_out[i] = T(gvv);
  }
#pragma unroll
  for (int q = 0; q < 5; ++q) sh[q][threadIdx.x] = e5[q];
  __syncthreads();
  for (int w = k256 / 2; w > 0; w >>= 1) {
    if (int(threadIdx.x) < w) {
#pragma unroll
      for (int q = 0; q < 5; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x < 5) part5[int64_t(blockIdx.x) * 5 + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ void final_reduce_kernel(const double* __restrict__ partial, const unsigned* __restrict__ negcnt, int64_t n,
                                    const int* __restrict__ chol_info, double n_points, double* __restrict__ out,
                                    const double* __restrict__ prep_scal) {
  // fixed-order tree: thread t sums elements t, t+256, ... then a fixed LDS tree -> bitwise reproducible.
  // out[0..8) is the vector a data-parallel evaluation all-reduces (comm.hip): {sum E, n_points, n_neg_var, chol flag,
  // failure flag, 0, 0, 0}
  __shared__ double sh[k256];
  __shared__ double sn[k256];
  double s = 0.0, c = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += k256) {
    s += partial[i];
    c += double(negcnt[i]);
  }
  sh[threadIdx.x] = s;
  sn[threadIdx.x] = c;
  __syncthreads();
  for (int w = k256 / 2; w > 0; w >>= 1) {
    if (int(threadIdx.x) < w) {
      sh[threadIdx.x] += sh[threadIdx.x + w];
      sn[threadIdx.x] += sn[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sh[0];
    out[1] = n_points;
    out[2] = sn[0];
    out[3] = (chol_info && *chol_info != 0) ? 1.0 : 0.0;
    out[4] = out[5] = out[6] = out[7] = 0.0;
    // behind the all-reduced 8-vector: this rank's prep scalars and chol_info, so that ONE copy brings an evaluation's results
    // to the host (round 2: three copies, ~20 us of host latency each at the small-problem floor)
    if (prep_scal) {
      for (int q = 0; q < 4; ++q) out[8 + q] = prep_scal[q];
      out[12] = chol_info ? double(*chol_info) : 0.0;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// standalone Kuf assembly (SVA:216): M x len column-major, HBM-write bound (s*(M + d) bytes per point).
// The pairwise distances of a 16-point x 16-inducing tile are one MFMA chain over the features,
//   r2 = |x|^2 + |z|^2 - 2 x.z  (accumulator preloaded with the two norms, B operand = -2 z),
// which leaves the VALU only the kernel function itself (f64 MFMA and VALU do not co-execute, and the direct
// (x - z)^2 form costs 2 d VALU instructions per element).  SE folds everything into the exponent:
//   acc = log(variance) - r2/2,  K = exp(min(acc, log variance)).
// A 256-thread workgroup owns 256 inducing rows x JB points; a wave owns 64 rows: its -2z fragments stay in
// registers for the whole block, the scaled x tile and its norms sit in LDS.  MFMA column c of block b is inducing
// row (b / VEC) 16 VEC + c VEC + b % VEC, so a lane ends up with VEC consecutive rows of one column of Kuf and a
// wave's store instruction writes 256 contiguous bytes of each of 4 columns.
// Rounding: |x|^2 + |z|^2 - 2 x.z carries an absolute error of a few ulp of (|x|^2 + |z|^2) in r2, i.e. of that size
// relative in K — parity with kernelmatrix() is tested at 1e-12 (f64) / 2e-5 (f32).
// ---------------------------------------------------------------------------------------------
// NBLK: 16-row blocks per wave (a wave owns 16 NBLK inducing rows, a workgroup 64 NBLK).  4 by default; the 64-feature build
// (32 < d <= 64, round 4: those dimensions used to take a scalar one-thread-per-row kernel at 0.16 TB/s) keeps its z fragments in
// registers too, KS = 16 values per block, and therefore owns fewer rows per wave: NBLK = VEC (2 in f64, 4 in fp32).
template <typename T, int DREG, int FAMILY, int NBLK = 4>
__global__ void __launch_bounds__(k256, 2) kuf_kernel(KernelParams kp, const T* __restrict__ zs, int64_t M, int64_t Mp,
                                                    const T* __restrict__ x, int64_t ldx, int64_t off, int64_t len,
                                                    T* __restrict__ K) {
  // Points per block: 256, one staged point per thread.  The 64-feature build (round 5) takes 128: its f64 x image
  // was 64 x 272 x 8 B = 139 KiB - ONE workgroup per CU, one wave per SIMD, every LDS read -> MFMA chain -> exp -> store serialised
  // (Hd64: 0.91 TB/s) - and is 74 KiB now: two workgroups per CU.
  constexpr int VEC = Vec16<T>::N, JB = (DREG == 64 ? 128 : 256), XLD = JB + 16, KS = DREG / 4, WROWS = 16 * NBLK;
  static_assert(NBLK % VEC == 0, "a lane stores VEC consecutive rows");
  using V = typename Vec16<T>::type;
  using acc_t = typename Mfma16<T>::acc_t;
  static_assert(JB <= k256, "at most one staged point per thread");
  __shared__ T xs[DREG * XLD];   // scaled inputs of the block, feature-major
  __shared__ T xn[JB];           // c0 + c1 |x|^2
  const int d = kp.d;
  const T* __restrict__ invl = static_cast<const T*>(kp.invl);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, kq = lane >> 4;
  // consecutive workgroups walk down the inducing rows of the same JB columns: the workgroups in flight write whole
  // columns, i.e. one contiguous region of Kuf, instead of a 1-2 KB piece out of every column
  const int nI = int((M + 4 * WROWS - 1) / (4 * WROWS));
  const int64_t j0 = int64_t(blockIdx.x / nI) * JB;
  const int64_t ibase = (int64_t(blockIdx.x % nI) * 4 + wave) * WROWS;
  const T variance = T(kp.variance);
  const T c1 = (FAMILY == KSE) ? T(-0.5) : T(1);
  const T c0 = (FAMILY == KSE) ? T(log(kp.variance)) : T(0);
  const T bscale = (FAMILY == KSE) ? T(1) : T(-2);
  if (tid < JB) {
    int64_t g = j0 + tid;
    g = g < len ? g : len - 1;
    T s = T(0);
#pragma unroll
    for (int f = 0; f < DREG; ++f) {
      const T v = (f < d) ? x[int64_t(f) * ldx + off + g] * invl[f] : T(0);
      xs[f * XLD + tid] = v;
      s = fma(v, v, s);
    }
    xn[tid] = fma(c1, s, c0);
  }
  T zb[NBLK][KS], zn[NBLK];
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    const int64_t i = ibase + (b / VEC) * (16 * VEC) + c * VEC + (b % VEC);
    // the row norm from the fragment values themselves: lane (c, kq) holds the features 4 q + kq of row c, the four kq groups meet
    // by two xor-shuffles (round 4 re-read all d features per lane for it: 64 more loads per lane and block at d = 64)
    T s = T(0);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int f = 4 * q + kq;
      const T v = (f < d && i < Mp) ? zs[int64_t(f) * Mp + i] : T(0);
      zb[b][q] = bscale * v;
      s = fma(v, v, s);
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    zn[b] = c1 * s;
  }
  __syncthreads();
  if (ibase >= M) return;
  const bool vec_ok = (M % VEC == 0);
  for (int jg = 0; jg < JB / 16; ++jg) {
    const int64_t jb = j0 + jg * 16;
    if (jb >= len) break;
    T a[KS], xr[4];
#pragma unroll
    for (int q = 0; q < KS; ++q) a[q] = xs[(4 * q + kq) * XLD + jg * 16 + c];
#pragma unroll
    for (int r = 0; r < 4; ++r) xr[r] = xn[jg * 16 + Mfma16<T>::row(lane, r)];
    acc_t acc[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[b][r] = xr[r] + zn[b];
#pragma unroll
    for (int q = 0; q < KS; ++q)
#pragma unroll
      for (int b = 0; b < NBLK; ++b) acc[b] = Mfma16<T>::mma(a[q], zb[b][q], acc[b]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t j = jb + Mfma16<T>::row(lane, r);
      if (j >= len) continue;
#pragma unroll
      for (int g = 0; g < NBLK / VEC; ++g) {
        V out;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const T v = acc[g * VEC + e][r];
          out[e] = (FAMILY == KSE) ? kexp(v > c0 ? c0 : v) : kappa<T>(FAMILY, v < T(0) ? T(0) : v, variance);
        }
        const int64_t i = ibase + g * (16 * VEC) + c * VEC;
        T* dst = K + j * M + i;
        if (vec_ok && i + VEC <= M) {
          *reinterpret_cast<V*>(dst) = out;
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (i + e < M) dst[e] = out[e];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Kuf assembly, column-owning form (the default).  Same arithmetic as kuf_kernel; what changes is WHO writes WHAT, WHEN.
// Measured with write-only kernels (tools/ubench/store_pattern.hip, 8.19 GB, same box): the 256-row x 256-point blocks of
// kuf_kernel cap at 4.6-4.85 TB/s whatever the per-instruction width, because every 8 KiB column is written as four 2 KiB
// pieces by four workgroups at four different times; a workgroup that owns WHOLE 8 KiB column pieces and finishes each
// piece before it moves to the next 16 points reaches 6.0-6.2 TB/s (4 KiB pieces: 5.5-5.7, 2 KiB: 4.8-5.1; row chunk as
// the outer loop: 5.5-5.7; nontemporal stores: slower everywhere).  So: a workgroup owns RW rows (8 KiB of a column where
// LDS allows) x 128 points; the -2z fragments and row norms of its RW rows sit in LDS (they no longer fit registers), the
// loop order is 16-point group outside, 256-row chunk inside.
// ---------------------------------------------------------------------------------------------
template <typename T, int DREG, int FAMILY, int NBLK = 4>
__global__ void __launch_bounds__(512, 2) kuf_cols_kernel(KernelParams kp, const T* __restrict__ zs, int64_t M, int64_t Mp,
                                                        const T* __restrict__ x, int64_t ldx, int64_t off, int64_t len,
                                                        T* __restrict__ K, int RW, int nR) {
  // 512 threads = 8 waves share one z image (the LDS footprint, not registers, limits residency).  Persistent: workgroup
  // w keeps row range w % nR for the whole launch (z image built once, ONE barrier per launch) and its two halves
  // (waves 0-3, 4-7) take 16-point groups 2 (w / nR) + half, + 2 gridDim / nR, ...; within a group the four waves of a
  // half cover 256 rows per chunk.  The x fragments of a group come straight from global memory (prefetched one group
  // ahead) with the norms by wave shuffles, so the main loop has no barrier: with a staged x tile and two barriers per
  // 128 points, 49 % of the wave cycles were spent waiting (SQ_WAIT_ANY) behind whichever wave the store queue held back.
  constexpr int VEC = Vec16<T>::N, KS = DREG / 4, NTH = 512, WROWS = 16 * NBLK, CH = 4 * WROWS;   // a wave owns WROWS rows of a CH-row chunk
  static_assert(NBLK % VEC == 0, "a lane stores VEC consecutive rows");
  using V = typename Vec16<T>::type;
  using acc_t = typename Mfma16<T>::acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* zl = reinterpret_cast<T*>(smem_raw);   // [DREG][RW]  bscale * scaled z of the workgroup's rows
  T* znl = zl + DREG * RW;                  // [RW]        c1 |z|^2
  // f64 SE, d > 32: exp from the 64-entry table of 2^(j/64) + a degree-5 polynomial (kexp_tab, as the strips' pre-generation: ~15 VALU
  // instructions and one LDS read against kexp's ~20).  Round 6, same box, TB/s with kexp -> with the table:
  // d = 8 5.06 -> 5.02, d = 17 4.14 -> 4.19, d = 32 3.46 -> 3.42, d = 64 2.51 -> 2.59 (profiles/round6/kuf_exptab_ab.log) - the five
  // instructions are ~4 % of a d = 64 tile, whose 16 f64 MFMAs per 256 entries (256 of ~390 issue cycles per 64 entries; f64 MFMA and
  // VALU do not co-execute) are what bounds it: 3.2 TB/s at 100 % issue
  constexpr bool kTab = SVGP_PREGEN_EXPTAB && FAMILY == KSE && sizeof(T) == 8 && DREG > 32;
  __shared__ double s_exptab[kTab ? 64 : 1];
  if (kTab && threadIdx.x < 64) s_exptab[threadIdx.x] = exp2(double(threadIdx.x) * 0.015625);
  const int d = kp.d;
  const T* __restrict__ invl = static_cast<const T*>(kp.invl);
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, half = tid >> 8, c = lane & 15, kq = lane >> 4;
  const T variance = T(kp.variance);
  const T c1 = (FAMILY == KSE) ? T(-0.5) : T(1);
  const T c0 = (FAMILY == KSE) ? T(log(kp.variance)) : T(0);
  const T bscale = (FAMILY == KSE) ? T(1) : T(-2);
  const bool vec_ok = (M % VEC == 0);
  const int nch = RW / CH;
  const int64_t r0 = int64_t(blockIdx.x % nR) * RW;
  for (int rl = tid; rl < RW; rl += NTH) {
    const int64_t i = r0 + rl;
    T s = T(0);
#pragma unroll
    for (int f = 0; f < DREG; ++f) {
      const T v = (f < d && i < Mp) ? zs[int64_t(f) * Mp + i] : T(0);
      zl[f * RW + rl] = bscale * v;
      s = fma(v, v, s);
    }
    znl[rl] = c1 * s;
  }
  T il[KS];
#pragma unroll
  for (int q = 0; q < KS; ++q) il[q] = (4 * q + kq < d) ? invl[4 * q + kq] : T(0);
  const int64_t ngroups = (len + 15) / 16;
  const int64_t gstep = 2 * int64_t(gridDim.x / nR);
  int64_t grp = 2 * int64_t(blockIdx.x / nR) + half;
  auto load_x = [&](int64_t g16, T (&dst)[KS]) {   // feature 4q + kq of point 16 g16 + c (MFMA A operand), unscaled
    int64_t g = g16 * 16 + c;
    g = g < len ? g : len - 1;
#pragma unroll
    for (int q = 0; q < KS; ++q) dst[q] = (4 * q + kq < d) ? x[int64_t(4 * q + kq) * ldx + off + g] : T(0);
  };
  T an[KS];
  if (grp < ngroups) load_x(grp, an);
  __syncthreads();
  for (; grp < ngroups; grp += gstep) {
    const int64_t jb = grp * 16;
    T a[KS], xr[4];
    T s2 = T(0);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      a[q] = an[q] * il[q];
      s2 = fma(a[q], a[q], s2);
    }
    if (grp + gstep < ngroups) load_x(grp + gstep, an);   // in flight during this group's arithmetic
    s2 += __shfl_xor(s2, 16);
    s2 += __shfl_xor(s2, 32);
    const T xnv = fma(c1, s2, c0);               // c0 + c1 |x|^2 of point c, in every lane of column c
#pragma unroll
    for (int r = 0; r < 4; ++r) xr[r] = __shfl(xnv, Mfma16<T>::row(lane, r));
    // destination of (r, g = 0, chunk 0): the chunk and g offsets are added as constants below
    T* dst0[4];
    bool live[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t j = jb + Mfma16<T>::row(lane, r);
      live[r] = j < len;
      dst0[r] = K + (live[r] ? j : jb) * M + r0 + wave * WROWS + c * VEC;
    }
    for (int rc = 0; rc < nch; ++rc) {
      const int rl0 = rc * CH + wave * WROWS;
      const int64_t ibase = r0 + rl0;
      if (ibase >= M) break;   // wave-uniform
      acc_t acc[NBLK];
      T zb[NBLK][KS];
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        const int rl = rl0 + (b / VEC) * (16 * VEC) + c * VEC + (b % VEC);
        const T zn = znl[rl];
#pragma unroll
        for (int q = 0; q < KS; ++q) zb[b][q] = zl[(4 * q + kq) * RW + rl];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[b][r] = xr[r] + zn;
      }
#pragma unroll
      for (int q = 0; q < KS; ++q)
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[b] = Mfma16<T>::mma(a[q], zb[b][q], acc[b]);
      // full tiles (16 live points, 64 rows inside M, vector-aligned columns): straight-line stores, no per-store masks or
      // branches (same box: 5.32 -> 5.45 TB/s at H); the ragged edges of the batch / of M take the guarded path
      if (vec_ok && jb + 16 <= len && ibase + WROWS <= M) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int g = 0; g < NBLK / VEC; ++g) {
            V out;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
              const T v = acc[g * VEC + e][r];
              if constexpr (kTab) out[e] = T(kexp_tab(double(v > c0 ? c0 : v), s_exptab));
              else out[e] = (FAMILY == KSE) ? kexp(v > c0 ? c0 : v) : kappa<T>(FAMILY, v < T(0) ? T(0) : v, variance);
            }
            *reinterpret_cast<V*>(dst0[r] + rc * CH + g * (16 * VEC)) = out;
          }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!live[r]) continue;
#pragma unroll
        for (int g = 0; g < NBLK / VEC; ++g) {
          V out;
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const T v = acc[g * VEC + e][r];
            if constexpr (kTab) out[e] = T(kexp_tab(double(v > c0 ? c0 : v), s_exptab));
            else out[e] = (FAMILY == KSE) ? kexp(v > c0 ? c0 : v) : kappa<T>(FAMILY, v < T(0) ? T(0) : v, variance);
          }
          const int64_t i = ibase + g * (16 * VEC) + c * VEC;
          T* dst = dst0[r] + rc * CH + g * (16 * VEC);
          if (vec_ok && i + VEC <= M) {
            *reinterpret_cast<V*>(dst) = out;
          } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
              if (i + e < M) dst[e] = out[e];
          }
        }
      }
    }
  }
}

// Kuf for 32 < d <= 64 (round 3; the standalone Kuf metric kernels keep their z fragments in registers / LDS for d <= 32):
// one thread per inducing row and 16 points, direct differences.  Off the ELBO path (svgp_kuf only).
template <typename T>
__global__ void __launch_bounds__(k256) kuf_generic_kernel(KernelParams kp, const T* __restrict__ zs, int64_t M, int64_t Mp,
                                                           const T* __restrict__ x, int64_t ldx, int64_t off, int64_t len,
                                                           T* __restrict__ K) {
  const int64_t i = int64_t(blockIdx.x) * k256 + threadIdx.x;
  const int64_t j0 = int64_t(blockIdx.y) * 16;
  if (i >= M) return;
  const T* __restrict__ invl = static_cast<const T*>(kp.invl);
  for (int jj = 0; jj < 16 && j0 + jj < len; ++jj) {
    const int64_t j = j0 + jj;
    T r2 = T(0);
    for (int f = 0; f < kp.d; ++f) {
      const T df = zs[int64_t(f) * Mp + i] - x[int64_t(f) * ldx + off + j] * invl[f];
      r2 = fma(df, df, r2);
    }
    K[i + j * M] = kappa<T>(kp.family, r2, T(kp.variance));
  }
}

template <typename T, int NT, int BK, int NTHR, int MINW = 2, int PAD = 16, bool GRAD = false, bool BIGD = false, bool SEG = false>
void launch_strip_t(hipStream_t s, const StripArgs& a, int grid, int64_t nstrips) {
  using G = TileGemm<T, NT, BK, NTHR, PAD>;
  // the strip's x image (<= 64 feature rows: SVGP_MAX_D) aliases the staging buffers
  const size_t lds = (SVGP_ASYNC && G::kAsync) ? G::ASYNC_LDS_BYTES : G::LDS_BYTES;
  static_assert(G::LDS_BYTES >= size_t(64) * NT * sizeof(T) && (!(SVGP_ASYNC && G::kAsync) || G::ASYNC_LDS_BYTES >= size_t(64) * NT * sizeof(T)),
                "x image must fit the staging buffers");
  static_assert(G::LDS_BYTES >= size_t(5) * NT * sizeof(double), "the five per-strip sums reuse the staging buffers");
  auto kern = strip_kernel<T, NT, BK, NTHR, MINW, PAD, GRAD, BIGD, SEG>;
  set_max_lds(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, s, a, nstrips);
}

// the product shapes (256 threads): d <= 16 -> the round-3 kernel, 16 < d <= 64 -> its wide-input twin (SVGP_PREGEN_MFMA_BIGD=0:
// the scalar per-feature generation inside the d <= 16 kernel, as round 3 - A/B knob)
template <typename T, int NT, bool GRAD>
void launch_strip_d(hipStream_t s, const StripArgs& a, int grid, int64_t nstrips) {
  static const bool bigd_on = exp_int("SVGP_PREGEN_MFMA_BIGD", 1) != 0;   // experiments build: A/B
  if (a.kp.d > 16 && bigd_on) launch_strip_t<T, NT, 16, 256, 2, 16, GRAD, true>(s, a, grid, nstrips);
  else launch_strip_t<T, NT, 16, 256, 2, 16, GRAD, false>(s, a, grid, nstrips);
}

}  // namespace

// Strip geometry.  Default: 64-point strips on 256-thread workgroups, two workgroups per CU, so the two
// waves on a SIMD belong to different workgroups and do not park at the same barrier.
// SVGP_STRIP_NT=128 selects the 128-point / 512-thread build, SVGP_STRIP_BK=32 the 32-deep k-step (tuning knobs).
// Measured and rejected for f64: 64 x 64 per wave on one workgroup per CU (1 wave/SIMD): 54.9 ms vs 42.1 ms at H.
// (every knob below is a constant - its default - in the product build: knobs.hpp)
static int env_int(const char* name, int dflt) { return exp_int(name, dflt); }

int strip_nt(int dtype, int64_t Mp, int64_t /*len*/) {
  // f64: 64-point strips; f32: 128-point strips (a wave then owns 64 x 64 = 16 MFMA tiles, the same 64 accumulator
  // VGPRs as the f64 wave, and twice the MFMA work per barrier and per byte of T/U).  SVGP_STRIP_NT overrides.
  static const int forced = env_int("SVGP_STRIP_NT", 0);
  if (forced == 64 || forced == 128) return forced;
  // measured (same box): H32 22.4 -> 20.5 ms, C3 84.1 -> 77.8 ms, C5 5.9 -> 5.4 ms; C4 (Mp = 8192) 131.7 -> 136.0 ms
  return dtype == 0 ? 64 : env_int("SVGP_F32_NT", Mp <= 2048 ? 128 : 64);
}

size_t strip_work_bytes(int dtype, int64_t Mp, int nt, int grid) {
  return size_t(grid) * size_t(Mp) * size_t(nt) * (dtype == 0 ? 8 : 4);
}

int strip_grid(int dtype, int nt, int64_t nstrips, int num_cus) {
  // two workgroups per CU (measured: a third f32 workgroup fits but is 7 % slower; SVGP_WG_PER_CU overrides)
  const int per_cu = env_int("SVGP_WG_PER_CU", (nt <= 64 || dtype == 1) ? 2 : 1);
  int64_t cap = int64_t(num_cus) * per_cu;
  // A/B knob (round 4): fewer workgroups than slots = a scratch working set below the 256 MiB Infinity Cache
  // (H: 512 workgroups x 512 KiB = exactly 256 MiB); profiles/round4/strip_grid_ab.log
  static const int forced_grid = env_int("SVGP_STRIP_GRID", 0);
  if (forced_grid > 0 && forced_grid < cap) cap = forced_grid;
  return int(nstrips < cap ? nstrips : cap);
}

// Strip schedule of one batch.  A batch with fewer regular-width strips than half the workgroup slots (minibatches:
// 4096 points are 64 strips for 512 slots) leaves most of the chip idle, so it runs as half-width strips instead (the
// narrower build: half the MFMA work per strip, twice the T/U bytes per flop) — measured at M = 1024 on 4096 / 16384
// points: 0.86 -> 0.55 ms / 0.93 -> 0.76 ms (f64), 0.78 -> 0.47 / 0.80 -> 0.51 ms (f32).  Per-point arithmetic does not
// depend on the strip width, so the ELBO is bitwise the same (tests/test_gpu_parity.py).
// Measured and rejected: running the last partial round of a large batch as a second, half-width launch (N = 1e5:
// 4.59 vs 4.62 ms, N = 2e5: 8.05 vs 8.31 ms) — the single launch's dynamic queue already hands the last strips to the
// workgroups that finish first, and those then run alone on their CU at ~0.65 of the paired strip time.
// Large batches whose strip count is not a multiple of the workgroup slots (C2, C4: 1563 strips on 512 slots = 3.05
// rounds) used to pay a whole strip time for the last 27 strips.  Now the whole rounds run as one launch and the
// remainder as half-width strips in a SECOND launch on a second stream, concurrently: its workgroups become resident as
// the main launch's finish (or before them - either way the short jobs fill the ragged end).  A sequential second launch
// did nothing (kernel boundaries are barriers).  Same-box A/B (tools/cfg_ab.sh): C4 127.8 -> 123.3 ms, C2 1.16 -> 1.135 ms
// (27 left-over strips = 5 % of a round); H32 16.87 -> 16.98 ms (133 left-over strips = 26 % of a round: the dynamic queue
// of one launch already spreads those), hence the 15 % threshold.
StripPlan strip_plan(int dtype, int64_t Mp, int64_t len, int num_cus) {
  StripPlan p{};
  const int W = strip_nt(dtype, Mp, len);
  const int W2 = W / 2 >= 32 ? W / 2 : 0;   // f64: 64 -> 32; f32: 128 -> 64, 64 -> 32
  const int64_t S = (len + W - 1) / W;
  p.nt = W; p.nstrips = S; p.points = len; p.grid = strip_grid(dtype, W, S, num_cus);
  static const int enabled = env_int("SVGP_TAIL", 1);
  if (!W2 || !enabled) return p;
  const int64_t S2 = (len + W2 - 1) / W2;
  const int64_t G = strip_grid(dtype, W, INT64_MAX, num_cus);
  if (S2 <= strip_grid(dtype, W2, INT64_MAX, num_cus)) {   // small batch: everything as half-width strips
    p.nstrips = 0; p.points = 0; p.grid = 0;
    p.nt_tail = W2; p.nstrips_tail = S2; p.grid_tail = strip_grid(dtype, W2, S2, num_cus);
    return p;
  }
  static const int ctail = env_int("SVGP_CTAIL", 1);
  const int64_t rem = S % G;
  if (ctail && S > G && rem > 0 && rem * 100 < G * 15) {   // a last round less than 15 % full
    const int64_t whole = S - rem;
    p.nstrips = whole; p.points = whole * W; p.grid = int(G);
    const int64_t left = len - p.points;
    p.nt_tail = W2; p.nstrips_tail = (left + W2 - 1) / W2; p.grid_tail = strip_grid(dtype, W2, p.nstrips_tail, num_cus);
    p.concurrent_tail = true;
  }
  return p;
}

StripPlan strip_plan_single(int dtype, int64_t Mp, int64_t len, int num_cus) {
  StripPlan p = strip_plan(dtype, Mp, len, num_cus);
  if (!p.concurrent_tail) return p;
  StripPlan q{};
  q.nt = p.nt;
  q.nstrips = (len + p.nt - 1) / p.nt;
  q.points = len;
  q.grid = strip_grid(dtype, p.nt, q.nstrips, num_cus);
  return q;
}

void launch_strip(int dtype, hipStream_t s, const StripArgs& a, int nt, int grid, int64_t nstrips) {
  static const bool bk32 = env_int("SVGP_STRIP_BK", 16) == 32;
  if (nt == 32 && dtype == 0) {
    launch_strip_d<double, 32, false>(s, a, grid, nstrips);
  } else if (nt == 32) {
    launch_strip_d<float, 32, false>(s, a, grid, nstrips);
  } else if (nt == 64) {
    if (dtype == 0) launch_strip_d<double, 64, false>(s, a, grid, nstrips);
    else launch_strip_d<float, 64, false>(s, a, grid, nstrips);   // BK = 32 measured identical
  } else if (dtype == 1 && env_int("SVGP_F32_THREADS", 256) == 256) {
    launch_strip_d<float, 128, false>(s, a, grid, nstrips);
  }
#ifdef SVGP_EXPERIMENTS   // the 512-thread, 128-point builds (SVGP_STRIP_NT=128 / SVGP_F32_THREADS=512): measured and rejected, round 1-3
  else if (dtype == 0) {
    if (bk32 && a.kp.d <= 8) launch_strip_t<double, 128, 32, 512>(s, a, grid, nstrips);
    else launch_strip_t<double, 128, 16, 512>(s, a, grid, nstrips);
  } else {
    if (bk32 && a.kp.d <= 8) launch_strip_t<float, 128, 32, 512>(s, a, grid, nstrips);
    else launch_strip_t<float, 128, 16, 512>(s, a, grid, nstrips);
  }
#else
  else {
    (void)bk32;
    leave_note("internal: no strip kernel for this (dtype, width)");   // unreachable: strip_nt() only plans the widths above
  }
#endif
}

// the segmented strips (a.seg_*): nt = 32 / 64 (f64), 32 / 64 / 128 (f32); d <= 16 (wider inputs take the one-launch path).
// grad: the value-and-gradient build (phase 1 per panel with the point-major A, then phase 3 + moments in the closing launch;
// `work` holds 2 x nstrips scratch strips: A, then Kuf)
void launch_strip_seg(int dtype, hipStream_t s, const StripArgs& a, int nt, int grid, int64_t nstrips, bool grad) {
  if (grad) {
    if (dtype == 0) {
      if (nt == 32) launch_strip_t<double, 32, 16, 256, 2, 16, true, false, true>(s, a, grid, nstrips);
      else launch_strip_t<double, 64, 16, 256, 2, 16, true, false, true>(s, a, grid, nstrips);
    } else {
      if (nt == 32) launch_strip_t<float, 32, 16, 256, 2, 16, true, false, true>(s, a, grid, nstrips);
      else if (nt == 64) launch_strip_t<float, 64, 16, 256, 2, 16, true, false, true>(s, a, grid, nstrips);
      else launch_strip_t<float, 128, 16, 256, 2, 16, true, false, true>(s, a, grid, nstrips);
    }
    return;
  }
  if (dtype == 0) {
    if (nt == 32) launch_strip_t<double, 32, 16, 256, 2, 16, false, false, true>(s, a, grid, nstrips);
    else launch_strip_t<double, 64, 16, 256, 2, 16, false, false, true>(s, a, grid, nstrips);
  } else {
    if (nt == 32) launch_strip_t<float, 32, 16, 256, 2, 16, false, false, true>(s, a, grid, nstrips);
    else if (nt == 64) launch_strip_t<float, 64, 16, 256, 2, 16, false, false, true>(s, a, grid, nstrips);
    else launch_strip_t<float, 128, 16, 256, 2, 16, false, false, true>(s, a, grid, nstrips);
  }
}
// doubles of state per strip a segmented launch saves / restores (kernels.hpp: StripArgs::seg_state)
size_t strip_seg_state_doubles(int dtype, int nt) {
  const int nj = nt / 2 / 16 > 0 ? nt / 2 / 16 : 1;   // TileGemm::NJ of the 256-thread builds (WC = 2)
  (void)dtype;
  return size_t(256) * 2 * nj;
}

void launch_strip_grad(int dtype, hipStream_t s, const StripArgs& a, int nt, int grid, int64_t nstrips) {
  // the strips leave (mu, v); launch_point_grads follows (both likelihood routes, one instantiation per shape)
  if (dtype == 0) {
    if (nt == 32) launch_strip_d<double, 32, true>(s, a, grid, nstrips);
    else launch_strip_d<double, 64, true>(s, a, grid, nstrips);
  } else {
    if (nt == 32) launch_strip_d<float, 32, true>(s, a, grid, nstrips);
    else if (nt == 64) launch_strip_d<float, 64, true>(s, a, grid, nstrips);
    else launch_strip_d<float, 128, true>(s, a, grid, nstrips);
  }
}

int point_grad_blocks(int64_t len) { return int((len + k256 - 1) / k256); }

void launch_point_grads(int dtype, hipStream_t s, const LikParams& lp, const double* mom_mu, const double* mom_var, const void* y,
                        int64_t off, int64_t len, double scale, const double* n_global_dev, double num_data, void* gmu_out,
                        void* gv_out, double* part5, unsigned* strip_queue, int64_t pad_to) {
  // strip_queue (nullable): zeroed for the NEXT launch of the strips; pad_to: g_mu, g_v of the points [len, pad_to) are written as zeros
  // (pad_to <= the 256-point blocks of the launch: the weighted SYRK reads g_v over the chunk padded to 128 points)
  const int nb = point_grad_blocks(len);
  if (dtype == 0)
    hipLaunchKernelGGL(point_grad_kernel<double>, dim3(nb), dim3(k256), 0, s, lp, mom_mu, mom_var, (const double*)y, off, len, scale,
                       n_global_dev, num_data, (double*)gmu_out, (double*)gv_out, part5, strip_queue, pad_to);
  else
    hipLaunchKernelGGL(point_grad_kernel<float>, dim3(nb), dim3(k256), 0, s, lp, mom_mu, mom_var, (const float*)y, off, len, scale,
                       n_global_dev, num_data, (float*)gmu_out, (float*)gv_out, part5, strip_queue, pad_to);
}

int expect_blocks(int64_t len) {
  const int64_t b = (len + k256 - 1) / k256;
  return int(b < 1024 ? b : 1024);
}

void launch_expect(int dtype, hipStream_t s, const LikParams& lp, const double* mom_mu, const double* mom_var,
                   const void* y, int64_t off, int64_t len, double* partial, unsigned* negcnt, void* mu_out,
                   void* var_out) {
  const int nb = expect_blocks(len);
  if (dtype == 0)
    hipLaunchKernelGGL(expect_kernel<double>, dim3(nb), dim3(k256), 0, s, lp, mom_mu, mom_var, (const double*)y, off, len,
                       partial, negcnt, (double*)mu_out, (double*)var_out);
  else
    hipLaunchKernelGGL(expect_kernel<float>, dim3(nb), dim3(k256), 0, s, lp, mom_mu, mom_var, (const float*)y, off, len,
                       partial, negcnt, (float*)mu_out, (float*)var_out);
}

void launch_final_reduce(hipStream_t s, const double* partial, const unsigned* negcnt, int64_t n, const int* chol_info,
                         double n_points, double* out, const double* prep_scal) {
  hipLaunchKernelGGL(final_reduce_kernel, dim3(1), dim3(k256), 0, s, partial, negcnt, n, chol_info, n_points, out, prep_scal);
}

// The column-owning Kuf kernel's launch for one padded feature count.
template <typename T, int DREG, int FAMILY, int NBLK = 4>
static void launch_kuf_cols(hipStream_t s, const KernelParams& kp, const T* zs, int64_t M, int64_t Mp, const T* x, int64_t ldx,
                            int64_t off, int64_t len, T* Kuf) {
  // rows per workgroup: 8 KiB of a column where the z image ((DREG + 1) values per row) fits 72 KiB of LDS (two workgroups
  // per CU), never less than 256 rows, never more than the (256-padded) matrix
  static const int forced = env_int("SVGP_KUF_RW", 0);
  // 36 KiB of z image per workgroup (four 512-thread workgroups per CU by LDS): 4 KiB column pieces for d <= 8.  The
  // arithmetic needs the resident waves more than the stores need 8 KiB pieces (same box, H f64: RW 1024 4.88, RW 512
  // 5.26, RW 256 4.95 TB/s; C4 f32: RW 2048 4.5, RW 1024 4.9 TB/s)
  // (d > 8: 72 KiB, or the pieces shrink to 2 KiB: C3 4.0-4.15 vs 4.5-4.65 TB/s)
  int64_t rw = int64_t((DREG > 8 ? 73728 : 36864) / ((DREG + 1) * sizeof(T))) / 256 * 256;
  const int64_t cap = int64_t(8192 / sizeof(T)), mrows = (M + 255) / 256 * 256;
  rw = rw > cap ? cap : rw;
  if (forced > 0) rw = int64_t(forced) / 256 * 256;
  rw = rw < 256 ? 256 : rw;
  rw = rw > mrows ? mrows : rw;
  const int nR = int((M + rw - 1) / rw);
  const size_t lds = size_t(DREG + 1) * size_t(rw) * sizeof(T);
  auto kern = kuf_cols_kernel<T, DREG, FAMILY, NBLK>;
  set_max_lds(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  const int64_t npairs = (len + 31) / 32;   // a workgroup's two halves take one 16-point group each per step
  int per_cu = 0, dev = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, lds) != hipSuccess || per_cu < 1) per_cu = 1;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  static const int forced_wg = env_int("SVGP_KUF_WG_PER_CU", 0);
  if (forced_wg > 0 && forced_wg < per_cu) per_cu = forced_wg;
  int64_t slots = int64_t(cus) * per_cu / nR * nR;
  slots = slots < nR ? nR : slots;
  const dim3 grid((unsigned)(npairs * nR < slots ? npairs * nR : slots));
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, kp, zs, M, Mp, x, ldx, off, len, Kuf, int(rw), nR);
}

template <typename T, int FAMILY>
static void launch_kuf_f(hipStream_t s, const KernelParams& kp, const T* zs, int64_t M, int64_t Mp, const T* x, int64_t ldx,
                         int64_t off, int64_t len, T* Kuf) {
  // Which kernel (same-box A/B runs, tools/kuf_ab.sh; TB/s, column-owning vs 256 x 256 blocks):
  //   H   f64 d 8  SE   M 1024 (8 KiB columns)   4.98-4.99 vs 4.04-4.16      C2 f64 d 8 SE M 512   4.28 vs 3.9
  //   C3  f32 d 16 M52  M 2048 (8 KiB columns)   4.70      vs 4.17-4.18      H32 f32 d 8 SE M 1024 4.97 vs 5.28
  //   C4  f32 d 8  SE   M 8192 (32 KiB columns)  4.5-4.7   vs 5.15-5.2
  // i.e. the column-owning kernel wins where the arithmetic per element is heavy enough to need its barrier-free,
  // persistent structure (f64; f32 with d > 8) and the column fits one 8 KiB piece; the light f32 d <= 8 case is purely
  // store-bound in both and the block kernel's higher residency wins by a few per cent.
  static const int forced_v = env_int("SVGP_KUF_V1", -1);   // A/B knob: 1 = block kernel, 0 = column-owning kernel
  // d > 32 (round 5): the column-owning kernel too wherever a column is one 8 KiB piece - its z image then takes 98 / 130 KiB of LDS
  // in f64 (48 / 64 feature rows x 256 inducing rows: ONE 512-thread workgroup per CU, which its 200+ VGPRs allow anyway) and is built
  // once per launch, where the block kernel re-reads 64 KiB of z and 64 KiB of x per 128 x 128 block of Kuf
  static const int cols_wide = env_int("SVGP_KUF_COLS_WIDE", 1);   // experiments build, A/B: 0 = the block kernel for every d > 32
  const bool one_piece = size_t(M) * sizeof(T) <= 8192;
  // (same box, profiles/round5/kuf_wide_cols.log: Hd64 f64 1.35 -> 2.55 TB/s; fp32 d = 64 2.09 -> 1.98: fp32 keeps the block kernel)
  const bool v1 = kp.d > 32 ? !(sizeof(T) == 8 && one_piece && cols_wide)
                            : (forced_v >= 0 ? forced_v != 0 : (!one_piece || (sizeof(T) == 4 && kp.d <= 8)));
  if (v1) {
    const dim3 grid((unsigned)(((len + 255) / 256) * ((M + 255) / 256)));
#define SVGP_KUF_LAUNCH(DREG) \
  hipLaunchKernelGGL((kuf_kernel<T, DREG, FAMILY>), grid, dim3(k256), 0, s, kp, zs, M, Mp, x, ldx, off, len, Kuf)
    if (kp.d > 32) {   // 64 feature rows: VEC blocks of 16 rows per wave, 128 points per block
      constexpr int NB64 = Vec16<T>::N, WGROWS = 64 * NB64;
      const dim3 g64((unsigned)(((len + 127) / 128) * ((M + WGROWS - 1) / WGROWS)));
      hipLaunchKernelGGL((kuf_kernel<T, 64, FAMILY, NB64>), g64, dim3(k256), 0, s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
      return;
    }
    if (kp.d <= 4) SVGP_KUF_LAUNCH(4);
    else if (kp.d <= 8) SVGP_KUF_LAUNCH(8);
    else if (kp.d <= 16) SVGP_KUF_LAUNCH(16);
    else SVGP_KUF_LAUNCH(32);
#undef SVGP_KUF_LAUNCH
    return;
  }
  // feature rows padded to 4 / 8 / 16 / 20 / 24 / 32 / 48 / 64 (round 5: 20 and 24 - at 16 < d <= 32 the kernel is bound by the f64 distance chain,
  // KS = DREG / 4 MFMAs per 16 x 16 tile, plus the kernel function on the VALU, not by the stores: d = 17 ran eight MFMAs where five do)
  if (kp.d <= 4) launch_kuf_cols<T, 4, FAMILY>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
  else if (kp.d <= 8) launch_kuf_cols<T, 8, FAMILY>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
  else if (kp.d <= 16) launch_kuf_cols<T, 16, FAMILY>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
  else if (kp.d <= 32) {
#ifdef SVGP_EXPERIMENTS
    // A/B: two 16-row blocks per wave (128-row chunks) - 121-ish VGPRs instead of 139-169, i.e. two workgroups per CU
    static const int nblk2 = exp_int("SVGP_KUF_NBLK2", 0);
    if constexpr (sizeof(T) == 8) {
      if (nblk2) {
        if (kp.d <= 20) launch_kuf_cols<T, 20, FAMILY, 2>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
        else if (kp.d <= 24) launch_kuf_cols<T, 24, FAMILY, 2>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
        else launch_kuf_cols<T, 32, FAMILY, 2>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
        return;
      }
    }
#endif
    if (kp.d <= 20) launch_kuf_cols<T, 20, FAMILY>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
    else if (kp.d <= 24) launch_kuf_cols<T, 24, FAMILY>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
    else launch_kuf_cols<T, 32, FAMILY>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
  }
  else if constexpr (sizeof(T) == 8) {
    if (kp.d <= 48) launch_kuf_cols<T, 48, FAMILY>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
    else launch_kuf_cols<T, 64, FAMILY>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
  }
}

template <typename T>
static void launch_kuf_t(hipStream_t s, const KernelParams& kp, const T* zs, int64_t M, int64_t Mp, const T* x, int64_t ldx,
                         int64_t off, int64_t len, T* Kuf) {
#ifdef SVGP_EXPERIMENTS
  static const bool generic_knob = exp_int("SVGP_KUF_GENERIC", 0) == 1;   // A/B: the round-3 path for d > 32
  if (kp.d > 32 && generic_knob) {
    hipLaunchKernelGGL(kuf_generic_kernel<T>, dim3((unsigned)((M + 255) / 256), (unsigned)((len + 15) / 16)), dim3(k256), 0, s, kp, zs, M, Mp, x, ldx,
                       off, len, Kuf);
    return;
  }
#endif
  if (kp.family == KSE) launch_kuf_f<T, KSE>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
  else if (kp.family == KM32) launch_kuf_f<T, KM32>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
  else launch_kuf_f<T, KM52>(s, kp, zs, M, Mp, x, ldx, off, len, Kuf);
}

void launch_kuf(int dtype, hipStream_t s, const KernelParams& kp, const void* zs, int64_t M, int64_t Mp, const void* x,
                int64_t ldx, int64_t off, int64_t len, void* Kuf) {
  if (dtype == 0)
    launch_kuf_t<double>(s, kp, static_cast<const double*>(zs), M, Mp, static_cast<const double*>(x), ldx, off, len,
                         static_cast<double*>(Kuf));
  else
    launch_kuf_t<float>(s, kp, static_cast<const float*>(zs), M, Mp, static_cast<const float*>(x), ldx, off, len,
                        static_cast<float*>(Kuf));
}

}  // namespace svgp
