// prep.hip — the M-sized, data-independent part of posterior(sva) (reference
// src/SparseVariationalApproximationModule.jl:160-187 and src/utils.jl:15-18) and of _prior_kl (:364-373):
//   Kuu = k(z, z) + jitter I            (src/utils.jl:17 ∘ cov(::FiniteGP))
//   Lk  = cholesky(Kuu).L               blocked right-looking: LDS-resident 128x128 POTF2 + MFMA TRSM/SYRK tiles
//   T   = blkdiag(inv(L_II)) * [-L_strict | I]   so that the data-sized trsm (SVA:217) becomes pure GEMM panels
//   U   = Lq'                           (SVA:183-184, B = Lq for NonCentered)
//   KL scalars                          (SVA:364-373)
// Everything is padded to Mp = ceil(M/128)*128 with an identity block so no kernel needs edge tiles.
#define SVGP_DIAG_TU_PREP
#include "device_common.hpp"
#include "kernels.hpp"
#include "knobs.hpp"

#include <hip/hip_ext.h>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace svgp {
namespace {

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void scale_inputs_kernel(const T* __restrict__ z, int layout, int d, int64_t M, int64_t Mp,
                                    const T* __restrict__ invl, T* __restrict__ zs) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int f = blockIdx.y;
  if (i >= Mp) return;
  T v = T(0);
  if (i < M) v = (layout == 1 ? z[int64_t(f) * M + i] : z[i * d + f]) * invl[f];  // RowVecs : ColVecs/Vec
  zs[int64_t(f) * Mp + i] = v;
}

template <typename T>
__global__ void transpose_colvecs_kernel(const T* __restrict__ x, int d, int64_t n, int64_t ldx, T* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int f = 0; f < d; ++f) out[int64_t(f) * ldx + i] = x[i * d + f];
}

// Kuu = k(z, z) + jitter I on the M x M block, identity on the padding.  Only the 128-tiles on and below the diagonal are
// written: the factorisation (and every later reader of L) never touches the tiles above it.  A thread owns one row i and JB
// consecutive columns: its scaled z_i stays in registers (d <= 8) and z_j is wave-uniform, so an element costs one pass over the
// features instead of 2 d loads (round 2: one element per thread, 287 us at M = 8192 fp32 = 0.9 TB/s of stores).
template <typename T, int DREG>
__global__ void __launch_bounds__(k256) kuu_kernel(KernelParams kp, const T* __restrict__ zs, int64_t M, int64_t Mp, T jitter,
                                                   T* __restrict__ K) {
  constexpr int JB = 16;
  const int64_t i = int64_t(blockIdx.x) * k256 + threadIdx.x;
  const int64_t j0 = int64_t(blockIdx.y) * JB;
  if (i >= Mp || j0 / kNB > (int64_t(blockIdx.x) * k256 + k256 - 1) / kNB) return;   // the whole block lies above the diagonal tiles
  T zi[DREG > 0 ? DREG : 1];
  if constexpr (DREG > 0) {
#pragma unroll
    for (int f = 0; f < DREG; ++f) zi[f] = (f < kp.d) ? zs[int64_t(f) * Mp + i] : T(0);
  }
  for (int jj = 0; jj < JB; ++jj) {
    const int64_t j = j0 + jj;
    if (j / kNB > i / kNB) continue;   // tile above the diagonal
    T v;
    if (i < M && j < M) {
      T r2 = T(0);
      if constexpr (DREG > 0) {
#pragma unroll
        for (int f = 0; f < DREG; ++f) {
          const T df = zi[f] - ((f < kp.d) ? zs[int64_t(f) * Mp + j] : T(0));
          r2 = fma(df, df, r2);
        }
      } else {
        for (int f = 0; f < kp.d; ++f) {
          const T df = zs[int64_t(f) * Mp + i] - zs[int64_t(f) * Mp + j];
          r2 = fma(df, df, r2);
        }
      }
      v = kappa<T>(kp.family, r2, T(kp.variance));
      if (i == j) v += jitter;
    } else {
      v = (i == j) ? T(1) : T(0);
    }
    K[i + j * Mp] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// POTF2 + TRTRI of one 128x128 diagonal block, LDS resident, MFMA blocked (16-wide panels).
//   per panel p:  wave 0 factors the 16x16 diagonal block D in registers (lane i owns row i; pivots and
//                 columns travel by v_readlane) and forms inv(D) row-parallel from X D = I;
//                 all waves: L[t,p] = A[t,p] inv(D)'  and  A[ti,tj] -= L[ti,p] L[tj,p]'  as 16x16x4 MFMAs.
//   inverse:      X = inv(L) by 16-blocks, one wave per block column (no barriers):
//                 X[ti,tj] = -inv(D_ti) * sum_{tj<=s<ti} L[ti,s] X[s,tj]; the accumulator of the sum is fed
//                 straight back as the B operand of the second product (register r of a 16x16 result is
//                 k-slab r of a B fragment when A is read with k = Mfma16::row(lane, r)).
// L overwrites the lower triangle of the block in A; inv(L) goes to the same block of Tm (upper = 0).
// info: LAPACK semantics, order of the first non-positive pivot (kept if an earlier panel already failed).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_t(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_t(float v, int srclane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}

// 1/sqrt(d) from the hardware estimate (v_rsq: ~24 bits) and Newton steps; with the residual corrections at the use
// sites the pivot and the scaled column are within 1 ulp of sqrt / divide, at a third of their dependent latency
// (the libm sqrt + divide chain was the longest part of every one of the 128 serial column steps of a block).
__device__ __forceinline__ double krsqrt(double d) {
  double r = __builtin_amdgcn_rsq(d);
  r = fma(0.5 * r, fma(-d * r, r, 1.0), r);
  r = fma(0.5 * r, fma(-d * r, r, 1.0), r);
  return r;
}
__device__ __forceinline__ float krsqrt(float d) {
  float r = __builtin_amdgcn_rsqf(d);
  r = fmaf(0.5f * r, fmaf(-d * r, r, 1.0f), r);
  return r;
}

// (s_memtime stamps of the diagnostic build: SVGP_STAMP / SVGP_STAMPW, diag.hpp)

// wave 0's part of a block step: factor the 16x16 diagonal block at offset o (already updated) and invert it.
// 16x16 Cholesky AND its inverse in registers, left-looking: lane l15 (every 16-lane row of the wave redundantly) owns row
// l15 of L and column l15 of X = inv(L).  Step j needs row j of L in every lane:
//   L[i][j] = (A[i][j] - sum_k L[i][k] L[j][k]) / L[j][j]      X[j][c] = (delta_jc - sum_k L[j][k] X[k][c]) / L[j][j]
// Round 3: the broadcast is a DPP row_newbcast (lane j of each 16-lane row to the whole row: v_mov_b32_dpp, pure VALU).  Rounds
// 1-2 used v_readlane into an SGPR: s_memtime showed ~312-340 cycles per column step whatever the instruction count (43 -> 27
// instructions changed nothing) - the VALU -> SGPR -> VALU round trips on the dependent chain; with DPP a step is 216 cycles
// (fp32 factor16: 5000 -> 3460 cycles; the 128 steps of a block are the critical path of the whole factorisation; prep at
// M = 1024, same box: 0.585 -> 0.523 ms f64).  fp32 takes the hardware
// v_rsq_f32 (1 ulp) as it is; fp64 keeps the Newton refinements (v_rsq_f64 delivers ~26 bits).  No per-step selects: an X
// lane starts from delta_jc and stays exactly zero above the diagonal; the diagonal entry is t r = sqrt(t) like any other
// entry of its column; L rows hold unused garbage right of the diagonal.
#ifndef SVGP_F16_DPP
#define SVGP_F16_DPP 1   // 0: v_readlane broadcasts (A/B and bisection builds)
#endif
template <int J>
__device__ __forceinline__ float row_bcast(float v) {
#if SVGP_F16_DPP
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + J, 0xf, 0xf, true));
#else
  return readlane_t(v, J);
#endif
}
template <int J>
__device__ __forceinline__ double row_bcast(double v) {
#if SVGP_F16_DPP
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + J, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + J, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
#else
  return readlane_t(v, J);
#endif
}
// acc -= bcast_J(v) * w in ONE instruction: the DP ALU takes a row_newbcast DPP operand directly (v_fmac_f64_dpp, gfx90a+; the
// 32-bit form since gfx9), so a column of the factor costs two instructions per earlier column instead of four (f64: two
// v_mov_b32_dpp at 8 cycles each + two FMAs - the broadcasts were 55 % of a column step's issue time).  Inline assembly: the
// compiler selects the three-operand v_fma_f64, which has no DPP form.  NOP: the hazard recogniser does not look inside inline
// assembly - a DPP read of a VGPR the previous instructions wrote needs 2 wait states (the k = J - 1 term only).
template <int J, bool NOP>
__device__ __forceinline__ void fmac_bcast(double& acc, double v, double w) {
#if SVGP_F16_DPP == 2   // A/B: one v_mov_b64_dpp + a plain FMA
  const double zero = 0.0;
  const double b = __builtin_amdgcn_update_dpp(zero, v, 0x150 + J, 0xf, 0xf, true);
  acc = fma(-b, w, acc);
#elif SVGP_F16_DPP
  if constexpr (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(w), "n"(J));
  else asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(w), "n"(J));
#else
  acc = fma(-row_bcast<J>(v), w, acc);
#endif
}
template <int J, bool NOP>
__device__ __forceinline__ void fmac_bcast(float& acc, float v, float w) {
#if SVGP_F16_DPP
  if constexpr (NOP) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(w), "n"(J));
  else asm volatile("v_fmac_f32_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(w), "n"(J));
#else
  acc = fmaf(-row_bcast<J>(v), w, acc);
#endif
}
__device__ __forceinline__ double pivot_scale(double t, double d, double rj) {
  double dj = d * rj;
  dj = fma(fma(-dj, dj, d), 0.5 * rj, dj);              // sqrt(d)
  const double res = t * rj;
  return fma(fma(-res, dj, t), rj, res);                // t / sqrt(d)
}
__device__ __forceinline__ float pivot_scale(float t, float, float rj) { return t * rj; }
__device__ __forceinline__ double kpivot_rsqrt(double d) { return krsqrt(d); }
__device__ __forceinline__ float kpivot_rsqrt(float d) { return __builtin_amdgcn_rsqf(d); }

// The serial tail of a column step - pivot broadcast -> 1/sqrt -> scale - as numbered stages of one or two independent
// instructions, so that factor16_step can issue ONE stage between two groups of the next step's dot products: a lone wave issues in
// order, and left to itself the compiler emits the ~25 dependent f64 operations of the tail back to back (each waiting ~9 cycles
// for the one before), then the independent work.
template <typename T>
struct PivotTail;
template <>
struct PivotTail<double> {
  // Round 5: 7 dependent stages instead of 13.  Rounds 3-4 refined r = 1/sqrt(d) with two Newton steps, formed sqrt(d) with a residual
  // correction and divided t and sx by it with another (S0 .. S12).  The quotient t / sqrt(d) needs none of the intermediate values:
  // with r0 = v_rsq_f64(d) (relative error eps ~ 2^-26) and e = 1 - d r0^2 (= -2 eps - eps^2, formed with one FMA from the rounded
  // product d r0: absolute error 2^-53),  1 / sqrt(d) = r0 (1 - e)^(-1/2) = r0 (1 + e / 2 + 3 e^2 / 8 + O(e^3)), so
  //   t / sqrt(d) = t r0 + (t r0) e (1/2 + 3/8 e)          (truncation 5/16 e^3 ~ 2^-77; rounding: within 1 ulp of the quotient)
  // - the same accuracy class as before ("within 1 ulp of sqrt / divide"), half the dependent chain.  SVGP_POTF2_TAIL13=1 (build-time
  // A/B) keeps the round-4 tail.
#if defined(SVGP_POTF2_TAIL13) && SVGP_POTF2_TAIL13
  static constexpr int NST = 13;
  double t, sx, d, r, u, h, e, dj, rt, rs, c1, c2;
  template <int J, int S>
  __device__ __forceinline__ void stage(double (&row)[16], double (&x)[16], int& bad) {
    if constexpr (S == 0) { d = row_bcast<J>(t); if (!(d > 0.0) && !bad) bad = J + 1; }
    if constexpr (S == 1) r = __builtin_amdgcn_rsq(d);                      // ~26 bits
    if constexpr (S == 2 || S == 5) { u = -d * r; h = 0.5 * r; }
    if constexpr (S == 3 || S == 6) e = fma(u, r, 1.0);
    if constexpr (S == 4 || S == 7) r = fma(h, e, r);                       // two Newton steps: 1/sqrt(d)
    if constexpr (S == 8) { dj = d * r; rt = t * r; rs = sx * r; h = 0.5 * r; }
    if constexpr (S == 9) c1 = fma(-dj, dj, d);
    if constexpr (S == 10) dj = fma(c1, h, dj);                             // sqrt(d)
    if constexpr (S == 11) { c1 = fma(-rt, dj, t); c2 = fma(-rs, dj, sx); }
    if constexpr (S == 12) { row[J] = fma(c1, r, rt); x[J] = fma(c2, r, rs); }   // t / sqrt(d), sx / sqrt(d)
  }
#else
  static constexpr int NST = 7;
  double t, sx, d, r, u, e, w, rt, rs;
  template <int J, int S>
  __device__ __forceinline__ void stage(double (&row)[16], double (&x)[16], int& bad) {
    if constexpr (S == 0) { d = row_bcast<J>(t); if (!(d > 0.0) && !bad) bad = J + 1; }
    if constexpr (S == 1) r = __builtin_amdgcn_rsq(d);                      // r0, ~26 bits
    if constexpr (S == 2) { u = d * r; rt = t * r; rs = sx * r; }
    if constexpr (S == 3) e = fma(-u, r, 1.0);                              // 1 - d r0^2
    if constexpr (S == 4) w = fma(e, 0.375, 0.5);
    if constexpr (S == 5) w = w * e;                                        // e / 2 + 3 e^2 / 8
    if constexpr (S == 6) { row[J] = fma(rt, w, rt); x[J] = fma(rs, w, rs); }   // t / sqrt(d), sx / sqrt(d)
  }
  // (measured too, profiles/round5/potf2_ab.md: a 6-stage form - e (t r0 / 2 + e 3/8 t r0) as two FMAs on pre-scaled copies - with the
  // k = J - 1 term landing on the pre-added sum of the bulk accumulators: one stage and one add less on the chain, three instructions more
  // per column, and SLOWER (factor16 4.4k -> 4.8k cycles): from J = 3 on a column step is bound by the ISSUE of its 2 J DPP FMACs plus the
  // tail's instructions, not by the chain's latency)
#endif
};
template <>
struct PivotTail<float> {
  static constexpr int NST = 3;
  float t, sx, d, r;
  template <int J, int S>
  __device__ __forceinline__ void stage(float (&row)[16], float (&x)[16], int& bad) {
    if constexpr (S == 0) { d = row_bcast<J>(t); if (!(d > 0.0f) && !bad) bad = J + 1; }
    if constexpr (S == 1) r = __builtin_amdgcn_rsqf(d);                     // v_rsq_f32, 1 ulp
    if constexpr (S == 2) { row[J] = t * r; x[J] = sx * r; }
  }
};

// Column step J, software-pipelined by hand: the sums of step J + 1 over the columns k < J (final since step k) are formed
// BESIDE the serial tail of step J, one tail stage per column k, pinned in that order by scheduling barriers; step J + 1 then
// only adds its k = J term.  pt / ps carry those partial sums (two accumulators each).
template <typename T, int J, int K>
__device__ __forceinline__ void factor16_bulk(T (&row)[16], T (&x)[16], T (&n)[2], T (&m)[2], PivotTail<T>& tail, int& bad) {
  if constexpr (K < J || K < PivotTail<T>::NST) {
    if constexpr (K < PivotTail<T>::NST) tail.template stage<J, K>(row, x, bad);
    if constexpr (K < J && J + 1 < 16) {
      fmac_bcast<J + 1, false>(n[K & 1], row[K], row[K]);   // n -= L[J+1][K] * row[K]
      fmac_bcast<J + 1, false>(m[K & 1], row[K], x[K]);     // m -= L[J+1][K] * x[K]
    }
    __builtin_amdgcn_sched_barrier(0);
    factor16_bulk<T, J, K + 1>(row, x, n, m, tail, bad);
  }
}
// Broadcasts: a DPP operand costs 8 cycles of issue in fp32 and ~16 on the DP ALU (a column step of 2J DPP-FMACs + 27 tail
// instructions takes 307 cycles in f64).  Measured and rejected: finished columns parked in a 16 x 16 LDS buffer and read back as
// same-address (broadcast) LDS reads for all but the k = J - 1 term - 7.6k instead of 4.9k cycles per factor (f64; 6.0k vs 3.6k
// fp32): the write -> read round trip of every step lands on the chain.  Round 5, measured and rejected: the BULK terms' broadcast through
// an SGPR (two v_readlane_b32 + two plain v_fma_f64 with an SGPR-pair operand instead of two DPP FMACs; the k = J - 1 term stays DPP):
// fp32 factor16 3.6k -> 4.3k cycles, prep at M = 128 66-68 -> 72-74 us f64, 62 -> 65-73 us fp32 (profiles/round5/potf2_sgpr.log) - the
// v_readlane -> SGPR -> VALU round trip costs more than the DPP operand's 8 / 16 issue cycles even off the dependent chain.
template <typename T, int J>
__device__ __forceinline__ void factor16_step(T (&row)[16], T (&x)[16], T (&pt)[2], T (&ps)[2], int& bad) {
  T t1 = pt[1], s1 = ps[1];
  if constexpr (J >= 1) {
    fmac_bcast<J, true>(t1, row[J - 1], row[J - 1]);   // L[J][J-1] comes from lane J of this lane's 16-lane row
    fmac_bcast<J, false>(s1, row[J - 1], x[J - 1]);
  }
  PivotTail<T> tail;
  tail.t = pt[0] + t1;
  tail.sx = ps[0] + s1;
  T n[2] = {J + 1 < 16 ? row[J + 1 < 16 ? J + 1 : 15] : T(0), T(0)}, m[2] = {J + 1 < 16 ? x[J + 1 < 16 ? J + 1 : 15] : T(0), T(0)};
  __builtin_amdgcn_sched_barrier(0);
  factor16_bulk<T, J, 0>(row, x, n, m, tail, bad);
  pt[0] = n[0]; pt[1] = n[1]; ps[0] = m[0]; ps[1] = m[1];
  // nothing moves across a column step: left alone the compiler hoists the broadcasts of ALL later steps (row[k] is final from
  // step k on) - up to 120 live broadcast values, 256 VGPRs + 214 AGPRs, every broadcast parked in an AGPR and fetched back
  // (700 v_accvgpr moves on the chain of the f64 kernel: 6.8k -> 5.5k cycles per 16 x 16 factor without them)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k <= J; ++k) asm volatile("" : "+v"(row[k]), "+v"(x[k]));
  asm volatile("" : "+v"(pt[0]), "+v"(pt[1]), "+v"(ps[0]), "+v"(ps[1]));
}

// Round 5, measured and rejected (profiles/round5/potf2_ab.md): a SPLIT factor - lanes 0-31 running the L recurrence and lanes 32-63 the X
// recurrence in the same instructions (one DPP FMAC per (J, k) instead of two; the pivot and each new column of L duplicated into the
// upper half by v_permlane32_swap_b32).  Correct (same residuals), but SLOWER: factor16 4.4k -> 5.7k cycles f64, 3.7k -> 4.3k fp32.  The
// column step is bound by the LATENCY of its dependent chain (~27 cycles per dependent f64 operation for a lone wave), not by the issue
// cycles of the DPP FMACs, and the two duplications add two stages to every column.
template <typename T>
__device__ __forceinline__ int factor16(T* __restrict__ sm, T* __restrict__ dinv, int LD, int DL, int o, int p, int lane) {
  const int l15 = lane & 15;
  T row[16], x[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    row[c] = sm[(o + l15) * LD + o + c];
    x[c] = (l15 == c) ? T(1) : T(0);
  }
  int bad = 0;
  T pt[2] = {row[0], T(0)}, ps[2] = {x[0], T(0)};
  factor16_step<T, 0>(row, x, pt, ps, bad);  factor16_step<T, 1>(row, x, pt, ps, bad);  factor16_step<T, 2>(row, x, pt, ps, bad);
  factor16_step<T, 3>(row, x, pt, ps, bad);  factor16_step<T, 4>(row, x, pt, ps, bad);  factor16_step<T, 5>(row, x, pt, ps, bad);
  factor16_step<T, 6>(row, x, pt, ps, bad);  factor16_step<T, 7>(row, x, pt, ps, bad);  factor16_step<T, 8>(row, x, pt, ps, bad);
  factor16_step<T, 9>(row, x, pt, ps, bad);  factor16_step<T, 10>(row, x, pt, ps, bad); factor16_step<T, 11>(row, x, pt, ps, bad);
  factor16_step<T, 12>(row, x, pt, ps, bad); factor16_step<T, 13>(row, x, pt, ps, bad); factor16_step<T, 14>(row, x, pt, ps, bad);
  factor16_step<T, 15>(row, x, pt, ps, bad);
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c <= l15) sm[(o + l15) * LD + o + c] = row[c];
      dinv[(p * 16 + c) * DL + l15] = x[c];
    }
  }
  return bad;
}

// Block step p, with lookahead: after the panel  L[t,p] = A[t,p] inv(D_p)'  (all waves), wave 0 updates only the next
// diagonal block and factors it (the serial chain above) WHILE the other waves work through the step's remaining items - the rest
// of the trailing update  A[ti,tj] -= L[ti,p] L[tj,p]'  and row p of the block inverse  X[p,tj] = -inv(D_p) sum_{tj<=s<p} L[p,s] X[s,tj]
// (the accumulator of the sum is fed straight back as the B operand of the second product: register r of a 16x16 result
// is k-slab r of a B fragment when A is read with k = Mfma16::row(lane, r); X[ti,tj]' lives in the strictly upper block
// (tj,ti) of the LDS image) - pulled from an LDS counter (potf2_items; wave 0 joins when its factor is done).
// Round 3, s_memtime stamps (tools/potf2_time.py; cycles per 128-block, f64 / fp32): 97k / 76k -> 84k / 69k.
//   * the register factor: 6.8k -> 4.9k (f64) per 16 x 16 block - the compiler had hoisted the broadcasts of all later column
//     steps (256 VGPRs + 214 AGPRs, 700 v_accvgpr moves on the chain: now pinned per step), and the broadcasts fold into
//     v_fmac_f64_dpp (factor16_step);
//   * the workers: a static round-robin left wave 0 waiting 1-5k cycles at EVERY step's barrier (early steps: 27 / 20 / 14
//     trailing tiles; late steps: the inverse rows, up to 8 chained products per tile) - now a work queue, longest items first.
// What is left per block (f64): 8 x 4.9k factor + 8 x 0.8k diagonal-tile update + 8 x 0.8-1.3k panel + load 7k + store 9k.
// Round 3, measured and rejected (s_memtime stamps on wave 0 and on a worker wave, fp32, same box): a restructured body - the
// 32 leading columns loaded first and the rest beside the first factor, finished L panels / inverse rows written out by the
// workers during the steps (LDS-only barriers), tiles two at a time, wave 0 taking a share of the tiles after its factor, the
// block inverse accumulated right-looking (no dot-product chains), a conflict-free LDS stride - was SLOWER end to end (prep at
// M = 1024: 0.595 vs 0.523 ms f64, 0.549 vs 0.507 ms fp32): every store moved into the steps lengthened them.  Also rejected:
// 32 loads in flight or a predicate skipping the blocks above the diagonal in the load phase (one CU's share of the memory pipe
// either way); the software-pipelined column step by itself (the chain is issue-bound: a DPP operand costs 8 / 16 cycles);
// trailing tiles pulled three at a time with their LDS reads and MFMA chains interleaved (a tile still costs ~800 cycles: the
// coarser items only unbalance the waves - 86k / 72k instead of 84k / 69k cycles per block); moving only the 36 blocks on and
// below the diagonal in the load and store phases, one block per workgroup pass (load 7.0k -> 6.6k, store 8.8k -> 10.0k: 128-byte
// segments instead of 512-byte ones cost what the fewer instructions save).
// The body is a device function of a 256-thread workgroup (smem_raw: potf2_lds_bytes<T>() of dynamic LDS) so that the
// trailing-update kernels can run it on the NEXT diagonal block the moment that block is up to date (potrf_t below).
// A[ti, tj] -= L[ti, p] L[tj, p]' on the LDS image of the 128-block: every LDS read of the tile first (the old tile is the
// accumulator's start: no read-modify-write behind the MFMAs)
template <typename T>
__device__ __forceinline__ void potf2_update_tile(T* __restrict__ sm, int p, int ti, int tj, int lane) {
  constexpr int LD = kNB + 1;
  using M16 = Mfma16<T>;
  using acc_t = typename M16::acc_t;
  const int l15 = lane & 15, g = lane >> 4, o = 16 * p, bi = 16 * ti, bj = 16 * tj;
  T fa[4], fb[4];
  acc_t acc;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = 4 * s + g;
    fa[s] = sm[(bi + l15) * LD + o + k];
    fb[s] = sm[(bj + l15) * LD + o + k];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = sm[(bi + M16::row(lane, r)) * LD + bj + l15];
  // (round 5: the sign is applied HERE - negated at the load, every fragment read was waited for on the spot: three LDS round trips per
  // tile instead of one)
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = M16::mma(-fa[s], fb[s], acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) sm[(bi + M16::row(lane, r)) * LD + bj + l15] = acc[r];
}

// Work items of block step p besides wave 0's chain, longest first: the rows of the block inverse X[p, tj] (tj < p: p - tj + 1 tile
// products each), then the trailing tiles (one product each).  Pulled from an LDS counter by the workers at once and by wave 0
// when its factor is done (the round-2 static round-robin left wave 0 waiting 1-5k cycles at every step's barrier: the workers'
// ~190 tile products per block at 600-950 cycles each exceed wave 0's 8 x 4k chain).  The inverse rows prefetch the operand
// fragments of the next product before the MFMAs of the current one.
// Round 5, measured and rejected (profiles/round5/potf2_ab.md): the queue pop pipelined (the atomic for the next item issued before the
// current item's work) with the trailing tiles software-pipelined over two operand sets - fp32 unchanged, f64 items 8.5k -> 9.7k cycles in
// block step 0 (two operand sets in a kernel already at 256 VGPRs + AGPRs).
template <typename T>
__device__ __forceinline__ void potf2_items(T* __restrict__ sm, T* __restrict__ dinv, int* __restrict__ queue, int p, int lane) {
  constexpr int NB = kNB, LD = NB + 1, NBLK = NB / 16, DL = 17;
  using M16 = Mfma16<T>;
  using acc_t = typename M16::acc_t;
  const int l15 = lane & 15, g = lane >> 4, o = 16 * p;
  const bool last = (p + 1 == NBLK);
  const int n = NBLK - p - 1, ntr = n * (n + 1) / 2 - (last ? 0 : 1), nit = p + ntr;
  auto inverse_tile = [&](int tj2) {   // inverse tile X[p, tj2]
    T a0[4], b0[4], a1[4], b1[4];   // two operand sets with static names (a runtime-indexed pair lives in scratch memory)
    auto ld = [&](int sb, T (&a)[4], T (&b)[4]) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int k = 4 * s4 + g;
        a[s4] = sm[(o + l15) * LD + 16 * sb + k];
        b[s4] = (sb == tj2) ? dinv[(tj2 * 16 + k) * DL + l15] : sm[(16 * tj2 + l15) * LD + 16 * sb + k];
      }
    };
    acc_t acc = {0, 0, 0, 0};
    ld(tj2, a0, b0);
    for (int sb = tj2; sb < p; sb += 2) {
      if (sb + 1 < p) ld(sb + 1, a1, b1);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = M16::mma(a0[s4], b0[s4], acc);
      if (sb + 1 < p) {
        if (sb + 2 < p) ld(sb + 2, a0, b0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = M16::mma(a1[s4], b1[s4], acc);
      }
    }
    acc_t x = {0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) x = M16::mma(dinv[(p * 16 + l15) * DL + M16::row(lane, s4)], acc[s4], x);
#pragma unroll
    for (int r = 0; r < 4; ++r) sm[(16 * tj2 + l15) * LD + o + M16::row(lane, r)] = -x[r];
  };
  for (;;) {
    int it = 0;
    if (lane == 0) it = atomicAdd(queue, 1);
    it = __builtin_amdgcn_readfirstlane(it);
    if (it >= nit) break;
    if (it < p) {
      inverse_tile(it);
    } else {        // trailing tile number q of the lower triangle of the trailing block ((0, 0) is wave 0's unless last)
      int q = it - p + (last ? 0 : 1), ti = 0;
      while (q >= ti + 1) { q -= ti + 1; ++ti; }
      potf2_update_tile<T>(sm, p, p + 1 + ti, p + 1 + q, lane);
    }
  }
}

template <typename T>
constexpr size_t potf2_lds_bytes() { return (size_t(kNB) * (kNB + 1) + 8 * 16 * 17) * sizeof(T); }

// NW (round 5): waves of the workgroup that take part - 4 (256 threads: the round 1-4 shape) or 8 (512 threads).  Wave 0 runs the column
// chain; with seven worker waves instead of three the block steps whose tiles outlast the chain (steps 0-2 and the last) become
// chain-bound, the panel solves take one pass and the load / store phases have twice the loads in flight.
template <typename T, int NW = 4>
__device__ __forceinline__ void potf2_body(T* __restrict__ A, T* __restrict__ Tm, int64_t ld, int* __restrict__ info, int pbase,
                                              unsigned char* __restrict__ smem_raw) {
  constexpr int NB = kNB, LD = NB + 1, NBLK = NB / 16, DL = 17, NTH = 64 * NW;
  static_assert((NB * NB) % (NTH * 16) == 0, "the load / store passes move 16 elements per thread");
  using M16 = Mfma16<T>;
  using acc_t = typename M16::acc_t;
  T* sm = reinterpret_cast<T*>(smem_raw);   // [128][129] row-major block; strictly-upper 16-blocks later hold X'
  T* dinv = sm + NB * LD;                   // [8][16][17]  inverses of the 16x16 diagonal blocks
  __shared__ int failed;
  __shared__ int queue;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  SVGP_STAMP(0);
  if (tid == 0) failed = (*info != 0) ? -1 : 0;
  {
    // 64 elements per thread, 16 loads in flight at a time (a plain loop waits for every load before its LDS store).  32 in flight,
    // or skipping the 16-blocks above the diagonal with a predicate, are not faster (7.5k cycles of the f64 block's 88k either way:
    // one CU's share of the memory pipe)
    constexpr int U = 16;
    for (int e0 = tid; e0 < NB * NB; e0 += NTH * U) {
      T v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * NTH;
        v[u] = A[(e % NB) + int64_t(e / NB) * ld];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * NTH;
        sm[(e % NB) * LD + e / NB] = v[u];
      }
    }
  }
  __syncthreads();
  if (failed) return;  // an earlier panel already reported the first bad pivot
  SVGP_STAMP(1);
  if (wave == 0) {
    const int bad = factor16(sm, dinv, LD, DL, 0, 0, lane);
    if (bad && lane == 0) failed = pbase + bad;
    SVGP_STAMP(42);
  }
  __syncthreads();

  for (int p = 0; p < NBLK; ++p) {
    const int o = 16 * p;
    SVGP_STAMP(2 + 4 * p);
    if (failed) {
      if (tid == 0) *info = failed;
      return;
    }
    // ---- panel: L[t, p] = A[t, p] inv(D_p)' ----
    for (int t = p + 1 + wave; t < NBLK; t += NW) {
      acc_t acc = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = 4 * s + g;
        acc = M16::mma(sm[(16 * t + l15) * LD + o + k], dinv[(p * 16 + l15) * DL + k], acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sm[(16 * t + M16::row(lane, r)) * LD + o + l15] = acc[r];
    }
    if (tid == 0) queue = 0;   // the step's item queue
    __syncthreads();
    SVGP_STAMP(3 + 4 * p);
    const bool last = (p + 1 == NBLK);          // no next factor
    if (wave == 0 && !last) {
      potf2_update_tile<T>(sm, p, p + 1, p + 1, lane);
      SVGP_STAMP(44 + 2 * p);
      const int bad = factor16(sm, dinv, LD, DL, o + 16, p + 1, lane);
      if (bad && lane == 0) failed = pbase + o + 16 + bad;
      SVGP_STAMP(45 + 2 * p);
    }
    SVGP_STAMPW(64 + 4 * p);
    // 512 threads: waves w and w + 4 share a SIMD (a workgroup's waves are dealt cyclically over the four), and f64 MFMA and VALU work
    // never co-execute on one: with wave 4 pulling tiles beside it wave 0's column chain ran 5.2 k instead of 4.4 k cycles per 16 x 16
    // factor (stamps, profiles/round5/potf2_ab.md).  Wave 4 therefore sits out the steps that carry a factor: six workers.
    if (!(NW == 8 && wave == 4 && !last)) potf2_items<T>(sm, dinv, &queue, p, lane);
    SVGP_STAMPW(66 + 4 * p);
    __syncthreads();
    SVGP_STAMP(4 + 4 * p);
  }
  if (failed) {   // a bad pivot in the last block
    if (tid == 0) *info = failed;
    return;
  }
  SVGP_STAMP(40);
  {
    constexpr int U = 16;
    for (int e0 = tid; e0 < NB * NB; e0 += NTH * U) {
      T lv[U], xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * NTH, r = e % NB, c = e / NB;
        lv[u] = sm[r * LD + c];
        xv[u] = ((r >> 4) == (c >> 4)) ? dinv[((r >> 4) * 16 + (r & 15)) * DL + (c & 15)] : sm[c * LD + r];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * NTH, r = e % NB, c = e / NB;
        if (r >= c) {   // Tm is zero above the diagonal on entry and stays so (launch_potrf's contract): nothing to store there
          A[r + int64_t(c) * ld] = lv[u];
          Tm[r + int64_t(c) * ld] = xv[u];
        }
      }
    }
  }
  SVGP_STAMP(41);
}


template <typename T>
__global__ void __launch_bounds__(kThreads, 2) potf2_kernel(T* __restrict__ A, T* __restrict__ Tm, int64_t ld,
                                                             int* __restrict__ info, int pbase) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  potf2_body<T, kThreads / 64>(A, Tm, ld, info, pbase, smem_raw);
}

// ------------------------------------------------------------------------------------------------
// MFMA tile kernels of the blocked Cholesky (and of the T panels, which ride in its launches).
// ------------------------------------------------------------------------------------------------
enum : int { MODE_TRSM = 0, MODE_SYRK = 1, MODE_COL = 2 };
#ifndef SVGP_CHOL_TILE_ASYNC
#define SVGP_CHOL_TILE_ASYNC 1
#endif

__device__ __forceinline__ void tri_index(int b, int& i, int& j) {  // b -> (i >= j) in row-major triangle order
  i = 0;
  while (b >= i + 1) {
    b -= i + 1;
    ++i;
  }
  j = b;
}

// Latency-bound small-grid variant of the TRSM / SYRK tiles: one 128x128 f64 tile product is 14 us of MFMA time on a
// single CU, and a panel step of a small Kuu has only a handful of tiles, so each output tile is split into 128/NT column
// chunks on 256-thread workgroups (f64: 4 x (128 x 32), f32: 2 x (128 x 64)) that run on different CUs
// (rocprofv3, M = 1024 f64: TRSM 22.5 -> 11.9 us, SYRK 28 -> 12-19 us per panel step).
// Round 3 - what each launch of the blocked factorisation carries (potrf_t):
//   MODE_TRSM launch of panel p:  tiles [0, n): L[i, p] = A[i, p] inv(L_pp)' for the n row blocks below the diagonal;
//                                 tiles [n, n + p): the T panels of block ROW p, T[p, J] = -inv(L_pp) L[p, J] for J < p (they
//                                 need nothing but inv(L_pp) and the finished row p of L: formerly a launch of their own at the end).
//   MODE_SYRK launch of panel p:  A[i, j] -= L[i, p] L[j, p]' for p < j <= i; with FUSE the workgroup that completes tile
//                                 (p+1, p+1) - the last of its NCH column chunks to arrive, counted in sync[p] - goes straight on
//                                 to factor that diagonal block (potf2_body) while the other workgroups finish the trailing
//                                 update: the next panel's factorisation no longer waits for the whole update, nor for a launch.
// (the fused f64 form asks for one workgroup per CU: potf2's LDS image of an f64 block - 146 KiB - allows no second one anyway, and
// the unrolled DPP factor wants more than the 256 VGPRs a two-workgroup bound leaves it: 282 spilled registers otherwise)
// Round 4, two-level blocking (potrf_t, large Kuu): MODE_COL is the trailing update restricted to block column p + 1 - tiles (i, p + 1),
// i > p - so that the second block column of a 256-wide outer panel can be factored before the REST of the trailing matrix is touched;
// that rest then takes ONE update of rank 256 (kb = 2: the contraction runs over the two block columns p - 1, p) instead of two of
// rank 128: half the read-modify-write traffic of the trailing matrix, which is what bounds these launches (a 128 x 128 tile read
// and written per 4.2 MFLOP).
// P8 (round 5, FUSE only): the kernel is launched with 512 threads.  Waves 0-3 are the 256-thread tile kernel unchanged (TileGemm
// indexes by threadIdx.x < 256); waves 4-7 are PASSENGERS that only execute the tile loop's barriers (TileGemm::*_barrier_count: same
// count by construction of the loops) and then join the block factorisation, which so has seven worker waves instead of three.
// One workgroup per CU (the block's LDS image), so potrf_t takes this form only where the launch fits the chip in one round anyway.
template <typename T, int MODE, int NT, bool FUSE = false, bool P8 = false>
__global__ void __launch_bounds__(P8 ? kThreads : k256, P8 ? 2 : ((FUSE && sizeof(T) == 8) ? 1 : 2)) chol_tile_kernel(T* __restrict__ A, T* __restrict__ Tm, int64_t ld, int p, int n,
                                                            int* __restrict__ info, unsigned* __restrict__ sync, int kb = 1) {
  using G = TileGemm<T, NT, 16, k256>;
  using QRegs = typename G::QRegs;
  constexpr int NB = kNB, NCH = NB / NT;
  static_assert(!P8 || (FUSE && MODE == MODE_SYRK), "passenger waves exist for the fused trailing update only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  typename G::Acc acc;
  acc.zero();
  const int chunk = blockIdx.x % NCH, tile = blockIdx.x / NCH;
  const bool passenger = P8 && threadIdx.x >= k256;   // wave-uniform
  if (passenger) {
    const int nst = (MODE == MODE_SYRK ? kb : 1) * (NB / 16);
    const int nbar = (SVGP_CHOL_TILE_ASYNC && G::kAsync) ? G::async_barrier_count(nst) : G::loop_barrier_count(nst);
    for (int q = 0; q < nbar; ++q) __builtin_amdgcn_s_barrier();
  } else {
  if (MODE == MODE_TRSM && (p < 0 || tile >= n)) {   // T[p, J] = -inv(L_pp) L[p, J]
    int J = tile - n;
    if (p < 0) {   // all block rows in one launch (large Kuu: potrf_t): tile -> (row I >= 1, column J < I)
      int ti, tj;
      tri_index(tile, ti, tj);
      p = ti + 1;
      J = tj;
    }
    const T* P = Tm + int64_t(p) * NB + int64_t(p) * NB * ld;
    const T* Q = A + int64_t(p) * NB + (int64_t(J) * NB + chunk * NT) * ld;   // element (k, c) at Q[k + c * ld]
    auto qload = [&](int t, QRegs& r) { G::load_q_trans(r, Q + int64_t(t) * 16, ld); };
    G::loop(acc, P, ld, NB / 16, qload, smem);
    T* C = Tm + int64_t(p) * NB + (int64_t(J) * NB + chunk * NT) * ld;
#pragma unroll
    for (int a = 0; a < G::MI; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < G::NJ; ++b) C[G::acc_row(a, r) + int64_t(G::acc_col(b)) * ld] = -acc.v[a][b][r];
    return;
  }
  int i, j;
  const T* P;
  if (MODE == MODE_TRSM) {       // L[i, p] = A[i, p] inv(L_pp)', computed as X' = inv(L_pp) A[i, p]' so that stores run along columns of A
    i = p + 1 + tile;
    j = p;
    P = Tm + int64_t(p) * NB + int64_t(p) * NB * ld;
  } else if (MODE == MODE_COL) {   // A[i, p + 1] -= L[i, p] L[p + 1, p]'  (block column p + 1 only)
    i = p + 1 + tile;
    j = p + 1;
    P = A + int64_t(j) * NB + int64_t(p) * NB * ld;
  } else {                       // A[i, j] -= L[i, p - kb + 1 .. p] L[j, p - kb + 1 .. p]'
    int ti, tj;
    tri_index(tile, ti, tj);
    i = p + 1 + ti;
    j = p + 1 + tj;
    P = A + int64_t(j) * NB + int64_t(p - kb + 1) * NB * ld;
  }
  const int kcol = (MODE == MODE_SYRK) ? p - kb + 1 : p;   // first block column of the contraction
  const T* Q = A + int64_t(i) * NB + int64_t(kcol) * NB * ld + chunk * NT;
  // SVGP_CHOL_TILE_ASYNC (build-time A/B, default 1): both operands global -> LDS by DMA through three buffers (the strips' loop) instead
  // of the two-buffer register-staged loop: these 8-step products run at the latency of their loads.  fp32 (64-column chunks): prep
  // M = 512 0.252 -> 0.245 ms, 2048 1.09 -> 1.06, 4096 1.89 -> 1.77 (same box, three runs).  The f64 tile shape (four 32-column chunks: one
  // column tile per wave) has no asynchronous loop and keeps the two-buffer one; two 64-column f64 chunks ON the asynchronous loop were
  // measured too: M = 1024 0.535 -> 0.615 ms, 4096 2.23 -> 2.48 - half the workgroups per tile costs more than the deeper pipeline gives.
  if constexpr (SVGP_CHOL_TILE_ASYNC && G::kAsync) {
    auto qsrc = [&](int t) { return Q + int64_t(t) * 16 * ld; };
    G::template loop_tri_async<0>(acc, P, ld, (MODE == MODE_SYRK ? kb : 1) * (NB / 16), qsrc, smem, ld);
  } else {
    const typename G::QOff qoff = G::q_offsets(ld);
    auto qload = [&](int t, QRegs& r) { G::load_q(r, Q + int64_t(t) * 16 * ld, qoff); };
    G::loop(acc, P, ld, (MODE == MODE_SYRK ? kb : 1) * (NB / 16), qload, smem);
  }
  T* C = A + int64_t(i) * NB + int64_t(j) * NB * ld + chunk * NT;
#pragma unroll
  for (int a = 0; a < G::MI; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int b = 0; b < G::NJ; ++b) {
        T* dst = C + G::acc_col(b) + int64_t(G::acc_row(a, r)) * ld;
        if (MODE == MODE_TRSM) *dst = acc.v[a][b][r];
        else *dst -= acc.v[a][b][r];
      }
  }   // (!passenger)
  if constexpr (FUSE && (MODE == MODE_SYRK || MODE == MODE_COL)) {
    if (tile != 0) return;         // tile 0 = (p+1, p+1): the next diagonal block
    __shared__ int is_last;
    __threadfence();               // release: this chunk's stores are visible device-wide before the count goes up
    __syncthreads();
    if (threadIdx.x == 0) is_last = (__hip_atomic_fetch_add(&sync[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == unsigned(NCH - 1));
    __syncthreads();
    if (!is_last) return;
    __threadfence();               // acquire: the other chunks' stores (other CUs, possibly other XCDs) before the loads below
    potf2_body<T, P8 ? 8 : 4>(A + int64_t(p + 1) * NB * (ld + 1), Tm + int64_t(p + 1) * NB * (ld + 1), ld, info, (p + 1) * NB, smem_raw);
  }
}

// The large-grid form of the trailing update (trailing matrices of >= 256 tiles fill the chip by themselves): whole
// 128 x 128 tiles on 512-thread workgroups, A[i, j] -= L[i, p] L[j, p]' computed transposed so that stores run along columns
// of A (half the operand traffic of the chunked form).  FUSE: workgroup 0 owns tile (p+1, p+1) and factors it right after its
// update.  potf2_body is a 256-thread routine: waves 4-7 leave first (s_barrier only waits for the waves of a workgroup that
// have not ended).
template <typename T, bool FUSE>
__global__ void __launch_bounds__(kThreads, 2) syrk128_kernel(T* __restrict__ A, T* __restrict__ Tm, int64_t ld, int p,
                                                               int* __restrict__ info, int kb = 1) {
  using G = TileGemm<T, kNB, 16>;
  using QRegs = typename G::QRegs;
  constexpr int NB = kNB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  typename G::Acc acc;
  acc.zero();
  int ti, tj;
  tri_index(blockIdx.x, ti, tj);
  const int i = p + 1 + ti, j = p + 1 + tj;
  const T* P = A + int64_t(j) * NB + int64_t(p - kb + 1) * NB * ld;   // kb = 2: rank-256 update over block columns p - 1, p
  const T* Q = A + int64_t(i) * NB + int64_t(p - kb + 1) * NB * ld;
  const typename G::QOff qoff = G::q_offsets(ld);
  auto qload = [&](int t, QRegs& r) { G::load_q(r, Q + int64_t(t) * 16 * ld, qoff); };
  G::loop(acc, P, ld, kb * (NB / 16), qload, smem);
  T* C = A + int64_t(i) * NB + int64_t(j) * NB * ld;
#pragma unroll
  for (int a = 0; a < G::MI; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int b = 0; b < G::NJ; ++b) C[G::acc_col(b) + int64_t(G::acc_row(a, r)) * ld] -= acc.v[a][b][r];
  if constexpr (FUSE) {
    if (blockIdx.x != 0) return;   // the owner of tile (p+1, p+1) goes on to factor it
    __threadfence();               // the tile this workgroup just wrote is re-read below (through L2: drop stale L1 lines)
    __syncthreads();
    potf2_body<T, kThreads / 64>(A + int64_t(p + 1) * NB * (ld + 1), Tm + int64_t(p + 1) * NB * (ld + 1), ld, info, (p + 1) * NB, smem_raw);
  }
}

// Rank-256 trailing update on 256 x 256 tiles (round 6; fp32: the C4 regime).  The 128 x 128 x 128 tiles of
// syrk128_kernel move 61 KB per MFLOP across the L2 <-> fabric boundary (two 64 KiB operand panels and 64 KiB of C read and written per
// 4.2 MFLOP) and run at 0.45 of the MFMA rate inside a round at ~4.6 TB/s - bandwidth-bound (profiles/round5/c4_chol_bulk.md).  Here a
// 512-thread workgroup owns a 256 x 256 tile of the trailing matrix and contracts over TWO panels (block columns p - 1, p: the two-level
// schedule of potrf_t): 512 KiB of operands + 512 KiB of C per 33.5 MFLOP = 30 KB per MFLOP.  A wave holds 128 x 64 of the tile
// (8 x 4 MFMA tiles: 128 accumulator VGPRs); the operand k-rows (256 consecutive floats of a column of L: 1 KiB) travel global ->
// registers -> LDS through two buffers, one barrier per 16-deep step; row stride 272 floats = 16 banks: the four k-rows of a fragment read
// sit on disjoint banks.  D rows are the C rows (contiguous in memory): a lane reads-modifies-writes float4 pieces.
// Tile (ti >= tj) in units of 256 rows from block row p + 1; diagonal tiles skip their upper-right quadrant.  FUSE: the workgroup of
// tile (0, 0) factors the next diagonal block (potf2_body, 8 waves) as syrk128_kernel's does.
template <bool FUSE>
__global__ void __launch_bounds__(kThreads, 2) syrk256_kernel(float* __restrict__ A, float* __restrict__ Tm, int64_t ld, int p,
                                                               int* __restrict__ info) {
  using M16 = Mfma16<float>;
  using acc_t = M16::acc_t;
  using V4 = float __attribute__((ext_vector_type(4)));
  constexpr int NB = kNB, TS = 256, BK = 16, LDS_LD = TS + 16, KTOT = 2 * NB, NSTEP = KTOT / BK;
  constexpr int TILE = BK * LDS_LD;            // floats per operand tile in LDS
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);   // [2 buffers][Q tile | P tile]
  int ti, tj;
  tri_index(blockIdx.x, ti, tj);
  const int64_t i0 = int64_t(p + 1 + 2 * ti) * NB, j0 = int64_t(p + 1 + 2 * tj) * NB, kc0 = int64_t(p - 1) * NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;     // wave grid 2 x 4: rows wr * 128, columns wc * 64
  const bool idle = (ti == tj) && wr == 0 && wc >= 2;   // upper-right quadrant of a diagonal tile: never read afterwards
  const float* __restrict__ Qg = A + i0 + kc0 * ld;   // element (k, r) at Qg[k * ld + r]: rows of C
  const float* __restrict__ Pg = A + j0 + kc0 * ld;   // element (k, c) at Pg[k * ld + c]: columns of C
  // staging: thread t moves k-rows t / 64 and t / 64 + 8 of each operand tile, 4 floats at column (t % 64) * 4
  const int srow = tid >> 6, scol = (tid & 63) * 4;
  V4 sq[2], sp[2];
  auto gload = [&](int step) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t k = int64_t(step) * BK + srow + 8 * h;
      sq[h] = *reinterpret_cast<const V4*>(Qg + k * ld + scol);
      sp[h] = *reinterpret_cast<const V4*>(Pg + k * ld + scol);
    }
  };
  auto sstore = [&](int buf) {
    float* q = smem + buf * 2 * TILE;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<V4*>(q + (srow + 8 * h) * LDS_LD + scol) = sq[h];
      *reinterpret_cast<V4*>(q + TILE + (srow + 8 * h) * LDS_LD + scol) = sp[h];
    }
  };
  acc_t acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = acc_t{0, 0, 0, 0};
  gload(0);
  sstore(0);
  __syncthreads();
  const float* fa0 = smem + g * LDS_LD + wr * 128 + l15;          // A operand (D rows = C rows): Q tile
  const float* fb0 = smem + TILE + g * LDS_LD + wc * 64 + l15;    // B operand (D columns = C columns): P tile
#pragma unroll 1
  for (int step = 0; step < NSTEP; ++step) {
    const int buf = step & 1;
    if (step + 1 < NSTEP) gload(step + 1);
    if (!idle) {
      const float* fa = fa0 + buf * 2 * TILE;
      const float* fb = fb0 + buf * 2 * TILE;
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        float av[8], bv[4];
#pragma unroll
        for (int a = 0; a < 8; ++a) av[a] = fa[ks * 4 * LDS_LD + a * 16];
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = fb[ks * 4 * LDS_LD + b * 16];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = M16::mma(av[a], bv[b], acc[a][b]);
      }
    }
    if (step + 1 < NSTEP) sstore(buf ^ 1);   // (last read in step - 1, behind that step's barrier)
    __syncthreads();
  }
  if (!idle) {
    // C[r][c] at A[(i0 + r) + (j0 + c) * ld]; D tile (a, b): rows wr * 128 + a * 16 + 4 g + q, column wc * 64 + b * 16 + l15
    float* __restrict__ Cw = A + i0 + wr * 128 + 4 * g + (j0 + wc * 64 + l15) * ld;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        V4* dst = reinterpret_cast<V4*>(Cw + a * 16 + int64_t(b) * 16 * ld);
        V4 c = *dst;
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] -= acc[a][b][q];
        *dst = c;
      }
  }
  if constexpr (FUSE) {
    if (blockIdx.x != 0) return;   // the owner of tile (0, 0), which holds block (p + 1, p + 1), goes on to factor it
    __threadfence();               // the block this workgroup just wrote is re-read below (through L2: drop stale L1 lines)
    __syncthreads();
    potf2_body<float, kThreads / 64>(A + int64_t(p + 1) * NB * (ld + 1), Tm + int64_t(p + 1) * NB * (ld + 1), ld, info, (p + 1) * NB, smem_raw);
  }
}
constexpr size_t kSyrk256LdsBytes = size_t(2) * 2 * 16 * (256 + 16) * sizeof(float);

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_q_kernel(const T* __restrict__ Lq, int64_t ldq, const T* __restrict__ m, int64_t M, int64_t Mp,
                              T* __restrict__ U, T* __restrict__ mp) {
  // U[j, k] = Lq[k, j] for k >= j (both < M), else 0.  32x32 LDS transpose keeps both sides coalesced.
  __shared__ T tile[32][33];
  const int64_t bj = int64_t(blockIdx.x) * 32, bk = int64_t(blockIdx.y) * 32;
  if (mp && blockIdx.y == 0 && threadIdx.y == 0) {
    const int64_t i = bj + threadIdx.x;
    mp[i] = (i < M) ? m[i] : T(0);
  }
  if (bj / kNB > bk / kNB) return;   // 128-tiles below the diagonal of the upper-triangular U: never read (phase 2 starts at the diagonal block)
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int64_t k = bk + threadIdx.x, j = bj + r;  // read Lq[k + j*M]: consecutive threads -> consecutive k
    tile[r][threadIdx.x] = (k < M && j < M && k >= j) ? Lq[k + j * ldq] : T(0);
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int64_t j = bj + threadIdx.x, k = bk + r;  // write U[j + k*Mp]: consecutive threads -> consecutive j
    U[j + k * Mp] = tile[threadIdx.x][r];
  }
}

template <typename T>
__global__ void kl_colsq_kernel(const T* __restrict__ Lq, int64_t ldq, int64_t M, double* __restrict__ colsq) {
  __shared__ double sh[k256];
  const int64_t j = blockIdx.x;
  double s = 0.0;
  for (int64_t i = j + threadIdx.x; i < M; i += k256) {
    const double v = double(Lq[i + j * ldq]);
    s = fma(v, v, s);
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = k256 / 2; w > 0; w >>= 1) {
    if (int(threadIdx.x) < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) colsq[j] = sh[0];
}

template <typename T>
__global__ void kl_final_kernel(const T* __restrict__ Lq, int64_t ldq, const T* __restrict__ m, const T* __restrict__ Lk, int64_t M,
                                int64_t Mp, const double* __restrict__ colsq, double* __restrict__ scal) {
  __shared__ double sh[4][k256];
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int64_t i = threadIdx.x; i < M; i += k256) {
    s0 += colsq[i];
    const double mi = double(m[i]);
    s1 = fma(mi, mi, s1);
    s2 += log(double(Lq[i + i * ldq]));
    s3 += log(double(Lk[i + i * Mp]));
  }
  sh[0][threadIdx.x] = s0;
  sh[1][threadIdx.x] = s1;
  sh[2][threadIdx.x] = s2;
  sh[3][threadIdx.x] = s3;
  __syncthreads();
  for (int w = k256 / 2; w > 0; w >>= 1) {
    if (int(threadIdx.x) < w)
      for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x < 4) scal[threadIdx.x] = sh[threadIdx.x][0];
}

template <typename T>
__global__ void extract_lower_kernel(const T* __restrict__ A, int64_t Mp, int64_t M, T* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (i >= M) return;
  out[i + j * M] = (i >= j) ? A[i + j * Mp] : T(0);
}

// ------------------------------------------------------------------------------------------------
// x := L \ x (trans = 0) or L' \ x (trans = 1), blocked with the inverted diagonal blocks in Tm.
// One workgroup; x (length Mp) lives in LDS.  Used by posterior(sva) for α (SVA:182), not by the ELBO.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(k256) trsv_kernel(const T* __restrict__ L, const T* __restrict__ Tm, int64_t Mp,
                                                         int trans, T* __restrict__ x) {
  constexpr int NB = kNB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* xs = reinterpret_cast<T*>(smem_raw);  // [Mp]
  T* rs = xs + Mp;                          // [NB] right-hand side of the current block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nP = int(Mp / NB);
  for (int64_t i = tid; i < Mp; i += k256) xs[i] = x[i];
  __syncthreads();
  for (int step = 0; step < nP; ++step) {
    const int I = trans ? nP - 1 - step : step;
    if (!trans) {
      // r_i = x_i - sum_{k < I*NB} L[i, k] x_k ; thread pair (i, h) strides k by 2 -> coalesced along i
      const int i = tid % NB, h = tid / NB;
      double s = 0.0;
      for (int64_t k = h; k < int64_t(I) * NB; k += 2) s = fma(double(L[int64_t(I) * NB + i + k * Mp]), double(xs[k]), s);
      __shared__ double part[2][NB];
      part[h][i] = s;
      __syncthreads();
      if (tid < NB) rs[tid] = T(double(xs[I * NB + tid]) - part[0][tid] - part[1][tid]);
      __syncthreads();
      // x_I = inv(L_II) r : thread pair per row, coalesced along rows
      double t = 0.0;
      for (int k = h; k <= i; k += 2) t = fma(double(Tm[int64_t(I) * NB + i + (int64_t(I) * NB + k) * Mp]), double(rs[k]), t);
      part[h][i] = t;
      __syncthreads();
      if (tid < NB) xs[I * NB + tid] = T(part[0][tid] + part[1][tid]);
      __syncthreads();
    } else {
      // r_i = x_i - sum_{k >= (I+1)*NB} L[k, i] x_k : one wave per row i, lanes along k (contiguous)
      for (int i = wave; i < NB; i += 4) {
        double s = 0.0;
        const T* col = L + (int64_t(I) * NB + i) * Mp;
        for (int64_t k = int64_t(I + 1) * NB + lane; k < Mp; k += 64) s = fma(double(col[k]), double(xs[k]), s);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) rs[i] = T(double(xs[I * NB + i]) - s);
      }
      __syncthreads();
      // x_I = inv(L_II)' r : x_c = sum_{r >= c} Winv[r, c] r_r ; wave per column, lanes along r
      for (int c = wave; c < NB; c += 4) {
        double s = 0.0;
        const T* col = Tm + int64_t(I) * NB + (int64_t(I) * NB + c) * Mp;
        for (int r = c + lane; r < NB; r += 64) s = fma(double(col[r]), double(rs[r]), s);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) xs[I * NB + c] = T(s);
      }
      __syncthreads();
    }
  }
  for (int64_t i = tid; i < Mp; i += k256) x[i] = xs[i];
}


// ------------------------------------------------------------------------------------------------
// X := Lk \ X in place, X lower-triangular Mp x Mp column-major (Centered: B = Lk \ Lq, SVA:133).
// One workgroup per 128-column tile; row panels sequentially, X_I = T[I, 0:(I+1)128] * [X_<I ; X_I].
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads, 2) trsm_mat_kernel(const T* __restrict__ Tm, T* __restrict__ X, int64_t Mp) {
  using G = TileGemm<T, kNB, 16>;
  using QRegs = typename G::QRegs;
  constexpr int NB = kNB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int ct = blockIdx.x, nP = int(Mp / NB);
  T* Xc = X + int64_t(ct) * NB * Mp;  // element (k, c) at Xc[k + c*Mp]
  for (int I = ct; I < nP; ++I) {     // rows above the diagonal tile are zero
    typename G::Acc acc;
    acc.zero();
    const int t0 = ct * (NB / 16);    // X[k, cols] = 0 for k < ct*128
    auto qload = [&](int t, QRegs& r) { G::load_q_trans(r, Xc + int64_t(t0 + t) * 16, Mp); };
    G::loop(acc, Tm + int64_t(I) * NB + int64_t(t0) * 16 * Mp, Mp, (I + 1) * (NB / 16) - t0, qload, smem);
#pragma unroll
    for (int a = 0; a < G::MI; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < G::NJ; ++b) Xc[int64_t(I) * NB + G::acc_row(a, r) + int64_t(G::acc_col(b)) * Mp] = acc.v[a][b][r];
    __syncthreads();
  }
}

template <typename T>
__global__ void pad_lower_kernel(const T* __restrict__ Lq, int64_t M, int64_t Mp, T* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (i >= Mp) return;
  out[i + j * Mp] = (i < M && j < M && i >= j) ? Lq[i + j * M] : T(0);
}

template <typename T>
__global__ void shift_vec_kernel(const T* __restrict__ m, T shift, int64_t M, int64_t Mp, T* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < Mp) out[i] = (i < M) ? m[i] + shift : T(0);
}

// ------------------------------------------------------------------------------------------------
// cov(f, xa, xb) = k(xa, xb) - Aa'Ab + Ca'Cb  (SVA:223-228, :255-264) from the k-major factors the strip
// kernel wrote.  Tile (j-block of xb) x (i-block of xa) so that stores run along columns of the output.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads, 2) cov_assemble_kernel(KernelParams kp, const T* __restrict__ xa, int64_t ldxa,
                                                                    int64_t na, const T* __restrict__ xb, int64_t ldxb,
                                                                    int64_t nb, const T* __restrict__ Aa,
                                                                    const T* __restrict__ Ca, int64_t lda,
                                                                    const T* __restrict__ Ab, const T* __restrict__ Cb,
                                                                    int64_t ldb, int64_t Mp, T* __restrict__ out) {
  using G = TileGemm<T, kNB, 16>;
  using QRegs = typename G::QRegs;
  constexpr int NB = kNB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int64_t i0 = int64_t(blockIdx.x) * NB, j0 = int64_t(blockIdx.y) * NB;
  typename G::Acc acc;
  acc.zero();
  const int nsteps = int(Mp / 16);
  {
    const typename G::QOff qoff = G::q_offsets(lda);
    auto qload = [&](int t, QRegs& r) { G::load_q(r, Aa + int64_t(t) * 16 * lda + i0, qoff); };
    G::loop(acc, Ab + j0, ldb, nsteps, qload, smem);
  }
#pragma unroll
  for (int a = 0; a < G::MI; ++a)
#pragma unroll
    for (int b = 0; b < G::NJ; ++b) acc.v[a][b] = -acc.v[a][b];
  {
    const typename G::QOff qoff = G::q_offsets(lda);
    auto qload = [&](int t, QRegs& r) { G::load_q(r, Ca + int64_t(t) * 16 * lda + i0, qoff); };
    G::loop(acc, Cb + j0, ldb, nsteps, qload, smem);
  }
  const T* __restrict__ invl = static_cast<const T*>(kp.invl);
#pragma unroll
  for (int a = 0; a < G::MI; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int b = 0; b < G::NJ; ++b) {
        const int64_t j = j0 + G::acc_row(a, r), i = i0 + G::acc_col(b);
        if (i < na && j < nb) {
          T r2 = T(0);
          for (int f = 0; f < kp.d; ++f) {
            const T df = (xa[int64_t(f) * ldxa + i] - xb[int64_t(f) * ldxb + j]) * invl[f];
            r2 = fma(df, df, r2);
          }
          out[i + j * na] = kappa<T>(kp.family, r2, T(kp.variance)) + acc.v[a][b][r];
        }
      }
}

void dbg(const char* name, hipStream_t s) {
  static const bool on = [] { const char* e = getenv("SVGP_DEBUG_SYNC"); return e && e[0] == '1'; }();
  if (!on) return;
  hipError_t e = hipPeekAtLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) leave_note(std::string("SVGP_DEBUG_SYNC: ") + name + ": " + hipGetErrorString(e));   // -> svgp_last_error
}

template <typename T>
void potrf_t(hipStream_t s, T* A, T* Tm, int64_t Mp, int* info, unsigned* sync, hipEvent_t* row_events, const RowHook* hook, int ncus) {
  using G = TileGemm<T, kNB, 16>;
  const int nP = int(Mp / kNB);
  constexpr size_t lds_potf2 = potf2_lds_bytes<T>();
  constexpr int CNT = sizeof(T) == 8 ? 32 : 64, NCH = kNB / CNT;
  using GS = TileGemm<T, CNT, 16, k256>;
  constexpr size_t lds_tile = (SVGP_CHOL_TILE_ASYNC && GS::kAsync) ? (GS::ASYNC_LDS_BYTES > GS::LDS_BYTES ? GS::ASYNC_LDS_BYTES : GS::LDS_BYTES) : GS::LDS_BYTES;
  constexpr size_t lds_fused_s = lds_tile > lds_potf2 ? lds_tile : lds_potf2;
  constexpr size_t lds_fused_l = G::LDS_BYTES > lds_potf2 ? G::LDS_BYTES : lds_potf2;
  static const bool fuse_on = exp_int("SVGP_CHOL_FUSE", 1) != 0;   // A/B knob (experiments build)
  set_max_lds(reinterpret_cast<const void*>(potf2_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_potf2));
  set_max_lds(reinterpret_cast<const void*>(syrk128_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(G::LDS_BYTES));
  set_max_lds(reinterpret_cast<const void*>(syrk128_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_fused_l));
  set_max_lds(reinterpret_cast<const void*>(chol_tile_kernel<T, MODE_TRSM, CNT>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_tile));
  set_max_lds(reinterpret_cast<const void*>(chol_tile_kernel<T, MODE_SYRK, CNT>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_tile));
  set_max_lds(reinterpret_cast<const void*>(chol_tile_kernel<T, MODE_SYRK, CNT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_fused_s));
  set_max_lds(reinterpret_cast<const void*>(chol_tile_kernel<T, MODE_SYRK, CNT, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_fused_s));
  // the 512-thread fused form (seven worker waves in the block factorisation) holds one workgroup per CU: taken where the launch fits
  // the chip in one round that way (SVGP_POTF2_WAVES=4 in the experiments build: always the 256-thread form)
  // (`ncus`: the calling context's device - a process-wide static captured the first caller's, wrong for the other members of a
  // single-process svgp_group on unlike devices)
  static const bool p8_on = exp_int("SVGP_POTF2_WAVES", 8) == 8;
  auto potf2 = [&](int p) {
    hipLaunchKernelGGL(potf2_kernel<T>, dim3(1), dim3(kThreads), lds_potf2, s, A + int64_t(p) * kNB * (Mp + 1), Tm + int64_t(p) * kNB * (Mp + 1), Mp,
                       info, p * kNB);
    dbg("potf2", s);
  };
  // The T panels of block row p ride in panel p's TRSM launch while Kuu is small (the launch is latency-bound anyway: M = 1024
  // saves the 25 us of a launch of their own); from 17 panels on they would lengthen 60-odd serial launches instead (C4: +0.33 ms
  // in the TRSM launches against the 0.15 ms of one launch over all 2016 tiles at the end), so a large Kuu keeps the one launch.
  const bool t_inside = nP <= 16;
  // ---- two-level blocking for a large fp32 Kuu (round 4 schedule, round 6 kernel) -----------------------------------------------------
  // Per outer panel (block columns a, b = a + 1):  TRSM(a);  update of block column b alone + factorisation of (b, b) in the same
  // launch;  TRSM(b);  ONE rank-256 update of everything right of b on 256 x 256 tiles (syrk256_kernel: half the bytes per flop of the
  // 128 x 128 x 128 tiles) with the factorisation of the next diagonal block fused in.  Taken pair by pair while the rank-256 launch
  // beats the two rank-128 launches it replaces (it runs one workgroup per CU: whole rounds of `ncus` tiles); the rest of the matrix
  // takes the one-level steps.  Round 4 built this schedule on the 128 x 128 tiles (kb = 2) and found it no faster than one level
  // (M = 8192 4.89 / 5.00 ms, profiles/round4/chol_two_level.md): K only amortises the C tile's read-modify-write, not the operand stream.
  // Round 6, with the 256 x 256 tiles (profiles/round6/c4_chol_256.md; same box, fp32, ms per factorisation, one level / two levels):
  // M = 8192 4.57 / 4.73, 6144 2.74 / 2.87, 4096 1.59 / 1.59 - NOT ADOPTED.  A tile takes 100-110 us for its 57 us of MFMA work (0.55 of the
  // rate: the 512 KiB read-modify-write of C is not hidden behind anything with one 202-VGPR workgroup per CU, ~30 us at the ~18 GB/s
  // a CU draws across the fabric), a launch is whole rounds of 256 tiles (496 tiles: 200 us), and the column step puts a 40-44 us launch
  // (its fused block factorisation is no longer hidden inside a big update) on the chain of every pair: 262 us for the first pair against
  // ~300 us one-level, break-even from the fifth pair on, a loss where the cost rule's optimistic 62 us per round still took them.
  // Experiments build: SVGP_CHOL_TWO_LEVEL=1 (tests/test_gpu_round6.py runs it).
  // the big fused update: whole 128 x 128 tiles on 512 threads (kbb = 2: rank-256, the two-level form)
  auto big_update = [&](int nt, int pp, int kbb) {
    hipLaunchKernelGGL((syrk128_kernel<T, true>), dim3(nt), dim3(kThreads), lds_fused_l, s, A, Tm, Mp, pp, info, kbb);
  };
  // Measured and rejected forms of this schedule, no longer in the tree (profiles/round6/removed_variants.patch): a two-stream look-ahead
  // with the bulk update one panel behind the chain (round 5: bitwise the same factor, 8-15 % slower - two cross-stream dependencies
  // per panel cost more than the idle time they reclaim; profiles/round5/chol_lookahead_ab.log), one launch per panel with a flag
  // hand-over (round 4, chol_chain_kernel: no faster - the TRSM launch already starts the instant the fused launch ends;
  // profiles/round4/chol_chain.md), the big update on the asynchronous three-buffer loop (round 4) and in an XCD-aware tile order
  // (round 5: no gain, profiles/round5/syrk_xcd_ab.log).
  potf2(0);
  auto step1 = [&](int p) {   // one-level panel step p: TRSM (+ T panels), trailing update with the next block factorisation
    const int n = nP - p - 1, nt_p = t_inside ? p : 0;
    // block row p of T (inv(L_pp) from the block factorisation, T[p, J < p] from the TRSM launch) is final behind that launch: the
    // strips' phase 1 of panel p may start (api.hip: SegRun - the row hook below enqueues the waiter on its own stream).  The event
    // rides on the launch itself (hipExtLaunchKernelGGL's stop event = the dispatch packet's own completion signal): a separate
    // hipEventRecord is a barrier packet of its own, ~5 us of the chain per panel (kernel traces, minibatch_step.md section 8).
    // SVGP_ROW_EVENT_EXT=0: the separate record (A/B).
    static const bool ev_ext = exp_int("SVGP_ROW_EVENT_EXT", 1) != 0;   // (experiments build)
    const bool want_ev = row_events && t_inside;
    bool ev_done = false;
    if (n + nt_p > 0) {   // the panel below the diagonal (and the T panels of block row p)
      if (want_ev && ev_ext) {
        hipExtLaunchKernelGGL((chol_tile_kernel<T, MODE_TRSM, CNT>), dim3((n + nt_p) * NCH), dim3(k256), lds_tile, s, nullptr, row_events[p], 0,
                              A, Tm, Mp, p, n, info, sync, 1);
        ev_done = true;
      } else {
        hipLaunchKernelGGL((chol_tile_kernel<T, MODE_TRSM, CNT>), dim3((n + nt_p) * NCH), dim3(k256), lds_tile, s, A, Tm, Mp, p, n, info, sync);
      }
      dbg("chol trsm + T panels", s);
    }
    if (want_ev) {
      if (!ev_done) (void)hipEventRecord(row_events[p], s);
      if (hook && hook->fn) hook->fn(hook->user, p);   // the waiters of this row are enqueued now, not after the whole chain (host time)
    }
    if (n == 0) return;
    const int nt = n * (n + 1) / 2;
    const bool large = nt >= 256;   // a trailing matrix that fills the chip by itself: full 128 x 128 tiles (half the operand traffic)
    // the fused f64 form of the large grid would cost the second resident workgroup (potf2's LDS image of an f64 block is 146 KiB)
    const bool fused = fuse_on && !(large && sizeof(T) == 8);
    if (fused && large) {
      big_update(nt, p, 1);
    } else if (fused) {
      if (p8_on && nt * NCH <= ncus)
        hipLaunchKernelGGL((chol_tile_kernel<T, MODE_SYRK, CNT, true, true>), dim3(nt * NCH), dim3(kThreads), lds_fused_s, s, A, Tm, Mp, p, n, info, sync);
      else
        hipLaunchKernelGGL((chol_tile_kernel<T, MODE_SYRK, CNT, true>), dim3(nt * NCH), dim3(k256), lds_fused_s, s, A, Tm, Mp, p, n, info, sync);
    } else {
      if (large) hipLaunchKernelGGL((syrk128_kernel<T, false>), dim3(nt), dim3(kThreads), G::LDS_BYTES, s, A, Tm, Mp, p, info);
      else hipLaunchKernelGGL((chol_tile_kernel<T, MODE_SYRK, CNT>), dim3(nt * NCH), dim3(k256), lds_tile, s, A, Tm, Mp, p, n, info, sync);
      potf2(p + 1);
    }
    dbg("chol syrk", s);
  };
  int p = 0;
  if constexpr (sizeof(T) == 4 && kExperiments) {   // MEASURED AND NOT ADOPTED (below): the product build takes the one-level steps throughout
    // (read per call: tests and A/B runs change them inside one process)
    const int two_level = exp_int("SVGP_CHOL_TWO_LEVEL", 0);         // 1 = two-level pairs where the cost rule takes them
    const double t256_us = exp_double("SVGP_CHOL_T256_US", 105.0);   // a round of 256 x 256 x 256 tiles (one per CU): measured 100-110 us
    const double t128_us = exp_double("SVGP_CHOL_T128_US", 15.3);    // a round of 128 x 128 x 128 tiles (profiles/round5/c4_syrk_occupancy.log)
    if (two_level && fuse_on && !t_inside && ncus > 0) {
      set_max_lds(reinterpret_cast<const void*>(chol_tile_kernel<T, MODE_COL, CNT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_fused_s));
      constexpr size_t lds256 = kSyrk256LdsBytes > lds_potf2 ? kSyrk256LdsBytes : lds_potf2;
      set_max_lds(reinterpret_cast<const void*>(syrk256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds256));
      while (p + 2 < nP) {
        const int a = p, b = a + 1, na = nP - a - 1, nb = nP - b - 1;
        if (nb % 2 != 0) { step1(p++); continue; }   // (256-row tiles from block row a + 2 on: an even number of block rows)
        const int n2 = nb / 2, t256 = n2 * (n2 + 1) / 2;
        const double cost256 = double((t256 + ncus - 1) / ncus) * t256_us + 15.0;   // + the column launch the pair adds to the chain
        const double cost128 = (double(na) * (na + 1) / 2 + double(nb) * (nb + 1) / 2) * t128_us / double(ncus);
        if (cost256 >= cost128) { step1(p++); continue; }
        hipLaunchKernelGGL((chol_tile_kernel<T, MODE_TRSM, CNT>), dim3(na * NCH), dim3(k256), lds_tile, s, A, Tm, Mp, a, na, info, sync, 1);
        dbg("chol trsm (a)", s);
        // block column b: A[i, b] -= L[i, a] L[b, a]' for i >= b, then the workgroup that completes tile (b, b) factors it
        hipLaunchKernelGGL((chol_tile_kernel<T, MODE_COL, CNT, true>), dim3(na * NCH), dim3(k256), lds_fused_s, s, A, Tm, Mp, a, na, info, sync, 1);
        dbg("chol column update + potf2 (b)", s);
        hipLaunchKernelGGL((chol_tile_kernel<T, MODE_TRSM, CNT>), dim3(nb * NCH), dim3(k256), lds_tile, s, A, Tm, Mp, b, nb, info, sync, 1);
        dbg("chol trsm (b)", s);
        hipLaunchKernelGGL(syrk256_kernel<true>, dim3(t256), dim3(kThreads), lds256, s, reinterpret_cast<float*>(A), reinterpret_cast<float*>(Tm), Mp, b, info);
        dbg("chol rank-256 update + potf2 (next a)", s);
        p += 2;
      }
    }
  }
  for (; p < nP; ++p) step1(p);
  if (!t_inside) {
    hipLaunchKernelGGL((chol_tile_kernel<T, MODE_TRSM, CNT>), dim3((nP * (nP - 1) / 2) * NCH), dim3(k256), lds_tile, s, A, Tm, Mp, -1, 0, info, sync);
    dbg("T panels", s);
  }
}

}  // namespace

// ctx.hpp's error macros (host-only translation units) fetch the pending note through this
std::string take_note_text() { return take_note(); }

#define SVGP_DISPATCH(dtype, expr_d, expr_f) \
  do {                                       \
    if ((dtype) == 0) { expr_d; } else { expr_f; } \
  } while (0)

void launch_scale_inputs(int dtype, hipStream_t s, const void* z, int layout, int d, int64_t M, int64_t Mp,
                         const void* invl, void* zs) {
  dim3 grid((unsigned)((Mp + 255) / 256), (unsigned)d);
  SVGP_DISPATCH(dtype,
                hipLaunchKernelGGL(scale_inputs_kernel<double>, grid, dim3(256), 0, s, (const double*)z, layout, d, M, Mp, (const double*)invl, (double*)zs),
                hipLaunchKernelGGL(scale_inputs_kernel<float>, grid, dim3(256), 0, s, (const float*)z, layout, d, M, Mp, (const float*)invl, (float*)zs));
}

void launch_transpose_colvecs(int dtype, hipStream_t s, const void* x, int d, int64_t n, int64_t ldx, void* out) {
  dim3 grid((unsigned)((n + 255) / 256));
  SVGP_DISPATCH(dtype,
                hipLaunchKernelGGL(transpose_colvecs_kernel<double>, grid, dim3(256), 0, s, (const double*)x, d, n, ldx, (double*)out),
                hipLaunchKernelGGL(transpose_colvecs_kernel<float>, grid, dim3(256), 0, s, (const float*)x, d, n, ldx, (float*)out));
}

void launch_kuu(int dtype, hipStream_t s, const KernelParams& kp, const void* zs, int64_t M, int64_t Mp, double jitter,
                void* Kuu) {
  dim3 grid((unsigned)((Mp + 255) / 256), (unsigned)(Mp / 16));
  if (kp.d <= 8) {
    SVGP_DISPATCH(dtype,
                  hipLaunchKernelGGL((kuu_kernel<double, 8>), grid, dim3(256), 0, s, kp, (const double*)zs, M, Mp, jitter, (double*)Kuu),
                  hipLaunchKernelGGL((kuu_kernel<float, 8>), grid, dim3(256), 0, s, kp, (const float*)zs, M, Mp, float(jitter), (float*)Kuu));
  } else {
    SVGP_DISPATCH(dtype,
                  hipLaunchKernelGGL((kuu_kernel<double, 0>), grid, dim3(256), 0, s, kp, (const double*)zs, M, Mp, jitter, (double*)Kuu),
                  hipLaunchKernelGGL((kuu_kernel<float, 0>), grid, dim3(256), 0, s, kp, (const float*)zs, M, Mp, float(jitter), (float*)Kuu));
  }
}

// T must be zero above the diagonal on entry (the model's buffer is cleared once at creation): the factorisation writes the
// lower triangles of the inverted diagonal blocks and the T panels below them only
void launch_potrf(int dtype, hipStream_t s, void* A, void* T, int64_t Mp, int* info, unsigned* sync, int num_cus, hipEvent_t* row_events,
                  const RowHook* hook) {
  SVGP_DISPATCH(dtype, potrf_t<double>(s, (double*)A, (double*)T, Mp, info, sync, row_events, hook, num_cus),
                potrf_t<float>(s, (float*)A, (float*)T, Mp, info, sync, row_events, hook, num_cus));
}
int potrf_max_row_events() { return 16; }   // block rows of T are final one by one only while they ride in the TRSM launches (nP <= 16)

void launch_pack_q_ld(int dtype, hipStream_t s, const void* Lq, int64_t ldq, const void* m, int64_t M, int64_t Mp, void* U,
                       void* mp) {
  dim3 grid((unsigned)(Mp / 32), (unsigned)(Mp / 32)), block(32, 8);
  SVGP_DISPATCH(dtype,
                hipLaunchKernelGGL(pack_q_kernel<double>, grid, block, 0, s, (const double*)Lq, ldq, (const double*)m, M, Mp, (double*)U, (double*)mp),
                hipLaunchKernelGGL(pack_q_kernel<float>, grid, block, 0, s, (const float*)Lq, ldq, (const float*)m, M, Mp, (float*)U, (float*)mp));
}

void launch_pack_q(int dtype, hipStream_t s, const void* Lq, const void* m, int64_t M, int64_t Mp, void* U, void* mp) {
  launch_pack_q_ld(dtype, s, Lq, M, m, M, Mp, U, mp);
}

void launch_kl_terms_ld(int dtype, hipStream_t s, const void* Lq, int64_t ldq, const void* m, const void* Lk, int64_t M,
                        int64_t Mp, double* scal) {
  double* colsq = scal + 8;  // caller provides 8 + M doubles
  if (dtype == 0) {
    hipLaunchKernelGGL(kl_colsq_kernel<double>, dim3((unsigned)M), dim3(k256), 0, s, (const double*)Lq, ldq, M, colsq);
    hipLaunchKernelGGL(kl_final_kernel<double>, dim3(1), dim3(k256), 0, s, (const double*)Lq, ldq, (const double*)m, (const double*)Lk, M, Mp, colsq, scal);
  } else {
    hipLaunchKernelGGL(kl_colsq_kernel<float>, dim3((unsigned)M), dim3(k256), 0, s, (const float*)Lq, ldq, M, colsq);
    hipLaunchKernelGGL(kl_final_kernel<float>, dim3(1), dim3(k256), 0, s, (const float*)Lq, ldq, (const float*)m, (const float*)Lk, M, Mp, colsq, scal);
  }
}

void launch_kl_terms(int dtype, hipStream_t s, const void* Lq, const void* m, const void* Lk, int64_t M, int64_t Mp,
                     double* scal) {
  launch_kl_terms_ld(dtype, s, Lq, M, m, Lk, M, Mp, scal);
}

void launch_extract_lower(int dtype, hipStream_t s, const void* A, int64_t Mp, int64_t M, void* out) {
  dim3 grid((unsigned)((M + 255) / 256), (unsigned)M);
  SVGP_DISPATCH(dtype,
                hipLaunchKernelGGL(extract_lower_kernel<double>, grid, dim3(256), 0, s, (const double*)A, Mp, M, (double*)out),
                hipLaunchKernelGGL(extract_lower_kernel<float>, grid, dim3(256), 0, s, (const float*)A, Mp, M, (float*)out));
}

void launch_trsv2(int dtype, hipStream_t s, const void* L, const void* Tm, int64_t Mp, int trans, void* x) {
  const size_t lds = size_t(Mp + kNB) * (dtype == 0 ? 8 : 4);
  SVGP_DISPATCH(dtype,
                hipLaunchKernelGGL(trsv_kernel<double>, dim3(1), dim3(k256), lds, s, (const double*)L, (const double*)Tm, Mp, trans, (double*)x),
                hipLaunchKernelGGL(trsv_kernel<float>, dim3(1), dim3(k256), lds, s, (const float*)L, (const float*)Tm, Mp, trans, (float*)x));
}

void launch_trsm_mat(int dtype, hipStream_t s, const void* Tm, int64_t Mp, void* X) {
  using Gd = TileGemm<double, kNB, 16>;
  using Gf = TileGemm<float, kNB, 16>;
  set_max_lds(reinterpret_cast<const void*>(trsm_mat_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, int(Gd::LDS_BYTES));
  set_max_lds(reinterpret_cast<const void*>(trsm_mat_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, int(Gf::LDS_BYTES));
  dim3 grid((unsigned)(Mp / kNB));
  SVGP_DISPATCH(dtype,
                hipLaunchKernelGGL(trsm_mat_kernel<double>, grid, dim3(kThreads), Gd::LDS_BYTES, s, (const double*)Tm, (double*)X, Mp),
                hipLaunchKernelGGL(trsm_mat_kernel<float>, grid, dim3(kThreads), Gf::LDS_BYTES, s, (const float*)Tm, (float*)X, Mp));
}

void launch_pad_lower(int dtype, hipStream_t s, const void* Lq, int64_t M, int64_t Mp, void* out) {
  dim3 grid((unsigned)((Mp + 255) / 256), (unsigned)Mp);
  SVGP_DISPATCH(dtype,
                hipLaunchKernelGGL(pad_lower_kernel<double>, grid, dim3(256), 0, s, (const double*)Lq, M, Mp, (double*)out),
                hipLaunchKernelGGL(pad_lower_kernel<float>, grid, dim3(256), 0, s, (const float*)Lq, M, Mp, (float*)out));
}

void launch_shift_vec(int dtype, hipStream_t s, const void* m, double shift, int64_t M, int64_t Mp, void* out) {
  dim3 grid((unsigned)((Mp + 255) / 256));
  SVGP_DISPATCH(dtype,
                hipLaunchKernelGGL(shift_vec_kernel<double>, grid, dim3(256), 0, s, (const double*)m, shift, M, Mp, (double*)out),
                hipLaunchKernelGGL(shift_vec_kernel<float>, grid, dim3(256), 0, s, (const float*)m, float(shift), M, Mp, (float*)out));
}

void launch_cov_assemble(int dtype, hipStream_t s, const KernelParams& kp, const void* xa, int64_t ldxa, int64_t na,
                         const void* xb, int64_t ldxb, int64_t nb, const void* Aa, const void* Ca, int64_t lda,
                         const void* Ab, const void* Cb, int64_t ldb, int64_t Mp, void* out) {
  using Gd = TileGemm<double, kNB, 16>;
  using Gf = TileGemm<float, kNB, 16>;
  set_max_lds(reinterpret_cast<const void*>(cov_assemble_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, int(Gd::LDS_BYTES));
  set_max_lds(reinterpret_cast<const void*>(cov_assemble_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, int(Gf::LDS_BYTES));
  dim3 grid((unsigned)((na + kNB - 1) / kNB), (unsigned)((nb + kNB - 1) / kNB));
  SVGP_DISPATCH(dtype,
                hipLaunchKernelGGL(cov_assemble_kernel<double>, grid, dim3(kThreads), Gd::LDS_BYTES, s, kp, (const double*)xa, ldxa, na, (const double*)xb, ldxb, nb, (const double*)Aa, (const double*)Ca, lda, (const double*)Ab, (const double*)Cb, ldb, Mp, (double*)out),
                hipLaunchKernelGGL(cov_assemble_kernel<float>, grid, dim3(kThreads), Gf::LDS_BYTES, s, kp, (const float*)xa, ldxa, na, (const float*)xb, ldxb, nb, (const float*)Aa, (const float*)Ca, lda, (const float*)Ab, (const float*)Cb, ldb, Mp, (float*)out));
}

}  // namespace svgp
