// grad.hip — reverse-mode gradient of the ELBO (the reference differentiates elbo with Zygote:
// examples/a-regression/script.jl:188-194, test/SparseVariationalApproximationModule.jl:170-175): the kernels of the parts
// that are NOT the strip kernel's phase 3.  Hand-derived adjoint of the forward path (oracle/svgp_oracle.py: elbo_grad is
// the same derivation; api.hip: grad_enqueue is the schedule).  With A = Lk \ Kuf, B the whitened factor, W = A diag(2 g_v) A':
//   data-sized     W (SYRK over the points, split-K slices)                         gemm_pm_kernel
//                  kernel-parameter / inducing-input reductions of P o dK           kgrad_mfma_kernel
//                  (P = alpha g_mu' + 2 (R A) diag(g_v) is formed there from the strips' UNSCALED R A: the strips take the
//                  variance from the same product, so their phase 2 is gone - 4 GEMM units per point in all: trsm 1 + R A 2 + W 1)
//   M-sized        Linv = Lk^-1 by recursive doubling (round 3)                     linv_init / linv_step kernels
//                  alpha = Linv' m~,  R = Linv' (B B' - I)                          linv_t_gemv, gemm_pm (M x M x M form)
//                  Lq_bar = tril(W B) - dKL/dB,  Lk_bar = -tril(alpha a' + R W)     gemm_pm, finish_mm2
//                  Kuu_bar = sym(Linv' Phi(Lk' Lk_bar) Linv)                        gemm_pm x 3, phi, symmetrize
//                  their Kuu part of the kernel-parameter gradients                 kgrad_mfma_kernel (uu), kgrad_reduce, finish_kgrad
// Round 2 applied Lk^-T by blocked substitution (four chains of nP panels, 0.18 ms each at M = 1024 whatever the batch);
// with the explicit inverse every M-sized step is a GEMM over the whole chip.
#include <cstdlib>

#include "device_common.hpp"
#include "kernels.hpp"
#include "knobs.hpp"
#include "lik.hpp"

namespace svgp {
namespace {

__global__ void set_f64_kernel(double* dst, double value) { *dst = value; }
__global__ void set2_f64_kernel(double* dst, double a, double b) { dst[0] = a; dst[1] = b; }
// status slots of the (all-reduced) gradient scalars, and - behind the all-reduced region - this rank's own prep scalars
// {sum Lq^2, m'm, sum log diag Lq, sum log diag Lk} and chol_info, so that one copy brings everything fp64 to the host
__global__ void grad_status_kernel(double* sums, const int* chol_info, double n_points, const double* prep_scal, double* prep_out) {
  sums[5] = n_points;
  sums[6] = (chol_info && *chol_info != 0) ? 1.0 : 0.0;
  sums[7] = 0.0;
  if (prep_out) {
    for (int q = 0; q < 4; ++q) prep_out[q] = prep_scal[q];
    prep_out[4] = chol_info ? double(*chol_info) : 0.0;
  }
}

// out[q] += sum over blocks of partial[b][q]: wave q sums its column (lane l takes blocks l, l + 64, ... in order, then a
// fixed shuffle tree), so the result does not depend on anything but the inputs (a single thread walking 1024 strips
// cost 170 us per chunk: 2.7 ms of an H evaluation)
__global__ void __launch_bounds__(320) sum5_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
  const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0.0;
  for (int b = lane; b < nblocks; b += 64) s += partial[b * 5 + q];
  for (int w = 32; w > 0; w >>= 1) s += __shfl_down(s, w);
  if (lane == 0) out[q] += s;
}

// ---- M x M helpers of the fused gradient path (svgp_elbo_grad, api.hip: grad_enqueue) ---------------------------------
// out (row-major, FULL symmetric) = sum over slices of the lower triangle of G (row-major; only entries c <= r are read, also
// inside the diagonal tiles, so the result is exactly symmetric), minus `eye` on the diagonal.  One 32 x 32 tile of the lower
// triangle per workgroup: the slice sums are read along rows, the mirror image leaves through an LDS transposition, so both
// the reads and the two writes run along rows (round 2 read the mirror half with stride Mp: 58 us for 7 slices at M = 1024).
template <typename T>
__global__ void __launch_bounds__(k256) sym_from_lower_kernel(const T* __restrict__ G, int nslices, int64_t Mp, T eye, T* __restrict__ out) {
  __shared__ T tile[32][33];
  int br = 0, b = blockIdx.x;
  while (b >= br + 1) { b -= br + 1; ++br; }
  const int bc = b;                                   // bc <= br
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int q = ty; q < 32; q += 8) {
    const int64_t r = int64_t(br) * 32 + q, c = int64_t(bc) * 32 + tx;
    T v = T(0);
    if (c <= r) {
      for (int s = 0; s < nslices; ++s) v += G[int64_t(s) * Mp * Mp + r * Mp + c];
      if (r == c) v -= eye;
      out[r * Mp + c] = v;
    }
    tile[q][tx] = v;
  }
  __syncthreads();
  for (int q = ty; q < 32; q += 8) {
    // mirror: element (c', r') of the output with c' in the tile's columns, r' in its rows = tile[r' - r0][c' - c0]
    const int64_t cp = int64_t(bc) * 32 + q, rp = int64_t(br) * 32 + tx;
    if (cp < rp) out[cp * Mp + rp] = tile[tx][q];
  }
}

// avec[c] = sum over slices of rowpart[s][1][c]  ( = (A g_mu)_c or (Kuf g_mu)_c, the data part of m_bar ): 64 columns per
// workgroup, thread (c, g) sums slices g, g + 4, ... in order, the four groups are combined in a fixed order (one thread per
// column walking 256 slices ran at the memory latency: 35 us)
__global__ void __launch_bounds__(k256) avec_kernel(const double* __restrict__ rp_uf, int ns, int64_t stride, int64_t Mp, double* __restrict__ avec) {
  __shared__ double sh[4][64];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = int64_t(blockIdx.x) * 64 + cl;
  double s = 0.0;
  if (i < Mp)
    for (int q = g; q < ns; q += 4) s += rp_uf[q * stride + Mp + i];
  sh[g][cl] = s;
  __syncthreads();
  if (g == 0 && i < Mp) avec[i] = ((sh[0][cl] + sh[1][cl]) + sh[2][cl]) + sh[3][cl];
}

// Lq_bar = tril(G1) - klw dKL/dLq  with G1 = 2 W Lq (row-major);  Lk_bar = -tril(G2 + alpha a')  with G2 = 2 R W (row-major)
template <typename T>
__global__ void finish_mm2_kernel(const T* __restrict__ G1, const T* __restrict__ G2, const T* __restrict__ alpha,
                                  const double* __restrict__ avec, int64_t Mp, int64_t M, const T* __restrict__ Lq, int64_t ldq,
                                  T klw, T* __restrict__ Lq_bar, T* __restrict__ BbarRM, T* __restrict__ LkbarRM) {
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= Mp) return;
  const bool low = c <= r;
  LkbarRM[r * Mp + c] = low ? -(G2[r * Mp + c] + alpha[r] * T(avec[c])) : T(0);
  T v = T(0);
  if (r < M && low) {
    const T l = Lq[r + c * ldq];
    v = G1[r * Mp + c] - klw * (c == r ? l - T(1) / l : l);   // klw * d KL / d Lq
  }
  if (BbarRM) BbarRM[r * Mp + c] = v;                    // Centered: adjoint of B = Lk \\ Lq, row-major for the solve
  else if (r < M && c < M) Lq_bar[r + c * M] = v;
}

// out[slice][r][c] += sum_{i in slice} w_i Xt[i][r] Yt[i][c]   (Xt, Yt point-major [n][Mp]; lower tiles only)
// NT = 128 / 512 threads: one workgroup per 128 x 128 output tile; NT = 64 / 256 threads: two 128 x 64 halves on
// separate workgroups (the strip kernel's geometry: the two waves of a SIMD belong to different workgroups)
// flags (round 3, the M x M x M products of the gradient's tail): kMmFull - all nP x nP tiles instead of the lower ones;
// kMmXLow / kMmYLow - Xt[i][r] (Yt[i][c]) vanishes for i above the diagonal tile of r (c), i.e. the contraction starts at that
// tile; kMmXUp / kMmYUp - it vanishes BELOW that tile, the contraction ends there (triangular operands: half the k-steps)
template <typename T, int NT, int NTHR>
__global__ void __launch_bounds__(NTHR, 2) gemm_pm_kernel(const T* __restrict__ Xt, const T* __restrict__ Yt,
                                                           const T* __restrict__ w, T wscale, int64_t Mp, int64_t n,
                                                           int64_t slice_len, T* __restrict__ out, int overwrite, int flags) {
  using G = TileGemm<T, NT, 16, NTHR>;
  using QRegs = typename G::QRegs;
  constexpr int NCH = kNB / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int item = blockIdx.x, slice = blockIdx.y;   // work item = (tile, K slice)
  const int chunk = item % NCH;
  int ti = 0, tj, b = item / NCH;
  if (flags & kMmFull) {
    const int nP = int(Mp / kNB);
    ti = b / nP;
    tj = b % nP;
  } else {
    while (b >= ti + 1) { b -= ti + 1; ++ti; }
    tj = b;
  }
  int64_t i0 = int64_t(slice) * slice_len;
  int64_t i1 = i0 + slice_len;
  i1 = i1 < n ? i1 : n;
  if (flags & (kMmXLow | kMmYLow | kMmXUp | kMmYUp)) {
    int64_t lo = 0, hi = n;
    if (flags & kMmXLow) lo = int64_t(ti) * kNB;
    if ((flags & kMmYLow) && int64_t(tj) * kNB > lo) lo = int64_t(tj) * kNB;
    if (flags & kMmXUp) hi = int64_t(ti + 1) * kNB;
    if ((flags & kMmYUp) && int64_t(tj + 1) * kNB < hi) hi = int64_t(tj + 1) * kNB;
    i0 = i0 > lo ? i0 : lo;
    i1 = i1 < hi ? i1 : hi;
  }
  typename G::Acc acc;
  acc.zero();
  if (i1 > i0) {
    const typename G::QOff qoff = G::q_offsets(Mp);
    const T* yq = Yt + i0 * Mp + int64_t(tj) * kNB + chunk * NT;
    auto qload = [&](int t, QRegs& r) {
      G::load_q(r, yq + int64_t(t) * 16 * Mp, qoff);
      if (w) {
#pragma unroll
        for (int p = 0; p < G::Q_PASSES; ++p) {
          int kk, c;
          G::q_coord(p, kk, c);
          const T wk = w[i0 + int64_t(t) * 16 + kk] * wscale;
#pragma unroll
          for (int e = 0; e < G::VEC; ++e) r.v[p][e] *= wk;
        }
      }
    };
    G::loop(acc, Xt + i0 * Mp + int64_t(ti) * kNB, Mp, int((i1 - i0) / 16), qload, smem);
  }
  T* o = out + int64_t(slice) * Mp * Mp + int64_t(ti) * kNB * Mp + int64_t(tj) * kNB + chunk * NT;
#pragma unroll
  for (int i = 0; i < G::MI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < G::NJ; ++j) {
        T* e = o + int64_t(G::acc_row(i, r)) * Mp + G::acc_col(j);
        *e = overwrite ? acc.v[i][j][r] : *e + acc.v[i][j][r];
      }
}

template <typename T>
struct StepWeights {   // the 16 weights of k-step t
  static constexpr bool on = true;
  const T* w0;
  __device__ __forceinline__ const T* operator()(int t) const { return w0 + t * 16; }
};
// The data-sized SYRK on the fully asynchronous loop (device_common.hpp: loop_tri_async_w; round 3): both operand tiles and
// the 16 weights of a step travel global -> LDS by DMA through three buffers, the weights scale the Q fragments as they are
// read (2 v_mul per 8 MFMAs in f64).  Same work items, same partial sums layout as gemm_pm_kernel; lower tiles only.
// UW (round 4): UNIFORM weights - every point of the chunk carries the same weight (the Gaussian likelihood: g_v = -scale / (2 sigma^2)
// whatever the point), so W = w A A' and the loop is the unweighted one; the weight meets the accumulators once, in the epilogue.
// The per-k weights cost the f64 kernel 7 % (8 v_mul_f64 per 32 MFMAs on a pipe that does not co-execute them: rocprofv3, H, 1267 ->
// 1174 us per 65 536-point chunk).  `uw` = {weight without the data scale, data scale or 0, num_data}: with a device-resident batch
// size (collective calls) the scale is num_data / *n_global_dev, read here.  The caller zeroes the rows of At between the chunk's
// last point and the end of its last 16-point k-step (the weighted form met them with zero weights).
struct UniformW { double w; double scale; double num_data; const double* n_global_dev; };
template <typename T, int NT, bool UW = false>
__global__ void __launch_bounds__(k256, 2) syrk_async_kernel(const T* __restrict__ At, const T* __restrict__ w, T wscale, int64_t Mp, int64_t n,
                                                             int64_t slice_len, T* __restrict__ out, int overwrite, UniformW uw) {
  using G = TileGemm<T, NT, 16, k256>;
  static_assert(G::kAsync, "tile shape without an asynchronous loop");
  constexpr int NCH = kNB / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int chunk = blockIdx.x % NCH;
  int ti = 0, tj, b = blockIdx.x / NCH;
  while (b >= ti + 1) { b -= ti + 1; ++ti; }
  tj = b;
  const int64_t i0 = int64_t(blockIdx.y) * slice_len;
  int64_t i1 = i0 + slice_len;
  i1 = i1 < n ? i1 : n;
  typename G::Acc acc;
  acc.zero();
  if (i1 > i0) {
    const T* yq = At + i0 * Mp + int64_t(tj) * kNB + chunk * NT;
    const T* w0 = w + i0;
    auto qsrc = [&](int t) { return yq + int64_t(t) * 16 * Mp; };
    if constexpr (UW) {
      (void)w0;
      G::template loop_tri_async<0>(acc, At + i0 * Mp + int64_t(ti) * kNB, Mp, int((i1 - i0 + 15) / 16), qsrc, smem, Mp);
    } else {
      StepWeights<T> wsrc{w0};
      G::template loop_tri_async_w<0>(acc, At + i0 * Mp + int64_t(ti) * kNB, Mp, int((i1 - i0) / 16), qsrc, wsrc, smem, Mp);
    }
  }
  if constexpr (UW) {
    const double sc = uw.n_global_dev ? (uw.num_data > 0.0 ? uw.num_data / *uw.n_global_dev : 1.0) : uw.scale;
    wscale = T(double(wscale) * uw.w * sc);
  }
  T* o = out + int64_t(blockIdx.y) * Mp * Mp + int64_t(ti) * kNB * Mp + int64_t(tj) * kNB + chunk * NT;
#pragma unroll
  for (int i = 0; i < G::MI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < G::NJ; ++j) {
        T* e = o + int64_t(G::acc_row(i, r)) * Mp + G::acc_col(j);
        const T v = wscale * acc.v[i][j][r];   // (the gradient's factor 2: exact)
        *e = overwrite ? v : *e + v;
      }
}

// out[r][c] (+)= sum_s part[s][r][c] over the lower 128-tiles (the tiles gemm_pm_kernel computes) or over all of them,
// slices in a fixed order
template <typename T>
__global__ void sum_slices_lower_kernel(const T* __restrict__ part, int ns, int64_t Mp, T* __restrict__ out, int full, int overwrite) {
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= Mp || (!full && c / kNB > r / kNB)) return;
  T v = T(0);
  for (int s = 0; s < ns; ++s) v += part[int64_t(s) * Mp * Mp + r * Mp + c];
  out[r * Mp + c] = overwrite ? v : out[r * Mp + c] + v;
}

// ---- explicit inverse of the Cholesky factor (round 3) ---------------------------------------------------------------
// The gradient's tail applied Lk^-T four times by blocked substitution (solve_t_kernel): M / 32 workgroups, each a chain of
// nP panels at the per-k-step latency - 4 x 0.18 ms of a 3.8 ms training step at M = 1024.  With Linv = Lk^-1 in hand every
// one of them is a plain (triangular-operand) GEMM over all the chip.  Linv is built by recursive doubling from the 128-blocks
// potf2 already inverted: for adjacent block ranges lo, hi of s panels,
//     Linv[hi, lo] = -Linv[hi, hi] (L[hi, lo] Linv[lo, lo])
// - two launches per level, log2(nP) levels, every tile of a level in parallel.  Both storage orders are kept: LinvRM
// (row-major: the k-major operand "Xt[k][r] = Linv[k][r]" of the products Linv' X) and LinvCM (column-major: the left
// operand of step 2).  Numerics: as the blocked substitution it replaces, the products carry a forward error of order
// eps cond(Lk); the gradient parity tests bound the end result.
template <typename T>
__global__ void linv_init_kernel(const T* __restrict__ Tm, int64_t Mp, T* __restrict__ LinvRM, T* __restrict__ LinvCM) {
  const int64_t o = int64_t(blockIdx.y) * kNB * (Mp + 1);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kNB * kNB; e += gridDim.x * blockDim.x) {
    const int r = e % kNB, c = e / kNB;
    const T v = Tm[o + r + int64_t(c) * Mp];   // inv(L_II)[r][c], zero above the diagonal
    LinvCM[o + r + int64_t(c) * Mp] = v;
    LinvRM[o + c + int64_t(r) * Mp] = v;
  }
}

template <typename T, int NT, int STEP>
__global__ void __launch_bounds__(k256, 2) linv_step_kernel(const T* __restrict__ L, T* __restrict__ LinvRM, T* __restrict__ LinvCM,
                                                            T* __restrict__ Ytmp, int64_t Mp, int nP, int s) {
  using G = TileGemm<T, NT, 16, k256>;
  using QRegs = typename G::QRegs;
  constexpr int NB = kNB, NCH = NB / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const int a = int(blockIdx.y) * 2 * s, hi0 = a + s;
  const int hn = (nP - hi0) < s ? (nP - hi0) : s;
  const int chunk = blockIdx.x % NCH, tile = blockIdx.x / NCH;
  const int tr = tile / s, tc = tile % s;
  if (tr >= hn) return;
  const int64_t row0 = int64_t(hi0 + tr) * NB, col0 = int64_t(a + tc) * NB + chunk * NT;
  typename G::Acc acc;
  acc.zero();
  const typename G::QOff qoff = G::q_offsets(Mp);
  if (STEP == 1) {   // Ytmp[hi, lo] = L[hi, lo] Linv[lo, lo]: contraction over the lo rows j >= the column's diagonal tile
    const T* P = L + row0 + (int64_t(a + tc) * NB) * Mp;                 // element (k, row) at P[k * Mp + row]
    const T* Q = LinvRM + (int64_t(a + tc) * NB) * Mp + col0;            // element (k, col) at Q[k * Mp + col]
    auto qload = [&](int t, QRegs& r) { G::load_q(r, Q + int64_t(t) * 16 * Mp, qoff); };
    G::loop(acc, P, Mp, (s - tc) * (NB / 16), qload, smem);
  } else {           // Linv[hi, lo] = -Linv[hi, hi] Ytmp[hi, lo]: contraction over the hi columns k <= the row's diagonal tile
    const T* P = LinvCM + row0 + (int64_t(hi0) * NB) * Mp;
    const T* Q = Ytmp + (int64_t(hi0) * NB) * Mp + col0;
    auto qload = [&](int t, QRegs& r) { G::load_q(r, Q + int64_t(t) * 16 * Mp, qoff); };
    G::loop(acc, P, Mp, (tr + 1) * (NB / 16), qload, smem);
  }
#pragma unroll
  for (int i = 0; i < G::MI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < G::NJ; ++j) {
        const int64_t row = row0 + G::acc_row(i, r), col = col0 + G::acc_col(j);
        if (STEP == 1) {
          Ytmp[row * Mp + col] = acc.v[i][j][r];
        } else {
          LinvRM[row * Mp + col] = -acc.v[i][j][r];
          LinvCM[row + col * Mp] = -acc.v[i][j][r];
        }
      }
}

// out[r] = sum_{k >= r} LinvRM[k][r] v[k]  ( = (Lk^-T v)_r ), fp64 accumulation, in two stages: workgroup (bx, by) sums the 128
// rows k of panel by for its 64 columns r (thread (r, g) takes k = g, g + 4, ...: a row of LinvRM is read contiguously, the
// loads of a thread are independent), the panel partials are then added in a fixed order.  (One stage with a serial loop over
// all k per thread ran at the L2 latency: 88 us at M = 1024.)
template <typename T, typename VT>
__global__ void __launch_bounds__(k256) linv_t_gemv_kernel(const T* __restrict__ LinvRM, const VT* __restrict__ v, int64_t Mp,
                                                           double* __restrict__ part, int notrans) {
  // notrans: the same sum over the COLUMN-major copy, restricted to k <= r:  out[r] = sum_{k <= r} Linv[r][k] v[k] = (Lk^-1 v)_r
  __shared__ double sh[4][64];
  const int rl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t r = int64_t(blockIdx.x) * 64 + rl, k0 = int64_t(blockIdx.y) * kNB;
  double acc = 0.0;
  if (notrans ? (k0 < int64_t(blockIdx.x + 1) * 64) : (k0 + kNB > int64_t(blockIdx.x) * 64)) {   // panels wholly on the zero side contribute nothing
#pragma unroll 8
    for (int kk = g; kk < kNB; kk += 4) {
      const int64_t k = k0 + kk;
      const double a = (notrans ? k <= r : k >= r) ? double(LinvRM[k * Mp + r]) : 0.0;
      acc = fma(a, double(v[k]), acc);
    }
  }
  sh[g][rl] = acc;
  __syncthreads();
  if (g == 0) part[int64_t(blockIdx.y) * Mp + r] = ((sh[0][rl] + sh[1][rl]) + sh[2][rl]) + sh[3][rl];
}
template <typename T>
__global__ void gemv_finish_kernel(const double* __restrict__ part, int np, int64_t Mp, T* __restrict__ out) {
  const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= Mp) return;
  double s = 0.0;
  for (int q = 0; q < np; ++q) s += part[int64_t(q) * Mp + r];
  out[r] = T(s);
}

// kernel value and its derivative w.r.t. r^2 (both including the variance)
template <typename T, int FAMILY>
__device__ __forceinline__ void kappa_and_d(T r2, T variance, T& k, T& dk) {
  if (FAMILY == KSE) {
    k = variance * kexp(T(-0.5) * r2);
    dk = T(-0.5) * k;
  } else if (FAMILY == KM32) {
    const T s = T(1.7320508075688772935) * ksqrt(r2);
    const T ex = variance * kexp(-s);
    k = (T(1) + s) * ex;
    dk = T(-1.5) * ex;
  } else {
    const T s = T(2.2360679774997896964) * ksqrt(r2);
    const T ex = variance * kexp(-s);
    k = (T(1) + s + T(5.0 / 3.0) * r2) * ex;
    dk = T(-5.0 / 6.0) * (T(1) + s) * ex;
  }
}

// Row-wise reductions of W = Pt o dK/dr^2 against the points of a slice (the interface of launch_kgrad):
//   rowpart[slice][0][i] += sum_j W_ij           rowpart[slice][1][i] += sum_j K_ij g_mu_j   (m_bar: the caller applies Lk^-1)
//   rowpart[slice][2+f][i] += sum_j W_ij xs_fj    scalpart[slice][rb][0] += sum P_ij K_ij,  [1+f] += sum W_ij u_fij^2
// Pt is point-major [n][Mp].  Rounds 2-5 computed them on the VALU - a lane owned one or two rows with all d feature slots in
// registers (kgrad_kernel), two / four lanes a row (kgrad_wide_kernel), a wave a 16-feature group (kgrad_wide2_kernel): 75 (d = 8)
// to 450 (d = 64) VALU instructions per 64 entries, 219 us per 65 536-point chunk at d = 8 and 2.9 ms at d = 64 (M = 1024, f64).
// Those kernels left the tree in round 6 (profiles/round6/kgrad_valu_kernels_removed.patch).
// ---- the kernel-gradient reductions on the MFMA (round 6) ----------------------------------------------------
// Both data-sized contractions are MFMAs and the VALU is left the kernel function itself:
//   (1) r2 tile (16 points x 16 inducing rows) = |x|^2 + |z|^2 - 2 x.z on the MFMA (accumulator preloaded with the norms, as the
//       strips' pre-generation and kuf_cols_kernel do): D[point slot][inducing row];
//   (2) VALU, 4 entries per lane: kernel function and its derivative, P = alpha g_mu' + 2 (R A) diag(g_v), W = P o dK/dr2, the row
//       sums R_i = sum_j W_ij, (Kuf g_mu)_i, sum P o K;
//   (3) Q (16 rows x 16 features) += W (16 rows x 4 points) X (4 points x 16 features): register r of the D tile of (1), as it
//       stands, IS the A operand of a 16x16x4 MFMA whose four k-slots are the points {g + 4 r} (g = lane / 16) - W never moves
//       between lanes - and the B operand is read from the staged x tile at those points.
// The lengthscale sums IL_f = sum_ij W_ij (z_fi - x_fj)^2 are not accumulated entry by entry (4 d VALU instructions per entry):
//     IL_f = sum_i (z_fi^2 R_i - 2 z_fi Q_fi) + sum_j x_fj^2 C_j,        C_j = sum_i W_ij  (over the workgroup's rows: linear, so
// per-workgroup partials add up), C_j by a DPP butterfly over the 16 lanes of a row group.  d <= 8: the eight spare columns of the one
// feature tile carry x_f^2 instead and give sum_j W_ij x_fj^2 directly.  The expansion cancels where |z - x| << |z|: relative error
// eps (|z|^2 + |x|^2) / |z - x|^2 in units of the scaled inputs - the same exposure as the MFMA distances themselves - so both x and
// z are taken RELATIVE TO A CENTRE c: distances and IL do not change, |z|, |x| become the spread of the data instead of its distance
// from the origin, and the row sums leave as Q = Q' + c R (fp64).  (Found by the one-point, one-inducing-point case of
// test_gradient_degenerate_shapes: z = x + 1e-3, where the fp32 build lost IL altogether.)  The centre of a feature is the MIDDLE OF THE
// RANGE of the workgroup's 64 inducing rows (the valid ones).  What is left of the cancellation in IL_f is (|z_fi - c_f| / lengthscale)^2
// times the fp32 rounding of the accumulated sums: with many inducing points per lengthscale (d = 1, M >= 128 over 8 lengthscales) that was
// 10-45 x the error of the entry-by-entry VALU sums of rounds 2-5 on the fp32 lengthscale gradient (profiles/round6/fuzz_grad.md).  So for
// d <= 4 the SPARE FEATURE SLOTS of the one feature tile carry the same feature about further centres - 8 / d centres per feature, evenly
// spaced over the rows' range - and every inducing row takes IL_f and sum_j W_ij x_fj from the slot whose centre is nearest to it:
// |z_fi - c| <= range / (2 * 8 / d), at no cost in the loop (the MFMAs of the slots were issued anyway, on zeros).
// A wave owns 16 inducing rows for the whole slice of points (no cross-wave sums but the scalars), a workgroup 64 rows; the staged
// tile is point-major with row stride DL + 16 bytes... (+6 / +4 elements): the r2 operand reads are conflict-free, and the pad columns
// carry g_mu, 2 g_v and c1 |x|^2 of the point.  fp32: per staged block (128 points) sums in fp32 (MFMA accumulators included), totals fp64.
// Rows per workgroup: 64 for every d.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row, left in every lane: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
template <typename T>
__device__ __forceinline__ T row16_sum(T v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

template <typename T, int FAMILY, int DL>
__global__ void __launch_bounds__(k256, 2) kgrad_mfma_kernel(KernelParams kp, const T* __restrict__ zs, int64_t Mp,
                                                             const T* __restrict__ x, int64_t ldx, int64_t xoff, int prescaled,
                                                             int64_t n, int64_t nvalid, const T* __restrict__ Pt,
                                                             const T* __restrict__ gmu, const T* __restrict__ gv,
                                                             const T* __restrict__ alpha, int64_t slice_len,
                                                             double* __restrict__ rowpart, double* __restrict__ scalpart, int kmb, int64_t M) {
  using M16 = Mfma16<T>;
  using acc_t = typename M16::acc_t;
  constexpr bool kF64 = (sizeof(T) == 8);
  constexpr bool SQ = (DL == 8);                    // x_f^2 in the spare columns 8..15 of the feature tile
  constexpr int JB = 128, KS = DL / 4, CT = (DL + 15) / 16, NTILE = JB / 16;
  constexpr int NQ = (CT <= 2) ? 2 : 1;             // independent accumulator sets of step (3) (a lone tile would be a chain of dependent MFMAs)
  constexpr int XLD = DL + (kF64 ? 6 : 4);          // f64: 16 rows x two k-slots on 32 distinct 8-byte banks; fp32: 16 x 4 on 64 banks
  constexpr int GM = DL, GV = DL + 1, XN = DL + 2;  // pad columns of a staged point
  constexpr T c1 = (FAMILY == KSE) ? T(-0.5) : T(1);      // SE: the tile holds -r2 / 2, the argument of exp
  constexpr T ascale = (FAMILY == KSE) ? T(1) : T(-2);
  __shared__ __attribute__((aligned(16))) T xt[JB * XLD];
  __shared__ double sred[4][2 + 16 * CT];
  __shared__ T cz[DL];                              // the centre of a feature slot (zeros for the unused ones)
  __shared__ T flo[SQ ? 8 : 1], finv[SQ ? 8 : 1];   // SQ: low end of a feature's range over the workgroup's rows, centres per unit length
  __shared__ int sfeat[SQ ? 8 : 1];                 // SQ: the feature a slot carries (d: none)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)), l15 = lane & 15, g = lane >> 4;
  const int d = kp.d;
  const T* __restrict__ invl = static_cast<const T*>(kp.invl);
  const int64_t iw0 = int64_t(blockIdx.y) * 64 + wave * 16, i = iw0 + l15;   // this lane's inducing row (tile column / A-operand row)
  // SQ: slot s = k d + f < nslot carries feature f about its centre number k (k = 0: the one the distances use - the middle position)
  const int nrep = SQ ? 8 / d : 1, nslot = SQ ? nrep * d : d;
  auto slot_pos = [&](int k) { return (k + nrep / 2) % nrep; };   // position of centre k along the range
  if (int(threadIdx.x) < DL) cz[threadIdx.x] = T(0);
  if (SQ && int(threadIdx.x) < 8) sfeat[threadIdx.x] = int(threadIdx.x) < nslot ? int(threadIdx.x) % d : d;
  __syncthreads();
  for (int f = wave; f < d; f += 4) {   // range of feature f over the valid rows of the workgroup (a wave per feature, a lane per row)
    const int64_t ri = int64_t(blockIdx.y) * 64 + lane;
    const T v = ri < M ? zs[int64_t(f) * Mp + ri] : T(0);
    T lo = ri < M ? v : T(INFINITY), hi = ri < M ? v : T(-INFINITY);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      lo = fmin(lo, __shfl_xor(lo, o));
      hi = fmax(hi, __shfl_xor(hi, o));
    }
    if (!(lo <= hi)) lo = hi = T(0);   // padding rows only
    if constexpr (SQ) {
      const T step = (hi - lo) / T(nrep);
      if (lane < nrep) cz[lane * d + f] = lo + (T(slot_pos(lane)) + T(0.5)) * step;
      if (lane == 0) {
        flo[f] = lo;
        finv[f] = hi > lo ? T(nrep) / (hi - lo) : T(0);
      }
    } else {
      if (lane == 0) cz[f] = T(0.5) * (lo + hi);
    }
  }
  __syncthreads();
  const int64_t j0 = int64_t(blockIdx.x) * slice_len;
  int64_t j1 = j0 + slice_len;
  j1 = j1 < n ? j1 : n;
  // z fragments of the wave's 16 rows (B operand of step (1): B[k = feature 4 q + g][col = row l15]) and c1 |z_i|^2
  T za[KS], zn = T(0);
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int f = 4 * q + g;
    const T v = (f < d) ? zs[int64_t(f) * Mp + i] - cz[f] : T(0);
    za[q] = ascale * v;
    zn = fma(v, v, zn);
  }
  zn += __shfl_xor(zn, 16);
  zn += __shfl_xor(zn, 32);
  zn *= c1;
  const T al = alpha ? alpha[i] : T(0);
  // the staged slot a lane reads as A operand of step (1): D row s of the tile is point g + 4 r for f64 (s = g + 4 r) and must be made
  // the same point for fp32 (s = 4 g + r), so that the k-slots of step (3) are the points g + 4 r in both
  const int slot_pt = kF64 ? l15 : ((l15 >> 2) + 4 * (l15 & 3));
  const T* __restrict__ xa = xt + slot_pt * XLD + g;         // + (16 t) XLD + 4 q
  const T* __restrict__ xb = xt + g * XLD + (SQ ? (l15 & 7) : l15);   // + (16 t + 4 r) XLD + 16 c
  const T* __restrict__ prow = Pt + i;
  // totals (fp64) and, fp32 builds, the staged block's sums
  double Rd = 0.0, MBd = 0.0, S1d = 0.0, ILd[SQ ? 1 : CT], Qd[kF64 ? 1 : CT][kF64 ? 1 : 4];
  acc_t Q[NQ][CT];
  T Rl = T(0), MBl = T(0), S1l = T(0), ILx[SQ ? 1 : CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    if constexpr (!SQ) { ILd[c] = 0.0; ILx[c] = T(0); }
    if constexpr (!kF64) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Qd[c][r] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) Q[q][c] = acc_t{0, 0, 0, 0};
  }
  if constexpr (SQ) ILd[0] = 0.0;
  (void)ILx;
  for (int64_t jb = j0; jb < j1; jb += JB) {
    __syncthreads();
    // stage the block: scaled inputs point-major, (g_mu, 2 g_v) of the point (zeros beyond the slice: P = 0 there), then c1 |x|^2
    for (int e = threadIdx.x; e < JB * DL; e += k256) {
      const int c = e % JB, sl = e / JB;                    // point, feature slot
      const int f = SQ ? sfeat[sl] : sl;                    // the slot's feature (d: none)
      int64_t gg = jb + c;
      gg = gg < nvalid ? gg : nvalid - 1;
      T v = T(0);
      if (f < d) v = (prescaled ? x[int64_t(f) * ldx + xoff + gg] : x[int64_t(f) * ldx + xoff + gg] * invl[f]) - cz[sl];
      xt[c * XLD + sl] = v;
    }
    if (threadIdx.x < JB) {
      const int64_t gg = jb + threadIdx.x;
      const bool ok = gg < j1;
      xt[threadIdx.x * XLD + GM] = (ok && gmu) ? gmu[gg] : T(0);
      xt[threadIdx.x * XLD + GV] = ok ? (alpha ? T(2) * gv[gg] : T(1)) : T(0);
    }
    __syncthreads();
    if (threadIdx.x < JB) {
      const T* __restrict__ xp = xt + threadIdx.x * XLD;
      T s = T(0);
#pragma unroll
      for (int f = 0; f < DL; ++f) {
        const T v = (!SQ || f < d) ? xp[f] : T(0);   // (SQ: slots d .. 7 repeat features about other centres)
        s = fma(v, v, s);
      }
      xt[threadIdx.x * XLD + XN] = c1 * s;
    }
    __syncthreads();
    // P of a tile is fetched one tile ahead (4 entries per lane: points g + 4 r of the tile, row i)
    T pn[4];
    auto fetch = [&](int t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int64_t j = jb + 16 * t + g + 4 * r;
        j = j < j1 ? j : j1 - 1;
        pn[r] = prow[j * Mp];
      }
    };
    fetch(0);
#pragma unroll 1
    for (int t = 0; t < NTILE; ++t) {
      if (jb + 16 * t >= j1) break;   // (workgroup-uniform)
      T pv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pv[r] = pn[r];
      if (t + 1 < NTILE) fetch(t + 1);   // (a tile wholly beyond j1 re-reads the last point: never used)
      const T* __restrict__ xat = xa + (16 * t) * XLD;
      const T* __restrict__ xbt = xb + (16 * t) * XLD;
      // (1) the distance tile
      acc_t a, a2 = {0, 0, 0, 0};
      T gm[4], gv2[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T* __restrict__ xp = xt + (16 * t + g + 4 * r) * XLD;
        gm[r] = xp[GM];
        gv2[r] = xp[GV];
        a[r] = xp[XN] + zn;
      }
      if constexpr (KS >= 4) {   // two chains (even / odd feature slabs)
#pragma unroll
        for (int q = 0; q < KS; q += 2) {
          a = M16::mma(xat[4 * q], za[q], a);
          a2 = M16::mma(xat[4 * q + 4], za[q + 1], a2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] += a2[r];
      } else {
#pragma unroll
        for (int q = 0; q < KS; ++q) a = M16::mma(xat[4 * q], za[q], a);
      }
      // (2) kernel function, P, W
      T w[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T p = fma(gv2[r], pv[r], al * gm[r]);   // (alpha == nullptr: 1 * pv + 0; beyond the slice: 0)
        if constexpr (FAMILY == KSE) {
          // unit variance and without the factor -1/2 of dK/dr2: both are applied to the sums at the end (sum P o K = sum_i of the R_i here)
          const T e = kexp(a[r] > T(0) ? T(0) : a[r]);
          w[r] = p * e;
          if (kmb) MBl = fma(e, gm[r], MBl);
        } else {
          T k, dk;
          kappa_and_d<T, FAMILY>(a[r] < T(0) ? T(0) : a[r], T(1), k, dk);
          w[r] = p * dk;
          S1l = fma(p, k, S1l);
          if (kmb) MBl = fma(k, gm[r], MBl);
        }
        Rl += w[r];
      }
      // (3) the feature sums
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        [[maybe_unused]] T cs = T(0);
        if constexpr (!SQ) cs = row16_sum(w[r]);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          T xv = xbt[(4 * r) * XLD + 16 * c];
          if constexpr (SQ) xv = (l15 & 8) ? xv * xv : xv;
          else ILx[c] = fma(xv * xv, cs, ILx[c]);
          Q[r % NQ][c] = M16::mma(w[r], xv, Q[r % NQ][c]);
        }
      }
    }
    if constexpr (!kF64) {   // the block's fp32 sums join the fp64 totals
      Rd += double(Rl); MBd += double(MBl); S1d += double(S1l);
      Rl = MBl = S1l = T(0);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if constexpr (!SQ) { ILd[c] += double(ILx[c]); ILx[c] = T(0); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          T v = Q[0][c][r];
          if constexpr (NQ == 2) v += Q[1][c][r];
          Qd[c][r] += double(v);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) Q[q][c] = acc_t{0, 0, 0, 0};
      }
    }
  }
  // ---- closing sums ----
  // scale of everything that is linear in W: the variance, and for SE the -1/2 of dK/dr2 left out above
  const double var = kp.variance, wsc = (FAMILY == KSE) ? -0.5 * var : var;
  double R = kF64 ? double(Rl) : Rd, MB = kF64 ? double(MBl) : MBd, S1 = kF64 ? double(S1l) : S1d;
  R += __shfl_xor(R, 16); R += __shfl_xor(R, 32);          // row l15, all its points
  MB += __shfl_xor(MB, 16); MB += __shfl_xor(MB, 32);
  double* rp = rowpart + int64_t(blockIdx.x) * (2 + DL) * Mp;
  if (g == 0) {
    rp[i] += wsc * R;
    if (kmb) rp[Mp + i] += var * MB;
  }
  // Q in the D layout: lane (feature l15 + 16 c, group g), register r <-> row M16::row(lane, r) of the wave's 16
  __syncthreads();
  if (g == 0) sred[wave][2 + l15] = R;
  __syncthreads();
  double ilw[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int sl = SQ ? (l15 & 7) : (l15 + 16 * c);                                // this lane's feature slot
    const int f = SQ ? sfeat[sl] : sl;
    [[maybe_unused]] const int pos = SQ ? slot_pos(sl / d) : 0;
    double il = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = M16::row(lane, r);
      double q;
      if constexpr (kF64) { q = double(Q[0][c][r]); if constexpr (NQ == 2) q += double(Q[1][c][r]); }
      else q = Qd[c][r];
      const T zrow = (f < d) ? zs[int64_t(f) * Mp + iw0 + row] : T(0);
      bool sel = f < d;
      if constexpr (SQ) {   // the row takes feature f from the slot whose centre is nearest to it
        if (sel) {
          int prow = int((zrow - flo[f]) * finv[f]);
          prow = prow < 0 ? 0 : (prow > nrep - 1 ? nrep - 1 : prow);
          sel = prow == pos;
        }
      }
      if (SQ && (l15 & 8)) {
        il += sel ? q : 0.0;                                 // sum_j W_ij x_fj^2 (about the slot's centre)
      } else if (sel) {
        const double cf = double(cz[sl]), Rr = sred[wave][2 + row];
        const double zf = double(zrow - cz[sl]);             // (the centred value as the tiles saw it)
        il += zf * (zf * Rr - 2.0 * q);
        rp[int64_t(2 + f) * Mp + iw0 + row] += wsc * (q + cf * Rr);                           // sum_j W_ij x_fj, uncentred
      }
    }
    if constexpr (SQ) il += __shfl_xor(il, 8);               // lanes l15 < 8: feature l15 complete over the lane's rows
    else il += kF64 ? double(ILx[c]) : ILd[c];               // + this lane's share of sum_j x_fj^2 C_j
    il += __shfl_xor(il, 16);
    il += __shfl_xor(il, 32);
    ilw[c] = il;
  }
  // sum P o K of the wave: SE - the sum of its R_i; Matern - accumulated
  if constexpr (FAMILY == KSE) {
    S1 = (g == 0) ? R : 0.0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) S1 += __shfl_xor(S1, o);
  } else {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) S1 += __shfl_xor(S1, o);
  }
  __syncthreads();
  if (lane == 0) sred[wave][0] = S1;
  if (g == 0 && (!SQ || l15 < 8)) {
#pragma unroll
    for (int c = 0; c < CT; ++c) sred[wave][2 + l15 + 16 * c] = ilw[c];
  }
  __syncthreads();
  if (int(threadIdx.x) <= DL) {
    const int q = threadIdx.x;
    const int slot = q == 0 ? 0 : 1 + q;
    double v = ((sred[0][slot] + sred[1][slot]) + sred[2][slot]) + sred[3][slot];
    if constexpr (SQ) {   // feature q - 1: its further slots
      if (q >= 1 && q - 1 < d)
        for (int k = 1; k < nrep; ++k) v += ((sred[0][slot + k * d] + sred[1][slot + k * d]) + sred[2][slot + k * d]) + sred[3][slot + k * d];
    }
    double* sp = scalpart + (int64_t(blockIdx.x) * gridDim.y + blockIdx.y) * (1 + DL);
    if (q == 0) sp[0] += var * v;
    else if (q - 1 < d) sp[q] += wsc * v;
  }
}

// ---- small M x M helpers -------------------------------------------------------------------------------------
template <typename T>
__global__ void lower_to_rowmajor_kernel(const T* __restrict__ L, int64_t Mp, T* __restrict__ out) {
  // out[k][r] = L[k + r*Mp] for k >= r else 0   (32x32 LDS transpose)
  __shared__ T tile[32][33];
  const int64_t br = int64_t(blockIdx.x) * 32, bk = int64_t(blockIdx.y) * 32;
  for (int q = threadIdx.y; q < 32; q += blockDim.y) {
    const int64_t k = bk + threadIdx.x, r = br + q;
    tile[q][threadIdx.x] = (k >= r) ? L[k + r * Mp] : T(0);
  }
  __syncthreads();
  for (int q = threadIdx.y; q < 32; q += blockDim.y) out[(bk + q) * Mp + br + threadIdx.x] = tile[threadIdx.x][q];
}

template <typename T>
__global__ void symmetrize_kernel(const T* __restrict__ St, int64_t Mp, T* __restrict__ H) {
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c < Mp) H[r * Mp + c] = T(0.5) * (St[r * Mp + c] + St[c * Mp + r]);
}

// Phi = tril(X) with the diagonal halved, in place on a row-major Mp x Mp matrix
template <typename T>
__global__ void phi_kernel(T* __restrict__ X, int64_t Mp) {
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c < Mp) {
    const T v = X[r * Mp + c];
    X[r * Mp + c] = c < r ? v : (c == r ? T(0.5) * v : T(0));
  }
}

// Centered chain rule helpers ------------------------------------------------------------------------------
// vec[i] = avec[i] - mtilde[i]   (adjoint of the whitened mean; avec = A g_mu), zero padded
template <typename T>
__global__ void mbar_kernel(const double* __restrict__ avec, const T* __restrict__ mt, double klw, int64_t M, int64_t Mp,
                            T* __restrict__ vec) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= Mp) return;
  vec[i] = (i < M) ? T(avec[i] - klw * double(mt[i])) : T(0);
}

// LkbarRM[r][c] -= (R B')[r][c] + rbar[r] mtilde[c]  for c <= r
template <typename T>
__global__ void lbar_adjust_kernel(T* __restrict__ LkbarRM, const T* __restrict__ RBt, const T* __restrict__ rbar,
                                   const T* __restrict__ mt, int64_t Mp) {
  const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c < Mp && c <= r) LkbarRM[r * Mp + c] -= RBt[r * Mp + c] + rbar[r] * mt[c];
}

// user-layout lower triangle of a matrix held COLUMN-major in an Mp x Mp buffer (element (r, c) at R[c Mp + r])
template <typename T>
__global__ void cm_tril_to_user_kernel(const T* __restrict__ R, int64_t Mp, int64_t M, T* __restrict__ out) {
  const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (r < M) out[r + c * M] = (c <= r) ? R[c * Mp + r] : T(0);
}

// Slice partials of the kernel-gradient reductions, summed in a fixed order by many workgroups (a single pass of M threads
// over 256 slices x (2 + d) rows was 0.37-0.43 ms of latency per gradient):
//   red[q][i] = sum_s rp_uf[s][q][i] + (q == 1 ? 0 : 2 sum_s rp_uu[s][q][i])        blockIdx.y = q < 2 + dreg
//   red_s[q]  = sum_b sp_uf[b][q] + sum_b sp_uu[b][q]                                blockIdx.y = 2 + dreg (q strided over x)
// (row 1 is m_bar's data part, which has no Kuu term; the factor 2 is the symmetric Kuu's.)
__global__ void __launch_bounds__(256) kgrad_reduce_kernel(int d, int dreg, int64_t Mp, const double* __restrict__ rp_uf, int ns_uf,
                                                           const double* __restrict__ rp_uu, int ns_uu,
                                                           const double* __restrict__ sp_uf, int nsp_uf,
                                                           const double* __restrict__ sp_uu, int nsp_uu,
                                                           double* __restrict__ red, double* __restrict__ red_s) {
  __shared__ double sh[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t stride = int64_t(2 + dreg) * Mp;
  if (int(blockIdx.y) < 2 + dreg) {
    const int q = blockIdx.y;
    const int64_t i = int64_t(blockIdx.x) * 64 + lane;
    double a = 0, b = 0;
    if (i < Mp) {
      for (int s = wave; s < ns_uf; s += 4) a += rp_uf[s * stride + q * Mp + i];
      if (q != 1)
        for (int s = wave; s < ns_uu; s += 4) b += rp_uu[s * stride + q * Mp + i];
    }
    sh[threadIdx.x] = a + 2.0 * b;
    __syncthreads();
    if (wave == 0 && i < Mp) red[q * Mp + i] = (sh[lane] + sh[64 + lane]) + (sh[128 + lane] + sh[192 + lane]);
    return;
  }
  for (int q = blockIdx.x; q <= d; q += gridDim.x) {
    double a = 0;
    for (int b = threadIdx.x; b < nsp_uf; b += 256) a += sp_uf[int64_t(b) * (1 + dreg) + q];
    for (int b = threadIdx.x; b < nsp_uu; b += 256) a += sp_uu[int64_t(b) * (1 + dreg) + q];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (int(threadIdx.x) < w) sh[threadIdx.x] += sh[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) red_s[q] = sh[0];
    __syncthreads();
  }
}

// final assembly of the kernel-parameter / inducing-input gradients from the reduced partials
template <typename T>
__global__ void finish_kgrad_kernel(int d, int64_t M, int64_t Mp, const T* __restrict__ zs, const double* __restrict__ invl,
                                    const double* __restrict__ red, const double* __restrict__ red_s, const T* __restrict__ m,
                                    double klw, int layout_z, double variance, T* __restrict__ z_bar, T* __restrict__ m_bar,
                                    double* __restrict__ scal_out, const double* __restrict__ avec) {
  // scal_out[0] = (sum P K (uf) + sum H K (uu)) / variance, scal_out[1 + f] = il_bar_f
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < M) {
    const double R = red[i];
    if (m_bar) m_bar[i] = T(avec[i] - klw * double(m[i]));   // A g_mu - klw * d KL / d m
    for (int f = 0; f < d; ++f) {
      const double zf = double(zs[int64_t(f) * Mp + i]);
      const double g = 2.0 * invl[f] * (zf * R - red[(2 + f) * Mp + i]);
      if (layout_z == 1) z_bar[int64_t(f) * M + i] = T(g);   // RowVecs: M x d column-major
      else z_bar[i * d + f] = T(g);                           // ColVecs / Vec
    }
  }
  if (blockIdx.x == 0 && int(threadIdx.x) <= d) {
    const int q = threadIdx.x;
    scal_out[q] = (q == 0) ? red_s[0] / variance : 2.0 / invl[q - 1] * red_s[q];
  }
}

// 64 rows per workgroup for every d
template <typename T, int FAMILY>
void launch_kgrad_f(hipStream_t s, const KernelParams& kp, const T* zs, int64_t Mp, const T* x, int64_t ldx, int64_t xoff,
                    int prescaled, int64_t n, int64_t nvalid, const T* Pt, const T* gmu, const T* gv, const T* alpha,
                    int64_t slice_len, int nslices, double* rowpart, double* scalpart, int kmb, int64_t M) {
  dim3 grid((unsigned)nslices, (unsigned)(Mp / 64));
#define SVGP_KGM(DL) hipLaunchKernelGGL((kgrad_mfma_kernel<T, FAMILY, DL>), grid, dim3(k256), 0, s, kp, zs, Mp, x, ldx, xoff, prescaled, n, \
                                        nvalid, Pt, gmu, gv, alpha, slice_len, rowpart, scalpart, kmb, M)
  if (kp.d <= 8) SVGP_KGM(8);
  else if (kp.d <= 16) SVGP_KGM(16);
  else if (kp.d <= 32) SVGP_KGM(32);
  else SVGP_KGM(64);
#undef SVGP_KGM
}

}  // namespace

// ---- launchers -----------------------------------------------------------------------------------------------
#define GD(dtype, T, ...)               \
  do {                                  \
    if ((dtype) == 0) { using T = double; __VA_ARGS__; } else { using T = float; __VA_ARGS__; } \
  } while (0)

int grad_dreg(int d) { return d <= 8 ? 8 : (d <= 16 ? 16 : (d <= 32 ? 32 : 64)); }
// workgroups along the rows of the kernel-gradient reductions: launch_kgrad_f's grid
int grad_rowblocks(int dtype, int d, int64_t Mp) {
  (void)dtype; (void)d;
  return int(Mp / 64);
}

void launch_set_f64(hipStream_t s, double* dst, double value) { hipLaunchKernelGGL(set_f64_kernel, dim3(1), dim3(1), 0, s, dst, value); }
struct Vec64 { double v[64]; };
__global__ void setvec_f64_kernel(double* __restrict__ dst, Vec64 vals, int n) {
  if (int(threadIdx.x) < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}
// dst[0 .. n) = vals (n <= 64), the values travelling as kernel arguments: unlike a hipMemcpyAsync from pageable host memory this
// does not block the enqueueing thread until the stream gets there
void launch_setvec_f64(hipStream_t s, double* dst, const double* vals, int n) {
  Vec64 v{};
  for (int i = 0; i < n && i < 64; ++i) v.v[i] = vals[i];
  hipLaunchKernelGGL(setvec_f64_kernel, dim3(1), dim3(64), 0, s, dst, v, n < 64 ? n : 64);
}
void launch_set2_f64(hipStream_t s, double* dst, double a, double b) { hipLaunchKernelGGL(set2_f64_kernel, dim3(1), dim3(1), 0, s, dst, a, b); }
void launch_grad_status(hipStream_t s, double* sums, const int* chol_info, double n_points, const double* prep_scal, double* prep_out) {
  hipLaunchKernelGGL(grad_status_kernel, dim3(1), dim3(1), 0, s, sums, chol_info, n_points, prep_scal, prep_out);
}

void launch_sum5(hipStream_t s, const double* partial, int nblocks, double* sums) {
  hipLaunchKernelGGL(sum5_kernel, dim3(1), dim3(320), 0, s, partial, nblocks, sums);
}

void launch_sym_from_lower(int dtype, hipStream_t s, const void* G, int nslices, int64_t Mp, double eye, void* out) {
  const int64_t nb = Mp / 32;
  GD(dtype, T, hipLaunchKernelGGL(sym_from_lower_kernel<T>, dim3((unsigned)(nb * (nb + 1) / 2)), dim3(k256), 0, s, (const T*)G, nslices, Mp,
                                  T(eye), (T*)out));
}

void launch_avec(hipStream_t s, const double* rp_uf, int ns, int64_t stride, int64_t Mp, double* avec) {
  hipLaunchKernelGGL(avec_kernel, dim3((unsigned)((Mp + 63) / 64)), dim3(k256), 0, s, rp_uf, ns, stride, Mp, avec);
}

void launch_finish_mm2(int dtype, hipStream_t s, const void* G1, const void* G2, const void* alpha, const double* avec, int64_t Mp,
                       int64_t M, const void* Lq, int64_t ldq, double klw, void* Lq_bar, void* BbarRM, void* LkbarRM) {
  dim3 grid((unsigned)((Mp + 255) / 256), (unsigned)Mp);
  GD(dtype, T, hipLaunchKernelGGL(finish_mm2_kernel<T>, grid, dim3(256), 0, s, (const T*)G1, (const T*)G2, (const T*)alpha, avec, Mp,
                                  M, (const T*)Lq, ldq, T(klw), (T*)Lq_bar, (T*)BbarRM, (T*)LkbarRM));
}

void launch_sum_slices_lower(int dtype, hipStream_t s, const void* part, int ns, int64_t Mp, void* out, int full, int overwrite) {
  dim3 grid((unsigned)((Mp + 255) / 256), (unsigned)Mp);
  GD(dtype, T, hipLaunchKernelGGL(sum_slices_lower_kernel<T>, grid, dim3(256), 0, s, (const T*)part, ns, Mp, (T*)out, full, overwrite));
}

void launch_linv(int dtype, hipStream_t s, const void* L, const void* Tm, int64_t Mp, void* LinvRM, void* LinvCM, void* Ytmp) {
  const int nP = int(Mp / kNB);
  GD(dtype, T, {
    constexpr int NT = sizeof(T) == 8 ? 32 : 64, NCH = kNB / NT;   // the latency-bound chunking of the Cholesky tiles
    using G = TileGemm<T, NT, 16, k256>;
    set_max_lds(reinterpret_cast<const void*>(linv_step_kernel<T, NT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, int(G::LDS_BYTES));
    set_max_lds(reinterpret_cast<const void*>(linv_step_kernel<T, NT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, int(G::LDS_BYTES));
    hipLaunchKernelGGL(linv_init_kernel<T>, dim3(16, (unsigned)nP), dim3(k256), 0, s, (const T*)Tm, Mp, (T*)LinvRM, (T*)LinvCM);
    for (int sz = 1; sz < nP; sz *= 2) {
      const dim3 grid((unsigned)(sz * sz * NCH), (unsigned)((nP + 2 * sz - 1) / (2 * sz)));
      hipLaunchKernelGGL((linv_step_kernel<T, NT, 1>), grid, dim3(k256), G::LDS_BYTES, s, (const T*)L, (T*)LinvRM, (T*)LinvCM, (T*)Ytmp, Mp, nP, sz);
      hipLaunchKernelGGL((linv_step_kernel<T, NT, 2>), grid, dim3(k256), G::LDS_BYTES, s, (const T*)L, (T*)LinvRM, (T*)LinvCM, (T*)Ytmp, Mp, nP, sz);
    }
  });
}

void launch_linv_t_gemv(int dtype, hipStream_t s, const void* LinvRM, const void* v, int64_t Mp, void* out, double* part, int notrans,
                        int vec_f64) {
  const int nP = int(Mp / kNB);
  if (vec_f64 && dtype != 0) {   // fp32 matrix, fp64 vector in and out (the accumulated Kuf g_mu of the kernel-gradient reductions)
    hipLaunchKernelGGL((linv_t_gemv_kernel<float, double>), dim3((unsigned)(Mp / 64), (unsigned)nP), dim3(k256), 0, s, (const float*)LinvRM, (const double*)v, Mp, part, notrans);
    hipLaunchKernelGGL(gemv_finish_kernel<double>, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, s, part, nP, Mp, (double*)out);
    return;
  }
  GD(dtype, T, {
    hipLaunchKernelGGL((linv_t_gemv_kernel<T, T>), dim3((unsigned)(Mp / 64), (unsigned)nP), dim3(k256), 0, s, (const T*)LinvRM, (const T*)v, Mp, part, notrans);
    hipLaunchKernelGGL(gemv_finish_kernel<T>, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, s, part, nP, Mp, (T*)out);
  });
}

void launch_gemm_pm(int dtype, hipStream_t s, const void* Xt, const void* Yt, const void* w, double wscale, int64_t Mp,
                    int64_t n, int64_t slice_len, int nslices, void* out, int overwrite, int flags) {
  const int nP = int(Mp / kNB), ntiles = (flags & kMmFull) ? nP * nP : nP * (nP + 1) / 2;
  // f64: 128 x 64 halves on 256-thread workgroups (same-box: H value-and-gradient 141.0 -> 138.4 ms); f32: no difference
  static const int forced = exp_int("SVGP_GEMM_PM_NT", 0);   // tuning knob (experiments build)
  const int nt = forced ? forced : (dtype == 0 ? 64 : 128);
  // the weighted SYRK of the gradient (Xt == Yt, weights 2 g_v folded by the caller into w): asynchronous loop (knob)
  static const int async_knob = exp_int("SVGP_SYRK_ASYNC", 1);   // (experiments build)
  if (async_knob && w && Xt == Yt && flags == 0 && !forced) {
    GD(dtype, T, {
      constexpr int NT = sizeof(T) == 8 ? 64 : 128;
      using G = TileGemm<T, NT, 16, k256>;
      auto kern = syrk_async_kernel<T, NT>;
      set_max_lds(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(G::ASYNC_W_LDS_BYTES));
      hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles * (kNB / NT)), (unsigned)nslices), dim3(k256), G::ASYNC_W_LDS_BYTES, s,
                         (const T*)Xt, (const T*)w, T(wscale), Mp, n, slice_len, (T*)out, overwrite, UniformW{});
    });
    return;
  }
  GD(dtype, T, {
    if (nt == 64) {
      using G = TileGemm<T, 64, 16, k256>;
      auto kern = gemm_pm_kernel<T, 64, k256>;
      set_max_lds(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(G::LDS_BYTES));
      hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles * 2), (unsigned)nslices), dim3(k256), G::LDS_BYTES, s, (const T*)Xt,
                         (const T*)Yt, (const T*)w, T(wscale), Mp, n, slice_len, (T*)out, overwrite, flags);
    } else {
      using G = TileGemm<T, kNB, 16, kThreads>;
      auto kern = gemm_pm_kernel<T, kNB, kThreads>;
      set_max_lds(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(G::LDS_BYTES));
      hipLaunchKernelGGL(kern, dim3((unsigned)ntiles, (unsigned)nslices), dim3(kThreads), G::LDS_BYTES, s, (const T*)Xt,
                         (const T*)Yt, (const T*)w, T(wscale), Mp, n, slice_len, (T*)out, overwrite, flags);
    }
  });
}

void launch_syrk_uniform(int dtype, hipStream_t s, const void* At, double w, double scale, const double* n_global_dev, double num_data,
                         double wscale, int64_t Mp, int64_t n, int64_t slice_len, int nslices, void* out, int overwrite) {
  const int nP = int(Mp / kNB), ntiles = nP * (nP + 1) / 2;
  GD(dtype, T, {
    constexpr int NT = sizeof(T) == 8 ? 64 : 128;
    using G = TileGemm<T, NT, 16, k256>;
    auto kern = syrk_async_kernel<T, NT, true>;
    set_max_lds(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(G::ASYNC_LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles * (kNB / NT)), (unsigned)nslices), dim3(k256), G::ASYNC_LDS_BYTES, s, (const T*)At,
                       (const T*)nullptr, T(wscale), Mp, n, slice_len, (T*)out, overwrite, UniformW{w, scale, num_data, n_global_dev});
  });
}

void launch_kgrad(int dtype, hipStream_t s, const KernelParams& kp, const void* zs, int64_t Mp, int64_t M, const void* x, int64_t ldx,
                  int64_t xoff, int prescaled, int64_t n, int64_t nvalid, const void* Pt, const void* gmu,
                  const void* gv, const void* alpha, int64_t slice_len, int nslices, double* rowpart, double* scalpart, int kmb) {
  GD(dtype, T, {
    if (kp.family == KSE)
      launch_kgrad_f<T, KSE>(s, kp, (const T*)zs, Mp, (const T*)x, ldx, xoff, prescaled, n, nvalid, (const T*)Pt,
                             (const T*)gmu, (const T*)gv, (const T*)alpha, slice_len, nslices, rowpart, scalpart, kmb, M);
    else if (kp.family == KM32)
      launch_kgrad_f<T, KM32>(s, kp, (const T*)zs, Mp, (const T*)x, ldx, xoff, prescaled, n, nvalid, (const T*)Pt,
                              (const T*)gmu, (const T*)gv, (const T*)alpha, slice_len, nslices, rowpart, scalpart, kmb, M);
    else
      launch_kgrad_f<T, KM52>(s, kp, (const T*)zs, Mp, (const T*)x, ldx, xoff, prescaled, n, nvalid, (const T*)Pt,
                              (const T*)gmu, (const T*)gv, (const T*)alpha, slice_len, nslices, rowpart, scalpart, kmb, M);
  });
}

void launch_lower_to_rowmajor(int dtype, hipStream_t s, const void* L, int64_t Mp, void* out) {
  dim3 grid((unsigned)(Mp / 32), (unsigned)(Mp / 32)), block(32, 8);
  GD(dtype, T, hipLaunchKernelGGL(lower_to_rowmajor_kernel<T>, grid, block, 0, s, (const T*)L, Mp, (T*)out));
}

void launch_symmetrize(int dtype, hipStream_t s, const void* St, int64_t Mp, void* H) {
  dim3 grid((unsigned)((Mp + 255) / 256), (unsigned)Mp);
  GD(dtype, T, hipLaunchKernelGGL(symmetrize_kernel<T>, grid, dim3(256), 0, s, (const T*)St, Mp, (T*)H));
}

void launch_phi(int dtype, hipStream_t s, void* X, int64_t Mp) {
  dim3 grid((unsigned)((Mp + 255) / 256), (unsigned)Mp);
  GD(dtype, T, hipLaunchKernelGGL(phi_kernel<T>, grid, dim3(256), 0, s, (T*)X, Mp));
}

void launch_mbar(int dtype, hipStream_t s, const double* avec, const void* mt, double klw, int64_t M, int64_t Mp, void* vec) {
  GD(dtype, T, hipLaunchKernelGGL(mbar_kernel<T>, dim3((unsigned)((Mp + 255) / 256)), dim3(256), 0, s, avec,
                                  (const T*)mt, klw, M, Mp, (T*)vec));
}

void launch_lbar_adjust(int dtype, hipStream_t s, void* LkbarRM, const void* RBt, const void* rbar, const void* mt, int64_t Mp) {
  dim3 grid((unsigned)((Mp + 255) / 256), (unsigned)Mp);
  GD(dtype, T, hipLaunchKernelGGL(lbar_adjust_kernel<T>, grid, dim3(256), 0, s, (T*)LkbarRM, (const T*)RBt, (const T*)rbar,
                                  (const T*)mt, Mp));
}

void launch_cm_tril_to_user(int dtype, hipStream_t s, const void* R, int64_t Mp, int64_t M, void* out) {
  dim3 grid((unsigned)((M + 255) / 256), (unsigned)M);
  GD(dtype, T, hipLaunchKernelGGL(cm_tril_to_user_kernel<T>, grid, dim3(256), 0, s, (const T*)R, Mp, M, (T*)out));
}

// The collective gradient's compute-dtype block {z_bar | m_bar | Lq_bar}: Lq_bar is lower triangular, so only its M (M + 1) / 2
// entries travel (SURVEY 8 f1 counts M^2 / 2 for the all-reduce; round 3 sent the full M x M block, half of it zeros).  Packed by
// columns: entry (r >= c) at head + c M - c (c - 1) / 2 + (r - c).  dir 0: gblk -> packed, dir 1: packed -> gblk (upper untouched: zero).
template <typename T>
__global__ void pack_tril_kernel(T* __restrict__ gblk, T* __restrict__ packed, int64_t head, int64_t M, int dir) {
  const int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (c == 0 && r < head) {   // z_bar | m_bar: copied as they are (head = M d + M entries, by the first grid row and its neighbours)
    if (dir == 0) packed[r] = gblk[r]; else gblk[r] = packed[r];
  }
  if (c == 0)
    for (int64_t q = r + int64_t(gridDim.x) * blockDim.x; q < head; q += int64_t(gridDim.x) * blockDim.x) {
      if (dir == 0) packed[q] = gblk[q]; else gblk[q] = packed[q];
    }
  if (r >= M || r < c) return;
  const int64_t pi = head + c * M - c * (c - 1) / 2 + (r - c), gi = head + c * M + r;
  if (dir == 0) packed[pi] = gblk[gi]; else gblk[gi] = packed[pi];
}

__global__ void add_f64_kernel(double* p, double v) { *p += v; }
void launch_pack_tril(int dtype, hipStream_t s, void* gblk, void* packed, int64_t head, int64_t M, int dir) {
  dim3 grid((unsigned)((M + 255) / 256), (unsigned)M);
  GD(dtype, T, hipLaunchKernelGGL(pack_tril_kernel<T>, grid, dim3(256), 0, s, (T*)gblk, (T*)packed, head, M, dir));
}

void launch_add_f64(hipStream_t s, double* p, double v) { hipLaunchKernelGGL(add_f64_kernel, dim3(1), dim3(1), 0, s, p, v); }

void launch_finish_kgrad(int dtype, hipStream_t s, int d, int64_t M, int64_t Mp, const void* zs, const double* invl,
                         const double* rp_uf, int ns_uf, const double* rp_uu, int ns_uu, const double* sp_uf, int nsp_uf,
                         const double* sp_uu, int nsp_uu, const void* m, double klw, int layout_z, double variance,
                         void* z_bar, void* m_bar, double* scal_out, double* red, const double* avec) {
  const int dreg = grad_dreg(d);
  double* red_s = red + int64_t(2 + dreg) * Mp;
  hipLaunchKernelGGL(kgrad_reduce_kernel, dim3((unsigned)((Mp + 63) / 64), (unsigned)(3 + dreg)), dim3(256), 0, s, d, dreg, Mp, rp_uf,
                     ns_uf, rp_uu, ns_uu, sp_uf, nsp_uf, sp_uu, nsp_uu, red, red_s);
  dim3 grid((unsigned)((M + 255) / 256));
  GD(dtype, T, hipLaunchKernelGGL(finish_kgrad_kernel<T>, grid, dim3(256), 0, s, d, M, Mp, (const T*)zs, invl, red, red_s,
                                  (const T*)m, klw, layout_z, variance, (T*)z_bar, (T*)m_bar, scal_out, avec));
}

}  // namespace svgp
