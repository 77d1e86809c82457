// Device-side building blocks shared by every kernel of libsvgp_mi355x (gfx950 only).
//   * Mfma16<T>: the 16x16x4 MFMA (f64 / f32-in-f32-acc) with its C/D lane map
//   * TileGemm<T, NT, BK>: a 128 x NT output tile per 256-thread workgroup, K streamed through
//     double-buffered LDS tiles, operands given as "k-major" matrices (element (k, i) at k*ld + i),
//     which is exactly a column-major Julia matrix read along its columns.
//   * the stationary kernel functions of KernelFunctions.jl (SE, Matern-3/2, Matern-5/2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace svgp {

// allow a kernel to use up to the whole 160 KiB LDS of a CU as dynamic shared memory
inline void set_max_lds(const void* fn, hipFuncAttribute attr, int bytes) {
  hipError_t e = hipFuncSetAttribute(fn, attr, bytes);
  if (e != hipSuccess) fprintf(stderr, "[svgp] hipFuncSetAttribute(%d bytes) failed: %s\n", bytes, hipGetErrorString(e));
}

constexpr int kWave = 64;
constexpr int kThreads = 512;  // tile kernels: 8 waves, two per SIMD, sharing one 128 x NT tile
constexpr int k256 = 256;      // small helper kernels
constexpr int kNB = 128;       // row-panel height == Cholesky block == diagonal-inverse block

template <typename T>
struct Mfma16;

// v_mfma_f64_16x16x4_f64: A[i][k] in lane 16k+i, B[k][j] in lane 16k+j, D[i][j]: lane 16(i%4)+j, reg i/4.
template <>
struct Mfma16<double> {
  using acc_t = double __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};

// v_mfma_f32_16x16x4_f32: same A/B maps, D[i][j]: lane 16(i/4)+j, reg i%4.
template <>
struct Mfma16<float> {
  using acc_t = float __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

template <typename T>
struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
  using type = T __attribute__((ext_vector_type(16 / sizeof(T))));
};

// ---------------------------------------------------------------------------------------------
// kernel functions  [KernelFunctions.jl]: κ(r²) with r² already scaled by the inverse lengthscales
// ---------------------------------------------------------------------------------------------
enum : int { KSE = 0, KM32 = 1, KM52 = 2 };

__device__ __forceinline__ double kexp(double v) { return exp(v); }
__device__ __forceinline__ float kexp(float v) { return expf(v); }
__device__ __forceinline__ double ksqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float ksqrt(float v) { return sqrtf(v); }

template <typename T>
__device__ __forceinline__ T kappa(int family, T r2, T variance) {
  if (family == KSE) return variance * kexp(T(-0.5) * r2);
  T r = ksqrt(r2);
  if (family == KM32) {
    T s = T(1.7320508075688772935) * r;
    return variance * (T(1) + s) * kexp(-s);
  }
  T s = T(2.2360679774997896964) * r;
  return variance * (T(1) + s + T(5.0 / 3.0) * r2) * kexp(-s);
}

// ---------------------------------------------------------------------------------------------
// TileGemm: acc(128 x NT) += P(128 x K) * Q(K x NT)
// ---------------------------------------------------------------------------------------------
template <typename T, int NT, int BK>
struct TileGemm {
  static constexpr int NB = kNB;
  static constexpr int WR = 2, WC = 4;           // wave grid: a wave owns 64 rows x NT/4 columns
  static constexpr int MI = NB / WR / 16;        // 16-row tiles per wave (4)
  static constexpr int NJ = NT / WC / 16;        // 16-col tiles per wave (2 @ NT=128, 1 @ NT=64)
  static constexpr int PLD = NB + 16;            // LDS leading dims: +16 elements makes the 4 k-rows a
  static constexpr int QLD = NT + 16;            // 64-lane fragment read touches land on disjoint banks
  static constexpr int VEC = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  using acc_t = typename Mfma16<T>::acc_t;

  static constexpr int P_TPR = NB / VEC;                 // threads per k-row of the P tile
  static constexpr int P_RPP = kThreads / P_TPR;         // k-rows per pass
  static constexpr int P_PASSES = BK / P_RPP;
  static constexpr int Q_TPR = NT / VEC;
  static constexpr int Q_RPP = kThreads / Q_TPR;
  static constexpr int Q_PASSES = (BK / Q_RPP) > 0 ? (BK / Q_RPP) : 1;
  static_assert(BK % P_RPP == 0, "BK must be a multiple of the P rows per pass");
  static_assert(BK % Q_RPP == 0, "BK must be a multiple of the Q rows per pass");

  static constexpr int P_TILE = BK * PLD;
  static constexpr int Q_TILE = BK * QLD;
  static constexpr int STAGE = P_TILE + Q_TILE;          // elements per buffer
  static constexpr size_t LDS_BYTES = 2 * size_t(STAGE) * sizeof(T);

  struct Acc {
    acc_t v[MI][NJ];
    __device__ __forceinline__ void zero() {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[i][j] = acc_t{0, 0, 0, 0};
    }
  };

  struct PRegs { V v[P_PASSES]; };
  struct QRegs { V v[Q_PASSES]; };

  // --- global -> registers, k-major source: element (k, i) at src[k*ld + i] -------------------
  static __device__ __forceinline__ void load_p(PRegs& r, const T* __restrict__ src, int64_t ld) {
    const int t = threadIdx.x;
    const int kk0 = t / P_TPR, c = (t % P_TPR) * VEC;
#pragma unroll
    for (int p = 0; p < P_PASSES; ++p)
      r.v[p] = *reinterpret_cast<const V*>(src + int64_t(kk0 + p * P_RPP) * ld + c);
  }
  static __device__ __forceinline__ void store_p(const PRegs& r, T* __restrict__ Ps) {
    const int t = threadIdx.x;
    const int kk0 = t / P_TPR, c = (t % P_TPR) * VEC;
#pragma unroll
    for (int p = 0; p < P_PASSES; ++p) *reinterpret_cast<V*>(Ps + (kk0 + p * P_RPP) * PLD + c) = r.v[p];
  }
  static __device__ __forceinline__ void load_q(QRegs& r, const T* __restrict__ src, int64_t ld) {
    const int t = threadIdx.x;
    const int kk0 = t / Q_TPR, c = (t % Q_TPR) * VEC;
#pragma unroll
    for (int p = 0; p < Q_PASSES; ++p)
      r.v[p] = *reinterpret_cast<const V*>(src + int64_t(kk0 + p * Q_RPP) * ld + c);
  }
  // transposed source: element (k, j) at src[j*ld + k] (contiguous along k).  Prep kernels only.
  static __device__ __forceinline__ void load_q_trans(QRegs& r, const T* __restrict__ src, int64_t ld) {
    const int t = threadIdx.x;
    const int kk0 = t / Q_TPR, c = (t % Q_TPR) * VEC;
#pragma unroll
    for (int p = 0; p < Q_PASSES; ++p)
#pragma unroll
      for (int e = 0; e < VEC; ++e) r.v[p][e] = src[int64_t(c + e) * ld + (kk0 + p * Q_RPP)];
  }
  static __device__ __forceinline__ void store_q(const QRegs& r, T* __restrict__ Qs) {
    const int t = threadIdx.x;
    const int kk0 = t / Q_TPR, c = (t % Q_TPR) * VEC;
#pragma unroll
    for (int p = 0; p < Q_PASSES; ++p) *reinterpret_cast<V*>(Qs + (kk0 + p * Q_RPP) * QLD + c) = r.v[p];
  }
  // coordinates of element e of pass p held by this thread in a Q tile
  static __device__ __forceinline__ void q_coord(int p, int& kk, int& c) {
    const int t = threadIdx.x;
    kk = t / Q_TPR + p * Q_RPP;
    c = (t % Q_TPR) * VEC;
  }

  // --- one BK-deep slab of MFMAs out of LDS ---------------------------------------------------
  static __device__ __forceinline__ void compute(Acc& acc, const T* __restrict__ Ps, const T* __restrict__ Qs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int l15 = lane & 15, lk = lane >> 4;
    const T* pa = Ps + lk * PLD + wr * (MI * 16) + l15;
    const T* pb = Qs + lk * QLD + wc * (NJ * 16) + l15;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      T a[MI], b[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = pa[ks * 4 * PLD + i * 16];
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = pb[ks * 4 * QLD + j * 16];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc.v[i][j] = Mfma16<T>::mma(a[i], b[j], acc.v[i][j]);
    }
  }

  // element (row, col) of acc.v[i][j][r] inside the 128 x NT tile
  static __device__ __forceinline__ int acc_row(int i, int r) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave / WC) * (MI * 16) + i * 16 + Mfma16<T>::row(lane, r);
  }
  static __device__ __forceinline__ int acc_col(int j) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave % WC) * (NJ * 16) + j * 16 + (lane & 15);
  }

  // --- the K loop: QLoad::operator()(step, QRegs&) produces the Q tile of a step ---------------
  // P tile of step t is Pbase + t*BK*ldp.  Ends with all waves past their last LDS read.
  template <typename QLoad>
  static __device__ __forceinline__ void loop(Acc& acc, const T* __restrict__ Pbase, int64_t ldp, int nsteps,
                                              QLoad&& qload, T* __restrict__ smem) {
    if (nsteps <= 0) return;
    PRegs pr;
    QRegs qr;
    load_p(pr, Pbase, ldp);
    qload(0, qr);
    store_p(pr, smem);
    store_q(qr, smem + P_TILE);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
      T* cur = smem + (t & 1) * STAGE;
      T* nxt = smem + ((t + 1) & 1) * STAGE;
      const bool more = (t + 1 < nsteps);
      if (more) {
        load_p(pr, Pbase + int64_t(t + 1) * BK * ldp, ldp);
        qload(t + 1, qr);
      }
      compute(acc, cur, cur + P_TILE);
      if (more) {
        store_p(pr, nxt);
        store_q(qr, nxt + P_TILE);
      }
      __syncthreads();
    }
  }
};

}  // namespace svgp
