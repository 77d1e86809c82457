// Device-side building blocks shared by every kernel of libsvgp_mi355x (gfx950 only).
//   * Mfma16<T>: the 16x16x4 MFMA (f64 / f32-in-f32-acc) with its C/D lane map
//   * TileGemm<T, NT, BK, NTHR>: a 128 x NT output tile per NTHR-thread workgroup, K streamed through
//     double-buffered LDS tiles, operands given as "k-major" matrices (element (k, i) at k*ld + i),
//     which is exactly a column-major Julia matrix read along its columns.
//   * the stationary kernel functions of KernelFunctions.jl (SE, Matern-3/2, Matern-5/2).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "diag.hpp"
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

namespace svgp {

// A note for the next error message of the calling context (the library never prints: include/svgp_mi355x.h).  Host-side
// helpers that have no context at hand (set_max_lds, the SVGP_DEBUG_SYNC checks of prep.hip) leave their diagnosis here;
// KCHECK / HIPC in ctx.hpp append it to svgp_last_error() when the launch they guard fails.
// Per THREAD (a process-global note left by one context's thread surfaced in an unrelated context's next error): a context
// is used by one thread at a time, and the launch helper that leaves a note and the macro that collects it run on that thread.
inline std::string& note_text() { static thread_local std::string s; return s; }
inline void leave_note(const std::string& s) {
  if (note_text().size() < 2048) note_text() += (note_text().empty() ? "" : "; ") + s;
}
inline std::string take_note() {
  std::string s;
  s.swap(note_text());
  return s;
}

// allow a kernel to use up to the whole 160 KiB LDS of a CU as dynamic shared memory.  hipFuncSetAttribute is issued ONCE
// per (kernel, device) and size - not on every launch (it takes a lock inside the runtime): the largest size
// set so far is remembered, a launch that needs no more than that costs one mutex and a short scan.
inline void set_max_lds(const void* fn, hipFuncAttribute attr, int bytes) {
  struct Entry { const void* fn; int dev; int bytes; };
  static std::mutex mu;
  static std::vector<Entry> seen;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(mu);
  for (Entry& e : seen)
    if (e.fn == fn && e.dev == dev) {
      if (bytes <= e.bytes) return;
      if (hipFuncSetAttribute(fn, attr, bytes) == hipSuccess) e.bytes = bytes;
      else leave_note("hipFuncSetAttribute(" + std::to_string(bytes) + " bytes of LDS) failed");
      return;
    }
  const hipError_t err = hipFuncSetAttribute(fn, attr, bytes);
  if (err == hipSuccess) seen.push_back(Entry{fn, dev, bytes});
  else leave_note(std::string("hipFuncSetAttribute(") + std::to_string(bytes) + " bytes of LDS) failed: " + hipGetErrorString(err));
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

// exp(v) for the kernel functions (v <= 0 there).  Cody-Waite reduction by ln2 (hi/lo), degree-13 Taylor
// polynomial on |r| <= ln2/2 (truncation 4e-18), v_ldexp: ~20 VALU instructions, <= 2 ulp; f64 MFMA does not
// co-execute with VALU work, so the Kuf generation inside the strip kernel pays for every instruction.
__device__ __forceinline__ double kexp(double v) {
  const double n = rint(v * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, v);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;            // 1/13!
  p = fma(p, r, 2.08767569878681e-09);          // 1/12!
  p = fma(p, r, 2.505210838544172e-08);         // 1/11!
  p = fma(p, r, 2.755731922398589e-07);         // 1/10!
  p = fma(p, r, 2.7557319223985893e-06);        // 1/9!
  p = fma(p, r, 2.48015873015873e-05);          // 1/8!
  p = fma(p, r, 1.984126984126984e-04);         // 1/7!
  p = fma(p, r, 1.3888888888888889e-03);        // 1/6!
  p = fma(p, r, 8.333333333333333e-03);         // 1/5!
  p = fma(p, r, 4.1666666666666664e-02);        // 1/4!
  p = fma(p, r, 1.6666666666666666e-01);        // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, int(n));                      // n < -1074 flushes to 0, as exp does
}
// fp32: v_exp_f32 on v*log2(e) with the rounding residual of that product applied as a first-order correction
// (5 instructions, <= 2 ulp; libm's expf costs ~12 with its range handling).  Results below the normal range flush to 0.
__device__ __forceinline__ float kexp(float v) {
  const float hi = v * 1.44269504088896341f;
  const float lo = fmaf(v, 1.44269504088896341f, -hi) + v * 1.92596299112661746e-8f;   // log2(e) = hi + lo split
  const float e = __builtin_amdgcn_exp2f(hi);
  return fmaf(e, lo * 0.693147180559945309f, e);
}
// kexp_tab - exp(v), v <= 0, from a 64-entry table tab[j] = 2^(j/64) (LDS) + a degree-5 polynomial: v = (64 m + j) ln2 / 64 + r, |r| <= ln2 / 128: 2^m * tab[j] * (1 + r + ... + r^5 / 120) (truncation 3.5e-17): ~15 VALU
// instructions and one LDS read against kexp's ~20 - the pre-generation is VALU-bound and f64 VALU blocks the MFMA pipe of the
// partner workgroup too (s_memtime stamps: 140-158k of a forward strip's 2230k ticks).  Same box, three repetitions, H strip ms:
// 33.44 / 33.41 / 33.57 with kexp, 33.35 / 33.33 / 33.27 with the table (C2 1.13 vs 1.13): -0.5 %, parity tests unchanged; 212 VGPRs either way
__device__ __forceinline__ double kexp_tab(double v, const double* __restrict__ tab) {
  const double nd = rint(v * 92.332482616893658);               // 64 / ln2
  const int n = int(nd);
  double r = fma(nd, -1.08304246932675596327e-02, v);           // ln2_hi / 64 (kexp's split, exact in binary)
  r = fma(nd, -2.98158582698529328128e-12, r);                  // ln2_lo / 64
  const double t = tab[n & 63];
  double p = fma(r, 8.333333333333333e-03, 4.1666666666666664e-02);
  p = fma(p, r, 1.6666666666666666e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p *= r;                                                       // exp(r) - 1
  return ldexp(fma(t, p, t), n >> 6);
}

__device__ __forceinline__ double ksqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float ksqrt(float v) { return __builtin_amdgcn_sqrtf(v); }   // v_sqrt_f32, 1 ulp

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
// TileGemm: acc(128 x NT) += P(128 x K) * Q(K x NT) on an NTHR-thread workgroup
// ---------------------------------------------------------------------------------------------
template <typename T, int NT, int BK, int NTHR = kThreads, int PAD = 16>
struct TileGemm {
  static constexpr int NB = kNB;
  static constexpr int WR = 2, WC = NTHR / 128;  // wave grid: a wave owns 64 rows x NT/WC columns
  static constexpr int MI = NB / WR / 16;        // 16-row tiles per wave (4)
  static constexpr int NJ = NT / WC / 16;        // 16-col tiles per wave (2 for 128 cols x 8 waves or 64 cols x 4 waves)
  static constexpr int PLD = NB + PAD;           // LDS leading dims: +16 elements makes the 4 k-rows a
  static constexpr int QLD = NT + PAD;           // 64-lane fragment read touches land on disjoint banks
  static constexpr int VEC = Vec16<T>::N;
  using V = typename Vec16<T>::type;
  using acc_t = typename Mfma16<T>::acc_t;

  static constexpr int P_TPR = NB / VEC;                 // threads per k-row of the P tile
  static constexpr int P_RPP = NTHR / P_TPR;             // k-rows per pass
  static constexpr int P_PASSES = BK / P_RPP;
  static constexpr int Q_TPR = NT / VEC;
  static constexpr int Q_RPP = NTHR / Q_TPR;
  static constexpr int Q_PASSES = (BK / Q_RPP) > 0 ? (BK / Q_RPP) : 1;
  static_assert(BK % P_RPP == 0, "BK must be a multiple of the P rows per pass");
  static_assert(BK % Q_RPP == 0 || Q_RPP % BK == 0, "BK and the Q rows per pass must divide one another");
  static constexpr bool Q_PARTIAL = Q_RPP > BK;   // narrow tiles (f32, NT = 32): only the first BK * Q_TPR threads carry Q data

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
  // `src` must be wave-uniform: the per-thread part of the address is a fixed 32-bit byte offset (POff/QOff),
  // so a step costs scalar pointer arithmetic only (f64 MFMA does not co-execute with VALU work).
  struct POff { uint32_t o[P_PASSES]; };
  struct QOff { uint32_t o[Q_PASSES]; };
  static __device__ __forceinline__ POff p_offsets(int64_t ld) {
    POff r;
    const int t = threadIdx.x;
    const int kk0 = t / P_TPR, c = (t % P_TPR) * VEC;
#pragma unroll
    for (int p = 0; p < P_PASSES; ++p) r.o[p] = uint32_t((int64_t(kk0 + p * P_RPP) * ld + c) * sizeof(T));
    return r;
  }
  static __device__ __forceinline__ QOff q_offsets(int64_t ld) {
    QOff r;
    const int t = threadIdx.x;
    const int kk0 = t / Q_TPR, c = (t % Q_TPR) * VEC;
#pragma unroll
    for (int p = 0; p < Q_PASSES; ++p) r.o[p] = uint32_t((int64_t(kk0 + p * Q_RPP) * ld + c) * sizeof(T));
    return r;
  }
  static __device__ __forceinline__ void load_p(PRegs& r, const T* __restrict__ src, const POff& off) {
    if constexpr (diag::ablate<1>) {   // diagnostic build: skip the P-tile loads
      asm volatile("" : "+v"(r.v[0]));
      return;
    }
    const char* base = reinterpret_cast<const char*>(src);
#pragma unroll
    for (int p = 0; p < P_PASSES; ++p) r.v[p] = *reinterpret_cast<const V*>(base + off.o[p]);
  }
  static __device__ __forceinline__ void store_p(const PRegs& r, T* __restrict__ Ps) {
    const int t = threadIdx.x;
    const int kk0 = t / P_TPR, c = (t % P_TPR) * VEC;
#pragma unroll
    for (int p = 0; p < P_PASSES; ++p) *reinterpret_cast<V*>(Ps + (kk0 + p * P_RPP) * PLD + c) = r.v[p];
  }
  // --- P tile global -> LDS without registers (LDS-DMA, global_load_lds_dwordx4): a k-row of the f64 tile is 128 x 8 B =
  // 64 lanes x 16 B, exactly one wave instruction, and the padded LDS rows (PLD) stay legal because no instruction's
  // bytes cross a row.  Wave w moves rows w, w + NW, ...  The LDS address is wave-uniform (M0), the global address is
  // the wave-uniform tile base plus a fixed per-lane 32-bit offset.  Removes the tile's ds_write pass and its staging
  // registers from the step; completion is covered by the vmcnt(0) of the step's closing __syncthreads().
#ifndef SVGP_DMA_P
#define SVGP_DMA_P 1
#endif
#ifndef SVGP_DMA_ASM
#define SVGP_DMA_ASM 1
#endif
  static constexpr int NW = NTHR / 64;
  static constexpr bool kDmaP = SVGP_DMA_P && (NB * sizeof(T) == 1024) && (BK % NW == 0);
  static constexpr int D_ROWS = BK / NW;
  struct DOff { uint32_t o[D_ROWS]; };
  static __device__ __forceinline__ DOff d_offsets(int64_t ld) {
    DOff r;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < D_ROWS; ++q) r.o[q] = uint32_t((int64_t(wave + q * NW) * ld + lane * VEC) * sizeof(T));
    return r;
  }
  // One instruction: LDS destination in M0 (wave-uniform), source = scalar base + 32-bit per-lane offset.  Written as inline
  // assembly because the builtin materialises the full 64-bit address in VGPRs (a v_lshl_add_u64 per instruction - f64 VALU
  // work is MFMA time); the hardware's vmcnt still counts these, the compiler merely does not know (it can only over-wait).
  // M0 is written without the compiler's knowledge: a kernel that uses this loop must hold no other M0 user (the builtin
  // LDS-DMA of the two-buffer loop, ds_gws, s_movrel) - the asynchronous kernels do not.  Same-box A/B: H 33.9 -> 33.3 ms,
  // H32 17.46 -> 16.86, C3 67.5 -> 65.7, C5 4.81 -> 4.74, C2 1.19 -> 1.15 ms; VALU instructions per 32-MFMA step 17 -> 11.
  static __device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, const T* lds) {
#if SVGP_DMA_ASM
    const uint32_t l = uint32_t(uintptr_t((__attribute__((address_space(3))) T*)(lds)));
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(voff), "s"(sbase) : "memory");
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(sbase) + voff),
                                     (__attribute__((address_space(3))) void*)(lds), 16, 0, 0);
#endif
  }
  static __device__ __forceinline__ void dma_p(const T* __restrict__ src, const DOff& off, T* __restrict__ Ps) {
    if constexpr (diag::ablate<1>) return;
    const int wv = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
#pragma unroll
    for (int q = 0; q < D_ROWS; ++q) glds16(src, off.o[q], Ps + (wv + q * NW) * PLD);
  }
  static __device__ __forceinline__ void load_q(QRegs& r, const T* __restrict__ src, const QOff& off) {
    if constexpr (diag::ablate<2>) {   // diagnostic build: skip the Q-tile loads
      asm volatile("" : "+v"(r.v[0]));
      return;
    }
    const char* base = reinterpret_cast<const char*>(src);
    if (Q_PARTIAL && int(threadIdx.x) >= BK * Q_TPR) return;
#pragma unroll
    for (int p = 0; p < Q_PASSES; ++p) r.v[p] = *reinterpret_cast<const V*>(base + off.o[p]);
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
    if (Q_PARTIAL && t >= BK * Q_TPR) return;
#pragma unroll
    for (int p = 0; p < Q_PASSES; ++p) *reinterpret_cast<V*>(Qs + (kk0 + p * Q_RPP) * QLD + c) = r.v[p];
  }
  // coordinates of element e of pass p held by this thread in a Q tile
  static __device__ __forceinline__ void q_coord(int p, int& kk, int& c) {
    const int t = threadIdx.x;
    kk = t / Q_TPR + p * Q_RPP;
    c = (t % Q_TPR) * VEC;
  }

  // --- MFMA fragments: one k-slab (4 k's) of the tile pair in LDS --------------------------------
  struct Frag {
    T a[MI], b[NJ];
  };
  // per-thread element offsets of its fragment inside a P / Q tile (slab 0)
  // Row tiles are interleaved over the two wave rows (wave row wr owns 16-row tiles 2i + wr, i < MI) so that
  // skipping the structurally-zero tiles of a triangular diagonal block stays balanced between them.
  static __device__ __forceinline__ int wave_row() { return __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)) / WC; }
  static __device__ __forceinline__ int frag_a_off() {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (lane >> 4) * PLD + (wave / WC) * 16 + (lane & 15);
  }
  static __device__ __forceinline__ int frag_b_off() {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (lane >> 4) * QLD + (wave % WC) * (NJ * 16) + (lane & 15);
  }
  // fa / fb already point at this thread's fragment of slab 0 of buffer 0; BUF and KSLAB are immediates
  // ILO..IHI (compile time) is the range of this wave's row tiles i that are not structurally zero in the step
  template <int BUF, int KSLAB, int ILO = 0, int IHI = MI - 1>
  static __device__ __forceinline__ void load_frag(Frag& f, const T* __restrict__ fa, const T* __restrict__ fb) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
      if (i >= ILO && i <= IHI) f.a[i] = fa[BUF * STAGE + KSLAB * 4 * PLD + i * 32];
    if (ILO <= IHI) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) f.b[j] = fb[BUF * STAGE + KSLAB * 4 * QLD + j * 16];
    }
  }
  template <int ILO = 0, int IHI = MI - 1>
  static __device__ __forceinline__ void mma_frag(Acc& acc, const Frag& f) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
      if (i >= ILO && i <= IHI) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc.v[i][j] = Mfma16<T>::mma(f.a[i], f.b[j], acc.v[i][j]);
      }
  }

  // element (row, col) of acc.v[i][j][r] inside the 128 x NT tile
  static __device__ __forceinline__ int acc_row(int i, int r) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (2 * i + wave / WC) * 16 + Mfma16<T>::row(lane, r);
  }
  static __device__ __forceinline__ int acc_col(int j) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave % WC) * (NJ * 16) + j * 16 + (lane & 15);
  }

  // Workgroup barriers a thread executes inside one call of the loops below - for kernels that keep PASSENGER waves at the loops'
  // barriers (prep.hip: the 512-thread fused factorisation kernels, whose waves 4-7 only join the block factorisation afterwards):
  // loop / loop_tri: the prologue's, one per step, the closing one; loop_tri_async(_w): the prologue's, one per step but the last, the
  // closing one.  KEEP IN STEP WITH THE LOOPS.  A miscount does NOT deadlock: passengers of a workgroup that does not go on to the
  // block factorisation simply exit, and exited waves drop out of s_barrier; in the workgroup that does go on the passengers' barrier
  // phases would shift into potf2_body's - a SILENT LDS race, caught only by the tests that check the factor
  // itself (every ELBO parity case with 2 <= M / 128 <= 16 runs the 512-thread form; tests/test_gpu_round5.py::test_large_kuu_factorisation
  // and tools/chol_check.py compare the factor).  The counts are spelled as prologue + per-step + closing so that an edit of a loop
  // has one obvious line to change with it.
  static constexpr int kLoopPrologueBarriers = 1, kLoopClosingBarriers = 1;   // loop_tri: store tile 0 | __syncthreads, ..., closing __syncthreads; one per step in step()
  static constexpr int kAsyncPrologueBarriers = 1, kAsyncClosingBarriers = 1; // loop_tri_async_w: wait_barrier before the first fragments, closing __syncthreads; astep: one per step but the last
  static constexpr int loop_barrier_count(int nsteps) { return nsteps <= 0 ? 0 : kLoopPrologueBarriers + nsteps + kLoopClosingBarriers; }
  static constexpr int async_barrier_count(int nsteps) { return nsteps <= 0 ? 0 : kAsyncPrologueBarriers + (nsteps - 1) + kAsyncClosingBarriers; }

  // --- the K loop -----------------------------------------------------------------------------------
  // P tile of step t is Pbase + t*BK*ldp (Pbase wave-uniform); QLoad::operator()(step, QRegs&) produces the
  // Q tile of a step.  Software pipeline, written so that the steady state issues (almost) nothing but
  // MFMA, LDS and global-memory instructions (measured: f64 MFMA never co-executes with VALU work,
  // SQ_VALU_MFMA_COEXEC_CYCLES = 0, so every VALU instruction in the loop is stolen MFMA time):
  //   * fragments ping-pong between two register sets (no copies); the step loop is unrolled by two so the
  //     LDS buffer of a step is a compile-time immediate;
  //   * fragments of slab ks+1 are read from LDS while the MFMAs of slab ks execute;
  //   * the MFMAs of the LAST slab of step t are issued AFTER the barrier that ends the step, behind the first
  //     fragment reads of step t+1, so the write -> barrier -> read latency hides under them;
  //   * global loads of tile t+2 are in flight during step t+1 and land in LDS just before its barrier.
  // Ends with all waves past their last LDS read.
  // NLO/NHI: tile range of the NEXT step (its slab-0 fragments are read behind this step's barrier)
  template <int BUF, int ILO, int IHI, typename QLoad>
  static __device__ __forceinline__ void step(Acc& acc, Frag (&f)[2], PRegs& pr, QRegs& qr, const T* __restrict__ Pbase,
                                              int64_t pstride, const POff& poff, const DOff& doff, int t, int nsteps,
                                              QLoad& qload, T* __restrict__ smem, const T* __restrict__ fa,
                                              const T* __restrict__ fb) {
    constexpr int KS = BK / 4;
    static_assert(KS % 2 == 0, "the fragment ping-pong needs an even number of k-slabs per step");
    const bool more = (t + 1 < nsteps);
#pragma unroll
    for (int ks = 0; ks + 1 < KS; ++ks) {
      if (ks == 0) load_frag<BUF, 1, ILO, IHI>(f[1], fa, fb);
      if (ks == 1) load_frag<BUF, 2, ILO, IHI>(f[0], fa, fb);
      if (ks == 2) load_frag<BUF, 3, ILO, IHI>(f[1], fa, fb);
      if (ks == 3) load_frag<BUF, 4 < KS ? 4 : 0, ILO, IHI>(f[0], fa, fb);
      if (ks == 4) load_frag<BUF, 5 < KS ? 5 : 0, ILO, IHI>(f[1], fa, fb);
      if (ks == 5) load_frag<BUF, 6 < KS ? 6 : 0, ILO, IHI>(f[0], fa, fb);
      if (ks == 6) load_frag<BUF, 7 < KS ? 7 : 0, ILO, IHI>(f[1], fa, fb);
      mma_frag<ILO, IHI>(acc, f[ks & 1]);
    }
    if (more) {
      T* Pn = smem + (BUF ^ 1) * STAGE;
      if constexpr (!kDmaP) store_p(pr, Pn);
      store_q(qr, Pn + P_TILE);
    }
    __syncthreads();   // with LDS-DMA: also the vmcnt(0) that lands the P tile of step t + 1
    if (more) {
      load_frag<BUF ^ 1, 0>(f[0], fa, fb);   // all tiles: the next step's range is not known at compile time
      if (t + 2 < nsteps) {
        if constexpr (kDmaP) {
          // Q first: a generated Q tile consumes its own loads at once, which would drain a DMA issued before it
          qload(t + 2, qr);
          dma_p(Pbase + int64_t(t + 2) * pstride, doff, smem + BUF * STAGE);   // this step's buffer is free again
        } else {
          load_p(pr, Pbase + int64_t(t + 2) * pstride, poff);
          qload(t + 2, qr);
        }
      }
    }
    mma_frag<ILO, IHI>(acc, f[(KS - 1) & 1]);
  }

  // The NB/BK = 8 steps that multiply a triangular diagonal block (lower: T's inv(L_II), upper: U's block of B') are
  // written out with compile-time tile ranges.  A wave row wr owns tiles 2i + wr; lower step SD touches tiles >= SD,
  // i.e. i >= SD/2 for the slower wave row (the barrier makes a step as long as its slower row, so both rows use the
  // same range: no branch on wr); upper step SD touches tiles <= SD, i.e. i <= SD/2.  20 of 32 tile-steps remain.
  // Straight-line code: no per-step dispatch, no per-MFMA predicates (both measured slower).
#define SVGP_DSTEP(B, LO, HI, TT) step<B, LO, HI>(acc, f, pr, qr, Pbase, pstride, poff, doff, (TT), nsteps, qload, smem, fa, fb)

  template <typename QLoad>
  static __device__ __forceinline__ void loop(Acc& acc, const T* __restrict__ Pbase, int64_t ldp, int nsteps,
                                              QLoad&& qload, T* __restrict__ smem) {
    loop_tri<0>(acc, Pbase, ldp, nsteps, qload, smem);
  }

  // TRI = 0: every step is a full tile.  TRI = +1: the LAST NB/BK steps multiply a lower-triangular diagonal block.
  // TRI = -1: the FIRST NB/BK steps multiply an upper-triangular diagonal block.  (nsteps is a multiple of NB/BK.)
  template <int TRI, typename QLoad>
  static __device__ __forceinline__ void loop_tri(Acc& acc, const T* __restrict__ Pbase, int64_t ldp, int nsteps,
                                                  QLoad&& qload, T* __restrict__ smem) {
    if (nsteps <= 0) return;
    static_assert(BK <= 32, "step() enumerates at most 8 k-slabs");
    constexpr int ND = NB / BK;
    static_assert(TRI == 0 || (ND == 8 && MI == 4), "triangular steps are written out for BK = 16, 128-row panels");
    const POff poff = p_offsets(ldp);
    const DOff doff = d_offsets(ldp);
    const int64_t pstride = int64_t(BK) * ldp;
    const T* fa = smem + frag_a_off();
    const T* fb = smem + P_TILE + frag_b_off();
    PRegs pr;
    QRegs qr;
    if constexpr (kDmaP) {
      qload(0, qr);
      dma_p(Pbase, doff, smem);
    } else {
      load_p(pr, Pbase, poff);
      qload(0, qr);
      store_p(pr, smem);
    }
    store_q(qr, smem + P_TILE);
    __syncthreads();
    if (nsteps > 1) {
      if constexpr (kDmaP) {
        qload(1, qr);
        dma_p(Pbase + pstride, doff, smem + STAGE);
      } else {
        load_p(pr, Pbase + pstride, poff);
        qload(1, qr);
      }
    }
    Frag f[2];
    load_frag<0, 0>(f[0], fa, fb);
    int t = 0;
    if (TRI < 0) {
      SVGP_DSTEP(0, 0, 0, 0); SVGP_DSTEP(1, 0, 0, 1); SVGP_DSTEP(0, 0, 1, 2); SVGP_DSTEP(1, 0, 1, 3);
      SVGP_DSTEP(0, 0, 2, 4); SVGP_DSTEP(1, 0, 2, 5); SVGP_DSTEP(0, 0, 3, 6); SVGP_DSTEP(1, 0, 3, 7);
      t = ND;
    }
    const int nreg = (TRI > 0) ? nsteps - ND : nsteps;
    for (; t + 1 < nreg; t += 2) {
      step<0, 0, MI - 1>(acc, f, pr, qr, Pbase, pstride, poff, doff, t, nsteps, qload, smem, fa, fb);
      step<1, 0, MI - 1>(acc, f, pr, qr, Pbase, pstride, poff, doff, t + 1, nsteps, qload, smem, fa, fb);
    }
    if (t < nreg) step<0, 0, MI - 1>(acc, f, pr, qr, Pbase, pstride, poff, doff, t, nsteps, qload, smem, fa, fb);
    if (TRI > 0) {
      SVGP_DSTEP(0, 0, 3, nreg + 0); SVGP_DSTEP(1, 0, 3, nreg + 1); SVGP_DSTEP(0, 1, 3, nreg + 2); SVGP_DSTEP(1, 1, 3, nreg + 3);
      SVGP_DSTEP(0, 2, 3, nreg + 4); SVGP_DSTEP(1, 2, 3, nreg + 5); SVGP_DSTEP(0, 3, 3, nreg + 6); SVGP_DSTEP(1, 3, 3, nreg + 7);
    }
    __syncthreads();  // the tail MFMAs read no LDS, but callers reuse the staging buffers right away
  }
#undef SVGP_DSTEP

  // ==================================================================================================
  // Fully asynchronous K loop (f64, NT = 64): BOTH operand tiles travel global -> LDS by LDS-DMA, through THREE LDS
  // buffers, so a tile has two whole steps to land instead of one; the loop body holds no ordinary vector-memory
  // instruction, no ds_write and no staging register, and the only wait is one counted `s_waitcnt vmcnt(N)` in front
  // of the step's raw `s_barrier` (a `__syncthreads()` would drain the tile still in flight).
  //   * P tile: as dma_p (one 1 KiB k-row per instruction).
  //   * Q tile: its source is a contiguous [BK][NT] block (the scratch strip), k-rows of 512 B; an instruction moves a PAIR
  //     of rows into 1 KiB of LDS.  Rows 2p and 2p+1 then sit 128 dwords apart = the same banks, so the odd row is stored
  //     with its columns XOR 16 (the swizzle is applied to the per-lane SOURCE address; LDS stays linear): the four k-rows
  //     of a fragment read land on disjoint banks, and the fragment of column tile j of an odd row is read from tile j^1.
  // Tile t lives in buffer t % 3.  After the barrier that ends step t buffer t % 3 is free and receives tile t + 3.
  // ==================================================================================================
  // fp32 (128-point strips): the P tile's k-rows are 512 B too and use the same row-pair + XOR-16 image (the fragment of
  // row tile 2i + wr of an odd k-row is read from tile 2i + (wr ^ 1)).
  static constexpr bool kPairP = (NB * sizeof(T) == 512);   // P k-rows of 512 B: row-pair image; 1 KiB rows: padded rows (PLD)
  // Q k-rows of 512 B travel in pairs, of 256 B (fp32, NT = 64: the strips of models with Mp > 2048) four to an instruction:
  // a KiB of LDS then holds QR rows whose banks coincide, so row r of the unit is stored with its 16-column blocks XOR r
  // (swizzle on the per-lane SOURCE address); the four k-rows of a fragment read then hit 64 distinct banks, and the
  // fragment of column tile jt of unit-row r is read from block jt ^ r.
#ifndef SVGP_ASYNC_QUAD
#define SVGP_ASYNC_QUAD 1   // 0: the 256-byte-row tiles (fp32 NT = 64) keep the two-buffer loop (A/B builds)
#endif
  // Q k-rows of 1 KiB (f64, NT = 128: the 512-thread strip of the round-3 A/B) are one row per instruction like the P tile's and
  // get the same +16 padding between rows instead of a swizzle
  static constexpr bool kRowQ = (NT * sizeof(T) == 1024);
  static constexpr int QR = kRowQ ? 1 : ((NT * sizeof(T) == 256) ? 4 : 2);
  static constexpr bool kAsync = (SVGP_DMA_P != 0) && (NB * sizeof(T) == 1024 || kPairP) &&
                                 (NT * sizeof(T) == 512 || (SVGP_ASYNC_QUAD && NT * sizeof(T) == 256) || (kRowQ && NTHR == 512)) &&
                                 BK == 16 && (NTHR == 256 || (kRowQ && NTHR == 512)) && NJ % 2 == 0;
  static constexpr int NBUF = 3;
  static constexpr int QPP = kRowQ ? QLD : QR * NT;         // one row unit (row / pair / quad) of the Q tile in LDS (elements)
  static constexpr int QA_TILE = (BK / QR) * QPP;
  static constexpr int QSLAB = (4 / QR) * QPP;              // LDS distance between the k-slabs (4 k-rows) of a Q tile
  static constexpr int PPP = 2 * NB;                        // one row pair of the P tile (pair image)
  static constexpr int PA_TILE = kPairP ? (BK / 2) * PPP : P_TILE;
  static constexpr int DQ = (BK / QR) / NW;                 // Q instructions per wave and tile
  static constexpr int DP = kPairP ? (BK / 2) / NW : BK / NW;   // P instructions per wave and tile
  static constexpr int DMA_PER_TILE = DP + DQ;              // DMA instructions per wave and tile
  static constexpr size_t ASYNC_LDS_BYTES = size_t(NBUF) * (PA_TILE + QA_TILE) * sizeof(T);
  // Optional per-k weights (the SYRK W = A diag(w) A'): the 16 weights of a tile travel by the same DMA path (one
  // global_load_lds_dword per wave and tile: 256 B, of which the first 16 weights are the tile's; every wave issues it, to the
  // same slot, so that all waves count the same number of DMAs per tile) and scale the Q fragments as they are read.
  static constexpr int W_TILE = 256 / int(sizeof(T));   // elements of one weight slot
  static constexpr size_t ASYNC_W_LDS_BYTES = ASYNC_LDS_BYTES + size_t(NBUF) * 256;
  // dynamic LDS a kernel that may take either loop has to ask for
  static constexpr size_t MAX_LDS_BYTES = (kAsync && ASYNC_LDS_BYTES > LDS_BYTES) ? ASYNC_LDS_BYTES : LDS_BYTES;

  struct AOff { uint32_t p[DP]; uint32_t q[DQ]; };
  // ldq: leading dimension of the Q source (its k-rows are ldq elements apart; NT for a contiguous tile)
  static __device__ __forceinline__ AOff a_offsets(int64_t ldp, int64_t ldq) {
    AOff r;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane >> 5, slot = (lane & 31) * VEC;   // this lane's 16 bytes inside a pair's KiB: row 0/1, column slot
#pragma unroll
    for (int q = 0; q < DP; ++q) {
      if constexpr (kPairP) r.p[q] = uint32_t((int64_t((wave + q * NW) * 2 + row) * ldp + (slot ^ (row * 16))) * sizeof(T));
      else r.p[q] = uint32_t((int64_t(wave + q * NW) * ldp + lane * VEC) * sizeof(T));
    }
    constexpr int LPR = 64 / QR;                           // lanes per k-row of a Q row unit
    const int qrow = lane / LPR, qslot = (lane % LPR) * VEC;
#pragma unroll
    for (int q = 0; q < DQ; ++q)
      r.q[q] = uint32_t((int64_t((wave + q * NW) * QR + qrow) * ldq + (qslot ^ (qrow * 16))) * sizeof(T));
    return r;
  }
  static __device__ __forceinline__ void dma_tile(const T* __restrict__ psrc, const T* __restrict__ qsrc, const AOff& off,
                                                  T* __restrict__ Pb, T* __restrict__ Qb) {
    const int wv = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    if constexpr (!diag::ablate<1>) {   // (diagnostic builds: bit 1 no P-tile DMA, bit 2 no Q-tile DMA - timing only)
#pragma unroll
      for (int q = 0; q < DP; ++q) glds16(psrc, off.p[q], Pb + (wv + q * NW) * (kPairP ? PPP : PLD));
    }
    if constexpr (!diag::ablate<2>) {
#pragma unroll
      for (int q = 0; q < DQ; ++q) glds16(qsrc, off.q[q], Qb + (wv + q * NW) * QPP);
    }
  }
  static __device__ __forceinline__ void dma_w(const T* __restrict__ wsrc, T* __restrict__ Wb) {
    const uint32_t l = uint32_t(uintptr_t((__attribute__((address_space(3))) T*)(Wb)));
    const uint32_t voff = (threadIdx.x & 63) * 4;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" ::"s"(l), "v"(voff), "s"(wsrc) : "memory");
  }
  // all but the wave's N newest vector-memory operations done, every LDS read returned, then the workgroup barrier
  template <int N>
  static __device__ __forceinline__ void wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
  }
  // this thread's fragment origins in buffer 0: column tile j of an odd k-row is read from tile j ^ 1, so even and odd
  // tiles get their own origin (b0: j even, b1: j odd); the same for the row tiles of a pair-image P tile
  struct AFrag { const T* a; const T* b0; const T* b1; };
  static __device__ __forceinline__ AFrag afrag(const T* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, l15 = lane & 15, odd = g & 1;
    AFrag r;
    if constexpr (kPairP) r.a = smem + (g >> 1) * PPP + odd * NB + ((wave / WC) ^ odd) * 16 + l15;
    else r.a = smem + g * PLD + (wave / WC) * 16 + l15;
    // k-row g of a slab is row qr of unit qu; its column tile jt sits in block jt ^ qr
    const int qu = g / QR, qr = g % QR, jt0 = (wave % WC) * NJ;
    const T* qb = smem + NBUF * PA_TILE + qu * QPP + qr * NT + l15;
    r.b0 = qb + ((jt0 + 0) ^ qr) * 16;
    r.b1 = qb + ((jt0 + 1) ^ qr) * 16;
    return r;
  }
  template <int KSLAB, int ILO = 0, int IHI = MI - 1>
  static __device__ __forceinline__ void load_afrag(Frag& f, const T* __restrict__ fa, const T* __restrict__ fb0,
                                                    const T* __restrict__ fb1) {
    constexpr int ASLAB = kPairP ? KSLAB * 2 * PPP : KSLAB * 4 * PLD;
#pragma unroll
    for (int i = 0; i < MI; ++i)
      if (i >= ILO && i <= IHI) f.a[i] = fa[ASLAB + i * 32];
    if (ILO <= IHI) {
      static_assert(NJ % 2 == 0, "column tiles come in even / odd pairs");
      static_assert(QR == 2 || NJ == 2, "the quad image is written for two column tiles per wave");
#pragma unroll
      for (int j = 0; j < NJ; j += 2) {   // (jt0 + j) ^ qr = ((jt0 ^ qr) + j) for even j: the XOR only touches the bits below
        f.b[j] = fb0[KSLAB * QSLAB + j * 16];
        f.b[j + 1] = fb1[KSLAB * QSLAB + j * 16];
      }
    }
  }
  // one k-step on tile t (buffer b, advanced on return); QSrc: t -> wave-uniform pointer to the contiguous Q tile of step t;
  // WSrc: NoWeights, or t -> wave-uniform pointer to the 16 weights of step t (>= 64 dwords readable behind it)
  struct NoWeights { static constexpr bool on = false; __device__ __forceinline__ const T* operator()(int) const { return nullptr; } };
  template <int KSLAB, bool W>
  static __device__ __forceinline__ void scale_b(Frag& f, const T* __restrict__ wl) {
    if constexpr (W) {
      const T wk = wl[KSLAB * 4];
#pragma unroll
      for (int j = 0; j < NJ; ++j) f.b[j] *= wk;
    }
  }
  template <int ILO, int IHI, typename QSrc, typename WSrc>
  static __device__ __forceinline__ void astep(Acc& acc, Frag (&f)[2], const T* __restrict__ Pbase, int64_t pstride,
                                               const AOff& off, int t, int nsteps, QSrc& qsrc, WSrc& wsrc, T* __restrict__ smem,
                                               int& b, const AFrag& fr) {
    constexpr int KS = BK / 4;
    constexpr bool W = WSrc::on;
    constexpr int PER_TILE = DMA_PER_TILE + (W ? 1 : 0);
    static_assert(KS == 4, "written for 16-deep steps");
    const T* fa = fr.a + b * PA_TILE;
    const T* fb0 = fr.b0 + b * QA_TILE;
    const T* fb1 = fr.b1 + b * QA_TILE;
    T* Ws = smem + NBUF * (PA_TILE + QA_TILE);
    const T* wl = Ws + b * W_TILE + ((threadIdx.x & 63) >> 4);   // this lane's k-row of slab 0
    load_afrag<1, ILO, IHI>(f[1], fa, fb0, fb1);
    mma_frag<ILO, IHI>(acc, f[0]);
    scale_b<1, W>(f[1], wl);
    load_afrag<2, ILO, IHI>(f[0], fa, fb0, fb1);
    mma_frag<ILO, IHI>(acc, f[1]);
    scale_b<2, W>(f[0], wl);
    load_afrag<3, ILO, IHI>(f[1], fa, fb0, fb1);
    mma_frag<ILO, IHI>(acc, f[0]);
    scale_b<3, W>(f[1], wl);
    if (t + 1 < nsteps) {
      // tile t + 1 must be in LDS for every wave: of this wave's DMAs only the newest group (tile t + 2) may still fly
      if (t + 2 < nsteps) wait_barrier<PER_TILE>();
      else wait_barrier<0>();
      const int bn = (b + 1 == NBUF) ? 0 : b + 1;
      load_afrag<0>(f[0], fr.a + bn * PA_TILE, fr.b0 + bn * QA_TILE, fr.b1 + bn * QA_TILE);   // all tiles: next range unknown here
      scale_b<0, W>(f[0], Ws + bn * W_TILE + ((threadIdx.x & 63) >> 4));
      if (t + 3 < nsteps) {   // buffer b (tile t) is free now
        dma_tile(Pbase + int64_t(t + 3) * pstride, qsrc(t + 3), off, smem + b * PA_TILE, smem + NBUF * PA_TILE + b * QA_TILE);
        if constexpr (W) dma_w(wsrc(t + 3), Ws + b * W_TILE);
      }
      b = bn;
    }
    mma_frag<ILO, IHI>(acc, f[1]);
  }
  // ---- a weighted twin of astep (the SYRK W = A diag(w) A'), round 4, MEASURED AND REJECTED (kept as an A/B build, SVGP_WSTEP).
  // A separate function, NOT a branch of astep: the strips' kernels, which instantiate the unweighted step, change their register
  // allocation with any edit of it (212 -> 255 VGPRs + 4 spills when this was first written as one function).  The idea: fetch
  // the four weights a lane needs in a step in ONE go behind the barrier and scale the next step's slab-0 fragments AFTER the
  // closing MFMAs.  rocprofv3, H, f64, per 65 536-point chunk (profiles/round4/syrk_weights.md): round-3 step 1267 us, this twin
  // 1402 (with the scheduling barrier) / 1436 (without), a timing-only build with NO weights 1174.  So the weights cost 7 % and
  // it is the 8 v_mul_f64 per step themselves (f64 VALU never co-executes with the MFMA pipe), not where they wait; the product
  // keeps the round-3 step and drops the weights altogether where they are uniform (Gaussian likelihood: grad.hip).
#ifndef SVGP_WSTEP
#define SVGP_WSTEP 0   // 0 = the round-3 step (weights read per slab: the product), 1 = the twin below with the scheduling barrier, 2 = without
#endif
  struct WRegs { T w[4]; };
  static __device__ __forceinline__ void load_w(WRegs& wr, const T* __restrict__ wl) {
#pragma unroll
    for (int q = 0; q < 4; ++q) wr.w[q] = wl[q * 4];
  }
  template <int KSLAB>
  static __device__ __forceinline__ void scale_bw(Frag& f, const WRegs& wr) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) f.b[j] *= wr.w[KSLAB];
  }
  template <typename QSrc, typename WSrc>
  static __device__ __forceinline__ void astep_w(Acc& acc, Frag (&f)[2], WRegs& wr, const T* __restrict__ Pbase, int64_t pstride,
                                                 const AOff& off, int t, int nsteps, QSrc& qsrc, WSrc& wsrc, T* __restrict__ smem,
                                                 int& b, const AFrag& fr) {
    constexpr int PER_TILE = DMA_PER_TILE + 1;
    static_assert(BK / 4 == 4, "written for 16-deep steps");
    const T* fa = fr.a + b * PA_TILE;
    const T* fb0 = fr.b0 + b * QA_TILE;
    const T* fb1 = fr.b1 + b * QA_TILE;
    T* Ws = smem + NBUF * (PA_TILE + QA_TILE);
    load_afrag<1>(f[1], fa, fb0, fb1);
    mma_frag(acc, f[0]);
    scale_bw<1>(f[1], wr);
    load_afrag<2>(f[0], fa, fb0, fb1);
    mma_frag(acc, f[1]);
    scale_bw<2>(f[0], wr);
    load_afrag<3>(f[1], fa, fb0, fb1);
    mma_frag(acc, f[0]);
    scale_bw<3>(f[1], wr);
    if (t + 1 < nsteps) {
      if (t + 2 < nsteps) wait_barrier<PER_TILE>();
      else wait_barrier<0>();
      const int bn = (b + 1 == NBUF) ? 0 : b + 1;
      load_w(wr, Ws + bn * W_TILE + ((threadIdx.x & 63) >> 4));   // this step's weights are all consumed
      load_afrag<0>(f[0], fr.a + bn * PA_TILE, fr.b0 + bn * QA_TILE, fr.b1 + bn * QA_TILE);
      if (t + 3 < nsteps) {   // buffer b (tile t) is free now
        dma_tile(Pbase + int64_t(t + 3) * pstride, qsrc(t + 3), off, smem + b * PA_TILE, smem + NBUF * PA_TILE + b * QA_TILE);
        dma_w(wsrc(t + 3), Ws + b * W_TILE);
      }
      b = bn;
      mma_frag(acc, f[1]);
#ifndef SVGP_WSTEP
#define SVGP_WSTEP 1   // A/B builds: 0 = the round-3 step (weights read per slab), 1 = this with the scheduling barrier, 2 = without it
#endif
#if SVGP_WSTEP == 1
      __builtin_amdgcn_sched_barrier(0);   // the closing MFMAs first: the fresh LDS reads land under them
#endif
      scale_bw<0>(f[0], wr);
    } else {
      mma_frag(acc, f[1]);
    }
  }
#define SVGP_ASTEP(LO, HI, TT) astep<LO, HI>(acc, f, Pbase, pstride, off, (TT), nsteps, qsrc, wsrc, smem, b, fr)
  template <int TRI, typename QSrc>
  static __device__ __forceinline__ void loop_tri_async(Acc& acc, const T* __restrict__ Pbase, int64_t ldp, int nsteps,
                                                        QSrc&& qsrc, T* __restrict__ smem, int64_t ldq = NT) {
    NoWeights nw;
    loop_tri_async_w<TRI>(acc, Pbase, ldp, nsteps, qsrc, nw, smem, ldq);
  }
  template <int TRI, typename QSrc, typename WSrc>
  static __device__ __forceinline__ void loop_tri_async_w(Acc& acc, const T* __restrict__ Pbase, int64_t ldp, int nsteps,
                                                          QSrc&& qsrc, WSrc&& wsrc, T* __restrict__ smem, int64_t ldq = NT) {
    if (nsteps <= 0) return;
    constexpr int ND = NB / BK;
    constexpr bool W = std::remove_reference_t<WSrc>::on;
    constexpr int PER_TILE = DMA_PER_TILE + (W ? 1 : 0);
    static_assert(TRI == 0 || (ND == 8 && MI == 4), "triangular steps are written out for BK = 16, 128-row panels");
    const AOff off = a_offsets(ldp, ldq);
    const int64_t pstride = int64_t(BK) * ldp;
    const AFrag fr = afrag(smem);
    T* Qs = smem + NBUF * PA_TILE;
    T* Ws = smem + NBUF * (PA_TILE + QA_TILE);
    dma_tile(Pbase, qsrc(0), off, smem, Qs);
    if constexpr (W) dma_w(wsrc(0), Ws);
    if (nsteps > 1) {
      dma_tile(Pbase + pstride, qsrc(1), off, smem + PA_TILE, Qs + QA_TILE);
      if constexpr (W) dma_w(wsrc(1), Ws + W_TILE);
    }
    if (nsteps > 2) {
      dma_tile(Pbase + 2 * pstride, qsrc(2), off, smem + 2 * PA_TILE, Qs + 2 * QA_TILE);
      if constexpr (W) dma_w(wsrc(2), Ws + 2 * W_TILE);
    }
    if (nsteps > 2) wait_barrier<2 * PER_TILE>();
    else if (nsteps > 1) wait_barrier<PER_TILE>();
    else wait_barrier<0>();
    Frag f[2];
    load_afrag<0>(f[0], fr.a, fr.b0, fr.b1);
    scale_b<0, W>(f[0], Ws + ((threadIdx.x & 63) >> 4));
    int b = 0, t = 0;
    if (TRI < 0) {
      SVGP_ASTEP(0, 0, 0); SVGP_ASTEP(0, 0, 1); SVGP_ASTEP(0, 1, 2); SVGP_ASTEP(0, 1, 3);
      SVGP_ASTEP(0, 2, 4); SVGP_ASTEP(0, 2, 5); SVGP_ASTEP(0, 3, 6); SVGP_ASTEP(0, 3, 7);
      t = ND;
    }
    const int nreg = (TRI > 0) ? nsteps - ND : nsteps;
    if constexpr (W && TRI == 0 && SVGP_WSTEP != 0) {
      WRegs wr;
      load_w(wr, Ws + ((threadIdx.x & 63) >> 4));   // f[0] was scaled above through scale_b (once per loop)
      for (; t < nreg; ++t) astep_w(acc, f, wr, Pbase, pstride, off, t, nsteps, qsrc, wsrc, smem, b, fr);
    } else {
      for (; t < nreg; ++t) astep<0, MI - 1>(acc, f, Pbase, pstride, off, t, nsteps, qsrc, wsrc, smem, b, fr);
    }
    if (TRI > 0) {
      SVGP_ASTEP(0, 3, nreg + 0); SVGP_ASTEP(0, 3, nreg + 1); SVGP_ASTEP(1, 3, nreg + 2); SVGP_ASTEP(1, 3, nreg + 3);
      SVGP_ASTEP(2, 3, nreg + 4); SVGP_ASTEP(2, 3, nreg + 5); SVGP_ASTEP(3, 3, nreg + 6); SVGP_ASTEP(3, 3, nreg + 7);
    }
    __syncthreads();  // callers reuse the LDS right away; also drains nothing (every DMA was waited for)
  }
#undef SVGP_ASTEP
};

}  // namespace svgp
