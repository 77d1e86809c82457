// diag.hpp - everything the DIAGNOSTIC builds add to the kernels, in one place (round 6; rounds 1-5 scattered 45 `#if`
// blocks through strip.hip / prep.hip / device_common.hpp).  The product and the experiments builds define none of the macros below:
// diag::ablate<BIT> is false (the `if constexpr` around it folds away) and the stamp macros expand to nothing.
//   tools/build_ablate.sh strip N       -DSVGP_ABLATE=N       timing-only ablations of the strip kernels - WRONG results by construction:
//                                                             bit 1 no P-tile loads / DMA, 2 no Q-tile loads / DMA, 4 no scratch stores of A,
//                                                             16 cheap column sums, 32 no point-major A, 64 no point-major R A, 128 no K-dot,
//                                                             256 the value-and-gradient strips overwrite Kuf by A in place (one scratch strip)
//   tools/build_ablate.sh stripstamps x -DSVGP_STRIP_STAMPS   s_memtime stamps at the phase boundaries of a strip (tools/strip_stamps*.py)
//   tools/build_ablate.sh stamps x      -DSVGP_POTF2_STAMPS   s_memtime stamps inside the block factorisation (tools/potf2_time.py)
// A translation unit that owns stamp storage defines SVGP_DIAG_TU_STRIP / SVGP_DIAG_TU_PREP before including this file.
#pragma once
#include <hip/hip_runtime.h>

namespace svgp {
namespace diag {
#ifdef SVGP_ABLATE
constexpr int kAblate = SVGP_ABLATE;
#else
constexpr int kAblate = 0;
#endif
template <int BIT>
inline constexpr bool ablate = (kAblate & BIT) != 0;
}  // namespace diag
}  // namespace svgp

// ---- strip stamps (strip.hip) -----------------------------------------------------------------------------------------------------
#if defined(SVGP_STRIP_STAMPS) && defined(SVGP_DIAG_TU_STRIP)
namespace svgp {
__device__ unsigned long long g_strip_stamps[128];
__shared__ unsigned long long s_strip_stamps[128];   // accumulated in LDS (a global read-modify-write per stamp costs ~2k cycles)
// per-workgroup timeline of the LAST launch: for workgroups 0, 37, 74, ... (16 of them) the clock at the start of each of its first
// 10 strips, at its end (slot 10), and its XCC id (slot 11): first-strip cost, lockstep, spread over the chip
__device__ unsigned long long g_wg_times[16][12];
}  // namespace svgp
extern "C" int svgp_debug_strip_stamps(unsigned long long* out) {
  return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(svgp::g_strip_stamps), sizeof(svgp::g_strip_stamps)));
}
extern "C" int svgp_debug_wg_times(unsigned long long* out) {
  return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(svgp::g_wg_times), sizeof(svgp::g_wg_times)));
}
// stamp i of the strip in flight: sums over every strip of workgroup 37 but its first
#define SVGP_SSTAMP(i) do { if (stamping && threadIdx.x == 0) s_strip_stamps[i] += clock64(); } while (0)
#define SVGP_SSTAMP_KERNEL_BEGIN()                            \
  int strips_done = 0;                                        \
  if (threadIdx.x < 128) s_strip_stamps[threadIdx.x] = 0;     \
  __syncthreads()
#define SVGP_SSTAMP_STRIP_BEGIN()                                                                                          \
  const bool stamping = (blockIdx.x == 37 && strips_done >= 1);                                                            \
  if (stamping && threadIdx.x == 0) s_strip_stamps[127] += 1;                                                              \
  if (threadIdx.x == 0 && blockIdx.x % 37 == 0 && blockIdx.x / 37 < 16 && strips_done < 10)                                \
    g_wg_times[blockIdx.x / 37][strips_done] = clock64();                                                                  \
  ++strips_done
#define SVGP_SSTAMP_KERNEL_END()                                                                                           \
  do {                                                                                                                     \
    if (blockIdx.x == 37 && threadIdx.x < 128) g_strip_stamps[threadIdx.x] = s_strip_stamps[threadIdx.x];                  \
    if (threadIdx.x == 0 && blockIdx.x % 37 == 0 && blockIdx.x / 37 < 16) {                                                \
      g_wg_times[blockIdx.x / 37][10] = clock64();                                                                         \
      g_wg_times[blockIdx.x / 37][11] = (unsigned long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf) | \
                                        ((unsigned long long)strips_done << 8); /* HW_REG_XCC_ID */                        \
      for (int q = strips_done; q < 10; ++q) g_wg_times[blockIdx.x / 37][q] = 0;                                           \
    }                                                                                                                      \
  } while (0)
#else
#define SVGP_SSTAMP(i)
#define SVGP_SSTAMP_KERNEL_BEGIN()
#define SVGP_SSTAMP_STRIP_BEGIN()
#define SVGP_SSTAMP_KERNEL_END()
#endif

// ---- block-factorisation stamps (prep.hip) ----------------------------------------------------------------------------------------
#if defined(SVGP_POTF2_STAMPS) && defined(SVGP_DIAG_TU_PREP)
namespace svgp {
__device__ unsigned long long g_potf2_stamps[128];
}
extern "C" int svgp_debug_potf2_stamps(unsigned long long* out) {
  return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(svgp::g_potf2_stamps), sizeof(svgp::g_potf2_stamps)));
}
#define SVGP_STAMP(i) do { if (threadIdx.x == 0) g_potf2_stamps[i] = clock64(); } while (0)
#define SVGP_STAMPW(i) do { if (threadIdx.x == 64) g_potf2_stamps[i] = clock64(); } while (0)   // a worker wave
#else
#define SVGP_STAMP(i)
#define SVGP_STAMPW(i)
#endif
