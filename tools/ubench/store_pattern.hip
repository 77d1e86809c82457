// store_pattern.hip — what does the Kuf kernel's store pattern cost by itself?  Write-only kernels over an M x n column-major
// f64 matrix (M = 1024, n = 1e6: 8.19 GB), no arithmetic:
//   fill      : plain grid-stride 16-B stores (1 KiB contiguous per wave instruction): the write roofline of the box
//   kuf       : the Kuf kernel's mapping: a 256-thread workgroup owns 256 rows x 256 columns, a wave instruction writes
//               4 columns x 256 B; consecutive workgroups walk down the rows of the same 256 columns
//   kuf_nt    : the same with nontemporal stores
//   col1k     : a workgroup owns the same 256 x 256 block but a wave instruction writes 1 KiB of ONE column
//               (wave w: rows 0..127 of columns w, w+4, ...; then rows 128..255)
//   col1k_nt  : nontemporal
//   col8k     : a workgroup owns ALL 1024 rows of 64 columns: a wave instruction writes 1 KiB, a workgroup pass 4 KiB,
//               whole 8 KiB columns finished back to back
// build: hipcc -O3 --offload-arch=gfx950 store_pattern.hip -o store_pattern ; run: ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
using V2 = double __attribute__((ext_vector_type(2)));
constexpr int64_t M = 1024, N = 1000000;

template <bool NT> __device__ __forceinline__ void st(V2* p, V2 v) {
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
__global__ void fill_k(V2* K, int64_t n2) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n2; i += int64_t(gridDim.x) * blockDim.x) K[i] = V2{1.0, 2.0};
}
template <bool NT> __global__ void __launch_bounds__(256) kuf_k(double* K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
  const int nI = int(M / 256);
  const int64_t j0 = int64_t(blockIdx.x / nI) * 256, ibase = (int64_t(blockIdx.x % nI) * 4 + wave) * 64;
  for (int jg = 0; jg < 16; ++jg)
    for (int r = 0; r < 4; ++r) {
      const int64_t j = j0 + jg * 16 + kq + 4 * r;
      if (j >= N) continue;
      for (int g = 0; g < 2; ++g) st<NT>(reinterpret_cast<V2*>(K + j * M + ibase + g * 32 + c * 2), V2{double(j), double(g)});
    }
}
template <bool NT> __global__ void __launch_bounds__(256) col1k_k(double* K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nI = int(M / 256);
  const int64_t j0 = int64_t(blockIdx.x / nI) * 256, ibase = int64_t(blockIdx.x % nI) * 256;
  for (int jj = wave; jj < 256; jj += 4) {
    const int64_t j = j0 + jj;
    if (j >= N) break;
    for (int h = 0; h < 2; ++h) st<NT>(reinterpret_cast<V2*>(K + j * M + ibase + h * 128 + lane * 2), V2{double(j), double(h)});
  }
}
template <bool NT> __global__ void __launch_bounds__(256) col8k_k(double* K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j0 = int64_t(blockIdx.x) * 64;
  for (int jj = 0; jj < 64; ++jj) {
    const int64_t j = j0 + jj;
    if (j >= N) break;
    for (int h = 0; h < 2; ++h) st<NT>(reinterpret_cast<V2*>(K + j * M + (h * 4 + wave) * 128 + lane * 2), V2{double(j), double(h)});
  }
}
// the Kuf kernel's per-instruction pattern (4 columns x 256 B) but a workgroup owns ALL M rows of JB columns.
// ORDER 0: for 16-column group { for 256-row chunk { ... } }  (a column's 8 KiB completes within 4 consecutive chunk steps)
// ORDER 1: for 256-row chunk { for 16-column group { ... } }  (z fragments could stay in registers per chunk)
template <int JB, int ORDER> __global__ void __launch_bounds__(256) own_k(double* K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
  const int64_t j0 = int64_t(blockIdx.x) * JB;
  constexpr int NG = JB / 16, NC = int(M / 256);
  for (int o = 0; o < NG * NC; ++o) {
    const int jg = ORDER == 0 ? o / NC : o % NG, rc = ORDER == 0 ? o % NC : o / NG;
    const int64_t ibase = rc * 256 + wave * 64;
    for (int r = 0; r < 4; ++r) {
      const int64_t j = j0 + jg * 16 + kq + 4 * r;
      if (j >= N) continue;
      for (int g = 0; g < 2; ++g) *reinterpret_cast<V2*>(K + j * M + ibase + g * 32 + c * 2) = V2{double(j), double(g)};
    }
  }
}
// a workgroup owns RW rows (a 8*RW-byte piece of every column) of JB columns; consecutive workgroups walk down the rows
template <int JB, int RW> __global__ void __launch_bounds__(256) part_k(double* K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
  constexpr int NR = int(M / RW), NG = JB / 16, NC = RW / 256;
  const int64_t j0 = int64_t(blockIdx.x / NR) * JB, r0 = int64_t(blockIdx.x % NR) * RW;
  for (int jg = 0; jg < NG; ++jg)
    for (int rc = 0; rc < NC; ++rc) {
      const int64_t ibase = r0 + rc * 256 + wave * 64;
      for (int r = 0; r < 4; ++r) {
        const int64_t j = j0 + jg * 16 + kq + 4 * r;
        if (j >= N) continue;
        for (int g = 0; g < 2; ++g) *reinterpret_cast<V2*>(K + j * M + ibase + g * 32 + c * 2) = V2{double(j), double(g)};
      }
    }
}
// own_k<128, 0> plus what the Kuf kernel does between its stores: WHAT bit 1: 8 f64 MFMAs per 8 stores, bit 2: 12 LDS reads,
// bit 4: ~320 dependent-chain f64 FMAs (the exp polynomials).  Which of them slows the store stream?
using A4 = double __attribute__((ext_vector_type(4)));
template <int WHAT, int LDSPAD = 0> __global__ void __launch_bounds__(256) mix_k(double* K, double seed) {
  __shared__ double lds[2048 + LDSPAD];   // LDSPAD limits residency: 2048 + 2048 doubles = 32 KiB -> 5 workgroups per CU, ...
  if (LDSPAD && seed == 123.0) lds[2048 + threadIdx.x] = seed;
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = seed * i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, kq = lane >> 4;
  const int64_t j0 = int64_t(blockIdx.x) * 128;
  for (int o = 0; o < 8 * 4; ++o) {
    const int jg = o / 4, rc = o % 4;
    const int64_t ibase = rc * 256 + wave * 64;
    A4 acc[4];
    for (int b = 0; b < 4; ++b) acc[b] = A4{seed, seed, seed, seed};
    double zb[8];
    for (int q = 0; q < 8; ++q) zb[q] = (WHAT & 2) ? lds[(q * 256 + rc * 64 + lane + jg) & 2047] : seed + q;
    if (WHAT & 1)
      for (int q = 0; q < 2; ++q)
        for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(seed, zb[b * 2 + q], acc[b], 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
      const int64_t j = j0 + jg * 16 + kq + 4 * r;
      if (j >= N) continue;
      for (int g = 0; g < 2; ++g) {
        double v0 = acc[2 * g][r] + zb[g], v1 = acc[2 * g + 1][r] + zb[4 + g];
        if (WHAT & 4)
          for (int t = 0; t < 20; ++t) { v0 = fma(v0, 0.999, 1e-3); v1 = fma(v1, 0.999, 1e-3); }
        *reinterpret_cast<V2*>(K + j * M + ibase + g * 32 + c * 2) = V2{v0, v1};
      }
    }
  }
}
template <typename F> void timeit(const char* name, F launch) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> ts;
  for (int i = 0; i < 12; ++i) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float t; hipEventElapsedTime(&t, a, b); ts.push_back(t); }
  std::sort(ts.begin() + 2, ts.end());
  const float med = ts[2 + 5];
  printf("%-10s median %.3f ms  %.0f GB/s   (min %.3f)\n", name, med, M * N * 8.0 / med / 1e6, ts[2]);
}
int main() {
  double* K; if (hipMalloc(&K, M * N * 8) != hipSuccess) return 1;
  const unsigned gk = unsigned(((N + 255) / 256) * (M / 256));
  timeit("fill", [&] { hipLaunchKernelGGL(fill_k, dim3(256 * 8), dim3(256), 0, 0, reinterpret_cast<V2*>(K), M * N / 2); });
  timeit("kuf", [&] { hipLaunchKernelGGL(kuf_k<false>, dim3(gk), dim3(256), 0, 0, K); });
  timeit("kuf_nt", [&] { hipLaunchKernelGGL(kuf_k<true>, dim3(gk), dim3(256), 0, 0, K); });
  timeit("col1k", [&] { hipLaunchKernelGGL(col1k_k<false>, dim3(gk), dim3(256), 0, 0, K); });
  timeit("col1k_nt", [&] { hipLaunchKernelGGL(col1k_k<true>, dim3(gk), dim3(256), 0, 0, K); });
  timeit("col8k", [&] { hipLaunchKernelGGL(col8k_k<false>, dim3(unsigned((N + 63) / 64)), dim3(256), 0, 0, K); });
  timeit("col8k_nt", [&] { hipLaunchKernelGGL(col8k_k<true>, dim3(unsigned((N + 63) / 64)), dim3(256), 0, 0, K); });
  timeit("own64_o0", [&] { hipLaunchKernelGGL((own_k<64, 0>), dim3(unsigned((N + 63) / 64)), dim3(256), 0, 0, K); });
  timeit("own64_o1", [&] { hipLaunchKernelGGL((own_k<64, 1>), dim3(unsigned((N + 63) / 64)), dim3(256), 0, 0, K); });
  timeit("own128_o0", [&] { hipLaunchKernelGGL((own_k<128, 0>), dim3(unsigned((N + 127) / 128)), dim3(256), 0, 0, K); });
  timeit("own128_o1", [&] { hipLaunchKernelGGL((own_k<128, 1>), dim3(unsigned((N + 127) / 128)), dim3(256), 0, 0, K); });
  timeit("own256_o0", [&] { hipLaunchKernelGGL((own_k<256, 0>), dim3(unsigned((N + 255) / 256)), dim3(256), 0, 0, K); });
  timeit("own256_o1", [&] { hipLaunchKernelGGL((own_k<256, 1>), dim3(unsigned((N + 255) / 256)), dim3(256), 0, 0, K); });
  timeit("part64_512", [&] { hipLaunchKernelGGL((part_k<64, 512>), dim3(unsigned((N + 63) / 64) * 2), dim3(256), 0, 0, K); });
  timeit("part64_256", [&] { hipLaunchKernelGGL((part_k<64, 256>), dim3(unsigned((N + 63) / 64) * 4), dim3(256), 0, 0, K); });
  timeit("part16_256", [&] { hipLaunchKernelGGL((part_k<16, 256>), dim3(unsigned((N + 15) / 16) * 4), dim3(256), 0, 0, K); });
  timeit("part256_512", [&] { hipLaunchKernelGGL((part_k<256, 512>), dim3(unsigned((N + 255) / 256) * 2), dim3(256), 0, 0, K); });
  timeit("mix_none", [&] { hipLaunchKernelGGL((mix_k<0>), dim3(unsigned((N + 127) / 128)), dim3(256), 0, 0, K, 1.0); });
  timeit("mix_mfma", [&] { hipLaunchKernelGGL((mix_k<1>), dim3(unsigned((N + 127) / 128)), dim3(256), 0, 0, K, 1.0); });
  timeit("mix_lds", [&] { hipLaunchKernelGGL((mix_k<2>), dim3(unsigned((N + 127) / 128)), dim3(256), 0, 0, K, 1.0); });
  timeit("mix_fma", [&] { hipLaunchKernelGGL((mix_k<4>), dim3(unsigned((N + 127) / 128)), dim3(256), 0, 0, K, 1.0); });
  timeit("mix_all", [&] { hipLaunchKernelGGL((mix_k<7>), dim3(unsigned((N + 127) / 128)), dim3(256), 0, 0, K, 1.0); });
  timeit("mix_all_5wg", [&] { hipLaunchKernelGGL((mix_k<7, 2048>), dim3(unsigned((N + 127) / 128)), dim3(256), 0, 0, K, 1.0); });
  timeit("mix_all_4wg", [&] { hipLaunchKernelGGL((mix_k<7, 3072>), dim3(unsigned((N + 127) / 128)), dim3(256), 0, 0, K, 1.0); });
  timeit("mix_all_3wg", [&] { hipLaunchKernelGGL((mix_k<7, 4608>), dim3(unsigned((N + 127) / 128)), dim3(256), 0, 0, K, 1.0); });
  timeit("mix_all_2wg", [&] { hipLaunchKernelGGL((mix_k<7, 7168>), dim3(unsigned((N + 127) / 128)), dim3(256), 0, 0, K, 1.0); });
  timeit("own16_o0", [&] { hipLaunchKernelGGL((own_k<16, 0>), dim3(unsigned((N + 15) / 16)), dim3(256), 0, 0, K); });
  timeit("fill", [&] { hipLaunchKernelGGL(fill_k, dim3(256 * 8), dim3(256), 0, 0, reinterpret_cast<V2*>(K), M * N / 2); });
  return 0;
}
