// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 rate on MI355X under the
// conditions the strip kernel creates (waves per SIMD, LDS fragment reads + waits between MFMA groups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: pure MFMA, 1: + ds_read + lgkmcnt(0) per 8 MFMA (like compute()), 2: mode 1 + barrier per 32 MFMA
__global__ void __launch_bounds__(512) k64(double* out, int iters) {
  __shared__ double lds[16 * 144 * 2];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 144 * 2; i += blockDim.x) lds[i] = 1e-3 * (i % 7);
  __syncthreads();
  d4 acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = d4{0, 0, 0, 0};
  double a[4] = {1.0 + lane, 2.0, 3.0, 4.0}, b[2] = {0.5, 0.25};
  const double* pa = lds + (lane >> 4) * 144 + (lane & 15);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = pa[ks * 4 * 144 + i * 16];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = pa[16 * 144 + ks * 4 * 144 + j * 16];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
    }
    if (MODE == 2) __syncthreads();
  }
  double s = 0;
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 123.456) out[0] = s;
}

template <int MODE>
__global__ void __launch_bounds__(512) k32(float* out, int iters) {
  __shared__ float lds[16 * 144 * 2];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 144 * 2; i += blockDim.x) lds[i] = 1e-3f * (i % 7);
  __syncthreads();
  f4 acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = f4{0, 0, 0, 0};
  float a[4] = {1.0f + lane, 2.0f, 3.0f, 4.0f}, b[2] = {0.5f, 0.25f};
  const float* pa = lds + (lane >> 4) * 144 + (lane & 15);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = pa[ks * 4 * 144 + i * 16];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = pa[16 * 144 + ks * 4 * 144 + j * 16];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
    }
    if (MODE == 2) __syncthreads();
  }
  float s = 0;
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 123.456f) out[0] = s;
}

template <typename F>
double timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* o; hipMalloc(&o, 64);
  const int iters = 20000;
  for (int threads : {256, 512, 1024}) {
    for (int wgs_per_cu : {1, 2}) {
      if (threads * wgs_per_cu > 1024) continue;
      const int grid = 256 * wgs_per_cu;
      const double flops = double(grid) * (threads / 64) * iters * 32.0 * 2048.0;
      double t0 = timeit([&] { hipLaunchKernelGGL(k64<0>, dim3(grid), dim3(threads), 0, 0, o, iters); });
      double t1 = timeit([&] { hipLaunchKernelGGL(k64<1>, dim3(grid), dim3(threads), 0, 0, o, iters); });
      double t2 = timeit([&] { hipLaunchKernelGGL(k64<2>, dim3(grid), dim3(threads), 0, 0, o, iters); });
      printf("f64 threads %4d x %d WG/CU (%d waves/SIMD): pure %.1f TF | +ds_read/wait %.1f TF | +barrier %.1f TF\n", threads, wgs_per_cu,
             threads * wgs_per_cu / 256, flops / t0 / 1e9, flops / t1 / 1e9, flops / t2 / 1e9);
      double u0 = timeit([&] { hipLaunchKernelGGL(k32<0>, dim3(grid), dim3(threads), 0, 0, (float*)o, iters); });
      double u1 = timeit([&] { hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(threads), 0, 0, (float*)o, iters); });
      double u2 = timeit([&] { hipLaunchKernelGGL(k32<2>, dim3(grid), dim3(threads), 0, 0, (float*)o, iters); });
      printf("f32 threads %4d x %d WG/CU (%d waves/SIMD): pure %.1f TF | +ds_read/wait %.1f TF | +barrier %.1f TF\n", threads, wgs_per_cu,
             threads * wgs_per_cu / 256, flops / u0 / 1e9, flops / u1 / 1e9, flops / u2 / 1e9);
    }
  }
  return 0;
}
