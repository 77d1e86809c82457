// What does s_memtime count, and at what clock does a lone workgroup run?  One wave runs N dependent v_fma_f32 (known issue cost:
// 4 cycles each on a wave64... the dependent latency may be more); the kernel's duration comes from HIP events, the s_memtime delta
// from the kernel.  Runs with 1 workgroup and with one workgroup per CU (the chip busy), for long (ms) kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void chain(float* out, long long* ticks, int n) {
  float v = threadIdx.x * 1e-3f;
  const long long t0 = __builtin_readcyclecounter();
  const long long m0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) v = fmaf(v, 1.0000001f, 1e-7f);
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long m1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
  if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = m1 - m0; }
}
int main() {
  float* out; long long* ticks;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&ticks, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1, 1, 256, 1024, 1}) {
    for (int n : {2000, 20000}) {
      hipLaunchKernelGGL(chain, dim3(grid), dim3(64), 0, 0, out, ticks, n);   // warm
      hipDeviceSynchronize();
      hipEventRecord(e0); hipLaunchKernelGGL(chain, dim3(grid), dim3(64), 0, 0, out, ticks, n); hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
      const double fmas = 64.0 * n;
      printf("grid %4d n %6d: %.3f ms | readcyclecounter %lld (%.2f per fma, %.0f MHz) | s_memtime %lld (%.2f per fma, %.0f MHz) | ns per dependent fma %.2f\n",
             grid, n, ms, h[0], h[0] / fmas, h[0] / (ms * 1e3), h[1], h[1] / fmas, h[1] / (ms * 1e3), ms * 1e6 / fmas);
    }
  }
  return 0;
}
