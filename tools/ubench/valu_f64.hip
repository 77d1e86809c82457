// Microbenchmark: sustained VALU rate of v_fma_f64 / v_fma_f32 on MI355X (wave64, 8 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void __launch_bounds__(256) k(T* out, int iters, T a, T b) {
  T acc[16];
  for (int j = 0; j < 16; ++j) acc[j] = T(threadIdx.x + j);
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = fma(acc[j], a, b);
  T s = 0;
  for (int j = 0; j < 16; ++j) s += acc[j];
  if (s == T(123.456)) out[0] = s;
}
template <typename T> void run(const char* name) {
  T* o; hipMalloc(&o, 64);
  const int iters = 20000, grid = 256 * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<T>, dim3(grid), dim3(256), 0, 0, o, iters, T(0.999), T(0.001)); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<T>, dim3(grid), dim3(256), 0, 0, o, iters, T(0.999), T(0.001)); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fmas = double(grid) * 256 * iters * 16;
  const double wave_instr_per_simd = fmas / 64 / 1024;
  printf("%s: %.1f TFLOP/s, %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, 2 * fmas / ms / 1e9, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
}
int main() { run<double>("v_fma_f64"); run<float>("v_fma_f32"); return 0; }
