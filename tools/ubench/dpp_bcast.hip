// checks the DPP row_newbcast broadcast against v_readlane, right after VALU writes of the source (hazard test)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int J, int NOP>
__device__ __forceinline__ double row_bcast(double v) {
  if constexpr (NOP == 0) asm volatile("s_nop 0" : "+v"(v));
  if constexpr (NOP == 3) asm volatile("s_nop 3" : "+v"(v));
  if constexpr (NOP == 7) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(v));
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + J, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + J, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ float row_bcastf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + J, 0xf, 0xf, true));
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int NOP>
__global__ void k(const double* in, double* out, int* nbad) {
  const int lane = threadIdx.x & 63;
  double v = in[blockIdx.x * 64 + lane];
  int bad = 0;
  for (int it = 0; it < 64; ++it) {
    v = fma(v, 1.0000001, 0.37 * it);           // fresh f64 VALU result right before the broadcast
    const double a = row_bcast<5, NOP>(v);
    const double b = __shfl(v, (lane & 48) + 5);
    if (a != b) ++bad;
    v = fma(a, 0.5, v);
    const double w = v * v;
    const double c = row_bcast<11, NOP>(w);
    const double d = __shfl(w, (lane & 48) + 11);
    if (c != d) ++bad;
    v = v * 0.999 + c * 1e-3;
  }
  out[blockIdx.x * 64 + lane] = v;
  if (bad) atomicAdd(nbad, bad);
}
__global__ void kf(const double* in, double* out, int* nbad) {
  const int lane = threadIdx.x & 63;
  float v = float(in[blockIdx.x * 64 + lane]);
  int bad = 0;
  for (int it = 0; it < 64; ++it) {
    v = fmaf(v, 1.0001f, 0.37f * it);
    const float a = row_bcastf<5>(v);
    const float b = __shfl(v, (lane & 48) + 5);
    if (a != b) ++bad;
    v = fmaf(a, 0.5f, v);
    const float w = v * v;
    const float c = row_bcastf<11>(w);
    const float d = __shfl(w, (lane & 48) + 11);
    if (c != d) ++bad;
    v = v * 0.999f + c * 1e-3f;
  }
  out[blockIdx.x * 64 + lane] = v;
  if (bad) atomicAdd(nbad, bad);
}
int main() {
  const int n = 64 * 1024;
  double *in, *out; int* nb;
  hipMalloc(&in, n * 8); hipMalloc(&out, n * 8); hipMalloc(&nb, 4);
  double* h = (double*)malloc(n * 8);
  for (int i = 0; i < n; ++i) h[i] = drand48() * 3 - 1;
  hipMemcpy(in, h, n * 8, hipMemcpyHostToDevice);
  int bad;
#define RUN(name, kern) hipMemset(nb, 0, 4); hipLaunchKernelGGL(kern, dim3(n / 64), dim3(64), 0, 0, in, out, nb); hipMemcpy(&bad, nb, 4, hipMemcpyDeviceToHost); printf("%s mismatches: %d\n", name, bad);
  RUN("f64 no nop", (k<-1>)); RUN("f64 s_nop 0", (k<0>)); RUN("f64 s_nop 3", (k<3>)); RUN("f64 s_nop 7", (k<7>)); RUN("f32", kf);
  return 0;
}
