// Microbenchmark 2: which ingredient of the strip kernel's step costs MFMA time?  f64, 256 threads x 2 WG/CU.
//  A: MFMA + frag reads (pipelined) + barrier per 32 MFMA
//  B: A + 6 ds_write_b128 per step (the next tile)           C: B + 6 global_load_dwordx4 per step (L2-resident)
//  D: C with LDS sized like the real kernel (61 KB per WG)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int PLD = 144, QLD = 80, STAGE = 16 * (PLD + QLD);

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(const double* __restrict__ g, double* out, int iters) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * STAGE; i += 256) lds[i] = 1e-3 * (i % 7);
  __syncthreads();
  d4 acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = d4{0, 0, 0, 0};
  const double* fa = lds + (lane >> 4) * PLD + (wave / 2) * 64 + (lane & 15);
  const double* fb = lds + 16 * PLD + (lane >> 4) * QLD + (wave % 2) * 32 + (lane & 15);
  double* wp = lds + (tid / 64) * PLD + (tid % 64) * 2;
  double* wq = lds + 16 * PLD + (tid / 32) * QLD + (tid % 32) * 2;
  const char* gp = reinterpret_cast<const char*>(g) + size_t(blockIdx.x % 64) * 65536 + tid * 16;
  d2 st[6];
  for (int p = 0; p < 6; ++p) st[p] = d2{1.0 + p, 2.0};
  double a[2][4], b[2][2];
  for (int i = 0; i < 4; ++i) a[0][i] = fa[i * 16];
  for (int j = 0; j < 2; ++j) b[0][j] = fb[j * 16];
  for (int it = 0; it < iters; ++it) {
    const int cur = (it & 1) * STAGE, nxt = STAGE - cur;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int s = (ks + 1) & 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) a[s][i] = fa[cur + (ks + 1) * 4 * PLD + i * 16];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[s][j] = fb[cur + (ks + 1) * 4 * QLD + j * 16];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks & 1][i], b[ks & 1][j], acc[i * 2 + j], 0, 0, 0);
    }
    if (MODE >= 1) {
#pragma unroll
      for (int p = 0; p < 4; ++p) *reinterpret_cast<d2*>(wp + nxt + p * 4 * PLD) = st[p];
#pragma unroll
      for (int p = 0; p < 2; ++p) *reinterpret_cast<d2*>(wq + nxt + p * 8 * QLD) = st[4 + p];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) a[0][i] = fa[nxt + i * 16];
#pragma unroll
    for (int j = 0; j < 2; ++j) b[0][j] = fb[nxt + j * 16];
    if (MODE >= 2) {
#pragma unroll
      for (int p = 0; p < 6; ++p) st[p] = *reinterpret_cast<const d2*>(gp + ((it * 6 + p) % 16) * 4096);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][i], b[1][j], acc[i * 2 + j], 0, 0, 0);
  }
  double s = 0;
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 123.456) out[0] = s + st[0][0];
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
  double *o, *g;
  hipMalloc(&o, 64); hipMalloc(&g, 64 * 65536 + 65536); hipMemset(g, 0, 64 * 65536 + 65536);
  const int iters = 20000, grid = 512;
  const double flops = double(grid) * 4 * iters * 32.0 * 2048.0;
  for (size_t lds : {size_t(2 * STAGE * 8), size_t(61440)}) {
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    double t0 = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), lds, 0, g, o, iters); });
    double t1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), lds, 0, g, o, iters); });
    double t2 = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), lds, 0, g, o, iters); });
    printf("LDS %6zu B/WG, 2 WG/CU: A(pipelined+barrier) %.1f TF | B(+6 ds_write_b128) %.1f TF | C(+6 global loads) %.1f TF\n", lds,
           flops / t0 / 1e9, flops / t1 / 1e9, flops / t2 / 1e9);
  }
  // one WG per CU for reference
  {
    size_t lds = 61440;
    const double fl1 = double(256) * 4 * iters * 32.0 * 2048.0;
    double t2 = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), lds, 0, g, o, iters); });
    printf("1 WG/CU (1 wave/SIMD): C %.1f TF\n", fl1 / t2 / 1e9);
  }
  return 0;
}
