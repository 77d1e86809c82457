// diagnostic: what dynamic LDS sizes does this device accept?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) { extern __shared__ float s[]; s[threadIdx.x] = threadIdx.x; __syncthreads(); if (o) o[0] = s[0]; }
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu optin %zu perMP %zu CUs %d\n", p.sharedMemPerBlock, p.sharedMemPerBlockOptin, p.sharedMemPerMultiprocessor, p.multiProcessorCount);
  for (int kb : {64, 65, 72, 80, 128, 132, 160}) {
    hipError_t e1 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), kb * 1024, 0, nullptr);
    hipError_t e2 = hipGetLastError(); hipError_t e3 = hipDeviceSynchronize();
    printf("%d KB: attr=%s launch=%s sync=%s\n", kb, hipGetErrorName(e1), hipGetErrorName(e2), hipGetErrorName(e3));
  }
  return 0;
}
