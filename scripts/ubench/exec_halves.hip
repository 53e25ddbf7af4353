// Micro-benchmark: does a wave64 VALU instruction cost less when one 32-lane half of EXEC is empty?
// Three masks over the same dependent FMA chain: all 64 lanes, lanes 0..31 only, even lanes only (32 lanes spread).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, int mode, int iters) {
  const int lane = threadIdx.x & 63;
  const bool on = mode == 0 ? true : (mode == 1 ? lane < 32 : (lane & 1) == 0);
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999f, c = 1e-3f;
  if (on) {
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 64; ++k) a = __builtin_fmaf(a, b, c);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
  float* d;
  const int blocks = 256 * 8, threads = 256, iters = 2000;
  hipMalloc(&d, sizeof(float) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const char* names[3] = {"all 64 lanes", "lanes 0..31", "even lanes"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, d, mode, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-14s %.3f ms\n", names[mode], ms);
    }
  return 0;
}
