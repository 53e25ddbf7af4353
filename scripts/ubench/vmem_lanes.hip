// Micro-benchmark: what does a per-lane 16-byte gather (global_load_dwordx4, every lane its own 64-byte item, cache-resident)
// cost per wave instruction as a function of the number of active lanes?  If the vector-memory path charges a wave
// instruction whatever its EXEC mask, a traversal kernel that runs with half of its lanes pays twice per ray.
// Items: a 16 KB table per block (L1-resident), random item per lane and step; 4 loads (one 64-byte item) per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ tab, float* out, int active, int iters, int table_items) {
  const int lane = threadIdx.x & 63;
  float4 acc = make_float4(0, 0, 0, 0);
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  const float4* base = tab + (size_t)(blockIdx.x % 64) * table_items * 4;
  if (lane < active) {
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
      s = s * 1664525u + 1013904223u;
      const float4* g = base + (size_t)((s >> 8) % (unsigned)table_items) * 4;
      float4 a = g[0], b = g[1], c = g[2], d = g[3];
      acc.x += a.x + b.y, acc.y += c.z + d.w, acc.z += a.w + c.x, acc.w += b.z + d.y;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
  const int blocks = 256 * 8, threads = 256, iters = 4000;
  float* d;
  (void)hipMalloc(&d, sizeof(float) * blocks * threads);
  for (int table_items : {256, 16384}) {   // 16 KB (L1) / 1 MB per table x 64 tables (L2)
    float4* tab;
    std::vector<float> h((size_t)64 * table_items * 16, 1.0f);
    (void)hipMalloc(&tab, h.size() * sizeof(float));
    (void)hipMemcpy(tab, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
      for (int active : {64, 48, 32, 16, 8}) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(gather, dim3(blocks), dim3(threads), 0, 0, tab, d, active, iters, table_items);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)blocks * (threads / 64) * iters * 4;  // dwordx4 wave instructions
        if (rep)
          printf("table %6d items: %2d active lanes: %8.3f ms  %6.2f cycles per wave-instruction per CU (2.4 GHz)  %7.1f GB/s useful\n", table_items,
                 active, ms, ms * 1e-3 * 2.4e9 / (winstr / 256.0), winstr * active * 16 / (ms * 1e-3) / 1e9);
      }
    (void)hipFree(tab);
  }
  return 0;
}
