// Micro-benchmark: 64 random 64-byte items per wave step (a traversal's node fetch), two ways:
//   own : every lane loads ITS item with four global_load_dwordx4 (what k_trace does): each instruction touches 64 different lines
//   quad: four instructions as well, but in instruction j lane L loads part (L & 3) of the item of lane (L >> 2) + 16 j: every
//         aligned group of four lanes reads one contiguous 64-byte item, an instruction touches 16 lines
// If the address path works on (up to) 64 contiguous bytes per cycle the second form needs a quarter of its cycles.
// Tables: 16 KB per block (L1) and 64 x 1 MB (L2).  `xfer`: the quad form followed by the hand-over of the parts to the owning
// lane through LDS (one ds_write_b128 + four ds_read... per lane), i.e. what a kernel would really do.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ tab, float* out, int iters, int table_items) {
  __shared__ float4 xfer[256 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 acc = make_float4(0, 0, 0, 0);
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  const float4* base = tab + (size_t)(blockIdx.x % 64) * table_items * 4;
  float4* my = xfer + wave * 256;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    const unsigned item = (s >> 8) % (unsigned)table_items;
    float4 a, b, c, d;
    if (MODE == 0) {
      const float4* g = base + (size_t)item * 4;
      a = g[0], b = g[1], c = g[2], d = g[3];
    } else {
      const unsigned i0 = (unsigned)__shfl((int)item, (lane >> 2)), i1 = (unsigned)__shfl((int)item, (lane >> 2) + 16);
      const unsigned i2 = (unsigned)__shfl((int)item, (lane >> 2) + 32), i3 = (unsigned)__shfl((int)item, (lane >> 2) + 48);
      a = base[(size_t)i0 * 4 + (lane & 3)], b = base[(size_t)i1 * 4 + (lane & 3)];
      c = base[(size_t)i2 * 4 + (lane & 3)], d = base[(size_t)i3 * 4 + (lane & 3)];
      if (MODE == 2) {  // hand the parts to the owners: part p of item of lane o sits at my[o * 4 + p]
        my[((lane >> 2)) * 4 + (lane & 3)] = a, my[((lane >> 2) + 16) * 4 + (lane & 3)] = b;
        my[((lane >> 2) + 32) * 4 + (lane & 3)] = c, my[((lane >> 2) + 48) * 4 + (lane & 3)] = d;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        a = my[lane * 4 + 0], b = my[lane * 4 + 1], c = my[lane * 4 + 2], d = my[lane * 4 + 3];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    acc.x += a.x + b.y, acc.y += c.z + d.w, acc.z += a.w + c.x, acc.w += b.z + d.y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
  const int blocks = 256 * 6, threads = 256, iters = 4000;
  float* d;
  (void)hipMalloc(&d, sizeof(float) * blocks * threads);
  for (int table_items : {256, 16384}) {
    float4* tab;
    std::vector<float> h((size_t)64 * table_items * 16, 1.0f);
    (void)hipMalloc(&tab, h.size() * sizeof(float));
    (void)hipMemcpy(tab, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
      for (int mode = 0; mode < 3; ++mode) {
        (void)hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(threads), 0, 0, tab, d, iters, table_items);
        else if (mode == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(threads), 0, 0, tab, d, iters, table_items);
        else hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(threads), 0, 0, tab, d, iters, table_items);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double steps = (double)blocks * (threads / 64) * iters;  // wave steps (64 items of 64 bytes each)
        if (rep)
          printf("table %6d items, %s: %8.3f ms  %7.1f cycles per wave step per CU (2.4 GHz)  %7.1f GB/s\n", table_items,
                 mode == 0 ? "own " : (mode == 1 ? "quad" : "xfer"), ms, ms * 1e-3 * 2.4e9 / (steps / 256.0), steps * 4096 / (ms * 1e-3) / 1e9);
      }
    (void)hipFree(tab);
  }
  return 0;
}
