// Micro-benchmark (round 5, replaces vmem_quads.hip's table): what the vector-memory path delivers to a traversal-like access
// pattern -- every lane of a wave fetches ITS OWN item (64 / 80 / 128 bytes: four / five / eight global_load_dwordx4) from a
// table -- with the residency and the occupancy stated, next to a fully coalesced control.
//
// VERDICT round 4 on vmem_quads.hip: its "L1" case gave every block its own 16 KB table at 6 blocks per CU = 96 KB per CU
// against a 32 KB L1 (not L1-resident); there was no coalesced control and no occupancy sweep.  Here:
//   tables   16 KB  ONE table for all blocks            -> vector L1 (32 KB per CU)
//            2 MB   one table for all blocks            -> L2 (4 MB per XCD)
//            64 MB  one table                           -> Infinity Cache (256 MB)
//            2 GB   one table                           -> HBM
//   patterns own<W>  lane L loads the W 16-byte words of item idx[L]                    (k_trace's load site)
//            dep<W>  the same, but the next index comes out of the item just loaded     (a traversal step: latency x parallelism)
//            coal    instruction j of a step loads 1 KB contiguous (lane L: word j * 64 + L of a random 4 KB page): the control
//   blocks per CU 1 / 2 / 4 / 6 (256 threads each; grid = 256 x k)
// Printed per row: ms, cycles per wave step and CU at the nominal 2.4 GHz, bytes per clock and CU, TB/s over the chip.
// Run under rocprofv3 --pmc (scripts/ubench/run_r5.sh) for TCP_TOTAL_CACHE_ACCESSES / TCP_TCC_READ_REQ / TCP_PENDING_STALL_CYCLES /
// TA_ADDR_STALLED_BY_TC|TD / TD_TD_BUSY / TA_TA_BUSY per kernel (the template arguments are in the kernel names).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <int W, bool DEP>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ tab, float* out, int iters, unsigned mask) {
  float4 acc = make_float4(0, 0, 0, 0);
  unsigned s = (threadIdx.x + blockIdx.x * 256u) * 2654435761u + 12345u;
  unsigned item = (s >> 7) & mask;
  const unsigned salt = s * 747796405u + 2891336453u;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    const float4* g = tab + (size_t)item * W;
    float4 v[W];
#pragma unroll
    for (int k = 0; k < W; k++) v[k] = g[k];
#pragma unroll
    for (int k = 0; k < W; k++) acc.x += v[k].y, acc.y += v[k].z;
    if (DEP) {
      item = (((__float_as_uint(v[0].x) ^ salt) * 2654435761u) >> 7) & mask;  // the next address needs this item (word 0 holds a random index; the
                                                                              // per-lane salt keeps the lanes' chains apart: without it they merge into a few cycles of the random map)
    } else {
      s = s * 1664525u + 1013904223u;
      item = (s >> 7) & mask;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
}
// control: the same bytes per step (4 KB per wave), every instruction one contiguous KB
__global__ __launch_bounds__(256) void coalesced(const float4* __restrict__ tab, float* out, int iters, unsigned mask) {
  const unsigned lane = threadIdx.x & 63u;
  float4 acc = make_float4(0, 0, 0, 0);
  unsigned s = ((threadIdx.x >> 6) + blockIdx.x * 4u) * 2654435761u + 12345u;  // wave-uniform
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    const unsigned page = ((s >> 7) & mask) & ~63u;  // 64 items of 64 B = one 4 KB page
    const float4* g = tab + (size_t)page * 4 + lane;
    const float4 a = g[0], b = g[64], c = g[128], d = g[192];
    acc.x += a.y + b.y, acc.y += c.z + d.z;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  float* d;
  (void)hipMalloc(&d, sizeof(float) * 256 * 8 * 256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  struct Tab { const char* name; size_t bytes; };
  const Tab tabs[4] = {{"16 KB (L1)", 16u << 10}, {"2 MB (L2)", 2u << 20}, {"64 MB (MALL)", 64u << 20}, {"2 GB (HBM)", (size_t)2 << 30}};
  for (const Tab& t : tabs) {
    // the table as 16-byte words; word k of every item: .x = a random index (for dep<W>), the rest 1.0
    const size_t words = t.bytes / 16;
    std::vector<float4> h(words);
    unsigned r = 99991u;
    for (size_t i = 0; i < words; i++) {
      r = r * 1664525u + 1013904223u;
      float4 v = make_float4(0.f, 1.f, 1.f, 1.f);
      unsigned idx = r >> 5;
      memcpy(&v.x, &idx, 4);
      h[i] = v;
    }
    float4* tab;
    if (hipMalloc(&tab, t.bytes) != hipSuccess) {
      printf("table %s: allocation failed\n", t.name);
      continue;
    }
    (void)hipMemcpy(tab, h.data(), t.bytes, hipMemcpyHostToDevice);
    for (int bpc : {1, 2, 4, 6}) {
      if (quick && bpc != 6 && bpc != 1) continue;
      const int blocks = 256 * bpc, iters = t.bytes >= (64u << 20) ? 1500 : 4000;
      for (int mode = 0; mode < 7; mode++) {
        // item size per mode: the number of items in the table is a power of two below bytes / item size
        const int W = (mode == 0 || mode == 3 || mode == 6) ? 4 : ((mode == 1 || mode == 4) ? 5 : 8);
        unsigned items = 1;
        while ((size_t)items * 2 * W * 16 <= t.bytes) items *= 2;
        const unsigned mask = items - 1;
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
          (void)hipEventRecord(e0);
          switch (mode) {
            case 0: hipLaunchKernelGGL((gather<4, false>), dim3(blocks), dim3(256), 0, 0, tab, d, iters, mask); break;
            case 1: hipLaunchKernelGGL((gather<5, false>), dim3(blocks), dim3(256), 0, 0, tab, d, iters, mask); break;
            case 2: hipLaunchKernelGGL((gather<8, false>), dim3(blocks), dim3(256), 0, 0, tab, d, iters, mask); break;
            case 3: hipLaunchKernelGGL((gather<4, true>), dim3(blocks), dim3(256), 0, 0, tab, d, iters, mask); break;
            case 4: hipLaunchKernelGGL((gather<5, true>), dim3(blocks), dim3(256), 0, 0, tab, d, iters, mask); break;
            case 5: hipLaunchKernelGGL((gather<8, true>), dim3(blocks), dim3(256), 0, 0, tab, d, iters, mask); break;
            default: hipLaunchKernelGGL(coalesced, dim3(blocks), dim3(256), 0, 0, tab, d, iters, mask); break;
          }
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
          (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const char* names[7] = {"own  64 B", "own  80 B", "own 128 B", "dep  64 B", "dep  80 B", "dep 128 B", "coalesced 4 KB"};
        const double steps = (double)blocks * 4 * iters, bytes = steps * 64.0 * W * 16.0;
        const double cyc = ms * 1e-3 * 2.4e9 / (steps / 256.0);
        printf("table %-13s %d blocks/CU  %-15s %8.3f ms  %7.1f cycles per wave step per CU  %6.2f B/clk/CU  %6.2f TB/s  %6.1f G items/s\n", t.name, bpc,
               names[mode], ms, cyc, 64.0 * W * 16.0 / cyc, bytes / (ms * 1e-3) / 1e12, steps * 64.0 / (ms * 1e-3) / 1e9);
      }
    }
    (void)hipFree(tab);
  }
  return 0;
}
