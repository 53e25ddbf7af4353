// bvh_gpu.hip -- BVH build on the GPU: the alternative to the host binned-SAH builder
// (bvh_build.cpp) for edit -> re-render loops and scenes with millions of primitives (a 4.8 M-piece hair scene).
// Replaces the same call as the host builder: Embree's rtcCommitScene behind Scene::CommitScene
// (src/raytracer/raytracer_impl.cc:136-147,181-192, src/scene.cc:96-104).
//
// Linear BVH: 63-bit Morton codes of the primitive-box centres (21 bits per axis over the centroid bounds), one radix
// sort, Karras' parallel hierarchy (one thread per internal node, ties between equal codes broken by position),
// bottom-up boxes with one atomic counter per node, then emission in the traversal format of dscene.h -- 64-byte nodes
// holding both children's boxes, leaves of <= kMaxLeaf primitives of one kind (a subtree that small and uniform is
// emitted as a leaf of its parent).  Hits do not depend on the shape of the tree (intersection contract, dtrace.h), so
// a frame rendered over this tree is bit-identical to one rendered over the SAH tree; only the traversal cost differs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "host_scene.h"

namespace pb {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t float_ordered(float f) {  // monotone float -> uint
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_float(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// centroid bounds: cb[0..2] = min (ordered uint), cb[3..5] = max
__global__ void k_lbvh_bounds(const float* __restrict__ lo, const float* __restrict__ hi, uint32_t n, uint32_t* cb) {
  __shared__ uint32_t smin[3], smax[3];
  if (threadIdx.x < 3) smin[threadIdx.x] = 0xFFFFFFFFu, smax[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    for (int a = 0; a < 3; a++) {
      const uint32_t c = float_ordered(0.5f * (lo[3 * (size_t)i + a] + hi[3 * (size_t)i + a]));
      mn[a] = c < mn[a] ? c : mn[a], mx[a] = c > mx[a] ? c : mx[a];
    }
  for (int a = 0; a < 3; a++) atomicMin(&smin[a], mn[a]), atomicMax(&smax[a], mx[a]);
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&cb[threadIdx.x], smin[threadIdx.x]), atomicMax(&cb[3 + threadIdx.x], smax[threadIdx.x]);
}

__device__ __forceinline__ uint64_t spread21(uint64_t x) {  // bit i -> bit 3i
  x &= 0x1FFFFFull;
  x = (x | (x << 32)) & 0x1F00000000FFFFull;
  x = (x | (x << 16)) & 0x1F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

__global__ void k_lbvh_keys(const float* __restrict__ lo, const float* __restrict__ hi, uint32_t n, const uint32_t* cb,
                            uint64_t* keys, uint32_t* ids) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t code = 0;
  for (int a = 0; a < 3; a++) {
    const float mn = ordered_float(cb[a]), mx = ordered_float(cb[3 + a]);
    const float c = 0.5f * (lo[3 * (size_t)i + a] + hi[3 * (size_t)i + a]);
    const float ext = mx - mn;
    float q = ext > 0.f ? (c - mn) / ext : 0.f;
    q = q < 0.f ? 0.f : (q > 1.f ? 1.f : q);
    uint32_t cell = (uint32_t)(q * 2097151.0f);
    code |= spread21(cell) << (2 - a);
  }
  keys[i] = code, ids[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j; equal keys continue with the positions
__device__ __forceinline__ int delta(const uint64_t* __restrict__ k, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = k[i], b = k[j];
  if (a != b) return __clzll((long long)(a ^ b));
  return 64 + __clz(i ^ j);
}

// Karras 2012, one thread per internal node i in [0, n-2].  Children: index < n-1 -> internal node, else leaf (idx-(n-1)).
__global__ void k_lbvh_hierarchy(const uint64_t* __restrict__ keys, int n, uint32_t* left, uint32_t* right, uint32_t* parent,
                                 uint32_t* first, uint32_t* last) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t == 1) break;
  }
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo_i = i < j ? i : j, hi_i = i < j ? j : i;
  const uint32_t lc = (lo_i == gamma) ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;
  const uint32_t rc = (hi_i == gamma + 1) ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
  left[i] = lc, right[i] = rc;
  first[i] = (uint32_t)lo_i, last[i] = (uint32_t)hi_i;
  parent[lc] = (uint32_t)i, parent[rc] = (uint32_t)i;
  if (i == 0) parent[0] = kNone;
}

// Bottom-up: box, uniform-kind flag (0/1 = all of that kind, 2 = mixed) and depth per node; the second thread to arrive
// at a node processes it.  Arrays are indexed like `parent` (internal 0..n-2, leaves n-1..2n-2).
__global__ void k_lbvh_fit(const float* __restrict__ lo, const float* __restrict__ hi, const uint8_t* __restrict__ kinds,
                           const uint32_t* __restrict__ ids, int n, const uint32_t* __restrict__ left,
                           const uint32_t* __restrict__ right, const uint32_t* __restrict__ parent, float* box, uint8_t* kind,
                           uint32_t* height, uint32_t* visits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = ids[i];
  uint32_t node = (uint32_t)(n - 1 + i);
  for (int a = 0; a < 3; a++) box[6 * (size_t)node + a] = lo[3 * (size_t)g + a], box[6 * (size_t)node + 3 + a] = hi[3 * (size_t)g + a];
  kind[node] = kinds[g], height[node] = 0u;
  __threadfence();
  for (uint32_t p = parent[node]; p != kNone; p = parent[p]) {
    if (atomicAdd(&visits[p], 1u) == 0u) return;  // the sibling subtree is not finished: its thread continues
    __threadfence();
    const uint32_t l = left[p], r = right[p];
    for (int a = 0; a < 3; a++) {
      box[6 * (size_t)p + a] = fminf(box[6 * (size_t)l + a], box[6 * (size_t)r + a]);
      box[6 * (size_t)p + 3 + a] = fmaxf(box[6 * (size_t)l + 3 + a], box[6 * (size_t)r + 3 + a]);
    }
    const uint8_t kl = kind[l], kr = kind[r];
    kind[p] = (kl == kr) ? kl : (uint8_t)2;
    const uint32_t hl = height[l], hr = height[r];
    height[p] = 1u + (hl > hr ? hl : hr);
    __threadfence();
  }
}

// traversal nodes: node i of the hierarchy -> BvhNode i; a child subtree of <= kMaxLeaf primitives of one kind is a leaf
__global__ void k_lbvh_emit(int n, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                            const uint32_t* __restrict__ first, const uint32_t* __restrict__ last, const float* __restrict__ box,
                            const uint8_t* __restrict__ kind, BvhNode* nodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  BvhNode nd;
  nd.pad[0] = nd.pad[1] = 0u;
  const uint32_t ch[2] = {left[i], right[i]};
  uint32_t ref[2];
  for (int c = 0; c < 2; c++) {
    const uint32_t k = ch[c];
    for (int a = 0; a < 3; a++) {  // stored widened, like BvhNode::set_box
      const float l = box[6 * (size_t)k + a], h = box[6 * (size_t)k + 3 + a];
      nd.lo[a][c] = l - (fabsf(l) * 1.52587890625e-05f + 1e-30f);
      nd.hi[a][c] = h + (fabsf(h) * 1.52587890625e-05f + 1e-30f);
    }
    uint32_t f, cnt;
    if (k >= (uint32_t)(n - 1)) f = k - (uint32_t)(n - 1), cnt = 1u;
    else f = first[k], cnt = last[k] - first[k] + 1u;
    if (cnt <= (uint32_t)kMaxLeaf && kind[k] != 2) ref[c] = kLeafBit | (kind[k] ? kCurveBit : 0u) | (f << 3) | (cnt - 1u);
    else ref[c] = k;
  }
  nd.c0 = ref[0], nd.c1 = ref[1];
  nodes[i] = nd;
}

#define GPU_CHK(x)                 \
  do {                             \
    hipError_t e_ = (x);           \
    if (e_ != hipSuccess) return e_; \
  } while (0)

template <typename T>
struct Tmp {
  T* p = nullptr;
  ~Tmp() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

}  // namespace

// lo/hi/kinds: host arrays of n primitive boxes.  nodes_out: device array of max(n-1, 1) BvhNode (allocated by the
// caller); order_out: host, slot -> primitive index; depth_out: stack depth a traversal needs.
hipError_t build_bvh_gpu(hipStream_t st, const std::vector<float>& lo, const std::vector<float>& hi,
                         const std::vector<uint8_t>& kinds, BvhNode* nodes_out, std::vector<uint32_t>* order_out,
                         uint32_t* depth_out) {
  const uint32_t n = (uint32_t)kinds.size();
  order_out->resize(n);
  *depth_out = 1;
  if (n == 0) return hipSuccess;
  if (n == 1) {  // a single leaf under the root (c1 empty), like the host builder
    BvhNode nd;
    memset(&nd, 0, sizeof(nd));
    const float nan3[3] = {NAN, NAN, NAN};
    nd.set_box(0, lo.data(), hi.data());
    nd.set_box(1, nan3, nan3);
    nd.c0 = kLeafBit | (kinds[0] ? kCurveBit : 0u) | 0u;
    nd.c1 = kEmptyChild;
    (*order_out)[0] = 0;
    GPU_CHK(hipMemcpyAsync(nodes_out, &nd, sizeof(nd), hipMemcpyHostToDevice, st));
    return hipStreamSynchronize(st);
  }
  Tmp<float> d_lo, d_hi, d_box;
  Tmp<uint8_t> d_kinds, d_kind;
  Tmp<uint32_t> d_cb, d_ids, d_ids2, d_left, d_right, d_parent, d_first, d_last, d_height, d_visits;
  Tmp<uint64_t> d_keys, d_keys2;
  GPU_CHK(d_lo.alloc(3 * (size_t)n)); GPU_CHK(d_hi.alloc(3 * (size_t)n)); GPU_CHK(d_kinds.alloc(n));
  GPU_CHK(d_cb.alloc(6)); GPU_CHK(d_ids.alloc(n)); GPU_CHK(d_ids2.alloc(n)); GPU_CHK(d_keys.alloc(n)); GPU_CHK(d_keys2.alloc(n));
  GPU_CHK(d_left.alloc(n)); GPU_CHK(d_right.alloc(n)); GPU_CHK(d_first.alloc(n)); GPU_CHK(d_last.alloc(n));
  GPU_CHK(d_parent.alloc(2 * (size_t)n)); GPU_CHK(d_height.alloc(2 * (size_t)n)); GPU_CHK(d_kind.alloc(2 * (size_t)n));
  GPU_CHK(d_box.alloc(12 * (size_t)n)); GPU_CHK(d_visits.alloc(n));
  GPU_CHK(hipMemcpyAsync(d_lo.p, lo.data(), 12 * (size_t)n, hipMemcpyHostToDevice, st));
  GPU_CHK(hipMemcpyAsync(d_hi.p, hi.data(), 12 * (size_t)n, hipMemcpyHostToDevice, st));
  GPU_CHK(hipMemcpyAsync(d_kinds.p, kinds.data(), n, hipMemcpyHostToDevice, st));
  const uint32_t init_cb[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  GPU_CHK(hipMemcpyAsync(d_cb.p, init_cb, sizeof(init_cb), hipMemcpyHostToDevice, st));
  GPU_CHK(hipMemsetAsync(d_visits.p, 0, sizeof(uint32_t) * n, st));
  const uint32_t grid = (n + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(k_lbvh_bounds, dim3(grid < 2048u ? grid : 2048u), dim3(kThreads), 0, st, d_lo.p, d_hi.p, n, d_cb.p);
  hipLaunchKernelGGL(k_lbvh_keys, dim3(grid), dim3(kThreads), 0, st, d_lo.p, d_hi.p, n, d_cb.p, d_keys.p, d_ids.p);
  size_t tmp_bytes = 0;
  GPU_CHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys.p, d_keys2.p, d_ids.p, d_ids2.p, n, 0, 63, st));
  Tmp<unsigned char> d_tmp;
  GPU_CHK(d_tmp.alloc(tmp_bytes));
  GPU_CHK(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_keys.p, d_keys2.p, d_ids.p, d_ids2.p, n, 0, 63, st));
  hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(grid), dim3(kThreads), 0, st, d_keys2.p, (int)n, d_left.p, d_right.p, d_parent.p,
                     d_first.p, d_last.p);
  hipLaunchKernelGGL(k_lbvh_fit, dim3(grid), dim3(kThreads), 0, st, d_lo.p, d_hi.p, d_kinds.p, d_ids2.p, (int)n, d_left.p, d_right.p,
                     d_parent.p, d_box.p, d_kind.p, d_height.p, d_visits.p);
  hipLaunchKernelGGL(k_lbvh_emit, dim3(grid), dim3(kThreads), 0, st, (int)n, d_left.p, d_right.p, d_first.p, d_last.p, d_box.p,
                     d_kind.p, nodes_out);
  GPU_CHK(hipGetLastError());
  uint32_t root_height = 0;
  GPU_CHK(hipMemcpyAsync(order_out->data(), d_ids2.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
  GPU_CHK(hipMemcpyAsync(&root_height, d_height.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  GPU_CHK(hipStreamSynchronize(st));
  *depth_out = root_height + 1;
  return hipSuccess;
}

}  // namespace pb
