// dtrace_quad.h -- ONE ray per QUAD of lanes: the latency-bound ends of a render (device, Q tree of triangle-only scenes).
//
// Why: small launches and the tail of a chunk are not bound by throughput but by the chain of their longest rays -- (steps of
// the ray) x (latency of a dependent step: fetch the node, ~160 instructions of box tests and ordering on ONE lane, fetch the next
// item) -- while 59 of 64 lanes idle (k_tail: 4.9 lanes per instruction; every k_trace launch below ~64 k rays carries ~0.3 ms
// that does not shrink with its ray count: profiles/README.md).  Here four lanes serve one ray and shorten the chain itself:
//   * lane j of the quad tests child j of the 4-wide node: 6 dequantisations and one slab test instead of 24 and four; the
//     nearest hit child and the push order of the others come from two / three DPP quad permutes instead of a sorting network;
//   * SPECULATION: as soon as the node's child references are known -- before the box tests -- lane j fetches child j's item
//     (the child node, or the leaf's TriPair).  When child a is chosen its item is already on its way in lane a's registers: a
//     chosen leaf is tested by lane a itself (the hit goes to the other three through ds_bpermute), a chosen node is handed to
//     the quad the same way, so the next step starts without a memory round trip of its own.  Four times the bytes -- of launches
//     whose problem is not bytes.
// Measured and not kept (profiles/README.md, round 4): a second register set that keeps the item of the child a lane pushed last
// for the pop that follows (+12 % on the hook kernels: the copies and the ballot cost more instructions than the load they save);
// one loop with one item -- node or leaf -- per turn (+15 %); one leaf-test path for fetched and popped leaves (no change).  What
// bounds a lone wave is the number of instructions it issues (~1 us per step with warm caches), not its loads.
// The stack belongs to the quad (the leader's LDS column; entries are written by the lane that holds the child).  Box tests are
// box_test4q's arithmetic for one child, the leaf test is tri_pair_accept: hits do not depend on the visiting order
// (intersection contract, dtrace.h), so results are bit-identical to every other traversal's.
#pragma once

#include "dtrace.h"

namespace pb {

// quad permutes (DPP): the value of lane (lane & ~3) + perm[lane & 3]
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
constexpr int kQp1032 = 0xB1, kQp2301 = 0x4E, kQp1230 = 0x39, kQp3012 = 0x93;
__device__ __forceinline__ uint32_t quad_min_u32(uint32_t v) {
  uint32_t w = quad_perm<kQp1032>(v);
  v = v < w ? v : w;
  w = quad_perm<kQp2301>(v);
  return v < w ? v : w;
}
__device__ __forceinline__ float quad_from(float v, uint32_t src_lane) { return __shfl(v, (int)src_lane); }
__device__ __forceinline__ uint32_t quad_from(uint32_t v, uint32_t src_lane) { return (uint32_t)__shfl((int)v, (int)src_lane); }
__device__ __forceinline__ float4 quad_from(const float4& v, uint32_t src_lane) {
  return make_float4(quad_from(v.x, src_lane), quad_from(v.y, src_lane), quad_from(v.z, src_lane), quad_from(v.w, src_lane));
}

// the slab test of box_test4q for ONE child (byte j of the quantised bound words), operation for operation
__device__ __forceinline__ void box_test1q(const float4& w0, const float4& w1, const float4& w2, uint32_t j, V3 o, V3 inv, float& ta, float& tb) {
  const float e = 1.52587890625e-05f;
  const uint32_t sh = 8u * j;
  auto q = [sh](float w) { return (float)((__float_as_uint(w) >> sh) & 255u); };
  float p = (__builtin_fmaf(q(w1.z), w0.w, w0.x) - o.x) * inv.x, r = (__builtin_fmaf(q(w2.y), w0.w, w0.x) - o.x) * inv.x;
  float a = __builtin_fminf(p, r), b = __builtin_fmaxf(p, r);
  p = (__builtin_fmaf(q(w1.w), w1.x, w0.y) - o.y) * inv.y, r = (__builtin_fmaf(q(w2.z), w1.x, w0.y) - o.y) * inv.y;
  a = __builtin_fmaxf(a, __builtin_fminf(p, r)), b = __builtin_fminf(b, __builtin_fmaxf(p, r));
  p = (__builtin_fmaf(q(w2.x), w1.y, w0.z) - o.z) * inv.z, r = (__builtin_fmaf(q(w2.w), w1.y, w0.z) - o.z) * inv.z;
  a = __builtin_fmaxf(a, __builtin_fminf(p, r)), b = __builtin_fminf(b, __builtin_fmaxf(p, r));
  ta = __builtin_fmaf(-fabsf(a), e, a), tb = __builtin_fmaf(fabsf(b), e, b);
}

// All four lanes of a quad call this with the SAME ray and stay together.  stack: the quad's LDS column (entry i at
// stack[i * stride]; the same pointer on the four lanes), spill likewise.  Returns true for an any-hit ray that is occluded.
template <int MODE>
__device__ __forceinline__ bool traverse_quad(const DScene& sc, V3 o, V3 d, float tmin, float tmax, Hit& hit, uint32_t* stack, uint32_t stride,
                                              uint32_t* overflow, bool any_rt, uint32_t* spill, uint32_t spill_stride) {
  const bool any_ray = MODE == 2 ? any_rt : (MODE == 1);
  hit.slot = kNone, hit.t = tmax, hit.u = 0.f, hit.v = 0.f;
  if (sc.num_nodes == 0) return false;
  const uint32_t lane = __lane_id(), j = lane & 3u, qbase = lane & ~3u;
  const V3 inv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const float4* const items = sc.wide;
  auto put = [&](uint32_t pos, uint32_t ref) {
    if (pos < (uint32_t)kSimpleLdsStack) stack[pos * stride] = ref;
    else if (pos < (uint32_t)kStackDepth) spill[(pos - (uint32_t)kSimpleLdsStack) * spill_stride] = ref;
    else *overflow = 1u;
  };
  auto get = [&](uint32_t pos) { return pos < (uint32_t)kSimpleLdsStack ? stack[pos * stride] : spill[(pos - (uint32_t)kSimpleLdsStack) * spill_stride]; };
  uint32_t sp = 0u, next = 0u;      // the root is always an inner node
  uint32_t holder = 4u;             // lane of the quad that has prefetched `next`'s item (4: nobody)
  float4 P0 = make_float4(0.f, 0.f, 0.f, 0.f), P1 = P0, P2 = P0, P3 = P0, P4 = P0;  // this lane's prefetched child item
  bool occluded = false;
  for (;;) {
    // ---- `next` is an inner node: its four words on every lane
    float4 N0, N1, N2, N3;
    if (holder < 4u) {
      N0 = quad_from(P0, qbase + holder), N1 = quad_from(P1, qbase + holder), N2 = quad_from(P2, qbase + holder), N3 = quad_from(P3, qbase + holder);
    } else {
      const float4* g = items + 4u * next;
      N0 = g[0], N1 = g[1], N2 = g[2], N3 = g[3];
    }
    const uint32_t ref = __float_as_uint(j == 0u ? N3.x : (j == 1u ? N3.y : (j == 2u ? N3.z : N3.w)));
    // speculation: this lane's child item, before the box test says whether the ray goes there
    if (ref != kEmptyChild) {
      const float4* g = (ref & kLeafBit) ? items + sc.q_tri0 + kTriPairWords * ((ref & 0x3FFFFFFFu) >> 3) : items + 4u * ref;
      P0 = g[0], P1 = g[1], P2 = g[2], P3 = g[3];
      if (ref & kLeafBit) P4 = g[4];
    }
    float ta, tb;
    box_test1q(N0, N1, N2, j, o, inv, ta, tb);
    const bool h = ta <= tb && tb >= tmin && ta <= hit.t && ref != kEmptyChild;
    const int bits = (int)__float_as_uint(ta);
    const uint32_t key = h ? (((uint32_t)(bits < 0 ? 0 : bits) & ~3u) | j) : kWideMiss;
    const uint32_t k1 = quad_perm<kQp1230>(key), k2 = quad_perm<kQp2301>(key), k3 = quad_perm<kQp3012>(key);
    const uint32_t kmin = quad_min_u32(key);
    const uint32_t nhit = (key != kWideMiss ? 1u : 0u) + (k1 != kWideMiss ? 1u : 0u) + (k2 != kWideMiss ? 1u : 0u) + (k3 != kWideMiss ? 1u : 0u);
    if (nhit > 1u) {
      // the other hit children go on the quad's stack, farthest first: each by the lane that holds it
      if (h && key != kmin) {
        const uint32_t farther = (k1 != kWideMiss && k1 > key ? 1u : 0u) + (k2 != kWideMiss && k2 > key ? 1u : 0u) + (k3 != kWideMiss && k3 > key ? 1u : 0u);
        put(sp + farther, ref);
      }
      sp += nhit - 1u;
      if (sp > (uint32_t)kStackDepth) sp = (uint32_t)kStackDepth;
    }
    if (nhit != 0u) {
      holder = kmin & 3u;
      next = quad_from(ref, qbase + holder);
    } else {
      holder = 4u;
      next = kEmptyChild;
    }
    // ---- leaves (and pops) until the next inner node
    for (;;) {
      if (next == kEmptyChild) {
        if (sp == 0u) return occluded;
        sp--;
        next = get(sp);  // (the four lanes read the same word)
        holder = 4u;
      }
      if (!(next & kLeafBit)) break;
      uint32_t nt = 0u;
      bool occ = false;
      if (holder < 4u) {
        // the lane that fetched the leaf tests it; its hit goes to the other three
        Hit hq = hit;
        if (j == holder) occ = tri_pair_accept_s<MODE == 1, false>(sc, P0, P1, P2, P3, P4, o.x, o.y, o.z, d.x, d.y, d.z, inv.x, inv.y, inv.z, tmin, any_ray, hq, nt);
        const uint32_t src = qbase + holder;
        hit.t = quad_from(hq.t, src), hit.u = quad_from(hq.u, src), hit.v = quad_from(hq.v, src), hit.slot = quad_from(hq.slot, src);
        occ = quad_from(occ ? 1u : 0u, src) != 0u;
      } else {
        const float4* g = items + sc.q_tri0 + kTriPairWords * ((next & 0x3FFFFFFFu) >> 3);
        const float4 w0 = g[0], w1 = g[1], w2 = g[2], w3 = g[3], w4 = g[4];
        occ = tri_pair_accept_s<MODE == 1, false>(sc, w0, w1, w2, w3, w4, o.x, o.y, o.z, d.x, d.y, d.z, inv.x, inv.y, inv.z, tmin, any_ray, hit, nt);  // (the four lanes alike)
      }
      if (occ) return true;
      next = kEmptyChild;
    }
  }
}

}  // namespace pb
