// dtrace_wp.h -- wave-pooled traversal: every wave keeps kWpRays rays in its OWN slice of LDS and works on them in full batches
// of one phase (device, Q tree, triangle-only scenes).
//
// Why: the phase-voting traversal (dtrace_pv.h) binds a ray to the registers of one lane, so a turn serves only the lanes
// whose ray is in the voted phase (36.7 of 64 lanes per node turn, 22.4 per triangle turn on C2) and the kernel is bound by
// the wave-instructions it issues.  Two rays per lane (dtrace_pv2.h) and a block-wide pool with shared queues (dtrace_q.h)
// were measured first: the former loses its 19 % more lanes to the exchange instructions and the lower occupancy, the latter
// reaches 57 lanes per step but pays ~8 dependent LDS round trips of queue protocol per step between sixteen contending waves
// (VALU pipes 44 % busy).  Here the pool is PRIVATE to a wave: no other wave touches it, so the queues are plain rings of
// 8-bit slot numbers whose heads and tails live in scalar registers -- no atomics, no sentinels, no spinning -- and a step is
//     take up to 64 tickets of one ring -> read those rays' state (origin, 1 / d, interval, current item, stack pointer) from
//     LDS -> fetch the item -> one node step / one triangle test for every lane -> write back -> append each ray's ticket to
//     the ring of its next phase (ballot + mbcnt rank).
// A ray is held by no lane between two steps, so every step runs with min(64, ring length) lanes whatever phases the other
// rays are in.  Per ray 16 + 16 + 16 bytes of state and kWpStack stack entries in LDS; the direction (needed by triangle tests
// only) is fetched again from the ray's record in HBM, and an accepted hit's (t, u, v, code) is written through to the hit
// record (Sink::accept) instead of being kept (Sink::kKeepUV: kept in LDS, for sinks that want the whole hit at the end).
// Hits do not depend on the visiting order (intersection contract, dtrace.h): bit-identical results.
//
// MEASURED (round 4, C2 at 8 spp, same box, k_trace ms): phase-voting 7.7 | this file 8.7 (node / triangle steps at 50.5 / 46.9 of
// 64 lanes, 36 % fewer steps than the phase-voting kernel has turns, but 246 VALU instructions per step against 190 per turn --
// ticket, state and ring traffic -- and 12 waves per CU against 24: VALU pipes 47 % busy) | one step = node batch + triangle
// batch + refill with their loads in flight together 8.9 | two batches per wave software-pipelined (168 VGPRs) 10.4 | rays per
// wave 112 / 140 / 192, LDS stack 4 / 8, refill at 32: 8.7 - 9.2.  Not selected by default (PBRHIP_TRACEWP=1); kept as the
// measured alternative, covered by tests/test_gpu_parity.py::test_alternative_traversals_bit_exact.  See profiles/README.md.
#pragma once

#include "dtrace_pv.h"

namespace pb {

static_assert(kWpRays <= 256 && kWpRays >= 64, "tickets are bytes; a step wants 64 rays");
enum : uint32_t { kWNode = 0, kWTri = 1, kWDone = 2, kWFree = 3 };
// meta word of a ray: stack pointer | primitives of the current leaf still to test after the current one | flags
constexpr uint32_t kWmSp = 0xFFu, kWmRemShift = 8u, kWmRem = 7u << 8, kWmAny = 1u << 11, kWmOccluded = 1u << 12;

template <bool UV>
struct alignas(16) WavePool {
  float4 a0[kWpRays];                  // origin, tmin
  float4 a1[kWpRays];                  // 1 / d, current hit distance (the ray's tmax)
  uint4 m[kWpRays];                    // current item (16-byte index into DScene::wide), meta, the ray's tag, hit code
  uint32_t stack[kWpStack * kWpRays];  // entry i of ray r: stack[i * kWpRays + r]
  float2 uv[UV ? kWpRays : 1];         // the hit's (u, v) when the sink wants it at the end
  uint8_t ring[4][256];                // tickets (slot numbers) per phase
};

__device__ __forceinline__ uint32_t wp_rank(unsigned long long m) {
  return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// lanes of one wave hand data to each other through LDS: the hardware executes a wave's DS operations in order, the compiler
// must not reorder them across this point either
__device__ __forceinline__ void wp_sync() { asm volatile("" ::: "memory"); }

// Sink (besides load(), dtrace_pv.h):
//   V3   dir(tag)                          the ray's direction again (its record in HBM)
//   void accept(tag, t, u, v, code)        a closest-hit ray has a new nearest hit (!kKeepUV)
//   void finish(tag, hit, occluded)        the ray is done; hit.u / hit.v are valid only for kKeepUV sinks
template <int MODE, bool UV, typename Sink>
__device__ __forceinline__ void trace_wp(const DScene& sc, uint32_t n, uint32_t* head, Sink& sink, WavePool<UV>& S, uint32_t* spill,
                                         uint32_t spill_stride, uint32_t* overflow, unsigned long long* stats = nullptr) {
  // spill: this wave's part of the spill area; entry i >= kWpStack of ray r: spill[(i - kWpStack) * spill_stride + r]
  const uint32_t lane = __lane_id();
  if (sc.num_nodes == 0) {
    for (;;) {  // empty scene: every ray misses
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(head, 64u);
      base = (uint32_t)__shfl((int)base, 0);
      if (base >= n) break;
      const uint32_t idx = base + lane;
      if (idx < n) {
        uint32_t tag;
        V3 o, d;
        float tmin, tmax;
        sink.load(idx, tag, o, d, tmin, tmax);
        Hit h = {tmax, 0.f, 0.f, kNone};
        sink.finish(tag, h, false);
      }
    }
    return;
  }
  const float4* const items = sc.wide;
  const uint32_t waves_total = gridDim.x * (blockDim.x >> 6);
  uint32_t batch = n / waves_total;
  if (batch >= 64u) {
    batch = (kPvGuide ? n / (waves_total * kPvGuide) : batch) & ~63u;
    batch = batch > kPvBatch ? kPvBatch : (batch < 64u ? 64u : batch);
  } else {
    batch = (n + waves_total - 1u) / waves_total;  // few rays: spread them over all waves
    batch = batch < 1u ? 1u : batch;
  }
  uint32_t batch_cur = 0u, batch_end = 0u;  // this wave's claim on the launch's ray queue
  bool exhausted = (n == 0u);
  // ring heads / tails (wave-uniform: scalar registers)
  uint32_t hN = 0u, tN = 0u, hT = 0u, tT = 0u, hD = 0u, tD = 0u, hF = 0u, tF = kWpRays;
  for (uint32_t i = lane; i < kWpRays; i += 64u) S.ring[kWFree][i] = (uint8_t)i;
  wp_sync();
  uint32_t n_step[3] = {0u, 0u, 0u}, n_lane[3] = {0u, 0u, 0u};

  // the common tail of a node step and a triangle step: pop when the ray has no next item, decode the next item
  auto next_item = [&](uint32_t r, uint32_t next, uint32_t& cur, uint32_t& meta) -> uint32_t {
    if (next == kEmptyChild) {
      uint32_t sp = meta & kWmSp;
      if (sp == 0u) return kWDone;
      sp--;
      next = sp < (uint32_t)kWpStack ? S.stack[sp * kWpRays + r] : spill[(sp - (uint32_t)kWpStack) * spill_stride + r];
      meta = (meta & ~kWmSp) | sp;
    }
    if (next & kLeafBit) {
      cur = sc.q_tri0 + kTriPairWords * ((next & 0x3FFFFFFFu) >> 3);
      return kWTri;
    }
    cur = 4u * next;
    return kWNode;
  };

  for (;;) {
    const uint32_t nN = tN - hN, nT = tT - hT, nD = tD - hD, nF = tF - hF;
    const bool have_rays = batch_cur < batch_end || !exhausted;
    const uint32_t nslots = nD + (have_rays ? nF : 0u);
    int step;  // 0 = node, 1 = triangle, 2 = deliveries / new rays
    if (nslots >= kWpRefillAt) step = 2;
    else if (nT >= 64u) step = 1;  // (primitives first: a hit shortens the ray's interval for its later box tests)
    else if (nN >= 64u) step = 0;
    else if (nN == 0u && nT == 0u) {
      if (nslots == 0u) break;  // nothing in flight, nothing to deliver, nothing to fetch
      step = 2;
    } else {
      step = nT > nN ? 1 : 0;
    }
    uint32_t r = 0u, dest = 4u;  // this lane's ray (slot number) and the ring it goes to next (4 = none)
    if (step == 0) {
      const uint32_t k = nN < 64u ? nN : 64u;
      if (lane < k) {
        r = S.ring[kWNode][(hN + lane) & 255u];
        const float4 o4 = S.a0[r], i4 = S.a1[r];
        const uint2 cm = *reinterpret_cast<const uint2*>(&S.m[r]);
        uint32_t cur = cm.x, meta = cm.y;
        const float4* g = items + cur;
        const float4 D0 = g[0], D1 = g[1], D2 = g[2], D3w = g[3];
        uint32_t key[4];
        wide_node_keys(D0, D1, D2, D3w, V3(o4.x, o4.y, o4.z), make_float4(i4.x, i4.y, i4.z, 0.f), o4.w, i4.w, key);
        uint32_t next = kEmptyChild;
        if (key[0] != kWideMiss) next = wide_ref(D3w, key[0]);
        uint32_t sp = meta & kWmSp;
#pragma unroll
        for (int j = 3; j >= 1; j--) {  // the other hit children go on the ray's stack, farthest first
          if (key[j] == kWideMiss) continue;
          const uint32_t ref = wide_ref(D3w, key[j]);
          if (sp < (uint32_t)kWpStack) S.stack[sp * kWpRays + r] = ref, sp++;
          else if (sp < (uint32_t)kStackDepth) spill[(sp - (uint32_t)kWpStack) * spill_stride + r] = ref, sp++;
          else *overflow = 1u;
        }
        meta = (meta & ~kWmSp) | sp;
        dest = next_item(r, next, cur, meta);
        *reinterpret_cast<uint2*>(&S.m[r]) = make_uint2(cur, meta);
      }
      hN += k;
      n_step[0]++, n_lane[0] += k;
    } else if (step == 1) {
      const uint32_t k = nT < 64u ? nT : 64u;
      if (lane < k) {
        r = S.ring[kWTri][(hT + lane) & 255u];
        const float4 o4 = S.a0[r], i4 = S.a1[r];
        const uint4 mm = S.m[r];
        uint32_t cur = mm.x, meta = mm.y;
        const V3 d = sink.dir(mm.z);
        const float4* g = items + cur;  // a triangle leaf of the Q tree: one TriPair, both triangles in one packed test
        const float4 w0 = g[0], w1 = g[1], w2 = g[2], w3 = g[3], w4 = g[4];
        const bool any_ray = (MODE == 1) || (MODE == 2 && (meta & kWmAny) != 0u);
        Hit h = {i4.w, 0.f, 0.f, mm.w};
        uint32_t nt = 0u;
        const bool occ = tri_pair_accept<MODE == 1, false>(sc, w0, w1, w2, w3, w4, V3(o4.x, o4.y, o4.z), d, V3(i4.x, i4.y, i4.z), o4.w, any_ray, h, nt);
        if (occ) {
          meta |= kWmOccluded;
          dest = kWDone;
        } else {
          if (h.slot != mm.w || h.t != i4.w) {  // a new nearest hit
            S.a1[r].w = h.t;
            S.m[r].w = h.slot;
            if (UV) S.uv[UV ? r : 0u] = make_float2(h.u, h.v);
            else sink.accept(mm.z, h.t, h.u, h.v, h.slot);
          }
          dest = next_item(r, kEmptyChild, cur, meta);  // leaf done: pop
        }
        *reinterpret_cast<uint2*>(&S.m[r]) = make_uint2(cur, meta);
      }
      hT += k;
      n_step[1]++, n_lane[1] += k;
    } else {
      // ---- deliveries, then new rays for the slots of this batch while this wave has some
      const uint32_t kd = nD < 64u ? nD : 64u;
      const uint32_t kf = have_rays ? ((64u - kd) < nF ? (64u - kd) : nF) : 0u;
      const uint32_t k = kd + kf;
      if (lane < kd) r = S.ring[kWDone][(hD + lane) & 255u];
      else if (lane < k) r = S.ring[kWFree][(hF + lane - kd) & 255u];
      hD += kd, hF += kf;
      n_step[2]++, n_lane[2] += k;
      if (lane < kd) {
        const uint4 mm = S.m[r];
        Hit h = {S.a1[r].w, 0.f, 0.f, mm.w};
        if (UV) {
          const float2 w = S.uv[UV ? r : 0u];
          h.u = w.x, h.v = w.y;
        }
        sink.finish(mm.z, h, (mm.y & kWmOccluded) != 0u);
      }
      if (batch_cur == batch_end && !exhausted) {
        uint32_t base = 0u;
        if (lane == 0u) base = atomicAdd(head, batch);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        batch_cur = base < n ? base : n;
        batch_end = (base + batch) < n ? (base + batch) : n;
        if (batch_cur >= n) exhausted = true;
        if (kPvGuide && batch >= 64u) {  // guided self-scheduling (trace_pv)
          uint32_t nb = ((n - batch_end) / (waves_total * kPvGuide)) & ~63u;
          batch = nb > kPvBatch ? kPvBatch : (nb < 64u ? 64u : nb);
        }
      }
      const uint32_t avail = batch_end - batch_cur;
      const uint32_t take = k < avail ? k : avail;
      if (lane < take) {
        uint32_t tag;
        V3 o, d;
        float tmin, tmax;
        const bool a = sink.load(batch_cur + lane, tag, o, d, tmin, tmax);
        const bool any_ray = (MODE == 1) || (MODE == 2 && a);
        S.a0[r] = make_float4(o.x, o.y, o.z, tmin);
        S.a1[r] = make_float4(1.0f / d.x, 1.0f / d.y, 1.0f / d.z, tmax);
        S.m[r] = make_uint4(0u /* the root is always an inner node */, any_ray ? kWmAny : 0u, tag, kNone);
        if (UV) S.uv[UV ? r : 0u] = make_float2(0.f, 0.f);
        dest = kWNode;
      } else if (lane < k) {
        dest = kWFree;
      }
      batch_cur += take;
    }
    // ---- tickets to the rings of the next phase
    wp_sync();
    {
      const unsigned long long mN = __ballot(dest == kWNode), mT = __ballot(dest == kWTri), mD = __ballot(dest == kWDone), mF = __ballot(dest == kWFree);
      if (dest == kWNode) S.ring[kWNode][(tN + wp_rank(mN)) & 255u] = (uint8_t)r;
      if (dest == kWTri) S.ring[kWTri][(tT + wp_rank(mT)) & 255u] = (uint8_t)r;
      if (dest == kWDone) S.ring[kWDone][(tD + wp_rank(mD)) & 255u] = (uint8_t)r;
      if (dest == kWFree) S.ring[kWFree][(tF + wp_rank(mF)) & 255u] = (uint8_t)r;
      tN += (uint32_t)__popcll(mN), tT += (uint32_t)__popcll(mT), tD += (uint32_t)__popcll(mD), tF += (uint32_t)__popcll(mF);
    }
    wp_sync();
  }
  if (stats && lane == 0u) {
    atomicAdd(&stats[kStatPvItNode], (unsigned long long)n_step[0]), atomicAdd(&stats[kStatPvLnNode], (unsigned long long)n_lane[0]);
    atomicAdd(&stats[kStatPvItTri], (unsigned long long)n_step[1]), atomicAdd(&stats[kStatPvLnTri], (unsigned long long)n_lane[1]);
    atomicAdd(&stats[kStatPvItCurve], (unsigned long long)n_step[2]), atomicAdd(&stats[kStatPvLnCurve], (unsigned long long)n_lane[2]);
  }
}

}  // namespace pb
