// dtrace_q.h -- pooled traversal: the rays of a block live in LDS, its waves are stateless workers that take FULL batches of
// same-phase work from shared queues (device, Q tree, triangle-only scenes).
//
// Why: the phase-voting traversal (dtrace_pv.h) keeps a ray in the registers of ONE lane from fetch to delivery, so a turn of
// the wave serves only the lanes whose ray happens to be in the voted phase: 36.7 of 64 lanes per node turn, 22.4 per
// triangle turn on C2, and the kernel is bound by the wave-instructions it issues (VALU issue 0.93).  Two rays per lane
// (dtrace_pv2.h) raises that to 43.8 / 26.6 and loses it again to the exchange instructions and the lower occupancy.
// Here the binding of rays to lanes is given up: a block owns kQRays ray slots in LDS (origin, 1 / d, direction, current hit,
// traversal stack: 116 bytes per ray) and four queues of slot numbers --
//     node: rays whose next item is an inner node          tri:  rays whose next item is a triangle of a leaf
//     done: finished rays whose result is to be delivered  free: empty slots
// -- and a wave repeatedly claims up to 64 entries of ONE queue, reads those rays' state from LDS, does that one step for all
// of them (every lane busy, no vote, no idle phase), writes the state back and appends each ray to the queue of its next
// phase.  A queue entry is a ticket: a ray is in at most one queue and in the hands of at most one lane, so its LDS state
// needs no lock.  Queues are rings (capacity >= the number of tickets: they cannot overflow).  A producer reserves entries with
// one LDS atomic per wave (tail), stores them and publishes them on a semaphore (avail); a consumer takes k entries from the
// semaphore -- giving them back if there were fewer: no compare-and-swap loop, sixteen waves contend for these words -- and
// then k tickets from the head counter; "empty" sentinels in the entries cover the window in which a producer that reserved
// earlier publishes later.
// Results are bit-identical to the other traversals': hits do not depend on the visiting order (dtrace.h).
#pragma once

#include "dtrace_pv.h"

namespace pb {

constexpr int kQBlock = 1024;      // threads per block: 16 waves, one block per CU
constexpr uint32_t kQRays = 1280;  // ray slots per block
constexpr int kQStack = 8;         // stack entries per ray in LDS (deeper ones: the global spill area)
enum : uint32_t { kQNode = 0, kQTri = 1, kQDone = 2, kQFree = 3, kQNum = 4 };
constexpr uint32_t kQCap = 2048;  // ring capacity (a power of two >= kQRays: every ray is in at most one queue, so a ring cannot overflow)
constexpr uint16_t kQEmpty = 0xFFFFu;
// meta word of a ray: stack pointer | primitives of the current leaf still to test after the current one | flags
constexpr uint32_t kQmSp = 0xFFu, kQmRemShift = 8u, kQmRem = 7u << 8, kQmAny = 1u << 11, kQmOccluded = 1u << 12;
// counters: head[q] = tickets handed to consumers, tail[q] = entries reserved by producers, avail[q] = entries published and not
// yet claimed (a semaphore: a consumer takes k with one atomic subtraction and gives them back if there were fewer)
enum : uint32_t { kQcHead = 0, kQcTail = 4, kQcAvail = 8, kQcLive = 12, kQcPending = 13, kQcNum = 16 };

struct QPool {
  float4 a0[kQRays];                 // origin, tmin
  float4 a1[kQRays];                 // 1 / d, current hit distance (the ray's tmax)
  float4 a2[kQRays];                 // d, current item (16-byte index into DScene::wide)
  float4 a3[kQRays];                 // hit u, v, the ray's tag, hit code
  uint32_t meta[kQRays];
  uint32_t stack[kQStack * kQRays];  // entry i of ray r: stack[i * kQRays + r]
  uint16_t q[kQNum][kQCap];
  uint32_t ctr[kQcNum];              // + rays in flight, rays claimed from the launch's queue but not installed
};
static_assert(sizeof(QPool) <= 160 * 1024, "the pool is one CU's LDS");

__device__ __forceinline__ uint32_t q_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint16_t q_ld16(const uint16_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void q_st16(uint16_t* p, uint16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t q_rank(unsigned long long m) {
  return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// every lane with `pred` appends its slot number r to queue `which`: reserve (one LDS atomic per wave), store, publish
__device__ __forceinline__ void q_push(QPool& S, uint32_t which, bool pred, uint32_t r, uint32_t lane) {
  const unsigned long long m = __ballot(pred);
  if (m == 0ull) return;
  const uint32_t n = (uint32_t)__popcll(m), first = (uint32_t)__builtin_ctzll(m);
  uint32_t base = 0u;
  if (lane == first) base = atomicAdd(&S.ctr[kQcTail + which], n);
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)first);
  asm volatile("" ::: "memory");  // (the ray's state is written before its ticket: DS operations of a wave execute in order)
  if (pred) q_st16(&S.q[which][(base + q_rank(m)) & (kQCap - 1u)], (uint16_t)r);
  asm volatile("" ::: "memory");
  if (lane == first) atomicAdd(&S.ctr[kQcAvail + which], n);
}

template <int MODE, typename Sink>
__device__ __forceinline__ void trace_pool(const DScene& sc, uint32_t n, uint32_t* head, Sink& sink, QPool& S, uint32_t* spill,
                                           uint32_t spill_stride, uint32_t* overflow, unsigned long long* stats = nullptr) {
  // stats: null, or the render's statistics array: steps and rays per step kind go to kStatPvIt* / kStatPvLn* (curve = deliveries / new rays)
  uint32_t n_step[3] = {0u, 0u, 0u}, n_lane[3] = {0u, 0u, 0u}, n_poll = 0u, n_lost = 0u;
  // spill: this block's part of the spill area; entry i >= kQStack of ray r: spill[(i - kQStack) * spill_stride + r]
  const uint32_t lane = __lane_id();
  for (uint32_t i = threadIdx.x; i < kQCap; i += blockDim.x) {
    S.q[kQNode][i] = kQEmpty, S.q[kQTri][i] = kQEmpty, S.q[kQDone][i] = kQEmpty, S.q[kQFree][i] = i < kQRays ? (uint16_t)i : kQEmpty;
  }
  if (threadIdx.x < kQcNum) S.ctr[threadIdx.x] = (threadIdx.x == kQcTail + kQFree || threadIdx.x == kQcAvail + kQFree) ? kQRays : 0u;
  __syncthreads();
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
        sink.done(tag, h, false);
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
  uint32_t polls = 0u;

  for (;;) {
    // ---- which queue: full batches first (free slots when this wave has rays for them, then deliveries, nodes, triangles)
    uint32_t c = 0u;
    if (lane < 6u) c = q_ld(&S.ctr[kQcAvail + lane]);
    uint32_t av[kQNum];
#pragma unroll
    for (int i = 0; i < (int)kQNum; i++) {
      const int a = __builtin_amdgcn_readlane((int)c, i);  // (transiently negative while another wave gives a claim back)
      av[i] = a > 0 ? (uint32_t)a : 0u;
    }
    const uint32_t live = (uint32_t)__builtin_amdgcn_readlane((int)c, 4), pending = (uint32_t)__builtin_amdgcn_readlane((int)c, 5);
    const bool have_rays = batch_cur < batch_end || !exhausted;
    if (!have_rays) av[kQFree] = 0u;
    uint32_t which = kQNum;
    if (av[kQFree] >= 64u) which = kQFree;
    else if (av[kQDone] >= 64u) which = kQDone;
    else if (av[kQNode] >= 64u) which = kQNode;
    else if (av[kQTri] >= 64u) which = kQTri;
    else {
      uint32_t best = 0u;
#pragma unroll
      for (int i = 0; i < (int)kQNum; i++)
        if (av[i] > best) best = av[i], which = (uint32_t)i;
    }
    if (which == kQNum) {
      if (!have_rays && live == 0u && pending == 0u) break;  // nothing in flight and nobody can bring more
      if (++polls > (1u << 26)) {  // (a bug guard: never spin forever)
        *overflow = 2u;
        break;
      }
      n_poll++;
      __builtin_amdgcn_s_sleep(4);
      continue;
    }
    // (selects, not indexed reads: the counters stay in scalar registers)
    const uint32_t avw = which == kQNode ? av[kQNode] : (which == kQTri ? av[kQTri] : (which == kQDone ? av[kQDone] : av[kQFree]));
    const uint32_t k = avw < 64u ? avw : 64u;
    // claim k published entries: take them from the semaphore (and give them back if another wave was faster), then the tickets
    uint32_t hdw = 0u;
    {
      int was = 0;
      if (lane == 0u) was = (int)atomicSub(&S.ctr[kQcAvail + which], k);
      was = __builtin_amdgcn_readfirstlane(was);
      if (was < (int)k) {
        if (lane == 0u) atomicAdd(&S.ctr[kQcAvail + which], k);
        n_lost++;
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      if (lane == 0u) hdw = atomicAdd(&S.ctr[kQcHead + which], k);
      hdw = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdw);
    }
    polls = 0u;
    {
      const int kind = which == kQNode ? 0 : (which == kQTri ? 1 : 2);
      n_step[kind]++, n_lane[kind] += k;
    }
    uint32_t r = kQEmpty;
    if (lane < k) {
      uint16_t* e = &S.q[which][(hdw + lane) & (kQCap - 1u)];
      for (uint32_t s = 0; s < (1u << 22); s++) {  // (a producer that published later may have reserved earlier: its store is on its way)
        r = q_ld16(e);
        if (r != kQEmpty) break;
      }
      q_st16(e, kQEmpty);
      if (r == kQEmpty) *overflow = 2u;
    }
    const bool active = r != kQEmpty;
    uint32_t dest = kQNum;  // the queue this lane's ray goes to next

    if (which == kQNode || which == kQTri) {
      uint32_t m = 0u, cur = 0u, next = kEmptyChild;
      bool advance = false;
      if (which == kQNode) {
        if (active) {
          const float4 o4 = S.a0[r], i4 = S.a1[r];
          cur = __float_as_uint(S.a2[r].w), m = S.meta[r];
          const float4* g = items + cur;
          const float4 D0 = g[0], D1 = g[1], D2 = g[2], D3w = g[3];
          uint32_t key[4];
          wide_node_keys(D0, D1, D2, D3w, V3(o4.x, o4.y, o4.z), make_float4(i4.x, i4.y, i4.z, 0.f), o4.w, i4.w, key);
          advance = true;
          if (key[0] != kWideMiss) next = wide_ref(D3w, key[0]);
          uint32_t sp = m & kQmSp;
#pragma unroll
          for (int j = 3; j >= 1; j--) {  // the other hit children go on the ray's stack, farthest first
            if (key[j] == kWideMiss) continue;
            const uint32_t ref = wide_ref(D3w, key[j]);
            if (sp < (uint32_t)kQStack) S.stack[sp * kQRays + r] = ref, sp++;
            else if (sp < (uint32_t)kStackDepth) spill[(sp - (uint32_t)kQStack) * spill_stride + r] = ref, sp++;
            else *overflow = 1u;
          }
          m = (m & ~kQmSp) | sp;
        }
      } else {
        if (active) {
          const float4 o4 = S.a0[r], i4 = S.a1[r], d4 = S.a2[r];
          cur = __float_as_uint(d4.w), m = S.meta[r];
          const float4* g = items + cur;
          const float4 D0 = g[0], D1 = g[1], D2 = g[2];
          const bool any_ray = (MODE == 1) || (MODE == 2 && (m & kQmAny) != 0u);
          float t, u, v;
          bool ok = tri_test(ld3(D0), ld3(D1), ld3(D2), V3(o4.x, o4.y, o4.z), V3(d4.x, d4.y, d4.z), V3(i4.x, i4.y, i4.z), o4.w, t, u, v) && (t <= i4.w);
          const uint32_t code = __float_as_uint(D2.w);
          if (ok && !any_ray && t == i4.w) {  // tie: the smaller canonical primitive id wins
            const uint32_t held = __float_as_uint(S.a3[r].w);
            if (held != kNone) ok = q_gid(sc, code) < q_gid(sc, held);
          }
          if (ok) {
            S.a1[r].w = t;
            S.a3[r].x = u, S.a3[r].y = v, S.a3[r].w = __uint_as_float(code);
          }
          if (any_ray && ok) {
            m |= kQmOccluded;
            dest = kQDone;
          } else if (m & kQmRem) {  // next primitive of the same leaf
            m -= 1u << kQmRemShift;
            cur += 3u;
            dest = kQTri;
          } else {
            advance = true;  // leaf done: pop
          }
        }
      }
      if (advance) {
        if (next == kEmptyChild) {
          uint32_t sp = m & kQmSp;
          if (sp == 0u) {
            dest = kQDone;
          } else {
            sp--;
            next = sp < (uint32_t)kQStack ? S.stack[sp * kQRays + r] : spill[(sp - (uint32_t)kQStack) * spill_stride + r];
            m = (m & ~kQmSp) | sp;
          }
        }
        if (next != kEmptyChild) {
          if (next & kLeafBit) {
            cur = sc.q_tri0 + 3u * ((next & 0x3FFFFFFFu) >> 3);
            m = (m & ~kQmRem) | ((next & 7u) << kQmRemShift);
            dest = kQTri;
          } else {
            cur = 4u * next;
            dest = kQNode;
          }
        }
      }
      if (active) {
        S.a2[r].w = __uint_as_float(cur);
        S.meta[r] = m;
      }
    } else {
      // ---- deliveries and / or new rays
      bool want_slot = active;
      if (which == kQDone) {
        if (active) {
          const float4 h4 = S.a3[r];
          const Hit h = {S.a1[r].w, h4.x, h4.y, __float_as_uint(h4.w)};
          sink.done(__float_as_uint(h4.z), h, (S.meta[r] & kQmOccluded) != 0u);
        }
        if (lane == 0u) atomicSub(&S.ctr[kQcLive], k);
      }
      // the slots of this batch take new rays while this wave has some (its claim on the launch's queue, renewed when empty)
      const unsigned long long want_mask = __ballot(want_slot);
      const uint32_t need = (uint32_t)__popcll(want_mask);
      if (batch_cur == batch_end && !exhausted) {
        uint32_t base = 0u;
        if (lane == 0u) base = atomicAdd(head, batch);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)base, 0));
        batch_cur = base < n ? base : n;
        batch_end = (base + batch) < n ? (base + batch) : n;
        if (batch_cur >= n) exhausted = true;
        if (kPvGuide && batch >= 64u) {  // guided self-scheduling (trace_pv)
          uint32_t nb = ((n - batch_end) / (waves_total * kPvGuide)) & ~63u;
          batch = nb > kPvBatch ? kPvBatch : (nb < 64u ? 64u : nb);
        }
        if (lane == 0u && batch_end > batch_cur) atomicAdd(&S.ctr[kQcPending], batch_end - batch_cur);
      }
      const uint32_t avail = batch_end - batch_cur;
      const uint32_t take = need < avail ? need : avail;
      const uint32_t rank = q_rank(want_mask);
      if (want_slot && rank < take) {
        uint32_t tag;
        V3 o, d;
        float tmin, tmax;
        const bool a = sink.load(batch_cur + rank, tag, o, d, tmin, tmax);
        const bool any_ray = (MODE == 1) || (MODE == 2 && a);
        S.a0[r] = make_float4(o.x, o.y, o.z, tmin);
        S.a1[r] = make_float4(1.0f / d.x, 1.0f / d.y, 1.0f / d.z, tmax);
        S.a2[r] = make_float4(d.x, d.y, d.z, __uint_as_float(0u));  // the root is always an inner node
        S.a3[r] = make_float4(0.f, 0.f, __uint_as_float(tag), __uint_as_float(kNone));
        S.meta[r] = any_ray ? kQmAny : 0u;
        dest = kQNode;
      } else if (want_slot) {
        dest = kQFree;
      }
      batch_cur += take;
      if (lane == 0u && take) {
        atomicAdd(&S.ctr[kQcLive], take);
        atomicSub(&S.ctr[kQcPending], take);
      }
    }
    q_push(S, kQNode, dest == kQNode, r, lane);
    q_push(S, kQTri, dest == kQTri, r, lane);
    q_push(S, kQDone, dest == kQDone, r, lane);
    q_push(S, kQFree, dest == kQFree, r, lane);
  }
  if (stats && lane == 0u) {
    atomicAdd(&stats[kStatPvItNode], (unsigned long long)n_step[0]), atomicAdd(&stats[kStatPvLnNode], (unsigned long long)n_lane[0]);
    atomicAdd(&stats[kStatPvItTri], (unsigned long long)n_step[1]), atomicAdd(&stats[kStatPvLnTri], (unsigned long long)n_lane[1]);
    atomicAdd(&stats[kStatPvItCurve], (unsigned long long)n_step[2]), atomicAdd(&stats[kStatPvLnCurve], (unsigned long long)n_lane[2]);
    atomicAdd(&stats[kStatPvItRefill], (unsigned long long)n_poll + ((unsigned long long)n_lost << 32));
  }
}

}  // namespace pb
