// dtrace_pv2.h -- the persistent phase-voting traversal of dtrace_pv.h with TWO rays per lane (device, Q tree only).
//
// Why: trace_pv is bound by the number of wave-instructions it issues (VALU issue 0.93 of a launch on C2) at 33.5 of 64 lanes
// per instruction: a lane whose ray waits at a primitive idles through the node turns, a lane whose ray is finished idles until
// the next refill.  Here a lane owns two rays, A and B, each with the complete per-ray state of trace_pv (18 registers) and its
// own traversal stack.  The wave votes as before -- but a lane takes part in a turn when EITHER of its rays is in the voted
// state: if it is B, the lane first exchanges A and B (v_swap_b32, one instruction per register, no temporaries), so that the
// code of a turn always works on A.  The item a turn works on (node / triangle / curve piece) is loaded after the vote, for the
// ray that takes part: one set of item registers per lane, not two.  A turn costs ~20 instructions more and serves about 1.4x
// the lanes; the kernel runs at fewer waves per SIMD (each wave carries twice the rays), so the rays in flight stay the same.
//
// Hits do not depend on the order in which boxes and primitives are visited (intersection contract, dtrace.h), so every
// result is bit-identical to trace_pv's and to the one-ray-per-lane traversal's.
#pragma once

#include "dtrace_pv.h"

namespace pb {

#ifndef PB_REFILL2
#define PB_REFILL2 32
#endif
#ifndef PB_REFILL2_CURVES
#define PB_REFILL2_CURVES 24
#endif

// pk: state (bits 0-2) | primitives of the current leaf still to test after the current one (3-5) | any-hit ray (6) |
// which of the lane's two stacks / ribbon frames belongs to this ray (7; it travels with the ray when A and B are exchanged)
constexpr uint32_t kPkState = 7u, kPkRemShift = 3u, kPkRem = 7u << 3, kPkAny = 1u << 6, kPkSid = 1u << 7;

struct PvRay {
  uint32_t tag, cur, pk, sp;
  float ox, oy, oz, dx, dy, dz, ix, iy, iz, tmin;
  float t, u, v;  // current hit (t = the ray's tmax)
  uint32_t slot;
  uint32_t steps;  // STATS only
};

template <bool STATS>
__device__ __forceinline__ void pv_swap(PvRay& a, PvRay& b) {
#define PB_SW(f) asm volatile("v_swap_b32 %0, %1" : "+v"(a.f), "+v"(b.f))
  PB_SW(tag); PB_SW(cur); PB_SW(pk); PB_SW(sp);
  PB_SW(ox); PB_SW(oy); PB_SW(oz); PB_SW(dx); PB_SW(dy); PB_SW(dz); PB_SW(ix); PB_SW(iy); PB_SW(iz); PB_SW(tmin);
  PB_SW(t); PB_SW(u); PB_SW(v); PB_SW(slot);
  if (STATS) PB_SW(steps);
#undef PB_SW
}

template <int MODE, bool STATS, bool CURVES, typename Sink>
__device__ __forceinline__ void trace_pv2(const DScene& sc, uint32_t n, uint32_t* head, Sink& sink, uint32_t* stk_base,
                                          uint32_t stride, uint32_t* spill, uint32_t spill_stride, TravStats& st,
                                          uint32_t* overflow, float* frame = nullptr, const float4* top = nullptr,
                                          uint32_t ntop = 0) {
  // stk_base: 2 x kPv2LdsStack entries per lane in LDS (entry i of stack s: stk_base[(s * kPv2LdsStack + i) * stride]);
  // spill: 2 x (kStackDepth - kPv2LdsStack) entries per thread; frame (CURVES): 2 x 10 words per lane in LDS
  static_assert(!Sink::kWalk, "walking sinks use trace_pv");
  constexpr int kLds = kPv2LdsStack;
  constexpr uint32_t kSpillPerRay = (uint32_t)(kStackDepth - kLds);
  const uint32_t lane = __lane_id();
  auto rank_in = [](unsigned long long m) {
    return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  };
  uint32_t batch_cur = 0, batch_end = 0;
  const uint32_t waves_total = gridDim.x * (blockDim.x >> 6);
  uint32_t batch = n / (2u * waves_total);  // (a wave holds 128 rays)
  if (batch >= 64u) {
    batch = (kPvGuide ? n / (waves_total * kPvGuide) : batch) & ~63u;
    batch = batch > kPvBatch ? kPvBatch : (batch < 64u ? 64u : batch);
  } else {
    batch = (n + waves_total - 1u) / waves_total;  // fewer rays than resident ray slots: spread them thin (trace_pv)
    batch = batch < 1u ? 1u : batch;
  }
  bool exhausted = (n == 0);
  if (sc.num_nodes == 0) {
    if (n == 0) return;
    for (;;) {  // empty scene: every ray misses
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(head, 64u);
      base = (uint32_t)__shfl((int)base, 0);
      if (base >= n) break;
      uint32_t idx = base + lane;
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

  PvRay A = {0u, 0u, kStIdle, 0u, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, kNone, 0u};
  PvRay B = A;
  B.pk = kStIdle | kPkSid;
  const float4* const items = sc.wide;

  for (;;) {
    const uint32_t sa = A.pk & kPkState, sb = B.pk & kPkState;
    const bool idle_a = sa == kStIdle || sa >= kStDone, idle_b = sb == kStIdle || sb >= kStDone;
    const unsigned long long idle_mask = __ballot(idle_a || idle_b);
    const int n_idle = __popcll(idle_mask);
    constexpr int kRefillAt = CURVES ? PB_REFILL2_CURVES : PB_REFILL2;
    bool advance = false, have_next = false;
    uint32_t next = 0;
    if (n_idle >= kRefillAt && !exhausted) {
      // ---- refill: one ray slot per lane with a free one (a lane with two free slots is served again by the next refill)
      const unsigned long long t_refill = STATS ? wall_clock64() : 0ull;
      if (!idle_a && idle_b) pv_swap<STATS>(A, B);
      const bool slot_free = idle_a || idle_b;  // A is the free slot now
      if (batch_cur == batch_end) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(head, batch);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)base, 0));
        batch_cur = base < n ? base : n;
        batch_end = (base + batch) < n ? (base + batch) : n;
        if (batch_cur >= n) exhausted = true;
        if (kPvGuide && batch >= 64u) {  // guided self-scheduling (trace_pv)
          uint32_t nb = ((n - batch_end) / (waves_total * kPvGuide)) & ~63u;
          batch = nb > kPvBatch ? kPvBatch : (nb < 64u ? 64u : nb);
        }
      }
      bool fresh = false;
      const uint32_t avail = batch_end - batch_cur;
      const uint32_t take = (uint32_t)n_idle < avail ? (uint32_t)n_idle : avail;
      const uint32_t rank = rank_in(idle_mask);
      const bool finishing = slot_free && (A.pk & kPkState) >= kStDone;
      const bool occluded = (A.pk & kPkState) == kStDoneOccluded;
      const bool taking = slot_free && rank < take;
      if (CURVES && finishing) A.slot = q_final_code(sc, A.slot);  // a curve hit of the Q tree gets its hit code here
      V3 o(0.f), d(0.f);
      float tmax = 0.f;
      bool a = false;
      if constexpr (sink_splits<Sink>()) {
        typename Sink::Pending pend = {};
        if (finishing) pend = sink.done_issue(A.tag, occluded);
        uint32_t entry = 0u;
        if (taking) entry = sink.load_entry(batch_cur + rank);
        if (finishing) {
          Hit h = {A.t, A.u, A.v, A.slot};
          sink.done_finish(A.tag, pend, h, occluded);
        }
        if (taking) a = sink.load_ray(batch_cur + rank, entry, A.tag, o, d, A.tmin, tmax), fresh = true;
      } else {
        if (finishing) {
          Hit h = {A.t, A.u, A.v, A.slot};
          sink.done(A.tag, h, occluded);
        }
        if (taking) a = sink.load(batch_cur + rank, A.tag, o, d, A.tmin, tmax), fresh = true;
      }
      if (finishing) A.pk = (A.pk & kPkSid) | kStIdle;
      if (fresh) {
        A.ox = o.x, A.oy = o.y, A.oz = o.z, A.dx = d.x, A.dy = d.y, A.dz = d.z;
        A.ix = 1.0f / d.x, A.iy = 1.0f / d.y, A.iz = 1.0f / d.z;
        if (CURVES) {
          const RayFrame f = ray_frame(d);
          const float w[10] = {f.dn.x, f.dn.y, f.dn.z, f.bx.x, f.bx.y, f.bx.z, f.by.x, f.by.y, f.by.z, f.inv_len};
          float* fr = frame + ((A.pk & kPkSid) ? 10u * stride : 0u);
#pragma unroll
          for (int k = 0; k < 10; k++) fr[(uint32_t)k * stride] = w[k];
        }
        A.t = tmax, A.u = 0.f, A.v = 0.f, A.slot = kNone, A.sp = 0u;
        if (STATS) A.steps = 0u;
        const bool any_ray = (MODE == 1) || (MODE == 2 && a);
        A.pk = (A.pk & kPkSid) | kStNode | (any_ray ? kPkAny : 0u);
        A.cur = 0u;  // the root is always an inner node
      }
      batch_cur += take;
      if (STATS) {
        __builtin_amdgcn_s_waitcnt(0x0070);
        if (lane == 0) st.it_refill++, st.refill_ticks += (uint32_t)(wall_clock64() - t_refill);
      }
      continue;
    }
    const unsigned long long node_mask = __ballot(sa == kStNode || sb == kStNode);
    const unsigned long long tri_mask = __ballot(sa == kStTri || sb == kStTri);
    const unsigned long long curve_mask = CURVES ? __ballot(sa == kStCurve || sb == kStCurve) : 0ull;
    if ((node_mask | tri_mask | curve_mask) == 0ull) break;  // queue exhausted, nothing in flight
    const int n_node = __popcll(node_mask), n_tri = __popcll(tri_mask), n_curve = CURVES ? __popcll(curve_mask) : 0;
    const int w_node = n_node * PB_W_NODE, w_tri = n_tri * PB_W_TRI, w_curve = n_curve * PB_W_CURVE;
    const int phase = (w_node >= w_tri && w_node >= w_curve) ? 0 : ((!CURVES || w_tri >= w_curve) ? 1 : 2);
    if (STATS && lane == 0) {
      if (phase == 0) st.it_node++, st.ln_node += n_node;
      else if (phase == 1) st.it_tri++, st.ln_tri += n_tri;
      else st.it_curve++, st.ln_curve += n_curve;
    }
    const uint32_t want = phase == 0 ? kStNode : (phase == 1 ? kStTri : kStCurve);
    const bool use_b = sa != want && sb == want;
    if (use_b) pv_swap<STATS>(A, B);
    const bool mine = sa == want || sb == want;
    const bool any_ray = (MODE == 1) || (MODE == 2 && (A.pk & kPkAny) != 0u);
    const uint32_t sid = (A.pk & kPkSid) ? 1u : 0u;
    if (phase == 0) {
      // ---- NODE turn
      if (mine) {
        float4 D0, D1, D2, D3w;
        if (!CURVES && kTopNodesWide > 0 && A.cur < 4u * ntop) {
          const float4* g = top + A.cur;
          D0 = g[0], D1 = g[1], D2 = g[2], D3w = g[3];
        } else {
          const float4* g = items + A.cur;
          D0 = g[0], D1 = g[1], D2 = g[2], D3w = g[3];
        }
        if (STATS) (any_ray ? st.anodes : st.nodes)++, A.steps++;
        uint32_t k[4];
        const float4 inv4 = make_float4(A.ix, A.iy, A.iz, 0.f);
        wide_node_keys(D0, D1, D2, D3w, V3(A.ox, A.oy, A.oz), inv4, A.tmin, A.t, k);
        auto ref_of = [&](uint32_t key) { return wide_ref(D3w, key); };
        advance = true;
        have_next = k[0] != kWideMiss;
        next = ref_of(k[0]);
        if (k[1] != kWideMiss) {  // the other hit children go on the ray's stack, farthest first
          uint32_t* const sb_lds = stk_base + sid * ((uint32_t)kLds * stride);
          uint32_t* const sb_spill = spill + sid * (kSpillPerRay * spill_stride);
          const uint32_t r3 = ref_of(k[3]), r2 = ref_of(k[2]), r1 = ref_of(k[1]);
          const uint32_t p3 = A.sp, p2 = p3 + ((k[3] != kWideMiss) ? 1u : 0u), p1 = p2 + ((k[2] != kWideMiss) ? 1u : 0u);
          if (__ballot(p1 >= (uint32_t)kLds) == 0ull) {
            // (an entry written for a child that was not hit lies above the new top or is overwritten by the next one)
            sb_lds[p3 * stride] = r3;
            sb_lds[p2 * stride] = r2;
            sb_lds[p1 * stride] = r1;
            A.sp = p1 + 1u;
          } else {
            auto put = [&](uint32_t pos, uint32_t ref) {
              if (pos < (uint32_t)kLds) sb_lds[pos * stride] = ref;
              else if (pos < (uint32_t)kStackDepth) sb_spill[(pos - (uint32_t)kLds) * spill_stride] = ref;
              else *overflow = 1u;
            };
            if (k[3] != kWideMiss) put(p3, r3);
            if (k[2] != kWideMiss) put(p2, r2);
            put(p1, r1);
            A.sp = p1 + 1u < (uint32_t)kStackDepth ? p1 + 1u : (uint32_t)kStackDepth;
          }
        }
      }
    } else {
      // ---- TRI / CURVE turn: one primitive per lane
      if (mine) {
        const bool is_curve = CURVES && phase == 2;
        const float4* g = items + A.cur;
        const float4 D0 = g[0], D1 = g[1];
        float4 D2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!is_curve) D2 = g[2];
        if (STATS) A.steps++;
        float t = 0.f, u = 0.f, v = 0.f;
        bool ok = false;
        const V3 o(A.ox, A.oy, A.oz), inv(A.ix, A.iy, A.iz);
        if (!CURVES) {
          // a triangle leaf of a triangle-only scene's Q tree: one TriPair, both triangles in one packed test (dtrace.h)
          const float4 w3 = g[3], w4 = g[4];
          Hit h = {A.t, A.u, A.v, A.slot};
          uint32_t nt = 0u;
          const bool occ = tri_pair_accept<MODE == 1, STATS>(sc, D0, D1, D2, w3, w4, o, V3(A.dx, A.dy, A.dz), inv, A.tmin, any_ray, h, nt);
          if (STATS) (any_ray ? st.atris : st.tris) += nt;
          A.t = h.t, A.u = h.u, A.v = h.v, A.slot = h.slot;
          if (occ) A.pk = (A.pk & ~kPkState) | kStDoneOccluded;
          else advance = true;
        } else if (!is_curve) {
          if (STATS) (any_ray ? st.atris : st.tris)++;
          ok = tri_test(ld3(D0), ld3(D1), ld3(D2), o, V3(A.dx, A.dy, A.dz), inv, A.tmin, t, u, v) && (t <= A.t);
        } else {
          if (STATS) (any_ray ? st.acurves : st.curves)++;
          const float* fr = frame + (sid ? 10u * stride : 0u);
          RayFrame f;
          f.dn = V3(fr[0], fr[stride], fr[2 * stride]), f.bx = V3(fr[3 * stride], fr[4 * stride], fr[5 * stride]);
          f.by = V3(fr[6 * stride], fr[7 * stride], fr[8 * stride]), f.inv_len = fr[9 * stride];
          ok = segment_test(D0, D1, (A.cur - sc.q_pt0) & 3u, o, f, inv, A.tmin, A.t, t, u, v);
        }
        // the hit code: a triangle slot carries its code, a curve hit is held as its point (dscene.h)
        const uint32_t code = is_curve ? (kQPointHit | (A.cur - sc.q_pt0)) : __float_as_uint(D2.w);
        if (ok && !any_ray && t == A.t && A.slot != kNone) ok = q_gid(sc, code) < q_gid(sc, A.slot);
        if (ok) A.t = t, A.u = u, A.v = v, A.slot = code;
        if (!CURVES) {
          // (handled above)
        } else if (any_ray && ok) {
          A.pk = (A.pk & ~kPkState) | kStDoneOccluded;
          if (STATS) st.ahist[A.steps <= 16u ? 0 : (28 - __clz(A.steps - 1u) > 7 ? 7 : 28 - __clz(A.steps - 1u))]++, st.amax_steps = A.steps > st.amax_steps ? A.steps : st.amax_steps;
        } else if (A.pk & kPkRem) {  // next primitive of the same leaf
          A.pk -= 1u << kPkRemShift;
          A.cur += is_curve ? 1u : 3u;
        } else {
          advance = true;  // leaf done: pop
        }
      }
    }
    // ---- common tail: pop / finish / decode the ray's next item
    if (advance) {
      if (!have_next) {
        if (A.sp == 0u) {
          A.pk = (A.pk & ~kPkState) | kStDone;
          advance = false;
          if (STATS) {
            const int b = A.steps <= 16u ? 0 : (28 - __clz(A.steps - 1u) > 7 ? 7 : 28 - __clz(A.steps - 1u));
            if (any_ray) st.ahist[b]++, st.amax_steps = A.steps > st.amax_steps ? A.steps : st.amax_steps;
            else st.hist[b]++, st.max_steps = A.steps > st.max_steps ? A.steps : st.max_steps;
          }
        } else {
          A.sp--;
          const uint32_t* const sb_lds = stk_base + sid * ((uint32_t)kLds * stride);
          next = sb_lds[(A.sp < (uint32_t)kLds ? A.sp : (uint32_t)(kLds - 1)) * stride];
          if (A.sp >= (uint32_t)kLds) next = spill[(sid * kSpillPerRay + (A.sp - (uint32_t)kLds)) * spill_stride];
        }
      }
      if (advance) {
        if (next & kLeafBit) {
          const uint32_t first = (next & 0x3FFFFFFFu) >> 3;
          const bool curve = CURVES && (next & kCurveBit);
          A.pk = (A.pk & ~(kPkState | kPkRem)) | (curve ? kStCurve : kStTri) | (CURVES ? ((next & 7u) << kPkRemShift) : 0u);
          A.cur = curve ? sc.q_pt0 + first : sc.q_tri0 + (CURVES ? 3u : kTriPairWords) * first;
        } else {
          A.cur = 4u * next;
          A.pk = (A.pk & ~(kPkState | kPkRem)) | kStNode;
        }
      }
    }
  }
  // rays that finished after the queue ran dry
#pragma unroll
  for (int r = 0; r < 2; r++) {
    if ((A.pk & kPkState) >= kStDone) {
      if (CURVES) A.slot = q_final_code(sc, A.slot);
      Hit h = {A.t, A.u, A.v, A.slot};
      sink.done(A.tag, h, (A.pk & kPkState) == kStDoneOccluded);
    }
    if (r == 0) pv_swap<STATS>(A, B);
  }
}

}  // namespace pb
