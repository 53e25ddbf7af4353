// dtrace_pv8.h -- the persistent, phase-voting traversal (dtrace_pv.h) over the O tree: eight children per node, 80-byte items
// (dscene.h::Node8 / TriPair), children visited in octant order, ONE stack entry per node (device; triangle-only scenes).
//
// Why (round 5): the counters of k_trace on the 4-wide Q tree say the VALU pipes are half idle (a wave64 VALU
// instruction issues over 2 cycles, not 4) while the waves wait on the vector-memory path -- a ray is a chain of ~9 DEPENDENT item
// fetches (6.8 nodes + 2.x leaves on C2) of 64-80 bytes each.  The O tree spends some of the idle issue slots on fewer, denser
// steps: one 80-byte node carries eight quantised boxes (node visits per ray x ~0.65, bytes per ray down as well: 4.5 x 80 against
// 6.8 x 64), and the order in which the hit children are visited comes from the ray's direction signs and the SLOT a child was put
// in by the builder (dscene.h), not from sorted entry distances: no sorting network, no reference selects, and what is left of a
// node after its nearest child is ONE 8-byte stack entry (base, masks, remaining hits) instead of up to seven references.
//
// Per-lane state machine as in dtrace_pv.h:  IDLE -> NODE <-> TRI, the next item (an 80-byte node or TriPair: five 16-byte words,
// one load site) prefetched as soon as it is known; the wave votes for the phase with the most lanes; finished rays are delivered
// and idle lanes refilled in bulk.  The leaf test is dtrace.h::tri_pair_accept, the box arithmetic box_test4q's (fma(q, s, org),
// then the binary tree's operations): hits are bit-identical to every other traversal (the intersection contract, dtrace.h).
#pragma once

#include "dtrace_pv.h"

namespace pb {

#ifndef PB_LDS_STACK8
#define PB_LDS_STACK8 13
#endif
constexpr int kPv8LdsStack = PB_LDS_STACK8;  // stack entries (8 bytes each) per lane kept in LDS; deeper ones spill to the group's global area
#ifndef PB_LDS_STACK8_CURVES
#define PB_LDS_STACK8_CURVES 10
#endif
constexpr int kPv8LdsStackCurves = PB_LDS_STACK8_CURVES;  // ... of the kernels of scenes with curves (next to 10 words of ray frame per lane)
#ifndef PB_W_CURVE8
#define PB_W_CURVE8 2
#endif
#ifndef PB_W_NODE8
#define PB_W_NODE8 1  // weights of one lane in the phase vote (node : leaf), as PB_W_NODE / PB_W_TRI of dtrace_pv.h
#endif
#ifndef PB_W_TRI8
#define PB_W_TRI8 2
#endif
constexpr int kStackDepth8 = 32;             // entries per ray: one per level of the O tree (a deeper tree is refused at commit)
static_assert((size_t)(kStackDepth8 - (kPv8LdsStack < kPv8LdsStackCurves ? kPv8LdsStack : kPv8LdsStackCurves)) * 2 <= (size_t)kStackDepth, "the spill area holds kStackDepth words per resident thread");

// The slab test on the eight quantised boxes of a Node8 (its five words w0..w4, dscene.h) against the ray interval [tmin, tmax]:
// bit s of the result = the box in slot s is hit.  Per pair of slots the arithmetic is box_test4q's, operation for operation: a
// bound is rebuilt as fma(q, s, org) -- the builder has checked with this very expression that the result encloses the binary
// tree's widened box -- and then tested with the binary tree's packed subtraction, multiplication, min / max and the 2^-16
// widening of the interval.  Bytes of empty slots are tested like the others; the caller masks the result with the node's
// `present` bits.
__device__ __forceinline__ uint32_t box_test8q(const float4& w0, const float4& w1, const float4& w2, const float4& w3, const float4& w4, float o_x,
                                               float o_y, float o_z, const float4& inv, float tmin, float tmax) {
  const float e = 1.52587890625e-05f;
  // (the entry / exit bounds are selected by the sign of 1 / d as whole words before the conversion: dtrace.h::box_test4q)
  const bool nx = (int)__float_as_uint(inv.x) < 0, ny = (int)__float_as_uint(inv.y) < 0, nz = (int)__float_as_uint(inv.z) < 0;
  const f2 sx = {w0.w, w0.w}, sy = {w1.x, w1.x}, sz = {w1.y, w1.y}, gx = {w0.x, w0.x}, gy = {w0.y, w0.y}, gz = {w0.z, w0.z};
  const f2 ox = {o_x, o_x}, oy = {o_y, o_y}, oz = {o_z, o_z}, ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
  uint32_t hits = 0u;
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {  // word hh of each bound: slots 4 hh .. 4 hh + 3
    const uint32_t qlx = __float_as_uint(hh ? w2.y : w2.x), qly = __float_as_uint(hh ? w2.w : w2.z), qlz = __float_as_uint(hh ? w3.y : w3.x);
    const uint32_t qhx = __float_as_uint(hh ? w3.w : w3.z), qhy = __float_as_uint(hh ? w4.y : w4.x), qhz = __float_as_uint(hh ? w4.w : w4.z);
    const uint32_t lx = nx ? qhx : qlx, ly = ny ? qhy : qly, lz = nz ? qhz : qlz;
    const uint32_t hx = nx ? qlx : qhx, hy = ny ? qly : qhy, hz = nz ? qlz : qhz;
#pragma unroll
    for (int hl = 0; hl < 2; hl++) {  // slots 4 hh + 2 hl, + 1: bytes 2 hl, 2 hl + 1
      const int h = 2 * hh + hl;
      auto two = [hl](uint32_t w) { return hl ? f2{(float)((w >> 16) & 255u), (float)(w >> 24)} : f2{(float)(w & 255u), (float)((w >> 8) & 255u)}; };
      const f2 px = (__builtin_elementwise_fma(two(lx), sx, gx) - ox) * ix, qx = (__builtin_elementwise_fma(two(hx), sx, gx) - ox) * ix;
      const f2 py = (__builtin_elementwise_fma(two(ly), sy, gy) - oy) * iy, qy = (__builtin_elementwise_fma(two(hy), sy, gy) - oy) * iy;
      const f2 pz = (__builtin_elementwise_fma(two(lz), sz, gz) - oz) * iz, qz = (__builtin_elementwise_fma(two(hz), sz, gz) - oz) * iz;
      const float m0 = __builtin_fmaxf(__builtin_fmaxf(px.x, py.x), pz.x), m1 = __builtin_fmaxf(__builtin_fmaxf(px.y, py.y), pz.y);
      const float n0 = __builtin_fminf(__builtin_fminf(qx.x, qy.x), qz.x), n1 = __builtin_fminf(__builtin_fminf(qx.y, qy.y), qz.y);
      const float a0 = __builtin_fmaf(-fabsf(m0), e, m0), a1 = __builtin_fmaf(-fabsf(m1), e, m1);
      const float b0 = __builtin_fmaf(fabsf(n0), e, n0), b1 = __builtin_fmaf(fabsf(n1), e, n1);
      // (the comparisons of box_test2 / wide_node_keys, as they are: a NaN entry or exit distance -- a ray with a NaN component -- hits
      // NO box.  Folding them into max(a, tmin) <= min(b, tmax) lets such a ray hit EVERY box: v_max / v_min drop the NaN, and one NaN
      // ray then walks all 332 574 items of the C2 tree, 340 ms of a 3 ms launch: measured, profiles/README.md)
      hits |= ((a0 <= b0 && b0 >= tmin && a0 <= tmax) ? 1u : 0u) << (2 * h);
      hits |= ((a1 <= b1 && b1 >= tmin && a1 <= tmax) ? 2u : 0u) << (2 * h);
    }
  }
  return hits;
}
// bit s of x moves to bit s ^ m (m = the ray's direction signs, 3 bits): the hit mask in visiting order
__device__ __forceinline__ uint32_t octant_permute(uint32_t x, uint32_t m) {
  x = (m & 1u) ? (((x & 0x55u) << 1) | ((x >> 1) & 0x55u)) : x;
  x = (m & 2u) ? (((x & 0x33u) << 2) | ((x >> 2) & 0x33u)) : x;
  x = (m & 4u) ? (((x & 0x0Fu) << 4) | (x >> 4)) : x;
  return x;
}

// Sink, MODE, STATS: as trace_pv (dtrace_pv.h).  stk: this lane's LDS stack (stk[i * stride], 8-byte entries); spill: its global
// spill entries (spill[(i - kPv8LdsStack) * spill_stride]).  CURVES: the scene has curve leaves (frame: 10 words per lane in LDS,
// frame[k * stride]: the ray's RayFrame, written when the ray is fetched).
template <int MODE, bool STATS, bool CURVES, typename Sink>
__device__ __forceinline__ void trace_pv8(const DScene& sc, uint32_t n, uint32_t* head, Sink& sink, uint2* stk, uint32_t stride, uint2* spill,
                                          uint32_t spill_stride, TravStats& st, uint32_t* overflow, float* frame = nullptr) {
  constexpr int kLds = CURVES ? kPv8LdsStackCurves : kPv8LdsStack;
  const uint32_t lane = __lane_id();
  auto rank_in = [](unsigned long long m) {
    return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  };
  // the ray queue is handed out as in trace_pv: guided batches, thin spreading of small launches
  uint32_t batch_cur = 0, batch_end = 0;
  const uint32_t waves_total = gridDim.x * (blockDim.x >> 6);
  uint32_t batch = n / waves_total;
  if (batch >= 64u) {
    batch = (kPvGuide ? n / (waves_total * kPvGuide) : batch) & ~63u;
    batch = batch > kPvBatch ? kPvBatch : (batch < 64u ? 64u : batch);
  } else {
    batch = (n + waves_total - 1u) / waves_total;
    batch = batch < 1u ? 1u : batch;
  }
  bool exhausted = (n == 0);
  const float4* const items = sc.wide8;

  // per-lane state
  uint32_t state = kStIdle, tag = 0;
  float o_x = 0.f, o_y = 0.f, o_z = 0.f;
  V3 d(0.f);
  float4 inv4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float tmin = 0.f;  // (the current tmax of the ray is hit.t)
  Hit hit = {0.f, 0.f, 0.f, kNone};
  int sp = 0;
  uint32_t steps = 0;
  bool any_ray = (MODE == 1);
  uint32_t oct = 0u;                  // the ray's direction signs: bit a = d[a] < 0; CURVES: bit 3 = the second piece of the curve leaf is next
#ifdef PB_TRAV_TWICE  // diagnostic build: every ray is traversed twice before it is delivered (what does the traversal itself cost?)
  bool second = false;
  float tmax0 = 0.f;
#endif
  uint32_t g_base = 0u, g_bits = 0u;  // what is left of the node the ray is in: first child word | hits in visiting order (8) | imask << 8 | tmask << 16 | cmask << 24
  float4 D0 = make_float4(0, 0, 0, 0), D1 = D0, D2 = D0, D3 = D0, D4 = D0;  // the prefetched item: a Node8, a TriPair or a curve leaf

  unsigned long long t_turn = STATS ? __builtin_readcyclecounter() : 0ull;  // STATS: the turn's cycles go to what it did
  int did = -1;
  for (;;) {
    if (STATS) {
      const unsigned long long t_now = __builtin_readcyclecounter();
      if (lane == 0 && did >= 0) st.cyc[did] += t_now - t_turn;
      t_turn = t_now;
    }
    bool advance = false, need_load = false;
    uint32_t cur = 0u;
    unsigned long long idle_mask = __ballot(state == kStIdle || state >= kStDone);
    int n_idle = __popcll(idle_mask);
    constexpr int kRefillAt = Sink::kWalk ? PB_WALK_REFILL : (CURVES ? kPvRefillIdleCurves : kPvRefillIdle);
    const int n_busy_now = 64 - n_idle;
    if ((n_idle >= kRefillAt || (Sink::kWalk && n_busy_now == 0)) && (!exhausted || (Sink::kWalk && __ballot(state >= kStDone) != 0ull))) {
      // ---- refill idle lanes from the queue (trace_pv's protocol)
      const unsigned long long t_refill = STATS ? wall_clock64() : 0ull;
      did = 3;
      if (!exhausted && batch_cur == batch_end) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(head, batch);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)base, 0));
        batch_cur = base < n ? base : n;
        batch_end = (base + batch) < n ? (base + batch) : n;
        if (batch_cur >= n) exhausted = true;
        if (kPvGuide && batch >= 64u) {
          uint32_t nb = ((n - batch_end) / (waves_total * kPvGuide)) & ~63u;
          batch = nb > kPvBatch ? kPvBatch : (nb < 64u ? 64u : nb);
        }
      }
      bool fresh = false;
      uint32_t taken = 0u;
      V3 o(o_x, o_y, o_z);
      if constexpr (sink_splits<Sink>()) {
        const bool finishing = state >= kStDone;
        const uint32_t avail0 = batch_end - batch_cur;
        const uint32_t take0 = (uint32_t)n_idle < avail0 ? (uint32_t)n_idle : avail0;
        const uint32_t rank0 = rank_in(idle_mask);
        const bool taking = (state == kStIdle || finishing) && rank0 < take0;
        typename Sink::Pending pend = {};
        if (finishing) pend = sink.done_issue(tag, state == kStDoneOccluded);
        uint32_t entry = 0u;
        if (taking) entry = sink.load_entry(batch_cur + rank0);
        if (finishing) {
          sink.done_finish(tag, pend, hit, state == kStDoneOccluded);
          state = kStIdle;
        }
        if (taking) {
          float tmax;
          const bool a = sink.load_ray(batch_cur + rank0, entry, tag, o, d, tmin, tmax);
          any_ray = (MODE == 1) || (MODE == 2 && a);
          hit.t = tmax;
          fresh = true;
        }
        taken = take0;
      } else {
        if (state >= kStDone) {
          if constexpr (Sink::kWalk) {
            float tmax = 0.f;
            fresh = sink.next(tag, hit, o, d, tmin, tmax);
            hit.t = tmax;
            if (!fresh) state = kStIdle;
          } else {
            sink.done(tag, hit, state == kStDoneOccluded);
            state = kStIdle;
          }
        }
        if constexpr (Sink::kWalk) {
          idle_mask = __ballot(state == kStIdle);
          n_idle = __popcll(idle_mask);
        }
        uint32_t avail = batch_end - batch_cur;
        uint32_t take = (uint32_t)n_idle < avail ? (uint32_t)n_idle : avail;
        uint32_t rank = rank_in(idle_mask);
        if (state == kStIdle && rank < take) {
          if constexpr (Sink::kWalk) {
            sink.start(batch_cur + rank, tag, hit, o, d);
            state = kStDone;  // its first step runs at the next refill, together with the other lanes'
          } else {
            float tmax;
            bool a = sink.load(batch_cur + rank, tag, o, d, tmin, tmax);
            any_ray = (MODE == 1) || (MODE == 2 && a);
            hit.t = tmax;
            fresh = true;
          }
        }
        taken = take;
      }
      o_x = o.x, o_y = o.y, o_z = o.z;
      if (fresh) {
        inv4 = make_float4(1.0f / d.x, 1.0f / d.y, 1.0f / d.z, 0.f);
        oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
        if (CURVES) {
          const RayFrame f = ray_frame(d);
          const float w[10] = {f.dn.x, f.dn.y, f.dn.z, f.bx.x, f.bx.y, f.bx.z, f.by.x, f.by.y, f.by.z, f.inv_len};
#pragma unroll
          for (int k = 0; k < 10; k++) frame[(uint32_t)k * stride] = w[k];
        }
        hit.u = 0.f, hit.v = 0.f, hit.slot = kNone;
        sp = 0, steps = 0, g_bits = 0u;
        state = kStNode, cur = 0u, need_load = true;  // word 0: the root node
#ifdef PB_TRAV_TWICE
        second = false, tmax0 = hit.t;
#endif
      }
      batch_cur += taken;
      if (STATS) {
        __builtin_amdgcn_s_waitcnt(0x0070);
        if (lane == 0) st.it_refill++, st.refill_ticks += (uint32_t)(wall_clock64() - t_refill);
      }
    } else {
      if (n_idle == 64) break;  // queue exhausted and every lane done
      const int n_node = __popcll(__ballot(state == kStNode)), n_tri = __popcll(__ballot(state == kStTri));
      const int n_curve = CURVES ? __popcll(__ballot(state == kStCurve)) : 0;
      const int w_node = n_node * PB_W_NODE8, w_tri = n_tri * PB_W_TRI8, w_curve = n_curve * PB_W_CURVE8;
      const int phase = (w_node >= w_tri && w_node >= w_curve) ? 0 : ((!CURVES || w_tri >= w_curve) ? 1 : 2);
      did = phase;
      if (STATS && lane == 0) {
        if (phase == 0) st.it_node++, st.ln_node += n_node;
        else if (phase == 1) st.it_tri++, st.ln_tri += n_tri;
        else st.it_curve++, st.ln_curve += n_curve;
      }
      if (phase == 0) {
        // ---- NODE phase: the eight boxes of the node in D0..D4
        if (state == kStNode) {
          if (STATS) (any_ray ? st.anodes : st.nodes)++, steps++;
          const uint32_t masks = __float_as_uint(D1.w), present = (masks | (masks >> 8) | (masks >> 16)) & 255u;
          uint32_t hits = box_test8q(D0, D1, D2, D3, D4, o_x, o_y, o_z, inv4, tmin, hit.t) & present;
          hits = octant_permute(hits, oct & 7u);
          if (hits) {
            if (g_bits & 255u) {  // the rest of the node above goes on the stack: one entry
              const uint2 e = make_uint2(g_base, g_bits);
              if (sp < kLds) stk[(uint32_t)sp * stride] = e, sp++;
              else if (sp < kStackDepth8) spill[(uint32_t)(sp - kLds) * spill_stride] = e, sp++;
              else *overflow = 1u;
            }
            g_base = __float_as_uint(D1.z);
            g_bits = hits | (masks << 8);
          }
          advance = true;
        }
      } else if (phase == 1) {
        if (state == kStTri) {
          // ---- TRI phase: the leaf's one or two triangles in ONE packed test (TriPair; dtrace.h::tri_pair_accept)
          if (STATS) steps++;
          uint32_t nt = 0u;
          const bool occ = tri_pair_accept_s<MODE == 1, STATS>(sc, D0, D1, D2, D3, D4, o_x, o_y, o_z, d.x, d.y, d.z, inv4.x, inv4.y, inv4.z, tmin, any_ray, hit, nt);
          if (STATS) (any_ray ? st.atris : st.tris) += nt;
          if (occ) {
            state = kStDoneOccluded;
            if (STATS) st.ahist[steps <= 16u ? 0 : (28 - __clz(steps - 1u) > 7 ? 7 : 28 - __clz(steps - 1u))]++, st.amax_steps = steps > st.amax_steps ? steps : st.amax_steps;
          } else {
            advance = true;
          }
        }
      } else if (CURVES && state == kStCurve) {
        // ---- CURVE phase: one linear piece of the leaf (D0 D1 D2 = P0 P1 P2, D3 = codes | piece indices | count): piece 0 = P0 P1,
        // then -- a second turn -- piece 1 = P1 P2
        if (STATS) steps++, (any_ray ? st.acurves : st.curves)++;
        const bool two = (oct & 8u) != 0u;  // this turn tests the leaf's second piece
        const uint32_t meta = __float_as_uint(D3.z);
        RayFrame f;
        f.dn = V3(frame[0], frame[stride], frame[2 * stride]), f.bx = V3(frame[3 * stride], frame[4 * stride], frame[5 * stride]);
        f.by = V3(frame[6 * stride], frame[7 * stride], frame[8 * stride]), f.inv_len = frame[9 * stride];
        float t, u, v;
        bool ok = segment_test(two ? D1 : D0, two ? D2 : D1, two ? ((meta >> 8) & 255u) : (meta & 255u), V3(o_x, o_y, o_z), f, V3(inv4.x, inv4.y, inv4.z), tmin, hit.t, t, u, v);
        const uint32_t code = __float_as_uint(two ? D3.y : D3.x);
        if (ok && !any_ray && t == hit.t && hit.slot != kNone) ok = q_gid(sc, code) < q_gid(sc, hit.slot);  // tie: the smaller canonical id wins
        if (ok) hit.t = t, hit.u = u, hit.v = v, hit.slot = code;
        if (any_ray && ok) {
          state = kStDoneOccluded, oct &= 7u;
          if (STATS) st.ahist[steps <= 16u ? 0 : (28 - __clz(steps - 1u) > 7 ? 7 : 28 - __clz(steps - 1u))]++, st.amax_steps = steps > st.amax_steps ? steps : st.amax_steps;
        } else if (!two && (meta >> 16) != 0u) {
          oct |= 8u;  // the second piece of the same leaf: no load
        } else {
          oct &= 7u;
          advance = true;
        }
      }
    }
    // ---- common tail: the next child of the current node, or of the node on top of the stack; then ONE load site
    if (advance) {
      if (!(g_bits & 255u)) {
        if (sp == 0) {
          state = kStDone;
          advance = false;
          if (STATS) {
            const int b = steps <= 16u ? 0 : (28 - __clz(steps - 1u) > 7 ? 7 : 28 - __clz(steps - 1u));
            if (any_ray) st.ahist[b]++, st.amax_steps = steps > st.amax_steps ? steps : st.amax_steps;
            else st.hist[b]++, st.max_steps = steps > st.max_steps ? steps : st.max_steps;
          }
        } else {
          sp--;
          uint2 e = stk[(uint32_t)(sp < kLds ? sp : kLds - 1) * stride];
          if (sp >= kLds) e = spill[(uint32_t)(sp - kLds) * spill_stride];
          g_base = e.x, g_bits = e.y;
        }
      }
      if (advance) {
        const uint32_t k = (uint32_t)__builtin_ctz(g_bits);  // (the low byte is not empty)
        g_bits &= g_bits - 1u;
        const uint32_t s = k ^ (oct & 7u), below = (1u << s) - 1u;
        const uint32_t five = ((g_bits >> 8) | (g_bits >> 16)) & below & 255u, four = (g_bits >> 24) & below;
        cur = g_base + 5u * (uint32_t)__builtin_popcount(five) + (CURVES ? 4u * (uint32_t)__builtin_popcount(four) : 0u);
        state = ((g_bits >> (16u + s)) & 1u) ? kStTri : ((CURVES && ((g_bits >> (24u + s)) & 1u)) ? kStCurve : kStNode);
        need_load = true;
      }
    }
#ifdef PB_TRAV_TWICE
    if ((state == kStDone || state == kStDoneOccluded) && !second) {
      second = true, hit.t = tmax0, hit.u = 0.f, hit.v = 0.f, hit.slot = kNone, sp = 0, g_bits = 0u;
      state = kStNode, cur = 0u, need_load = true;
    }
#endif
    if (need_load) {
      const float4* g = items + cur;
      D0 = g[0], D1 = g[1], D2 = g[2], D3 = g[3];
      if (!CURVES || state != kStCurve) D4 = g[4];
    }
  }
  if constexpr (!Sink::kWalk) {
    if (state >= kStDone) sink.done(tag, hit, state == kStDoneOccluded);  // rays that finished after the queue ran dry
  }
}

}  // namespace pb
