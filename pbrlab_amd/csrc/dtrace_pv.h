// dtrace_pv.h -- persistent, phase-voting BVH traversal for wave64 (device).
//
// Why: a one-ray-per-lane while/if traversal on gfx950 measured 9.6 active lanes per VALU instruction (15 %)
// -- lanes at a leaf idle while others walk nodes, and finished rays idle until the slowest lane of the wave
// ends (profiles/README.md, round 1a).  Here every lane owns a small state machine
//     IDLE -> NODE(internal node) <-> TRI(one triangle of a leaf) [<-> CURVE]
// whose next work item (a 64-byte node or a 64-byte primitive slot -- same footprint) is prefetched into
// registers as soon as it is known.  Each iteration the WAVE votes (ballot + popcount) for the phase with the
// most lanes and executes only that phase's code, so at least half of the busy lanes advance per iteration and
// the two code paths never serialise.  When enough lanes are idle they are refilled from the ray queue
// (persistent threads): the wave grabs rays in batches with one atomicAdd and deals them out by ballot rank.
//
// The intersection contract is dtrace.h's (same tests, same tie rule), so results are bit-identical to the
// simple traversal used by the test hooks.
#pragma once

#include "dtrace.h"

namespace pb {

#ifndef PB_BATCH
#define PB_BATCH 512
#endif
constexpr uint32_t kPvBatch = PB_BATCH;   // rays grabbed per atomicAdd, at most
#ifndef PB_BATCH_HEADS
#define PB_BATCH_HEADS 256  // (A/B, eight heads: 64 / 128 / 256 rays: eighth of C2 7.86 / 7.86 / 7.78 ms, frame 46.5 / 45.4 / 44.8; one head with 512: 8.49 / 44.8 -- profiles/README.md)
#endif
constexpr uint32_t kPvBatchHeads = PB_BATCH_HEADS;  // ... when the queue has several heads (k_trace)
constexpr uint32_t kPvHeadStride = 64;  // == kernels.h::kHeadStride: words between two heads (a cache line of its own each)
#ifndef PB_GUIDE
#define PB_GUIDE 2
#endif
constexpr uint32_t kPvGuide = PB_GUIDE;   // a grab takes 1 / (kPvGuide x waves) of what is left (>= 64 rays); 0 = fixed batches
#ifndef PB_LDS_STACK
#define PB_LDS_STACK 16
#endif
constexpr int kPvLdsStack = PB_LDS_STACK;  // stack entries per lane kept in LDS
#ifndef PB_LDS_STACK_DEEP
#define PB_LDS_STACK_DEEP ((PB_CURVE_RECORDS && PB_CURVE_TWO) ? 20 : 16)  // (five blocks per CU when both pieces of a curve record are tested in one turn: 30.7 KB of LDS each)
#endif
// ... and for the Q tree of scenes with curves (hair: trees 14-16 levels deep, up to three entries per level): a wave whose
// lanes are on both sides of the LDS / spill boundary runs both push paths every node turn
constexpr int kPvLdsStackDeep = PB_LDS_STACK_DEEP;
template <bool CURVES, bool WIDE>
__host__ __device__ constexpr int pv_lds_stack() { return (CURVES && WIDE) ? kPvLdsStackDeep : kPvLdsStack; }
#ifndef PB_REFILL
#define PB_REFILL 32
#endif
constexpr int kPvRefillIdle = PB_REFILL;         // refill when at least this many lanes are idle (triangle-only scenes)
#ifndef PB_REFILL_CURVES
#define PB_REFILL_CURVES 24
#endif
constexpr int kPvRefillIdleCurves = PB_REFILL_CURVES;  // the same for scenes with curves (A/B on C4, round 3: 32 -> 216.6 ms per frame, 24 -> 213.0, 16 -> 209)
// Weights of one lane in the phase vote.  Measured on C2 (A/B, k_trace ms per frame): node:tri = 2:1 64.1, 1:1 60.9,
// 3:4 59.0, 1:2 57.9, 1:3 58.0 -- primitives first: a lane parked at a leaf holds a shorter tmax for its own later box
// tests and returns to the (much more frequent) node phase, so the node phase runs with more lanes.
#ifndef PB_WALK_REFILL
#define PB_WALK_REFILL 56
#endif
#ifndef PB_W_NODE
#define PB_W_NODE 1
#endif
#ifndef PB_W_TRI
#define PB_W_TRI 2
#endif
constexpr uint32_t kRemSecond = 8u;  // rem bit: the record's second piece is next
#ifndef PB_W_CURVE
#define PB_W_CURVE 2  // (round 3, Q tree: 1 -> 216.6 ms per C4 frame, 2 -> 213.5; together with the refill at 24 idle lanes 210.2)
#endif

// kStDone / kStDoneOccluded: the ray is finished, its result still sits in the lane's registers.  Results are delivered
// (sink.done: a store, or a read-modify-write of the path's radiance) together, at the next refill, instead of by the one
// or two lanes that happen to finish in an iteration -- the delivery code costs the same however few lanes run it.
enum : uint32_t { kStIdle = 0, kStNode = 1, kStTri = 2, kStCurve = 3, kStDone = 4, kStDoneOccluded = 5 };

// Sink: what to do with a finished ray.  closest: store the hit record; shadow: resolve the contribution.
//   bool load(uint32_t idx, uint32_t& tag, V3& o, V3& d, float& tmin, float& tmax)   returns "any-hit ray"
//   void done(uint32_t tag, const Hit& h, bool occluded)
// A WALKING sink (Sink::kWalk, the random walk of subsurface scattering: k_sss_walk) chains rays: its queue entries arrive
// with a hit that is already known, and every finished ray is followed by the next one of the same walk:
//   void start(uint32_t idx, uint32_t& tag, Hit& h, V3& o, V3& d)                     the entry's ray and its known hit
//   bool next(uint32_t tag, const Hit& h, V3& o, V3& d, float& tmin, float& tmax)     one step; true = trace this ray next
// MODE: 0 = every ray wants its closest hit, 1 = every ray is an any-hit (shadow) ray, 2 = per ray (load()'s
// return value): closest-hit rays of bounce k+1 and shadow rays of bounce k share one launch and one drain.
// Traversal stack: the first kPvLdsStack entries of each lane live in LDS (stk_base[i * stride]), deeper ones
// spill to a per-thread global area (spill[(i - kPvLdsStack) * spill_stride]); keeping the LDS part small is
// what lets 6 blocks (24 waves) share a CU.
// a sink with `static constexpr bool kSplit = true` (and no walk) offers done_issue / done_finish / load_entry / load_ray
template <typename Sink, typename = void>
struct SinkSplits { static constexpr bool value = false; };
template <typename Sink>
struct SinkSplits<Sink, decltype((void)Sink::kSplit)> { static constexpr bool value = Sink::kSplit && !Sink::kWalk; };
template <typename Sink>
__device__ constexpr bool sink_splits() { return SinkSplits<Sink>::value; }
// a sink with `static constexpr bool kSuspend = true` takes part in the suspension of long rays (kernels.h::PathState::susp_turns):
//   uint32_t susp_turns();  uint32_t* susp_out();  const uint32_t* susp_in();  static constexpr bool kResumes (its queue can hold suspended rays)
//   bool suspendable(uint32_t tag)                                   a closest-hit ray whose path can skip a shading round
//   void suspended(uint32_t tag, uint32_t rec, V3 o, V3 d)           marks the path: its ray's state is record `rec` of susp_out
//   uint32_t resume_index(uint32_t tag)                              the record of a ray whose load() returned a tag with kTagResume
// Tag bits of such a sink: kTagShadow (its own), kTagNoSuspend, kTagResume; the low 28 bits are the path slot.
template <typename Sink, typename = void>
struct SinkSuspends { static constexpr bool value = false; };
template <typename Sink>
struct SinkSuspends<Sink, decltype((void)Sink::kSuspend)> { static constexpr bool value = Sink::kSuspend && !Sink::kWalk; };
template <typename Sink>
__device__ constexpr bool sink_suspends() { return SinkSuspends<Sink>::value; }
constexpr uint32_t kTagShadow = 0x80000000u, kTagNoSuspend = 0x40000000u, kTagResume = 0x20000000u, kTagHeld = 0x10000000u;
constexpr uint32_t kSuspRecWords = 72;  // == kernels.h::kSuspWords: hit (4 words) | cur, state | rem << 8 | sp << 16, 0, 0 | the stack

// WIDE: the Q tree (sc.wide: QNode, dscene.h: four children per 64-byte node with quantised boxes, compact triangle slots,
// curve pieces as chains of 16-byte points) instead of the binary one -- half the dependent fetches per ray and half the bytes
// per fetch; hit children are visited nearest first (sorted by entry distance), like the binary tree's.  `cur` then counts
// 16-byte words of sc.wide instead of 64-byte items.
template <int MODE, bool STATS, bool CURVES, bool WIDE, uint32_t NHEADS = 1u, typename Sink>
__device__ __forceinline__ void trace_pv(const DScene& sc, uint32_t n, uint32_t* head, Sink& sink, uint32_t* stk_base,
                                         uint32_t stride, uint32_t* spill, uint32_t spill_stride, TravStats& st,
                                         uint32_t* overflow, float* frame = nullptr, const float4* top = nullptr,
                                         uint32_t ntop = 0) {
  constexpr uint32_t nheads = NHEADS;
  // NHEADS (round 6): the queue [0, n) is cut into nheads equal ranges, head[h] counts inside range h; a wave draws from the head of its
  // block (block b: head b % nheads -- blocks go round the XCDs) and, when that range is dealt, from the next ones.  Why: ONE counter
  // sustains ~88 atomics per microsecond, which is what made the batches 512 rays -- ~0.4 ms of work for one wave -- and the waves that
  // drew their last batch late found the queue empty up to 0.3 ms after the first wave had (per-wave timelines: profiles/README.md);
  // eight counters take eight times the rate, so the batches can be a quarter of that.
  // top / ntop: LDS copy of nodes 0 .. ntop-1 (the breadth-first top of the tree, 64 bytes each), or none
  // frame (CURVES): 10 words per lane in LDS (frame[k * stride]), the ray's RayFrame, written when the ray is fetched
  constexpr int kLds = pv_lds_stack<CURVES, WIDE>();  // stack entries of this lane that live in LDS
  const uint32_t lane = __lane_id();
  // number of set bits of a wave mask below this lane (v_mbcnt: no per-lane mask has to stay in registers)
  auto rank_in = [](unsigned long long m) {
    return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  };
  // wave-uniform batch cursor.  Rays are grabbed kPvBatch at a time when there are plenty; when the queue is
  // short the batch shrinks to one ray per lane so that the work spreads over all resident waves instead of
  // being walked serially by a few (a 20 k-ray launch took 1.5 ms with fixed 512-ray batches).
  uint32_t batch_cur = 0, batch_end = 0;
  const uint32_t waves_total = gridDim.x * (blockDim.x >> 6);
  uint32_t hsel = nheads > 1u ? blockIdx.x % nheads : 0u, heads_left = nheads;  // the head this wave draws from now; heads not yet found empty
  const uint32_t kBatchCap = nheads > 1u ? kPvBatchHeads : kPvBatch;
  uint32_t batch = n / waves_total;
  if (batch >= 64u) {
    batch = (kPvGuide ? n / (waves_total * kPvGuide) : batch) & ~63u;
    batch = batch > kBatchCap ? kBatchCap : (batch < 64u ? 64u : batch);
  } else {
    // fewer rays than resident lanes: spread them thin (a few lanes per wave, every SIMD busy).  A launch like this is
    // bound by the latency of its longest ray, and a ray advances fastest when its wave has no other phase to vote for
    // (tail launches of 10 k rays took 100-300 us with one full wave per 64 rays).
    batch = (n + waves_total - 1u) / waves_total;
    batch = batch < 1u ? 1u : batch;
  }
  bool exhausted = (n == 0) || (sc.num_nodes == 0);
  if (sc.num_nodes == 0 && n != 0) {
    // empty scene: every ray misses
    for (;;) {
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
  if (Sink::kWalk && sc.num_nodes == 0) return;  // (nothing can be inside a medium of an empty scene)

  // per-lane state
  uint32_t state = kStIdle, tag = 0;
  V3 o(0.f), d(0.f), inv(0.f);
  float4 inv4 = make_float4(0.f, 0.f, 0.f, 0.f);  // WIDE: the same 1 / d as one aligned register quadruple (box_test4's packed operands)
  float tmin = 0.f;  // (the current tmax of the ray is hit.t)
  Hit hit = {0.f, 0.f, 0.f, kNone};
  int sp = 0;
  uint32_t steps = 0;  // STATS: node visits + primitive tests of the lane's current ray
  uint32_t rem = 0;  // primitives of the current leaf still to test after the current one
  bool any_ray = (MODE == 1);
  uint32_t cur = 0;  // index of the current 64-byte item (node; TRI/CURVE: num_nodes + slot)
  float4 D0 = make_float4(0, 0, 0, 0), D1 = D0, D2 = D0;  // prefetched node / primitive slot (a curve piece of the Q tree: its two points in D0, D1)
  float2 D3 = make_float2(0, 0);                           // a node's two child references
  float4 D3w = D0;                                         // WIDE: the four child references of a QNode (D0..D2 = origin, steps, quantised bounds)
  float4 D4 = D0;                                          // WIDE, triangle-only scenes: the fifth word of a triangle leaf (TriPair, dscene.h: D0..D3w + D4)
  // the array the traversal walks: the binary tree's nodes then the slots (64-byte items), or the Q tree (16-byte words)
  const float4* const items = WIDE ? sc.wide : reinterpret_cast<const float4*>(sc.nodes);
  const uint32_t slot0 = sc.num_nodes;  // binary tree: item index of slot 0

  unsigned long long t_turn = STATS ? __builtin_readcyclecounter() : 0ull;  // STATS: the turn's cycles go to what it did
  int did = -1;
  uint32_t drain_turns = 0u;  // loop turns since this wave found the queue empty (wave-uniform)
  // index of this wave in the grid (wave-uniform: a scalar register; the suspend record of lane l is 64 x wave_id + l)
  uint32_t wave_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  asm volatile("" : "+s"(wave_id));  // (computed HERE: left to itself the compiler sinks it behind the loop and keeps threadIdx.x alive -- in scratch -- for it)
  for (;;) {
    if (STATS) {
      const unsigned long long t_now = __builtin_readcyclecounter();
      if (lane == 0 && did >= 0) st.cyc[did] += t_now - t_turn;
      t_turn = t_now;
    }
    // `advance`: the lane needs a new item; `next` is its reference when have_next, else it is popped
    bool advance = false, have_next = false, need_load = false;
    uint32_t next = 0;
    unsigned long long idle_mask = __ballot(state == kStIdle || state >= kStDone);
    int n_idle = __popcll(idle_mask);
    // (a walking sink has work for its finished lanes even when the queue is empty: their walks go on)
    // (a walking sink's step is ~3x the work of tracing a ray: it waits for more lanes to be ready for theirs)
    constexpr int kRefillAt = Sink::kWalk ? PB_WALK_REFILL : (CURVES ? kPvRefillIdleCurves : kPvRefillIdle);
    const int n_busy_now = 64 - n_idle;
    if ((n_idle >= kRefillAt || (Sink::kWalk && n_busy_now == 0)) && (!exhausted || (Sink::kWalk && __ballot(state >= kStDone) != 0ull))) {
      // ---- refill idle lanes from the queue
      const unsigned long long t_refill = STATS ? wall_clock64() : 0ull;
      did = 3;
      if (!exhausted && batch_cur == batch_end) {
        for (;;) {  // (wave-uniform: the next head when this one's range is dealt)
          // range of head hsel: [hsel * nq, + nq) -- the last one takes the remainder; the head counts inside its range
          const uint32_t nq = n / nheads, lo_h = hsel * nq, len_h = hsel + 1u == nheads ? n - lo_h : nq;
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(head + hsel * kPvHeadStride, batch);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)base, 0));
          if (base < len_h) {  // (a head that has overshot its range keeps growing by what later visitors add: far below 2^32)
            const uint32_t end = (base + batch) < len_h ? base + batch : len_h;
            batch_cur = lo_h + base, batch_end = lo_h + end;
            // guided self-scheduling: the batches shrink as the range empties (down to one wave-full), so that the waves run
            // dry at about the same time
            if (kPvGuide && batch >= 64u) {
              const uint32_t nb = ((len_h - end) / ((waves_total / nheads + 1u) * kPvGuide)) & ~63u;
              batch = nb > kBatchCap ? kBatchCap : (nb < 64u ? 64u : nb);
            }
            break;
          }
          if (--heads_left == 0u) {
            exhausted = true, st.t_exhausted = wall_clock64();
            batch_cur = batch_end = n;
            break;
          }
          hsel = hsel + 1u == nheads ? 0u : hsel + 1u;
          batch = 64u;  // (a guest in another range: small bites)
        }
      }
      bool fresh = false;  // the lane has a new ray in (o, d, tmin, hit.t)
      uint32_t taken = 0u;  // queue entries handed out in this refill
      if (WIDE && CURVES && state >= kStDone) hit.slot = q_final_code(sc, hit.slot);  // a curve hit of the Q tree gets its hit code here, with the other lanes' 
      if constexpr (sink_splits<Sink>()) {
        // A splitting sink hands out its loads first and uses them afterwards: what the delivery of the finished rays needs
        // (done_issue) and the queue entries of the new rays (load_entry) are in flight together, one memory round trip
        // instead of three before the new rays' own loads.
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
        idle_mask = __ballot(state == kStIdle);  // (lanes whose walk goes on take no new entry)
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
      if (fresh) {
        inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        if (WIDE) inv4 = make_float4(inv.x, inv.y, inv.z, 0.f);
        if (CURVES) {
          const RayFrame f = ray_frame(d);
          const float w[10] = {f.dn.x, f.dn.y, f.dn.z, f.bx.x, f.bx.y, f.bx.z, f.by.x, f.by.y, f.by.z, f.inv_len};
#pragma unroll
          for (int k = 0; k < 10; k++) frame[(uint32_t)k * stride] = w[k];
        }
        bool resumed = false;
        if constexpr (sink_suspends<Sink>()) resumed = Sink::kResumes && (tag & kTagResume) != 0u;
        if (resumed) {
          // a ray suspended by the previous launch: it goes on where it stopped (its item is fetched again at the load site below)
          if constexpr (sink_suspends<Sink>()) {
            tag &= ~kTagResume;
            const uint32_t* rec = sink.susp_in() + (size_t)sink.resume_index(tag) * kSuspRecWords;
            hit.t = __uint_as_float(rec[0]), hit.u = __uint_as_float(rec[1]), hit.v = __uint_as_float(rec[2]), hit.slot = rec[3];
            cur = rec[4];
            const uint32_t meta = rec[5];
            state = meta & 255u, rem = (meta >> 8) & 255u, sp = (int)(meta >> 16);
            for (int i = 0; i < sp; i++) {
              const uint32_t v = rec[8 + i];
              if (i < kLds) stk_base[(uint32_t)i * stride] = v;
              else spill[(uint32_t)(i - kLds) * spill_stride] = v;
            }
            steps = 0, need_load = true;
          }
        } else {
          hit.u = 0.f, hit.v = 0.f, hit.slot = kNone;
          sp = 0, steps = 0;
          advance = true, have_next = true, next = 0u;  // root is always an internal node
          if constexpr (Sink::kWalk && WIDE) {
            // a random walk's ray starts below the root where its instance has an entry (dscene.h::SssEntry): `next` becomes the entry node
            // and the foreign references the ray's interval meets go on the stack
            sink.entry(sc, o, d, inv, tmin, hit.t, next, [&](uint32_t ref) {
              if (sp < kLds) stk_base[(uint32_t)sp * stride] = ref;
              else spill[(uint32_t)(sp - kLds) * spill_stride] = ref;
              sp++;
            });
          }
        }
      }
      batch_cur += taken;
      if (STATS) {
        // (the tick count is taken after the new rays' loads have landed: the wave waits for them before it goes on)
        __builtin_amdgcn_s_waitcnt(0x0070);
        if (lane == 0) st.it_refill++, st.refill_ticks += (uint32_t)(wall_clock64() - t_refill);
      }
    } else {
      if (n_idle == 64) break;  // queue exhausted and every lane done
      if constexpr (sink_suspends<Sink>()) {
        // The drain: the queue is empty, this wave's last rays are finishing one by one and the chip runs nearly empty while they
        // do.  After susp_turns more turns the wave stops as soon as every ray it still traces can be suspended (closest-hit rays:
        // below); the launch that follows takes them up again in its bulk phase.
        if (exhausted && sink.susp_turns() != 0u) {
          if (drain_turns < sink.susp_turns()) drain_turns++;
          else if (__ballot(state >= kStNode && state <= kStCurve && !sink.suspendable(tag)) == 0ull) break;
        }
      }
      unsigned long long node_mask = __ballot(state == kStNode);
      unsigned long long tri_mask = __ballot(state == kStTri);
      int n_node = __popcll(node_mask), n_tri = __popcll(tri_mask);
      int n_curve = CURVES ? __popcll(__ballot(state == kStCurve)) : 0;
      // vote: the phase that advances the most lanes per instruction issued (a node step is cheaper than a primitive step)
      const int w_node = n_node * PB_W_NODE, w_tri = n_tri * PB_W_TRI, w_curve = n_curve * PB_W_CURVE;
      const int phase = (w_node >= w_tri && w_node >= w_curve) ? 0 : ((!CURVES || w_tri >= w_curve) ? 1 : 2);
      did = phase;
      if (STATS && lane == 0) {
        if (phase == 0) st.it_node++, st.ln_node += n_node;
        else if (phase == 1) st.it_tri++, st.ln_tri += n_tri;
        else st.it_curve++, st.ln_curve += n_curve;
      }
      if (phase == 0) {
        // ---- NODE phase
        if (WIDE && state == kStNode) {
          if (STATS) (any_ray ? st.anodes : st.nodes)++, steps++;
#ifdef PB_DIAG_EXTRA_VALU  // diagnostic build: PB_DIAG_EXTRA_VALU dependent fma more per node turn (is the kernel bound by its VALU instructions?)
          {
            float x = D0.x;
#pragma unroll
            for (int k2 = 0; k2 < PB_DIAG_EXTRA_VALU; k2++) asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(x) : "v"(D0.y));
            if (x == 1.2345e-30f) tmin = x;
          }
#endif
#ifdef PB_DIAG_EXTRA_LOAD  // diagnostic build: one more 16-byte load per lane and node turn (the node's own first word: an L1 hit)
          {
            const float4* qd = items + cur;
            asm volatile("" : "+v"(qd));
            const float4 x = *qd;
            if (x.x == 1.2345e-30f) tmin = x.y;
          }
#endif
          uint32_t k[4];
          wide_node_keys(D0, D1, D2, D3w, o, inv4, tmin, hit.t, k);
          auto ref_of = [&](uint32_t key) { return wide_ref(D3w, key); };
          advance = true;
          have_next = k[0] != kWideMiss;
          next = ref_of(k[0]);
          if (!CURVES && k[1] != kWideMiss) {  // the other hit children go on the stack, farthest first
            // (triangle-only scenes: shallow trees, the LDS part of the stack nearly always has room)
            if (sp + 3 <= kLds) {
              // (an entry written for a child that was not hit lies above the new top or is overwritten by the next one)
              uint32_t p = (uint32_t)sp;
              stk_base[p * stride] = ref_of(k[3]);
              p += (k[3] != kWideMiss) ? 1u : 0u;
              stk_base[p * stride] = ref_of(k[2]);
              p += (k[2] != kWideMiss) ? 1u : 0u;
              stk_base[p * stride] = ref_of(k[1]);
              sp = (int)p + 1;
            } else {
#pragma unroll
              for (int j = 3; j >= 1; j--) {
                if (k[j] == kWideMiss) continue;
                const uint32_t farc = ref_of(k[j]);
                if (sp < kLds) {
                  stk_base[(uint32_t)sp * stride] = farc;
                  sp++;
                } else if (sp < kStackDepth) {
                  spill[(uint32_t)(sp - kLds) * spill_stride] = farc;
                  sp++;
                } else {
                  *overflow = 1u;
                }
              }
            }
          } else if (k[1] != kWideMiss) {
            const uint32_t r3 = ref_of(k[3]), r2 = ref_of(k[2]), r1 = ref_of(k[1]);
            const uint32_t p3 = (uint32_t)sp, p2 = p3 + ((k[3] != kWideMiss) ? 1u : 0u), p1 = p2 + ((k[2] != kWideMiss) ? 1u : 0u);
            // ONE of the two paths per wave: a wave whose lanes sit on both sides of the LDS / spill boundary used to run both
            // every node turn (hair: trees 14-16 levels deep, up to three entries per level)
            if (__ballot(p1 >= (uint32_t)kLds) == 0ull) {
              // (an entry written for a child that was not hit lies above the new top or is overwritten by the next one)
              stk_base[p3 * stride] = r3;
              stk_base[p2 * stride] = r2;
              stk_base[p1 * stride] = r1;
              sp = (int)p1 + 1;
            } else {
              auto put = [&](uint32_t pos, uint32_t ref) {
                if (pos < (uint32_t)kLds) stk_base[pos * stride] = ref;
                else if (pos < (uint32_t)kStackDepth) spill[(pos - (uint32_t)kLds) * spill_stride] = ref;
                else *overflow = 1u;
              };
              if (k[3] != kWideMiss) put(p3, r3);
              if (k[2] != kWideMiss) put(p2, r2);
              put(p1, r1);
              sp = (int)(p1 + 1u < (uint32_t)kStackDepth ? p1 + 1u : (uint32_t)kStackDepth);
            }
          }
        } else if (state == kStNode) {
          if (STATS) (any_ray ? st.anodes : st.nodes)++, steps++;
          uint32_t c0 = __float_as_uint(D3.x), c1 = __float_as_uint(D3.y);
          float t0, t1;
          bool h0, h1;
          box_test2(D0, D1, D2, o, inv, tmin, hit.t, h0, h1, t0, t1);
          bool swap = h1 && (!h0 || t1 < t0);
          uint32_t nearc = swap ? c1 : c0, farc = swap ? c0 : c1;
          advance = true;
          have_next = h0 || h1;
          next = nearc;
          if (h0 && h1) {
            if (sp < kLds) {
              stk_base[(uint32_t)sp * stride] = farc;
              sp++;
            } else if (sp < kStackDepth) {
              spill[(uint32_t)(sp - kLds) * spill_stride] = farc;
              sp++;
            } else {
              *overflow = 1u;
            }
          }
        }
      } else {
        // ---- TRI / CURVE phase: one primitive per lane
        const bool is_tri = state == kStTri, is_curve = CURVES && state == kStCurve;
        const bool mine = (phase == 1) ? is_tri : is_curve;
        if (WIDE && !CURVES && mine) {
          // a triangle leaf of the Q tree of a triangle-only scene: its one or two triangles in ONE packed test (TriPair; dtrace.h::tri_pair_accept)
          if (STATS) steps++;
          uint32_t nt = 0u;
          const bool occ = tri_pair_accept<MODE == 1, STATS>(sc, D0, D1, D2, D3w, D4, o, d, V3(inv4.x, inv4.y, inv4.z), tmin, any_ray, hit, nt);
          if (STATS) (any_ray ? st.atris : st.tris) += nt;
          if (occ) {
            state = kStDoneOccluded;
            if (STATS) st.ahist[steps <= 16u ? 0 : (28 - __clz(steps - 1u) > 7 ? 7 : 28 - __clz(steps - 1u))]++, st.amax_steps = steps > st.amax_steps ? steps : st.amax_steps;
          } else {
            advance = true;  // leaf done: pop
          }
        } else if (PB_CURVE_RECORDS && PB_CURVE_TWO && WIDE && CURVES && mine && is_curve) {
          // A curve leaf of the Q tree is a RECORD (round 6, dscene.h): the end points a0 a1 (b0 b1) of its one or two pieces in D0, D1
          // (D2, D3w) (rem & kCurvePairBit: a second piece; the pieces' indices in their cubics are the low bits of cur and of rem).
          // PB_CURVE_TWO: both pieces in this one turn (needs 88-91 registers: five blocks per CU); otherwise one piece per turn -- the
          // second piece's turn needs no fetch, the record is in the lane's registers (rem & kRemSecond) -- at the register budget of
          // rounds 3-5 (six blocks per CU).  Piece by piece the operations, their order and the accept rule are the same: the same bits.
          const bool two = (rem & kCurvePairBit) != 0u;
          RayFrame f;
          f.dn = V3(frame[0], frame[stride], frame[2 * stride]), f.bx = V3(frame[3 * stride], frame[4 * stride], frame[5 * stride]);
          f.by = V3(frame[6 * stride], frame[7 * stride], frame[8 * stride]), f.inv_len = frame[9 * stride];
          const V3 i3(inv4.x, inv4.y, inv4.z);
          const uint32_t pt = (cur & ~3u) - sc.q_pt0;  // point index of the first piece
          bool occ = false;
          const bool second = false;
          if (STATS) steps++, (any_ray ? st.acurves : st.curves)++;
          {
            // (one piece per turn: the piece of this turn is in D0, D1 -- the second piece was moved there when the first was done)
            const V3 pa = segment_project(D0, o, f), pb = segment_project(D1, o, f);
            float t, u, v;
            bool ok = segment_core(D0, D1, pa, pb, second ? (rem & 3u) : (cur & 3u), o, f.inv_len, i3, tmin, hit.t, t, u, v);
            const uint32_t code = kQPointHit | (pt + (second ? 2u : 0u));
            if (ok && !any_ray && t == hit.t && hit.slot != kNone) ok = q_gid(sc, code) < q_gid(sc, hit.slot);
            if (ok) hit.t = t, hit.u = u, hit.v = v, hit.slot = code;
            occ = any_ray && ok;
          }
#if PB_CURVE_TWO
          if (two && !occ) {
            if (STATS) steps++, (any_ray ? st.acurves : st.curves)++;
            const V3 pa = segment_project(D2, o, f), pb = segment_project(D3w, o, f);
            float t, u, v;
            bool ok = segment_core(D2, D3w, pa, pb, rem & 3u, o, f.inv_len, i3, tmin, hit.t, t, u, v);
            const uint32_t code = kQPointHit | (pt + 2u);
            if (ok && !any_ray && t == hit.t && hit.slot != kNone) ok = q_gid(sc, code) < q_gid(sc, hit.slot);
            if (ok) hit.t = t, hit.u = u, hit.v = v, hit.slot = code;
            occ = any_ray && ok;
          }
#endif
          if (occ) {
            state = kStDoneOccluded;
            if (STATS) st.ahist[steps <= 16u ? 0 : (28 - __clz(steps - 1u) > 7 ? 7 : 28 - __clz(steps - 1u))]++, st.amax_steps = steps > st.amax_steps ? steps : st.amax_steps;
          } else if (!PB_CURVE_TWO && two && !second) {
            rem |= kRemSecond;  // the second piece: next curve turn, nothing to fetch -- it moves to where the turn's piece is expected
            D0 = D2, D1 = D3w;
          } else {
            advance = true;  // leaf done: pop
          }
        } else if (mine) {
          if (STATS) steps++;
          float t, u, v;
          bool ok;
          if (!CURVES || is_tri) {
            if (STATS) (any_ray ? st.atris : st.tris)++;
            ok = tri_test(ld3(D0), ld3(D1), ld3(D2), o, d, WIDE ? V3(inv4.x, inv4.y, inv4.z) : inv, tmin, t, u, v) && (t <= hit.t);
          } else {
            if (STATS) (any_ray ? st.acurves : st.curves)++;
            RayFrame f;
            f.dn = V3(frame[0], frame[stride], frame[2 * stride]), f.bx = V3(frame[3 * stride], frame[4 * stride], frame[5 * stride]);
            f.by = V3(frame[6 * stride], frame[7 * stride], frame[8 * stride]), f.inv_len = frame[9 * stride];
            // (a curve record of the Q tree, one piece per turn: the turn's piece is in D0, D1 -- the second piece, kRemSecond, was moved
            // there when the first was done; the pieces' indices in their cubics are the low bits of cur and of rem)
            const uint32_t sub = !WIDE ? __float_as_uint(D2.x) : (PB_CURVE_RECORDS ? ((rem & kRemSecond) ? (rem & 3u) : (cur & 3u)) : ((cur - sc.q_pt0) & 3u));
            ok = segment_test(D0, D1, sub, o, f, WIDE ? V3(inv4.x, inv4.y, inv4.z) : inv, tmin, hit.t, t, u, v);
          }
          // the hit code: slot + routing bits (dscene.h); Q tree: a triangle slot carries its code, a curve hit is held as its point
          const uint32_t qpt = PB_CURVE_RECORDS ? (cur & ~3u) - sc.q_pt0 + ((rem & kRemSecond) ? 2u : 0u) : cur - sc.q_pt0;
          const uint32_t code = !WIDE ? ((cur - slot0) | __float_as_uint(D2.w))
                                      : ((CURVES && is_curve) ? (kQPointHit | qpt) : __float_as_uint(D2.w));
          if (ok && !any_ray && t == hit.t && hit.slot != kNone)
            ok = WIDE ? q_gid(sc, code) < q_gid(sc, hit.slot) : sc.shade[cur - slot0].gid < sc.shade[hit.slot & kHitSlotMask].gid;
          if (ok) {
            hit.t = t, hit.u = u, hit.v = v, hit.slot = code;
          }
          if (any_ray && ok) {
            state = kStDoneOccluded;
            if (STATS) st.ahist[steps <= 16u ? 0 : (28 - __clz(steps - 1u) > 7 ? 7 : 28 - __clz(steps - 1u))]++, st.amax_steps = steps > st.amax_steps ? steps : st.amax_steps;
          } else if (PB_CURVE_RECORDS && WIDE && CURVES && is_curve) {
            if ((rem & (kCurvePairBit | kRemSecond)) == kCurvePairBit) {
              rem |= kRemSecond;  // the record's second piece: next curve turn, nothing to fetch -- it moves to where the turn's piece is expected
              D0 = D2, D1 = D3w;
            } else {
              advance = true;  // leaf done: pop
            }
          } else if (rem != 0u) {  // next primitive of the same leaf
            rem--, cur += WIDE ? ((CURVES && is_curve) ? 1u : 3u) : 1u;
            need_load = true;
          } else {
            advance = true;  // leaf done: pop
          }
        }
      }
    }
    // ---- common tail: pop / finish / decode the next item, then ONE load site for every lane that moved
    if (advance) {
      if (!have_next) {
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
          // the LDS read is unconditional (a clamped index), the spill read the rare exception: one ds_read instead of a
          // flat load behind an address select
          next = stk_base[(uint32_t)(sp < kLds ? sp : kLds - 1) * stride];
          if (sp >= kLds) next = spill[(uint32_t)(sp - kLds) * spill_stride];
        }
      }
      if (advance) {
        need_load = true;
        if (next & kLeafBit) {
          const uint32_t first = (next & 0x3FFFFFFFu) >> 3;
          rem = next & 7u;
          state = (CURVES && (next & kCurveBit)) ? kStCurve : kStTri;
          if (WIDE && !CURVES) rem = 0u;  // (a triangle leaf of a triangle-only scene's Q tree is one item: TriPair)
          if (WIDE) cur = (CURVES && (next & kCurveBit)) ? sc.q_pt0 + first : sc.q_tri0 + (CURVES ? 3u : kTriPairWords) * first;
          else cur = first + slot0;  // slots follow the nodes in one array of 64-byte items
        } else {
          cur = WIDE ? 4u * next : next;
          state = kStNode;
        }
      }
    }
    if (WIDE && need_load) {
      if (!CURVES && kTopNodesWide > 0 && cur < 4u * ntop) {  // the top of the Q tree (triangle-only scenes): every ray passes through it
        const float4* g = top + cur;
        D0 = g[0], D1 = g[1], D2 = g[2], D3w = g[3];
      } else {
        // (a curve record: four words; the low bits of its address carry a piece index)
        const bool at_rec = PB_CURVE_RECORDS && CURVES && state == kStCurve;
        // (a ray that was suspended at the second piece of its record -- kRemSecond -- resumes with that piece where the turn expects it)
        const float4* g = items + (at_rec ? (cur & ~3u) + ((!PB_CURVE_TWO && (rem & kRemSecond)) ? 2u : 0u) : cur);
        D0 = g[0], D1 = g[1];
        if (!CURVES || state != kStCurve || at_rec) D2 = g[2];
        if (!CURVES || state == kStNode || at_rec) D3w = g[3];
        if (!CURVES && state == kStTri) D4 = g[4];
      }
    } else if (need_load) {
      if (kTopNodes > 0 && cur < ntop) {  // the top of the tree: every ray passes through it
        const float4* g = top + cur * 4u;
        D0 = g[0], D1 = g[1], D2 = g[2];
        D3 = *reinterpret_cast<const float2*>(g + 3);
      } else {
        const float4* g = reinterpret_cast<const float4*>(sc.nodes + cur);  // node, or slot cur - num_nodes
        D0 = g[0], D1 = g[1], D2 = g[2];
        if (state == kStNode) D3 = *reinterpret_cast<const float2*>(g + 3);
      }
    }
  }
  if constexpr (!Sink::kWalk) {
    if (state >= kStDone) {  // rays that finished after the queue ran dry
      if (WIDE && CURVES) hit.slot = q_final_code(sc, hit.slot);
      sink.done(tag, hit, state == kStDoneOccluded);
    }
  }
  if constexpr (sink_suspends<Sink>()) {
    // SUSPEND what is still being traced (the wave left its loop at the end of the drain): the stack, the current item and the hit
    // held so far go to the lane's record (one per resident thread: no counter), the path is marked (sink.suspended: its hit code
    // becomes kHitSuspended) and the ray goes on -- from this very point, so its hit is what it would have been -- in the next
    // iteration's launch.  The prefetched item is not kept (the resuming lane fetches it again), so this costs the loop no register.
    if (state >= kStNode && state <= kStCurve) {
      const uint32_t j = wave_id * 64u + __lane_id();
      uint32_t* rec = sink.susp_out() + (size_t)j * kSuspRecWords;
      rec[0] = __float_as_uint(hit.t), rec[1] = __float_as_uint(hit.u), rec[2] = __float_as_uint(hit.v), rec[3] = hit.slot;
      rec[4] = cur, rec[5] = state | (rem << 8) | ((uint32_t)sp << 16);
      for (int i = 0; i < sp; i++) rec[8 + i] = i < kLds ? stk_base[(uint32_t)i * stride] : spill[(uint32_t)(i - kLds) * spill_stride];
      sink.suspended(tag, j, o, d);
      if (STATS) (any_ray ? st.suspended_any : st.suspended)++;
    }
  }
}

}  // namespace pb
