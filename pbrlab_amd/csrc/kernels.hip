// kernels.hip -- the wavefront path tracer's device kernels (gfx950, wave64).
//
// One path = one (pixel, pass) sample.  Path state lives in HBM, in records indexed by the path slot
// (kernels.h::PathState); stages exchange *queues of slot ids*, compacted tile-wise (2048 entries, one atomic per
// tile and queue).  One iteration of the host loop (pbrhip.cpp::render_impl) =
//
//   k_trace           rtcIntersect1 (raytracer_impl.cc:268-278) for the live paths' rays AND rtcOccluded1 + the tail of
//                     DirectIllumination (shader-utils.h:192-208) for the shadow rays of the previous bounce
//   k_classify        routes each traced path to its closure queue (in-medium / principled / hair), drops misses
//   k_shade_principled GetRadiance head (render.cc:31-68: implicit light + MIS, Russian roulette) +
//                     CyclesPrincipledShader (cycles-principled-shader.cc:414-484) incl. SSS entry
//   k_shade_hair      HairShader             (hair-shader.cc:153-229)
//   k_sss_step        RandomWalkSubsurface loop body + exit (random-walk-sss.h:287-405)
//   k_compact         result words of the shade kernels -> next ray queue + shadow-ray queue
//   k_advance         queue flip
//
// k_tail runs the same per-path functions in a loop once few paths are left; k_generate (render.cc:160-171) and
// k_accumulate (render.cc:175-183) bracket a chunk of passes.
#include "dshade.h"
#include "kernels.h"
#include "dtrace_pv.h"
#include "dtrace_quad.h"

namespace pb {

constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
  return v;
}

__device__ __forceinline__ float4 mk4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }

// ------------------------------------------------------------------ the camera sample (render.cc:160-171)
// Path slot0 + j of a group: its pixel, its pass, the two draws of its own generator, the ray direction -- and the generator's
// state after them.  A function of j: the first k_trace (TraceSinkT<.., FIRST>) and the first shading (path_head) evaluate it;
// nothing of it is stored.
__device__ __forceinline__ void camera_sample(const PathState& P, uint32_t j, V3& dir, uint64_t& rng_state) {
  const uint32_t R = P.pass_run, q = j / R;  // (PathState::pass_run: runs of R passes of one pixel are adjacent)
  const uint32_t pass = P.first_pass + (q / P.npix) * R + j % R;
  const uint32_t gpix = P.pix_index[q % P.npix];
  const uint32_t x = gpix % P.width, y = gpix / P.width;
  Rng rng = rng_seed(((uint64_t)pass << 32) + (uint64_t)gpix, P.seed_seq);
  const float jx = draw(rng);
  const float jy = draw(rng);
  const V3 org(P.cam.org[0], P.cam.org[1], P.cam.org[2]);
  const V3 target(P.cam.x_corner + P.cam.dx * ((float)x + jx), P.cam.y_corner - P.cam.dy * ((float)y + jy), P.cam.z_corner);
  dir = normalize_raw(target - org);
  rng_state = rng.state;
}
// what is left of the generation pass: the radiance of the group's paths starts at 0 (origin, throughput, flags, direction,
// generator state and queue entry of the first bounce are implied: PathState::first)
__global__ __launch_bounds__(kBlock) void k_generate(PathState P, uint32_t npaths) {
  for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < npaths; j += gridDim.x * kBlock) P.L[P.slot0 + j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------ k_trace
// ONE persistent phase-voting traversal launch per wavefront iteration (dtrace_pv.h): it serves the closest-hit
// rays of this bounce (rtcIntersect1, raytracer_impl.cc:268-278; queue q_in) AND the shadow rays the previous
// bounce's shading produced (rtcOccluded1 + tail of DirectIllumination, shader-utils.h:192-208; queue q_shadow).
// The two sets are independent -- shading of this bounce waits for both -- so they share one drain.
//   closest ray: store the hit record
//   shadow ray:  kShNormal   L += c_vis when unoccluded
//                kShSssEntry A  = 0 + c_vis when unoccluded (first NEE of a path that entered the medium)
//                kShSssExit  L += unoccluded ? c_vis : c_occ
// SPLIT (dtrace_pv.h): the loads of a refill are issued together, then used.  A/B, frame ms: hair scene (C4) 236.2 -> 231.9,
// triangle-only scenes 57.2 -> 58.5 (C2, the delivery's L load becomes unconditional), 436 -> 436 (C3): curve scenes only.
// FIRST: the launch is a group's first (camera rays only: computed, not loaded; no shadow rays yet)
template <bool SPLIT, bool FIRST = false>
struct TraceSinkT {
  static constexpr bool kWalk = false;
  static constexpr bool kSplit = SPLIT;
  static constexpr bool kSuspend = true;
  const PathState& P;
  uint32_t n_closest, n_shadow;
  // resumable rays (dtrace_pv.h, kernels.h::PathState::susp_turns)
  __device__ __forceinline__ uint32_t susp_turns() const { return P.susp_turns; }
  static constexpr bool kResumes = !FIRST;  // (a group's first launch traces camera rays only)
  __device__ __forceinline__ uint32_t* susp_out() const { return P.susp_out; }
  __device__ __forceinline__ const uint32_t* susp_in() const { return P.susp_in; }
  // closest-hit rays of paths outside media; in scenes without media shadow rays as well (kernels.h::PathState::hold)
  __device__ __forceinline__ bool suspendable(uint32_t tag) const {
    return (tag & kTagNoSuspend) == 0u && (!(tag & kTagShadow) || (!FIRST && P.no_medium != 0u));
  }
  __device__ __forceinline__ void suspended(uint32_t tag, uint32_t rec, V3 o, V3 d) const {
    const uint32_t p = tag & kQPathMask;
    if (!FIRST && (tag & kTagShadow)) {
      // a shadow ray: its record index goes where a medium exit would keep its second contribution (unused without media), the path is
      // held, and the ray re-queues itself for the next launch (one atomic per wave; k_compact appends behind it)
      P.sh_e[p] = make_float4(__uint_as_float(rec), 0.f, 0.f, 0.f);
      P.hold[p] = 1u;
      const unsigned long long m = __ballot(true);
      const int leader = __ffsll((long long)m) - 1;
      uint32_t base = 0u;
      if ((int)__lane_id() == leader) base = atomicAdd(&P.counts[kCntShadow], (uint32_t)__popcll(m));
      base = (uint32_t)__shfl((int)base, leader);
      P.q_shadow[base + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = p | kQResume;
      return;
    }
    P.hit[p] = make_float4(__uint_as_float(rec), 0.f, 0.f, __uint_as_float(kHitSuspended));
    if (FIRST) {  // a camera ray was computed, not loaded: the launch that resumes it loads it like any other ray
      P.ray_o[p] = mk4(o, 0.0f);
      P.ray_d[p] = mk4(d, kInf);
    }
  }
  __device__ __forceinline__ uint32_t resume_index(uint32_t tag) const {
    const uint32_t p = tag & kQPathMask;
    return __float_as_uint((tag & kTagShadow) ? P.sh_e[p].x : P.hit[p].x);
  }
  // a trace-queue entry's flags as tag bits: a path inside a medium is never suspended (with media) / a held path's ray is not
  // traced again (without: kQHold = the same bit), a suspended ray resumes
  __device__ __forceinline__ uint32_t entry_tag(uint32_t entry) const {
    return (entry & kQPathMask) | ((entry & kQSssBit) ? (kTagNoSuspend | (P.no_medium ? kTagHeld : 0u)) : 0u) | ((entry & kQResume) ? kTagResume : 0u);
  }
  // a shadow-queue entry: kQResume = the ray was suspended (its path is held: cleared when the ray is delivered)
  static __device__ __forceinline__ uint32_t shadow_tag(uint32_t entry) {
    return (entry & kQPathMask) | kTagShadow | ((entry & kQResume) ? (kTagResume | kTagHeld) : 0u);
  }
  // queue position -> (shadow ray?, index in its queue): the shadow rays of the previous bounce first (shadow_first) or last
  __device__ __forceinline__ bool is_shadow(uint32_t idx, uint32_t& k) const {
    if (P.shadow_first) {
      k = idx < n_shadow ? idx : idx - n_shadow;
      return idx < n_shadow;
    }
    k = idx < n_closest ? idx : idx - n_closest;
    return idx >= n_closest;
  }
  struct Pending {
    float4 c, L;  // a shadow ray's pending contribution (+ mode) and its path's radiance
  };
  __device__ __forceinline__ Pending done_issue(uint32_t tag, bool occluded) const {
    Pending q = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    if ((tag & kTagShadow) && !(occluded && P.no_medium)) {
      const uint32_t p = tag & kQPathMask;
      q.c = P.sh_c[p], q.L = P.L[p];
    }
    return q;
  }
  __device__ __forceinline__ void done_finish(uint32_t tag, const Pending& q, const Hit& h, bool occluded) const {
    const uint32_t p = tag & kQPathMask;
    if (!(tag & kTagShadow)) {
      if (!(tag & kTagHeld)) P.hit[p] = make_float4(h.t, h.u, h.v, __uint_as_float(h.slot));  // (a held path's hit record stands)
      else if (P.stats) atomicAdd(&P.stats[kStatHeld], 1ull);  // (statistics renders: the entry was no ray)
      return;
    }
    if (tag & kTagHeld) P.hold[p] = 0u;  // (a shadow ray that was suspended: its path may be shaded again)
    if (occluded && P.no_medium) return;  // (an ordinary shadow ray that is occluded adds nothing)
    const uint32_t mode = __float_as_uint(q.c.w);
    if (mode == kShSssEntry) {
      if (!occluded) P.sss_A[p] = make_float4(0.0f + q.c.x, 0.0f + q.c.y, 0.0f + q.c.z, 0.0f);
    } else if (!occluded || mode == kShSssExit) {
      V3 add(q.c.x, q.c.y, q.c.z);
      if (occluded) add = ld3(P.sh_e[p]);
      P.L[p] = make_float4(q.L.x + add.x, q.L.y + add.y, q.L.z + add.z, q.L.w);
    }
  }
  __device__ __forceinline__ void camera_ray(uint32_t idx, uint32_t& tag, V3& o, V3& d, float& tmin, float& tmax) const {
    tag = P.slot0 + idx;
    uint64_t state;
    camera_sample(P, idx, d, state);
    o = V3(P.cam_org[0], P.cam_org[1], P.cam_org[2]), tmin = 0.0f, tmax = kInf;
  }
  __device__ __forceinline__ uint32_t load_entry(uint32_t idx) const {
    if (FIRST) return 0u;
    uint32_t k;
    return is_shadow(idx, k) ? P.q_shadow_in[k] : P.q_in[k];
  }
  __device__ __forceinline__ bool load_ray(uint32_t idx, uint32_t entry, uint32_t& tag, V3& o, V3& d, float& tmin, float& tmax) const {
    if (FIRST) {
      camera_ray(idx, tag, o, d, tmin, tmax);
      return false;
    }
    uint32_t k;
    if (!is_shadow(idx, k)) {
      tag = entry_tag(entry);
      const uint32_t p = entry & kQPathMask;
      const float4 o4 = P.ray_o[p], d4 = P.ray_d[p];
      o = ld3(o4), d = ld3(d4), tmin = o4.w, tmax = (tag & kTagHeld) ? -1.0f : d4.w;  // (a held path: an empty interval -- nothing is traced)
      return false;
    }
    tag = shadow_tag(entry);
    const uint32_t p = entry & kQPathMask;
    float4 o4 = P.ray_o[p], d4 = P.sh_d[p];
    o = ld3(o4), d = ld3(d4), tmin = o4.w, tmax = d4.w;
    return true;
  }
  __device__ __forceinline__ bool load(uint32_t idx, uint32_t& tag, V3& o, V3& d, float& tmin, float& tmax) const {
    if (FIRST) {
      camera_ray(idx, tag, o, d, tmin, tmax);
      return false;
    }
    uint32_t k;
    if (!is_shadow(idx, k)) {
      const uint32_t entry = P.q_in[k];
      tag = entry_tag(entry);
      const uint32_t p = entry & kQPathMask;
      const float4 o4 = P.ray_o[p], d4 = P.ray_d[p];
      o = ld3(o4), d = ld3(d4), tmin = o4.w, tmax = (tag & kTagHeld) ? -1.0f : d4.w;  // (a held path: an empty interval -- nothing is traced)
      return false;
    }
    const uint32_t entry = P.q_shadow_in[k];
    tag = shadow_tag(entry);
    const uint32_t p = entry & kQPathMask;
    float4 o4 = P.ray_o[p], d4 = P.sh_d[p];
    o = ld3(o4), d = ld3(d4), tmin = o4.w, tmax = d4.w;
    return true;
  }
  __device__ __forceinline__ void done(uint32_t tag, const Hit& h, bool occluded) const {
    const uint32_t p = tag & kQPathMask;
    if (!(tag & kTagShadow)) {
      if (!(tag & kTagHeld)) P.hit[p] = make_float4(h.t, h.u, h.v, __uint_as_float(h.slot));  // (a held path's hit record stands)
      else if (P.stats) atomicAdd(&P.stats[kStatHeld], 1ull);  // (statistics renders: the entry was no ray)
      return;
    }
    if (tag & kTagHeld) P.hold[p] = 0u;  // (a shadow ray that was suspended: its path may be shaded again)
    if (occluded && P.no_medium) return;  // (an ordinary shadow ray that is occluded adds nothing: its payload is not even read)
    const float4 c = P.sh_c[p];
    const uint32_t mode = __float_as_uint(c.w);
    if (mode == kShSssEntry) {
      if (!occluded) P.sss_A[p] = make_float4(0.0f + c.x, 0.0f + c.y, 0.0f + c.z, 0.0f);
    } else if (!occluded || mode == kShSssExit) {
      V3 add(c.x, c.y, c.z);
      if (occluded) add = ld3(P.sh_e[p]);
      float4 L = P.L[p];
      P.L[p] = make_float4(L.x + add.x, L.y + add.y, L.z + add.z, L.w);
    }
  }
};

using TraceSink = TraceSinkT<false>;

__device__ __forceinline__ void trace_stats_out(const PathState& P, const TravStats& st, uint32_t n_closest, uint32_t n_shadow) {
    uint32_t v[13] = {st.nodes, st.tris, st.curves, st.anodes, st.atris, st.acurves, st.it_node, st.it_tri, st.it_curve,
                      st.it_refill, st.ln_node, st.ln_tri, st.ln_curve};
    const uint32_t idx[13] = {kStatClosestNodes, kStatClosestTris, kStatClosestCurves, kStatShadowNodes, kStatShadowTris,
                              kStatShadowCurves, kStatPvItNode, kStatPvItTri, kStatPvItCurve, kStatPvItRefill,
                              kStatPvLnNode, kStatPvLnTri, kStatPvLnCurve};
    for (int i = 0; i < 13; i++) {
      uint32_t s = wave_sum(v[i]);
      if (__lane_id() == 0 && s) atomicAdd(&P.stats[idx[i]], (unsigned long long)s);
    }
    for (int i = 0; i < 8; i++) {
      uint32_t s = wave_sum(st.hist[i]);
      if (__lane_id() == 0 && s) atomicAdd(&P.stats[kStatStepHist0 + i], (unsigned long long)s);
    }
    for (int i = 0; i < 8; i++) {
      uint32_t s = wave_sum(st.ahist[i]);
      if (__lane_id() == 0 && s) atomicAdd(&P.stats[kStatAnyHist0 + i], (unsigned long long)s);
    }
    atomicMax(&P.stats[kStatAnyMaxSteps], (unsigned long long)st.amax_steps);
    atomicMax(&P.stats[kStatMaxSteps], (unsigned long long)st.max_steps);
    if (__lane_id() == 0)
      for (int i = 0; i < 4; i++)
        if (st.cyc[i]) atomicAdd(&P.stats[kStatCycNode + i], st.cyc[i]);
    {
      const uint32_t su = wave_sum(st.suspended), sa = wave_sum(st.suspended_any);
      if (__lane_id() == 0 && su) atomicAdd(&P.stats[kStatSuspended], (unsigned long long)su);
      if (__lane_id() == 0 && sa) atomicAdd(&P.stats[kStatSuspendedShadow], (unsigned long long)sa);
    }
    if (__lane_id() == 0) atomicMax(&P.stats[kStatMaxWaveIters], (unsigned long long)(st.it_node + st.it_tri + st.it_curve + st.it_refill));
    if (threadIdx.x == 0 && blockIdx.x == 0) {
      atomicAdd(&P.stats[kStatClosestRays], (unsigned long long)n_closest);
      atomicAdd(&P.stats[kStatShadowRays], (unsigned long long)n_shadow);
    }
}

// FIRST, scenes with curves (and the statistics instances): one block per CU fewer -- the camera sample (64-bit multiplies of the generator, a
// square root, two divisions) lives in the refill path and would spill into the traversal loop's register budget.  Triangle-only
// scenes: the same blocks per CU as the other launches since round 5 (79 VGPRs, nothing spilled: first launch of C2 5.66 -> 5.50 ms)
constexpr uint32_t trace_first_less(bool stats, bool curves) { return (stats || curves) ? 1u : 0u; }
template <bool STATS, bool CURVES, bool WIDE = false, bool FIRST = false>
__global__ __launch_bounds__(kBlock, trace_blocks_per_cu(CURVES, WIDE) - (FIRST ? trace_first_less(STATS, CURVES) : 0u)) void k_trace(PathState P, DScene sc) {
  __shared__ uint32_t stk[pv_lds_stack<CURVES, WIDE>() * kBlock];
  __shared__ float frm[CURVES ? 10 * kBlock : 1];
  // the top of the tree in LDS (triangle-only scenes: with the ribbon frames of curve scenes it would cost a block per CU)
  constexpr int kTopHere = WIDE ? kTopNodesWide : kTopNodes;
  constexpr bool kStageTop = !CURVES && kTopHere > 0;
  __shared__ float4 top[kStageTop ? kTopHere * 4 : 1];
  const uint32_t ntop = kStageTop ? (WIDE ? (sc.wide_top_nodes < (uint32_t)kTopHere ? sc.wide_top_nodes : (uint32_t)kTopHere) : sc.top_nodes) : 0u;
  if (kStageTop) {
    const float4* src = WIDE ? sc.wide : reinterpret_cast<const float4*>(sc.nodes);  // (binary node and Q node: 64 bytes each)
    for (uint32_t i = threadIdx.x; i < ntop * 4u; i += kBlock) top[i] = src[i];
    __syncthreads();
  }
  const uint32_t n_closest = P.counts[kCntIn], n_shadow = P.counts[kCntShadowIn];
  TravStats st = {};
  uint32_t overflow = 0u;
  TraceSinkT<CURVES, FIRST> sink = {P, n_closest, n_shadow};
  const unsigned long long t_start = P.wave_log ? wall_clock64() : 0ull;
  uint32_t wave_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * kBlock + threadIdx.x) >> 6));
  asm volatile("" : "+s"(wave_idx));  // (a scalar, computed before the loop: threadIdx.x does not stay alive across it)
  trace_pv<FIRST ? 0 : 2, STATS, CURVES, WIDE, kTraceHeads>(sc, n_closest + (FIRST ? 0u : n_shadow), P.heads, sink, stk + threadIdx.x, kBlock,
                             P.spill + blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock, st, &overflow,
                             CURVES ? frm + threadIdx.x : nullptr, top, ntop);
  if (overflow) P.counts[kCntOverflow] = 1u;
  if (P.wave_log && __lane_id() == 0 && P.wave_log_launch < kWaveLogLaunches) {
    const uint32_t w = wave_idx;
    if (w < kWaveLogWaves) {
      unsigned long long* o = P.wave_log + ((size_t)P.wave_log_launch * kWaveLogWaves + w) * 4;
      o[0] = t_start, o[1] = wall_clock64(), o[2] = STATS ? (st.it_refill | ((unsigned long long)st.refill_ticks << 32)) : st.t_exhausted, o[3] = st.it_node + st.it_tri + st.it_curve;
    }
  }
  if (STATS) trace_stats_out(P, st, n_closest, n_shadow);
}

// Small launches (the late iterations of a group, every launch of a small render) on one ray per QUAD of lanes (dtrace_quad.h).
// Rays are dealt to the quads in order (no queue: at most a few rays per quad).  Triangle-only Q trees.  Off by default
// (PBRHIP_QUAD_RAYS = the largest launch it takes): measured equal to k_trace on the benchmark frames, profiles/README.md.
template <bool STATS>
__global__ __launch_bounds__(kBlock) void k_trace_quad(PathState P, DScene sc) {
  __shared__ uint32_t stk[kSimpleLdsStack * (kBlock / 4)];
  const uint32_t n_closest = P.counts[kCntIn], n_shadow = P.counts[kCntShadowIn];
  const uint32_t n = n_closest + n_shadow;
  constexpr uint32_t quads = kBlock / 4;
  const uint32_t quad = threadIdx.x >> 2;
  uint32_t overflow = 0u;
  TraceSinkT<false> sink = {P, n_closest, n_shadow};
  for (uint32_t i = blockIdx.x * quads + quad; i < n; i += gridDim.x * quads) {
    uint32_t tag;
    V3 o, d;
    float tmin, tmax;
    const bool any = sink.load(i, tag, o, d, tmin, tmax);
    tag &= ~kTagResume;  // (a ray another launch suspended is traced from its start here: the same hit)
    Hit h;
    const bool occluded = traverse_quad<2>(sc, o, d, tmin, tmax, h, stk + quad, quads, &overflow, any, P.spill + blockIdx.x * quads + quad, gridDim.x * quads);
    if ((threadIdx.x & 3u) == 0u) sink.done(tag, h, occluded);
  }
  if (overflow) P.counts[kCntOverflow] = 1u;
  if (STATS) {
    TravStats st = {};
    trace_stats_out(P, st, n_closest, n_shadow);
  }
}

// ------------------------------------------------------------------ ordered stream compaction, few atomics
// One global atomicAdd per output queue per TILE of kTileItems entries (one queue word saturates at ~88
// atomics/us on this chip: a per-wave atomicAdd per append cost ~45 ms per kernel at 289 M entries).
// A block reads kItemsPerThread strided entries per thread, ranks them with wave ballots, reduces the
// per-(item row, wave) counts in LDS, reserves the tile's output range once, and scatters in input order.
// Tile sizes: k_classify gathers (its loads depend on the queue entry), more items per thread only cost it occupancy; k_compact
// streams, and at 8 items per thread its two counters were the bottleneck (51 k atomics per launch on one word: 0.84 ms for
// 105 M entries; 16 items -> 0.45, 32 -> 0.3).
#ifndef PB_CLASSIFY_ITEMS
#define PB_CLASSIFY_ITEMS 8
#endif
#ifndef PB_COMPACT_ITEMS
#define PB_COMPACT_ITEMS 32
#endif
#ifndef PB_CLASSIFY_RUNS
#define PB_CLASSIFY_RUNS 2  // the principled hits of a tile leave it in two runs (kHitMore first); 1: in queue order
#endif
constexpr int kClassifyItems = PB_CLASSIFY_ITEMS, kCompactItems = PB_COMPACT_ITEMS;
constexpr int kWavesPerBlock = kBlock / 64;

template <int NQ, int kItemsPerThread>
struct TileCompactor {
  uint32_t (*wcount)[kItemsPerThread][kWavesPerBlock];  // [NQ] in LDS
  uint32_t* base;                                        // [NQ] in LDS
  uint32_t rank[kItemsPerThread];
  // dest[j] in 0..NQ (0 = drop).  Call from all threads of the block.
  __device__ __forceinline__ void run(const uint32_t dest[kItemsPerThread], uint32_t* const counters[NQ]) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < kItemsPerThread; j++) {
      rank[j] = 0;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        unsigned long long m = __ballot(dest[j] == (uint32_t)(q + 1));
        if (lane == 0) wcount[q][j][wave] = (uint32_t)__popcll(m);
        if (dest[j] == (uint32_t)(q + 1)) rank[j] = (uint32_t)__popcll(m & lt);
      }
    }
    __syncthreads();
    // exclusive prefix over the (row, wave) counts in input order: wave q scans queue q's kItemsPerThread x kWavesPerBlock
    // counters, a few per lane, with a shuffle scan across the lanes (a one-thread loop over 128 counters was a quarter of
    // k_compact's time)
    static_assert(NQ <= kWavesPerBlock, "one wave per queue");
    constexpr int M = kItemsPerThread * kWavesPerBlock, C = (M + 63) / 64;
    if (M <= 32) {  // few counters: one thread per queue walks them
      if (threadIdx.x < NQ) {
        const int q = threadIdx.x;
        uint32_t total = 0;
        for (int j = 0; j < kItemsPerThread; j++)
          for (int w = 0; w < kWavesPerBlock; w++) {
            uint32_t c = wcount[q][j][w];
            wcount[q][j][w] = total;
            total += c;
          }
        base[q] = total ? atomicAdd(counters[q], total) : 0u;
      }
    } else if (wave < (uint32_t)NQ) {
      uint32_t* cnt = &wcount[wave][0][0];
      uint32_t v[C], sum = 0;
#pragma unroll
      for (int k = 0; k < C; k++) {
        const int at = (int)lane * C + k;
        v[k] = at < M ? cnt[at] : 0u;
        sum += v[k];
      }
      uint32_t incl = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= (uint32_t)d) incl += y;
      }
      uint32_t run = incl - sum;
#pragma unroll
      for (int k = 0; k < C; k++) {
        const int at = (int)lane * C + k;
        if (at < M) cnt[at] = run;
        run += v[k];
      }
      const uint32_t total = (uint32_t)__shfl((int)incl, 63);
      uint32_t* ctr = counters[0];  // (a select, not an indexed read: the pointer array stays in registers)
#pragma unroll
      for (int q = 1; q < NQ; q++)
        if (wave == (uint32_t)q) ctr = counters[q];
      if (lane == 0) base[wave] = total ? atomicAdd(ctr, total) : 0u;
    }
    __syncthreads();
  }
  __device__ __forceinline__ uint32_t slot(int j, uint32_t dest_j) const {
    return base[dest_j - 1] + wcount[dest_j - 1][j][threadIdx.x >> 6] + rank[j];
  }
};

// ------------------------------------------------------------------ k_classify
// Queue entries are path slots; bit 31 marks a path that is inside a medium (random-walk SSS).
// Routes every traced path: in-medium -> q_sss; miss -> dropped (render.cc:34, no environment light);
// hit -> q_hair / q_principled by the material kind denormalised into ShadeRec.flags.
// A hit path whose next Russian roulette (render.cc:66-68) is known to fail (kQDoomed: the shading that continued the
// path already knew the throughput and the generator state the roulette will use) and whose hit primitive is no light
// ends at the head of its shading without storing anything (path_head returns false), so it is dropped here instead of
// idling in a shading wave; likewise a hit on a primitive without material.  The routing bits come with the hit code.
// The principled hits of a tile (2048 entries) leave it in two runs -- first the hits whose shading fetches the long part of the
// ShadeRec (kHitMore: corner normals / texcoords: the smooth meshes), then the hits on flat triangles -- so that a shading wave is
// mostly one kind or the other: the same queue, the same lines touched per tile, fewer waves that run both sides of every branch
// (north star: "per-closure material sorting", here by what the hit code already says; separate queues lose: profiles/README.md).
__global__ __launch_bounds__(kBlock) void k_classify(PathState P, DScene sc) {
  constexpr int kItemsPerThread = kClassifyItems, kTileItems = kItemsPerThread * kBlock;
  __shared__ uint32_t wcount[4][kItemsPerThread][kWavesPerBlock];
  __shared__ uint32_t base[4];
  __shared__ uint32_t tile_principled[2];  // this tile's two runs: counted in LDS, reserved in the queue with ONE atomic
  const uint32_t n = P.counts[kCntIn];
  const uint32_t ntiles = (n + kTileItems - 1) / kTileItems;
  uint32_t* const counters[4] = {&P.counts[kCntSss], &tile_principled[0], &P.counts[kCntHair], &tile_principled[1]};
  uint32_t* const queues[4] = {P.q_sss, P.q_principled, P.q_hair, P.q_principled};
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    uint32_t p[kItemsPerThread], dest[kItemsPerThread];  // p: path slot | kQFirst (handed on to the shading queues)
    bool doomed[kItemsPerThread];
#pragma unroll
    for (int j = 0; j < kItemsPerThread; j++) {
      uint32_t i = tile * kTileItems + j * kBlock + threadIdx.x;
      dest[j] = 0, p[j] = 0, doomed[j] = false;
      if (i < n) {
        const uint32_t e = P.first ? ((P.slot0 + i) | kQFirst) : P.q_in[i];  // (a group's first bounce: entry i is path slot0 + i)
        p[j] = e & (kQPathMask | kQFirst | kQDoomed);  // (kQDoomed rides along: a held path hands it back, kernels.h::kRHold)
        dest[j] = (!P.no_medium && (e & kQSssBit)) ? 1u : 0xFFu;  // (without media bit 31 is kQHold: the path is routed by its hit like any other)
        doomed[j] = (e & kQDoomed) != 0u;
      }
    }
#pragma unroll
    for (int j = 0; j < kItemsPerThread; j++)
      if (dest[j] == 0xFFu) {
        const uint32_t code = __float_as_uint(P.hit[p[j] & kQPathMask].w);
        dest[j] = (code & kHitHair) ? 3u : ((PB_CLASSIFY_RUNS == 2 && (code & kHitMore)) ? 2u : 4u);
        if (code == kNone || (!(code & kHitLight) && (doomed[j] || (code & kHitNoMaterial)))) dest[j] = 0u;
        // a path whose ray was suspended (its ray goes on in the next launch) rides through the principled queue untouched: the shading
        // kernel turns its entry into a "resume" result word and k_compact re-queues it -- no atomic, no queue of its own
        if (code == kHitSuspended) dest[j] = 4u, p[j] |= kQResume;
      }
    if (threadIdx.x < 2) tile_principled[threadIdx.x] = 0u;
    __syncthreads();
    TileCompactor<4, kItemsPerThread> tc = {wcount, base, {}};
    tc.run(dest, counters);  // (base[1] = base[3] = 0: the two runs were counted from 0)
    if (threadIdx.x == 0) {
      const uint32_t a = tile_principled[0], b = tile_principled[1];
      const uint32_t g = (a + b) ? atomicAdd(&P.counts[kCntPrincipled], a + b) : 0u;
      base[1] = g, base[3] = g + a;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kItemsPerThread; j++)
      if (dest[j]) queues[dest[j] - 1][tc.slot(j, dest[j])] = p[j];
    __syncthreads();
  }
}

// ------------------------------------------------------------------ k_compact
// The shade kernels overwrite their queue entry with  path | kRShadow | kRAlive | kQSssBit | kQDoomed  instead of
// appending; this pass turns the three result lists into the next trace queue and the shadow-ray queue.
__global__ __launch_bounds__(kBlock) void k_compact(PathState P) {
  constexpr int kItemsPerThread = kCompactItems, kTileItems = kItemsPerThread * kBlock;
  __shared__ uint32_t wcount[2][kItemsPerThread][kWavesPerBlock];
  __shared__ uint32_t base[2];
  const uint32_t n0 = P.counts[P.direct ? kCntIn : kCntPrincipled], n1 = P.counts[kCntHair], n2 = P.counts[kCntSss];
  const uint32_t n = n0 + n1 + n2;
  const uint32_t ntiles = (n + kTileItems - 1) / kTileItems;
  uint32_t* const counters[2] = {&P.counts[kCntOut], &P.counts[kCntShadow]};
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    uint32_t e[kItemsPerThread];
#pragma unroll
    for (int j = 0; j < kItemsPerThread; j++) {
      uint32_t i = tile * kTileItems + j * kBlock + threadIdx.x;
      e[j] = 0;
      if (i < n) e[j] = (i < n0) ? P.q_principled[i] : ((i < n0 + n1) ? P.q_hair[i - n0] : P.q_sss[i - n0 - n1]);
    }
    // a "resume" word (kRResume without kRAlive: the path's ray was suspended, kernels.h) goes back to the trace queue with kQResume
    // (everything below is a function of e[j]: nothing else is kept per item)
    auto resumes = [](uint32_t w) { return (w & (kRAlive | kRResume)) == kRResume; };
    // ... and a "hold" word (kRHold alone: the path's shadow ray is suspended) with kQHold: its closest-hit ray is not traced again
    auto holds = [](uint32_t w) { return (w & (kRAlive | kRResume | kRHold)) == kRHold; };
    // two independent streams share one pass: run the compactor once per stream
    TileCompactor<2, kItemsPerThread> ta = {wcount, base, {}};
    {
      uint32_t dest[kItemsPerThread];
#pragma unroll
      for (int j = 0; j < kItemsPerThread; j++) dest[j] = ((e[j] & kRAlive) || resumes(e[j]) || holds(e[j])) ? 1u : 0u;
      ta.run(dest, counters);
#pragma unroll
      for (int j = 0; j < kItemsPerThread; j++)
        if (dest[j])
          P.q_out[ta.slot(j, 1u)] = resumes(e[j]) ? ((e[j] & (kRPathMask | kQDoomed)) | kQResume | ((e[j] & kRResumeFirst) ? kQFirst : 0u))
                                    : (holds(e[j]) ? ((e[j] & kRPathMask) | kQHold | ((e[j] & kRHoldDoomed) ? kQDoomed : 0u))
                                                   : (e[j] & (kRPathMask | kQSssBit | kQDoomed)));
      __syncthreads();
    }
    {
      uint32_t dest[kItemsPerThread];
#pragma unroll
      for (int j = 0; j < kItemsPerThread; j++) dest[j] = ((e[j] & kRShadow) && !resumes(e[j]) && !holds(e[j])) ? 2u : 0u;
      ta.run(dest, counters);
#pragma unroll
      for (int j = 0; j < kItemsPerThread; j++)
        if (dest[j]) P.q_shadow[ta.slot(j, 2u)] = e[j] & kRPathMask;
      __syncthreads();
    }
  }
}

// Head of GetRadiance for one path that hit something (render.cc:39-68): surface, implicit area light with
// MIS, Russian roulette.  Returns false when the path ends here.
struct PathHead {
  Hit h;
  V3 dir, thr;
  Surface s;
  Rng rng;
  uint32_t flags;
};
// Returns kHeadEnds when the path ends here, kHeadHeld when its shadow ray of the previous bounce is still suspended (PathState::hold:
// nothing is touched, the path waits one more iteration), else kHeadGoes.
enum : int { kHeadEnds = 0, kHeadGoes = 1, kHeadHeld = 2 };
// HOLD: the kernel can meet held paths (scenes without media only: compiled out of the kernel of scenes with media, kShadeMedia, which sits at the
// edge of its register class; the kShadeFull instances also serve scenes without media -- the statistics build of k_tail -- and keep the hold word
// initialised: a hair shading of the same path reads it)
template <bool HOLD = true>
__device__ __forceinline__ int path_head(const PathState& P, const DScene& sc, uint32_t p, uint64_t rng_inc, PathHead& c, bool first) {
  float4 h4 = P.hit[p];
  float4 o4 = make_float4(P.cam_org[0], P.cam_org[1], P.cam_org[2], 0.0f), t4 = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
  c.flags = first ? 0u : kFlagNotFirst;  // set by every head after the first (render.cc:43-61: depth-0 emission has weight 1)
  uint64_t rng_state;
  bool held = false;
  if (!first) {
    o4 = P.ray_o[p], t4 = P.thr[p];
    c.dir = ld3(P.ray_d[p]);
    if (HOLD) {
      const uint4 r4 = P.rng4[p];  // generator state | hold | - : one 16-byte word of the path's record
      rng_state = (uint64_t)r4.x | ((uint64_t)r4.y << 32);
      held = P.no_medium != 0u && r4.z != 0u;
    } else {
      rng_state = P.rng[p];
    }
  } else {
    camera_sample(P, p - P.slot0, c.dir, rng_state);  // the camera ray's values are implied or recomputed (PathState::first)
  }
  c.h.t = h4.x, c.h.u = h4.y, c.h.v = h4.z, c.h.slot = __float_as_uint(h4.w);
  c.s = make_surface(sc, ld3(o4), c.dir, c.h);
  c.thr = ld3(t4);
  if (!held && c.s.face == kFront && c.s.lightrec != kNone) {  // render.cc:43-62, LightManager::ImplicitAreaLight
    const float4* lr = reinterpret_cast<const float4*>(sc.lrecs + c.s.lightrec);
    float pdf_area = lr[0].w;
    V3 emission = ld3(lr[4]);
    float a2s = fabsf((c.h.t * c.h.t) / dot(c.s.n_s, c.dir));
    float w = (c.flags & kFlagNotFirst) ? power_heuristic(t4.w, pdf_area * a2s) : 1.0f;
    float4 L4 = P.L[p];
    P.L[p] = mk4(ld3(L4) + w * emission * c.thr, L4.w);
  }
  c.rng.state = rng_state, c.rng.inc = rng_inc;
  float rr = spectrum_norm(c.thr);  // render.cc:66-68 (Q1)
  float u = draw(c.rng);
  if (held) return kHeadHeld;  // (nothing was stored: the path is shaded when its shadow ray has been delivered)
  if (rr < u) return kHeadEnds;
  c.thr = c.thr * V3(1.0f / rr);
  c.flags |= kFlagNotFirst;
  return (c.s.flags & kSlotMatNone) == 0 ? kHeadGoes : kHeadEnds;  // shader.cc:11-17: no material -> throughput 0 -> path ends
}
// a continuing path's generator state, with its hold flag cleared (the 16-byte word of its record: kernels.h::PathState)
template <bool HOLD = true>
__device__ __forceinline__ void store_rng(const PathState& P, uint32_t p, uint64_t state) {
  if (HOLD) P.rng4[p] = make_uint4((uint32_t)state, (uint32_t)(state >> 32), 0u, 0u);
  else P.rng[p] = state;
}

// kQDoomed for a path that continues with throughput `thr` and generator state `state`: the head of its next shading
// draws once from that state and ends the path when max(thr) < u (path_head; render.cc:66-68)
__device__ __forceinline__ uint32_t doomed_bit(V3 thr, uint64_t state, uint64_t rng_inc) {
  Rng r;
  r.state = state, r.inc = rng_inc;
  return spectrum_norm(thr) < draw(r) ? kQDoomed : 0u;
}

// A doomed path can still pick up the emission of an area light its next ray hits (path_head adds it before the roulette);
// nothing else it does survives.  With a handful of light primitives that is decided here: if the ray misses every one
// of them (the very test, on the very operands, the traversal would run on those primitives) the path ends now and its ray
// is never traced.  Scenes with more light primitives than kLightPretest but at most kLightPretest lights (emissive meshes)
// get the conservative version of the same argument: a ray that misses the bounding box of every light -- the traversal's
// own slab test on boxes stored like BVH boxes (sc.light_boxes, two per node) -- cannot hit a light primitive either.
// Beyond that the ray is kept (a linear scan would not pay).
constexpr uint32_t kLightPretest = 8;
__device__ __forceinline__ bool misses_all_lights(const DScene& sc, V3 o, V3 d, float tmin) {
  if (sc.lights_transformed) return false;  // the light records hold local positions, the raytracer tests transformed ones
  if (sc.num_lrecs <= kLightPretest) {
    const V3 inv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    // two light primitives per test on packed fp32 (dtrace.h::tri_test_pair: per triangle tri_test's operations and bits); an odd
    // last one is tested twice
    for (uint32_t i = 0; i < sc.num_lrecs; i += 2u) {
      const float4* la = reinterpret_cast<const float4*>(sc.lrecs + i);
      const float4* lb = reinterpret_cast<const float4*>(sc.lrecs + (i + 1u < sc.num_lrecs ? i + 1u : i));
      const float4 a0 = la[0], a1 = la[1], a2 = la[2], b0 = lb[0], b1 = lb[1], b2 = lb[2];
      bool ok_a, ok_b;
      f2 t, u, v;
      tri_test_pair(make_float4(a0.x, b0.x, a0.y, b0.y), make_float4(a0.z, b0.z, a1.x, b1.x), make_float4(a1.y, b1.y, a1.z, b1.z),
                    make_float4(a2.x, b2.x, a2.y, b2.y), make_float4(a2.z, b2.z, 0.f, 0.f), o.x, o.y, o.z, d.x, d.y, d.z, inv.x, inv.y, inv.z, tmin, ok_a,
                    ok_b, t, u, v);
      if (ok_a || ok_b) return false;
    }
    return true;
  }
  if (sc.num_lights > kLightPretest) return false;
  const V3 inv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  for (uint32_t i = 0; i < (sc.num_lights + 1u) / 2u; i++) {
    const float4* np = reinterpret_cast<const float4*>(sc.light_boxes + i);
    bool h0, h1;
    float t0, t1;
    box_test2(np[0], np[1], np[2], o, inv, tmin, kInf, h0, h1, t0, t1);
    if (h0 || h1) return false;
  }
  return true;
}

__device__ __forceinline__ void count_pruned(const PathState& P) {  // P.stats is null unless the render collects statistics
  if (P.stats) atomicAdd(&P.stats[kStatPrunedRays], 1ull);
}

// writes one shadow-queue entry
__device__ __forceinline__ void put_shadow(const PathState& P, V3 pos, const Nee& n, V3 c_vis, V3 c_occ, uint32_t p,
                                           uint32_t mode, bool alive) {
  // ShadowRay (shader-utils.h:116-129): [kEps, max(kEps, dist - kEps)]  (Q9).  One shadow ray per path per
  // iteration at most, so the payload lives at the path's own slot.  Its origin and tmin are those of the path's next ray
  // (the same surface point, the same 1e-3 offset), so they are read from ray_o; a path that ends here still stores them.
  static_assert(kEps == 1e-3f, "shadow rays share (origin, tmin) with the continuation ray");
  if (!alive) P.ray_o[p] = mk4(pos, kEps);
  P.sh_d[p] = mk4(n.dir, smax(kEps, n.dist - kEps));
  P.sh_c[p] = mk4(c_vis, __uint_as_float(mode));
  if (mode == kShSssExit) P.sh_e[p] = mk4(c_occ, 0.f);  // what to add when the ray is occluded (only a medium exit has one)
}

// Block-cooperative copy of the light tables into LDS (LdsLightTables layout, dshade.h) when they are small enough; the
// caller synchronises.  The shading kernels sample a light per hit: five 16-byte reads of a record, two searches in
// cumulative tables -- from LDS they do not queue behind the path-state traffic of the vector-memory path.
__device__ __forceinline__ bool stage_light_tables(const DScene& sc, float* lds) {
  if (!(sc.num_lights > 0 && sc.num_lights <= kLdsLights && sc.num_lrecs <= kLdsLights)) return false;
  constexpr uint32_t kRecWords = sizeof(LightRec) / 4;
  for (uint32_t i = threadIdx.x; i < sc.num_lights; i += kBlock) {
    lds[i] = sc.light_cdf[i];
    lds[kLdsLights + 2u * i] = __uint_as_float(sc.light_heads[i].first);
    lds[kLdsLights + 2u * i + 1u] = __uint_as_float(sc.light_heads[i].count);
  }
  for (uint32_t i = threadIdx.x; i < sc.num_lrecs; i += kBlock) lds[3u * kLdsLights + i] = sc.lprim_cdf[i];
  for (uint32_t i = threadIdx.x; i < sc.num_lrecs * kRecWords; i += kBlock) lds[4u * kLdsLights + i] = reinterpret_cast<const float*>(sc.lrecs)[i];
  return true;
}

// ------------------------------------------------------------------ k_shade_principled
// CyclesPrincipledShader (cycles-principled-shader.cc:414-484) + the tail of GetRadiance (render.cc:76-87).
// One path; returns the result bits (kRShadow | kRAlive | kQSssBit | kQDoomed) its caller stores or acts on.
// lds_bsdf: the scene's closure sets staged in LDS by the caller (k_shade_principled when they fit), or null
// MODE (what the scene's materials can do, so that code no hit can reach -- and the registers it holds -- is compiled out of the
// wavefront kernel): kShadePlain: no medium, no texture; kShadeMedia: media (random-walk subsurface), no texture; kShadeFull.
enum : int { kShadePlain = 0, kShadeMedia = 1, kShadeFull = 2 };
template <int MODE = kShadeFull>
__device__ __forceinline__ uint32_t shade_principled_path(const PathState& P, const DScene& sc, uint32_t p, uint64_t rng_inc, bool first,
                                                          const PrincipledBsdf* lds_bsdf = nullptr, const float* lds_lights = nullptr) {
  {
    const bool active = true;
    bool alive = false, shadow = false;
    V3 sh_pos(0.f), c_vis(0.f);
    Nee nee;
    nee.dir = V3(0.f), nee.emission = V3(0.f), nee.dist = 0.f, nee.pdf_sigma = 0.f;
    uint32_t sh_mode = kShNormal, qbit = 0u;
    PathHead c;
    const int head = path_head<MODE != kShadeMedia>(P, sc, p, rng_inc, c, first);
    if (active && head == kHeadGoes) {
      const Hit& h = c.h;
      const V3 dir = c.dir, thr = c.thr;
      const Surface& s = c.s;
      Rng& rng = c.rng;
      V3 new_thr(0.f), next_dir = -dir;
      float new_pdf = 0.f;
      if (s.face != kAmbiguous) {  // :418-424
        V3 wo_g = -dir;
        Frame fr;
        fr.ez = (s.face == kFront) ? s.n_s : -s.n_s;
        branchless_onb(fr.ez, fr.ex, fr.ey);
        V3 wo = to_local(fr, wo_g);
        PrincipledBsdf b;
        if (lds_bsdf) {  // (LDS reads: off the vector-memory path the kernel is bound by)
          constexpr uint32_t kWords = sizeof(PrincipledBsdf) / 4;
          const auto* lw = (const __attribute__((address_space(3))) uint32_t*)reinterpret_cast<const uint32_t*>(lds_bsdf) + s.material * kWords;
          uint32_t w[kWords];
#pragma unroll
          for (uint32_t k = 0; k < kWords; k++) w[k] = lw[k];
          __builtin_memcpy(&b, w, sizeof(b));
        } else {
          b = sc.materials[s.material].bsdf;
        }
        const bool per_hit = MODE == kShadeFull && sc.materials[s.material].textured != 0u;
        if (per_hit) {  // ParamToBsdf per hit (cycles-principled-shader.cc:281-301)
          const PrincipledParam mp = sc.materials[s.material].param;
          V3 bc(mp.base_color[0], mp.base_color[1], mp.base_color[2]);
          V3 ssc(mp.subsurface_color[0], mp.subsurface_color[1], mp.subsurface_color[2]);
          if (mp.base_color_tex_id != kNone) bc = texture_fetch3(sc, mp.base_color_tex_id, s.tu, s.tv);
          if (mp.subsurface_color_tex_id != kNone) ssc = texture_fetch3(sc, mp.subsurface_color_tex_id, s.tu, s.tv);
          b = param_to_bsdf(mp, bc, ssc);
        }
        SampleWeight w = closure_sample_weight(wo, b);
        // DirectIllumination (shader-utils.h:166-212)
        V3 d1(0.f);
        shadow = nee_sample(sc, rng, s.pos, fr.ez, true, nee, lds_lights);
        if (shadow) {
          V3 f;
          float pdf;
          eval_bsdf(to_local(fr, nee.dir), wo, b, w, f, pdf);
          d1 = nee_contribution(nee, f, pdf);
          sh_pos = s.pos;
        }
        // SampleBsdf (:169-242)
        float select = draw(rng);
        int pick = pick_closure(select, w);
        V3 wi(0.f);
        bool sampled = true;
        if (pick == 0) {
          float u0 = draw(rng);
          float u1 = draw(rng);
          float pdf;
          lambert_sample(u0, u1, wi, pdf);
        } else if (pick == 2) {
          float u0 = draw(rng);
          float u1 = draw(rng);
          ggx_sample(wo, b.alpha_x, b.alpha_y, u0, u1, wi);
        } else if (pick == 3) {
          float u0 = draw(rng);
          float u1 = draw(rng);
          ggx_sample(wo, b.clearcoat_alpha_x, b.clearcoat_alpha_y, u0, u1, wi);
        } else if (MODE == kShadePlain) {
          sampled = false;  // unreachable: no material has a subsurface weight
        } else {
          // RandomWalkSubsurface entry (random-walk-sss.h:236-287)
          sampled = false;
          bool ok = (s.face == kFront);
          if (ok) {
            float u0 = draw(rng);
            float u1 = draw(rng);
            V3 tmp;
            float pdf;
            lambert_sample(u0, u1, tmp, pdf);
            tmp = -tmp;
            V3 gdir = to_global(fr, tmp);
            ok = !(dot(-s.n_g, gdir) <= 0.0f);
            if (ok) {
              // the medium's coefficients and the walk's first throughput (random-walk-sss.h:111-122, 243-258) depend on the
              // closure set alone: for a material without textures they were computed at commit with these very functions
              // (host and device share the f64r exp), otherwise they follow from this hit's closure set
              V3 sigt, sigs, wthr;
              if (per_hit) {
                medium_coefficients(b, sigt, sigs, wthr);
              } else {
                const float4* mc = reinterpret_cast<const float4*>(&sc.materials[s.material].sss_sigt);
                const float4 m0 = mc[0], m1 = mc[1], m2 = mc[2];
                sigt = V3(m0.x, m0.y, m0.z), sigs = V3(m0.w, m1.x, m1.y), wthr = V3(m1.z, m1.w, m2.x);
              }
              float e0 = draw(rng);
              float e1 = draw(rng);
              V3 chpdf;
              float t_scatter = sample_scatter_distance(wthr, sigs, sigt, e0, e1, chpdf);
              P.ray_o[p] = mk4(s.pos, 1e-3f);
              P.ray_d[p] = mk4(gdir, t_scatter);
              P.sss_sigt[p] = mk4(sigt, 0.f);
              P.sss_sigs[p] = mk4(sigs, __uint_as_float(sc.shade[h.slot & kHitSlotMask].instance_id));  // .w = entry instance id
              P.sss_thr[p] = mk4(wthr, __uint_as_float(0u));  // .w = step index (kept with the data every step rewrites)
              P.sss_ez[p] = mk4(fr.ez, 0.f);
              P.sss_A[p] = make_float4(0.f, 0.f, 0.f, 0.f);
              P.thr[p] = mk4(thr, 0.f);  // Russian-roulette-scaled path throughput, used again at the exit
              P.rng[p] = rng.state;
              alive = true;
              qbit = kQSssBit;
              sh_mode = kShSssEntry;
              c_vis = d1;  // raw: resolved into A when the shadow ray is traced (TraceSink::done)
            }
          }
          // failed entry: omega_in = f = pdf = 0 -> 0*0/0 = NaN -> throughput 0 (:217-220, :474-483)
        }
        if (sampled) {
          V3 f;
          float pdf;
          eval_bsdf(wi, wo, b, w, f, pdf);
          next_dir = to_global(fr, wi);
          float cos_i = fabsf(wi.z);
          new_thr = f * cos_i / pdf;
          new_pdf = pdf;
          if (!is_finite(new_thr) || !isfinite(new_pdf)) {
            new_thr = V3(0.f);
            new_pdf = 0.f;
          }
        }
        if (sh_mode == kShNormal) c_vis = thr * ((V3(0.f) + d1) + V3(0.f));  // render.cc:79
        if (!alive) {
          // render.cc:80-86
          V3 t2 = new_thr * thr;
          if (!is_black(t2)) {
            qbit = doomed_bit(t2, rng.state, rng_inc);
            if (!(qbit && misses_all_lights(sc, s.pos, next_dir, 1e-3f))) {
              alive = true;
              P.ray_o[p] = mk4(s.pos, 1e-3f);
              P.ray_d[p] = mk4(next_dir, kInf);
              P.thr[p] = mk4(t2, new_pdf);
              store_rng<MODE != kShadeMedia>(P, p, rng.state);
            } else {
              qbit = 0u;
              count_pruned(P);
            }
          }
        }
      }
    }
    if (shadow) put_shadow(P, sh_pos, nee, c_vis, V3(0.f), p, sh_mode, alive);
    return head == kHeadHeld ? kRHold : ((shadow ? kRShadow : 0u) | (alive ? kRAlive : 0u) | qbit);
  }
}
#ifndef PB_SHADE_WAVES
#define PB_SHADE_WAVES 4  // min waves per SIMD of the plain kernel: <= 128 VGPRs (round 5: 117-119 by itself since kernels.hip is compiled without the SLP vectoriser; five waves spill: 10.5 -> 12.7 ms on C2.  Rounds 2-4, 133-168 VGPRs: 3)
#endif
#ifndef PB_LDS_MATS
#define PB_LDS_MATS 64
#endif
constexpr uint32_t kLdsMats = PB_LDS_MATS;  // closure sets staged in LDS by the plain shading kernel (96 B each)
#ifndef PB_SHADE_WAVES_FULL
#define PB_SHADE_WAVES_FULL 3  // the same for the kernel of textured materials
#endif
#ifndef PB_SHADE_WAVES_MEDIA
#define PB_SHADE_WAVES_MEDIA 4  // ... of scenes with media and no textures (C3, C5): 128-130 VGPRs by itself, pinned to the class it sits at the edge of
#endif
template <int MODE>
__global__ __launch_bounds__(kBlock, MODE == kShadePlain ? PB_SHADE_WAVES : (MODE == kShadeMedia ? PB_SHADE_WAVES_MEDIA : PB_SHADE_WAVES_FULL)) void k_shade_principled(PathState P, DScene sc, uint64_t rng_inc) {
  __shared__ PrincipledBsdf lds_bsdf[kLdsMats ? kLdsMats : 1];
  __shared__ float lds_lights[kLdsMats ? kLdsLightWords : 1];
  const bool lights_staged = kLdsMats && stage_light_tables(sc, lds_lights);
  const bool staged = kLdsMats && sc.num_materials <= kLdsMats;
  if (staged) {
    constexpr uint32_t kWords = sizeof(PrincipledBsdf) / 4;
    static_assert(sizeof(PrincipledBsdf) % 4 == 0, "closure set is a whole number of words");
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds_bsdf);
    for (uint32_t i = threadIdx.x; i < sc.num_materials * kWords; i += kBlock)
      dst[i] = reinterpret_cast<const uint32_t*>(&sc.materials[i / kWords].bsdf)[i % kWords];
  }
  if (staged || lights_staged) __syncthreads();
  // P.direct: this bounce's rays were not classified (kernels.h) -- the entries are the trace queue's (a first bounce: entry i is path
  // slot0 + i) and k_classify's drop rule is applied here
  const bool direct = P.direct != 0u;
  const uint32_t n = P.counts[direct ? kCntIn : kCntPrincipled];
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    // (scenes with media are direct on a first bounce only: their kernel, at the edge of its register class, keeps the shorter code)
    constexpr bool kAnyBounce = MODE != kShadeMedia;
    const uint32_t e = direct ? ((!kAnyBounce || P.first) ? P.slot0 + i : P.q_in[i]) : P.q_principled[i];
    const uint32_t p = e & kQPathMask;
    uint32_t r = 0u;
    bool go = true;
    // a path whose closest-hit ray k_trace suspended is not shaded now: its result word says "resume" (kernels.h: kRResume; the
    // path's kQDoomed / kQFirst ride along) and k_compact puts it back into the trace queue
    if (direct) {
      const uint32_t code = __float_as_uint(P.hit[p].w);
      const bool doomed = kAnyBounce && (e & kQDoomed);
      go = !(code == kNone || (!(code & kHitLight) && (doomed || (code & kHitNoMaterial))));  // (k_classify's drop rule)
      if (code == kHitSuspended) {
        go = false, r = kRResume | kRResumeFirst;
        if (kAnyBounce && !P.first) r = kRResume | (e & kQDoomed) | ((e & kQFirst) ? kRResumeFirst : 0u);
      }
    } else if (e & kQResume) {
      go = false, r = kRResume | (e & kQDoomed) | ((e & kQFirst) ? kRResumeFirst : 0u);
    }
    if (go) r = shade_principled_path<MODE>(P, sc, p, rng_inc, P.first != 0u || (e & kQFirst) != 0u, staged ? lds_bsdf : nullptr, lights_staged ? lds_lights : nullptr);
    if (r == kRHold && (e & kQDoomed) && (kAnyBounce ? !P.first : !direct)) r |= kRHoldDoomed;  // (a held path hands its queue entry's kQDoomed back)
    P.q_principled[i] = p | r;
  }
}

// ------------------------------------------------------------------ k_shade_hair (hair-shader.cc:153-229)
__device__ __forceinline__ uint32_t shade_hair_path(const PathState& P, const DScene& sc, uint32_t p, uint64_t rng_inc, bool first,
                                                    const float* lds_lights = nullptr) {
  {
    const bool active = true;
    bool alive = false, shadow = false;
    uint32_t qbit = 0u;
    V3 sh_pos(0.f), c_vis(0.f);
    Nee nee;
    nee.dir = V3(0.f), nee.emission = V3(0.f), nee.dist = 0.f, nee.pdf_sigma = 0.f;
    PathHead c;
    const int head = path_head(P, sc, p, rng_inc, c, first);
    if (active && head == kHeadGoes) {
      const Hit& h = c.h;
      const V3 dir = c.dir, thr = c.thr;
      const Surface& s = c.s;
      Rng& rng = c.rng;
      if (s.face != kAmbiguous) {
        V3 wo_g = -dir;
        Frame fr;
        fr.ex = s.n_s;  // curve tangent
        fr.ey = vnormalize(cross(cross(wo_g, fr.ex), fr.ex));
        fr.ez = cross(fr.ex, fr.ey);
        V3 wo = to_local(fr, wo_g);
        HairBsdf hb = sc.materials[s.material].hair;
        hb.h = h.v;  // :183 (Q12)
        HairSetup S;
        hair_prepare(wo, hb, S);
        V3 d1(0.f);
        shadow = nee_sample(sc, rng, s.pos, fr.ex, false, nee, lds_lights);
        if (shadow) {
          V3 wl = to_local(fr, nee.dir);
          float pdf;
          V3 fcos = hair_eval(S, wl, hb, pdf);
          d1 = nee_contribution(nee, fcos / fabsf(wl.x), pdf);
          sh_pos = s.pos;
        }
        float us[4];
        us[0] = draw(rng), us[1] = draw(rng), us[2] = draw(rng), us[3] = draw(rng);
        V3 wi(0.f);
        float pdf = 0.f;
        V3 fcos = hair_sample(S, hb, us, wi, pdf);
        V3 next_dir = to_global(fr, wi);
        V3 new_thr = fcos / pdf;
        if (!is_finite(new_thr) || !isfinite(pdf)) {
          new_thr = V3(0.f);
          pdf = 0.f;
        }
        c_vis = thr * ((V3(0.f) + d1) + V3(0.f));
        V3 t2 = new_thr * thr;
        if (!is_black(t2)) {
          qbit = doomed_bit(t2, rng.state, rng_inc);
          if (!(qbit && misses_all_lights(sc, s.pos, next_dir, 1e-3f))) {
            alive = true;
            P.ray_o[p] = mk4(s.pos, 1e-3f);
            P.ray_d[p] = mk4(next_dir, kInf);
            P.thr[p] = mk4(t2, pdf);
            store_rng(P, p, rng.state);
          } else {
            qbit = 0u;
            count_pruned(P);
          }
        }
      }
    }
    if (shadow) put_shadow(P, sh_pos, nee, c_vis, V3(0.f), p, kShNormal, alive);
    return head == kHeadHeld ? kRHold : ((shadow ? kRShadow : 0u) | (alive ? kRAlive : 0u) | qbit);
  }
}
__global__ __launch_bounds__(kBlock) void k_shade_hair(PathState P, DScene sc, uint64_t rng_inc) {
  __shared__ float lds_lights[kLdsLightWords];
  const bool lights_staged = stage_light_tables(sc, lds_lights);
  if (lights_staged) __syncthreads();
  const uint32_t n = P.counts[kCntHair];
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const uint32_t e = P.q_hair[i], p = e & kQPathMask;
    uint32_t r = shade_hair_path(P, sc, p, rng_inc, P.first != 0u || (e & kQFirst) != 0u, lights_staged ? lds_lights : nullptr);
    if (r == kRHold && (e & kQDoomed)) r |= kRHoldDoomed;  // (a held path hands its queue entry's kQDoomed back)
    P.q_hair[i] = p | r;
  }
}

// ------------------------------------------------------------------ k_sss_step
// One iteration of RandomWalkSubsurface's loop after its TraceFirstHit1 (random-walk-sss.h:314-405),
// then either the next step's direction/distance sampling (:287-311) or the exit: second NEE + diffuse
// re-sample (cycles-principled-shader.cc:197-216) and the tail of CyclesPrincipledShader (:467-483).
// State of a random walk between two events (random-walk-sss.h:287-311): the bounded ray (org, dir, t_scatter), the
// medium (sigma_t, sigma_s), the walk's throughput, the step index and the path's generator.
struct WalkState {
  V3 org, dir, sigt, sigs, wthr;
  float t_scatter;
  uint32_t bounce;
  uint64_t rng_state;
};
// The event after a bounded ray that hit nothing: scatter at its end (random-walk-sss.h:333-366, :287-311): throughput
// update, Russian roulette, the step bound, then the next isotropic direction and scatter distance.  Returns false when
// the walk FAILS here (roulette, or more than 8192 steps); then `w` is left untouched.  Shared by the wavefront's step
// (sss_step_path) and by the fast-forward kernel (k_sss_walk), so both compute the very same values.
// chpdf_io: null, or the channel pdf of the pending step as the previous call left it (in: *have_chpdf says whether it holds
// one; out: the next step's) -- the same function of the same operands, so carrying it changes no bit.
// albedo: null, or sigma_s / sigma_t (safe_divide_spectrum) computed once for the walk.
__device__ __forceinline__ bool sss_scatter(WalkState& w, uint64_t rng_inc, V3* chpdf_io = nullptr, bool have_chpdf = false, const V3* albedo = nullptr) {
  // what sample_scatter_distance computed when it drew this step's distance
  const V3 chpdf = (chpdf_io && have_chpdf) ? *chpdf_io : scatter_channel_pdf(w.wthr, w.sigs, w.sigt, albedo);
  const V3 trans = attenuate_transmission(w.sigt, w.t_scatter);
  float pdf = dot(chpdf, w.sigt * trans);
  V3 wthr = w.wthr * (w.sigs * trans) / pdf;
  float pr = saturate(spectrum_norm(wthr));
  Rng rng = {w.rng_state, rng_inc};
  float q = draw(rng);
  if (q >= pr) return false;
  wthr = wthr / pr;
  const V3 org = w.org + w.t_scatter * w.dir;
  const uint32_t bounce = w.bounce + 1u;
  if (bounce > 8192u) return false;  // loop bound :287
  // next step: isotropic direction; g++ evaluates UniformSampleSphere(Draw(), Draw()) right-to-left,
  // so the FIRST draw is u2 (:296)
  float first = draw(rng);
  float second = draw(rng);
  V3 wi = vnormalize(uniform_sample_sphere(second, first));
  float e0 = draw(rng);
  float e1 = draw(rng);
  V3 chpdf_next;
  float t_scatter = sample_scatter_distance(wthr, w.sigs, w.sigt, e0, e1, chpdf_next, albedo);
  w.org = org, w.dir = wi, w.wthr = wthr, w.t_scatter = t_scatter, w.bounce = bounce, w.rng_state = rng.state;
  if (chpdf_io) *chpdf_io = chpdf_next;
  return true;
}

// hreg: the hit of the path's bounded ray when the caller holds it in registers, else it is read from P.hit.
__device__ __forceinline__ uint32_t sss_step_path(const PathState& P, const DScene& sc, uint32_t p, uint64_t rng_inc,
                                                  const Hit* hreg = nullptr, const float* lds_lights = nullptr) {
  {
    const bool active = true;
    bool alive = false, shadow = false;
    uint32_t qbit = 0u;
    V3 sh_pos(0.f), c_vis(0.f), c_occ(0.f);
    Nee nee;
    nee.dir = V3(0.f), nee.emission = V3(0.f), nee.dist = 0.f, nee.pdf_sigma = 0.f;
    if (active) {
      float4 o4 = P.ray_o[p], d4 = P.ray_d[p];
      float4 st4 = P.sss_sigt[p], ss4 = P.sss_sigs[p], wt4 = P.sss_thr[p];
      Hit h;
      if (hreg) {
        h = *hreg;
      } else {
        const float4 h4 = P.hit[p];
        h.t = h4.x, h.u = h4.y, h.v = h4.z, h.slot = __float_as_uint(h4.w);
      }
      V3 org = ld3(o4), dir = ld3(d4), sigt = ld3(st4), sigs = ld3(ss4), wthr = ld3(wt4);
      uint32_t bounce = __float_as_uint(wt4.w), entry_inst = __float_as_uint(ss4.w);
      Rng rng = {P.rng[p], rng_inc};
      bool hit = (h.slot != kNone);
      bool fail = false, exited = false;
      if (hit) {
        V3 chpdf = scatter_channel_pdf(wthr, sigs, sigt);  // what sample_scatter_distance computed when it drew this step's distance
        V3 trans = attenuate_transmission(sigt, h.t);
        float pdf = dot(chpdf, trans);
        wthr = wthr * trans / pdf;
        exited = true;
      } else {
        WalkState w = {org, dir, sigt, sigs, wthr, d4.w /* = t_scatter */, bounce, rng.state};
        if (!sss_scatter(w, rng_inc)) {
          fail = true;
        } else {
          P.ray_o[p] = mk4(w.org, 0.f);
          P.ray_d[p] = mk4(w.dir, w.t_scatter);
          P.sss_thr[p] = mk4(w.wthr, __uint_as_float(w.bounce));
          P.rng[p] = w.rng_state;
          alive = true;
          qbit = kQSssBit;
        }
      }
      V3 thr = ld3(P.thr[p]);
      V3 A = ld3(P.sss_A[p]);
      if (alive) {
        // (the walk goes on: nothing else to do)
      } else if (exited) {
        uint32_t exit_inst;
        Surface s = make_surface(sc, org, dir, h, &exit_inst);  // :369
        if (exit_inst != entry_inst) fail = true;               // :372 (Q6)
        if (s.face != kBack) fail = true;                      // :376
        if (!fail) {
          Frame fx;  // exit frame :382-394
          fx.ez = s.n_s;
          branchless_onb(fx.ez, fx.ex, fx.ey);
          V3 wo = to_local(fx, dir);
          PrincipledBsdf nb = default_bsdf();  // cycles-principled-shader.cc:198-200
          nb.enable_diffuse = 1;
          nb.diffuse_weight = wthr;
          SampleWeight w = closure_sample_weight(wo, nb);
          V3 d2(0.f);
          shadow = nee_sample(sc, rng, s.pos, s.n_s, true, nee, lds_lights);  // :202-212 (Q5)
          if (shadow) {
            V3 f;
            float pdf;
            eval_bsdf(to_local(fx, nee.dir), wo, nb, w, f, pdf);
            d2 = nee_contribution(nee, f, pdf);
            sh_pos = s.pos;
          }
          float select = draw(rng);
          int pick = pick_closure(select, w);
          V3 wi(0.f);
          float u0 = draw(rng);
          float u1 = draw(rng);
          if (pick == 0) {
            float pdf;
            lambert_sample(u0, u1, wi, pdf);
          } else {
            // diffuse weight 0 (NaN -> 0) falls through to the clearcoat branch with alpha (1,1) (Q7)
            ggx_sample(wo, nb.clearcoat_alpha_x, nb.clearcoat_alpha_y, u0, u1, wi);
          }
          V3 f;
          float pdf;
          eval_bsdf(wi, wo, nb, w, f, pdf);
          // local -> global with the ENTRY frame (cycles-principled-shader.cc:467-469)
          Frame fe;
          fe.ez = ld3(P.sss_ez[p]);
          branchless_onb(fe.ez, fe.ex, fe.ey);
          V3 next_dir = to_global(fe, wi);
          V3 new_thr = f * fabsf(wi.z) / pdf;
          if (!is_finite(new_thr) || !isfinite(pdf)) {
            new_thr = V3(0.f);
            pdf = 0.f;
          }
          c_vis = thr * (A + d2);
          c_occ = thr * (A + V3(0.f));
          if (!shadow) {
            float4 L4 = P.L[p];
            P.L[p] = mk4(ld3(L4) + c_occ, L4.w);
          }
          V3 t2 = new_thr * thr;
          if (!is_black(t2)) {
            qbit = doomed_bit(t2, rng.state, rng_inc);
            if (!(qbit && misses_all_lights(sc, s.pos, next_dir, 1e-3f))) {
              alive = true;
              P.ray_o[p] = mk4(s.pos, 1e-3f);
              P.ray_d[p] = mk4(next_dir, kInf);
              P.thr[p] = mk4(t2, pdf);
              P.rng[p] = rng.state;
            } else {
              qbit = 0u;
              count_pruned(P);
            }
          }
        }
      }
      if (fail) {
        // walk failed: path ends, the first NEE's contribution still counts (render.cc:79)
        float4 L4 = P.L[p];
        P.L[p] = mk4(ld3(L4) + thr * (A + V3(0.f)), L4.w);
      }
    }
    if (shadow) put_shadow(P, sh_pos, nee, c_vis, c_occ, p, kShSssExit, alive);
    return (shadow ? kRShadow : 0u) | (alive ? kRAlive : 0u) | qbit;
  }
}
__global__ __launch_bounds__(kBlock) void k_sss_step(PathState P, DScene sc, uint64_t rng_inc) {
  __shared__ float lds_lights[kLdsLightWords];
  const bool lights_staged = stage_light_tables(sc, lds_lights);
  if (lights_staged) __syncthreads();
  const uint32_t n = P.counts[kCntSss];
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const uint32_t p = P.q_sss[i];
    P.q_sss[i] = p | sss_step_path(P, sc, p, rng_inc, nullptr, lights_staged ? lds_lights : nullptr);
  }
}

// ------------------------------------------------------------------ k_sss_walk
// Fast-forward of the random walks (RandomWalkSubsurface's loop, random-walk-sss.h:287-405).  A walk alternates a bounded
// closest-hit ray with an event: most events are plain scatterings (the ray hit nothing: new direction, new distance),
// the last one is the exit (the ray hit the boundary) or the walk's failure.  One wavefront iteration per event cost a C3
// frame ~470 iterations of trace / classify / step / compact over mostly in-medium paths.  This kernel runs between
// k_classify and k_sss_step: a lane takes an in-medium path whose pending event is a scattering, keeps the walk's state in
// registers, applies the event (sss_scatter: the code the step kernel itself runs), traces the next bounded ray with the
// persistent phase-voting traversal (dtrace_pv.h), and repeats until the pending event is no plain scattering any more --
// a hit, or a scattering that fails (which is NOT applied).  It then leaves the path exactly as the wavefront would have
// left it before that event (ray, throughput, step index, generator state, hit record), and k_sss_step handles the event
// as it always did.  A path therefore spends one or two wavefront iterations inside a medium instead of one per step; the
// values computed are the same, in the same order.
#ifndef PB_WALK_CAP
#define PB_WALK_CAP 24u  // A/B on C3 (walk + step kernels, ms per frame): no cap 230, 128: 183, 64: 164, 32: 149, 24: 140, 16: 139, 12: 134 (but more iterations), 8: 138
#endif
constexpr uint32_t kWalkCap = PB_WALK_CAP;
constexpr uint32_t kWalkWords = 22;  // words of walk state per lane in LDS (WalkSink)
struct WalkSink {
  static constexpr bool kWalk = true;
  const PathState& P;
  uint64_t rng_inc;
  // What a walk carries besides its ray (which the traversal holds in registers anyway) lives in LDS, kWalkWords words per
  // lane (wl[k * kBlock]): throughput, scatter distance, step index, generator state, scatterings applied, path slot; the
  // medium's sigma_t / sigma_s (fetched once per launch instead of two gathers in front of every scattering) and the channel
  // pdf of the pending step as the previous scattering computed it.  This keeps the kernel's registers at the traversal's.
  float* wl;
  uint32_t n_rays;  // rays traced by this lane (STATS)
  __device__ __forceinline__ void start(uint32_t idx, uint32_t& tag, Hit& h, V3& o, V3& d) {
    tag = idx;
    const uint32_t p = P.q_sss[idx] & kQPathMask;
    const float4 h4 = P.hit[p];
    h.t = h4.x, h.u = h4.y, h.v = h4.z, h.slot = __float_as_uint(h4.w);
    wl[7 * kBlock] = __uint_as_float(0u), wl[8 * kBlock] = __uint_as_float(p);
    wl[21 * kBlock] = __uint_as_float(kNone);  // (the walk's instance: known once the walk is fetched)
    if (h.slot == kNone) {  // a scattering is pending: fetch the walk
      const float4 o4 = P.ray_o[p], d4 = P.ray_d[p], wt4 = P.sss_thr[p];
      const uint64_t r = P.rng[p];
      o = ld3(o4), d = ld3(d4);
      wl[0] = wt4.x, wl[kBlock] = wt4.y, wl[2 * kBlock] = wt4.z, wl[3 * kBlock] = d4.w, wl[4 * kBlock] = wt4.w;
      wl[5 * kBlock] = __uint_as_float((uint32_t)r), wl[6 * kBlock] = __uint_as_float((uint32_t)(r >> 32));
      const float4 st4 = P.sss_sigt[p], ss4 = P.sss_sigs[p];
      wl[9 * kBlock] = st4.x, wl[10 * kBlock] = st4.y, wl[11 * kBlock] = st4.z;
      wl[12 * kBlock] = ss4.x, wl[13 * kBlock] = ss4.y, wl[14 * kBlock] = ss4.z;
      wl[21 * kBlock] = ss4.w;  // the instance the walk entered (its bits)
      const V3 albedo = safe_divide_spectrum(ld3(ss4), ld3(st4));  // what scatter_channel_pdf derives from them every time
      wl[18 * kBlock] = albedo.x, wl[19 * kBlock] = albedo.y, wl[20 * kBlock] = albedo.z;
    }
  }
  __device__ __forceinline__ bool next(uint32_t tag, const Hit& h, V3& o, V3& d, float& tmin, float& tmax) {
    const uint32_t p = __float_as_uint(wl[8 * kBlock]);
    // A launch ends with its longest walk, so a walk is fast-forwarded by at most kWalkCap scatterings per launch; a longer
    // one is handed back (k_sss_step applies its pending scattering) and goes on in the next iteration's launch.
    if (h.slot == kNone && __float_as_uint(wl[7 * kBlock]) < kWalkCap) {
      WalkState w;
      w.org = o, w.dir = d;
      w.sigt = V3(wl[9 * kBlock], wl[10 * kBlock], wl[11 * kBlock]), w.sigs = V3(wl[12 * kBlock], wl[13 * kBlock], wl[14 * kBlock]);
      const bool carried = __float_as_uint(wl[7 * kBlock]) != 0u;  // a scattering of this launch left the pending step's channel pdf
      V3 chpdf(wl[15 * kBlock], wl[16 * kBlock], wl[17 * kBlock]);
      const V3 albedo(wl[18 * kBlock], wl[19 * kBlock], wl[20 * kBlock]);
      w.wthr = V3(wl[0], wl[kBlock], wl[2 * kBlock]), w.t_scatter = wl[3 * kBlock], w.bounce = __float_as_uint(wl[4 * kBlock]);
      w.rng_state = (uint64_t)__float_as_uint(wl[5 * kBlock]) | ((uint64_t)__float_as_uint(wl[6 * kBlock]) << 32);
      if (sss_scatter(w, rng_inc, &chpdf, carried, &albedo)) {
        o = w.org, d = w.dir, tmin = 0.f, tmax = w.t_scatter;
        wl[15 * kBlock] = chpdf.x, wl[16 * kBlock] = chpdf.y, wl[17 * kBlock] = chpdf.z;
        wl[0] = w.wthr.x, wl[kBlock] = w.wthr.y, wl[2 * kBlock] = w.wthr.z, wl[3 * kBlock] = w.t_scatter;
        wl[4 * kBlock] = __uint_as_float(w.bounce);
        wl[5 * kBlock] = __uint_as_float((uint32_t)w.rng_state), wl[6 * kBlock] = __uint_as_float((uint32_t)(w.rng_state >> 32));
        wl[7 * kBlock] = __uint_as_float(__float_as_uint(wl[7 * kBlock]) + 1u);
        n_rays++;
        return true;
      }
    }
    if (__float_as_uint(wl[7 * kBlock])) {  // hand the path back as the wavefront would have left it before the pending event
      P.ray_o[p] = mk4(o, 0.f);
      P.ray_d[p] = mk4(d, wl[3 * kBlock]);
      P.sss_thr[p] = make_float4(wl[0], wl[kBlock], wl[2 * kBlock], wl[4 * kBlock]);
      P.rng[p] = (uint64_t)__float_as_uint(wl[5 * kBlock]) | ((uint64_t)__float_as_uint(wl[6 * kBlock]) << 32);
      P.hit[p] = make_float4(h.t, h.u, h.v, __uint_as_float(h.slot));
    }
    return false;
  }
  // Where this walk's next ray starts (dscene.h::SssEntry): the entry node of the walk's instance when both ends of the ray lie inside the
  // instance's (widened) bounds, with the foreign references the ray's interval meets pushed on its stack; otherwise the root.  The box
  // test of a foreign reference is the binary tree's (box_test2's operations on the box the reference's parent presents): conservative and
  // monotone in the box, so no reference that can hold an accepted hit is dropped.
  template <typename Push>
  __device__ __forceinline__ void entry(const DScene& sc, V3 o, V3 d, V3 inv, float tmin, float tmax, uint32_t& next, Push push) const {
    const uint32_t inst = __float_as_uint(wl[21 * kBlock]);
    if (sc.sss_entries == nullptr) return;
    // the far end of the ray (any rounding is far inside the margins the bounds were widened by; a NaN fails the test below)
    const V3 end(o.x + tmax * d.x, o.y + tmax * d.y, o.z + tmax * d.z);
    // The lanes of a wave nearly always walk in ONE instance: the instance's record is fetched with scalar loads (a wave-uniform
    // address: one fetch for the wave instead of sixteen 16-byte loads per lane) -- a waterfall loop over the distinct instances.
    for (unsigned long long todo = __ballot(true); todo != 0ull;) {
      const uint32_t u = (uint32_t)__builtin_amdgcn_readlane((int)inst, __ffsll((long long)todo) - 1);
      const bool mine = inst == u;
      todo &= ~__ballot(mine);
      if (u >= sc.num_sss_entries) continue;
      // (the constant address space: read-only during the launch and wave-uniform -> s_load_dwordx4)
      typedef const __attribute__((address_space(4))) float* ConstF;
      const ConstF ef = (ConstF)(reinterpret_cast<uintptr_t>(sc.sss_entries + u));
      auto word = [&](uint32_t i) { return make_float4(ef[4 * i], ef[4 * i + 1], ef[4 * i + 2], ef[4 * i + 3]); };
      const float4 e0 = word(0), e1 = word(1);
      const uint32_t ent = __float_as_uint(e0.w), nf = __float_as_uint(e1.w);
      if (ent == 0u) continue;
      const bool inside = mine && o.x >= e0.x && o.y >= e0.y && o.z >= e0.z && o.x <= e1.x && o.y <= e1.y && o.z <= e1.z && end.x >= e0.x &&
                          end.y >= e0.y && end.z >= e0.z && end.x <= e1.x && end.y <= e1.y && end.z <= e1.z;
      if (inside) next = ent;
      for (uint32_t k = 0; k < nf; k++) {
        const float4 f0 = word(2 + 2 * k), f1 = word(3 + 2 * k);
        float t0 = (f0.x - o.x) * inv.x, t1 = (f1.x - o.x) * inv.x;
        float a = __builtin_fminf(t0, t1), b = __builtin_fmaxf(t0, t1);
        t0 = (f0.y - o.y) * inv.y, t1 = (f1.y - o.y) * inv.y;
        a = __builtin_fmaxf(a, __builtin_fminf(t0, t1)), b = __builtin_fminf(b, __builtin_fmaxf(t0, t1));
        t0 = (f0.z - o.z) * inv.z, t1 = (f1.z - o.z) * inv.z;
        a = __builtin_fmaxf(a, __builtin_fminf(t0, t1)), b = __builtin_fminf(b, __builtin_fmaxf(t0, t1));
        const float eps = 1.52587890625e-05f;
        a = __builtin_fmaf(-fabsf(a), eps, a), b = __builtin_fmaf(fabsf(b), eps, b);
        if (inside && a <= b && b >= tmin && a <= tmax) push(__float_as_uint(f0.w));
      }
    }
  }
  // (trace_pv only calls these for sinks that do not walk)
  __device__ __forceinline__ bool load(uint32_t, uint32_t&, V3&, V3&, float&, float&) const { return false; }
  __device__ __forceinline__ void done(uint32_t, const Hit&, bool) const {}
};
#ifndef PB_WALK_WAVES
#define PB_WALK_WAVES 3  // waves per SIMD of k_sss_walk: <= 168 VGPRs, nothing spilled
#endif
#ifndef PB_WALK_WAVES_TRI
#define PB_WALK_WAVES_TRI 4  // ... on the Q tree of a triangle-only scene (C3): 116-121 VGPRs, nothing spilled, 38.9 KB of LDS: four blocks per CU (round 5: 26.8 -> 25.9 ms per 64 spp of C3; round 6, measured and not kept: the entry node tested inside the step turn needs ~150 registers = three blocks, which costs what it saves)
#endif
constexpr uint32_t walk_blocks_per_cu(bool curves, bool wide) { return (!curves && wide) ? PB_WALK_WAVES_TRI : PB_WALK_WAVES; }
template <bool STATS, bool CURVES, bool WIDE = false>
__global__ __launch_bounds__(kBlock, STATS ? PB_WALK_WAVES : walk_blocks_per_cu(CURVES, WIDE)) void k_sss_walk(PathState P, DScene sc, uint64_t rng_inc) {
  __shared__ uint32_t stk[pv_lds_stack<CURVES, WIDE>() * kBlock];
  __shared__ float frm[CURVES ? 10 * kBlock : 1];
  __shared__ float walk[kWalkWords * kBlock];
  const uint32_t n = P.counts[kCntSss];
  TravStats st = {};
  uint32_t overflow = 0u;
  WalkSink sink = {P, rng_inc, walk + threadIdx.x, 0u};
  trace_pv<0, STATS, CURVES, WIDE>(sc, n, &P.counts[kCntWalkHead], sink, stk + threadIdx.x, kBlock,
                             P.spill + blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock, st, &overflow,
                             CURVES ? frm + threadIdx.x : nullptr);
  if (overflow) P.counts[kCntOverflow] = 1u;
  if (STATS) {
    const uint32_t a = wave_sum(sink.n_rays), nn = wave_sum(st.nodes), nt = wave_sum(st.tris + st.curves);
    if (__lane_id() == 0 && a) atomicAdd(&P.stats[kStatTailClosestRays], (unsigned long long)a);
    if (__lane_id() == 0) {
      atomicAdd(&P.stats[kStatWalkNodes], (unsigned long long)nn), atomicAdd(&P.stats[kStatWalkTris], (unsigned long long)nt);
      atomicAdd(&P.stats[kStatWalkTurns], (unsigned long long)(st.it_node + st.it_tri + st.it_curve));
      atomicAdd(&P.stats[kStatWalkSteps], (unsigned long long)st.it_refill);
      atomicAdd(&P.stats[kStatWalkCycTrav], st.cyc[0] + st.cyc[1] + st.cyc[2]), atomicAdd(&P.stats[kStatWalkCycStep], st.cyc[3]);
    }
  }
}

// ------------------------------------------------------------------ k_tail
// The tail of a chunk -- few live paths, many bounces left -- is bound by latency, not throughput: every wavefront
// iteration costs ~0.2 ms of launches and drains however few paths it moves (91 iterations on C2, 70 of them with
// < 64 Ki paths).  Below pbrhip_render_desc.tail_paths live paths the remaining bounces of every path therefore run in
// ONE launch: a lane owns a path and loops  shade -> (its shadow ray, resolved at once) -> closest-hit trace  until the
// path ends, calling the very functions the wavefront kernels call, in the order the wavefront applies them to that
// path (the shadow ray of bounce k is resolved before anything of bounce k+1 touches L), so results are unchanged.
// Input: q_in = the paths just traced by k_trace (their hit records are in P.hit).  Traversal is the plain per-lane
// one (dtrace.h::traverse): with a handful of lanes per wave there is nothing to vote on.
// Lanes: a wave starts with one path per lane (up to 64).  Once at most 32 of them are alive they are moved to the even
// lanes and every odd lane becomes its neighbour's helper: it traverses the path's shadow ray while the even lane
// traverses the continuation ray -- the two dependent-load chains of a bounce run side by side instead of one after the
// other (what bounds k_tail is the chain of the longest path, not throughput).
#ifndef PB_TAIL_OCTETS
#define PB_TAIL_OCTETS 1  // triangle-only Q trees: eight lanes per path once at most eight paths of a wave are alive (below)
#endif
#ifndef PB_TAIL_WAVES
#define PB_TAIL_WAVES 3  // min waves per SIMD of k_tail (<= 168 VGPRs: three blocks per CU hold 196 k lanes, so every path of a 256 Ki tail starts at once;
                         // A/B on C2: 2 -> 59.1 ms per frame / 12.1 ms for an eighth, 3 -> 58.6 / 11.8)
#endif
// MODE: what the scene's materials can do (shade_principled_path): the branches no path can take are compiled out
template <int MODE, bool STATS, bool CURVES, bool WIDE = false>
__global__ __launch_bounds__(kBlock, PB_TAIL_WAVES) void k_tail(PathState P, DScene sc, uint64_t rng_inc) {
  __shared__ uint32_t stk[kSimpleLdsStack * kBlock];
  uint32_t* const spill = P.spill + blockIdx.x * kBlock + threadIdx.x;  // stack entries beyond the LDS part (the group's spill area: this grid is smaller than k_trace's)
  const uint32_t spill_stride = gridDim.x * kBlock;
  const uint32_t n = P.counts[kCntIn];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = gridDim.x * (kBlock >> 6);
  uint32_t per_wave = (n + nwaves - 1u) / nwaves;  // spread thin: every SIMD busy, little divergence per wave
  per_wave = per_wave < 1u ? 1u : (per_wave > 64u ? 64u : per_wave);
  TravStats st = {};
  uint32_t overflow = 0u, n_closest = 0u, n_shadow = 0u;  // rays traced here (STATS)
  const TraceSink sink = {P, 0u, 0u};
  uint32_t* const stack = stk + threadIdx.x;
  constexpr uint32_t kHave = 0x80000000u, kMedium = 0x40000000u, kFirst = 0x20000000u;  // lane state = path slot | flags
  constexpr bool kOctets = PB_TAIL_OCTETS != 0 && WIDE && !CURVES;
  for (uint32_t base = wave * per_wave; base < n; base += nwaves * per_wave) {  // (wave-uniform loop)
    uint32_t state = 0u;
    if (lane < per_wave && base + lane < n) {
      const uint32_t e = P.first ? P.slot0 + base + lane : P.q_in[base + lane];
      state = (e & kQPathMask) | kHave | ((!P.no_medium && (e & kQSssBit)) ? kMedium : 0u) | ((P.first || (e & kQFirst)) ? kFirst : 0u);  // the tail starts at the very first bounce of tiny renders (kQFirst: a camera ray that was suspended on its way)
    }
    uint32_t team = 1u;  // lanes per path: 1, 2 (pairs) or 8 (octets)
    for (;;) {
      const unsigned long long act = __ballot((state & kHave) != 0u);
      if (act == 0ull) break;
      const uint32_t alive = (uint32_t)__popcll(act);
      const uint32_t want_team = (kOctets && alive <= 8u) ? 8u : (alive <= 32u ? 2u : 1u);
      if (want_team > team) {
        // move the k-th live path to lane team * k (its state is this one word; everything else lives in the path's slot)
        uint32_t src = lane;
        bool mine = false;
        uint32_t k = 0;
        for (unsigned long long m = act; m; m &= m - 1ull, k++)
          if (lane == want_team * k) src = (uint32_t)__builtin_ctzll(m), mine = true;
        const uint32_t moved = (uint32_t)__shfl((int)state, (int)src);
        state = mine ? moved : 0u;
        team = want_team;
      }
      const bool paired = team == 2u;
      const uint32_t p = state & kQPathMask;
      uint32_t r = 0u;
      if (state & kHave) {
        if (MODE != kShadePlain && (state & kMedium)) {
          r = sss_step_path(P, sc, p, rng_inc);
        } else {
          const uint32_t slot = __float_as_uint(P.hit[p].w);
          if (slot != kNone)  // a miss ends the path (render.cc:34)
            r = (slot & kHitHair) ? shade_hair_path(P, sc, p, rng_inc, (state & kFirst) != 0u)
                                  : shade_principled_path<MODE>(P, sc, p, rng_inc, (state & kFirst) != 0u);
        }
        state &= ~kFirst;
      }
      const bool want_shadow = (r & kRShadow) != 0u, want_closest = (r & kRAlive) != 0u;
      if (kOctets && team == 8u) {
        // lanes 0-3 of the octet: the path's continuation ray; lanes 4-7: its shadow ray -- one ray per quad (dtrace_quad.h)
        const bool upper = (lane & 4u) != 0u;
        float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f), d4 = o4, s4 = o4;
        if (want_shadow || want_closest) o4 = P.ray_o[p];
        if (want_closest) d4 = P.ray_d[p];
        if (want_shadow) s4 = P.sh_d[p];
        const int lead = (int)(lane & ~7u);
        const float ox = __shfl(o4.x, lead), oy = __shfl(o4.y, lead), oz = __shfl(o4.z, lead), ow = __shfl(o4.w, lead);
        const float cx = __shfl(d4.x, lead), cy = __shfl(d4.y, lead), cz = __shfl(d4.z, lead), cw = __shfl(d4.w, lead);
        const float sx = __shfl(s4.x, lead), sy = __shfl(s4.y, lead), sz = __shfl(s4.z, lead), sw = __shfl(s4.w, lead);
        const float dx = upper ? sx : cx, dy = upper ? sy : cy, dz = upper ? sz : cz, dw = upper ? sw : cw;
        const uint32_t wants = (uint32_t)__shfl((int)((want_closest ? 1u : 0u) | (want_shadow ? 2u : 0u)), lead);
        const bool go = (wants & (upper ? 2u : 1u)) != 0u;
        Hit h = {0.f, 0.f, 0.f, kNone};
        bool occluded = false;
        const uint32_t col = threadIdx.x & ~3u;
        if (go) occluded = traverse_quad<2>(sc, V3(ox, oy, oz), V3(dx, dy, dz), ow, dw, h, stk + col, kBlock, &overflow, upper,
                                            P.spill + blockIdx.x * kBlock + col, spill_stride);
        const bool occ_up = __shfl((int)occluded, lead + 4) != 0;
        if (want_shadow) {
          Hit none = {0.f, 0.f, 0.f, kNone};
          sink.done(p | 0x80000000u, none, occ_up);
          n_shadow++;
        }
        if (want_closest) {
          P.hit[p] = make_float4(h.t, h.u, h.v, __uint_as_float(h.slot));
          n_closest++;
        }
      } else if (!paired) {
        if (want_shadow) {
          const float4 o4 = P.ray_o[p], d4 = P.sh_d[p];
          Hit h;
          const bool occluded = traverse<true, false, CURVES, WIDE>(sc, ld3(o4), ld3(d4), o4.w, d4.w, h, stack, kBlock, st, &overflow, spill, spill_stride);
          sink.done(p | 0x80000000u, h, occluded);
          n_shadow++;
        }
        if (want_closest) {
          const float4 o4 = P.ray_o[p], d4 = P.ray_d[p];
          Hit h;
          traverse<false, false, CURVES, WIDE>(sc, ld3(o4), ld3(d4), o4.w, d4.w, h, stack, kBlock, st, &overflow, spill, spill_stride);
          P.hit[p] = make_float4(h.t, h.u, h.v, __uint_as_float(h.slot));
          n_closest++;
        }
      } else {
        // even lane: the path and its continuation ray; odd lane: the shadow ray of the even lane to its left
        const bool odd = (lane & 1u) != 0u;
        float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f), d4 = o4, s4 = o4;
        if (want_shadow || want_closest) o4 = P.ray_o[p];
        if (want_closest) d4 = P.ray_d[p];
        if (want_shadow) s4 = P.sh_d[p];
        const int left = (int)(lane & ~1u);
        const float sx = __shfl(s4.x, left), sy = __shfl(s4.y, left), sz = __shfl(s4.z, left), sw = __shfl(s4.w, left);
        const float ox = __shfl(o4.x, left), oy = __shfl(o4.y, left), oz = __shfl(o4.z, left), ow = __shfl(o4.w, left);
        const bool left_shadow = __shfl((int)want_shadow, left) != 0;
        bool go = want_closest;
        if (odd) o4 = make_float4(ox, oy, oz, ow), d4 = make_float4(sx, sy, sz, sw), go = left_shadow;
        Hit h = {0.f, 0.f, 0.f, kNone};
        bool occluded = false;
        if (go) occluded = traverse_mode<2, false, CURVES, WIDE>(sc, ld3(o4), ld3(d4), o4.w, d4.w, h, stack, kBlock, st, &overflow, odd, spill, spill_stride);
        const bool occ_right = __shfl((int)occluded, (int)(lane | 1u)) != 0;
        if (want_shadow) {
          Hit none = {0.f, 0.f, 0.f, kNone};
          sink.done(p | 0x80000000u, none, occ_right);  // (a shadow ray's result is the one bit)
          n_shadow++;
        }
        if (want_closest) {
          P.hit[p] = make_float4(h.t, h.u, h.v, __uint_as_float(h.slot));
          n_closest++;
        }
      }
      if (!want_closest) state = 0u;
      else state = (state & ~kMedium) | ((r & kQSssBit) ? kMedium : 0u);
    }
  }
  if (overflow) P.counts[kCntOverflow] = 1u;
  if (STATS) {
    const uint32_t a = wave_sum(n_closest), b = wave_sum(n_shadow);
    if (lane == 0 && a) atomicAdd(&P.stats[kStatTailClosestRays], (unsigned long long)a);
    if (lane == 0 && b) atomicAdd(&P.stats[kStatTailShadowRays], (unsigned long long)b);
  }
}

// ------------------------------------------------------------------ k_accumulate (render.cc:175-183, Q13)
// One thread per pixel of this rank's tiles; passes of the chunk are added in ascending order.
__global__ __launch_bounds__(kBlock) void k_accumulate(PathState P, const uint32_t* __restrict__ pix_index, uint32_t npix,
                                                       uint32_t npass, float* __restrict__ rgba,
                                                       uint32_t* __restrict__ count) {
  const uint32_t R = P.pass_run;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npix; i += gridDim.x * kBlock) {
    uint32_t g = pix_index[i];
    float4 acc = reinterpret_cast<float4*>(rgba)[g];
    uint32_t c = count[g];
    if (R == 1u) {
      for (uint32_t k = 0; k < npass; k++) {
        float4 L = P.L[(size_t)k * npix + i];
        acc.x += L.x, acc.y += L.y, acc.z += L.z, acc.w += 1.0f;
        c++;
      }
    } else {
      // runs of R passes of this pixel are adjacent (PathState::pass_run): a thread streams its own runs, eight radiances = one
      // 128-byte line per batch of loads; the sums stay in ascending pass order
      for (uint32_t k0 = 0; k0 < npass; k0 += R) {
        const float4* run = P.L + ((size_t)(k0 / R) * npix + i) * R;
        for (uint32_t r0 = 0; r0 < R; r0 += 8u) {
          float4 L[8];
#pragma unroll
          for (uint32_t r = 0; r < 8u; r++) L[r] = (r0 + r < R) ? run[r0 + r] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (uint32_t r = 0; r < 8u; r++)
            if (r0 + r < R) acc.x += L[r].x, acc.y += L[r].y, acc.z += L[r].z, acc.w += 1.0f, c++;
        }
      }
    }
    reinterpret_cast<float4*>(rgba)[g] = acc;
    count[g] = c;
  }
}

// ------------------------------------------------------------------ leaf functions as a test hook (pbrhip_leaf_eval)
// One item per thread: `in` = in_words floats per item, `out` = out_words per item; op = PBRHIP_LEAF_* (include/pbrhip.h).  The device's
// generator, fast math, samplers and closures on the inputs the committed vectors of the REFERENCE's own leaf code were taken at
// (tests/golden/ref_leaf_kats.npz): the -m gpu test compares the two directly, bit for bit.
__global__ __launch_bounds__(kBlock) void k_leaf_eval(uint32_t op, const float* __restrict__ in, uint32_t n, uint32_t in_words, float* __restrict__ out,
                                                      uint32_t out_words) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float* a = in + (size_t)i * in_words;
    float* o = out + (size_t)i * out_words;
    switch (op) {
      case 0: {  // PCG32: (initstate lo, hi, initseq lo, hi as bits) -> out_words draws
        const uint64_t st = (uint64_t)__float_as_uint(a[0]) | ((uint64_t)__float_as_uint(a[1]) << 32);
        const uint64_t sq = (uint64_t)__float_as_uint(a[2]) | ((uint64_t)__float_as_uint(a[3]) << 32);
        Rng r = rng_seed(st, sq);
        for (uint32_t k = 0; k < out_words; k++) o[k] = draw(r);
      } break;
      case 1: {  // fast math: (function as bits, x, y)
        const uint32_t f = __float_as_uint(a[0]);
        float sn, cs;
        fastm::fsincos(a[1], sn, cs);
        o[0] = f == 0 ? fastm::fsin(a[1]) : f == 1 ? fastm::fcos(a[1]) : f == 2 ? fastm::fexp(a[1]) : f == 3 ? fastm::flog(a[1]) : f == 4 ? fastm::fatan2(a[1], a[2])
             : f == 5 ? fastm::fasin(a[1]) : f == 6 ? fastm::fexp2(a[1]) : f == 7 ? fastm::flog2(a[1]) : f == 8 ? sn : cs;
      } break;
      case 2: o[0] = fresnel_dielectric_cos(a[0], a[1]); break;
      case 3: o[0] = power_heuristic(a[0], a[1]); break;
      case 4: {  // Lambert sample: (u0, u1) -> wi, f, pdf
        V3 wi(0.f);
        float pdf = 0.f;
        const float f = lambert_sample(a[0], a[1], wi, pdf);
        o[0] = wi.x, o[1] = wi.y, o[2] = wi.z, o[3] = f, o[4] = pdf;
      } break;
      case 5: {
        const V3 v = uniform_sample_sphere(a[0], a[1]);
        o[0] = v.x, o[1] = v.y, o[2] = v.z;
      } break;
      case 6: triangle_uniform_sampler(a[0], a[1], o[0], o[1]); break;
      case 7: {  // GGX eval: (wi, wo, ax, ay, distrib as bits) -> f, pdf
        float pdf = 0.f;
        o[0] = ggx_eval(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), a[6], a[7], (int)__float_as_uint(a[8]), pdf);
        o[1] = pdf;
      } break;
      case 8: {  // GGX sample: (wo, ax, ay, u0, u1, distrib as bits) -> wi, f, pdf (microfacet-ggx.h:247-286: eval of the sampled direction)
        const V3 wo(a[0], a[1], a[2]);
        V3 wi(0.f);
        float pdf = 0.f, f = 0.f;
        if (wo.z > 0.f) {
          const V3 m = microfacet_sample_stretched(wo, a[3], a[4], a[5], a[6]);
          const float cos_mo = dot(m, wo);
          if (cos_mo > 0) {
            wi = 2 * cos_mo * m - wo;
            f = ggx_eval(wi, wo, a[3], a[4], (int)__float_as_uint(a[7]), pdf);
          }
        }
        o[0] = wi.x, o[1] = wi.y, o[2] = wi.z, o[3] = f, o[4] = pdf;
      } break;
      case 9:
      case 10: {  // hair: (wi, wo, params[23]) -> f, pdf   /   (wo, params[23], us[4]) -> wi, f, pdf
        const float* p = a + (op == 9 ? 6 : 3);
        HairBsdf b;
        b.h = p[0];
        for (int k = 0; k < 4; k++) b.v[k] = p[1 + k];
        b.s = p[5], b.sigma_a = V3(p[6], p[7], p[8]), b.eta = p[9], b.alpha = p[10];
        for (int k = 0; k < 4; k++) b.tints[k] = V3(p[11 + 3 * k], p[12 + 3 * k], p[13 + 3 * k]);
        b.transparent_scale = p[22];
        HairSetup S;
        float pdf = 0.f;
        if (op == 9) {
          hair_prepare(V3(a[3], a[4], a[5]), b, S);
          const V3 f = hair_eval(S, V3(a[0], a[1], a[2]), b, pdf);
          o[0] = f.x, o[1] = f.y, o[2] = f.z, o[3] = pdf;
        } else {
          hair_prepare(V3(a[0], a[1], a[2]), b, S);
          const float us[4] = {a[26], a[27], a[28], a[29]};
          V3 wi(0.f);
          const V3 f = hair_sample(S, b, us, wi, pdf);
          o[0] = wi.x, o[1] = wi.y, o[2] = wi.z, o[3] = f.x, o[4] = f.y, o[5] = f.z, o[6] = pdf;
        }
      } break;
      default: break;
    }
  }
}
// Texture::FetchFloat3 on an array of (u, v) (test hook pbrhip_texture_fetch): the fetch the textured shading kernels run
__global__ __launch_bounds__(kBlock) void k_texture_fetch(DScene sc, uint32_t tex_id, const float* __restrict__ uv, uint32_t n, float* __restrict__ rgb) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const V3 c = texture_fetch3(sc, tex_id, uv[2 * i], uv[2 * i + 1]);
    rgb[3 * i] = c.x, rgb[3 * i + 1] = c.y, rgb[3 * i + 2] = c.z;
  }
}
void launch_texture_fetch(hipStream_t s, const DScene& sc, uint32_t tex_id, const float* uv, uint32_t n, float* rgb) {
  const uint32_t blocks = (n + kBlock - 1u) / kBlock;
  hipLaunchKernelGGL(k_texture_fetch, dim3(blocks < 1u ? 1u : (blocks > 1024u ? 1024u : blocks)), dim3(kBlock), 0, s, sc, tex_id, uv, n, rgb);
}
void launch_leaf_eval(hipStream_t s, uint32_t op, const float* in, uint32_t n, uint32_t in_words, float* out, uint32_t out_words) {
  const uint32_t blocks = (n + kBlock - 1u) / kBlock;
  hipLaunchKernelGGL(k_leaf_eval, dim3(blocks < 1u ? 1u : (blocks > 1024u ? 1024u : blocks)), dim3(kBlock), 0, s, op, in, n, in_words, out, out_words);
}

// ------------------------------------------------------------------ RenderLayer shards (multi-GPU exchange, multi.cpp)
// A rank's share of the frame is the pixel list `pix` (the blocks dealt to it).  pack: shard = [rgba of every listed
// pixel | count of every listed pixel]; unpack_add: layer[pix[i]] += shard[i] (the receiving layer holds zeros there, so
// this is the reduce(sum) of SURVEY 8e restricted to the terms that are not zero by construction).
__global__ __launch_bounds__(kBlock) void k_layer_pack(const uint32_t* __restrict__ pix, uint32_t npix,
                                                       const float4* __restrict__ rgba, const uint32_t* __restrict__ count,
                                                       float4* __restrict__ out_rgba, uint32_t* __restrict__ out_count) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npix; i += gridDim.x * kBlock) {
    const uint32_t g = pix[i];
    out_rgba[i] = rgba[g];
    out_count[i] = count[g];
  }
}
__global__ __launch_bounds__(kBlock) void k_layer_unpack_add(const uint32_t* __restrict__ pix, uint32_t npix,
                                                             const float4* __restrict__ in_rgba, const uint32_t* __restrict__ in_count,
                                                             float4* __restrict__ rgba, uint32_t* __restrict__ count) {
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < npix; i += gridDim.x * kBlock) {
    const uint32_t g = pix[i];
    const float4 a = rgba[g], b = in_rgba[i];
    rgba[g] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    count[g] += in_count[i];
  }
}

// ------------------------------------------------------------------ test hooks: Raytracer::FirstHitTrace1 / AnyHit1
// Same persistent phase-voting traversal as the render path, fed from a caller-supplied ray array.
__device__ __forceinline__ HookHit hook_result(const DScene& sc, V3 o, V3 d, const Hit& h) {
  HookHit r;  // TraceResult defaults (raytracer.h:9-17)
  r.ng[0] = 1.f, r.ng[1] = 0.f, r.ng[2] = 0.f, r.t = 1.f, r.u = 0.f, r.v = 0.f;
  r.instance_id = r.geom_id = r.prim_id = kNone;
  if (h.slot != kNone) {
    Surface s = make_surface(sc, o, d, h);
    const ShadeRec& sr = sc.shade[h.slot & kHitSlotMask];
    r.ng[0] = s.n_g.x, r.ng[1] = s.n_g.y, r.ng[2] = s.n_g.z;
    r.t = h.t, r.u = h.u, r.v = h.v;
    r.instance_id = sr.instance_id, r.geom_id = sr.geom_id, r.prim_id = sr.prim_id;
  }
  return r;
}
struct HookSink {
  static constexpr bool kWalk = false;
  const DScene& sc;
  const float4* rays;
  HookHit* hits;
  uint8_t* occ;
  __device__ __forceinline__ bool load(uint32_t idx, uint32_t& tag, V3& o, V3& d, float& tmin, float& tmax) const {
    tag = idx;
    float4 o4 = rays[2 * idx], d4 = rays[2 * idx + 1];
    o = ld3(o4), d = ld3(d4), tmin = o4.w, tmax = fminf(d4.w, INFINITY);  // raytracer_impl.cc:256
    return occ != nullptr;
  }
  __device__ __forceinline__ void done(uint32_t i, const Hit& h, bool occluded) const {
    if (occ) occ[i] = occluded ? 1 : 0;
    else hits[i] = hook_result(sc, ld3(rays[2 * i]), ld3(rays[2 * i + 1]), h);
  }
};
template <bool ANY, bool CURVES, bool WIDE>  // CURVES / WIDE: the variant k_trace runs for this scene (the Q tree, with or without curves)
__global__ __launch_bounds__(kBlock) void k_hook_pv(DScene sc, const float4* __restrict__ rays, uint32_t n, HookHit* hits,
                                                    uint8_t* occ, uint32_t* counts, uint32_t* spill) {
  __shared__ uint32_t stk[pv_lds_stack<CURVES, WIDE>() * kBlock];
  __shared__ float frm[CURVES ? 10 * kBlock : 1];
  TravStats st = {};
  uint32_t overflow = 0u;
  HookSink sink = {sc, rays, hits, occ};
  trace_pv<ANY ? 1 : 0, false, CURVES, WIDE>(sc, n, &counts[kCntHead], sink, stk + threadIdx.x, kBlock,
                             spill + blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock, st, &overflow, CURVES ? frm + threadIdx.x : nullptr);
  if (overflow) counts[kCntOverflow] = 1u;
}
// One ray per thread, plain stack traversal (dtrace.h): an independent second implementation, selected with
// PBRHIP_SIMPLE_TRAVERSAL=1, that must agree with the production traversal bit for bit.
template <bool CURVES, bool WIDE>
__global__ __launch_bounds__(kBlock) void k_hook_closest(DScene sc, const float4* __restrict__ rays, uint32_t n,
                                                         HookHit* __restrict__ out, uint32_t* overflow_flag, uint32_t* spill) {
  __shared__ uint32_t stk[kSimpleLdsStack * kBlock];
  TravStats st = {};
  uint32_t overflow = 0u;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float4 o4 = rays[2 * i], d4 = rays[2 * i + 1];
    Hit h;
    traverse<false, false, CURVES, WIDE>(sc, ld3(o4), ld3(d4), o4.w, fminf(d4.w, INFINITY), h, stk + threadIdx.x, kBlock, st, &overflow,
                                         spill + blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock);
    out[i] = hook_result(sc, ld3(o4), ld3(d4), h);
  }
  if (overflow) *overflow_flag = 1u;
}
template <bool CURVES, bool WIDE>
__global__ __launch_bounds__(kBlock) void k_hook_any(DScene sc, const float4* __restrict__ rays, uint32_t n,
                                                     uint8_t* __restrict__ out, uint32_t* overflow_flag, uint32_t* spill) {
  __shared__ uint32_t stk[kSimpleLdsStack * kBlock];
  TravStats st = {};
  uint32_t overflow = 0u;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    float4 o4 = rays[2 * i], d4 = rays[2 * i + 1];
    Hit h;
    out[i] = traverse<true, false, CURVES, WIDE>(sc, ld3(o4), ld3(d4), o4.w, fminf(d4.w, INFINITY), h, stk + threadIdx.x, kBlock, st,
                                                 &overflow, spill + blockIdx.x * kBlock + threadIdx.x, gridDim.x * kBlock)
                 ? 1
                 : 0;
  }
  if (overflow) *overflow_flag = 1u;
}

// One ray per QUAD of lanes (dtrace_quad.h), selected with PBRHIP_QUAD=1 for triangle-only Q trees: must agree bit for bit.
template <bool ANY>
__global__ __launch_bounds__(kBlock) void k_hook_quad(DScene sc, const float4* __restrict__ rays, uint32_t n, HookHit* __restrict__ hits,
                                                      uint8_t* __restrict__ occ, uint32_t* overflow_flag, uint32_t* spill) {
  __shared__ uint32_t stk[kSimpleLdsStack * (kBlock / 4)];
  uint32_t overflow = 0u;
  const uint32_t quad = threadIdx.x >> 2, quads = kBlock / 4;
  for (uint32_t i = blockIdx.x * quads + quad; i < n; i += gridDim.x * quads) {
    float4 o4 = rays[2 * i], d4 = rays[2 * i + 1];
    Hit h;
    const bool o = traverse_quad<ANY ? 1 : 0>(sc, ld3(o4), ld3(d4), o4.w, fminf(d4.w, INFINITY), h, stk + quad, quads, &overflow, ANY,
                                             spill + blockIdx.x * quads + quad, gridDim.x * quads);
    if ((threadIdx.x & 3u) == 0u) {
      if (ANY) occ[i] = o ? 1 : 0;
      else hits[i] = hook_result(sc, ld3(o4), ld3(d4), h);
    }
  }
  if (overflow) *overflow_flag = 1u;
}

// queue flip between iterations: counts[In] = counts[Out]; the per-iteration counters restart at 0
// ... and the host is told (ring: four words of pinned host memory, the stamp last): it sizes the launches after the next from these
__global__ void k_advance(uint32_t* counts, uint32_t* heads, uint32_t* ring, uint32_t stamp) {
  if (threadIdx.x == 0) {
    counts[kCntIn] = counts[kCntOut];
    counts[kCntShadowIn] = counts[kCntShadow];
    if (ring) {
      ring[0] = counts[kCntOut], ring[1] = counts[kCntShadow], ring[2] = counts[kCntOverflow];
      __threadfence_system();
      __hip_atomic_store(&ring[3], stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    counts[kCntOut] = 0, counts[kCntPrincipled] = 0, counts[kCntHair] = 0, counts[kCntSss] = 0, counts[kCntShadow] = 0;
    counts[kCntHead] = 0, counts[kCntWalkHead] = 0;
    for (uint32_t h = 0; h < kTraceHeads; h++) heads[h * kHeadStride] = 0;
  }
}

// ------------------------------------------------------------------ launchers
// the Q tree serves the scenes whose tree was built on the host (PBRHIP_WIDE=0: never; read per launch)
static inline bool use_wide(const DScene& sc) {
  const char* e = getenv("PBRHIP_WIDE");
  return sc.wide != nullptr && !(e && atoi(e) == 0);
}
// launches KERNEL<..., CURVES, WIDE> for this scene: (curves, binary), (no curves, binary), (curves, Q), (no curves, Q)
#define PB_LAUNCH_TRAV(KERNEL, PRE, curves, wide, ...)                                          \
  do {                                                                                          \
    if ((wide) && (curves)) hipLaunchKernelGGL((KERNEL<PRE, true, true>), __VA_ARGS__);         \
    else if (wide) hipLaunchKernelGGL((KERNEL<PRE, false, true>), __VA_ARGS__);                 \
    else if (curves) hipLaunchKernelGGL((KERNEL<PRE, true, false>), __VA_ARGS__);               \
    else hipLaunchKernelGGL((KERNEL<PRE, false, false>), __VA_ARGS__);                          \
  } while (0)
bool trace_uses_wide(const DScene& sc) { return use_wide(sc); }
#ifndef PB_TRACE_SMALL1_RAYS
#define PB_TRACE_SMALL1_RAYS 16000000u  // launches of at most this many rays (upper bound): PB_TRACE_SMALL1_BLOCKS blocks per CU
#define PB_TRACE_SMALL1_BLOCKS 4u
#define PB_TRACE_SMALL2_RAYS 4000000u   // ... and of at most this many: PB_TRACE_SMALL2_BLOCKS
#define PB_TRACE_SMALL2_BLOCKS 3u
#endif
#ifndef PB_QUAD_RAYS
#define PB_QUAD_RAYS 0u  // k_trace launches of at most this many rays run on k_trace_quad (PBRHIP_QUAD_RAYS overrides)
#endif
static inline uint32_t quad_rays() {
  const char* e = getenv("PBRHIP_QUAD_RAYS");
  return e ? (uint32_t)strtoul(e, nullptr, 10) : PB_QUAD_RAYS;
}
static inline bool use_quad() {
  const char* e = getenv("PBRHIP_QUAD");
  return e && e[0] == '1';
}
static inline uint32_t grid_for(uint32_t n, uint32_t cap) {
  uint32_t g = (n + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  return g < cap ? g : cap;
}
static inline uint32_t quad_grid(uint32_t n) {  // one ray per quad of lanes: 64 rays per block
  const uint32_t g = (n + 63u) / 64u;
  return g < 1u ? 1u : (g < 2048u ? g : 2048u);
}

void launch_generate(hipStream_t s, const PathState& P, uint32_t npaths) {
  hipLaunchKernelGGL(k_generate, dim3(grid_for(npaths, 8192)), dim3(kBlock), 0, s, P, npaths);
}
void launch_trace(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, bool stats) {
  // Persistent kernel: at most the resident set.  A launch with fewer rays than that would fill gets fewer waves, so that
  // every wave still has a queue to refill its lanes from (kRaysPerWave rays each): a wave that starts with one ray per
  // lane and nothing to refill from runs until its longest ray ends with most lanes idle, and with seven such waves per
  // SIMD the launch is bound by the instructions those mostly empty waves issue.
  const char* e = getenv("PBRHIP_RAYS_PER_WAVE");  // (tuning knob; read per launch)
  const uint32_t rays_per_wave = e ? (uint32_t)strtoul(e, nullptr, 10) : 4u;
  uint32_t blocks = (n_upper + 4u * rays_per_wave - 1u) / (4u * rays_per_wave);
  const bool curves = sc.num_curves != 0;
  const bool wide = use_wide(sc);
  uint32_t cap = 256u * trace_blocks_per_cu(curves, wide);
  if (const char* b = getenv("PBRHIP_TRACE_BLOCKS")) {  // (tuning knob, read per launch: resident blocks per CU, at most the kernel's)
    const uint32_t k = (uint32_t)strtoul(b, nullptr, 10);
    if (k >= 1u && 256u * k < cap) cap = 256u * k;
  }
  if (const char* b = getenv("PBRHIP_TRACE_BLOCKS_SMALL")) {  // "k,n": k blocks per CU for launches of at most n rays ("0,0": none)
    char* end = nullptr;
    const uint32_t k = (uint32_t)strtoul(b, &end, 10);
    const uint32_t lim = (end && *end == ',') ? (uint32_t)strtoul(end + 1, nullptr, 10) : 0u;
    if (k >= 1u && n_upper <= lim && 256u * k < cap) cap = 256u * k;
  } else if (wide && !curves) {
    // Fewer resident blocks for the launches that do not fill the chip for long (round 4, after the shading kernels got faster;
    // scripts/sched_ab.py, rank 0's share of the C2 frame): their drain -- every wave waiting for its longest ray -- runs at
    // fewer waves per SIMD, and the other path group's kernels find room on the CUs.
    if (n_upper <= PB_TRACE_SMALL2_RAYS) cap = std::min(cap, 256u * PB_TRACE_SMALL2_BLOCKS);
    else if (n_upper <= PB_TRACE_SMALL1_RAYS) cap = std::min(cap, 256u * PB_TRACE_SMALL1_BLOCKS);
  }
  if (P.first) {
    // a group's first launch: camera rays only, computed by the sink (TraceSinkT<.., FIRST>); always the phase-voting kernel
    if (trace_first_less(stats, curves)) cap = std::max(cap, 512u) - 256u;  // (its launch bounds: one block per CU fewer; never below one block per CU -- PBRHIP_TRACE_BLOCKS=1 used to make this 0)
    dim3 g(blocks < 1u ? 1u : (blocks < cap ? blocks : cap));
#define PB_LAUNCH_FIRST(ST)                                                                                      \
  do {                                                                                                           \
    if (wide && curves) hipLaunchKernelGGL((k_trace<ST, true, true, true>), g, dim3(kBlock), 0, s, P, sc);       \
    else if (wide) hipLaunchKernelGGL((k_trace<ST, false, true, true>), g, dim3(kBlock), 0, s, P, sc);           \
    else if (curves) hipLaunchKernelGGL((k_trace<ST, true, false, true>), g, dim3(kBlock), 0, s, P, sc);         \
    else hipLaunchKernelGGL((k_trace<ST, false, false, true>), g, dim3(kBlock), 0, s, P, sc);                    \
  } while (0)
    if (stats) PB_LAUNCH_FIRST(true);
    else PB_LAUNCH_FIRST(false);
#undef PB_LAUNCH_FIRST
    return;
  }
  if (wide && !curves && n_upper <= quad_rays()) {
    // a small launch: one ray per quad of lanes
    const dim3 gq(quad_grid(n_upper));
    if (stats) hipLaunchKernelGGL((k_trace_quad<true>), gq, dim3(kBlock), 0, s, P, sc);
    else hipLaunchKernelGGL((k_trace_quad<false>), gq, dim3(kBlock), 0, s, P, sc);
    return;
  }
  dim3 g(blocks < 1u ? 1u : (blocks < cap ? blocks : cap));
  if (stats) PB_LAUNCH_TRAV(k_trace, true, curves, wide, g, dim3(kBlock), 0, s, P, sc);
  else PB_LAUNCH_TRAV(k_trace, false, curves, wide, g, dim3(kBlock), 0, s, P, sc);
}
static inline uint32_t tiles_grid(uint32_t n_upper, int items_per_thread) {
  const uint32_t tile = (uint32_t)items_per_thread * kBlock;
  uint32_t g = (n_upper + tile - 1) / tile;
  return g < 1 ? 1 : (g < kShadeGridCap ? g : kShadeGridCap);
}
void launch_classify(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper) {
  hipLaunchKernelGGL(k_classify, dim3(tiles_grid(n_upper, kClassifyItems)), dim3(kBlock), 0, s, P, sc);
}
void launch_compact(hipStream_t s, const PathState& P, uint32_t n_upper) {
  hipLaunchKernelGGL(k_compact, dim3(tiles_grid(n_upper, kCompactItems)), dim3(kBlock), 0, s, P);
}
void launch_shade_principled(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, uint64_t rng_inc, bool media, bool textured) {
  const dim3 g(grid_for(n_upper, kShadeGridCap));
  if (textured) hipLaunchKernelGGL(k_shade_principled<kShadeFull>, g, dim3(kBlock), 0, s, P, sc, rng_inc);
  else if (media) hipLaunchKernelGGL(k_shade_principled<kShadeMedia>, g, dim3(kBlock), 0, s, P, sc, rng_inc);
  else hipLaunchKernelGGL(k_shade_principled<kShadePlain>, g, dim3(kBlock), 0, s, P, sc, rng_inc);
}
void launch_shade_hair(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, uint64_t rng_inc) {
  hipLaunchKernelGGL(k_shade_hair, dim3(grid_for(n_upper, kShadeGridCap)), dim3(kBlock), 0, s, P, sc, rng_inc);
}
void launch_sss_step(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, uint64_t rng_inc) {
  hipLaunchKernelGGL(k_sss_step, dim3(grid_for(n_upper, kShadeGridCap)), dim3(kBlock), 0, s, P, sc, rng_inc);
}
void launch_sss_walk(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, uint64_t rng_inc, bool stats) {
  const bool curves = sc.num_curves != 0;
  const char* ww = getenv("PBRHIP_WIDE_WALK");
  const bool wide = use_wide(sc) && !(ww && atoi(ww) == 0);
  // persistent: the resident blocks (<= kTraceGridCap: the walk shares k_trace's spill area)
  const uint32_t cap = 256u * (stats ? (uint32_t)PB_WALK_WAVES : walk_blocks_per_cu(curves, wide));
  const uint32_t blocks = (n_upper + 15u) / 16u;
  dim3 g(blocks < 1u ? 1u : (blocks < cap ? blocks : cap));
  if (stats) PB_LAUNCH_TRAV(k_sss_walk, true, curves, wide, g, dim3(kBlock), 0, s, P, sc, rng_inc);
  else PB_LAUNCH_TRAV(k_sss_walk, false, curves, wide, g, dim3(kBlock), 0, s, P, sc, rng_inc);
}
void launch_accumulate(hipStream_t s, const PathState& P, const uint32_t* pix_index, uint32_t npix, uint32_t npass,
                       float* rgba, uint32_t* count) {
  hipLaunchKernelGGL(k_accumulate, dim3(grid_for(npix, 8192)), dim3(kBlock), 0, s, P, pix_index, npix, npass, rgba, count);
}
#ifndef PB_TAIL_BLOCKS
#define PB_TAIL_BLOCKS 768u  // 3 blocks per CU (40 KB LDS stack each); A/B on C2: 256 -> 6.4 ms, 512 -> 5.4, 768 -> 4.8, 1024 -> 5.5
#endif
static_assert((size_t)(kStackDepth - kSimpleLdsStack) * PB_TAIL_BLOCKS * 256 <= kSpillWords, "k_tail's spill area (one stack per thread of its grid)");
static_assert(trace_blocks_per_cu(false, true) * 256u <= kTraceGridCap && trace_blocks_per_cu(true, true) * 256u <= kTraceGridCap,
              "the Q tree's k_trace grids fit the spill area sized by kTraceGridCap");
#define PB_COMMA ,
void launch_tail(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, uint64_t rng_inc, bool stats, bool media, bool textured) {
  uint32_t blocks = (n_upper + 3u) / 4u;  // one path per wave while that fits, at most 2 blocks per CU
  dim3 g(blocks < 1u ? 1u : (blocks < PB_TAIL_BLOCKS ? blocks : PB_TAIL_BLOCKS));
  const bool curves = sc.num_curves != 0;
  const bool wide = use_wide(sc);
  if (stats) PB_LAUNCH_TRAV(k_tail, kShadeFull PB_COMMA true, curves, wide, g, dim3(kBlock), 0, s, P, sc, rng_inc);
  else if (textured) PB_LAUNCH_TRAV(k_tail, kShadeFull PB_COMMA false, curves, wide, g, dim3(kBlock), 0, s, P, sc, rng_inc);
  else if (media) PB_LAUNCH_TRAV(k_tail, kShadeMedia PB_COMMA false, curves, wide, g, dim3(kBlock), 0, s, P, sc, rng_inc);
  else PB_LAUNCH_TRAV(k_tail, kShadePlain PB_COMMA false, curves, wide, g, dim3(kBlock), 0, s, P, sc, rng_inc);
}
void launch_layer_pack(hipStream_t s, const uint32_t* pix, uint32_t npix, const float* rgba, const uint32_t* count, float* shard) {
  if (!npix) return;
  hipLaunchKernelGGL(k_layer_pack, dim3(grid_for(npix, 8192)), dim3(kBlock), 0, s, pix, npix, reinterpret_cast<const float4*>(rgba),
                     count, reinterpret_cast<float4*>(shard), reinterpret_cast<uint32_t*>(shard + 4 * (size_t)npix));
}
void launch_layer_unpack_add(hipStream_t s, const uint32_t* pix, uint32_t npix, const float* shard, float* rgba, uint32_t* count) {
  if (!npix) return;
  hipLaunchKernelGGL(k_layer_unpack_add, dim3(grid_for(npix, 8192)), dim3(kBlock), 0, s, pix, npix,
                     reinterpret_cast<const float4*>(shard), reinterpret_cast<const uint32_t*>(shard + 4 * (size_t)npix),
                     reinterpret_cast<float4*>(rgba), count);
}
void launch_advance(hipStream_t s, const PathState& P, uint32_t* ring_slot, uint32_t stamp) { hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, s, P.counts, P.heads, ring_slot, stamp); }
// counts: kCntNum zeroed words (queue head + overflow flag); spill: traversal-stack spill area
void launch_hook_closest(hipStream_t s, const DScene& sc, const float4* rays, uint32_t n, HookHit* out, uint32_t* counts,
                         uint32_t* spill, bool simple) {
  // the variant of the traversal the render of this scene runs; the binary tree's hooks always carry the curve code
  const bool wide = use_wide(sc), curves = !wide || sc.num_curves != 0;
  if (simple) {
    if (wide && curves) hipLaunchKernelGGL((k_hook_closest<true, true>), dim3(grid_for(n, 4096)), dim3(kBlock), 0, s, sc, rays, n, out, counts + kCntOverflow, spill);
    else if (wide) hipLaunchKernelGGL((k_hook_closest<false, true>), dim3(grid_for(n, 4096)), dim3(kBlock), 0, s, sc, rays, n, out, counts + kCntOverflow, spill);
    else hipLaunchKernelGGL((k_hook_closest<true, false>), dim3(grid_for(n, 4096)), dim3(kBlock), 0, s, sc, rays, n, out, counts + kCntOverflow, spill);
    return;
  }
  if (wide && !curves && use_quad()) {
    hipLaunchKernelGGL((k_hook_quad<false>), dim3(quad_grid(n)), dim3(kBlock), 0, s, sc, rays, n, out, (uint8_t*)nullptr, counts + kCntOverflow, spill);
    return;
  }
  const dim3 g(grid_for(n, kTraceGridCap));
  if (wide && curves) hipLaunchKernelGGL((k_hook_pv<false, true, true>), g, dim3(kBlock), 0, s, sc, rays, n, out, (uint8_t*)nullptr, counts, spill);
  else if (wide) hipLaunchKernelGGL((k_hook_pv<false, false, true>), g, dim3(kBlock), 0, s, sc, rays, n, out, (uint8_t*)nullptr, counts, spill);
  else hipLaunchKernelGGL((k_hook_pv<false, true, false>), g, dim3(kBlock), 0, s, sc, rays, n, out, (uint8_t*)nullptr, counts, spill);
}
void launch_hook_any(hipStream_t s, const DScene& sc, const float4* rays, uint32_t n, uint8_t* out, uint32_t* counts,
                     uint32_t* spill, bool simple) {
  const bool wide = use_wide(sc), curves = !wide || sc.num_curves != 0;
  if (simple) {
    if (wide && curves) hipLaunchKernelGGL((k_hook_any<true, true>), dim3(grid_for(n, 4096)), dim3(kBlock), 0, s, sc, rays, n, out, counts + kCntOverflow, spill);
    else if (wide) hipLaunchKernelGGL((k_hook_any<false, true>), dim3(grid_for(n, 4096)), dim3(kBlock), 0, s, sc, rays, n, out, counts + kCntOverflow, spill);
    else hipLaunchKernelGGL((k_hook_any<true, false>), dim3(grid_for(n, 4096)), dim3(kBlock), 0, s, sc, rays, n, out, counts + kCntOverflow, spill);
    return;
  }
  if (wide && !curves && use_quad()) {
    hipLaunchKernelGGL((k_hook_quad<true>), dim3(quad_grid(n)), dim3(kBlock), 0, s, sc, rays, n, (HookHit*)nullptr, out, counts + kCntOverflow, spill);
    return;
  }
  const dim3 g(grid_for(n, kTraceGridCap));
  if (wide && curves) hipLaunchKernelGGL((k_hook_pv<true, true, true>), g, dim3(kBlock), 0, s, sc, rays, n, (HookHit*)nullptr, out, counts, spill);
  else if (wide) hipLaunchKernelGGL((k_hook_pv<true, false, true>), g, dim3(kBlock), 0, s, sc, rays, n, (HookHit*)nullptr, out, counts, spill);
  else hipLaunchKernelGGL((k_hook_pv<true, true, false>), g, dim3(kBlock), 0, s, sc, rays, n, (HookHit*)nullptr, out, counts, spill);
}

}  // namespace pb
