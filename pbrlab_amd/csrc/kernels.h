// kernels.h -- path-state layout in HBM and the kernel launch interface (host <-> kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dscene.h"

namespace pb {

// Per-path state, 16-byte words, grouped into records (below).
// N = paths of one chunk (pixels of this rank's tiles x passes in the chunk); slot = pass_local*npix + pixel.
//   ray_o (org.xyz, tmin) | ray_d (dir.xyz, tmax) | thr (throughput.rgb, bsdf pdf) | rng (PCG32 state, u64)   = rec, 64 B
//   hit (t, u, v, slot bits) | L (contribution.rgb, -)                                              arrays of their own
//   random-walk state, touched only by paths inside a medium (ssrec 64 B + sss_A):
//     sss_sigt (sigma_t.rgb) | sss_sigs (sigma_s.rgb, entry instance id) | sss_thr (walk throughput, step index)
//     sss_ez (entry frame normal) | sss_A (resolved first NEE)
//   queues (u32 path slots): q_in/q_out (ping-pong), q_principled, q_hair, q_sss, q_shadow; the shadow-ray payload
//     sits at the path's own slot: sh_d (dir, tmax) | sh_c (contribution if visible, mode) | sh_e (contribution if occluded:
//     medium exits only); origin and tmin are ray_o's (the shadow ray starts where the continuation ray starts)
// Path state in HBM.  After the first bounce the live paths are sparse in slot space (132.7 M camera paths of a C2 frame, 104 M
// later bounces in all): what a path-sized access costs then is the number of LINES it touches, not its bytes.  The words a kernel
// reads and writes together therefore share a record: rec (64 B per path) = ray origin + tmin | direction + tmax | throughput + pdf |
// generator state -- one line for the shading kernels' five loads and four stores, one for k_trace's ray; srec (32 B) = a shadow
// ray's direction + tmax | its pending contribution; ssrec (64 B) = what a random walk carries (sigma_t, sigma_s + entry instance,
// throughput + step index, entry frame).  hit and L stay arrays of their own: k_trace writes / k_accumulate reads
// them densely.  The members keep their names: PathView<T, STRIDE>::operator[] is the indexing every kernel already does.
template <typename T, int STRIDE>
struct PathView {
  T* base;
  __host__ __device__ __forceinline__ T& operator[](size_t i) const { return base[i * STRIDE]; }
};
struct PathState {
  PathView<float4, 4> ray_o, ray_d, thr;  // rec + 0 / 1 / 2
  PathView<uint64_t, 8> rng;              // rec + 3 (8 of its 16 bytes)
  float4 *L, *hit;
  PathView<float4, 4> sss_sigt, sss_sigs, sss_thr, sss_ez;  // the random walk's record (64 B per path): one line per walk start / step
  float4* sss_A;
  uint32_t *q_in, *q_out, *q_principled, *q_hair, *q_sss, *q_shadow, *q_shadow_in;
  PathView<float4, 2> sh_d, sh_c;         // srec + 0 / 1
  float4* sh_e;
  uint32_t* spill;               // traversal-stack spill area: (kStackDepth - LDS part) x resident threads
  uint32_t* counts;              // kCnt*
  uint32_t* heads;               // the kTraceHeads heads of k_trace's ray queue, kHeadStride words apart (reset by k_advance)
  unsigned long long* stats;     // kStat*; null unless the render collects statistics
  unsigned long long* wave_log;  // debugging (PBRHIP_WAVE_LOG): per k_trace launch and wave: start, end (100 MHz clock), rays, loop turns
  uint32_t wave_log_launch;      // index of this launch in wave_log
  // First bounce of a chunk: every path still has the camera position as its origin (tmin 0), throughput (1,1,1) and pdf 0,
  // so k_generate does not store ray_o / thr and the first trace and shading do not load them; the only per-path flag
  // ("not the first bounce", MIS weight of emission) is the same bit.
  float cam_org[3];
  uint32_t first;   // 1: a group's first bounce
  uint32_t direct;  // 1: k_classify is skipped -- every hit of this bounce takes the principled shader (a first bounce without hair; any bounce of
                    // a scene without hair and media when PBRHIP_DIRECT is set) --: k_shade_principled walks the trace queue (first bounce:
                    // the paths slot0 .. slot0 + n) itself and applies k_classify's drop rule
  // ... and the camera sample of path slot0 + j itself is a function of j (render.cc:160-171: pixel = pix_index[j % npix], pass =
  // first_pass + j / npix, two draws of the sample's own generator): the first k_trace and the first shading compute it
  // (kernels.hip::camera_sample) instead of reading a stored direction, generator state and queue entry (round 4: k_generate only
  // clears the radiance; 44 -> 16 bytes written and 44 fewer read per path)
  Camera cam;
  const uint32_t* pix_index;
  uint32_t npix, width, first_pass, slot0;
  // Path order inside a group (round 5).  pass_run = R >= 1: runs of R passes of ONE pixel are adjacent -- path j of the group is pixel
  // pix_index[(j / R) % npix], pass first_pass + (j / R / npix) * R + j % R.  R = 1: a wave's 64 paths are an 8 x 8 pixel patch of one pass
  // (scenes of surfaces: the order does not matter to them); scenes with curves: R = the largest power of two <= 64 that divides the
  // group's passes -- thin geometry likes a pixel's samples in one wave (hair k_trace -3 to -5 %).  A permutation: images do not change.
  uint32_t pass_run;
  uint64_t seed_seq;
  uint32_t no_medium;  // no material of the scene can enter a medium: every shadow ray is an ordinary one (kShNormal), so an occluded one has nothing to deliver
  // Resumable rays (round 6; dtrace_pv.h).  Every k_trace launch used to end with a drain of ~0.35 ms in which the chip ran nearly
  // empty while a few long rays finished.  Now a wave that has found the queue empty keeps going for susp_turns more loop turns and
  // then SUSPENDS the closest-hit rays it still holds: the traversal state (stack, current item, the hit held so far: one
  // kSuspWords-word record, susp_out) is stored, the path's hit record becomes (record index, -, -, kHitSuspended), the path goes
  // to no shading queue in this iteration (k_classify / the direct first shading re-queue it with kQResume) and the ray RESUMES at
  // the head of the next iteration's k_trace (susp_in = this launch's susp_out), inside that launch's bulk phase.  A suspended ray
  // loses nothing and its hit does not change (the traversal continues where it stopped), so images are bit-identical; the path
  // just skips one shading round.  The bounded rays of paths inside a medium are never suspended.  A SHADOW ray (scenes without media
  // only: no_medium) is suspended too: its payload sits at the path's own slot, which the path's next shading would overwrite, and its
  // contribution has to reach L before that shading's (float sums do not commute) -- so the suspending lane sets hold[p], re-queues the
  // ray itself (one atomic per wave: q_shadow) and the path's next shading, finding hold[p] set, does nothing but hand the path back
  // (result word kRHold -> trace-queue entry with kQHold: its closest-hit ray is NOT traced again, its hit record stands); the launch that
  // finishes the shadow ray clears hold[p].  susp_turns = 0: never (the launch before k_tail, the hooks).
  uint32_t susp_turns;             // drain turns before suspension (0: never)
  uint32_t* susp_out;              // this launch's records: one per resident thread of the launch (kSuspRecords)
  const uint32_t* susp_in;         // the previous launch's
  uint32_t shadow_first;           // k_trace takes the shadow rays of the previous bounce before this bounce's closest-hit rays
  PathView<uint4, 4> rng4;         // rec + 3 as one 16-byte word: generator state (x, y) | hold (z) | -
  PathView<uint32_t, 16> hold;     // rec + 3, third word: 1 while the path's shadow ray is suspended -- the path's next shading waits for it (above)
};
constexpr uint32_t kSuspWords = 72;  // hit (4) | cur, state | rem << 8 | sp << 16, -, - (4) | stack (kStackDepth = 64)
static_assert(kStackDepth <= 64, "a suspend record holds the whole traversal stack");

enum : uint32_t { kFlagNotFirst = 1u };
// queue entry = path slot | in-medium bit | "the Russian roulette at the head of this path's next shading fails" bit
constexpr uint32_t kQSssBit = 0x80000000u, kQDoomed = 0x40000000u, kQPathMask = 0x0FFFFFFFu;
// ... | "the path's pending shading is its FIRST bounce" (a camera ray that was suspended: its shading runs in a later iteration, next
// to other paths' later bounces) | "the path's ray is a suspended one: resume it" (trace queue only)
constexpr uint32_t kQFirst = 1u << 29, kQResume = 1u << 28;
// in a scene WITHOUT media bit 31 of a trace-queue entry means: the path is HELD -- its closest-hit ray is traced already (the hit record
// stands), it waits for its suspended shadow ray (PathState::hold)
constexpr uint32_t kQHold = kQSssBit;
constexpr uint32_t kHitSuspended = 0xFFFFFFFEu;  // hit code of a path whose closest-hit ray was suspended (kNone = 0xFFFFFFFF: a miss)
// shade-kernel result word (written over the kernel's own queue entry): path slot (28 bits) | flags
constexpr uint32_t kRPathMask = 0x0FFFFFFFu, kRShadow = 1u << 28, kRAlive = 1u << 29;
// ... or, for a path whose closest-hit ray was suspended: kRResume WITHOUT kRAlive (no shading produces that: a path inside a medium
// is alive) | kQDoomed of its queue entry | kRResumeFirst = its kQFirst; k_compact re-queues it with kQResume
constexpr uint32_t kRResume = kQSssBit, kRResumeFirst = kRShadow;
// ... or, for a HELD path (its shadow ray is suspended: PathState::hold): kRHold without kRAlive and without kRResume (no shading
// produces that: kQDoomed comes with kRAlive) | kRHoldDoomed = the kQDoomed of its queue entry; k_compact re-queues it with kQHold
constexpr uint32_t kRHold = kQDoomed, kRHoldDoomed = kRShadow;
constexpr uint64_t kMaxPathsInFlight = (1ull << 28) - 1;
enum : uint32_t { kShNormal = 0u, kShSssEntry = 1u, kShSssExit = 2u };
// The counters of one path group.  Round 6: the ones the kernels hit with atomics (one per tile and queue in k_classify / k_compact, one per
// batch in the traversal kernels) each sit in a 256-byte block of their own -- atomics on one cache line serialise (~88 per microsecond per
// line: what the eight heads of k_trace's queue taught, dtrace_pv.h); the others share the first block.  PB_CNT_SPREAD=0: packed, as before.
#ifndef PB_CNT_SPREAD
#define PB_CNT_SPREAD 1
#endif
enum : uint32_t {
  kCntIn = 0, kCntOverflow = 1, kCntShadowIn = 2,
  kCntStride = PB_CNT_SPREAD ? 64 : 1, kCntHot0 = PB_CNT_SPREAD ? 64 : 3,
  kCntOut = kCntHot0, kCntPrincipled = kCntHot0 + kCntStride, kCntHair = kCntHot0 + 2 * kCntStride, kCntSss = kCntHot0 + 3 * kCntStride,
  kCntShadow = kCntHot0 + 4 * kCntStride, kCntHead = kCntHot0 + 5 * kCntStride, kCntWalkHead = kCntHot0 + 6 * kCntStride,
  kCntNum = kCntHot0 + 7 * kCntStride
};
enum : uint32_t {
  kStatClosestRays = 0, kStatClosestNodes, kStatClosestTris, kStatClosestCurves,
  kStatShadowRays, kStatShadowNodes, kStatShadowTris, kStatShadowCurves,
  kStatPvItNode, kStatPvItTri, kStatPvItCurve, kStatPvItRefill, kStatPvLnNode, kStatPvLnTri, kStatPvLnCurve,
  kStatTailClosestRays, kStatTailShadowRays, kStatPrunedRays,
  kStatStepHist0, kStatStepHistLast = kStatStepHist0 + 7, kStatMaxSteps, kStatMaxWaveIters,
  kStatWalkNodes, kStatWalkTris, kStatWalkTurns, kStatWalkSteps,
  kStatAnyHist0, kStatAnyHistLast = kStatAnyHist0 + 7, kStatAnyMaxSteps,
  kStatHeld,  // trace-queue entries of HELD paths (no ray behind them: subtracted from the closest-hit rays)
  kStatSuspendedShadow,  // shadow rays suspended (counted once more as a shadow ray of the launch that resumes each)
  kStatSuspended,  // rays suspended (each is counted once more as a closest-hit ray of the launch that resumes it: subtracted there)
  kStatCycNode, kStatCycTri, kStatCycCurve, kStatCycRefill, kStatWalkCycTrav, kStatWalkCycStep,  // shader-clock cycles of the phase-voting waves' loop turns by what the turn did (lane 0 of every wave)
  kStatNum
};

// Resident 256-thread blocks per CU of the persistent traversal kernel = waves per SIMD (VGPR budget 512 / waves).
// Binary tree (trees built on the GPU, PBRHIP_WIDE=0), triangle-only scenes: 7 in rounds 1-5 (<= 72 VGPRs; A/B on C2 in round 1: 6 -> 48.4 ms,
// 7 -> 45.2, 8 spills -> 62.3); 6 since round 6 -- with the resume path of the suspended rays in the refill the kernel no longer fits
// 72 registers without scratch in its triangle phase.  Scenes with curves need more registers: 6 (7 spills: C4 254 -> 434 ms).
#ifndef PB_TRACE_BLOCKS
#define PB_TRACE_BLOCKS 6
#endif
#ifndef PB_TRACE_BLOCKS_CURVES
#define PB_TRACE_BLOCKS_CURVES 6
#endif
#ifndef PB_TRACE_BLOCKS_WIDE
#define PB_TRACE_BLOCKS_WIDE 6
#endif
#ifndef PB_TRACE_BLOCKS_WIDE_CURVES
// round 6: with curve records -- both pieces of a leaf in one turn (PB_CURVE_RECORDS, dscene.h) -- the kernel needs 91-93 registers: five
// waves per SIMD (at six, 36 bytes of scratch in the loop: +40 %); the one-piece-per-turn kernel of rounds 3-5 fits six
#define PB_TRACE_BLOCKS_WIDE_CURVES ((PB_CURVE_RECORDS && PB_CURVE_TWO) ? 5 : 6)
#endif
constexpr uint32_t kTraceBlocksPerCUWide = PB_TRACE_BLOCKS_WIDE, kTraceBlocksPerCUWideCurves = PB_TRACE_BLOCKS_WIDE_CURVES;  // the Q tree's kernels
constexpr uint32_t kTraceBlocksPerCU = PB_TRACE_BLOCKS, kTraceBlocksPerCUCurves = PB_TRACE_BLOCKS_CURVES;
constexpr uint32_t trace_blocks_per_cu(bool curves, bool wide) {
  return wide ? (curves ? kTraceBlocksPerCUWideCurves : kTraceBlocksPerCUWide) : (curves ? kTraceBlocksPerCUCurves : kTraceBlocksPerCU);
}
constexpr uint32_t kTraceGridCap = 256 * (kTraceBlocksPerCU > kTraceBlocksPerCUCurves ? kTraceBlocksPerCU : kTraceBlocksPerCUCurves);  // persistent traversal: at most the resident blocks (sizes the spill area)
// words of traversal-stack spill area one path group (or one hook call) needs: every resident thread of the largest traversal
// grid x the entries of its stack that do not live in LDS (the one-ray-per-lane kernels -- k_tail, the simple hooks -- run
// smaller grids: (kStackDepth - kSimpleLdsStack) x 4096 x 256 and x PB_TAIL_BLOCKS x 256)
constexpr size_t kSpillWords = (size_t)kStackDepth * kTraceGridCap * 256;
constexpr size_t kSuspRecords = (size_t)kTraceGridCap * 256;  // suspend records of one k_trace launch: one per resident thread
static_assert((size_t)(kStackDepth - kSimpleLdsStack) * 4096 * 256 <= kSpillWords, "spill area of the one-ray-per-lane hook grids (grid_for(n, 4096))");
constexpr uint32_t kShadeGridCap = 256 * 8;
constexpr int kMaxGroups = 8;
#ifndef PB_TRACE_HEADS
#define PB_TRACE_HEADS 8
#endif
constexpr uint32_t kTraceHeads = PB_TRACE_HEADS;  // heads of k_trace's ray queue: one per XCD (block b draws from head b % 8 first)
constexpr uint32_t kHeadStride = 64;  // words between two heads: each in a 256-byte block of its own (atomics on one cache line serialise: eight heads in ONE line were twice as slow as one head)
constexpr uint32_t kRingSlots = 16;  // iterations of one group the host may have enqueued and not yet heard of
constexpr uint32_t kWaveLogWaves = 8192, kWaveLogLaunches = 64;  // PBRHIP_WAVE_LOG buffer: launches x waves x 4 words  // concurrent path groups (one HIP stream each)

struct HookHit {  // == pbrhip_hit == TraceResult (raytracer.h:9-17)
  float ng[3];
  float t, u, v;
  uint32_t instance_id, geom_id, prim_id;
};

void launch_generate(hipStream_t s, const PathState& P, uint32_t npaths);  // clears the radiance of the group's paths (PathState::slot0 ...)
void launch_trace(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, bool stats);
bool trace_uses_wide(const DScene& sc);  // the traversal kernels walk the 4-wide tree of this scene (now: PBRHIP_WIDE is read per launch)
void launch_tail(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, uint64_t rng_inc, bool stats, bool media, bool textured);
void launch_classify(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper);
void launch_compact(hipStream_t s, const PathState& P, uint32_t n_upper);
void launch_shade_principled(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, uint64_t rng_inc, bool media, bool textured);
void launch_shade_hair(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, uint64_t rng_inc);
void launch_sss_step(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, uint64_t rng_inc);
void launch_sss_walk(hipStream_t s, const PathState& P, const DScene& sc, uint32_t n_upper, uint64_t rng_inc, bool stats);
void launch_accumulate(hipStream_t s, const PathState& P, const uint32_t* pix_index, uint32_t npix, uint32_t npass,
                       float* rgba, uint32_t* count);
void launch_advance(hipStream_t s, const PathState& P, uint32_t* ring_slot, uint32_t stamp);  // (also resets P.heads)  // ring_slot: 4 words of device-visible host memory (or null)
void launch_texture_fetch(hipStream_t s, const DScene& sc, uint32_t tex_id, const float* uv, uint32_t n, float* rgb);  // test hook: Texture::FetchFloat3
void launch_leaf_eval(hipStream_t s, uint32_t op, const float* in, uint32_t n, uint32_t in_words, float* out, uint32_t out_words);  // test hook: the device's leaf functions
// RenderLayer shard of a pixel list: shard = npix x rgba (16 B) followed by npix x count (4 B); 16-byte aligned
void launch_layer_pack(hipStream_t s, const uint32_t* pix, uint32_t npix, const float* rgba, const uint32_t* count, float* shard);
void launch_layer_unpack_add(hipStream_t s, const uint32_t* pix, uint32_t npix, const float* shard, float* rgba, uint32_t* count);
void launch_hook_closest(hipStream_t s, const DScene& sc, const float4* rays, uint32_t n, HookHit* out, uint32_t* counts,
                         uint32_t* spill, bool simple);
void launch_hook_any(hipStream_t s, const DScene& sc, const float4* rays, uint32_t n, uint8_t* out, uint32_t* counts,
                     uint32_t* spill, bool simple);

}  // namespace pb
