// dtrace.h -- ray/primitive tests and BVH2 traversal (device).
//
// Stands in for Embree's rtcIntersect1 / rtcOccluded1 behind pbrlab's Raytracer facade
// (src/raytracer/raytracer_impl.cc:268-287).  Intersection contract:
//   * a primitive hit is accepted for  tmin < t <= tmax  if t also lies inside the interval in which the ray crosses the
//     primitive's OWN box (hit_inside): a degenerate sliver can pass the triangle test with a meaningless distance far from
//     the sliver; with this rule the accepted set is a property of (ray, primitive) alone
//   * closest hit = smallest accepted t; equal t -> smaller canonical primitive id (instance, geom, prim)
//     => the answer is independent of BVH shape and traversal order (every stored box contains the validation boxes of the
//     primitives below it, and the slab arithmetic is monotone in the box)
//   * any-hit = "some primitive has an accepted hit"
//   * boxes are tested conservatively (interval widened by 2^-16 relative)
#pragma once

#include "dscene.h"

namespace pb {

struct Hit {
  float t, u, v;
  uint32_t slot;  // kNone = miss
};

struct TravStats {
  uint32_t nodes, tris, curves;     // closest-hit rays
  uint32_t anodes, atris, acurves;  // any-hit (shadow) rays
  // phase-voting traversal, lane 0 of each wave: iterations and participating lanes per phase
  uint32_t it_node, it_tri, it_curve, it_refill, ln_node, ln_tri, ln_curve;
  // per ray: steps (node visits + primitive tests) in power-of-two buckets (<= 16, 32, ... 1024, more) and the maximum
  uint32_t hist[8], max_steps;
  uint32_t ahist[8], amax_steps;  // the same for any-hit (shadow) rays
  uint32_t refill_ticks;          // 100 MHz ticks the wave spent in refills (lane 0)
  unsigned long long cyc[4];      // shader-clock cycles (s_memtime) of the wave's loop turns by what the turn did: node, triangle, curve phase, refill (lane 0)
  uint32_t suspended, suspended_any;  // closest-hit / any-hit rays this lane suspended (dtrace_pv.h)
  unsigned long long t_exhausted; // 100 MHz clock when this wave found the ray queue empty (0: never; always recorded, once per wave: PBRHIP_WAVE_LOG)
};

__device__ __forceinline__ V3 ld3(const float4& a) { return V3(a.x, a.y, a.z); }

// The validation of the intersection contract: lo / hi = the primitive's own box (a triangle's corners; a ribbon piece's end
// points widened by the larger end radius), widened here by 2^-17 relative + 1e-31 -- strictly inside what BvhNode::widen_*
// stores for any box that contains it, strictly outside the geometry -- and t has to lie in the ray's interval through it
// (widened by 2^-16 relative with the function box_test2 / box_test4 use).  Operation for operation the checker's hit_inside.
__device__ __forceinline__ float vbox_lo(float v) { return __builtin_fmaf(-fabsf(v), 7.62939453125e-06f, v) - 1e-31f; }
__device__ __forceinline__ float vbox_hi(float v) { return __builtin_fmaf(fabsf(v), 7.62939453125e-06f, v) + 1e-31f; }
__device__ __forceinline__ bool hit_inside(V3 lo, V3 hi, V3 o, V3 inv, float t) {
  float t0 = (vbox_lo(lo.x) - o.x) * inv.x, t1 = (vbox_hi(hi.x) - o.x) * inv.x;
  float a = __builtin_fminf(t0, t1), b = __builtin_fmaxf(t0, t1);
  t0 = (vbox_lo(lo.y) - o.y) * inv.y, t1 = (vbox_hi(hi.y) - o.y) * inv.y;
  a = __builtin_fmaxf(a, __builtin_fminf(t0, t1)), b = __builtin_fminf(b, __builtin_fmaxf(t0, t1));
  t0 = (vbox_lo(lo.z) - o.z) * inv.z, t1 = (vbox_hi(hi.z) - o.z) * inv.z;
  a = __builtin_fmaxf(a, __builtin_fminf(t0, t1)), b = __builtin_fminf(b, __builtin_fmaxf(t0, t1));
  const float e = 1.52587890625e-05f;
  a = __builtin_fmaf(-fabsf(a), e, a), b = __builtin_fmaf(fabsf(b), e, b);
  return a <= t && t <= b;
}

// Moeller-Trumbore; u,v are the barycentrics of v1,v2 (what Lerp3 expects, pbrlab_math.h:35-38); inv = 1 / d
__device__ __forceinline__ bool tri_test(V3 v0, V3 v1, V3 v2, V3 o, V3 d, V3 inv3, float tmin, float& t, float& u, float& v) {
  V3 e1 = v1 - v0, e2 = v2 - v0;
  V3 p = cross(d, e2);
  float det = dot(e1, p);
  if (!(det != 0.0f)) return false;
  float inv = 1.0f / det;
  V3 s = o - v0;
  float uu = dot(s, p) * inv;
  if (!(uu >= 0.0f && uu <= 1.0f)) return false;
  V3 q = cross(s, e1);
  float vv = dot(d, q) * inv;
  if (!(vv >= 0.0f && uu + vv <= 1.0f)) return false;
  float tt = dot(e2, q) * inv;
  if (!(tt > tmin)) return false;
  const V3 lo(__builtin_fminf(__builtin_fminf(v0.x, v1.x), v2.x), __builtin_fminf(__builtin_fminf(v0.y, v1.y), v2.y),
              __builtin_fminf(__builtin_fminf(v0.z, v1.z), v2.z));
  const V3 hi(__builtin_fmaxf(__builtin_fmaxf(v0.x, v1.x), v2.x), __builtin_fmaxf(__builtin_fmaxf(v0.y, v1.y), v2.y),
              __builtin_fmaxf(__builtin_fmaxf(v0.z, v1.z), v2.z));
  if (!hit_inside(lo, hi, o, inv3, tt)) return false;
  t = tt, u = uu, v = vv;
  return true;
}

// Two triangles at once on packed fp32 (v_pk_*): the triangle leaves of the Q tree hold one or two triangles, stored
// interleaved (TriPair, dscene.h: every pair of words = the same coordinate of triangle a and of triangle b), so that after the
// 16-byte loads the two values sit in an aligned register pair and every +, -, x of tri_test / hit_inside is ONE instruction for
// both.  Per triangle the operations and their order are tri_test's, so each half returns tri_test's bits; nothing is skipped on
// a failed condition (in SIMT an early exit saves nothing unless the whole wave takes it).  ok[i]: triangle i has an accepted hit
// in (tmin, +inf) -- the caller compares with the ray's tmax.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 f2s(float a) { return f2{a, a}; }
// (scalar fmas on purpose: v_fma_f32 takes |x| as an input modifier and issues at full rate, 2.7 cycles; the packed form needs a v_and
// per value for the |x| and a v_pk_fma_f32 at 4.7 per two -- scripts/ubench/valu_rate2.hip)
__device__ __forceinline__ f2 vbox_lo2(f2 x) { return f2{__builtin_fmaf(-fabsf(x.x), 7.62939453125e-06f, x.x), __builtin_fmaf(-fabsf(x.y), 7.62939453125e-06f, x.y)} - f2s(1e-31f); }  // vbox_lo, two at once
__device__ __forceinline__ f2 vbox_hi2(f2 x) { return f2{__builtin_fmaf(fabsf(x.x), 7.62939453125e-06f, x.x), __builtin_fmaf(fabsf(x.y), 7.62939453125e-06f, x.y)} + f2s(1e-31f); }
// (the ray comes as nine scalars: a V3 handed over by value survives as a 12-byte stack object here -- the splats below defeat
// its scalar replacement -- which the backend then parks in LDS, 3 KB per block)
__device__ __forceinline__ void tri_test_pair(const float4& w0, const float4& w1, const float4& w2, const float4& w3, const float4& w4, float o_x,
                                              float o_y, float o_z, float d_x, float d_y, float d_z, float i_x, float i_y, float i_z, float tmin,
                                              bool& ok_a, bool& ok_b, f2& t, f2& u, f2& v) {
  const f2 v0x = {w0.x, w0.y}, v0y = {w0.z, w0.w}, v0z = {w1.x, w1.y};
  const f2 v1x = {w1.z, w1.w}, v1y = {w2.x, w2.y}, v1z = {w2.z, w2.w};
  const f2 v2x = {w3.x, w3.y}, v2y = {w3.z, w3.w}, v2z = {w4.x, w4.y};
  const f2 dx = f2s(d_x), dy = f2s(d_y), dz = f2s(d_z), ox = f2s(o_x), oy = f2s(o_y), oz = f2s(o_z);
  const f2 e1x = v1x - v0x, e1y = v1y - v0y, e1z = v1z - v0z;
  const f2 e2x = v2x - v0x, e2y = v2y - v0y, e2z = v2z - v0z;
  const f2 px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;  // cross(d, e2)
  const f2 det = e1x * px + e1y * py + e1z * pz;
  const f2 inv = {1.0f / det.x, 1.0f / det.y};
  const f2 sx = ox - v0x, sy = oy - v0y, sz = oz - v0z;
  const f2 uu = (sx * px + sy * py + sz * pz) * inv;
  const f2 qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;  // cross(s, e1)
  const f2 vv = (dx * qx + dy * qy + dz * qz) * inv;
  const f2 tt = (e2x * qx + e2y * qy + e2z * qz) * inv;
  const f2 uv = uu + vv;
  // hit_inside on the triangles' own boxes
  const f2 lox = __builtin_elementwise_min(__builtin_elementwise_min(v0x, v1x), v2x), loy = __builtin_elementwise_min(__builtin_elementwise_min(v0y, v1y), v2y),
           loz = __builtin_elementwise_min(__builtin_elementwise_min(v0z, v1z), v2z);
  const f2 hix = __builtin_elementwise_max(__builtin_elementwise_max(v0x, v1x), v2x), hiy = __builtin_elementwise_max(__builtin_elementwise_max(v0y, v1y), v2y),
           hiz = __builtin_elementwise_max(__builtin_elementwise_max(v0z, v1z), v2z);
  const f2 e = f2s(1.52587890625e-05f);
  f2 t0 = (vbox_lo2(lox) - ox) * f2s(i_x), t1 = (vbox_hi2(hix) - ox) * f2s(i_x);
  f2 a = __builtin_elementwise_min(t0, t1), b = __builtin_elementwise_max(t0, t1);
  t0 = (vbox_lo2(loy) - oy) * f2s(i_y), t1 = (vbox_hi2(hiy) - oy) * f2s(i_y);
  a = __builtin_elementwise_max(a, __builtin_elementwise_min(t0, t1)), b = __builtin_elementwise_min(b, __builtin_elementwise_max(t0, t1));
  t0 = (vbox_lo2(loz) - oz) * f2s(i_z), t1 = (vbox_hi2(hiz) - oz) * f2s(i_z);
  a = __builtin_elementwise_max(a, __builtin_elementwise_min(t0, t1)), b = __builtin_elementwise_min(b, __builtin_elementwise_max(t0, t1));
  a = f2{__builtin_fmaf(-fabsf(a.x), e.x, a.x), __builtin_fmaf(-fabsf(a.y), e.x, a.y)}, b = f2{__builtin_fmaf(fabsf(b.x), e.x, b.x), __builtin_fmaf(fabsf(b.y), e.x, b.y)};
#ifdef PB_DIAG_NO_VALIDATE  // (diagnostic build: what the validation costs -- NOT the contract)
  ok_a = det.x != 0.0f && uu.x >= 0.0f && uu.x <= 1.0f && vv.x >= 0.0f && uv.x <= 1.0f && tt.x > tmin;
  ok_b = det.y != 0.0f && uu.y >= 0.0f && uu.y <= 1.0f && vv.y >= 0.0f && uv.y <= 1.0f && tt.y > tmin;
#else
  ok_a = det.x != 0.0f && uu.x >= 0.0f && uu.x <= 1.0f && vv.x >= 0.0f && uv.x <= 1.0f && tt.x > tmin && a.x <= tt.x && tt.x <= b.x;
  ok_b = det.y != 0.0f && uu.y >= 0.0f && uu.y <= 1.0f && vv.y >= 0.0f && uv.y <= 1.0f && tt.y > tmin && a.y <= tt.y && tt.y <= b.y;
#endif
  t = tt, u = uu, v = vv;
}
// one triangle leaf of the Q tree (the five words of its TriPair) against a ray whose current interval ends at hit.t with hit
// `hit` (slot == kNone: none yet): the reference order -- triangle a, then triangle b -- of the accept rule "t <= tmax, equal
// distances go to the smaller canonical id".  Returns true when the ray is an any-hit ray and a triangle was accepted.
__device__ __forceinline__ uint32_t q_gid(const DScene& sc, uint32_t code);
template <bool ANY_CT, bool STATS>
__device__ __forceinline__ bool tri_pair_accept_s(const DScene& sc, const float4& w0, const float4& w1, const float4& w2, const float4& w3, const float4& w4,
                                                  float o_x, float o_y, float o_z, float d_x, float d_y, float d_z, float i_x, float i_y, float i_z,
                                                  float tmin, bool any_rt, Hit& hit, uint32_t& ntested);
template <bool ANY_CT, bool STATS>
__device__ __forceinline__ bool tri_pair_accept(const DScene& sc, const float4& w0, const float4& w1, const float4& w2, const float4& w3, const float4& w4,
                                                V3 o, V3 d, V3 inv3, float tmin, bool any_rt, Hit& hit, uint32_t& ntested) {
  return tri_pair_accept_s<ANY_CT, STATS>(sc, w0, w1, w2, w3, w4, o.x, o.y, o.z, d.x, d.y, d.z, inv3.x, inv3.y, inv3.z, tmin, any_rt, hit, ntested);
}
// (scalar ray parameters: a V3 passed by value into the packed test survives as a 12-byte stack object)
template <bool ANY_CT, bool STATS>
__device__ __forceinline__ bool tri_pair_accept_s(const DScene& sc, const float4& w0, const float4& w1, const float4& w2, const float4& w3, const float4& w4,
                                                  float o_x, float o_y, float o_z, float d_x, float d_y, float d_z, float i_x, float i_y, float i_z,
                                                  float tmin, bool any_rt, Hit& hit, uint32_t& ntested) {
  bool ok_a, ok_b;
  f2 t, u, v;
  tri_test_pair(w0, w1, w2, w3, w4, o_x, o_y, o_z, d_x, d_y, d_z, i_x, i_y, i_z, tmin, ok_a, ok_b, t, u, v);
  const uint32_t code_a = __float_as_uint(w4.z), code_b = __float_as_uint(w4.w);
  ok_b = ok_b && code_b != kNone;  // (a leaf of one triangle stores it twice: the second copy is not a candidate)
  if (STATS) ntested += code_b != kNone ? 2u : 1u;
  const bool any_ray = ANY_CT || any_rt;
  bool acc_a = ok_a && t.x <= hit.t;
  if (!any_ray) {
    if (acc_a && t.x == hit.t && hit.slot != kNone) acc_a = q_gid(sc, code_a) < q_gid(sc, hit.slot);  // tie: the smaller canonical id wins
    if (acc_a) hit.t = t.x, hit.u = u.x, hit.v = v.x, hit.slot = code_a;
  }
  bool acc_b = ok_b && t.y <= hit.t;
  if (!any_ray) {
    if (acc_b && t.y == hit.t && hit.slot != kNone) acc_b = q_gid(sc, code_b) < q_gid(sc, hit.slot);
    if (acc_b) hit.t = t.y, hit.u = u.y, hit.v = v.y, hit.slot = code_b;
  }
  return any_ray && (acc_a || acc_b);
}

// dP/du of the cubic: what Embree reports as Ng for flat curves (hair-shader.cc:165-166 uses it as tangent)
__device__ __forceinline__ V3 bezier_tangent(const float4 cp[4], float u) {
  float s = 1.0f - u;
  float c0 = 3.0f * s * s, c1 = 6.0f * u * s, c2 = 3.0f * u * u;
  V3 p0 = ld3(cp[0]), p1 = ld3(cp[1]), p2 = ld3(cp[2]), p3 = ld3(cp[3]);
  return (p1 - p0) * c0 + (p2 - p1) * c1 + (p3 - p2) * c2;
}

// Ray-facing flat ribbon, 4 linear pieces per cubic (RTC_GEOMETRY_TYPE_FLAT_BEZIER_CURVE, raytracer_impl.cc:158-159);
// every piece is its own traversal primitive: a = B(i/4), b = B((i+1)/4) (xyz + radius, evaluated at commit).
// u = curve parameter, v in [-1,1] across the width.
// The part of the test that depends on the ray alone: unit direction, the two axes of the ray-facing plane, 1/|d|.  The
// phase-voting traversal computes it once per ray (at the refill) and keeps it in LDS; every piece test of that ray reads it.
struct RayFrame {
  V3 dn, bx, by;
  float inv_len;
};
__device__ __forceinline__ RayFrame ray_frame(V3 d) {
  RayFrame f;
  f.inv_len = 1.0f / sqrtf(dot(d, d));
  f.dn = d * f.inv_len;
  branchless_onb(f.dn, f.bx, f.by);
  return f;
}
__device__ __forceinline__ bool segment_test(const float4& a, const float4& b, uint32_t i, V3 o, const RayFrame& f, V3 inv3, float tmin,
                                             float tmax, float& t, float& u, float& v);
__device__ __forceinline__ bool segment_test(const float4& a, const float4& b, uint32_t i, V3 o, V3 d, V3 inv3, float tmin, float tmax,
                                             float& t, float& u, float& v) {
  return segment_test(a, b, i, o, ray_frame(d), inv3, tmin, tmax, t, u, v);
}
// ... in two steps, so that a leaf of two NEIGHBOURING pieces (points a b c: pieces a-b and b-c) projects its middle point once:
// segment_project = a point in the ray's frame (x, y across the ray, z along it); segment_core = the test on two projected points.
// The expressions are segment_test's own, so each piece gets segment_test's bits.
__device__ __forceinline__ V3 segment_project(const float4& a, V3 o, const RayFrame& f) {
  const V3 ra = V3(a.x, a.y, a.z) - o;
  return V3(dot(ra, f.bx), dot(ra, f.by), dot(ra, f.dn));
}
__device__ __forceinline__ bool segment_core(const float4& a, const float4& b, V3 pa, V3 pb, uint32_t i, V3 o, float inv_len, V3 inv3, float tmin, float tmax,
                                             float& t, float& u, float& v) {
  const float pxa = pa.x, pya = pa.y, pza = pa.z, pxb = pb.x, pyb = pb.y, pzb = pb.z;
  float ex = pxb - pxa, ey = pyb - pya;
  float len2 = ex * ex + ey * ey;
  if (!(len2 > 0.0f)) return false;
  float s = -(pxa * ex + pya * ey) / len2;
  if (!(s >= 0.0f && s <= 1.0f)) return false;
  float dist = (ey * pxa - ex * pya) / sqrtf(len2);
  float r = a.w + s * (b.w - a.w);
  if (!(r > 0.0f && fabsf(dist) <= r)) return false;
  float tt = (pza + s * (pzb - pza)) * inv_len;
  if (!(tt > tmin && tt <= tmax)) return false;
  const float rm = __builtin_fmaxf(fabsf(a.w), fabsf(b.w));
  const V3 lo(__builtin_fminf(a.x, b.x) - rm, __builtin_fminf(a.y, b.y) - rm, __builtin_fminf(a.z, b.z) - rm);
  const V3 hi(__builtin_fmaxf(a.x, b.x) + rm, __builtin_fmaxf(a.y, b.y) + rm, __builtin_fmaxf(a.z, b.z) + rm);
  if (!hit_inside(lo, hi, o, inv3, tt)) return false;
  t = tt, u = ((float)i + s) * 0.25f, v = dist / r;
  return true;
}
__device__ __forceinline__ bool segment_test(const float4& a, const float4& b, uint32_t i, V3 o, const RayFrame& f, V3 inv3, float tmin,
                                             float tmax, float& t, float& u, float& v) {
  const float inv_len = f.inv_len;
  const V3 dn = f.dn, bx = f.bx, by = f.by;
  V3 ra = V3(a.x, a.y, a.z) - o, rb = V3(b.x, b.y, b.z) - o;
  float pxa = dot(ra, bx), pya = dot(ra, by), pza = dot(ra, dn);
  float pxb = dot(rb, bx), pyb = dot(rb, by), pzb = dot(rb, dn);
  float ex = pxb - pxa, ey = pyb - pya;
  float len2 = ex * ex + ey * ey;
  if (!(len2 > 0.0f)) return false;
  float s = -(pxa * ex + pya * ey) / len2;
  if (!(s >= 0.0f && s <= 1.0f)) return false;
  float dist = (ey * pxa - ex * pya) / sqrtf(len2);
  float r = a.w + s * (b.w - a.w);
  if (!(r > 0.0f && fabsf(dist) <= r)) return false;
  float tt = (pza + s * (pzb - pza)) * inv_len;
  if (!(tt > tmin && tt <= tmax)) return false;
  const float rm = __builtin_fmaxf(fabsf(a.w), fabsf(b.w));
  const V3 lo(__builtin_fminf(a.x, b.x) - rm, __builtin_fminf(a.y, b.y) - rm, __builtin_fminf(a.z, b.z) - rm);
  const V3 hi(__builtin_fmaxf(a.x, b.x) + rm, __builtin_fmaxf(a.y, b.y) + rm, __builtin_fmaxf(a.z, b.z) + rm);
  if (!hit_inside(lo, hi, o, inv3, tt)) return false;
  t = tt, u = ((float)i + s) * 0.25f, v = dist / r;
  return true;
}

// Conservative slab test of BOTH children of a node against [tmin, tmax] (entry distances in t0, t1).  n0..n2 are the
// first three 16-byte words of a BvhNode: (lo.x pair, lo.y pair), (lo.z pair, hi.x pair), (hi.y pair, hi.z pair), each
// pair = (child 0, child 1).  The interval is widened by 2^-16 relative; the test only has to be conservative (hits do
// not depend on which boxes are visited), so the widening may use fma.
__device__ __forceinline__ void box_test2(const float4& n0, const float4& n1, const float4& n2, V3 o, V3 inv, float tmin,
                                          float tmax, bool& h0, bool& h1, float& t0, float& t1) {
  const f2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
  const f2 ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
  const f2 lx = {n0.x, n0.y}, ly = {n0.z, n0.w}, lz = {n1.x, n1.y};
  const f2 hx = {n1.z, n1.w}, hy = {n2.x, n2.y}, hz = {n2.z, n2.w};
  f2 p = (lx - ox) * ix, q = (hx - ox) * ix;
  f2 a = __builtin_elementwise_min(p, q), b = __builtin_elementwise_max(p, q);
  p = (ly - oy) * iy, q = (hy - oy) * iy;
  a = __builtin_elementwise_max(a, __builtin_elementwise_min(p, q)), b = __builtin_elementwise_min(b, __builtin_elementwise_max(p, q));
  p = (lz - oz) * iz, q = (hz - oz) * iz;
  a = __builtin_elementwise_max(a, __builtin_elementwise_min(p, q)), b = __builtin_elementwise_min(b, __builtin_elementwise_max(p, q));
  const float e = 1.52587890625e-05f;
  const float a0 = __builtin_fmaf(-fabsf(a.x), e, a.x), a1 = __builtin_fmaf(-fabsf(a.y), e, a.y);
  const float b0 = __builtin_fmaf(fabsf(b.x), e, b.x), b1 = __builtin_fmaf(fabsf(b.y), e, b.y);
  t0 = a0, t1 = a1;
  h0 = a0 <= b0 && b0 >= tmin && a0 <= tmax;
  h1 = a1 <= b1 && b1 >= tmin && a1 <= tmax;
}

// The slab test on the four quantised boxes of a wide node (QNode, dscene.h).  w0 = (org.x, org.y, org.z, s.x), w1 = (s.y, s.z,
// qlo.x, qlo.y), w2 = (qlo.z, qhi.x, qhi.y, qhi.z); byte i of a q word belongs to child i.  A bound is rebuilt as
// fma(q, s, org) (v_cvt_f32_ubyte + v_pk_fma_f32) -- the builder has checked with this very expression that the result
// encloses the binary tree's box -- and from there on the arithmetic is box_test2's, operation for operation: the test is
// monotone in the box like the binary tree's, rays parallel to an axis included.
// ta / tb: entry / exit distance of each box (widened like box_test2's); the caller compares them with the ray's interval.
__device__ __forceinline__ void box_test4q(const float4& w0, const float4& w1, const float4& w2, V3 o, const float4& inv, float ta[4],
                                           float tb[4]) {
  const float e = 1.52587890625e-05f;
  // Round 5: which of a box's two bounds on an axis the ray meets first is a property of the ray alone -- the sign bit of 1 / d
  // (of d: -0 included) --, so the entry bounds of all four children are SELECTED as whole words (four bytes = four children) before
  // the conversion, and per box the entry distance is one max3 and the exit distance one min3 over the axes instead of
  // min(p, q), max(p, q) per axis and then the combination: 8 instead of 32 min / max instructions per node, for 9 selects.  p <= q
  // holds after rounding whenever neither is NaN (lo <= hi, the same factor: rounding is monotone), so the distances are the
  // ones box_test2 computes; where a product is NaN (0 x inf: the ray is parallel to the axis and its origin lies ON the bound)
  // box_test2 keeps the other bound's value for both and this form drops the axis' entry or exit constraint: a weaker test, i.e.
  // conservative, which is all a box test has to be (hits do not depend on which boxes are visited).
  const bool nx = (int)__float_as_uint(inv.x) < 0, ny = (int)__float_as_uint(inv.y) < 0, nz = (int)__float_as_uint(inv.z) < 0;
  const uint32_t qlx = __float_as_uint(w1.z), qly = __float_as_uint(w1.w), qlz = __float_as_uint(w2.x);
  const uint32_t qhx = __float_as_uint(w2.y), qhy = __float_as_uint(w2.z), qhz = __float_as_uint(w2.w);
  const uint32_t lx = nx ? qhx : qlx, ly = ny ? qhy : qly, lz = nz ? qhz : qlz;  // the bound the ray enters through
  const uint32_t hx = nx ? qlx : qhx, hy = ny ? qly : qhy, hz = nz ? qlz : qhz;  // ... and leaves through
  const f2 sx = {w0.w, w0.w}, sy = {w1.x, w1.x}, sz = {w1.y, w1.y}, gx = {w0.x, w0.x}, gy = {w0.y, w0.y}, gz = {w0.z, w0.z};
  const f2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z}, ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
#pragma unroll
  for (int h = 0; h < 2; h++) {
    auto two = [h](uint32_t w) { return h ? f2{(float)((w >> 16) & 255u), (float)(w >> 24)} : f2{(float)(w & 255u), (float)((w >> 8) & 255u)}; };
    const f2 px = (__builtin_elementwise_fma(two(lx), sx, gx) - ox) * ix, qx = (__builtin_elementwise_fma(two(hx), sx, gx) - ox) * ix;
    const f2 py = (__builtin_elementwise_fma(two(ly), sy, gy) - oy) * iy, qy = (__builtin_elementwise_fma(two(hy), sy, gy) - oy) * iy;
    const f2 pz = (__builtin_elementwise_fma(two(lz), sz, gz) - oz) * iz, qz = (__builtin_elementwise_fma(two(hz), sz, gz) - oz) * iz;
    const float a0 = __builtin_fmaxf(__builtin_fmaxf(px.x, py.x), pz.x), a1 = __builtin_fmaxf(__builtin_fmaxf(px.y, py.y), pz.y);
    const float b0 = __builtin_fminf(__builtin_fminf(qx.x, qy.x), qz.x), b1 = __builtin_fminf(__builtin_fminf(qx.y, qy.y), qz.y);
    ta[2 * h] = __builtin_fmaf(-fabsf(a0), e, a0), ta[2 * h + 1] = __builtin_fmaf(-fabsf(a1), e, a1);
    tb[2 * h] = __builtin_fmaf(fabsf(b0), e, b0), tb[2 * h + 1] = __builtin_fmaf(fabsf(b1), e, b1);
  }
}

// One step at a wide node: the children the ray's interval [tmin, tmax] hits, nearest first.  k[0..3] ascending; a key is the
// child's entry distance (>= 0 as an integer; its two low bits hold the child index, wide_ref), 0xFFFFFFFF = not hit.
constexpr uint32_t kWideMiss = 0xFFFFFFFFu;
__device__ __forceinline__ void wide_node_keys(const float4& w0, const float4& w1, const float4& w2, const float4& refs, V3 o,
                                               const float4& inv, float tmin, float tmax, uint32_t k[4]) {
  float ta[4], tb[4];
  box_test4q(w0, w1, w2, o, inv, ta, tb);
  const uint32_t r[4] = {__float_as_uint(refs.x), __float_as_uint(refs.y), __float_as_uint(refs.z), __float_as_uint(refs.w)};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const bool h = ta[i] <= tb[i] && tb[i] >= tmin && ta[i] <= tmax && r[i] != kEmptyChild;
    const int bits = (int)__float_as_uint(ta[i]);
    k[i] = h ? ((uint32_t)(bits < 0 ? 0 : bits) & ~3u) | (uint32_t)i : kWideMiss;
  }
  auto cex = [](uint32_t& a, uint32_t& b) {
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo, b = hi;
  };
  cex(k[0], k[1]), cex(k[2], k[3]), cex(k[0], k[2]), cex(k[1], k[3]), cex(k[1], k[2]);
}
__device__ __forceinline__ uint32_t wide_ref(const float4& refs, uint32_t key) {
  const uint32_t a = __float_as_uint((key & 1u) ? refs.y : refs.x), b = __float_as_uint((key & 1u) ? refs.w : refs.z);
  return (key & 2u) ? b : a;
}

// canonical id of a hit held by a traversal of the Q tree (ties between equal distances go to the smaller one): a triangle's
// hit code carries its slot, a curve hit is held as kQPointHit | point
__device__ __forceinline__ uint32_t q_gid(const DScene& sc, uint32_t code) {
  const uint32_t c = (code & kQPointHit) ? sc.q_hitcode[code & ~kQPointHit] : code;
  return sc.shade[c & kHitSlotMask].gid;
}
// the hit code the other stages see
__device__ __forceinline__ uint32_t q_final_code(const DScene& sc, uint32_t code) {
  return (code != kNone && (code & kQPointHit)) ? sc.q_hitcode[code & ~kQPointHit] : code;
}

// Leaf processing.  any: returns true on the first accepted hit.
// MODE: 0 = closest hit, 1 = any hit, 2 = per lane (`any_rt`), as in trace_pv.
template <int MODE, bool STATS, bool CURVES, bool WIDE = false>
__device__ __forceinline__ bool leaf_test(const DScene& sc, uint32_t leaf, V3 o, V3 d, V3 inv, float tmin, float& best_t,
                                          Hit& hit, TravStats& st, bool any_rt) {
  const bool ANY = MODE == 2 ? any_rt : (MODE == 1);
  uint32_t first = (leaf & 0x3FFFFFFFu) >> 3, count = (leaf & 7u) + 1u;
  bool is_curve = (leaf & kCurveBit) != 0;
  if (WIDE && !CURVES) {  // Q tree of a triangle-only scene: one TriPair per leaf (dscene.h), both triangles in one packed test
    const float4* g = sc.wide + sc.q_tri0 + (size_t)first * kTriPairWords;
    const float4 w0 = g[0], w1 = g[1], w2 = g[2], w3 = g[3], w4 = g[4];
    uint32_t nt = 0u;
    const bool occluded = tri_pair_accept<MODE == 1, STATS>(sc, w0, w1, w2, w3, w4, o, d, inv, tmin, ANY, hit, nt);
    if (STATS) st.tris += nt;
    best_t = hit.t;
    return occluded;
  }
  // a curve record of the Q tree (dscene.h; PB_CURVE_RECORDS): one or two pieces at points P, P + 2; the low bits of `first` and of the
  // reference are their indices in their cubics
  const bool rec = PB_CURVE_RECORDS && WIDE && CURVES && is_curve;
  const uint32_t sub_a = first & 3u, sub_b = leaf & 3u;
  if (rec) first &= ~3u, count = (leaf & kCurvePairBit) ? 2u : 1u;
  for (uint32_t k = 0; k < count; k++) {
    const uint32_t s = first + (rec ? 2u * k : k);
    float t, u, v;
    bool ok;
    uint32_t code;
    if (WIDE) {  // Q tree of a scene with curves: 48-byte triangle slots / curve records or chains of curve points (dscene.h)
      if (!is_curve) {
        const float4* g = sc.wide + sc.q_tri0 + (size_t)s * 3;
        float4 a = g[0], b = g[1], c = g[2];
        if (STATS) st.tris++;
        ok = tri_test(ld3(a), ld3(b), ld3(c), o, d, inv, tmin, t, u, v) && (t <= best_t);
        code = __float_as_uint(c.w);
      } else {
        const float4* g = sc.wide + sc.q_pt0 + s;
        float4 a = g[0], b = g[1];
        if (STATS) st.curves++;
        ok = segment_test(a, b, rec ? (k ? sub_b : sub_a) : (s & 3u), o, d, inv, tmin, best_t, t, u, v);
        code = kQPointHit | s;
      }
    } else {
      const float4* g = sc.slots + (size_t)s * 4;
      float4 a = g[0], b = g[1], c = g[2];
      if (!CURVES || !is_curve) {
        if (STATS) st.tris++;
        ok = tri_test(ld3(a), ld3(b), ld3(c), o, d, inv, tmin, t, u, v) && (t <= best_t);
      } else {
        if (STATS) st.curves++;
        ok = segment_test(a, b, __float_as_uint(c.x), o, d, inv, tmin, best_t, t, u, v);
      }
      code = s | __float_as_uint(c.w);  // + routing bits (dscene.h)
    }
    if (!ok) continue;
    if (ANY) return true;
    if (t == best_t && hit.slot != kNone) {  // tie: the smaller canonical primitive id wins
      const bool less = WIDE ? q_gid(sc, code) < q_gid(sc, hit.slot) : sc.shade[s].gid < sc.shade[hit.slot & kHitSlotMask].gid;
      if (!less) continue;
    }
    best_t = t;
    hit.t = t, hit.u = u, hit.v = v, hit.slot = code;
  }
  return false;
}

// BVH2 traversal, near child first, far child on a per-lane stack (stack[i * stride]).
// WIDE: the 4-wide tree (sc.wide must not be null): nearest hit child next, the others pushed farthest first.
template <int MODE, bool STATS, bool CURVES, bool WIDE = false>
__device__ __forceinline__ bool traverse_mode(const DScene& sc, V3 o, V3 d, float tmin, float tmax, Hit& hit,
                                              uint32_t* stack, uint32_t stride, TravStats& st, uint32_t* overflow, bool any_rt,
                                              uint32_t* spill, uint32_t spill_stride) {
  // stack: entries 0 .. kSimpleLdsStack-1 (stack[i * stride], LDS); deeper ones in spill[(i - kSimpleLdsStack) * spill_stride]
  auto push = [&](int& sp, uint32_t v) {
    if (sp < kSimpleLdsStack) stack[(uint32_t)sp * stride] = v, sp++;
    else if (sp < kStackDepth) spill[(uint32_t)(sp - kSimpleLdsStack) * spill_stride] = v, sp++;
    else *overflow = 1u;
  };
  auto pop = [&](int& sp) {
    sp--;
    return sp < kSimpleLdsStack ? stack[(uint32_t)sp * stride] : spill[(uint32_t)(sp - kSimpleLdsStack) * spill_stride];
  };
  hit.slot = kNone;
  hit.t = tmax, hit.u = 0.f, hit.v = 0.f;
  if (sc.num_nodes == 0) return false;
  V3 inv(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  float best_t = tmax;
  int sp = 0;
  uint32_t cur = 0;
  for (;;) {
    // cur is an internal node
    if (WIDE) {
      const float4* np = sc.wide + (size_t)cur * 4u;  // (a wide node's reference is its 64-byte item index)
      const float4 w0 = np[0], w1 = np[1], w2 = np[2], refs = np[3];
      if (STATS) st.nodes++;
      uint32_t k[4];
      wide_node_keys(w0, w1, w2, refs, o, make_float4(inv.x, inv.y, inv.z, 0.f), tmin, best_t, k);
#pragma unroll
      for (int j = 3; j >= 1; j--) {
        if (k[j] == kWideMiss) continue;
        push(sp, wide_ref(refs, k[j]));
      }
      uint32_t next = k[0] == kWideMiss ? kEmptyChild : wide_ref(refs, k[0]);
      for (;;) {
        if (next == kEmptyChild) {
          if (sp == 0) {
            hit.slot = q_final_code(sc, hit.slot);
            return hit.slot != kNone;
          }
          next = pop(sp);
        }
        if (!(next & kLeafBit)) break;
        if (leaf_test<MODE, STATS, CURVES, true>(sc, next, o, d, inv, tmin, best_t, hit, st, any_rt)) return true;
        next = kEmptyChild;
      }
      cur = next;
      continue;
    }
    const float4* np = reinterpret_cast<const float4*>(sc.nodes + cur);
    float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    if (STATS) st.nodes++;
    uint32_t c0 = __float_as_uint(n3.x), c1 = __float_as_uint(n3.y);
    float t0, t1;
    bool h0, h1;
    box_test2(n0, n1, n2, o, inv, tmin, best_t, h0, h1, t0, t1);
    uint32_t next = kEmptyChild;
    if (h0 && h1) {
      uint32_t nearc = c0, farc = c1;
      if (t1 < t0) nearc = c1, farc = c0;
      push(sp, farc);
      next = nearc;
    } else if (h0) {
      next = c0;
    } else if (h1) {
      next = c1;
    }
    for (;;) {
      if (next == kEmptyChild) {
        if (sp == 0) return hit.slot != kNone;
        next = pop(sp);
      }
      if (!(next & kLeafBit)) break;
      if (leaf_test<MODE, STATS, CURVES>(sc, next, o, d, inv, tmin, best_t, hit, st, any_rt)) return true;
      next = kEmptyChild;
    }
    cur = next;
  }
}
template <bool ANY, bool STATS, bool CURVES, bool WIDE = false>
__device__ __forceinline__ bool traverse(const DScene& sc, V3 o, V3 d, float tmin, float tmax, Hit& hit,
                                         uint32_t* stack, uint32_t stride, TravStats& st, uint32_t* overflow, uint32_t* spill,
                                         uint32_t spill_stride) {
  return traverse_mode<ANY ? 1 : 0, STATS, CURVES, WIDE>(sc, o, d, tmin, tmax, hit, stack, stride, st, overflow, ANY, spill, spill_stride);
}

}  // namespace pb
