/*
 * orc_math.h -- ORACLE (test infrastructure only; never linked into the product).
 *
 * Plain-C restatement of pbrlab's leaf math, following the reference file:line cited at each
 * function.  Every expression keeps the reference's association order: the oracle is built
 * with -ffp-contract=off on baseline x86-64 so that each '*' and '+' rounds exactly once, like
 * the reference built with g++.
 *
 * Transcendental mode (orc_set_math_mode):
 *   ORC_MATH_LIBM    cosf/sinf/expf/logf of the host libm == what the reference calls (std::cos(float) etc.): "the reference's
 *                    arithmetic" on this host.
 *   ORC_MATH_GLIBCF  include/pbr_glibcf.h: the float functions of glibc (Ubuntu 22.04's 2.35 build) on x86-64 with FMA restated with explicit IEEE
 *                    double arithmetic (every one of the 2^32 arguments of each function gives the host libm's bits on this image:
 *                    tests/test_glibcf.py).  The HIP kernels compile the same header (round 5), so GPU-vs-oracle[GLIBCF] is
 *                    bit-exact on any host and GPU-vs-oracle[LIBM] is bit-exact where the libm is that glibc.
 *   ORC_MATH_F64R    the function evaluated in double precision and rounded once to float, by the fixed implementation of
 *                    include/pbr_f64r.h (double +, -, *, /, fma only; within 0.5 ulp + 2^-20 ulp of the true value:
 *                    tests/test_f64r.py): what the kernels compute when built with -DPBR_MATH_F64R (rounds 3-4's default).
 * sqrt and division are IEEE correctly rounded in all modes and on the device.
 */
#ifndef ORC_MATH_H_
#define ORC_MATH_H_

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORC_MATH_LIBM 0
#define ORC_MATH_F64R 1
#define ORC_MATH_GLIBCF 2 /* include/pbr_glibcf.h: glibc's own float functions restated (what the HIP kernels compute since round 5) */

extern int g_orc_math_mode;

/* src/pbrlab_math.h:7-11 */
#define ORC_PI 3.141592653589793f
#define ORC_PI_INV 0.318309886183f
#define ORC_EPS 1e-3f
#define ORC_INF 1.844E18f

#include "../include/pbr_f64r.h"   /* the fixed f64r implementation (correctly rounded), compiled verbatim by the HIP kernels' f64r build too */
#include "../include/pbr_glibcf.h" /* glibc's flt-32 functions restated bit for bit, compiled verbatim by the HIP kernels */

static inline float orc_cosf(float x) { return g_orc_math_mode == ORC_MATH_LIBM ? cosf(x) : (g_orc_math_mode == ORC_MATH_GLIBCF ? glibcf_cosf(x) : f64r_cosf(x)); }
static inline float orc_sinf(float x) { return g_orc_math_mode == ORC_MATH_LIBM ? sinf(x) : (g_orc_math_mode == ORC_MATH_GLIBCF ? glibcf_sinf(x) : f64r_sinf(x)); }
static inline float orc_expf(float x) { return g_orc_math_mode == ORC_MATH_LIBM ? expf(x) : (g_orc_math_mode == ORC_MATH_GLIBCF ? glibcf_expf(x) : f64r_expf(x)); }
static inline float orc_logf(float x) { return g_orc_math_mode == ORC_MATH_LIBM ? logf(x) : (g_orc_math_mode == ORC_MATH_GLIBCF ? glibcf_logf(x) : f64r_logf(x)); }

/* std::max(a,b) / std::min(a,b) exactly as libstdc++ defines them (NaN behaviour included). */
static inline float orc_max(float a, float b) { return (a < b) ? b : a; }
static inline float orc_min(float a, float b) { return (b < a) ? b : a; }
/* src/pbrlab-util.h:9-17 : Clamp(x,a,b) = max(a, min(b, x)) */
static inline float orc_clamp(float x, float a, float b) { return orc_max(a, orc_min(b, x)); }
static inline float orc_saturate(float x) { return orc_clamp(x, 0.0f, 1.0f); }
static inline float orc_sqr(float v) { return v * v; }
/* src/pbrlab_math.h:17 */
static inline float orc_safe_sqrtf(float f) { return sqrtf(orc_max(f, 0.0f)); }

static inline uint32_t orc_f2u(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float orc_u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* ---------------------------------------------------------------- float3 (nanort.h:313-404) */
typedef struct {
  float x, y, z;
} f3;

static inline f3 f3_make(float x, float y, float z) {
  f3 r = {x, y, z};
  return r;
}
static inline f3 f3_set1(float v) { return f3_make(v, v, v); }
static inline f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_mul(f3 a, f3 b) { return f3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 f3_div(f3 a, f3 b) { return f3_make(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline f3 f3_scale(f3 a, float s) { return f3_make(a.x * s, a.y * s, a.z * s); }
/* real3 / float goes through real3(float) and the element-wise operator/ */
static inline f3 f3_divs(f3 a, float s) { return f3_make(a.x / s, a.y / s, a.z / s); }
static inline f3 f3_neg(f3 a) { return f3_make(-a.x, -a.y, -a.z); }
static inline float f3_get(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline float f3_dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 f3_cross(f3 a, f3 b) {
  return f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float f3_length(f3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* nanort.h:379-390 : vectors no longer than FLT_EPSILON are returned unchanged (Q11) */
static inline f3 f3_normalize(f3 a) {
  float len = f3_length(a);
  if (fabsf(len) > FLT_EPSILON) {
    float inv = 1.0f / len;
    a.x *= inv;
    a.y *= inv;
    a.z *= inv;
  }
  return a;
}
/* render.cc:243-249 and raytracer_impl.cc:213-220 : unguarded normalise (Q11) */
static inline f3 f3_normalize_raw(f3 v) {
  float inv = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
  return f3_make(v.x * inv, v.y * inv, v.z * inv);
}
/* src/pbrlab_math.h:30-38 */
static inline f3 f3_lerp(f3 v0, f3 v1, float u) {
  return f3_add(f3_scale(v0, 1.0f - u), f3_scale(v1, u));
}
static inline f3 f3_lerp3(f3 v0, f3 v1, f3 v2, float u, float v) {
  return f3_add(f3_add(f3_scale(v0, 1.0f - u - v), f3_scale(v1, u)), f3_scale(v2, v));
}

/* src/pbrlab-util.h */
static inline float orc_average(f3 c) { return (c.x + c.y + c.z) / 3.f; }
/* std::max({a,b,c}) : first maximum by operator< */
static inline float orc_spectrum_norm(f3 c) {
  float m = c.x;
  if (m < c.y) m = c.y;
  if (m < c.z) m = c.z;
  return m;
}
static inline f3 orc_safe_divide_spectrum(f3 a, f3 b) {
  f3 c;
  c.x = (fabsf(b.x) < FLT_EPSILON) ? 0.0f : a.x / b.x;
  c.y = (fabsf(b.y) < FLT_EPSILON) ? 0.0f : a.y / b.y;
  c.z = (fabsf(b.z) < FLT_EPSILON) ? 0.0f : a.z / b.z;
  return c;
}
static inline float orc_rgb_to_y(f3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }
static inline int orc_is_black(f3 v) { return (fabsf(v.x) + fabsf(v.y) + fabsf(v.z)) < FLT_EPSILON; }
static inline int orc_is_finite3(f3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }

/* ------------------------------------------------------------ PCG32 (src/random/rng.h:17-69) */
typedef struct {
  uint64_t state, inc;
  uint64_t draws; /* oracle-only bookkeeping: number of Draw() calls */
} orc_rng;

static inline uint32_t orc_pcg32_next(orc_rng* r) {
  uint64_t old = r->state;
  r->state = old * 6364136223846793005ULL + r->inc;
  uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = (uint32_t)(old >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}
static inline void orc_rng_seed(orc_rng* r, uint64_t initstate, uint64_t initseq) {
  r->state = 0u;
  r->inc = (initseq << 1u) | 1u;
  r->draws = 0;
  orc_pcg32_next(r);
  r->state += initstate;
  orc_pcg32_next(r);
}
/* rng.h:54-65 : 23 random mantissa bits, [0,1) */
static inline float orc_rng_draw(orc_rng* r) {
  uint32_t u = (orc_pcg32_next(r) >> 9) | 0x3f800000u;
  r->draws++;
  return orc_u2f(u) - 1.0f;
}

/* -------------------------------------------- sampling (src/sampler/sampling-utils.h:10-66) */
static inline f3 orc_cosine_sample_hemisphere(float u1, float u2) {
  float a = u1 * 2.0f * ORC_PI, r = sqrtf(u2);
  return f3_make(orc_cosf(a) * r, orc_sinf(a) * r, sqrtf(orc_max(1.0f - u2, 0.0f)));
}
static inline f3 orc_uniform_sample_sphere(float u1, float u2) {
  float u = 2.0f * u2 - 1.0f;
  float norm = sqrtf(orc_max(0.0f, 1.0f - u * u));
  float theta = 2.0f * ORC_PI * u1;
  return f3_make(norm * orc_cosf(theta), u, norm * orc_sinf(theta));
}
static inline float orc_power_heuristic(float sampled_pdf, float other_pdf) {
  float r, mis;
  if (sampled_pdf > other_pdf) {
    r = other_pdf / sampled_pdf;
    mis = 1 / (1 + r * r);
  } else if (sampled_pdf < other_pdf) {
    r = sampled_pdf / other_pdf;
    mis = 1 - 1 / (1 + r * r);
  } else {
    mis = 0.5f;
  }
  return mis;
}
static inline void orc_triangle_uniform_sampler(float u1, float u2, float* a, float* b) {
  int flag = (u1 > u2);
  float M = flag ? u1 : u2;
  float m = (!flag) ? u1 : u2;
  *a = 1.0f - M;
  *b = M - m;
}

/* -------------------------- OIIO-derived fast math (src/pbrlab_math.h:96-341), explicit fmaf */
static inline float orc_fast_clamp(float x, float a, float b) { return orc_max(a, orc_min(b, x)); }

static inline float orc_fast_reduce_pi(float x, int* q_out) {
  int q = (int)rintf(x * (float)0.31830988618379067154);
  float qf = (float)q;
  x = fmaf(qf, -0.78515625f * 4, x);
  x = fmaf(qf, -0.00024187564849853515625f * 4, x);
  x = fmaf(qf, -3.7747668102383613586e-08f * 4, x);
  x = fmaf(qf, -1.2816720341285448015e-12f * 4, x);
  x = (float)1.57079632679489661923 - ((float)1.57079632679489661923 - x);
  *q_out = q;
  return x;
}
static inline float orc_fast_sin_poly(float x, float s) {
  float u = 2.6083159809786593541503e-06f;
  u = fmaf(u, s, -0.0001981069071916863322258f);
  u = fmaf(u, s, +0.00833307858556509017944336f);
  u = fmaf(u, s, -0.166666597127914428710938f);
  u = fmaf(s, u * x, x);
  return u;
}
static inline float orc_fast_cos_poly(float s) {
  float u = -2.71811842367242206819355e-07f;
  u = fmaf(u, s, +2.47990446951007470488548e-05f);
  u = fmaf(u, s, -0.00138888787478208541870117f);
  u = fmaf(u, s, +0.0416666641831398010253906f);
  u = fmaf(u, s, -0.5f);
  u = fmaf(u, s, +1.0f);
  return u;
}
/* pbrlab_math.h:135-161 */
static inline float orc_fast_sin(float x) {
  int q;
  x = orc_fast_reduce_pi(x, &q);
  float s = x * x;
  if ((q & 1) != 0) x = -x;
  float u = orc_fast_sin_poly(x, s);
  if (fabsf(u) > 1.0f) u = 0.0f;
  return u;
}
/* pbrlab_math.h:163-185 */
static inline float orc_fast_cos(float x) {
  int q;
  x = orc_fast_reduce_pi(x, &q);
  float s = x * x;
  float u = orc_fast_cos_poly(s);
  if ((q & 1) != 0) u = -u;
  if (fabsf(u) > 1.0f) u = 0.0f;
  return u;
}
/* pbrlab_math.h:187-215 */
static inline void orc_fast_sincos(float x, float* sine, float* cosine) {
  int q;
  x = orc_fast_reduce_pi(x, &q);
  float s = x * x;
  if ((q & 1) != 0) x = -x;
  float su = orc_fast_sin_poly(x, s);
  float cu = orc_fast_cos_poly(s);
  if ((q & 1) != 0) cu = -cu;
  if (fabsf(su) > 1.0f) su = 0.0f;
  if (fabsf(cu) > 1.0f) cu = 0.0f;
  *sine = su;
  *cosine = cu;
}
/* pbrlab_math.h:217-241 */
static inline float orc_fast_exp2(float xval) {
  float x = orc_fast_clamp(xval, -126.0f, 126.0f);
  int m = (int)x;
  x -= (float)m;
  x = 1.0f - (1.0f - x);
  float r = 1.33336498402e-3f;
  r = fmaf(x, r, 9.810352697968e-3f);
  r = fmaf(x, r, 5.551834031939e-2f);
  r = fmaf(x, r, 0.2401793301105f);
  r = fmaf(x, r, 0.693144857883f);
  r = fmaf(x, r, 1.0f);
  return orc_u2f(orc_f2u(r) + ((uint32_t)m << 23));
}
/* pbrlab_math.h:243-248 */
static inline float orc_fast_exp(float x) {
  return orc_fast_exp2(x * (float)(1 / 0.69314718055994530942));
}
/* pbrlab_math.h:250-278 */
static inline float orc_fast_atan2(float y, float x) {
  float a = fabsf(x);
  float b = fabsf(y);
  float k = (b == 0) ? 0.0f : ((a == b) ? 1.0f : (b > a ? a / b : b / a));
  float s = 1.0f - (1.0f - k);
  float t = s * s;
  float r = s * fmaf(0.430165678f, t, 1.0f) / fmaf(fmaf(0.0579354987f, t, 0.763007998f), t, 1.0f);
  if (b > a) r = 1.570796326794896557998982f - r;
  if (orc_f2u(x) & 0x80000000u) r = (float)ORC_PI - r;
  return copysignf(r, y);
}
/* pbrlab_math.h:280-294 */
static inline float orc_fast_asin(float x) {
  float f = fabsf(x);
  float m = (f < 1.0f) ? 1.0f - (1.0f - f) : 1.0f;
  float a = (float)1.57079632679489661923 -
            sqrtf(1.0f - m) *
                (1.5707963267f + m * (-0.213300989f + m * (0.077980478f + m * -0.02164095f)));
  return copysignf(a, x);
}
/* pbrlab_math.h:314-338 */
static inline float orc_fast_log2(float xval) {
  float x = orc_fast_clamp(xval, FLT_MIN, FLT_MAX);
  uint32_t bits = orc_f2u(x);
  int exponent = (int)(bits >> 23) - 127;
  float f = orc_u2f((bits & 0x007FFFFFu) | 0x3f800000u) - 1.0f;
  float f2 = f * f;
  float f4 = f2 * f2;
  float hi = fmaf(f, -0.00931049621349f, 0.05206469089414f);
  float lo = fmaf(f, 0.47868480909345f, -0.72116591947498f);
  hi = fmaf(f, hi, -0.13753123777116f);
  hi = fmaf(f, hi, 0.24187369696082f);
  hi = fmaf(f, hi, -0.34730547155299f);
  lo = fmaf(f, lo, 1.442689881667200f);
  return ((f4 * hi) + (f * lo)) + (float)exponent;
}
/* pbrlab_math.h:340-344 */
static inline float orc_fast_log(float x) { return orc_fast_log2(x) * (float)0.69314718055994530942; }

#endif /* ORC_MATH_H_ */
