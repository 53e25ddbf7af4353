/*
 * pbr_f64r.h -- the four transcendental functions of the hot path, in ONE fixed implementation that the HIP kernels
 * (pbrlab_amd/csrc/dmath.h) and the CPU checker of the test suite (its "f64r" arithmetic mode) compile verbatim.
 *
 * What they stand for: the reference calls std::cos / std::sin / std::exp / std::log on float (Lambert and GGX sampling
 * sampler/sampling-utils.h:10-14, closure/microfacet-ggx.h:55-118; random walk shader/random-walk-sss.h:116,183,192-194),
 * i.e. the host libm's float functions, whose last ulp is platform-defined.  "f64r" = the function evaluated in double
 * precision and rounded ONCE to float.  The functions below do that with fixed arithmetic only -- IEEE double +, -, *, one
 * division (log), fma, and integer bit operations; no libm / OCML call, no table, no Payne-Hanek path -- so that the same
 * bits come out of gcc on x86-64 and of hipcc on gfx950, and a scattering of the random walk costs tens instead of hundreds
 * of VALU instructions.
 *
 * Accuracy (tests/test_f64r.py, 10^7 samples per function against glibc's double functions): the double result is within
 * 2^-44 relative of the true value before the rounding to float, i.e. the float result is within 0.5 ulp + 2^-20 ulp.
 *   sin / cos   arguments |x| <= 1e5 (the callers pass [0, 2 pi]); reduction by k * pi/2 with a two-part constant
 *               (Cody-Waite), kernels on [-pi/4, pi/4]: the fdlibm minimax polynomials (k_sin.c / k_cos.c, Sun Microsystems,
 *               freely distributable)
 *   exp         any float: k = rint(x / ln 2), Taylor polynomial of degree 13 on |r| <= ln 2 / 2, scaled by 2^k; arguments below
 *               -110 give 0, above 100 give +inf (both beyond the float range)
 *   log         any float > 0 incl. denormals: x = 2^e * m, m in [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1), the fdlibm series
 *               in s^2 (e_log.c); log(0) = -inf, log(< 0) = NaN
 * NaN in, NaN out.
 *
 * The includer may define F64R_FN (function qualifiers; default `static inline`).
 */
#ifndef PBR_F64R_H_
#define PBR_F64R_H_

#include <math.h>
#include <stdint.h>

#ifndef F64R_FN
#define F64R_FN static inline
#endif

F64R_FN uint64_t f64r_bits(double d) {
  uint64_t u;
  __builtin_memcpy(&u, &d, 8);
  return u;
}
F64R_FN double f64r_from_bits(uint64_t u) {
  double d;
  __builtin_memcpy(&d, &u, 8);
  return d;
}
/* correctly rounded a * b + c (v_fma_f64 on the device; the FMA instruction or libm's exact fma on the host) */
F64R_FN double f64r_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

/* rint for |v| < 2^51: adding and subtracting 1.5 * 2^52 rounds to the nearest integer (ties to even); *lo = its low 32 bits */
F64R_FN double f64r_rint(double v, int32_t* lo) {
  const double magic = 6755399441055744.0;
  const double t = v + magic;
  *lo = (int32_t)(uint32_t)f64r_bits(t);
  return t - magic;
}

/* sin and cos of x (float) evaluated in double; *s, *c are the doubles BEFORE the rounding to float */
F64R_FN void f64r_sincos_d(double x, double* s, double* c) {
  if (!(x >= -1e9 && x <= 1e9)) { /* infinities, NaN and arguments far outside the supported range: one fixed NaN */
    *s = *c = (double)__builtin_nanf("");
    return;
  }
  int32_t q;
  const double k = f64r_rint(x * 6.36619772367581382433e-01, &q); /* x * 2 / pi */
  double r = f64r_fma(-k, 1.57079632673412561417e+00, x);         /* first 33 bits of pi / 2: k * that is exact */
  r = f64r_fma(-k, 6.07710050650619224932e-11, r);                /* the rest of pi / 2 */
  const double z = r * r;
  /* sin r = r + r^3 (S1 + z (S2 + ...)) */
  double ps = 1.58969099521155010221e-10;
  ps = f64r_fma(ps, z, -2.50507602534068634195e-08);
  ps = f64r_fma(ps, z, 2.75573137070700676789e-06);
  ps = f64r_fma(ps, z, -1.98412698298579493134e-04);
  ps = f64r_fma(ps, z, 8.33333333332248946124e-03);
  ps = f64r_fma(ps, z, -1.66666666666666324348e-01);
  const double sr = f64r_fma(r * z, ps, r);
  /* cos r = 1 - z / 2 + z^2 (C1 + z (C2 + ...)) */
  double pc = -1.13596475577881948265e-11;
  pc = f64r_fma(pc, z, 2.08757232129817482790e-09);
  pc = f64r_fma(pc, z, -2.75573143513906633035e-07);
  pc = f64r_fma(pc, z, 2.48015872894767294178e-05);
  pc = f64r_fma(pc, z, -1.38888888888741095749e-03);
  pc = f64r_fma(pc, z, 4.16666666666666019037e-02);
  const double cr = f64r_fma(z * z, pc, f64r_fma(-0.5, z, 1.0));
  /* quadrant k mod 4: (sin, cos) -> (s, c), (c, -s), (-s, -c), (-c, s) */
  const double a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;
  *s = (q & 2) ? -a : a;
  *c = ((q + 1) & 2) ? -b : b;
}
F64R_FN void f64r_sincosf(float x, float* s, float* c) {
  double sd, cd;
  f64r_sincos_d((double)x, &sd, &cd);
  *s = (float)sd, *c = (float)cd;
}
F64R_FN float f64r_sinf(float x) {
  double sd, cd;
  f64r_sincos_d((double)x, &sd, &cd);
  return (float)sd;
}
F64R_FN float f64r_cosf(float x) {
  double sd, cd;
  f64r_sincos_d((double)x, &sd, &cd);
  return (float)cd;
}

F64R_FN float f64r_expf(float xf) {
  const double x = (double)xf;
  if (!(x > -110.0)) return (x != x) ? xf : 0.0f; /* NaN stays; exp(-110) is far below the smallest float */
  if (x > 100.0) return __builtin_inff();
  int32_t q;
  const double k = f64r_rint(x * 1.44269504088896338700e+00, &q); /* x / ln 2 */
  double r = f64r_fma(-k, 6.93147180369123816490e-01, x);         /* ln 2, high part (k * that is exact) */
  r = f64r_fma(-k, 1.90821492927058770002e-10, r);                /* ln 2, low part */
  /* exp r, |r| <= 0.3466: Taylor, degree 13 */
  double p = 1.0 / 6227020800.0;
  p = f64r_fma(p, r, 1.0 / 479001600.0);
  p = f64r_fma(p, r, 1.0 / 39916800.0);
  p = f64r_fma(p, r, 1.0 / 3628800.0);
  p = f64r_fma(p, r, 1.0 / 362880.0);
  p = f64r_fma(p, r, 1.0 / 40320.0);
  p = f64r_fma(p, r, 1.0 / 5040.0);
  p = f64r_fma(p, r, 1.0 / 720.0);
  p = f64r_fma(p, r, 1.0 / 120.0);
  p = f64r_fma(p, r, 1.0 / 24.0);
  p = f64r_fma(p, r, 1.0 / 6.0);
  p = f64r_fma(p, r, 0.5);
  p = f64r_fma(p, r, 1.0);
  p = f64r_fma(p, r, 1.0);
  const double scale = f64r_from_bits((uint64_t)(int64_t)(q + 1023) << 52); /* 2^k, k in [-159, 145] */
  return (float)(p * scale);
}

F64R_FN float f64r_logf(float xf) {
  const double x = (double)xf; /* (a denormal float is a normal double) */
  if (!(x > 0.0)) return (x == 0.0) ? -__builtin_inff() : __builtin_nanf(""); /* log 0 = -inf; negatives and NaN: one fixed NaN */
  if (!(x < 1e300)) return xf;                                                 /* +inf */
  uint64_t b = f64r_bits(x);
  int32_t e = (int32_t)(b >> 52) - 1023;
  b = (b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull; /* m in [1, 2) */
  double m = f64r_from_bits(b);
  if (m > 1.41421356237309514547) m = m * 0.5, e = e + 1; /* m in [sqrt(1/2), sqrt(2)) */
  const double f = m - 1.0;
  const double s = f / (m + 1.0);
  const double z = s * s;
  /* log m = 2 s + 2 s z (L1 + z (L2 + ...)) = 2 atanh s */
  double p = 1.479819860511658591e-01;
  p = f64r_fma(p, z, 1.531383769920937332e-01);
  p = f64r_fma(p, z, 1.818357216161805012e-01);
  p = f64r_fma(p, z, 2.222219843214978396e-01);
  p = f64r_fma(p, z, 2.857142874366239149e-01);
  p = f64r_fma(p, z, 3.999999999940941908e-01);
  p = f64r_fma(p, z, 6.666666666666735130e-01);
  const double lm = f64r_fma(s * z, p, 2.0 * s);
  const double de = (double)e;
  return (float)f64r_fma(de, 6.93147180369123816490e-01, f64r_fma(de, 1.90821492927058770002e-10, lm));
}

#endif /* PBR_F64R_H_ */
