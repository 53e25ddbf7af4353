/*
 * pbr_glibcf.h -- cosf / sinf / expf / logf AS THE REFERENCE COMPUTES THEM on the platform this project runs on: bit for bit the
 * float functions of GNU libc (the algorithms it ships since 2.28; the build pinned is Ubuntu 22.04's 2.35) on x86-64 hardware with FMA, restated with explicit IEEE double arithmetic so that
 * gcc on the host and hipcc on gfx950 produce the same bits.  The HIP kernels (pbrlab_amd/csrc/dmath.h) and the CPU checker (its
 * "glibcf" arithmetic mode) compile this file verbatim.
 *
 * Why.  pbrlab calls std::cos / std::sin / std::exp / std::log on float (sampler/sampling-utils.h:10-14,
 * closure/microfacet-ggx.h:55-118, shader/random-walk-sss.h:116,183,192-194): the host libm's float functions, whose last ulp is
 * platform-defined.  Rounds 1-4 evaluated them correctly rounded ("f64r", include/pbr_f64r.h): independent of any libm, but a
 * last-ulp difference against the reference's libm flips a discrete decision of about one sample in 10^5, and on the dark,
 * heavy-tailed C5 frame those flips alone are 1.1e-3 of relative L2 at the configuration's 1024 spp -- above the 1e-4 bar
 * (round 5).  The only arithmetic that IS the reference's is the libm's own.
 *
 * What is restated.  glibc's flt-32 functions are Szabolcs Nagy's (ARM Optimized Routines, MIT licence; glibc
 * sysdeps/ieee754/flt-32/{e_expf.c, e_logf.c, s_sinf.c, s_cosf.c, s_sincosf.h} since 2.28): a table-driven exp2-style expf
 * (N = 32), a 16-interval logf, sinf / cosf with a shared degree-7 / degree-8 double polynomial after a fast x * 2/pi reduction
 * (and an integer 192-bit reduction for |x| >= 120), all evaluated in double and rounded once to float -- NOT correctly rounded
 * (0.50-0.56 ulp).  On x86-64 the dynamic loader picks the *_fma variants (the -fma.c files of sysdeps/x86_64/fpu/multiarch: the same C
 * compiled with -mfma -mavx2) on every CPU with FMA and AVX2, and there the compiler has CONTRACTED products and sums into fused
 * multiply-adds.  Which ones is not written in any source; it was read off the instruction sequences of Ubuntu's libm.so.6
 * 2.35-0ubuntu3.11 (`objdump -d`, functions behind the ifunc resolvers of expf / logf / sinf / cosf) and is reproduced below
 * with explicit fma() calls, operand order included; the constants are the published ones (they are also what that binary holds).
 *
 * Pin (tests/test_glibcf.py): against the host's libm, ALL 2^32 arguments of expf and logf and 2^32 of sinf / cosf: 0 differing
 * results on glibc 2.35 / x86-64 / FMA (this container and the GPU boxes).  On a host whose libm is something else the test says
 * so and skips: there the functions below are still what the GPU computes, and the checker's "glibcf" mode still equals the GPU
 * bit for bit -- only the statement "equal to the reference's own arithmetic" is then about another platform.
 *
 * The includer may define GLIBCF_FN (function qualifiers; default `static inline`) and GLIBCF_TAB (qualifiers of the tables;
 * default `static const`; device compilation: `static __device__ const`).
 */
#ifndef PBR_GLIBCF_H_
#define PBR_GLIBCF_H_

#include <stdint.h>

#ifndef GLIBCF_FN
#define GLIBCF_FN static inline
#endif
#ifndef GLIBCF_TAB
#define GLIBCF_TAB static const
#endif

GLIBCF_FN uint64_t glibcf_bits(double d) {
  uint64_t u;
  __builtin_memcpy(&u, &d, 8);
  return u;
}
GLIBCF_FN double glibcf_from_bits(uint64_t u) {
  double d;
  __builtin_memcpy(&d, &u, 8);
  return d;
}
GLIBCF_FN uint32_t glibcf_fbits(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}
GLIBCF_FN float glibcf_from_fbits(uint32_t u) {
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
/* correctly rounded a * b + c (v_fma_f64 on the device; the FMA instruction or libm's exact fma on the host) */
GLIBCF_FN double glibcf_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
GLIBCF_FN float glibcf_nan(void) { return glibcf_from_fbits(0x7fc00000u); } /* __math_invalidf: (x - x) / 0 */

/* ---------------------------------------------------------------------------------------------------------------- expf
 * exp(x) = 2^(k/N) * 2^(r/N), k = round(x * N / ln 2), N = 32; tab[i] = bits(2^(i/N)) - (i << 47) so that adding k << 47 puts k / N
 * into the exponent; the polynomial is 2^(r/N) ~ 1 + C2 r + r^2 (C1 + C0 r) with the 1 / N powers folded into the coefficients. */
GLIBCF_TAB uint64_t glibcf_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

GLIBCF_FN float glibcf_expf(float x) {
  const uint32_t ix = glibcf_fbits(x), abstop = (ix >> 20) & 0x7ffu;
  const double xd = (double)x;
  if (abstop > 0x42au) { /* |x| >= 88 or NaN (the binary compares the top 12 bits with those of 88.0f) */
    if (ix == 0xff800000u) return 0.0f;
    if (abstop > 0x7f7u) return x + x;                                 /* inf, NaN */
    if (x > 0x1.62e42ep6f) return glibcf_from_fbits(0x7f800000u);     /* __math_oflowf */
    if (x < -0x1.9fe368p6f) return 0.0f;                              /* __math_uflowf */
    if (x < -0x1.9d1d9ep6f) return glibcf_from_fbits(0x00000001u);    /* __math_may_uflowf: 0x1.4p-75f * 0x1.4p-75f, rounded */
  }
  const double invln2n = 0x1.71547652b82fep+5, shift = 0x1.8p+52;
  double kd = glibcf_fma(invln2n, xd, shift);     /* (the product x * N / ln 2 is never rounded on its own in the FMA build) */
  const uint64_t ki = glibcf_bits(kd);
  kd -= shift;
  const double r = glibcf_fma(invln2n, xd, -kd);
  const double s = glibcf_from_bits(glibcf_exp2f_tab[ki & 31u] + (ki << 47));
  const double z = glibcf_fma(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
  const double r2 = r * r;
  double y = glibcf_fma(0x1.62e42ff0c52d6p-6, r, 1.0);
  y = glibcf_fma(z, r2, y);
  y = y * s;
  return (float)y;
}

/* ---------------------------------------------------------------------------------------------------------------- logf
 * x = 2^k z, z in [OFF, 2 OFF), OFF = 0x3f330000; the interval of z picks (invc, logc) ~ (1 / c, log c) with c near its centre;
 * log x = log1p(z / c - 1) + log c + k ln 2, the log1p by a degree-4 polynomial in r = z * invc - 1. */
GLIBCF_TAB double glibcf_logf_tab[32] = {
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2, 0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2,
    0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3, 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4, 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5,
    0x1.0000000000000p+0, 0x0.0p+0,              0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5,  0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4,
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3,  0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3,  0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,
    0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2};

GLIBCF_FN float glibcf_logf(float x) {
  uint32_t ix = glibcf_fbits(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u > 0x7effffffu) {                      /* x < 0x1p-126, inf or NaN */
    if (ix * 2u == 0u) return glibcf_from_fbits(0xff800000u); /* __math_divzerof(1): -inf */
    if (ix == 0x7f800000u) return x;
    if ((ix & 0x80000000u) || ix * 2u > 0xfeffffffu) return glibcf_nan(); /* negative or NaN (the binary returns x for a quiet NaN argument's sign; NaN either way) */
    ix = glibcf_fbits(x * 0x1p23f);                           /* subnormal: normalise */
    ix -= 23u << 23;
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const uint32_t i = (tmp >> 19) & 15u;
  const int32_t k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double invc = glibcf_logf_tab[2u * i], logc = glibcf_logf_tab[2u * i + 1u];
  const double z = (double)glibcf_from_fbits(iz);
  const double y0 = glibcf_fma((double)k, 0x1.62e42fefa39efp-1, logc);
  const double r = glibcf_fma(z, invc, -1.0);
  double y = glibcf_fma(r, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2);
  const double r2 = r * r;
  const double t = r + y0;
  y = glibcf_fma(r2, -0x1.00ea348b88334p-2, y);
  y = glibcf_fma(r2, y, t);
  return (float)y;
}

/* ---------------------------------------------------------------------------------------------------------- sinf, cosf
 * x = n * pi/2 + r: for |x| < 120 n = round(x * 2/pi) through a 2^24-scaled product truncated to int32 (the binary's vcvttsd2si),
 * r = x - n * pi/2 in ONE fused operation; beyond, the 192-bit integer reduction with the bits of 4/pi.  On [-pi/4, pi/4]:
 *   sin r ~ r + r^3 S1 + r^5 (S2 + r^2 S3),    cos r ~ (C0 + r^2 C1) + r^4 C2 + r^6 (C3 + r^2 C4),
 * the quadrant selects the polynomial and the sign (glibc keeps a second coefficient table with the cosine's coefficients negated:
 * negating every coefficient negates every fused result exactly, so one table and a final negation give the same bits). */
GLIBCF_TAB uint32_t glibcf_inv_pio4[24] = {0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
                                           0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u,
                                           0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};

GLIBCF_FN double glibcf_sin_poly(double x, double x2) {
  const double s1 = glibcf_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);                              /* S2 + x2 * S3 */
  const double x3 = x2 * x;
  const double x5 = x2 * x3;
  const double s = glibcf_fma(x3, -0x1.555545995a603p-3, x);                                                  /* x + x3 * S1 */
  return glibcf_fma(s1, x5, s);
}
GLIBCF_FN double glibcf_cos_poly(double x2) {
  const double x4 = x2 * x2;
  const double c1 = glibcf_fma(x2, -0x1.ffffffd0c621cp-2, 1.0);                       /* C0 + x2 * C1 */
  const double c2 = glibcf_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);    /* C3 + x2 * C4 */
  const double x6 = x4 * x2;
  const double c = glibcf_fma(x4, 0x1.55553e1068f19p-5, c1);                          /* c1 + x4 * C2 */
  return glibcf_fma(c2, x6, c);
}
/* the reduced argument and the quadrant: *np = n (|x| < 120) or n + sign (beyond), *odd = n & 1 */
GLIBCF_FN double glibcf_reduce(float y, uint32_t abstop, int32_t* np, int32_t* odd) {
  const double x = (double)y;
  if (abstop <= 0x42eu) { /* |y| < 120 */
    const double r = x * 0x1.45f306dc9c883p+23;
    const int32_t n = ((int32_t)r + 0x800000) >> 24;
    *np = n, *odd = n & 1;
    return glibcf_fma(-(double)n, 0x1.921fb54442d18p+0, x);
  }
  const uint32_t xi0 = glibcf_fbits(y);
  const uint32_t* arr = &glibcf_inv_pio4[(xi0 >> 26) & 15u];
  const uint32_t shift = (xi0 >> 23) & 7u;
  const uint32_t xi = ((xi0 & 0x7fffffu) | 0x800000u) << shift;
  uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
  const uint64_t res1 = (uint64_t)xi * arr[4], res2 = (uint64_t)xi * arr[8];
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  const uint64_t n = (res0 + (1ull << 61)) >> 62;
  res0 -= n << 62;
  *np = (int32_t)n + (int32_t)(xi0 >> 31), *odd = (int32_t)(n & 1u);
  return (double)(int64_t)res0 * 0x1.921fb54442d18p-62;
}
GLIBCF_FN float glibcf_sinf(float y) {
  const uint32_t abstop = (glibcf_fbits(y) >> 20) & 0x7ffu;
  if (abstop <= 0x3f3u) { /* |y| < pi/4 */
    const double x = (double)y, x2 = x * x;
    if (abstop <= 0x397u) return y; /* |y| < 2^-12 */
    return (float)glibcf_sin_poly(x, x2);
  }
  if (abstop > 0x7f7u) return glibcf_nan();
  int32_t n, odd;
  const double x = glibcf_reduce(y, abstop, &n, &odd);
  const double x2 = x * x;
  if (!odd) return (float)glibcf_sin_poly(x * (((n + 1) & 2) ? -1.0 : 1.0), x2);   /* sign[n & 3] = {1, -1, -1, 1} */
  const double c = glibcf_cos_poly(x2);
  return (float)((n & 2) ? -c : c);
}
GLIBCF_FN float glibcf_cosf(float y) {
  const uint32_t abstop = (glibcf_fbits(y) >> 20) & 0x7ffu;
  if (abstop <= 0x3f3u) {
    const double x = (double)y, x2 = x * x;
    if (abstop <= 0x397u) return 1.0f;
    return (float)glibcf_cos_poly(x2);
  }
  if (abstop > 0x7f7u) return glibcf_nan();
  int32_t n, odd;
  const double x = glibcf_reduce(y, abstop, &n, &odd);
  const double x2 = x * x;
  if (odd) return (float)glibcf_sin_poly(x * (((n + 1) & 2) ? -1.0 : 1.0), x2);
  const double c = glibcf_cos_poly(x2);
  return (float)((n & 2) ? -c : c);
}

#endif /* PBR_GLIBCF_H_ */
