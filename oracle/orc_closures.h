/*
 * orc_closures.h -- ORACLE (test infrastructure only).
 * Plain-C restatement of pbrlab's closures: Lambert, dielectric Fresnel, Cycles/OSL GGX with
 * Heitz-d'Eon VNDF sampling, and the 4-lobe energy-conserving hair BSDF.  file:line = reference.
 */
#ifndef ORC_CLOSURES_H_
#define ORC_CLOSURES_H_

#include "orc_math.h"

/* ------------------------------------------------------- Lambert (src/closure/lambert.h:11-27) */
static inline float orc_lambert_pdf(f3 omega_in) { return omega_in.z * ORC_PI_INV; } /* Q8: no clamp */
static inline float orc_lambert_brdf_pdf(f3 omega_in, float* pdf) {
  *pdf = orc_lambert_pdf(omega_in);
  return ORC_PI_INV;
}
static inline float orc_lambert_sample(float u0, float u1, f3* omega_in, float* pdf) {
  *omega_in = orc_cosine_sample_hemisphere(u0, u1);
  return orc_lambert_brdf_pdf(*omega_in, pdf);
}

/* ---------------------------------------- Fresnel (src/closure/closure-util.h:10-29) */
static inline float orc_fresnel_dielectric_cos(float cos_, float eta) {
  if (fabsf(eta) < FLT_EPSILON) return 1.0f;
  if (cos_ < 0.0f) eta = 1.0f / eta;
  float c = fabsf(cos_);
  float g = eta * eta - 1 + c * c;
  if (g > 0) {
    g = sqrtf(g);
    float A = (g - c) / (g + c);
    float B = (c * (g + c) - 1) / (c * (g - c) + 1);
    return 0.5f * A * A * (1 + B * B);
  }
  return 1.0f;
}

/* ------------------------------------------ GGX (src/closure/microfacet-ggx.h) */
/* :48-53 */
static inline float orc_d_gtr1(f3 h, float alpha) {
  if (alpha >= 1.0f) return 1.0f / ORC_PI;
  float alpha2 = alpha * alpha;
  float t = 1.0f + (alpha2 - 1.0f) * h.z * h.z;
  return (alpha2 - 1.0f) / (ORC_PI * orc_logf(alpha2) * t);
}
/* :55-63 */
static inline float orc_d_gtr2(f3 h, float alpha2) {
  float c = h.z;
  float c2 = c * c;
  float c4 = c2 * c2;
  float tan2 = (1.0f - c2) / c2;
  return alpha2 / (ORC_PI * c4 * (alpha2 + tan2) * (alpha2 + tan2));
}
/* :65-118 */
static inline void orc_ggx_sample_slopes(float cos_theta_i, float sin_theta_i, float randu,
                                         float randv, float* slope_x, float* slope_y, float* G1i) {
  const float k2PI = 2.0f * ORC_PI;
  if (cos_theta_i >= 0.99999f) {
    float r = sqrtf(randu / (1.0f - randu));
    float phi = k2PI * randv;
    *slope_x = r * orc_cosf(phi);
    *slope_y = r * orc_sinf(phi);
    *G1i = 1.0f;
    return;
  }
  float tan_theta_i = sin_theta_i / cos_theta_i;
  float G1_inv = 0.5f * (1.0f + orc_safe_sqrtf(1.0f + tan_theta_i * tan_theta_i));
  *G1i = 1.0f / G1_inv;

  float A = 2.0f * randu * G1_inv - 1.0f;
  float AA = A * A;
  float tmp = 1.0f / (AA - 1.0f);
  float B = tan_theta_i;
  float BB = B * B;
  float D = orc_safe_sqrtf(BB * (tmp * tmp) - (AA - BB) * tmp);
  float slope_x_1 = B * tmp - D;
  float slope_x_2 = B * tmp + D;
  *slope_x = (A < 0.0f || slope_x_2 * tan_theta_i > 1.0f) ? slope_x_1 : slope_x_2;

  float S;
  if (randv > 0.5f) {
    S = 1.0f;
    randv = 2.0f * (randv - 0.5f);
  } else {
    S = -1.0f;
    randv = 2.0f * (0.5f - randv);
  }
  float z = (randv * (randv * (randv * 0.27385f - 0.73369f) + 0.46341f)) /
            (randv * (randv * (randv * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
  *slope_y = S * z * orc_safe_sqrtf(1.0f + (*slope_x) * (*slope_x));
}
/* :121-162 */
static inline f3 orc_microfacet_sample_stretched(f3 omega_i, float alpha_x, float alpha_y,
                                                 float randu, float randv, float* G1i) {
  f3 w = f3_normalize(f3_make(alpha_x * omega_i.x, alpha_y * omega_i.y, omega_i.z));
  float costheta_ = 1.0f, sintheta_ = 0.0f, cosphi_ = 1.0f, sinphi_ = 0.0f;
  if (w.z < 0.99999f) {
    costheta_ = w.z;
    sintheta_ = orc_safe_sqrtf(1.0f - costheta_ * costheta_);
    float invlen = 1.0f / sintheta_;
    cosphi_ = w.x * invlen;
    sinphi_ = w.y * invlen;
  }
  float slope_x = 0.f, slope_y = 0.f;
  orc_ggx_sample_slopes(costheta_, sintheta_, randu, randv, &slope_x, &slope_y, G1i);
  float tmp = cosphi_ * slope_x - sinphi_ * slope_y;
  slope_y = sinphi_ * slope_x + cosphi_ * slope_y;
  slope_x = tmp;
  slope_x = alpha_x * slope_x;
  slope_y = alpha_y * slope_y;
  return f3_normalize(f3_make(-slope_x, -slope_y, 1.0f));
}
/* :164-245  distrib: 1 = GTR1 (clearcoat), 2 = GTR2 */
static inline float orc_ggx_bsdf_pdf(f3 omega_in, f3 omega_out, float alpha_x, float alpha_y,
                                     int distrib, float* pdf) {
  float cos_n_o = omega_out.z;
  float cos_n_i = omega_in.z;
  if (cos_n_o > 0 && cos_n_i > 0) {
    f3 m = f3_normalize(f3_add(omega_in, omega_out));
    float alpha2 = alpha_x * alpha_y;
    float D = 0.f, G1o = 0.f, G1i = 0.f;
    if (fabsf(alpha_x - alpha_y) < FLT_EPSILON) {
      if (distrib == 1) {
        D = orc_d_gtr1(m, alpha_x);
        alpha2 = 0.0625f;
      } else {
        D = orc_d_gtr2(m, alpha2);
      }
      G1o = 2 / (1 + orc_safe_sqrtf(1 + alpha2 * (1 - cos_n_o * cos_n_o) / (cos_n_o * cos_n_o)));
      G1i = 2 / (1 + orc_safe_sqrtf(1 + alpha2 * (1 - cos_n_i * cos_n_i) / (cos_n_i * cos_n_i)));
    } else {
      float slope_x = -m.x / (m.z * alpha_x);
      float slope_y = -m.y / (m.z * alpha_y);
      float slope_len = 1 + slope_x * slope_x + slope_y * slope_y;
      float cosThetaM = m.z;
      float cosThetaM2 = cosThetaM * cosThetaM;
      float cosThetaM4 = cosThetaM2 * cosThetaM2;
      D = 1.f / ((slope_len * slope_len) * ORC_PI * alpha2 * cosThetaM4);

      float tanThetaO2 = (1.f - cos_n_o * cos_n_o) / (cos_n_o * cos_n_o);
      float cosPhiO = omega_out.x;
      float sinPhiO = omega_out.y;
      float alphaO2 = (cosPhiO * cosPhiO) * (alpha_x * alpha_x) + (sinPhiO * sinPhiO) * (alpha_y * alpha_y);
      alphaO2 /= cosPhiO * cosPhiO + sinPhiO * sinPhiO;
      G1o = 2 / (1 + orc_safe_sqrtf(1 + alphaO2 * tanThetaO2));

      float tanThetaI2 = (1 - cos_n_i * cos_n_i) / (cos_n_i * cos_n_i);
      float cosPhiI = omega_in.x;
      float sinPhiI = omega_in.y;
      float alphaI2 = (cosPhiI * cosPhiI) * (alpha_x * alpha_x) + (sinPhiI * sinPhiI) * (alpha_y * alpha_y);
      alphaI2 /= cosPhiI * cosPhiI + sinPhiI * sinPhiI;
      G1i = 2 / (1 + orc_safe_sqrtf(1 + alphaI2 * tanThetaI2));
    }
    float G = G1o * G1i;
    float common = D * 0.25f / cos_n_o / cos_n_i;
    float bsdf_f = G * common;
    if (distrib == 1) bsdf_f = 0.25f * bsdf_f;
    *pdf = G1o * common;
    return bsdf_f;
  }
  *pdf = 0.f;
  return 0.f;
}
/* :247-286  reflect-only.  When cos_n_o <= 0 or cos_m_o <= 0 neither *omega_in nor *pdf is
 * written (the caller's initial values survive) -- kept. */
static inline float orc_ggx_sample(f3 omega_out, float alpha_x, float alpha_y, float u0, float u1,
                                   int distrib, f3* omega_in, float* pdf) {
  float cos_n_o = omega_out.z;
  float ret = 0.f;
  if (cos_n_o > 0.f) {
    float G1o = 0.f;
    f3 m = orc_microfacet_sample_stretched(omega_out, alpha_x, alpha_y, u0, u1, &G1o);
    float cos_m_o = f3_dot(m, omega_out);
    if (cos_m_o > 0) {
      *omega_in = f3_sub(f3_scale(m, 2 * cos_m_o), omega_out);
      ret = orc_ggx_bsdf_pdf(*omega_in, omega_out, alpha_x, alpha_y, distrib, pdf);
    }
  }
  return ret;
}

/* ---------------------- hair BSDF (src/closure/energy-conserving-hair-bsdf.h, USE_FAST_MATH=1,
 * USE_IMPROVED_ROBE_EVALUATION=1).  The reference's std::cerr diagnostics are dropped (Q12). */
typedef struct {
  f3 sigma_a;
  float h;
  float v[4];
  float s;
  float eta;
  float alpha;
  f3 tints[4];
  float transparent_scale;
} orc_hair_bsdf;

/* :42-49 */
static inline float orc_hair_safe_asin(float x) {
  float ret = orc_fast_asin(x);
  if (isnan(ret)) return orc_fast_asin(orc_clamp(x, -1.0f, 1.0f));
  return ret;
}
/* :82-90 */
static inline float orc_horner(float x, const float* a, int n) {
  float f = a[n];
  for (int i = n - 1; i >= 0; i--) f = f * x + a[i];
  return f;
}
/* :92-170 (improved-lobe branch; note "+ 1.0f" sits outside the log -- kept) */
static inline float orc_safe_log_i0(float x) {
  x = fabsf(x);
  if (x < 7.5f) {
    static const float P[] = {1.00000003928615375e+00f, 2.49999576572179639e-01f,
                              2.77785268558399407e-02f, 1.73560257755821695e-03f,
                              6.96166518788906424e-05f, 1.89645733877137904e-06f,
                              4.29455004657565361e-08f, 3.90565476357034480e-10f,
                              1.48095934745267240e-11f};
    float x22 = x * x / 4.0f;
    return orc_fast_log(x22 * orc_horner(x22, P, 8)) + 1.0f;
  }
  static const float Q[] = {3.98942651588301770e-01f, 4.98327234176892844e-02f,
                            2.91866904423115499e-02f, 1.35614940793742178e-02f,
                            1.31409251787866793e-01f};
  float inv_x = 1.0f / x;
  float Px = orc_horner(inv_x, Q, 4);
  return x + 0.5f * orc_fast_log(Px * Px * inv_x);
}
/* :172-202 */
static inline float orc_hair_mp(float sin_theta_i, float cos_theta_i, float sin_theta_o,
                                float cos_theta_o, float v) {
  float ccv = cos_theta_i * cos_theta_o / v;
  float ssv = sin_theta_i * sin_theta_o / v;
  v = orc_clamp(v, 1e-5f, 1e4f);
  return orc_fast_exp(orc_safe_log_i0(ccv) - ssv - 1.0f / v + orc_fast_log(1.0f / v) -
                      orc_fast_log(1.0f - orc_fast_exp(-2.0f / v)));
}
/* :205-229 */
static inline float orc_fr_dielectric(float cos_theta_i, float eta_i, float eta_t) {
  cos_theta_i = orc_clamp(cos_theta_i, -1.0f, 1.0f);
  int entering = cos_theta_i > 0.0f;
  if (!entering) {
    float a = eta_i;
    eta_i = eta_t;
    eta_t = a;
    cos_theta_i = fabsf(cos_theta_i);
  }
  float sin_theta_i = sqrtf(orc_max(0.0f, 1.0f - cos_theta_i * cos_theta_i));
  float sin_theta_t = eta_i / eta_t * sin_theta_i;
  if (sin_theta_t >= 1.0f) return 1.0f;
  float cos_theta_t = sqrtf(orc_max(0.0f, 1.0f - sin_theta_t * sin_theta_t));
  float r_parl = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) /
                 ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
  float r_perp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) /
                 ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
  return (r_parl * r_parl + r_perp * r_perp) * 0.5f;
}
/* :231-255 */
static inline void orc_hair_ap(float cos_theta_o, float eta, float h, f3 T, f3 ap[4]) {
  float cos_gamma_o = orc_safe_sqrtf(1.0f - h * h);
  float cos_theta = cos_theta_o * cos_gamma_o;
  float f = orc_fr_dielectric(cos_theta, 1.0f, eta);
  ap[0] = f3_set1(f);
  ap[1] = f3_scale(T, orc_sqr(1.0f - f));
  ap[2] = f3_scale(f3_mul(ap[1], T), f);
  ap[3] = f3_div(f3_mul(f3_scale(ap[2], f), T), f3_sub(f3_set1(1.0f), f3_scale(T, f)));
  if (!isfinite(ap[3].x) || !isfinite(ap[3].y) || !isfinite(ap[3].z)) ap[3] = f3_set1(0.0f);
}
/* :257-289 */
static inline float orc_logistic(float x, float s) {
  x = fabsf(x);
  float numerator = orc_fast_exp(-x / s);
  return numerator / (s * orc_sqr(1.0f + numerator));
}
static inline float orc_logistic_cdf(float x, float s) { return 1.0f / (1.0f + orc_fast_exp(-x / s)); }
static inline float orc_trimmed_logistic(float x, float s, float a, float b) {
  return orc_logistic(x, s) / (orc_logistic_cdf(b, s) - orc_logistic_cdf(a, s));
}
static inline float orc_hair_phi(int p, float gamma_o, float gamma_t) {
  return 2.0f * (float)p * gamma_t - 2.0f * gamma_o + (float)p * ORC_PI;
}
static inline float orc_fmod_floor(float a, float b) { return a - floorf(a / b) * b; }
static inline float orc_hair_np(float phi, int p, float s, float gamma_o, float gamma_t) {
  float dphi = orc_fmod_floor(phi - orc_hair_phi(p, gamma_o, gamma_t), 2.0f * ORC_PI);
  if (dphi >= ORC_PI) dphi -= 2.0f * ORC_PI;
  return orc_trimmed_logistic(dphi, s, -ORC_PI, ORC_PI);
}

/* common set-up of Eval (:295-362) and Sample (:419-476) */
typedef struct {
  float sin_theta_o, cos_theta_o;
  float sin_o_crt[4], cos_o_crt[4];
  float phi_o, gamma_o, gamma_t;
  f3 ap[4];
  float ap_pdf[4];
} orc_hair_setup;

static inline void orc_hair_prepare(f3 omega_out, const orc_hair_bsdf* b, orc_hair_setup* S) {
  S->sin_theta_o = omega_out.x;
  S->cos_theta_o = orc_safe_sqrtf(1.0f - orc_sqr(S->sin_theta_o));
  float s2k[3], c2k[3];
  orc_fast_sincos(b->alpha, &s2k[0], &c2k[0]);
  for (int i = 1; i < 3; i++) {
    s2k[i] = 2.0f * s2k[i - 1] * c2k[i - 1];
    c2k[i] = orc_sqr(c2k[i - 1]) - orc_sqr(s2k[i - 1]);
  }
  float so = S->sin_theta_o, co = S->cos_theta_o;
  S->sin_o_crt[0] = so * c2k[1] - co * s2k[1];
  S->cos_o_crt[0] = co * c2k[1] + so * s2k[1];
  S->sin_o_crt[1] = so * c2k[0] + co * s2k[0];
  S->cos_o_crt[1] = co * c2k[0] - so * s2k[0];
  S->sin_o_crt[2] = so * c2k[2] + co * s2k[2];
  S->cos_o_crt[2] = co * c2k[2] - so * s2k[2];
  S->sin_o_crt[3] = so;
  S->cos_o_crt[3] = co;

  S->phi_o = orc_fast_atan2(omega_out.z, omega_out.y);

  float sin_theta_t = so / b->eta;
  float cos_theta_t = orc_safe_sqrtf(1.f - orc_sqr(sin_theta_t));
  float etap = sqrtf(b->eta * b->eta - orc_sqr(so)) / co;
  float sin_gamma_t = b->h / etap;
  float cos_gamma_t = orc_safe_sqrtf(1.0f - orc_sqr(sin_gamma_t));
  S->gamma_t = orc_hair_safe_asin(sin_gamma_t);
  float l = b->transparent_scale * 2.0f * cos_gamma_t / cos_theta_t;
  f3 T = f3_make(orc_fast_exp(-b->sigma_a.x * l), orc_fast_exp(-b->sigma_a.y * l),
                 orc_fast_exp(-b->sigma_a.z * l));
  S->gamma_o = orc_hair_safe_asin(b->h);
  orc_hair_ap(co, b->eta, b->h, T, S->ap);
  float sum = 0.0f;
  for (int i = 0; i < 4; i++) sum = sum + orc_rgb_to_y(S->ap[i]);
  for (int i = 0; i < 4; i++) S->ap_pdf[i] = orc_rgb_to_y(S->ap[i]) / sum;
}

/* lobe sum shared by Eval (:364-404) and Sample (:537-571): returns f*cos, *pdf */
static inline f3 orc_hair_lobes(const orc_hair_setup* S, const orc_hair_bsdf* b, float sin_theta_i,
                                float cos_theta_i, float phi, float* pdf) {
  float pdfs[4];
  f3 ret = f3_set1(0.0f);
  for (int p = 0; p < 3; p++) {
    float mpnp = orc_hair_mp(sin_theta_i, cos_theta_i, S->sin_o_crt[p], S->cos_o_crt[p], b->v[p]) *
                 orc_hair_np(phi, p, b->s, S->gamma_o, S->gamma_t);
    pdfs[p] = mpnp * S->ap_pdf[p];
    ret = f3_add(ret, f3_mul(f3_scale(S->ap[p], mpnp), b->tints[p]));
  }
  float mpnp = orc_hair_mp(sin_theta_i, cos_theta_i, S->sin_theta_o, S->cos_theta_o, b->v[3]) *
               (1.0f / (2.0f * ORC_PI));
  pdfs[3] = mpnp * S->ap_pdf[3];
  ret = f3_add(ret, f3_mul(f3_scale(S->ap[3], mpnp), b->tints[3]));
  *pdf = 0.0f;
  if (!isfinite(ret.x) || !isfinite(ret.y) || !isfinite(ret.z)) return f3_set1(0.0f);
  *pdf = (((0.0f + pdfs[0]) + pdfs[1]) + pdfs[2]) + pdfs[3];
  if (!isfinite(*pdf)) {
    *pdf = 0.0f;
    return f3_set1(0.0f);
  }
  return ret;
}

/* :295-405 */
static inline f3 orc_hair_eval(f3 omega_in, f3 omega_out, const orc_hair_bsdf* b, float* pdf) {
  orc_hair_setup S;
  orc_hair_prepare(omega_out, b, &S);
  float sin_theta_i = omega_in.x;
  float cos_theta_i = orc_safe_sqrtf(1.0f - orc_sqr(sin_theta_i));
  float phi_i = orc_fast_atan2(omega_in.z, omega_in.y);
  float phi = phi_i - S.phi_o;
  return orc_hair_lobes(&S, b, sin_theta_i, cos_theta_i, phi, pdf);
}
/* :407-417 (the isinf/isfinite pair there can never fire -- omitted) */
static inline float orc_sample_trimmed_logistic(float s, float a, float b, float u) {
  float T = orc_logistic_cdf(b, s) - orc_logistic_cdf(a, s);
  return -s * orc_fast_log(1.0f / (u * T + 1.0f / (1.0f + orc_fast_exp(-a / s))) - 1.0f);
}
/* :419-572 */
static inline f3 orc_hair_sample(f3 omega_out, const orc_hair_bsdf* b, const float us[4],
                                 f3* omega_in, float* pdf) {
  orc_hair_setup S;
  orc_hair_prepare(omega_out, b, &S);
  int p;
  float u0 = us[0];
  for (p = 0; p < 3; p++) {
    if (u0 < S.ap_pdf[p]) break;
    u0 -= S.ap_pdf[p];
  }
  float u1 = us[1], u2 = us[2];
  float u = 1.0f + b->v[p] * orc_fast_log(u1 + (1.0f - u1) * orc_fast_exp(-2.0f / b->v[p]));
  float sin_theta_i = -u * S.sin_o_crt[p] +
                      orc_safe_sqrtf(1.0f - orc_sqr(u)) * orc_fast_cos(2.0f * ORC_PI * u2) * S.cos_o_crt[p];
  float cos_theta_i = orc_safe_sqrtf(1.0f - orc_sqr(sin_theta_i));
  float dphi;
  if (p < 3) {
    dphi = orc_hair_phi(p, S.gamma_o, S.gamma_t) + orc_sample_trimmed_logistic(b->s, -ORC_PI, ORC_PI, us[3]);
  } else {
    dphi = 2.0f * ORC_PI * us[3];
  }
  float phi_i = S.phi_o + dphi;
  *omega_in = f3_make(sin_theta_i, cos_theta_i * orc_fast_cos(phi_i), cos_theta_i * orc_fast_sin(phi_i));
  return orc_hair_lobes(&S, b, sin_theta_i, cos_theta_i, dphi, pdf);
}

#endif /* ORC_CLOSURES_H_ */
