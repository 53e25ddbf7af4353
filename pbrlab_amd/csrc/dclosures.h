// dclosures.h -- closures of the MI355X path tracer (host + device): Lambert, dielectric Fresnel,
// Cycles/OSL GGX (GTR2 / clearcoat GTR1) with Heitz-d'Eon VNDF sampling, the 4-lobe energy-conserving
// hair BSDF.  file:line = the pbrlab code each function stands in for.
#pragma once

#include "dmath.h"

namespace pb {

// ------------------------------------------------------------------ Lambert (closure/lambert.h:11-27)
PB_HD float lambert_eval(V3 wi, float& pdf) {
  pdf = wi.z * kPiInv;  // Q8: unclamped
  return kPiInv;
}
PB_HD float lambert_sample(float u0, float u1, V3& wi, float& pdf) {
  wi = cosine_sample_hemisphere(u0, u1);
  return lambert_eval(wi, pdf);
}

// ------------------------------------------------------------------ closure/closure-util.h:10-29
PB_HD float fresnel_dielectric_cos(float cos_, float eta) {
  if (fabsf(eta) < kFltEps) return 1.0f;
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

// ------------------------------------------------------------------ closure/microfacet-ggx.h
PB_HD float d_gtr1(V3 h, float alpha) {  // :48-53
  if (alpha >= 1.0f) return 1.0f / kPi;
  float a2 = alpha * alpha;
  float t = 1.0f + (a2 - 1.0f) * h.z * h.z;
  return (a2 - 1.0f) / (kPi * f_log(a2) * t);
}
PB_HD float d_gtr2(V3 h, float a2) {  // :55-63
  float c2 = h.z * h.z;
  float c4 = c2 * c2;
  float tan2 = (1.0f - c2) / c2;
  return a2 / (kPi * c4 * (a2 + tan2) * (a2 + tan2));
}
PB_HD void ggx_sample_slopes(float cos_i, float sin_i, float randu, float randv, float& slope_x, float& slope_y) {  // :65-118
  const float k2PI = 2.0f * kPi;
  if (cos_i >= 0.99999f) {
    float r = sqrtf(randu / (1.0f - randu));
    float phi = k2PI * randv;
    slope_x = r * f_cos(phi);
    slope_y = r * f_sin(phi);
    return;
  }
  float tan_i = sin_i / cos_i;
  float G1_inv = 0.5f * (1.0f + safe_sqrt(1.0f + tan_i * tan_i));
  float A = 2.0f * randu * G1_inv - 1.0f;
  float AA = A * A;
  float tmp = 1.0f / (AA - 1.0f);
  float B = tan_i;
  float BB = B * B;
  float D = safe_sqrt(BB * (tmp * tmp) - (AA - BB) * tmp);
  float s1 = B * tmp - D;
  float s2 = B * tmp + D;
  slope_x = (A < 0.0f || s2 * tan_i > 1.0f) ? s1 : s2;
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
  slope_y = S * z * safe_sqrt(1.0f + slope_x * slope_x);
}
PB_HD V3 microfacet_sample_stretched(V3 wo, float ax, float ay, float randu, float randv) {  // :121-162
  V3 w = vnormalize(V3(ax * wo.x, ay * wo.y, wo.z));
  float costheta = 1.0f, sintheta = 0.0f, cosphi = 1.0f, sinphi = 0.0f;
  if (w.z < 0.99999f) {
    costheta = w.z;
    sintheta = safe_sqrt(1.0f - costheta * costheta);
    float invlen = 1.0f / sintheta;
    cosphi = w.x * invlen;
    sinphi = w.y * invlen;
  }
  float sx = 0.f, sy = 0.f;
  ggx_sample_slopes(costheta, sintheta, randu, randv, sx, sy);
  float tmp = cosphi * sx - sinphi * sy;
  sy = sinphi * sx + cosphi * sy;
  sx = tmp;
  sx = ax * sx;
  sy = ay * sy;
  return vnormalize(V3(-sx, -sy, 1.0f));
}
// :164-245.  distrib 1 = clearcoat GTR1 (alpha^2 = 0.0625 inside G, extra 0.25), 2 = GTR2 (Q8)
PB_HD float ggx_eval(V3 wi, V3 wo, float ax, float ay, int distrib, float& pdf) {
  float cos_o = wo.z, cos_i = wi.z;
  if (cos_o > 0 && cos_i > 0) {
    V3 m = vnormalize(wi + wo);
    float alpha2 = ax * ay;
    float D, G1o, G1i;
    if (fabsf(ax - ay) < kFltEps) {
      if (distrib == 1) {
        D = d_gtr1(m, ax);
        alpha2 = 0.0625f;
      } else {
        D = d_gtr2(m, alpha2);
      }
      G1o = 2 / (1 + safe_sqrt(1 + alpha2 * (1 - cos_o * cos_o) / (cos_o * cos_o)));
      G1i = 2 / (1 + safe_sqrt(1 + alpha2 * (1 - cos_i * cos_i) / (cos_i * cos_i)));
    } else {
      float slope_x = -m.x / (m.z * ax);
      float slope_y = -m.y / (m.z * ay);
      float slope_len = 1 + slope_x * slope_x + slope_y * slope_y;
      float cm2 = m.z * m.z;
      float cm4 = cm2 * cm2;
      D = 1.f / ((slope_len * slope_len) * kPi * alpha2 * cm4);
      float tanO2 = (1.f - cos_o * cos_o) / (cos_o * cos_o);
      float aO2 = (wo.x * wo.x) * (ax * ax) + (wo.y * wo.y) * (ay * ay);
      aO2 /= wo.x * wo.x + wo.y * wo.y;
      G1o = 2 / (1 + safe_sqrt(1 + aO2 * tanO2));
      float tanI2 = (1 - cos_i * cos_i) / (cos_i * cos_i);
      float aI2 = (wi.x * wi.x) * (ax * ax) + (wi.y * wi.y) * (ay * ay);
      aI2 /= wi.x * wi.x + wi.y * wi.y;
      G1i = 2 / (1 + safe_sqrt(1 + aI2 * tanI2));
    }
    float G = G1o * G1i;
    float common = D * 0.25f / cos_o / cos_i;
    float f = G * common;
    if (distrib == 1) f = 0.25f * f;
    pdf = G1o * common;
    return f;
  }
  pdf = 0.f;
  return 0.f;
}
// :247-286.  Leaves wi untouched when wo is below the horizon or the microfacet faces away.
PB_HD void ggx_sample(V3 wo, float ax, float ay, float u0, float u1, V3& wi) {
  if (wo.z > 0.f) {
    V3 m = microfacet_sample_stretched(wo, ax, ay, u0, u1);
    float cos_mo = dot(m, wo);
    if (cos_mo > 0) wi = 2 * cos_mo * m - wo;
  }
}

// ------------------------------------------------------------------ hair (closure/energy-conserving-hair-bsdf.h)
struct HairBsdf {  // hair-shader.cc:8-17
  V3 sigma_a;
  float h;
  float v[4];
  float s, eta, alpha;
  V3 tints[4];
  float transparent_scale;
};

PB_HD float hair_safe_asin(float x) {  // :42-49
  float r = fastm::fasin(x);
  if (isnan(r)) return fastm::fasin(clampf(x, -1.0f, 1.0f));
  return r;
}
PB_HD float safe_log_i0(float x) {  // :92-170, improved-lobe branch ("+ 1.0f" outside the log: sic)
  x = fabsf(x);
  if (x < 7.5f) {
    float t = x * x / 4.0f;
    float f = 1.48095934745267240e-11f;
    f = f * t + 3.90565476357034480e-10f;
    f = f * t + 4.29455004657565361e-08f;
    f = f * t + 1.89645733877137904e-06f;
    f = f * t + 6.96166518788906424e-05f;
    f = f * t + 1.73560257755821695e-03f;
    f = f * t + 2.77785268558399407e-02f;
    f = f * t + 2.49999576572179639e-01f;
    f = f * t + 1.00000003928615375e+00f;
    return fastm::flog(t * f) + 1.0f;
  }
  float ix = 1.0f / x;
  float p = 1.31409251787866793e-01f;
  p = p * ix + 1.35614940793742178e-02f;
  p = p * ix + 2.91866904423115499e-02f;
  p = p * ix + 4.98327234176892844e-02f;
  p = p * ix + 3.98942651588301770e-01f;
  return x + 0.5f * fastm::flog(p * p * ix);
}
PB_HD float hair_mp(float sin_i, float cos_i, float sin_o, float cos_o, float v) {  // :172-202
  float ccv = cos_i * cos_o / v;
  float ssv = sin_i * sin_o / v;
  v = clampf(v, 1e-5f, 1e4f);
  return fastm::fexp(safe_log_i0(ccv) - ssv - 1.0f / v + fastm::flog(1.0f / v) -
                     fastm::flog(1.0f - fastm::fexp(-2.0f / v)));
}
PB_HD float fr_dielectric(float cos_i, float eta_i, float eta_t) {  // :205-229
  cos_i = clampf(cos_i, -1.0f, 1.0f);
  if (!(cos_i > 0.0f)) {
    float a = eta_i;
    eta_i = eta_t;
    eta_t = a;
    cos_i = fabsf(cos_i);
  }
  float sin_i = sqrtf(smax(0.0f, 1.0f - cos_i * cos_i));
  float sin_t = eta_i / eta_t * sin_i;
  if (sin_t >= 1.0f) return 1.0f;
  float cos_t = sqrtf(smax(0.0f, 1.0f - sin_t * sin_t));
  float r_parl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
  float r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
  return (r_parl * r_parl + r_perp * r_perp) * 0.5f;
}
PB_HD float logistic(float x, float s) {  // :257-262
  x = fabsf(x);
  float n = fastm::fexp(-x / s);
  return n / (s * sqr(1.0f + n));
}
PB_HD float logistic_cdf(float x, float s) { return 1.0f / (1.0f + fastm::fexp(-x / s)); }  // :264-266
PB_HD float trimmed_logistic(float x, float s, float a, float b) {                           // :268-271
  return logistic(x, s) / (logistic_cdf(b, s) - logistic_cdf(a, s));
}
PB_HD float hair_phi(int p, float gamma_o, float gamma_t) {  // :273-275
  return 2.0f * (float)p * gamma_t - 2.0f * gamma_o + (float)p * kPi;
}
PB_HD float hair_np(float phi, int p, float s, float gamma_o, float gamma_t) {  // :277-289
  float a = phi - hair_phi(p, gamma_o, gamma_t);
  float b = 2.0f * kPi;
  float dphi = a - floorf(a / b) * b;
  if (dphi >= kPi) dphi -= 2.0f * kPi;
  return trimmed_logistic(dphi, s, -kPi, kPi);
}

// everything of Eval (:295-362) / Sample (:419-476) that depends on (wo, bsdf) only
struct HairSetup {
  float sin_o, cos_o;
  float sin_o_crt[4], cos_o_crt[4];
  float phi_o, gamma_o, gamma_t;
  V3 ap[4];
  float ap_pdf[4];
};
PB_HD void hair_prepare(V3 wo, const HairBsdf& b, HairSetup& S) {
  S.sin_o = wo.x;
  S.cos_o = safe_sqrt(1.0f - sqr(S.sin_o));
  float s2k[3], c2k[3];
  fastm::fsincos(b.alpha, s2k[0], c2k[0]);
  for (int i = 1; i < 3; i++) {
    s2k[i] = 2.0f * s2k[i - 1] * c2k[i - 1];
    c2k[i] = sqr(c2k[i - 1]) - sqr(s2k[i - 1]);
  }
  float so = S.sin_o, co = S.cos_o;
  S.sin_o_crt[0] = so * c2k[1] - co * s2k[1];
  S.cos_o_crt[0] = co * c2k[1] + so * s2k[1];
  S.sin_o_crt[1] = so * c2k[0] + co * s2k[0];
  S.cos_o_crt[1] = co * c2k[0] - so * s2k[0];
  S.sin_o_crt[2] = so * c2k[2] + co * s2k[2];
  S.cos_o_crt[2] = co * c2k[2] - so * s2k[2];
  S.sin_o_crt[3] = so;
  S.cos_o_crt[3] = co;
  S.phi_o = fastm::fatan2(wo.z, wo.y);
  float sin_t = so / b.eta;
  float cos_t = safe_sqrt(1.f - sqr(sin_t));
  float etap = sqrtf(b.eta * b.eta - sqr(so)) / co;
  float sin_gt = b.h / etap;
  float cos_gt = safe_sqrt(1.0f - sqr(sin_gt));
  S.gamma_t = hair_safe_asin(sin_gt);
  float l = b.transparent_scale * 2.0f * cos_gt / cos_t;
  V3 T(fastm::fexp(-b.sigma_a.x * l), fastm::fexp(-b.sigma_a.y * l), fastm::fexp(-b.sigma_a.z * l));
  S.gamma_o = hair_safe_asin(b.h);
  {  // Ap :231-255
    float cos_go = safe_sqrt(1.0f - b.h * b.h);
    float f = fr_dielectric(co * cos_go, 1.0f, b.eta);
    S.ap[0] = V3(f);
    S.ap[1] = sqr(1.0f - f) * T;
    S.ap[2] = S.ap[1] * T * f;
    S.ap[3] = S.ap[2] * f * T / (V3(1.0f) - T * f);
    if (!is_finite(S.ap[3])) S.ap[3] = V3(0.0f);
  }
  float sum = 0.0f;
  for (int i = 0; i < 4; i++) sum = sum + rgb_to_y(S.ap[i]);
  for (int i = 0; i < 4; i++) S.ap_pdf[i] = rgb_to_y(S.ap[i]) / sum;
}
// lobe sum of Eval (:364-404) and Sample (:537-571): returns f*cos
PB_HD V3 hair_lobes(const HairSetup& S, const HairBsdf& b, float sin_i, float cos_i, float phi, float& pdf) {
  float pdfs[4];
  V3 ret(0.0f);
  for (int p = 0; p < 3; p++) {
    float mpnp = hair_mp(sin_i, cos_i, S.sin_o_crt[p], S.cos_o_crt[p], b.v[p]) * hair_np(phi, p, b.s, S.gamma_o, S.gamma_t);
    pdfs[p] = mpnp * S.ap_pdf[p];
    ret = ret + mpnp * S.ap[p] * b.tints[p];
  }
  float mpnp = hair_mp(sin_i, cos_i, S.sin_o, S.cos_o, b.v[3]) * (1.0f / (2.0f * kPi));
  pdfs[3] = mpnp * S.ap_pdf[3];
  ret = ret + mpnp * S.ap[3] * b.tints[3];
  pdf = 0.0f;
  if (!is_finite(ret)) return V3(0.0f);
  pdf = (((0.0f + pdfs[0]) + pdfs[1]) + pdfs[2]) + pdfs[3];
  if (!isfinite(pdf)) {
    pdf = 0.0f;
    return V3(0.0f);
  }
  return ret;
}
PB_HD V3 hair_eval(const HairSetup& S, V3 wi, const HairBsdf& b, float& pdf) {  // :295-405
  float sin_i = wi.x;
  float cos_i = safe_sqrt(1.0f - sqr(sin_i));
  float phi_i = fastm::fatan2(wi.z, wi.y);
  return hair_lobes(S, b, sin_i, cos_i, phi_i - S.phi_o, pdf);
}
PB_HD float sample_trimmed_logistic(float s, float a, float b, float u) {  // :407-417
  float T = logistic_cdf(b, s) - logistic_cdf(a, s);
  return -s * fastm::flog(1.0f / (u * T + 1.0f / (1.0f + fastm::fexp(-a / s))) - 1.0f);
}
PB_HD V3 hair_sample(const HairSetup& S, const HairBsdf& b, const float us[4], V3& wi, float& pdf) {  // :419-572
  int p;
  float u0 = us[0];
  for (p = 0; p < 3; p++) {
    if (u0 < S.ap_pdf[p]) break;
    u0 -= S.ap_pdf[p];
  }
  float vp = p == 0 ? b.v[0] : (p == 1 ? b.v[1] : (p == 2 ? b.v[2] : b.v[3]));
  float so = p == 0 ? S.sin_o_crt[0] : (p == 1 ? S.sin_o_crt[1] : (p == 2 ? S.sin_o_crt[2] : S.sin_o_crt[3]));
  float co = p == 0 ? S.cos_o_crt[0] : (p == 1 ? S.cos_o_crt[1] : (p == 2 ? S.cos_o_crt[2] : S.cos_o_crt[3]));
  float u1 = us[1], u2 = us[2];
  float u = 1.0f + vp * fastm::flog(u1 + (1.0f - u1) * fastm::fexp(-2.0f / vp));
  float sin_i = -u * so + safe_sqrt(1.0f - sqr(u)) * fastm::fcos(2.0f * kPi * u2) * co;
  float cos_i = safe_sqrt(1.0f - sqr(sin_i));
  float dphi;
  if (p < 3)
    dphi = hair_phi(p, S.gamma_o, S.gamma_t) + sample_trimmed_logistic(b.s, -kPi, kPi, us[3]);
  else
    dphi = 2.0f * kPi * us[3];
  float phi_i = S.phi_o + dphi;
  wi = V3(sin_i, cos_i * fastm::fcos(phi_i), cos_i * fastm::fsin(phi_i));
  return hair_lobes(S, b, sin_i, cos_i, dphi, pdf);
}

}  // namespace pb
