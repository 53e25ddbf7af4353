// dshade.h -- per-hit shading building blocks (device): surface reconstruction, light sampling,
// principled closure set, random-walk SSS coefficients, hair set-up.  file:line = pbrlab code.
#pragma once

#include "dtrace.h"

namespace pb {

enum : int { kFront = 0, kBack = 1, kAmbiguous = 2 };

// SurfaceInfo (shader/shader-utils.h:18-41), rebuilt from the 16-byte hit record
struct Surface {
  V3 pos, n_s, n_g;
  float tu, tv;  // Scene::FetchMeshTexcoord (scene.cc:230-249)
  int face;
  uint32_t material, lightrec, flags;
};

// TraceResultToSufaceInfo (shader-utils.h:131-164) + Scene::FetchMeshShadingNormal (scene.cc:210-228,
// mesh/triangle-mesh.cc:62-101) + EmbreeRayToTraceResult's normalisation (raytracer_impl.cc:221-240).
// Reads ONE 128-byte ShadeRec line (plus the 64-byte control-point slot for curves).
__device__ __forceinline__ Surface make_surface(const DScene& sc, V3 org, V3 dir, const Hit& h, uint32_t* instance_id = nullptr) {
  Surface s;
  const float4* r = reinterpret_cast<const float4*>(sc.shade + (h.slot & kHitSlotMask));
  // words: ng[0..2] matflags(3) ns_flat[4..6] lightrec(7) | n[8..16] (17) uv[18..23] | gid(24) instance(25) geom(26) prim(27)  (dscene.h)
  const float4 r0 = r[0], r1 = r[1];
  float4 r2 = make_float4(0.f, 0.f, 0.f, 0.f), r3 = r2, r4 = r2, r5 = r2;
  if (h.slot & kHitMore) r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5];  // (the hit code says so: the four loads leave with the first two)
  const uint32_t mf = __float_as_uint(r0.w);
  s.material = (mf & 0x00FFFFFFu) == 0x00FFFFFFu ? kNone : (mf & 0x00FFFFFFu), s.flags = mf >> 24, s.lightrec = __float_as_uint(r1.w);
  if (instance_id) *instance_id = sc.shade[h.slot & kHitSlotMask].instance_id;
  s.tu = 0.f, s.tv = 0.f;  // curves: (0,0) (scene.cc:243-245)
  if (s.flags & kSlotIsCurve) {
    float4 cp[4] = {r2, r3, r4, r5};  // the cubic's control points (xyzr)
    s.n_g = normalize_raw(bezier_tangent(cp, h.u));
    s.n_s = s.n_g;  // scene.cc:222-223
  } else {
    s.n_g = V3(r0.x, r0.y, r0.z);
    if (s.flags & kSlotHasNormals) {
      V3 n0(r2.x, r2.y, r2.z), n1(r2.w, r3.x, r3.y), n2(r3.z, r3.w, r4.x);
      s.n_s = vnormalize(lerp3(n0, n1, n2, h.u, h.v));
    } else {
      s.n_s = V3(r1.x, r1.y, r1.z);
    }
    if (s.flags & kSlotHasUV) {  // TriangleMesh::FetchTexcoord, triangle-mesh.cc:126-156 (only textured scenes pay)
      float w0 = 1.0f - h.u - h.v;
      s.tu = w0 * r4.z + h.u * r5.x + h.v * r5.z;
      s.tv = w0 * r4.w + h.u * r5.y + h.v * r5.w;
    } else {
      s.tu = h.u, s.tv = h.v;
    }
  }
  s.pos = org + h.t * dir;
  float dg = dot(dir, s.n_g), ds = dot(dir, s.n_s);
  s.face = (dg < 0.0f && ds < 0.0f) ? kFront : ((dg > 0.0f && ds > 0.0f) ? kBack : kAmbiguous);
  return s;
}

// Texture::FetchFloat3 (texture.cc:43-68) -> BilinearFilter with clamp addressing (image-utils.cc:99-167):
// px = width * u (no half-texel offset); channels the image lacks read as 0
__device__ __forceinline__ V3 texture_fetch3(const DScene& sc, uint32_t tex_id, float u, float v) {
  const TexDesc t = sc.textures[tex_id];
  float uu = smax(u, 0.0f);
  uu = smin(uu, 1.0f);
  float vv = smax(v, 0.0f);
  vv = smin(vv, 1.0f);
  const int width = (int)t.width, height = (int)t.height, stride = (int)t.channels;
  const float px = (float)t.width * uu;
  const float py = (float)t.height * vv;
  int x0 = (int)px, y0 = (int)py;
  x0 = x0 < width - 1 ? x0 : width - 1;
  x0 = x0 > 0 ? x0 : 0;
  y0 = y0 < height - 1 ? y0 : height - 1;
  y0 = y0 > 0 ? y0 : 0;
  const int x1 = ((x0 + 1) >= width) ? (width - 1) : (x0 + 1);
  const int y1 = ((y0 + 1) >= height) ? (height - 1) : (y0 + 1);
  const float dx = px - (float)x0;
  const float dy = py - (float)y0;
  const float w0 = (1.0f - dx) * (1.0f - dy), w1 = (1.0f - dx) * dy, w2 = dx * (1.0f - dy), w3 = dx * dy;
  const float* p = sc.tex_pixels + t.offset;
  const int i00 = stride * (y0 * width + x0), i01 = stride * (y0 * width + x1);
  const int i10 = stride * (y1 * width + x0), i11 = stride * (y1 * width + x1);
  float c[3];
#pragma unroll
  for (int i = 0; i < 3; i++)
    c[i] = (i < stride) ? p[i00 + i] * w0 + p[i10 + i] * w1 + p[i01 + i] * w2 + p[i11 + i] * w3 : 0.f;
  return V3(c[0], c[1], c[2]);
}

// std::lower_bound on a float CDF, clamped to the last entry (Q10).  (Cdf: a pointer to floats in global memory or in LDS)
template <typename Cdf>
__device__ __forceinline__ uint32_t cdf_lower_bound(Cdf cdf, uint32_t n, float u) {
  uint32_t lo = 0, len = n;
  while (len > 0) {
    uint32_t half = len >> 1;
    if (cdf[lo + half] < u) {
      lo = lo + half + 1;
      len = len - half - 1;
    } else {
      len = half;
    }
  }
  return lo < n ? lo : n - 1;
}

// DirectIllumination up to the shadow ray (shader-utils.h:166-190) with LightManager::SampleAllLight
// (light-manager.h:79-170: 4 draws, none when the scene has no light).  Returns true when a shadow ray
// has to be traced; dir/dist describe it, pdf_sigma/emission feed nee_contribution().
struct Nee {
  V3 dir, emission;
  float dist, pdf_sigma;
};
// The light tables as the sampling reads them: from the scene in global memory, or from a copy in LDS (k_shade_principled stages
// them when the scene has at most kLdsLights lights and light primitives: LDS reads stay off the vector-memory path).
constexpr uint32_t kLdsLights = 32;
constexpr uint32_t kLdsLightWords = kLdsLights * (1 + 2 + 1 + sizeof(LightRec) / 4);  // light cdf, heads, primitive cdf, records
typedef const __attribute__((address_space(3))) float* LdsFloats;
struct GlobalLightTables {
  const DScene& sc;
  __device__ __forceinline__ const float* cdf() const { return sc.light_cdf; }
  __device__ __forceinline__ LightHead head(uint32_t li) const { return sc.light_heads[li]; }
  __device__ __forceinline__ const float* prim_cdf(uint32_t first) const { return sc.lprim_cdf + first; }
  __device__ __forceinline__ float4 rec(uint32_t idx, int k) const { return reinterpret_cast<const float4*>(sc.lrecs + idx)[k]; }
};
struct LdsLightTables {  // words: [0, 32) light cdf | [32, 96) heads | [96, 128) primitive cdf | [128, ...) records of 20 words
  LdsFloats w;
  __device__ __forceinline__ LdsFloats cdf() const { return w; }
  __device__ __forceinline__ LightHead head(uint32_t li) const {
    LightHead h;
    h.first = __float_as_uint(w[kLdsLights + 2u * li]), h.count = __float_as_uint(w[kLdsLights + 2u * li + 1u]);
    return h;
  }
  __device__ __forceinline__ LdsFloats prim_cdf(uint32_t first) const { return w + 3u * kLdsLights + first; }
  __device__ __forceinline__ float4 rec(uint32_t idx, int k) const {
    LdsFloats r = w + 4u * kLdsLights + idx * (uint32_t)(sizeof(LightRec) / 4) + 4u * (uint32_t)k;
    return make_float4(r[0], r[1], r[2], r[3]);
  }
};
template <typename Tables>
__device__ __forceinline__ bool nee_sample_from(const Tables& lt, uint32_t num_lights, Rng& rng, V3 pos, V3 global_normal, bool hemisphere, Nee& n) {
  if (num_lights == 0) return false;
  float u0 = draw(rng);
  uint32_t li = cdf_lower_bound(lt.cdf(), num_lights, u0);
  LightHead head = lt.head(li);
  float u1 = draw(rng);
  uint32_t pi = cdf_lower_bound(lt.prim_cdf(head.first), head.count, u1);
  float u2 = draw(rng);
  float u3 = draw(rng);
  float bu, bv;
  triangle_uniform_sampler(u2, u3, bu, bv);
  const uint32_t ri = head.first + pi;
  float4 a = lt.rec(ri, 0), b = lt.rec(ri, 1), c = lt.rec(ri, 2), nn = lt.rec(ri, 3), e = lt.rec(ri, 4);
  V3 light_pos = lerp3(ld3(a), ld3(b), ld3(c), bu, bv);  // FetchLocalPosition, triangle-mesh.cc:102-112
  V3 light_normal = ld3(nn);
  float pdf = a.w;
  n.emission = ld3(e);
  n.dir = vnormalize(light_pos - pos);
  n.dist = length(pos - light_pos);
  float wl_dot_nl = -dot(n.dir, light_normal);
  float wl_dot_np = dot(n.dir, global_normal);
  n.pdf_sigma = fabsf(pdf * n.dist * n.dist / (wl_dot_nl * wl_dot_np));
  return (!hemisphere) || (wl_dot_nl > 0.0f && wl_dot_np > 0.0f);
}
// lds_lights: the light tables staged in LDS (LdsLightTables layout), or null
__device__ __forceinline__ bool nee_sample(const DScene& sc, Rng& rng, V3 pos, V3 global_normal, bool hemisphere, Nee& n,
                                           const float* lds_lights = nullptr) {
  if (lds_lights) {
    const LdsLightTables lt = {(LdsFloats)lds_lights};
    return nee_sample_from(lt, sc.num_lights, rng, pos, global_normal, hemisphere, n);
  }
  const GlobalLightTables lt = {sc};
  return nee_sample_from(lt, sc.num_lights, rng, pos, global_normal, hemisphere, n);
}
// shader-utils.h:195-208
__device__ __forceinline__ V3 nee_contribution(const Nee& n, V3 bsdf_f, float bsdf_pdf) {
  float w = power_heuristic(n.pdf_sigma, bsdf_pdf);
  return bsdf_f * n.emission * w / n.pdf_sigma;
}

// ------------------------------------------------------------------ principled closure set
// SpecularColor (cycles-principled-shader.cc:54-61)
PB_HD V3 specular_color_fn(V3 wi, V3 wo, V3 color, float ior) {
  V3 h = vnormalize(wi + wo);
  float f0 = fresnel_dielectric_cos(1.0f, ior);
  float fh = (fresnel_dielectric_cos(dot(h, wo), ior) - f0) / (1.0f - f0);
  return color * (1.f - fh) + V3(fh);
}
struct SampleWeight {
  float diffuse, subsurface, specular, clearcoat;
};
// FetchClosureSampleWeight (:63-112), Q7
PB_HD SampleWeight closure_sample_weight(V3 wo, const PrincipledBsdf& b) {
  SampleWeight w;
  V3 refl(-wo.x, -wo.y, wo.z);
  w.diffuse = b.enable_diffuse ? rgb_to_y(b.diffuse_weight) : 0.f;
  w.subsurface = b.enable_subsurface ? rgb_to_y(b.subsurface_weight) : 0.f;
  w.specular = b.enable_specular ? rgb_to_y(b.specular_weight * specular_color_fn(refl, wo, b.specular_color, b.ior)) : 0.f;
  w.clearcoat =
      b.enable_clearcoat ? rgb_to_y(b.clearcoat_weight * specular_color_fn(refl, wo, b.clearcoat_color, b.clearcoat_ior)) : 0.f;
  float sum = 0.0f;
  sum += w.diffuse;
  sum += w.subsurface;
  sum += w.specular;
  sum += w.clearcoat;
  w.diffuse /= sum;
  w.subsurface /= sum;
  w.specular /= sum;
  w.clearcoat /= sum;
  if (!isfinite(w.diffuse)) w.diffuse = 0.f;
  if (!isfinite(w.subsurface)) w.subsurface = 0.f;
  if (!isfinite(w.specular)) w.specular = 0.f;
  if (!isfinite(w.clearcoat)) w.clearcoat = 0.f;
  return w;
}
// EvalBsdf (:114-155)
PB_HD void eval_bsdf(V3 wi, V3 wo, const PrincipledBsdf& b, const SampleWeight& w, V3& f, float& pdf) {
  f = V3(0.0f);
  pdf = 0.0f;
  if (b.enable_diffuse) {
    float p;
    float v = lambert_eval(wi, p);
    f = f + b.diffuse_weight * v;
    pdf += w.diffuse * p;
  }
  if (b.enable_specular) {
    float p;
    float v = ggx_eval(wi, wo, b.alpha_x, b.alpha_y, 2, p);
    f = f + b.specular_weight * specular_color_fn(wi, wo, b.specular_color, b.ior) * v;
    pdf += w.specular * p;
  }
  if (b.enable_clearcoat) {
    float p;
    float v = ggx_eval(wi, wo, b.clearcoat_alpha_x, b.clearcoat_alpha_y, 1, p);
    f = f + b.clearcoat_weight * specular_color_fn(wi, wo, b.clearcoat_color, b.clearcoat_ior) * v;
    pdf += w.clearcoat * p;
  }
}
// SampleBsdf's closure pick (:169-242): 0 diffuse, 1 subsurface, 2 specular, 3 clearcoat (also the
// fall-through when every weight is 0, Q7)
PB_HD int pick_closure(float select, const SampleWeight& w) {
  if (select < w.diffuse) return 0;
  if (select < w.diffuse + w.subsurface) return 1;
  if (select < w.diffuse + w.subsurface + w.specular) return 2;
  return 3;
}

// ------------------------------------------------------------------ ParamToBsdf, hoisted to commit time
// random-walk-sss.h:35-104 (BssrdfSetup with burley radius, mfp scaling, eq. 5)
PB_HD float burley_fitting5(float A) { return 1.85f - A + 7.0f * fabsf((A - 0.8f) * (A - 0.8f) * (A - 0.8f)); }
PB_HD void bssrdf_setup(V3& weight, V3 albedo, V3& radius, V3& diffuse_weight) {
  diffuse_weight = V3(0.0f);
  const float kMinRadius = 1e-8f;
  float kd[3] = {0, 0, 0}, w[3] = {weight.x, weight.y, weight.z}, r[3] = {radius.x, radius.y, radius.z};
  int channels = 3;
  for (int i = 0; i < 3; i++)
    if (r[i] < kMinRadius) {
      kd[i] = w[i];
      w[i] = 0.f;
      r[i] = 0.f;
      channels--;
    }
  weight = V3(w[0], w[1], w[2]);
  radius = V3(r[0], r[1], r[2]);
  if (channels < 3) diffuse_weight = V3(kd[0], kd[1], kd[2]);
  if (channels > 0) {
    V3 l = 0.25f * (1.0f / kPi) * radius;
    V3 s(burley_fitting5(albedo.x), burley_fitting5(albedo.y), burley_fitting5(albedo.z));
    radius = l / s;
  }
}
// cycles-principled-shader.cc:244-412; base_color / subsurface_color already resolved (parameter or texture, :281-301)
PB_HD PrincipledBsdf param_to_bsdf(const PrincipledParam& m, V3 base_color, V3 subsurface_color) {
  const V3 weight(1.f);
  float subsurface = m.subsurface;
  V3 subsurface_radius(m.subsurface_radius[0], m.subsurface_radius[1], m.subsurface_radius[2]);
  const float cutoff = kEps;
  PrincipledBsdf b = default_bsdf();
  float diffuse_w = (1.0f - saturate(m.metallic)) * (1.0f - saturate(m.transmission));
  float final_transmission = saturate(m.transmission) * (1.0f - saturate(m.metallic));
  float specular_w = (1.0f - final_transmission);
  V3 mixed = subsurface_color * subsurface + base_color * (1.0f - subsurface);
  if (average(mixed) > cutoff) {
    if (subsurface < cutoff && diffuse_w > cutoff) {
      b.enable_diffuse = 1;
      b.diffuse_weight = weight * base_color * diffuse_w;
    } else if (subsurface > cutoff) {
      b.enable_subsurface = 1;
      b.subsurface_weight = weight * mixed * diffuse_w;
      b.subsurface_albedo = mixed;
      b.subsurface_radius = subsurface_radius * subsurface;
      V3 add_diffuse(0.f);
      bssrdf_setup(b.subsurface_weight, b.subsurface_albedo, b.subsurface_radius, add_diffuse);
      if (!is_black(add_diffuse)) {
        b.enable_diffuse = 1;
        b.diffuse_weight = b.diffuse_weight + add_diffuse;
      }
    }
  }
  if (specular_w > cutoff && (m.specular > cutoff || m.metallic > cutoff)) {
    b.enable_specular = 1;
    b.specular_weight = weight * specular_w;
    b.ior = (2.0f / (1.0f - safe_sqrt(0.08f * m.specular))) - 1.0f;
    float aspect = safe_sqrt(1.0f - m.anisotropic * 0.9f);
    float r2 = m.roughness * m.roughness;
    b.alpha_x = r2 / aspect;
    b.alpha_y = r2 * aspect;
    float y = rgb_to_y(base_color);
    V3 rho_tint = y > 0.0f ? base_color / y : V3(0.0f);
    V3 rho_specular = lerp(V3(1.0f), rho_tint, m.specular_tint);
    b.specular_color = lerp(0.08f * m.specular * rho_specular, base_color, m.metallic);
  }
  if (m.clearcoat > cutoff) {
    b.enable_clearcoat = 1;
    b.clearcoat_weight = V3(0.25f * m.clearcoat);
    b.clearcoat_alpha_x = m.clearcoat_roughness * m.clearcoat_roughness;
    b.clearcoat_alpha_y = m.clearcoat_roughness * m.clearcoat_roughness;
    b.clearcoat_color = V3(0.04f);
    b.clearcoat_ior = 1.5f;
  }
  return b;
}

PB_HD PrincipledBsdf param_to_bsdf(const PrincipledParam& m) {
  return param_to_bsdf(m, V3(m.base_color[0], m.base_color[1], m.base_color[2]),
                       V3(m.subsurface_color[0], m.subsurface_color[1], m.subsurface_color[2]));
}

PB_HD float pow_n(float v, int n) {  // pbrlab_math.h:40-55
  if (n == 0) return 1.f;
  if (n == 1) return v;
  float h = pow_n(v, n / 2);
  return h * h * pow_n(v, n & 1);
}
// hair-shader.cc:19-151 : everything of ParamToBsdf that does not depend on the hit (h is filled per hit)
PB_HD HairBsdf hair_param_to_bsdf(const HairParam& m) {
  HairBsdf b;
  if (m.coloring_hair == 0) {
    float bn = m.azimuthal_roughness;  // Q12: sigma_a from RGB uses beta_n
    float c[3] = {m.base_color[0], m.base_color[1], m.base_color[2]}, r[3];
    for (int i = 0; i < 3; i++)
      r[i] = sqr(fastm::flog(c[i]) / (5.969f - 0.215f * bn + 2.532f * sqr(bn) - 10.73f * pow_n(bn, 3) +
                                      5.574f * pow_n(bn, 4) + 0.245f * pow_n(bn, 5)));
    b.sigma_a = V3(r[0], r[1], r[2]);
  } else {
    const float random_value = 0.5f;
    float factor = 1.f + 2.f * (random_value - 0.5f);
    float melanin = clampf(m.melanin, 0.0f, 1.0f) * factor;
    float redness = clampf(m.melanin_redness, 0.0f, 1.0f);
    melanin = -fastm::flog(smax(1.0f - melanin, 0.0001f));
    float eu = melanin * (1.0f - redness);
    float pheo = melanin * redness;
    b.sigma_a = V3(smax(0.0f, eu * 0.506f + pheo * 0.343f), smax(0.0f, eu * 0.841f + pheo * 0.733f),
                   smax(0.0f, eu * 1.653f + pheo * 1.924f));
  }
  b.h = 0.f;
  float bm = m.roughness;
  b.v[0] = sqr(0.726f * bm + 0.812f * sqr(bm) + 3.7f * pow_n(bm, 20));
  b.v[1] = 0.25f * b.v[0];
  b.v[2] = 4.0f * b.v[0];
  b.v[3] = b.v[2];
  float bn2 = sqr(m.azimuthal_roughness);
  b.s = sqrtf(kPi / 8.0f) * (0.265f * m.azimuthal_roughness + 1.194f * bn2 + 5.372f * pow_n(bn2, 11));
  b.eta = m.ior;
  b.alpha = m.shift * kPi / 180.f;
  b.tints[0] = V3(m.specular_tint[0], m.specular_tint[1], m.specular_tint[2]);
  b.tints[1] = V3(m.transmission_tint[0], m.transmission_tint[1], m.transmission_tint[2]);
  b.tints[2] = V3(m.second_specular_tint[0], m.second_specular_tint[1], m.second_specular_tint[2]);
  b.tints[3] = V3(1.f);
  b.transparent_scale = 1.f;
  return b;
}

// ------------------------------------------------------------------ random-walk SSS pieces (random-walk-sss.h)
PB_HD void scattering_from_albedo(float A, float d, float& sigma_t, float& sigma_s) {  // :111-122
  float a = 1.0f - f_exp(A * (-5.09406f + A * (2.61188f - A * 4.31805f)));
  float s = 1.9f - A + 3.5f * sqr(A - 0.8f);
  sigma_t = 1.0f / smax(d * s, 1e-16f);
  sigma_s = sigma_t * a;
}
// the medium a closure set describes and the walk's first throughput (random-walk-sss.h:111-122, 243-258)
PB_HD void medium_coefficients(const PrincipledBsdf& b, V3& sigt, V3& sigs, V3& wthr) {
  scattering_from_albedo(b.subsurface_albedo.x, b.subsurface_radius.x, sigt.x, sigs.x);
  scattering_from_albedo(b.subsurface_albedo.y, b.subsurface_radius.y, sigt.y, sigs.y);
  scattering_from_albedo(b.subsurface_albedo.z, b.subsurface_radius.z, sigt.z, sigs.z);
  wthr = safe_divide_spectrum(b.subsurface_weight, b.subsurface_albedo);
}
// SampleScatterDistance + SampleChannel (:141-188)
// the channel pdf is a function of (walk throughput, sigma_s, sigma_t) alone: the step that consumes it recomputes it from the
// stored throughput instead of keeping it in the path state
// albedo: null, or safe_divide_spectrum(sigma_s, sigma_t) computed before (a constant of the walk)
PB_HD V3 scatter_channel_pdf(V3 throughput, V3 sigma_s, V3 sigma_t, const V3* albedo_in = nullptr) {
  V3 albedo = albedo_in ? *albedo_in : safe_divide_spectrum(sigma_s, sigma_t);
  V3 w(fabsf(throughput.x * albedo.x), fabsf(throughput.y * albedo.y), fabsf(throughput.z * albedo.z));
  float sum = w.x + w.y + w.z;
  if (sum > 0.0f) return V3(w.x / sum, w.y / sum, w.z / sum);
  return V3(1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 3.0f);
}
PB_HD float sample_scatter_distance(V3 throughput, V3 sigma_s, V3 sigma_t, float u0, float u1, V3& channel_pdf, const V3* albedo_in = nullptr) {
  channel_pdf = scatter_channel_pdf(throughput, sigma_s, sigma_t, albedo_in);
  float st = (u0 < channel_pdf.x) ? sigma_t.x : ((u0 < channel_pdf.x + channel_pdf.y) ? sigma_t.y : sigma_t.z);
  return -f_log(1.0f - u1) / st;
}
PB_HD V3 attenuate_transmission(V3 sigma_t, float d) {  // :190-198
  return V3(f_exp(-sigma_t.x * d), f_exp(-sigma_t.y * d), f_exp(-sigma_t.z * d));
}

}  // namespace pb
