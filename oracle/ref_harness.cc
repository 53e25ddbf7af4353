// ref_harness.cc -- ORACLE support (test infrastructure only).
//
// A driver, written for this repo, that #includes the REFERENCE's own header-only leaf code where
// it lies under /root/reference/src and exports it through a C ABI, so tests can compare the C
// restatement (pbr_oracle.c) with the reference's actual arithmetic.  Built by oracle/Makefile into
// oracle/_ref/libref_leaf.so (git-ignored; travels to the GPU box as a prebuilt .so).
//
// Only reference files whose includes resolve inside /root/reference/src are used:
//   random/rng.h, sampler/sampling-utils.h, pbrlab_math.h, pbrlab-util.h, type.h (+ nanort.h),
//   closure/{lambert,closure-util,microfacet-ggx}.h, closure/energy‐conserving-hair-bsdf.h,
//   matrix.{h,cc}, render-tile.{h,cc}, curve-util.{h,cc}, mesh/triangle-mesh.{h,cc}, mesh/attribute.h,
//   texture.{h,cc}, image-utils.{h,cc}.
// Anything that needs mpark/variant.hpp or embree4/rtcore.h (shaders, Scene, LightManager, render.cc,
// raytracer) is NOT built: those headers are absent and no stand-ins are written.
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "closure/closure-util.h"
#include "closure/energy‐conserving-hair-bsdf.h"
#include "closure/lambert.h"
#include "closure/microfacet-ggx.h"
#include "curve-util.h"
#include "matrix.h"
#include "mesh/triangle-mesh.h"
#include "pbrlab-util.h"
#include "pbrlab_math.h"
#include "random/rng.h"
#include "render-tile.h"
#include "sampler/sampling-utils.h"
#include "texture.h"
#include "image-utils.h"
#include "type.h"

using namespace pbrlab;

extern "C" {

void ref_rng(uint64_t initstate, uint64_t initseq, uint32_t n, float* out) {
  RNG rng(initstate, initseq);
  for (uint32_t i = 0; i < n; i++) out[i] = rng.Draw();
}

float ref_fastmath(int op, float x, float y2) {
  float s, c;
  switch (op) {
    case 0: return fast_math::FastSin(x);
    case 1: return fast_math::FastCos(x);
    case 2: return fast_math::FastExp(x);
    case 3: return fast_math::FastLog(x);
    case 4: return fast_math::FastAtan2(x, y2);
    case 5: return fast_math::FastAsin(x);
    case 6: return fast_math::FastExp2(x);
    case 7: return fast_math::FastLog2(x);
    case 8: fast_math::FastSincos(x, &s, &c); return s;
    case 9: fast_math::FastSincos(x, &s, &c); return c;
  }
  return 0.f;
}

float ref_fresnel(float c, float eta) { return FresnelDielectricCos(c, eta); }
float ref_power_heuristic(float a, float b) { return PowerHeuristicWeight(a, b); }

void ref_lambert_sample(float u0, float u1, float out[5]) {
  float3 wi;
  float pdf = 0.f;
  const float f = LambertBrdfSample(float3(0.f, 0.f, 1.f), {u0, u1}, &wi, &pdf);
  out[0] = wi[0], out[1] = wi[1], out[2] = wi[2], out[3] = f, out[4] = pdf;
}

void ref_ggx_eval(const float wi[3], const float wo[3], float ax, float ay, int distrib, float out[2]) {
  float pdf = 0.f;
  out[0] = MicrofacetGGXBsdfPdf(float3(wi), float3(wo), ax, ay, distrib, &pdf);
  out[1] = pdf;
}

void ref_ggx_sample(const float wo[3], float ax, float ay, float u0, float u1, int distrib, float out[5]) {
  float3 wi(0.f);
  float pdf = 0.f;
  const float f = MicrofacetGGXSample(float3(wo), ax, ay, {u0, u1}, false, distrib, &wi, &pdf);
  out[0] = wi[0], out[1] = wi[1], out[2] = wi[2], out[3] = f, out[4] = pdf;
}

// params: h, v0..v3, s, sigma_a[3], eta, alpha, tints[12], transparent_scale (23 floats)
static void unpack_hair(const float* p, float* h, std::array<float, 4>* v, float* s, float3* sigma_a, float* eta,
                        float* alpha, std::array<float3, 4>* tints, float* ts) {
  *h = p[0];
  for (int i = 0; i < 4; i++) (*v)[size_t(i)] = p[1 + i];
  *s = p[5];
  *sigma_a = float3(p[6], p[7], p[8]);
  *eta = p[9];
  *alpha = p[10];
  for (int i = 0; i < 4; i++) (*tints)[size_t(i)] = float3(p[11 + i * 3], p[12 + i * 3], p[13 + i * 3]);
  *ts = p[22];
}

void ref_hair_eval(const float wi[3], const float wo[3], const float* params, float out[4]) {
  float h, s, eta, alpha, ts;
  std::array<float, 4> v;
  std::array<float3, 4> tints;
  float3 sigma_a;
  unpack_hair(params, &h, &v, &s, &sigma_a, &eta, &alpha, &tints, &ts);
  float pdf = 0.f;
  const float3 f =
      hair_bsdf::EnergyConservingHairBsdfCosPdf(float3(wi), float3(wo), h, v, s, sigma_a, eta, alpha, tints, ts, &pdf);
  out[0] = f[0], out[1] = f[1], out[2] = f[2], out[3] = pdf;
}

void ref_hair_sample(const float wo[3], const float* params, const float us[4], float out[7]) {
  float h, s, eta, alpha, ts;
  std::array<float, 4> v;
  std::array<float3, 4> tints;
  float3 sigma_a;
  unpack_hair(params, &h, &v, &s, &sigma_a, &eta, &alpha, &tints, &ts);
  float3 wi(0.f);
  float pdf = 0.f;
  const float3 f = hair_bsdf::EnergyConservingHairSample(float3(wo), h, v, s, sigma_a, eta, alpha, tints, ts,
                                                         {us[0], us[1], us[2], us[3]}, &wi, &pdf);
  out[0] = wi[0], out[1] = wi[1], out[2] = wi[2], out[3] = f[0], out[4] = f[1], out[5] = f[2], out[6] = pdf;
}

void ref_uniform_sphere(float u1, float u2, float out[3]) {
  const float3 v = UniformSampleSphere(u1, u2);
  out[0] = v[0], out[1] = v[1], out[2] = v[2];
}
// UniformSampleSphere(rng.Draw(), rng.Draw()) exactly as random-walk-sss.h:296 spells it: exposes the
// compiler's argument evaluation order.
void ref_uniform_sphere_from_rng(uint64_t initstate, uint64_t initseq, float out[3]) {
  RNG rng(initstate, initseq);
  const float3 v = UniformSampleSphere(rng.Draw(), rng.Draw());
  out[0] = v[0], out[1] = v[1], out[2] = v[2];
}
void ref_triangle_sampler(float u1, float u2, float out[2]) {
  const auto p = TriangleUniformSampler(u1, u2);
  out[0] = p.first, out[1] = p.second;
}
void ref_cosine_hemisphere(float u1, float u2, float out[3]) {
  const float3 v = CosineSampleHemisphere(u1, u2);
  out[0] = v[0], out[1] = v[1], out[2] = v[2];
}

// Matrix::MultV with the rows of a 3x3 (4th row/col = identity part)
void ref_mult_v(const float v[3], const float rows[9], float out[3]) {
  float m[4][4] = {{rows[0], rows[1], rows[2], 0.f},
                   {rows[3], rows[4], rows[5], 0.f},
                   {rows[6], rows[7], rows[8], 0.f},
                   {0.f, 0.f, 0.f, 1.f}};
  Matrix::MultV(v, m, out);
}

void ref_create_tiles(uint32_t width, uint32_t height, uint32_t* out, uint32_t* num_tiles) {
  std::vector<std::unique_ptr<RenderTile>> tiles;
  CreateTiles(width, height, 64, 64, &tiles);
  *num_tiles = uint32_t(tiles.size());
  if (out)
    for (size_t i = 0; i < tiles.size(); i++) {
      out[i * 4 + 0] = tiles[i]->sx, out[i * 4 + 1] = tiles[i]->tx;
      out[i * 4 + 2] = tiles[i]->sy, out[i * 4 + 3] = tiles[i]->ty;
    }
}

int ref_to_cubic_bezier(const float* cvs, const float* radii, uint32_t n, float* out_xyzr) {
  std::vector<float> c(cvs, cvs + 3 * n), r(radii, radii + n), bv, br;
  if (!ToCubicBezierCurve(c, r, &bv, &br)) return -1;
  for (size_t i = 0; i < br.size(); i++) {
    out_xyzr[i * 4 + 0] = bv[i * 3 + 0], out_xyzr[i * 4 + 1] = bv[i * 3 + 1], out_xyzr[i * 4 + 2] = bv[i * 3 + 2];
    out_xyzr[i * 4 + 3] = br[i];
  }
  return int(br.size() / 4);
}

// TriangleMesh fetches (mesh/triangle-mesh.cc): what=0 shading normal, 1 geometry normal, 2 local position,
// 3 face area (out[0])
void ref_triangle_fetch(const float* vertices_xyzw, uint32_t nv, const float* normals_xyzw, uint32_t nn,
                        const uint32_t* vid, const uint32_t* nid, uint32_t nfaces, uint32_t prim, float u, float v,
                        int what, float out[3]) {
  std::shared_ptr<Attribute> attr(new Attribute());
  attr->vertices.assign(vertices_xyzw, vertices_xyzw + 4 * nv);
  attr->normals.assign(normals_xyzw, normals_xyzw + 4 * nn);
  std::vector<uint32_t> vids(vid, vid + 3 * nfaces), nids;
  if (nid) nids.assign(nid, nid + 3 * nfaces);
  TriangleMesh mesh("m", attr, vids, nids, {}, {});
  float3 r(0.f);
  if (what == 0) r = mesh.FetchShadingNormal(prim, u, v);
  if (what == 1) r = mesh.FetchGeometryNormal(prim);
  if (what == 2) r = mesh.FetchLocalPosition(prim, u, v);
  if (what == 3) r = float3(mesh.FetchFaceArea(prim));
  out[0] = r[0], out[1] = r[1], out[2] = r[2];
}

// Texture::FetchFloat3 (texture.cc:65-67) -> BilinearFilter (image-utils.cc:99-167)
void ref_texture_fetch(const float* pixels, uint32_t width, uint32_t height, uint32_t channels, float u, float v,
                       float out[3]) {
  Texture tex(std::vector<float>(pixels, pixels + size_t(width) * height * channels), width, height, channels, "t");
  tex.FetchFloat3(u, v, out);
}
// LinerTosRGB / SrgbToLiner (image-utils.cc:10-38): the CLI's output stage (pc/pbrlab-cli.cc:56)
float ref_linear_to_srgb(float c) { return LinerTosRGB(c); }
float ref_srgb_to_linear(float c) { return SrgbToLiner(c); }

float ref_spectrum_norm(const float c[3]) { return SpectrumNorm(float3(c)); }
float ref_rgb_to_y(const float c[3]) { return RgbToY(float3(c)); }

}  // extern "C"
