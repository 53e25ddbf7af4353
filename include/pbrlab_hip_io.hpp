// pbrlab_hip_io.hpp -- header-only C++ shim with pbrlab's I/O names over the C ABI of libpbrhip_io (pbrhip_io.h): what a
// pbrlab caller includes instead of io/triangle-mesh-io.h, io/curve-mesh-io.h, io/image-io.h and image-utils.h.  With it and
// pbrlab_hip.hpp the reference's own callers -- pc/pc-common.cc and pc/pbrlab-cli.cc -- compile UNMODIFIED against this
// repository (tests/cpp/fwd/ holds one-line forwarding headers with the reference's file names; the `ref_cli` build recipe
// of the test infrastructure builds the reference's CLI that way; tests/test_reference_callers.py).
//
//   pbrlab::io::LoadTriangleMeshFromObj          src/io/triangle-mesh-io.h:14-17  (triangle-mesh-io.cc:214-325)
//   pbrlab::io::LoadCurveMeshAsCubicBezierCurve  src/io/curve-mesh-io.h:13-21     (curve-mesh-io.cc:32-138)
//   pbrlab::io::LoadImageFromFile / WritePNG     src/io/image-io.h:21-63          (image-io.cc:98-224)
//   pbrlab::SrgbToLiner / LinerTosRGB / LinerToSrgb  src/image-utils.h:9-47       (image-utils.cc:8-91)
#ifndef PBRLAB_HIP_IO_HPP_
#define PBRLAB_HIP_IO_HPP_

#include <cmath>

#include "pbrhip_io.h"
#include "pbrlab_hip.hpp"

namespace pbrlab {

// the sRGB transfer functions of image-utils.cc:8-38 (IEC 61966-2-1, evaluated in T like there)
template <typename T>
inline T SrgbToLiner(const T c_srgb) {
  if (c_srgb <= T(0.04045)) return c_srgb / T(12.92);
  return std::pow((c_srgb + T(0.055)) / (T(1.0) + T(0.055)), T(2.4));
}
template <typename T>
inline T LinerTosRGB(const T c_liner) {
  if (c_liner <= T(0.0031308)) return T(12.92) * c_liner;
  return std::pow((T(1.0) + T(0.055)) * c_liner, T(1.0 / 2.4)) - T(0.055);
}
namespace detail {
template <typename T, typename F>
inline void MapColorChannels(const std::vector<T>& src, size_t width, size_t height, size_t channels, std::vector<T>* out, F f) {
  assert(src.size() == width * height * channels);
  std::vector<T> res(width * height * channels);  // (src and out may be the same vector: pbrlab-cli.cc:56)
  for (size_t i = 0; i < res.size(); ++i) res[i] = (i % channels) < 3 ? f(src[i]) : src[i];  // alpha passes through
  out->swap(res);
}
}  // namespace detail
template <typename T>
inline void SrgbToLiner(const std::vector<T>& src, const size_t width, const size_t height, const size_t channels, std::vector<T>* out) {
  detail::MapColorChannels(src, width, height, channels, out, [](T c) { return SrgbToLiner(c); });
}
template <typename T>
inline void LinerToSrgb(const std::vector<T>& src, const size_t width, const size_t height, const size_t channels, std::vector<T>* out) {
  detail::MapColorChannels(src, width, height, channels, out, [](T c) { return LinerTosRGB(c); });
}

namespace io {

// io/triangle-mesh-io.h:14-17.  One TriangleMesh per OBJ shape, all sharing one Attribute; material ids index
// `material_params`, texture ids inside them index `textures` (this file's lists: the caller renumbers, pc-common.cc:115-139).
inline bool LoadTriangleMeshFromObj(const std::string& filename, std::vector<TriangleMesh>* meshes,
                                    std::vector<MaterialParameter>* material_params, std::vector<Texture>* textures) {
  pbrio_obj* o = nullptr;
  if (pbrio_obj_load(filename.c_str(), &o) != PBRHIP_OK) {
    std::cerr << pbrio_last_error() << std::endl;
    return false;
  }
  if (const char* w = pbrio_obj_warnings(o))
    if (*w) std::cerr << w << std::endl;
  auto attr = std::make_shared<Attribute>();
  std::vector<float>* const dst[3] = {&attr->vertices, &attr->normals, &attr->texcoords};
  for (int which = 0; which < 3; ++which) {
    const float* p = nullptr;
    const size_t n = pbrio_obj_attribute(o, which, &p);
    dst[which]->assign(p, p + n);
  }
  meshes->clear();
  for (uint32_t s = 0; s < pbrio_obj_num_shapes(o); ++s) {
    std::vector<uint32_t> ids[4];
    for (int which = 0; which < 4; ++which) {
      const uint32_t* p = nullptr;
      const size_t n = pbrio_obj_shape_ids(o, s, which, &p);
      ids[which].assign(p, p + n);
    }
    meshes->emplace_back(pbrio_obj_shape_name(o, s), attr, ids[0], ids[1], ids[2], ids[3]);
  }
  for (uint32_t t = 0; t < pbrio_obj_num_textures(o); ++t) {
    const float* px = nullptr;
    uint32_t w = 0, h = 0, c = 0;
    const char* name = "";
    pbrio_obj_texture(o, t, &px, &w, &h, &c, &name);
    textures->emplace_back(std::vector<float>(px, px + size_t(w) * h * c), w, h, c, name ? name : "");
  }
  for (uint32_t m = 0; m < pbrio_obj_num_materials(o); ++m) {
    pbrhip_principled_param pr;
    const char* name = "";
    pbrio_obj_material(o, m, &pr, &name);
    CyclesPrincipledBsdfParameter cp;
    cp.base_color = float3(pr.base_color), cp.subsurface = pr.subsurface, cp.subsurface_radius = float3(pr.subsurface_radius);
    cp.subsurface_color = float3(pr.subsurface_color), cp.metallic = pr.metallic, cp.specular = pr.specular;
    cp.specular_tint = pr.specular_tint, cp.roughness = pr.roughness, cp.anisotropic = pr.anisotropic;
    cp.anisotropic_rotation = pr.anisotropic_rotation, cp.sheen = pr.sheen, cp.sheen_tint = pr.sheen_tint;
    cp.clearcoat = pr.clearcoat, cp.clearcoat_roughness = pr.clearcoat_roughness, cp.ior = pr.ior;
    cp.transmission = pr.transmission, cp.transmission_roughness = pr.transmission_roughness;
    cp.base_color_tex_id = pr.base_color_tex_id, cp.subsurface_color_tex_id = pr.subsurface_color_tex_id;
    cp.name = name ? name : "";
    material_params->emplace_back(cp);
  }
  pbrio_obj_free(o);
  return true;
}

// io/curve-mesh-io.h:13-17
inline bool LoadCurveMeshAsCubicBezierCurve(const std::string& filepath, const bool memory_saving_mode,
                                            std::vector<float>* vertices_thickness, std::vector<uint32_t>* indices) {
  pbrio_curves* c = nullptr;
  if (pbrio_curves_load(filepath.c_str(), memory_saving_mode ? 1 : 0, &c) != PBRHIP_OK) {
    std::cerr << pbrio_last_error() << std::endl;
    return false;
  }
  const float* v = nullptr;
  const uint32_t* i = nullptr;
  const size_t nv = pbrio_curves_vertices(c, &v), ni = pbrio_curves_indices(c, &i);
  vertices_thickness->assign(v, v + nv);
  indices->assign(i, i + ni);
  pbrio_curves_free(c);
  return true;
}
// io/curve-mesh-io.h:18-20 (curve-mesh-io.cc:121-136: the first loader's result is not looked at, the mesh is named after
// the file and every segment starts without a material)
inline bool LoadCurveMeshAsCubicBezierCurve(const std::string& filepath, const bool memory_saving_mode, CubicBezierCurveMesh* curve_mesh) {
  auto attr = std::make_shared<CurveAttribute>();
  std::vector<uint32_t> indices;
  LoadCurveMeshAsCubicBezierCurve(filepath, memory_saving_mode, &attr->vertices, &indices);
  const std::vector<uint32_t> material_ids(indices.size(), uint32_t(-1));
  *curve_mesh = CubicBezierCurveMesh(filepath, attr, indices, material_ids);
  return curve_mesh != nullptr;
}

// io/image-io.h:21-37, the float instantiation: pixels as the reference's loader returns them (8-bit files / 255, HDR and EXR as they are)
inline bool LoadImageFromFile(const std::string& filename, const std::string& asset_path, std::vector<float>* pixels, size_t* width,
                              size_t* height, size_t* channels) {
  float* px = nullptr;
  if (pbrio_image_load(filename.c_str(), asset_path.c_str(), &px, width, height, channels) != PBRHIP_OK) return false;
  pixels->assign(px, px + *width * *height * *channels);
  pbrio_free(px);
  return true;
}
// io/image-io.h:51-63 (image-io.cc:172-224: a float image is written as bytes clamp(x * 256, 0, 255))
inline bool WritePNG(const std::string& filename, const std::string& asset_path, const std::vector<float>& pixels, const size_t width,
                     const size_t height, const size_t channels) {
  return pbrio_write_png_f32(filename.c_str(), asset_path.c_str(), pixels.data(), width, height, channels) == PBRHIP_OK;
}
inline bool WritePNG(const std::string& filename, const std::string& asset_path, const std::vector<unsigned char>& pixels,
                     const size_t width, const size_t height, const size_t channels) {
  return pbrio_write_png_u8(filename.c_str(), asset_path.c_str(), pixels.data(), width, height, channels) == PBRHIP_OK;
}

}  // namespace io
}  // namespace pbrlab

#endif  // PBRLAB_HIP_IO_HPP_
