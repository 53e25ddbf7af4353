// pbrlab_hip.hpp -- header-only C++ shim that keeps pbrlab's names AND value types (namespace pbrlab: float3, Attribute,
// TriangleMesh, CubicBezierCurveMesh, Texture, MaterialParameter, AreaLightParameter, MeshPtr, Scene, RenderLayer, Render)
// over the C ABI of libpbrhip (pbrhip.h).  What a pbrlab caller (pc/pbrlab-cli.cc:24-59, pc/pc-common.cc:100-270,
// pc/pbrlab-gui.cc:207-222, pc/glfw-window.cc:866-979) includes instead of scene.h / render.h: the call sequences of
// CreateSceneFromObj / CreateSceneFromCubicBezierCurve and of the GUI's material edit loop compile against it as they are
// (mpark::variant / mpark::get are std::variant / std::get here: the alias below; -DPBRLAB_HIP_NO_MPARK_ALIAS drops it).
//
// Differences a caller can observe: the library COPIES mesh, texture and material data when it is handed over (the
// reference's raytracer shares the mesh buffers, raytracer.h:34), and material edits made through
// Scene::FetchMeshMaterialParameters() reach the GPU at the next Render() call (the GUI applies them between calls too:
// EditQueue::EditAndPopAll, pc/pc-common.cc:57-84).
#ifndef PBRLAB_HIP_HPP_
#define PBRLAB_HIP_HPP_

#include <algorithm>
#include <atomic>
#include <cassert>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <variant>
#include <vector>

#include "pbrhip.h"

#ifndef PBRLAB_HIP_NO_MPARK_ALIAS
namespace mpark {  // pbrlab spells its variants mpark::variant / mpark::get<I> (src/material-param.h:14, src/mesh/mesh.h:9)
using std::get;
using std::get_if;
using std::holds_alternative;
using std::variant;
}  // namespace mpark
#endif

namespace pbrlab {

// src/type.h:8 (nanort::real3<float>): the part of it pbrlab's callers use
struct float3 {
  float3() : v{0.f, 0.f, 0.f} {}
  explicit float3(float s) : v{s, s, s} {}
  float3(float x, float y, float z) : v{x, y, z} {}
  explicit float3(const float* p) : v{p[0], p[1], p[2]} {}
  float x() const { return v[0]; }
  float y() const { return v[1]; }
  float z() const { return v[2]; }
  float operator[](int i) const { return v[i]; }
  float& operator[](int i) { return v[i]; }
  float v[3];
};

// src/mesh/attribute.h:7-18
struct Attribute {
  std::vector<float> vertices;   // 4(xyzw) * num vertices (w = 1.0f)
  std::vector<float> normals;    // 4(xyzw) * num normals or 0
  std::vector<float> texcoords;  // 2(uv) * num texcoords or 0
};
struct CurveAttribute {
  std::vector<float> vertices;  // 4(xyz + thickness) * num vertices
};

// src/mesh/triangle-mesh.h:15-62, triangle-mesh.cc:17-57 (the data side; the fetch functions run on the GPU)
class TriangleMesh {
public:
  TriangleMesh() : num_faces_(0) {}
  TriangleMesh(const std::string name, const std::shared_ptr<Attribute>& attribute, const std::vector<uint32_t> vertex_ids,
               const std::vector<uint32_t> normal_ids, const std::vector<uint32_t> texcoord_ids,
               const std::vector<uint32_t> material_ids)
      : num_faces_(uint32_t(vertex_ids.size() / 3)), vertex_ids_(vertex_ids), normal_ids_(normal_ids), texcoord_ids_(texcoord_ids),
        material_ids_(material_ids), pAttribute_(attribute), name_(name) {}
  const std::vector<uint32_t>& GetMaterials() const { return material_ids_; }
  uint32_t GetNumFaces() const { return num_faces_; }
  uint32_t GetNumVertices() const { return pAttribute_ ? uint32_t(pAttribute_->vertices.size() / 4) : 0u; }
  std::string GetName() const { return name_; }
  const std::vector<uint32_t>& GetVertexIds() const { return vertex_ids_; }
  const std::vector<float>& GetVertices() const { return pAttribute_->vertices; }
  void SetMaterialId(const uint32_t material_id, const uint32_t prim_id) { material_ids_.at(prim_id) = material_id; }
  // (not in the reference: what Scene::AddTriangleMesh hands to the library)
  const std::vector<uint32_t>& GetNormalIds() const { return normal_ids_; }
  const std::vector<uint32_t>& GetTexcoordIds() const { return texcoord_ids_; }
  const std::shared_ptr<Attribute>& GetAttribute() const { return pAttribute_; }

private:
  uint32_t num_faces_;
  std::vector<uint32_t> vertex_ids_, normal_ids_, texcoord_ids_;  // 3 * num_faces_ (normals / texcoords: or 0)
  std::vector<uint32_t> material_ids_;                            // num_faces_ or 0
  std::shared_ptr<Attribute> pAttribute_;
  std::string name_;
};

// src/mesh/cubic-bezier-curve-mesh.h:11-37
class CubicBezierCurveMesh {
public:
  CubicBezierCurveMesh() {}
  CubicBezierCurveMesh(const std::string& name, const std::shared_ptr<CurveAttribute> attribute,
                       const std::vector<uint32_t>& indices, const std::vector<uint32_t>& material_ids)
      : pAttribute_(attribute), indices_(indices), material_ids_(material_ids), name_(name) {}
  const std::vector<uint32_t>& GetIndices() const { return indices_; }
  uint32_t GetNumSegments() const { return uint32_t(indices_.size()); }
  uint32_t GetNumVertices() const { return pAttribute_ ? uint32_t(pAttribute_->vertices.size() / 4) : 0u; }
  const std::vector<uint32_t>& GetMaterials() const { return material_ids_; }
  std::string GetName() const { return name_; }
  const std::vector<float>& GetVertices() const { return pAttribute_->vertices; }
  void SetMaterialId(const uint32_t material_id, const uint32_t segment_id) {
    if (material_ids_.size() != indices_.size()) material_ids_.resize(indices_.size(), uint32_t(-1));
    material_ids_.at(segment_id) = material_id;
  }

private:
  std::shared_ptr<CurveAttribute> pAttribute_;
  std::vector<uint32_t> indices_, material_ids_;
  std::string name_;
};

// src/mesh/mesh.h:21-26
enum MeshType { kTriangleMesh = 0, kCubicBezierCurveMesh, kMeshNone };
using MeshPtr = std::variant<std::shared_ptr<TriangleMesh>, std::shared_ptr<CubicBezierCurveMesh>>;
inline std::string GetName(const MeshPtr& m) {
  return m.index() == kTriangleMesh ? std::get<kTriangleMesh>(m)->GetName() : std::get<kCubicBezierCurveMesh>(m)->GetName();
}
inline uint32_t GetNumPrimitive(const MeshPtr& m) {
  return m.index() == kTriangleMesh ? std::get<kTriangleMesh>(m)->GetNumFaces() : std::get<kCubicBezierCurveMesh>(m)->GetNumSegments();
}

// src/texture.h:11-44 (the data side; texture.cc:43-57 runs on the GPU)
class Texture {
public:
  Texture() : width_(0), height_(0), channels_(0) {}
  Texture(const std::vector<float>& pixels, const uint32_t width, const uint32_t height, const uint32_t channels,
          const std::string& name)
      : width_(width), height_(height), channels_(channels), pixels_(pixels), name_(name) {}
  bool Reset(const std::vector<float>& pixels, const uint32_t width, const uint32_t height, const uint32_t channels) {
    if (pixels.size() != size_t(width) * height * channels) return false;
    pixels_ = pixels, width_ = width, height_ = height, channels_ = channels;
    return true;
  }
  void SetName(const std::string& name) { name_ = name; }
  uint32_t GetWidth() const { return width_; }
  uint32_t GetHeight() const { return height_; }
  uint32_t GetChannels() const { return channels_; }
  std::string GetName() const { return name_; }
  const std::vector<float>& GetPixels() const { return pixels_; }  // (not in the reference)

private:
  uint32_t width_, height_, channels_;
  std::vector<float> pixels_;
  std::string name_;
};

// src/material-param.h:20-103: the same members, defaults and helper functions
enum MaterialParameterType { kCyclesPrincipledBsdfParameter = 0, kHairBsdfParameter };
struct CyclesPrincipledBsdfParameter {
  float3 base_color = {0.8f, 0.8f, 0.8f};
  float subsurface = 0.0f;
  float3 subsurface_radius = {1.0f, 1.0f, 1.0f};
  float3 subsurface_color = {0.7f, 0.1f, 0.1f};
  float metallic = 0.0f, specular = 0.5f, specular_tint = 0.0f, roughness = 0.5f, anisotropic = 0.0f, anisotropic_rotation = 0.0f;
  float sheen = 0.0f, sheen_tint = 0.5f, clearcoat = 0.0f, clearcoat_roughness = 0.03f, ior = 1.45f, transmission = 0.0f;
  float transmission_roughness = 0.0f;
  uint32_t base_color_tex_id = uint32_t(-1), subsurface_color_tex_id = uint32_t(-1);
  std::string name = "";
};
struct HairBsdfParameter {
  enum ColoringHair { kRGB = 0, kMelanin };
  ColoringHair coloring_hair = kMelanin;
  float3 base_color = {0.18f, 0.06f, 0.02f};
  float melanin = 0.5f, melanin_redness = 0.8f, melanin_randomize = 0.f;
  float roughness = 0.2f, azimuthal_roughness = 0.3f;
  float ior = 1.55f;
  float shift = 2.f;
  float3 specular_tint = {1.f, 1.f, 1.f}, second_specular_tint = {1.f, 1.f, 1.f}, transmission_tint = {1.f, 1.f, 1.f};
  std::string name = "";
};
using MaterialParameter = std::variant<CyclesPrincipledBsdfParameter, HairBsdfParameter>;
inline void SetMaterialName(const std::string& name, MaterialParameter* m) {
  if (m->index() == kCyclesPrincipledBsdfParameter) std::get<kCyclesPrincipledBsdfParameter>(*m).name = name;
  else std::get<kHairBsdfParameter>(*m).name = name;
}
inline std::string GetMaterialName(const MaterialParameter& m) {
  return m.index() == kCyclesPrincipledBsdfParameter ? std::get<kCyclesPrincipledBsdfParameter>(m).name : std::get<kHairBsdfParameter>(m).name;
}

// src/light-param.h:18-47
struct AreaLightParameter {
  float3 emission = float3(0.8f);
  std::string name;
};
enum LightType { kAreaLight = 0, kLightNone };
using LightParameter = std::variant<AreaLightParameter>;

// src/render-layer.h:11-26
struct RenderLayer {
  RenderLayer() = default;
  RenderLayer(size_t w, size_t h) { Resize(w, h), Clear(); }
  void Clear() {
    std::lock_guard<std::mutex> lock(mtx);
    std::fill(rgba.begin(), rgba.end(), 0.0f);
    std::fill(count.begin(), count.end(), 0u);
  }
  void Resize(size_t w, size_t h) {
    std::lock_guard<std::mutex> lock(mtx);
    width = w, height = h;
    rgba.resize(w * h * 4);
    count.resize(w * h);
  }
  size_t width = 0, height = 0;
  std::vector<float> rgba;
  std::vector<uint32_t> count;
  mutable std::mutex mtx;
};

namespace detail {
inline pbrhip_principled_param ToAbi(const CyclesPrincipledBsdfParameter& p) {
  pbrhip_principled_param q;
  std::memset(&q, 0, sizeof(q));
  for (int k = 0; k < 3; ++k)
    q.base_color[k] = p.base_color[k], q.subsurface_radius[k] = p.subsurface_radius[k], q.subsurface_color[k] = p.subsurface_color[k];
  q.subsurface = p.subsurface, q.metallic = p.metallic, q.specular = p.specular, q.specular_tint = p.specular_tint;
  q.roughness = p.roughness, q.anisotropic = p.anisotropic, q.anisotropic_rotation = p.anisotropic_rotation;
  q.sheen = p.sheen, q.sheen_tint = p.sheen_tint, q.clearcoat = p.clearcoat, q.clearcoat_roughness = p.clearcoat_roughness;
  q.ior = p.ior, q.transmission = p.transmission, q.transmission_roughness = p.transmission_roughness;
  q.base_color_tex_id = p.base_color_tex_id, q.subsurface_color_tex_id = p.subsurface_color_tex_id;
  return q;
}
inline pbrhip_hair_param ToAbi(const HairBsdfParameter& p) {
  pbrhip_hair_param q;
  std::memset(&q, 0, sizeof(q));
  q.coloring_hair = uint32_t(p.coloring_hair);
  for (int k = 0; k < 3; ++k) {
    q.base_color[k] = p.base_color[k], q.specular_tint[k] = p.specular_tint[k];
    q.second_specular_tint[k] = p.second_specular_tint[k], q.transmission_tint[k] = p.transmission_tint[k];
  }
  q.melanin = p.melanin, q.melanin_redness = p.melanin_redness, q.melanin_randomize = p.melanin_randomize;
  q.roughness = p.roughness, q.azimuthal_roughness = p.azimuthal_roughness, q.ior = p.ior, q.shift = p.shift;
  return q;
}
// what the library was last given for a material (compared byte for byte with the caller's current value before a render)
struct Pushed {
  uint32_t kind = 0;
  pbrhip_principled_param pr;
  pbrhip_hair_param hr;
};
inline Pushed Flatten(const MaterialParameter& m) {
  Pushed f;
  std::memset(&f.pr, 0, sizeof(f.pr)), std::memset(&f.hr, 0, sizeof(f.hr));
  f.kind = uint32_t(m.index());
  if (m.index() == kCyclesPrincipledBsdfParameter) f.pr = ToAbi(std::get<kCyclesPrincipledBsdfParameter>(m));
  else f.hr = ToAbi(std::get<kHairBsdfParameter>(m));
  return f;
}
}  // namespace detail

// src/scene.h:14-111
class Scene {
public:
  Scene() {
    if (pbrhip_abi_version() != PBRHIP_ABI_VERSION)  // (the library writes whole structs of ITS layout)
      throw std::runtime_error("libpbrhip.so was built from another version of pbrhip.h (struct layouts differ): rebuild");
    if (pbrhip_scene_create(&h_) != PBRHIP_OK) throw std::runtime_error(pbrhip_last_error());
  }
  ~Scene() { pbrhip_scene_destroy(h_); }
  Scene(const Scene&) = delete;
  Scene& operator=(const Scene&) = delete;

  // scene.h:19-24: AddTriangleMesh(triangle_mesh) as pc/pc-common.cc:159 calls it, or the constructor arguments of
  // TriangleMesh (name, attribute, vertex_ids, normal_ids, texcoord_ids, material_ids)
  template <class... Args>
  MeshPtr AddTriangleMesh(Args&&... args) {
    triangle_meshes_.emplace_back(std::make_shared<TriangleMesh>(args...));
    const TriangleMesh& m = *triangle_meshes_.back();
    const uint32_t nf = m.GetNumFaces();
    const Attribute empty;
    const Attribute& a = m.GetAttribute() ? *m.GetAttribute() : empty;
    uint32_t id;
    Check(pbrhip_scene_add_triangle_mesh(
        h_, a.vertices.data(), uint32_t(a.vertices.size() / 4), a.normals.data(), uint32_t(a.normals.size() / 4),
        a.texcoords.data(), uint32_t(a.texcoords.size() / 2), m.GetVertexIds().data(),
        m.GetNormalIds().size() == size_t(nf) * 3 ? m.GetNormalIds().data() : nullptr,
        m.GetTexcoordIds().size() == size_t(nf) * 3 ? m.GetTexcoordIds().data() : nullptr,
        m.GetMaterials().size() == size_t(nf) ? m.GetMaterials().data() : nullptr, nf, &id));
    mesh_ids_.emplace_back(triangle_meshes_.back().get(), id);
    return MeshPtr(triangle_meshes_.back());
  }
  // scene.h:26-32
  template <class... Args>
  MeshPtr AddCubicBezierCurveMesh(Args&&... args) {
    cubic_bezier_curve_meshes_.emplace_back(std::make_shared<CubicBezierCurveMesh>(args...));
    const CubicBezierCurveMesh& m = *cubic_bezier_curve_meshes_.back();
    const std::vector<float> none;
    const std::vector<float>& v = m.GetNumVertices() ? m.GetVertices() : none;
    uint32_t id;
    Check(pbrhip_scene_add_curve_mesh(h_, v.data(), uint32_t(v.size() / 4), m.GetIndices().data(),
                                      m.GetMaterials().size() == m.GetIndices().size() ? m.GetMaterials().data() : nullptr,
                                      m.GetNumSegments(), &id));
    mesh_ids_.emplace_back(cubic_bezier_curve_meshes_.back().get(), id);
    return MeshPtr(cubic_bezier_curve_meshes_.back());
  }
  // scene.h:34-37 (LightManager::AddLightParam)
  template <class... Args>
  uint32_t AddLightParam(Args&&... args) {
    const AreaLightParameter p(args...);
    uint32_t id;
    Check(pbrhip_scene_add_area_light(h_, p.emission.v, &id));
    return id;
  }
  // scene.h:39-44: a MaterialParameter, or either of its alternatives
  template <class... Args>
  uint32_t AddMaterialParam(Args&&... args) {
    // the library first: if it refuses the material nothing is appended and the two lists stay in step
    const uint32_t id = uint32_t(material_params_.size());
    MaterialParameter param(args...);
    const detail::Pushed f = detail::Flatten(param);
    uint32_t lib_id = uint32_t(-1);
    if (f.kind == kCyclesPrincipledBsdfParameter) Check(pbrhip_scene_add_principled_material(h_, &f.pr, &lib_id));
    else Check(pbrhip_scene_add_hair_material(h_, &f.hr, &lib_id));
    if (lib_id != id) throw std::runtime_error("pbrlab::Scene::AddMaterialParam: material ids of the library and of the shim differ");
    material_params_.emplace_back(std::move(param));
    pushed_.push_back(f);
    return id;
  }
  // scene.h:46-51: a Texture, or its constructor arguments (pixels, width, height, channels, name)
  template <class... Args>
  uint32_t AddTexture(Args&&... args) {
    const Texture t(args...);
    uint32_t id;
    Check(pbrhip_scene_add_texture(h_, t.GetPixels().data(), t.GetWidth(), t.GetHeight(), t.GetChannels(), &id));
    return id;
  }
  uint32_t AddMeshToLocalScene(const uint32_t local_scene_id, const MeshPtr& mesh_ptr) {
    const void* key = mesh_ptr.index() == kTriangleMesh ? static_cast<const void*>(std::get<kTriangleMesh>(mesh_ptr).get())
                                                        : static_cast<const void*>(std::get<kCubicBezierCurveMesh>(mesh_ptr).get());
    for (const auto& e : mesh_ids_)
      if (e.first == key) {
        uint32_t g;
        Check(pbrhip_scene_add_mesh_to_local_scene(h_, local_scene_id, e.second, &g));
        return g;
      }
    throw std::runtime_error("AddMeshToLocalScene: the mesh was not added to this scene");
  }
  // throws std::runtime_error on a size mismatch like scene.cc:64-94
  void AttachLightParamIdsToInstance(const uint32_t instance_id, const std::vector<std::vector<uint32_t>>& ids) {
    for (size_t g = 0; g < ids.size(); ++g)
      Check(pbrhip_scene_attach_light_ids(h_, instance_id, uint32_t(g), ids[g].data(), uint32_t(ids[g].size())));
  }
  void AttachMaterialParamIdsToInstance(const uint32_t instance_id, const std::vector<std::vector<uint32_t>>& ids) {
    for (size_t g = 0; g < ids.size(); ++g)
      Check(pbrhip_scene_attach_material_ids(h_, instance_id, uint32_t(g), ids[g].data(), uint32_t(ids[g].size())));
  }
  void CommitScene() { Check(pbrhip_scene_commit(h_)); }
  uint32_t CreateInstance(const uint32_t local_scene_id, const float transform[4][4]) {
    uint32_t id;
    Check(pbrhip_scene_create_instance(h_, local_scene_id, &transform[0][0], &id));
    return id;
  }
  uint32_t CreateLocalScene() {
    uint32_t id;
    Check(pbrhip_scene_create_local_scene(h_, &id));
    return id;
  }
  // scene.cc:251-259 (the header there names the arguments (bmax, bmin); every caller passes (bmin, bmax))
  void FetchSceneAABB(float* bmin, float* bmax) const { Check(pbrhip_scene_aabb(h_, bmin, bmax)); }

  // scene.h:81, scene.cc:206-208: the scene's material table; the GUI edits its elements between renders
  // (pc/glfw-window.cc:866-979 through EditQueue, pc/pc-common.cc:57-84).  An edit reaches the GPU at the next Render():
  // PushMaterialEdits() compares every element with what the library was last given and calls
  // pbrhip_scene_update_*_material for those that differ (a material cannot change its kind).
  std::vector<MaterialParameter>* FetchMeshMaterialParameters() { return &material_params_; }
  void PushMaterialEdits() const {
    if (material_params_.size() != pushed_.size()) throw std::runtime_error("materials are added with AddMaterialParam");
    for (size_t i = 0; i < material_params_.size(); ++i) {
      const detail::Pushed f = detail::Flatten(material_params_[i]);
      if (f.kind == pushed_[i].kind && std::memcmp(&f.pr, &pushed_[i].pr, sizeof(f.pr)) == 0 && std::memcmp(&f.hr, &pushed_[i].hr, sizeof(f.hr)) == 0)
        continue;
      if (f.kind == kCyclesPrincipledBsdfParameter) Check(pbrhip_scene_update_principled_material(h_, uint32_t(i), &f.pr));
      else Check(pbrhip_scene_update_hair_material(h_, uint32_t(i), &f.hr));
      pushed_[i] = f;
    }
  }

  pbrhip_scene* handle() const { return h_; }

  // A copy of this committed scene on another GPU (device memory is copied device-to-device: no second ingestion or
  // BVH build); for the multi-GPU Render() overload below.  Material edits made afterwards go to the scene they are made on.
  std::unique_ptr<Scene> Replicate(int device) const {
    PushMaterialEdits();
    pbrhip_scene* h = nullptr;
    Check(pbrhip_scene_replicate(h_, device, &h));
    std::unique_ptr<Scene> r(new Scene(h));
    r->material_params_ = material_params_, r->pushed_ = pushed_;
    return r;
  }

private:
  explicit Scene(pbrhip_scene* h) : h_(h) {}
  static void Check(int rc) {
    if (rc != PBRHIP_OK) throw std::runtime_error(pbrhip_last_error());
  }
  pbrhip_scene* h_ = nullptr;
  std::vector<std::shared_ptr<TriangleMesh>> triangle_meshes_;                  // scene.h:98
  std::vector<std::shared_ptr<CubicBezierCurveMesh>> cubic_bezier_curve_meshes_;  // scene.h:100
  std::vector<std::pair<const void*, uint32_t>> mesh_ids_;                      // mesh object -> the library's mesh id
  std::vector<MaterialParameter> material_params_;                              // scene.h:102
  mutable std::vector<detail::Pushed> pushed_;
};

namespace detail {
static_assert(sizeof(std::atomic_bool) == 1 && sizeof(std::atomic_size_t) == sizeof(size_t),
              "the library reads std::atomic_bool as a byte and stores std::atomic_size_t as a size_t");
// While the library renders, a watcher prints "finish pass N" as *finish_pass advances (render.cc:229 prints from the
// worker that completes a pass).
struct ProgressPrinter {
  explicit ProgressPrinter(const std::atomic_size_t* fin) : fin_(fin) {
    if (fin_) th_ = std::thread([this]() {
      size_t shown = 0;
      for (;;) {
        const bool last = stop_.load();
        for (const size_t now = fin_->load(); shown < now;) printf("finish pass %lu\n", (unsigned long)++shown);
        if (last) return;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
    });
  }
  ~ProgressPrinter() {
    stop_.store(true);
    if (th_.joinable()) th_.join();
  }
  const std::atomic_size_t* fin_;
  std::atomic_bool stop_{false};
  std::thread th_;
};
}  // namespace detail

// src/render.h:14-17.  Blocking; clears and resizes *layer.  cancel_render_flag is read live by the library (at every
// host round trip of its render loop; the reference polls it before every tile job, render.cc:217): setting it from
// another thread ends the call early with the completed passes in *layer.  *finish_pass advances while the call runs
// (render.cc:224-231).  Returns true like the reference (render.cc:240); a library error is reported on std::cerr and
// returns false.
inline bool Render(const Scene& scene, const uint32_t width, const uint32_t height, const uint32_t num_sample,
                   const std::atomic_bool& cancel_render_flag, RenderLayer* layer, std::atomic_size_t* finish_pass) {
  layer->Resize(width, height);  // PrepareRendering, render.cc:99-100 (the library clears)
  try {
    scene.PushMaterialEdits();  // edits made through FetchMeshMaterialParameters() since the last call
  } catch (const std::exception& e) {
    std::cerr << "pbrlab::Render: " << e.what() << std::endl;
    return false;
  }
  pbrhip_render_desc d = {};
  d.width = width, d.height = height, d.num_sample = num_sample;
  d.seed_seq = 1234567890;  // render.cc:215
  d.tile_world = 1;
  std::atomic_size_t local_fin(0);
  std::atomic_size_t* fin = finish_pass ? finish_pass : &local_fin;
  int rc;
  fin->store(0);  // render.cc:209: zero before any worker starts (a counter reused from the previous frame must not be reported)
  {
    detail::ProgressPrinter progress(fin);
    rc = pbrhip_render(scene.handle(), &d, reinterpret_cast<const volatile unsigned char*>(&cancel_render_flag),
                       layer->rgba.data(), layer->count.data(), reinterpret_cast<size_t*>(fin), nullptr);
  }
  if (rc != PBRHIP_OK) {
    std::cerr << "pbrlab::Render: " << pbrhip_last_error() << std::endl;
    return false;
  }
  return true;
}

// The same frame over several GPUs of this process: `scenes` = the committed scene and its replicas
// (Scene::Replicate), one per device.  Pixel blocks are dealt to the devices, the shards are gathered device-to-device
// over xGMI inside the library (pbrhip_render_multi); the image is bit-identical to the one-GPU frame.
inline bool Render(const std::vector<const Scene*>& scenes, const uint32_t width, const uint32_t height,
                   const uint32_t num_sample, const std::atomic_bool& cancel_render_flag, RenderLayer* layer,
                   std::atomic_size_t* finish_pass) {
  layer->Resize(width, height);
  pbrhip_render_desc d = {};
  d.width = width, d.height = height, d.num_sample = num_sample;
  d.seed_seq = 1234567890;
  d.tile_world = 1;
  std::vector<pbrhip_scene*> hs;
  try {
    for (const Scene* s : scenes) s->PushMaterialEdits(), hs.push_back(s->handle());
  } catch (const std::exception& e) {
    std::cerr << "pbrlab::Render: " << e.what() << std::endl;
    return false;
  }
  std::atomic_size_t local_fin(0);
  std::atomic_size_t* fin = finish_pass ? finish_pass : &local_fin;
  int rc;
  fin->store(0);  // render.cc:209: zero before any worker starts (a counter reused from the previous frame must not be reported)
  {
    detail::ProgressPrinter progress(fin);
    rc = pbrhip_render_multi(hs.data(), uint32_t(hs.size()), &d,
                             reinterpret_cast<const volatile unsigned char*>(&cancel_render_flag), layer->rgba.data(),
                             layer->count.data(), reinterpret_cast<size_t*>(fin), nullptr);
  }
  if (rc != PBRHIP_OK) {
    std::cerr << "pbrlab::Render: " << pbrhip_last_error() << std::endl;
    return false;
  }
  return true;
}

}  // namespace pbrlab
#endif  // PBRLAB_HIP_HPP_
