// pbrlab_hip.hpp -- header-only C++ shim that keeps pbrlab's names (namespace pbrlab: Scene, RenderLayer,
// Render) over the C ABI of libpbrhip (pbrhip.h).  What a pbrlab caller (pc/pbrlab-cli.cc:24-59,
// pc/pc-common.cc:100-270, pc/pbrlab-gui.cc:207-222) includes instead of scene.h / render.h.
//
// Mesh and material value types are reduced to what the callers use: meshes are built from the flat arrays of
// pbrlab::Attribute (src/mesh/attribute.h) and the id vectors of TriangleMesh (src/mesh/triangle-mesh.h:52-57).
#ifndef PBRLAB_HIP_HPP_
#define PBRLAB_HIP_HPP_

#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <iostream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pbrhip.h"

namespace pbrlab {

using CyclesPrincipledBsdfParameter = pbrhip_principled_param;  // src/material-param.h:24-49
using HairBsdfParameter = pbrhip_hair_param;                     // src/material-param.h:51-72

struct AreaLightParameter {  // src/light-param.h:20-23
  float emission[3] = {0.8f, 0.8f, 0.8f};
  std::string name;
};

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

// handle returned by Add*Mesh (stands in for pbrlab::MeshPtr, src/mesh/mesh.h:25-26)
struct MeshPtr {
  uint32_t id = PBRHIP_NONE;
};

// src/scene.h:14-111
class Scene {
public:
  Scene() {
    if (pbrhip_scene_create(&h_) != PBRHIP_OK) throw std::runtime_error(pbrhip_last_error());
  }
  ~Scene() { pbrhip_scene_destroy(h_); }
  Scene(const Scene&) = delete;
  Scene& operator=(const Scene&) = delete;

  // Scene::AddTriangleMesh(name, attribute, vertex_ids, normal_ids, texcoord_ids, material_ids)
  MeshPtr AddTriangleMesh(const std::string& /*name*/, const std::vector<float>& vertices_xyzw,
                          const std::vector<float>& normals_xyzw, const std::vector<float>& texcoords_uv,
                          const std::vector<uint32_t>& vertex_ids, const std::vector<uint32_t>& normal_ids,
                          const std::vector<uint32_t>& texcoord_ids, const std::vector<uint32_t>& material_ids) {
    const uint32_t nf = uint32_t(vertex_ids.size() / 3);
    MeshPtr m;
    Check(pbrhip_scene_add_triangle_mesh(
        h_, vertices_xyzw.data(), uint32_t(vertices_xyzw.size() / 4), normals_xyzw.data(),
        uint32_t(normals_xyzw.size() / 4), texcoords_uv.data(), uint32_t(texcoords_uv.size() / 2), vertex_ids.data(),
        normal_ids.size() == size_t(nf) * 3 ? normal_ids.data() : nullptr,
        texcoord_ids.size() == size_t(nf) * 3 ? texcoord_ids.data() : nullptr,
        material_ids.size() == size_t(nf) ? material_ids.data() : nullptr, nf, &m.id));
    return m;
  }
  // Scene::AddCubicBezierCurveMesh(name, attribute, indices, material_ids)
  MeshPtr AddCubicBezierCurveMesh(const std::string& /*name*/, const std::vector<float>& vertices_xyzr,
                                  const std::vector<uint32_t>& indices, const std::vector<uint32_t>& material_ids) {
    MeshPtr m;
    Check(pbrhip_scene_add_curve_mesh(h_, vertices_xyzr.data(), uint32_t(vertices_xyzr.size() / 4), indices.data(),
                                      material_ids.size() == indices.size() ? material_ids.data() : nullptr,
                                      uint32_t(indices.size()), &m.id));
    return m;
  }
  // Scene::AddTexture(pixels, width, height, channels, name) (scene.h:46-51, texture.cc:10-21)
  uint32_t AddTexture(const std::vector<float>& pixels, uint32_t width, uint32_t height, uint32_t channels,
                      const std::string& /*name*/ = "") {
    uint32_t id;
    Check(pbrhip_scene_add_texture(h_, pixels.data(), width, height, channels, &id));
    return id;
  }
  uint32_t AddLightParam(const AreaLightParameter& p) {
    uint32_t id;
    Check(pbrhip_scene_add_area_light(h_, p.emission, &id));
    return id;
  }
  uint32_t AddMaterialParam(const CyclesPrincipledBsdfParameter& p) {
    uint32_t id;
    Check(pbrhip_scene_add_principled_material(h_, &p, &id));
    return id;
  }
  uint32_t AddMaterialParam(const HairBsdfParameter& p) {
    uint32_t id;
    Check(pbrhip_scene_add_hair_material(h_, &p, &id));
    return id;
  }
  uint32_t AddMeshToLocalScene(uint32_t local_scene_id, const MeshPtr& mesh) {
    uint32_t g;
    Check(pbrhip_scene_add_mesh_to_local_scene(h_, local_scene_id, mesh.id, &g));
    return g;
  }
  // throws std::runtime_error on a size mismatch like scene.cc:64-94
  void AttachLightParamIdsToInstance(uint32_t instance_id, const std::vector<std::vector<uint32_t>>& ids) {
    for (size_t g = 0; g < ids.size(); ++g)
      Check(pbrhip_scene_attach_light_ids(h_, instance_id, uint32_t(g), ids[g].data(), uint32_t(ids[g].size())));
  }
  void AttachMaterialParamIdsToInstance(uint32_t instance_id, const std::vector<std::vector<uint32_t>>& ids) {
    for (size_t g = 0; g < ids.size(); ++g)
      Check(pbrhip_scene_attach_material_ids(h_, instance_id, uint32_t(g), ids[g].data(), uint32_t(ids[g].size())));
  }
  void CommitScene() { Check(pbrhip_scene_commit(h_)); }
  uint32_t CreateInstance(uint32_t local_scene_id, const float transform[4][4]) {
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
  // material edits between renders (EditQueue, pc/pc-common.cc:57-84)
  void UpdateMaterialParam(uint32_t id, const CyclesPrincipledBsdfParameter& p) {
    Check(pbrhip_scene_update_principled_material(h_, id, &p));
  }
  void UpdateMaterialParam(uint32_t id, const HairBsdfParameter& p) { Check(pbrhip_scene_update_hair_material(h_, id, &p)); }

  pbrhip_scene* handle() const { return h_; }

  // A copy of this committed scene on another GPU (device memory is copied device-to-device: no second ingestion or
  // BVH build); for the multi-GPU Render() overload below.
  std::unique_ptr<Scene> Replicate(int device) const {
    pbrhip_scene* h = nullptr;
    Check(pbrhip_scene_replicate(h_, device, &h));
    return std::unique_ptr<Scene>(new Scene(h));
  }

private:
  explicit Scene(pbrhip_scene* h) : h_(h) {}
  static void Check(int rc) {
    if (rc != PBRHIP_OK) throw std::runtime_error(pbrhip_last_error());
  }
  pbrhip_scene* h_ = nullptr;
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
  for (const Scene* s : scenes) hs.push_back(s->handle());
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
