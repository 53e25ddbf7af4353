// Written for this repo: walks, against include/pbrlab_hip.hpp, the call sequences pbrlab's own callers use --
//   CreateSceneFromObj               pc/pc-common.cc:100-186  (materials, textures, texture-id fix-up through mpark::get, per
//                                    shape: SetMaterialId, AddTriangleMesh(triangle_mesh), local scene, instance, `light*` shapes)
//   CreateSceneFromCubicBezierCurve  pc/pc-common.cc:188-233  (default HairBsdfParameter, SetMaterialName, AddCubicBezierCurveMesh)
//   CreateScene's tail               pc/pc-common.cc:262-268  (CommitScene, FetchSceneAABB)
//   the GUI's material edit loop     pc/glfw-window.cc:866-979 + EditQueue::EditAndPopAll pc/pc-common.cc:57-84
//                                    (FetchMeshMaterialParameters, GetMaterialName, *dst = *src between two Render() calls)
// -- with pbrlab's value types (TriangleMesh, Attribute, Texture, MaterialParameter, AreaLightParameter, float3, MeshPtr), then
// renders.  Checks: an edit made through FetchMeshMaterialParameters() changes the next frame exactly as a scene built with the
// edited material from the start; no edit, no change.  Exit code 0 = ok, 3 = no HIP device (expected on a CPU box).
#include <cmath>
#include <cstdio>

#include "pbrlab_hip.hpp"

namespace {

// what pbrlab::io::LoadTriangleMeshFromObj would return for a small .obj with a .mtl (io/triangle-mesh-io.cc:214-325): meshes that
// share one Attribute, per-face material indices into `material_params`, texture indices into `textures`
void LoadStandIn(std::vector<pbrlab::TriangleMesh>* triangle_meshes, std::vector<pbrlab::MaterialParameter>* material_params,
                 std::vector<pbrlab::Texture>* textures) {
  auto attr = std::make_shared<pbrlab::Attribute>();
  attr->vertices = {-1, -1, -1, 1, 1,  -1, -1, 1, 1,  -1, 1,  1, -1, -1, 1,  1,   // floor
                    -1, -1, -1, 1, 1,  -1, -1, 1, 1,  1,  -1, 1, -1, 1,  -1, 1,   // back wall
                    -.4f, .95f, -.4f, 1, .4f, .95f, -.4f, 1, .4f, .95f, .4f, 1, -.4f, .95f, .4f, 1};  // light
  attr->texcoords = {0, 0, 1, 0, 1, 1, 0, 1};
  pbrlab::CyclesPrincipledBsdfParameter wall, floor;
  wall.base_color = pbrlab::float3(0.7f, 0.3f, 0.2f), wall.specular = 0.8f, wall.roughness = 0.3f, wall.name = "Wall";
  floor.base_color_tex_id = 0, floor.name = "Floor";  // index into `textures`: fixed up by the caller (pc-common.cc:129-139)
  material_params->emplace_back(wall);
  material_params->emplace_back(floor);
  std::vector<float> px(4 * 4 * 3);
  for (size_t i = 0; i < px.size(); ++i) px[i] = 0.2f + 0.6f * float((i / 3 + i / 12) % 2);
  textures->emplace_back(px, 4u, 4u, 3u, "checker");
  triangle_meshes->emplace_back("floor", attr, std::vector<uint32_t>{0, 2, 1, 0, 3, 2}, std::vector<uint32_t>{},
                                std::vector<uint32_t>{0, 2, 1, 0, 3, 2}, std::vector<uint32_t>{1, 1});
  triangle_meshes->emplace_back("back", attr, std::vector<uint32_t>{4, 5, 6, 4, 6, 7}, std::vector<uint32_t>{}, std::vector<uint32_t>{},
                                std::vector<uint32_t>{0, 0});
  triangle_meshes->emplace_back("light_top", attr, std::vector<uint32_t>{8, 9, 10, 8, 10, 11}, std::vector<uint32_t>{},
                                std::vector<uint32_t>{}, std::vector<uint32_t>{0, 0});
}

void FixTextureId(const std::vector<uint32_t>& texture_ids, uint32_t* fixed_texture_id) {  // pc-common.cc:93-98
  if (*fixed_texture_id != uint32_t(-1)) *fixed_texture_id = texture_ids.at(*fixed_texture_id);
}

// pc/pc-common.cc:100-186, statement for statement (the loader call replaced by the stand-in above)
bool CreateSceneFromObj(pbrlab::Scene* scene, const pbrlab::MaterialParameter* replace_wall) {
  std::vector<pbrlab::TriangleMesh> triangle_meshes;
  std::vector<pbrlab::MaterialParameter> material_params;
  std::vector<pbrlab::Texture> textures;
  LoadStandIn(&triangle_meshes, &material_params, &textures);
  if (replace_wall) material_params[0] = *replace_wall;

  std::vector<uint32_t> material_ids;
  for (const auto& material_param : material_params) {
    const uint32_t material_id = scene->AddMaterialParam(material_param);
    material_ids.emplace_back(material_id);
  }
  std::vector<uint32_t> texture_ids;
  for (const auto& texture : textures) {
    const uint32_t tex_id = scene->AddTexture(texture);
    texture_ids.emplace_back(tex_id);
  }
  for (auto& material_param : material_params) {
    if (material_param.index() == pbrlab::kCyclesPrincipledBsdfParameter) {
      pbrlab::CyclesPrincipledBsdfParameter& cycles_material_param = mpark::get<pbrlab::kCyclesPrincipledBsdfParameter>(material_param);
      FixTextureId(texture_ids, &(cycles_material_param.base_color_tex_id));
      FixTextureId(texture_ids, &(cycles_material_param.subsurface_color_tex_id));
    }
  }
  for (auto& triangle_mesh : triangle_meshes) {
    const uint32_t num_face = triangle_mesh.GetNumFaces();
    const std::vector<uint32_t>& no_fix_material_ids = triangle_mesh.GetMaterials();
    if (no_fix_material_ids.size() != num_face) return false;
    for (uint32_t f_id = 0; f_id < num_face; ++f_id) triangle_mesh.SetMaterialId(material_ids.at(no_fix_material_ids.at(f_id)), f_id);
    const pbrlab::MeshPtr mesh_ptr = scene->AddTriangleMesh(triangle_mesh);
    const uint32_t local_scene_id = scene->CreateLocalScene();
    scene->AddMeshToLocalScene(local_scene_id, mesh_ptr);
    const float transform[4][4] = {{1.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 1.0f}};
    const uint32_t instance_id = scene->CreateInstance(local_scene_id, transform);
    if (triangle_mesh.GetName().substr(0, 5) == "light") {
      pbrlab::AreaLightParameter area_light_param = {};
      area_light_param.emission = pbrlab::float3(3.0f);
      const uint32_t light_id = scene->AddLightParam(area_light_param);
      std::vector<std::vector<uint32_t>> light_param_ids;
      light_param_ids.emplace_back(std::vector<uint32_t>(triangle_mesh.GetNumFaces()));
      for (uint32_t f = 0; f < triangle_mesh.GetNumFaces(); ++f) light_param_ids[0][f] = light_id;
      scene->AttachLightParamIdsToInstance(instance_id, light_param_ids);
    }
  }
  return true;
}

// pc/pc-common.cc:188-233 (the loader call replaced by three hand-made segments)
bool CreateSceneFromCubicBezierCurve(pbrlab::Scene* scene) {
  auto attr = std::make_shared<pbrlab::CurveAttribute>();
  for (int s = 0; s < 3; ++s)
    for (int c = 0; c < 4; ++c) {
      const float t = float(c) / 3.0f;
      attr->vertices.insert(attr->vertices.end(), {-0.6f + 0.6f * float(s), -0.9f + 1.2f * t, 0.2f * std::sin(3.0f * t + float(s)), 0.03f});
    }
  pbrlab::CubicBezierCurveMesh curve_mesh("strands", attr, std::vector<uint32_t>{0, 4, 8}, std::vector<uint32_t>{});
  pbrlab::MaterialParameter material_param = pbrlab::HairBsdfParameter();
  pbrlab::SetMaterialName("hair", &material_param);
  const uint32_t material_id = scene->AddMaterialParam(material_param);
  const uint32_t num_segments = curve_mesh.GetNumSegments();
  for (uint32_t seg_id = 0; seg_id < num_segments; ++seg_id) curve_mesh.SetMaterialId(material_id, seg_id);
  const pbrlab::MeshPtr mesh_ptr = scene->AddCubicBezierCurveMesh(curve_mesh);
  const uint32_t local_scene_id = scene->CreateLocalScene();
  scene->AddMeshToLocalScene(local_scene_id, mesh_ptr);
  const float transform[4][4] = {{1.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 1.0f}};
  const uint32_t instance_id = scene->CreateInstance(local_scene_id, transform);
  (void)instance_id;
  return pbrlab::GetNumPrimitive(mesh_ptr) == 3 && pbrlab::GetName(mesh_ptr) == "strands";
}

bool Build(pbrlab::Scene* scene, const pbrlab::MaterialParameter* replace_wall = nullptr) {
  if (!CreateSceneFromObj(scene, replace_wall) || !CreateSceneFromCubicBezierCurve(scene)) return false;
  scene->CommitScene();  // pc-common.cc:262
  float bmin[3], bmax[3];
  scene->FetchSceneAABB(bmin, bmax);
  return bmin[0] == -1.0f && bmax[1] == 1.0f;
}

bool RenderInto(const pbrlab::Scene& scene, pbrlab::RenderLayer* layer) {
  std::atomic_bool cancel(false);
  std::atomic_size_t finish_pass(7);  // (a counter reused from a previous frame: Render() starts it from 0)
  return pbrlab::Render(scene, 96, 64, 4, cancel, layer, &finish_pass) && finish_pass == 4;
}

}  // namespace

int main() {
  try {
    pbrlab::Scene scene;
    if (!Build(&scene)) return 10;
    pbrlab::RenderLayer a, b, c, d;
    if (!RenderInto(scene, &a)) return 11;
    double sum = 0;
    for (size_t i = 0; i < a.rgba.size(); i += 4) sum += a.rgba[i] + a.rgba[i + 1] + a.rgba[i + 2];
    if (!(sum > 0) || !std::isfinite(sum)) return 12;
    if (!RenderInto(scene, &b) || a.rgba != b.rgba) return 13;  // nothing edited: the same frame

    // the GUI's edit: pc/glfw-window.cc:869-870 + 900-979 builds an edited copy, EditQueue::EditAndPopAll (pc-common.cc:57-84)
    // assigns it over the scene's element between two Render() calls
    std::vector<pbrlab::MaterialParameter>* materials = scene.FetchMeshMaterialParameters();
    if (materials->size() != 3 || pbrlab::GetMaterialName((*materials)[0]) != "Wall" || pbrlab::GetMaterialName((*materials)[2]) != "hair") return 14;
    pbrlab::MaterialParameter* dst = &(*materials)[0];
    pbrlab::MaterialParameter* src = new pbrlab::MaterialParameter(*dst);
    {
      pbrlab::CyclesPrincipledBsdfParameter& m = mpark::get<pbrlab::kCyclesPrincipledBsdfParameter>(*src);
      m.base_color.v[0] = 0.1f, m.base_color.v[1] = 0.6f, m.base_color.v[2] = 0.9f;
      m.roughness = 0.05f, m.clearcoat = 0.5f;
    }
    const pbrlab::MaterialParameter edited = *src;
    (*dst) = (*src);  // CopyAndDeleteSrc, pc-common.cc:46-55
    delete src;
    if (!RenderInto(scene, &c) || c.rgba == a.rgba) return 15;  // the edit is in the next frame

    pbrlab::Scene fresh;  // the same scene built with the edited material from the start
    if (!Build(&fresh, &edited)) return 16;
    if (!RenderInto(fresh, &d)) return 17;
    if (c.rgba != d.rgba || c.count != d.count) return 18;

    // the hair material too (pc/glfw-window.cc:826-862 edits a HairBsdfParameter in place)
    mpark::get<pbrlab::kHairBsdfParameter>((*materials)[2]).melanin = 0.1f;
    pbrlab::RenderLayer e;
    if (!RenderInto(scene, &e) || e.rgba == c.rgba) return 19;
    printf("shim types ok: %zu materials, mean radiance %f\n", materials->size(), sum / (96 * 64 * 4 * 3));
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "shim: %s\n", e.what());
    return 3;
  }
}
