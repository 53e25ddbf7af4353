// GPU-side caller test of include/pbrlab_hip.hpp, written from the API tables of INTEGRATION.md (not from the reference's
// callers: those are compiled as they are by oracle/Makefile's `ref_cli`, tests/test_reference_callers.py).  It uses pbrlab's
// value types -- TriangleMesh, Attribute, Texture, MaterialParameter, AreaLightParameter, float3, MeshPtr -- the way the API
// is meant to be used: materials and textures first, texture ids renumbered to the scene's, one local scene + identity instance
// per mesh, an emissive instance, a curve mesh with a hair material; then the GUI-style material edit between two Render() calls.
// Checks: an edit made through FetchMeshMaterialParameters() changes the next frame exactly as a scene built with the edited
// material from the start; no edit, no change.  Exit code 0 = ok, 3 = no HIP device (expected on a CPU box).
#include <cmath>
#include <cstdio>
#include <functional>

#include "pbrlab_hip.hpp"

namespace {

using Ids = std::vector<uint32_t>;
const float kIdentity[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};

struct Part {  // one mesh of the test scene: a quad out of the shared vertex pool
  const char* name;
  uint32_t first_vertex;
  bool flipped, textured, emissive;
  uint32_t material;  // index into the test's own material list
};
const Part kParts[] = {{"floor", 0, true, true, false, 1}, {"back", 4, false, false, false, 0}, {"light_top", 8, false, false, true, 0}};

std::shared_ptr<pbrlab::Attribute> VertexPool() {
  auto a = std::make_shared<pbrlab::Attribute>();
  const float quads[3][4][3] = {{{-1, -1, -1}, {1, -1, -1}, {1, -1, 1}, {-1, -1, 1}},
                                {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}},
                                {{-.4f, .95f, -.4f}, {.4f, .95f, -.4f}, {.4f, .95f, .4f}, {-.4f, .95f, .4f}}};
  for (const auto& q : quads)
    for (const auto& v : q) a->vertices.insert(a->vertices.end(), {v[0], v[1], v[2], 1.0f});
  a->texcoords = {0, 0, 1, 0, 1, 1, 0, 1};
  return a;
}

std::vector<pbrlab::MaterialParameter> Materials(const pbrlab::MaterialParameter* replace_first) {
  pbrlab::CyclesPrincipledBsdfParameter wall, floor;
  wall.name = "Wall", wall.base_color = pbrlab::float3(0.7f, 0.3f, 0.2f), wall.specular = 0.8f, wall.roughness = 0.3f;
  floor.name = "Floor", floor.base_color_tex_id = 0;  // an index into the test's texture list until it is renumbered below
  std::vector<pbrlab::MaterialParameter> m = {wall, floor};
  if (replace_first) m[0] = *replace_first;
  return m;
}

pbrlab::Texture Checker() {
  std::vector<float> px(4 * 4 * 3);
  for (size_t i = 0; i < px.size(); ++i) px[i] = 0.2f + 0.6f * float((i / 3 + i / 12) % 2);
  return pbrlab::Texture(px, 4u, 4u, 3u, "checker");
}

// one local scene + identity instance per mesh (the only placement pbrlab's callers use)
uint32_t Place(pbrlab::Scene* scene, const pbrlab::MeshPtr& mesh) {
  const uint32_t local = scene->CreateLocalScene();
  scene->AddMeshToLocalScene(local, mesh);
  return scene->CreateInstance(local, kIdentity);
}

bool AddQuads(pbrlab::Scene* scene, const pbrlab::MaterialParameter* replace_first) {
  std::vector<pbrlab::MaterialParameter> mats = Materials(replace_first);
  // textures get scene-wide ids; the materials that point into the test's own list are renumbered BEFORE they are added
  const Ids tex_of_scene = {scene->AddTexture(Checker())};
  auto renumber = [&](uint32_t* id) {
    if (*id != uint32_t(-1)) *id = tex_of_scene.at(*id);
  };
  Ids mat_of_scene;
  for (auto& m : mats) {
    if (auto* p = std::get_if<pbrlab::CyclesPrincipledBsdfParameter>(&m)) renumber(&p->base_color_tex_id), renumber(&p->subsurface_color_tex_id);
    mat_of_scene.push_back(scene->AddMaterialParam(m));
  }
  const auto pool = VertexPool();
  for (const Part& part : kParts) {
    const uint32_t v = part.first_vertex;
    const Ids tri = part.flipped ? Ids{v, v + 2, v + 1, v, v + 3, v + 2} : Ids{v, v + 1, v + 2, v, v + 2, v + 3};
    const Ids uv = part.textured ? Ids{0, 2, 1, 0, 3, 2} : Ids{};
    pbrlab::TriangleMesh mesh(part.name, pool, tri, Ids{}, uv, Ids(2, uint32_t(-1)));
    if (mesh.GetNumFaces() != 2 || mesh.GetMaterials().size() != 2) return false;
    for (uint32_t f = 0; f < mesh.GetNumFaces(); ++f) mesh.SetMaterialId(mat_of_scene.at(part.material), f);
    const uint32_t instance = Place(scene, scene->AddTriangleMesh(mesh));
    if (part.emissive) {
      pbrlab::AreaLightParameter light = {};
      light.emission = pbrlab::float3(3.0f);
      const uint32_t light_id = scene->AddLightParam(light);
      scene->AttachLightParamIdsToInstance(instance, {Ids(mesh.GetNumFaces(), light_id)});
    }
  }
  return true;
}

bool AddStrands(pbrlab::Scene* scene) {
  auto attr = std::make_shared<pbrlab::CurveAttribute>();
  for (int s = 0; s < 3; ++s)
    for (int c = 0; c < 4; ++c) {
      const float t = float(c) / 3.0f;
      attr->vertices.insert(attr->vertices.end(), {-0.6f + 0.6f * float(s), -0.9f + 1.2f * t, 0.2f * std::sin(3.0f * t + float(s)), 0.03f});
    }
  pbrlab::CubicBezierCurveMesh strands("strands", attr, Ids{0, 4, 8}, Ids{});
  pbrlab::MaterialParameter hair = pbrlab::HairBsdfParameter();
  pbrlab::SetMaterialName("hair", &hair);
  const uint32_t hair_id = scene->AddMaterialParam(hair);
  for (uint32_t seg = 0; seg < strands.GetNumSegments(); ++seg) strands.SetMaterialId(hair_id, seg);
  const pbrlab::MeshPtr mesh = scene->AddCubicBezierCurveMesh(strands);
  Place(scene, mesh);
  return pbrlab::GetNumPrimitive(mesh) == 3 && pbrlab::GetName(mesh) == "strands";
}

bool Build(pbrlab::Scene* scene, const pbrlab::MaterialParameter* replace_wall = nullptr) {
  if (!AddQuads(scene, replace_wall) || !AddStrands(scene)) return false;
  scene->CommitScene();
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

    // a GUI-style edit: an edited copy is assigned over the scene's element between two Render() calls (INTEGRATION.md,
    // "material edits")
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
    *dst = edited;
    delete src;
    if (!RenderInto(scene, &c) || c.rgba == a.rgba) return 15;  // the edit is in the next frame

    pbrlab::Scene fresh;  // the same scene built with the edited material from the start
    if (!Build(&fresh, &edited)) return 16;
    if (!RenderInto(fresh, &d)) return 17;
    if (c.rgba != d.rgba || c.count != d.count) return 18;

    // the hair material too, edited in place
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
