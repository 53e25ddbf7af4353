// Compiles the C++ shim (include/pbrlab_hip.hpp) with plain g++ against libpbrhip.so and drives it the way
// pc/pbrlab-cli.cc + pc/pc-common.cc do.  Exit code 0 = rendered, 3 = no HIP device (expected on a CPU box).
#include <cmath>
#include <cstdio>

#include "pbrlab_hip.hpp"

int main() {
  try {
    pbrlab::Scene scene;
    // one quad `light` above one quad floor, shared attribute buffer like LoadTriangleMeshFromObj
    std::vector<float> v = {-1, 0, -1, 1, 1, 0, -1, 1, 1, 0, 1, 1, -1, 0, 1, 1,
                            -.5f, 1, -.5f, 1, .5f, 1, -.5f, 1, .5f, 1, .5f, 1, -.5f, 1, .5f, 1};
    pbrlab::CyclesPrincipledBsdfParameter white = {};
    white.base_color[0] = white.base_color[1] = white.base_color[2] = 0.8f;
    white.subsurface_radius[0] = white.subsurface_radius[1] = white.subsurface_radius[2] = 1.f;
    white.roughness = 0.5f, white.sheen_tint = 0.5f, white.clearcoat_roughness = 0.03f, white.ior = 1.45f;
    white.base_color_tex_id = white.subsurface_color_tex_id = PBRHIP_NONE;
    const uint32_t mat = scene.AddMaterialParam(white);
    const float I[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    auto floor = scene.AddTriangleMesh("floor", v, {}, {}, {0, 2, 1, 0, 3, 2}, {}, {}, {mat, mat});
    uint32_t ls = scene.CreateLocalScene();
    scene.AddMeshToLocalScene(ls, floor);
    scene.CreateInstance(ls, I);
    auto light = scene.AddTriangleMesh("light", v, {}, {}, {4, 5, 6, 4, 6, 7}, {}, {}, {mat, mat});
    ls = scene.CreateLocalScene();
    scene.AddMeshToLocalScene(ls, light);
    const uint32_t inst = scene.CreateInstance(ls, I);
    pbrlab::AreaLightParameter lp;
    lp.emission[0] = lp.emission[1] = lp.emission[2] = 3.0f;  // pc/pc-common.cc:174
    const uint32_t lid = scene.AddLightParam(lp);
    scene.AttachLightParamIdsToInstance(inst, {{lid, lid}});
    bool threw = false;
    try {
      scene.AttachLightParamIdsToInstance(inst, {{lid}});
    } catch (const std::runtime_error&) {
      threw = true;  // scene.cc:68-69
    }
    if (!threw) return 10;
    scene.CommitScene();
    pbrlab::RenderLayer layer;
    std::atomic_bool cancel(false);
    std::atomic_size_t finish_pass(0);
    if (!pbrlab::Render(scene, 64, 48, 4, cancel, &layer, &finish_pass)) return 11;
    if (finish_pass != 4 || layer.count[100] != 4 || layer.rgba[100 * 4 + 3] != 4.0f) return 12;
    double sum = 0;
    for (size_t i = 0; i < layer.rgba.size(); i += 4) sum += layer.rgba[i];
    if (!(sum > 0) || !std::isfinite(sum)) return 13;
    printf("shim ok: mean R %f\n", sum / (64 * 48 * 4));
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "shim: %s\n", e.what());
    return 3;
  }
}
