// Compiles the C++ shim (include/pbrlab_hip.hpp) with plain g++ against libpbrhip.so and drives it the way
// pc/pbrlab-cli.cc + pc/pc-common.cc do.  Exit code 0 = rendered, 3 = no HIP device (expected on a CPU box).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <thread>

#include "pbrlab_hip.hpp"

int main() {
  try {
    pbrlab::Scene scene;
    // one quad `light` above one quad floor, shared attribute buffer like LoadTriangleMeshFromObj (io/triangle-mesh-io.cc:238-262)
    auto attr = std::make_shared<pbrlab::Attribute>();
    attr->vertices = {-1, 0, -1, 1, 1, 0, -1, 1, 1, 0, 1, 1, -1, 0, 1, 1,
                      -.5f, 1, -.5f, 1, .5f, 1, -.5f, 1, .5f, 1, .5f, 1, -.5f, 1, .5f, 1};
    pbrlab::CyclesPrincipledBsdfParameter white;  // the defaults of material-param.h:24-49
    const uint32_t mat = scene.AddMaterialParam(white);
    const float I[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    const pbrlab::MeshPtr floor = scene.AddTriangleMesh("floor", attr, std::vector<uint32_t>{0, 2, 1, 0, 3, 2}, std::vector<uint32_t>{},
                                                        std::vector<uint32_t>{}, std::vector<uint32_t>{mat, mat});
    uint32_t ls = scene.CreateLocalScene();
    scene.AddMeshToLocalScene(ls, floor);
    scene.CreateInstance(ls, I);
    const pbrlab::MeshPtr light = scene.AddTriangleMesh("light", attr, std::vector<uint32_t>{4, 5, 6, 4, 6, 7}, std::vector<uint32_t>{},
                                                        std::vector<uint32_t>{}, std::vector<uint32_t>{mat, mat});
    ls = scene.CreateLocalScene();
    scene.AddMeshToLocalScene(ls, light);
    const uint32_t inst = scene.CreateInstance(ls, I);
    pbrlab::AreaLightParameter lp;
    lp.emission = pbrlab::float3(3.0f);  // pc/pc-common.cc:174
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
    // the same frame over three ranks of this process (all on the GPUs present: rank g on device g % ndev), shards
    // gathered inside the library: bit-identical to the one-GPU frame
    int ndev = 0;
    pbrhip_device_count(&ndev);
    std::vector<std::unique_ptr<pbrlab::Scene>> replicas;
    std::vector<const pbrlab::Scene*> all{&scene};
    for (int g = 1; g < 3; ++g) replicas.push_back(scene.Replicate(g % ndev)), all.push_back(replicas.back().get());
    pbrlab::RenderLayer multi;
    std::atomic_size_t fin_multi(0);
    if (!pbrlab::Render(all, 64, 48, 4, cancel, &multi, &fin_multi)) return 14;
    if (fin_multi != 4 || multi.rgba != layer.rgba || multi.count != layer.count) return 15;
    // cancel from another thread while the call runs (render.cc:217): the call returns early, every pixel holds exactly
    // finish_pass complete passes
    pbrlab::RenderLayer big;
    std::atomic_size_t fin_big(0);
    const uint32_t kSpp = 1u << 20;  // hours of work if nobody cancels
    const auto t0 = std::chrono::steady_clock::now();
    std::thread killer([&]() {
      std::this_thread::sleep_for(std::chrono::milliseconds(150));
      cancel.store(true);
    });
    const bool ok = pbrlab::Render(scene, 256, 256, kSpp, cancel, &big, &fin_big);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    killer.join();
    if (!ok) return 16;
    if (fin_big >= kSpp || ms > 5000.0) return 17;
    for (size_t i = 0; i < big.count.size(); ++i)
      if (big.count[i] != fin_big || big.rgba[4 * i + 3] != float(fin_big)) return 18;
    printf("cancelled after %.0f ms with %lu complete passes\n", ms, (unsigned long)fin_big.load());
    printf("shim ok: mean R %f\n", sum / (64 * 48 * 4));
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "shim: %s\n", e.what());
    return 3;
  }
}
