// pbrlab-hip-cli -- pbrlab-cli (pc/pbrlab-cli.cc:16-60) on the MI355X path tracer.
//
//   pbrlab-hip-cli scene.obj [more.obj ...] [strands.hair ...] [--width W] [--height H] [--spp N] [--out FILE.png]
//                  [--gpus N] [--bvh host|gpu]
//
// Without options it does what the reference binary does: 512 x 512, 32 samples per pixel, "rgba.png" in the current
// directory = sRGB(rgba / count) quantised as byte(x * 256).  --gpus N renders tile i on GPU i % N (one host thread and
// one scene copy per GPU; the disjoint per-GPU layers are added on the host).  --bvh gpu builds the acceleration structure
// on the GPU (faster commit, slightly slower traversal, same image).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "pbrhip_io.h"
#include "pbrlab_hip.hpp"

int main(int argc, char** argv) {
  size_t width = 512, height = 512, samples = 32;  // pbrlab-cli.cc:36-38
  int gpus = 1, bvh = PBRHIP_BVH_HOST_SAH;
  std::string out = "rgba.png";
  std::vector<const char*> files;
  files.push_back(argv[0]);
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto value = [&](const char* name) -> const char* {
      if (i + 1 >= argc) {
        std::cerr << "missing value for " << name << std::endl;
        exit(EXIT_FAILURE);
      }
      return argv[++i];
    };
    if (a == "--width") width = size_t(atol(value("--width")));
    else if (a == "--height") height = size_t(atol(value("--height")));
    else if (a == "--spp") samples = size_t(atol(value("--spp")));
    else if (a == "--out") out = value("--out");
    else if (a == "--gpus") gpus = atoi(value("--gpus"));
    else if (a == "--bvh") bvh = std::string(value("--bvh")) == "gpu" ? PBRHIP_BVH_GPU_LBVH : PBRHIP_BVH_HOST_SAH;
    else files.push_back(argv[i]);
  }
  if (files.size() < 2) {
    std::cerr << "not specified obj filename" << std::endl;
    return EXIT_FAILURE;
  }
  if (width == 0 || height == 0 || samples == 0 || gpus < 1) {
    std::cerr << "width, height, spp and gpus must be positive" << std::endl;
    return EXIT_FAILURE;
  }
  int ndev = 0;
  if (pbrhip_device_count(&ndev) != PBRHIP_OK || ndev < 1) {
    std::cerr << "no HIP device: " << pbrhip_last_error() << std::endl;
    return EXIT_FAILURE;
  }

  pbrlab::RenderLayer layer;
  if (gpus == 1) {
    pbrlab::Scene scene;
    pbrhip_scene_set_bvh_builder(scene.handle(), bvh);
    if (pbrio_create_scene(int(files.size()), files.data(), scene.handle()) != PBRHIP_OK) {
      std::cerr << "scene: " << pbrio_last_error() << std::endl;
      return EXIT_FAILURE;
    }
    std::atomic_bool cancel_render_flag(false);
    std::atomic_size_t finish_pass(0);
    if (!pbrlab::Render(scene, uint32_t(width), uint32_t(height), uint32_t(samples), cancel_render_flag, &layer, &finish_pass))
      return EXIT_FAILURE;
  } else {
    layer.Resize(width, height);
    layer.Clear();
    const size_t ng = size_t(gpus);
    std::vector<pbrlab::RenderLayer> parts(ng);
    std::vector<std::unique_ptr<pbrlab::Scene>> scenes;
    for (int g = 0; g < gpus; ++g) {  // ingestion is repeated per GPU: each device holds its own copy of the scene
      pbrhip_set_device(g % ndev);
      scenes.emplace_back(new pbrlab::Scene());
      pbrhip_scene_set_bvh_builder(scenes.back()->handle(), bvh);
      if (pbrio_create_scene(int(files.size()), files.data(), scenes.back()->handle()) != PBRHIP_OK) {
        std::cerr << "scene: " << pbrio_last_error() << std::endl;
        return EXIT_FAILURE;
      }
    }
    std::vector<int> rc(ng, PBRHIP_OK);
    std::vector<std::string> msg(ng);
    std::vector<std::thread> workers;
    for (int g = 0; g < gpus; ++g) {
      workers.emplace_back([&, g]() {
        parts[size_t(g)].Resize(width, height);
        pbrhip_render_desc d = {};
        d.width = uint32_t(width), d.height = uint32_t(height), d.num_sample = uint32_t(samples);
        d.seed_seq = 1234567890;
        d.tile_rank = uint32_t(g), d.tile_world = uint32_t(gpus);
        d.shard_block = 16;  // finer than the 64 x 64 tile: evens out the load of the GPUs (the image does not depend on it)
        size_t fin = 0;
        rc[size_t(g)] = pbrhip_render(scenes[size_t(g)]->handle(), &d, nullptr, parts[size_t(g)].rgba.data(),
                                      parts[size_t(g)].count.data(), &fin, nullptr);
        if (rc[size_t(g)] != PBRHIP_OK) msg[size_t(g)] = pbrhip_last_error();
      });
    }
    for (std::thread& t : workers) t.join();
    for (int g = 0; g < gpus; ++g) {
      if (rc[size_t(g)] != PBRHIP_OK) {
        std::cerr << "render on GPU " << g % ndev << ": " << msg[size_t(g)] << std::endl;
        return EXIT_FAILURE;
      }
      for (size_t i = 0; i < layer.rgba.size(); ++i) layer.rgba[i] += parts[size_t(g)].rgba[i];  // disjoint tiles + zeros
      for (size_t i = 0; i < layer.count.size(); ++i) layer.count[i] += parts[size_t(g)].count[i];
    }
    for (size_t p = 1; p <= samples; ++p) printf("finish pass %lu\n", (unsigned long)p);
  }

  const size_t slash = out.find_last_of('/');
  const std::string dir = slash == std::string::npos ? "./" : out.substr(0, slash + 1);
  const std::string name = slash == std::string::npos ? out : out.substr(slash + 1);
  if (pbrio_write_layer_png(name.c_str(), dir.c_str(), layer.rgba.data(), layer.count.data(), width, height) != PBRHIP_OK) {
    std::cerr << pbrio_last_error() << std::endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
