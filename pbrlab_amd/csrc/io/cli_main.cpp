// pbrlab-hip-cli -- pbrlab-cli (pc/pbrlab-cli.cc:16-60) on the MI355X path tracer.
//
//   pbrlab-hip-cli scene.obj [more.obj ...] [strands.hair ...] [--width W] [--height H] [--spp N] [--out FILE.png]
//                  [--gpus N] [--bvh host|gpu]
//
// Without options it does what the reference binary does: 512 x 512, 32 samples per pixel, "rgba.png" in the current
// directory = sRGB(rgba / count) quantised as byte(x * 256).  --gpus N deals 16 x 16 pixel blocks to N ranks, rank g on
// GPU g % (GPUs present): one host thread per rank, the scene is ingested once and copied device-to-device, the shards
// are gathered on the first GPU over xGMI inside the library (pbrhip_render_multi).  --bvh gpu builds the acceleration structure
// on the GPU (faster commit, slightly slower traversal, same image).
#include <atomic>
#include <cerrno>
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
    // a positive integer that fits the library's uint32_t fields
    auto number = [&](const char* name, unsigned long max) -> size_t {
      const char* v = value(name);
      char* end = nullptr;
      errno = 0;
      const unsigned long n = strtoul(v, &end, 10);
      if (errno || end == v || *end || v[0] == '-' || n == 0 || n > max) {
        std::cerr << name << " needs an integer in 1.." << max << ", got '" << v << "'" << std::endl;
        exit(EXIT_FAILURE);
      }
      return size_t(n);
    };
    if (a == "--width") width = number("--width", 0xFFFFFFFFul);
    else if (a == "--height") height = number("--height", 0xFFFFFFFFul);
    else if (a == "--spp") samples = number("--spp", 0xFFFFFFFFul);
    else if (a == "--out") out = value("--out");
    else if (a == "--gpus") gpus = int(number("--gpus", 1024));
    else if (a == "--bvh") bvh = std::string(value("--bvh")) == "gpu" ? PBRHIP_BVH_GPU_LBVH : PBRHIP_BVH_HOST_SAH;
    else files.push_back(argv[i]);
  }
  if (files.size() < 2) {
    std::cerr << "not specified obj filename" << std::endl;
    return EXIT_FAILURE;
  }
  if (uint64_t(width) * uint64_t(height) >= (1ull << 32)) {
    std::cerr << "width x height must stay below 2^32 pixels" << std::endl;
    return EXIT_FAILURE;
  }
  int ndev = 0;
  if (pbrhip_device_count(&ndev) != PBRHIP_OK || ndev < 1) {
    std::cerr << "no HIP device: " << pbrhip_last_error() << std::endl;
    return EXIT_FAILURE;
  }

  // the scene is ingested and its BVH built once; the other GPUs get device-to-device copies
  pbrlab::RenderLayer layer;
  pbrlab::Scene scene;
  pbrhip_scene_set_bvh_builder(scene.handle(), bvh);
  if (pbrio_create_scene(int(files.size()), files.data(), scene.handle()) != PBRHIP_OK) {
    std::cerr << "scene: " << pbrio_last_error() << std::endl;
    return EXIT_FAILURE;
  }
  std::atomic_bool cancel_render_flag(false);
  std::atomic_size_t finish_pass(0);
  if (gpus == 1) {
    if (!pbrlab::Render(scene, uint32_t(width), uint32_t(height), uint32_t(samples), cancel_render_flag, &layer, &finish_pass))
      return EXIT_FAILURE;
  } else {
    std::vector<std::unique_ptr<pbrlab::Scene>> replicas;
    std::vector<const pbrlab::Scene*> scenes{&scene};
    try {
      for (int g = 1; g < gpus; ++g) {  // rank g renders on GPU g % ndev
        replicas.push_back(scene.Replicate(g % ndev));
        scenes.push_back(replicas.back().get());
      }
    } catch (const std::exception& e) {
      std::cerr << "scene copy: " << e.what() << std::endl;
      return EXIT_FAILURE;
    }
    if (!pbrlab::Render(scenes, uint32_t(width), uint32_t(height), uint32_t(samples), cancel_render_flag, &layer, &finish_pass))
      return EXIT_FAILURE;
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
