// ASan/UBSan mutation fuzz of the standalone readers (CPU only, not part of the product)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <string>
#include <vector>
#include "hair_reader.h"
#include "image_codec.h"
#include "obj_reader.h"
static std::vector<uint8_t> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}
int main(int argc, char** argv) {
  std::mt19937 rng(123);
  const int iters = atoi(argv[1]);
  std::cerr.setstate(std::ios::failbit);
  size_t ok = 0, total = 0;
  for (int a = 2; a < argc; ++a) {
    const std::string path = argv[a];
    const std::vector<uint8_t> seed = slurp(path);
    const std::string ext = path.substr(path.find_last_of('.'));
    for (int it = 0; it < iters; ++it) {
      std::vector<uint8_t> d = seed;
      const int nm = 1 + int(rng() % 8);
      for (int m = 0; m < nm && !d.empty(); ++m) {
        const size_t pos = rng() % d.size();
        switch (rng() % 4) {
          case 0: d[pos] = uint8_t(rng()); break;
          case 1: d[pos] ^= uint8_t(1u << (rng() % 8)); break;
          case 2: d.resize(pos); break;
          case 3: d.insert(d.begin() + long(pos), uint8_t(rng())); break;
        }
      }
      std::vector<uint8_t> px8;
      std::vector<float> pxf;
      size_t w, h, c;
      std::string err;
      bool r = false;
      total++;
      if (ext == ".png") r = pbio::DecodePng(d.data(), d.size(), &px8, &w, &h, &c, &err);
      else if (ext == ".jpg") r = pbio::DecodeJpeg(d.data(), d.size(), &px8, &w, &h, &c, &err);
      else if (ext == ".exr") r = pbio::DecodeExr(d.data(), d.size(), &pxf, &w, &h, &err);
      else if (ext == ".hdr" || ext == ".pic") r = pbio::DecodeHdr(d.data(), d.size(), &pxf, &w, &h, &err);
      else if (ext == ".bmp") r = pbio::IsBmp(d.data(), d.size()) && pbio::DecodeBmp(d.data(), d.size(), &px8, &w, &h, &c, &err);
      else if (ext == ".tga") r = pbio::IsTga(d.data(), d.size()) && pbio::DecodeTga(d.data(), d.size(), &px8, &w, &h, &c, &err);
      else if (ext == ".ppm" || ext == ".pgm") r = pbio::IsPnm(d.data(), d.size()) && pbio::DecodePnm(d.data(), d.size(), &px8, &w, &h, &c, &err);
      else if (ext == ".gif") r = pbio::IsGif(d.data(), d.size()) && pbio::DecodeGif(d.data(), d.size(), &px8, &w, &h, &c, &err);
      else if (ext == ".psd") r = pbio::IsPsd(d.data(), d.size()) && pbio::DecodePsd(d.data(), d.size(), &px8, &w, &h, &c, &err);
      else if (ext == ".obj" || ext == ".hair" || ext == ".mtl") {
        const std::string tmp = std::string("/tmp/pbrio_fuzz/m") + ext;
        { std::ofstream o(tmp, std::ios::binary); o.write(reinterpret_cast<const char*>(d.data()), long(d.size())); }
        if (ext == ".hair") {
          std::vector<float> vt; std::vector<uint32_t> idx;
          r = pbio::LoadCurveMeshAsCubicBezierCurve(tmp, (it & 1) != 0, &vt, &idx);
        } else {
          pbio::ObjFile f;
          r = pbio::ReadObj(ext == ".obj" ? tmp : std::string("/tmp/pbrio_fuzz/host.obj"), "/tmp/pbrio_fuzz", &f);
        }
      }
      ok += r ? 1 : 0;
    }
  }
  printf("fuzz done: %zu inputs, %zu decoded\n", total, ok);
  return 0;
}
