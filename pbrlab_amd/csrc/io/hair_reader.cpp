// hair_reader.cpp -- see hair_reader.h
#include "hair_reader.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <limits>

namespace pbio {
namespace {

// 128-byte CyHair header (cyhair.h:8-18)
struct Header {
  char magic[4];
  uint32_t num_strands, total_points, flags, default_segments;
  float default_thickness, default_transparency, default_color[3];
  char information[88];
};
static_assert(sizeof(Header) == 128, "CyHair header is 128 bytes");

// Catmull-Rom (tightness 0.5) -> Bezier, one scalar channel; positions use the same expressions per component
// (nanort::real3 operators are componentwise).  Association order as in curve-util.cc.
void start_piece(float q[4], float p0, float p1, float p2) {  // :32-54
  const float tau = 0.5f, tau3 = tau / 3.0f;
  q[0] = p0;
  q[1] = ((tau + 1.0f) / 3.0f) * p0 + (2.0f / 3.0f) * p1 - tau3 * p2;
  q[2] = tau3 * (p0 - p2) + p1;
  q[3] = p1;
}
void middle_piece(float q[4], float p0, float p1, float p2, float p3) {  // :57-80
  const float tau = 0.5f, tau3 = tau / 3.0f;
  q[0] = p1;
  q[1] = tau3 * (p2 - p0) + p1;
  q[2] = tau3 * (p1 - p3) + p2;
  q[3] = p2;
}
void end_piece(float q[4], float p0, float p1, float p2) {  // :7-30
  const float tau = 0.5f, tau3 = tau / 3.0f;
  q[0] = p1;
  q[1] = tau3 * (p2 - p0) + p1;
  q[2] = (-tau3) * p0 + (2.0f / 3.0f) * p1 + ((tau + 1.0f) / 3.0f) * p2;
  q[3] = p2;
}

// bytes between the read position and the end of the file
size_t bytes_left(FILE* fp) {
  const long here = ftell(fp);
  if (here < 0 || fseek(fp, 0, SEEK_END) != 0) return 0;
  const long end = ftell(fp);
  fseek(fp, here, SEEK_SET);
  return end > here ? size_t(end - here) : 0;
}
template <typename T>
bool read_block(FILE* fp, std::vector<T>* dst, size_t count) {
  // the counts come from the file's header: a block the file cannot hold is a failed read (as it is in the reference,
  // whose fread comes up short), not an allocation of up to 2^32 * 3 floats
  if (count > bytes_left(fp) / sizeof(T)) return false;
  dst->resize(count);
  if (count == 0) return false;  // fread(ptr, 0, 1, fp) returns 0 in the reference too
  return fread(dst->data(), sizeof(T) * count, 1, fp) == 1;
}

}  // namespace

bool ToCubicBezierCurve(const std::vector<float>& cvs, const std::vector<float>& radii, std::vector<float>* bv,
                        std::vector<float>* br) {
  if (cvs.empty() || radii.empty()) return false;
  if (cvs.size() % 3 != 0) return false;
  if (bv->size() % 12 != 0 || br->size() % 4 != 0 || bv->size() != br->size() * 3) return false;
  const size_t n = cvs.size() / 3, nseg = n - 1;
  if (n < 3 || n != radii.size()) return false;

  float q[3][4], r[4];
  auto push = [&]() {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 3; ++j) bv->push_back(q[j][i]);
    for (int i = 0; i < 4; ++i) br->push_back(r[i]);
  };
  for (int j = 0; j < 3; ++j) start_piece(q[j], cvs[j], cvs[3 + j], cvs[6 + j]);
  start_piece(r, radii[0], radii[1], radii[2]);
  push();
  for (size_t s = 1; s + 1 < nseg; ++s) {
    const size_t k = s - 1;
    for (int j = 0; j < 3; ++j) middle_piece(q[j], cvs[3 * k + j], cvs[3 * k + 3 + j], cvs[3 * k + 6 + j], cvs[3 * k + 9 + j]);
    middle_piece(r, radii[k], radii[k + 1], radii[k + 2], radii[k + 3]);
    push();
  }
  if (nseg > 1) {
    const size_t k = nseg - 2;
    for (int j = 0; j < 3; ++j) end_piece(q[j], cvs[3 * k + j], cvs[3 * k + 3 + j], cvs[3 * k + 6 + j]);
    end_piece(r, radii[k], radii[k + 1], radii[k + 2]);
    push();
  }
  return true;
}

bool LoadCyHair(const std::string& filepath, bool is_y_up, std::vector<std::vector<float>>* vertices,
                std::vector<std::vector<float>>* thicknesses) {
  FILE* fp = fopen(filepath.c_str(), "rb");
  if (!fp) return false;
  Header h;
  if (fread(&h, 128, 1, fp) != 1 || memcmp(h.magic, "HAIR", 4) != 0) {
    fclose(fp);
    return false;
  }
  const bool has_segments = h.flags & 0x1, has_points = h.flags & 0x2, has_thickness = h.flags & 0x4;
  const bool has_transparency = h.flags & 0x8, has_color = h.flags & 0x10;
  const int default_segments = static_cast<int>(h.default_segments);
  if (!has_points || (default_segments < 1 && !has_segments)) {
    fclose(fp);
    return false;
  }
  std::vector<unsigned short> segments;
  std::vector<float> points, thick, skip;
  bool ok = true;
  if (has_segments) ok = read_block(fp, &segments, h.num_strands);
  if (ok) ok = read_block(fp, &points, size_t(h.total_points) * 3);
  if (ok && has_thickness) ok = read_block(fp, &thick, h.total_points);
  if (ok && has_transparency) ok = read_block(fp, &skip, h.total_points);
  if (ok && has_color) ok = read_block(fp, &skip, size_t(h.total_points) * 3);
  fclose(fp);
  if (!ok) return false;

  size_t offset = 0;
  for (size_t s = 0; s < h.num_strands; ++s) {
    const size_t nseg = segments.empty() ? size_t(default_segments) : segments.at(s);
    const size_t nv = nseg + 1;
    if (nv < 2) continue;  // (the reference does not advance the point offset here either)
    vertices->emplace_back();
    thicknesses->emplace_back();
    std::vector<float>& pv = vertices->back();
    std::vector<float>& pt = thicknesses->back();
    for (size_t i = 0; i < nv; ++i) {
      const size_t b = (offset + i) * 3;
      pv.push_back(points.at(b + 0));
      pv.push_back(points.at(b + (is_y_up ? 1 : 2)));
      pv.push_back(points.at(b + (is_y_up ? 2 : 1)));
      pt.push_back(thick.empty() ? h.default_thickness : thick.at(offset + i));
    }
    offset += nv;
  }
  return true;
}

bool LoadCurveMeshAsCubicBezierCurve(const std::string& filepath, bool memory_saving_mode, std::vector<float>* vt,
                                     std::vector<uint32_t>* indices) {
  const size_t dot = filepath.find_last_of('.');
  const size_t slash = filepath.find_last_of('/');
  const std::string ext = (dot == std::string::npos || (slash != std::string::npos && dot < slash)) ? "" : filepath.substr(dot);
  if (ext != ".hair") {
    std::cerr << "unknown data type" << std::endl;
    return false;
  }
  std::vector<std::vector<float>> strands, widths;
  try {
    LoadCyHair(filepath, true, &strands, &widths);  // result not checked in the reference: a failed load = no strands
  } catch (const std::out_of_range&) {
    std::cerr << "CyHair file [" << filepath << "] is truncated" << std::endl;
    return false;
  }
  if (widths.size() != strands.size()) return false;

  size_t base = 0;
  const float eps = std::numeric_limits<float>::epsilon();
  for (size_t s = 0; s < strands.size(); ++s) {
    std::vector<float> bv, bt;
    const bool ok = ToCubicBezierCurve(strands[s], widths[s], &bv, &bt);
    const size_t nv = bt.size();
    if (!ok || bv.size() != nv * 3 || nv % 4 != 0) return false;
    if (memory_saving_mode) {  // consecutive segments share their joint: 3 points per segment + the last one
      const size_t nseg = nv / 4;
      for (size_t g = 0; g < nseg; ++g) {
        indices->push_back(uint32_t(base + g * 3));
        for (size_t c = 0; c < 3; ++c) {
          const size_t v = g * 4 + c;
          vt->push_back(bv[3 * v + 0]), vt->push_back(bv[3 * v + 1]), vt->push_back(bv[3 * v + 2]);
          vt->push_back(bt[v]);
        }
        if (g > 0) {
          for (size_t a = 0; a < 3; ++a)
            if (!(fabsf(bv[3 * g * 4 + a] - bv[3 * g * 4 + a - 3]) < eps)) return false;
        }
      }
      vt->push_back(bv[3 * (nv - 1) + 0]), vt->push_back(bv[3 * (nv - 1) + 1]), vt->push_back(bv[3 * (nv - 1) + 2]);
      vt->push_back(bt[nv - 1]);
    } else {
      for (size_t v = 0; v < nv; ++v) {
        if (v % 4 == 0) indices->push_back(uint32_t(base + v));
        vt->push_back(bv[3 * v + 0]), vt->push_back(bv[3 * v + 1]), vt->push_back(bv[3 * v + 2]);
        vt->push_back(bt[v]);
      }
    }
    base = vt->size() / 4;
  }
  return true;
}

}  // namespace pbio
