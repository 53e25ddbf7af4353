// ref_io_harness.cc -- ORACLE support (test infrastructure only) for rows N1/N2 (scene ingestion, image output).
//
// A driver, written for this repo, around the REFERENCE's own host I/O code compiled unmodified where it lies
// under /root/reference/src:
//   io/tiny_obj_loader.{h,cc}  (vendored tinyobjloader 2.0.0: what io/triangle-mesh-io.cc:216-255 calls)
//   io/cyhair.{h,cc}, io/curve-mesh-io.{h,cc}, curve-util.{h,cc}, mesh/cubic-bezier-curve-mesh.{h,cc}
//   io/image-io.{h,cc} (+ vendored io/stb_image*.{h,cc}, io/tinyexr.{h,cc}, miniz.{h,c}), image-utils.{h,cc}
//   (tinyexr's writer is also exposed, as a generator of .exr test files)
// Built by oracle/Makefile into oracle/_ref/libref_io.so.  io/triangle-mesh-io.cc itself is NOT built: it needs
// material-param.h -> mpark/variant.hpp (absent; no stand-ins are written), so the MTL-key -> parameter conversion
// (triangle-mesh-io.cc:34-212) is checked against a restatement in the tests, not against this library.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "image-utils.h"
#include "io/curve-mesh-io.h"
#include "io/image-io.h"
#include "io/tiny_obj_loader.h"
#include "io/tinyexr.h"
#include "io/stb_image_write.h"

namespace {
struct ObjResult {
  bool ok = false;
  tinyobj::attrib_t attrib;
  std::vector<tinyobj::shape_t> shapes;
  std::vector<tinyobj::material_t> materials;
  std::string warn, err;
  // flattened
  std::vector<int32_t> idx;          // per corner: v, vn, vt
  std::vector<int32_t> shape_first;  // per shape: first corner (size shapes+1)
  std::vector<int32_t> mat_ids;      // per face, all shapes concatenated
  std::string text;                  // shape names / material names / unknown parameters, '\n' separated records
};
struct HairResult {
  bool ok = false;
  std::vector<float> vt;
  std::vector<uint32_t> indices;
};
struct ImageResult {
  bool ok = false;
  std::vector<float> px;
  size_t w = 0, h = 0, c = 0;
};
}  // namespace

extern "C" {

// tinyobj::LoadObj exactly as triangle-mesh-io.cc:232-236 calls it (triangulate = true)
void* refio_obj_load(const char* filename, const char* base_dir) {
  ObjResult* r = new ObjResult();
  const char* base_path = (base_dir == nullptr || std::string(base_dir) == "/") ? nullptr : base_dir;
  r->ok = tinyobj::LoadObj(&r->attrib, &r->shapes, &r->materials, &r->warn, &r->err, filename, base_path, true);
  r->shape_first.push_back(0);
  for (const auto& s : r->shapes) {
    for (const auto& i : s.mesh.indices) {
      r->idx.push_back(i.vertex_index);
      r->idx.push_back(i.normal_index);
      r->idx.push_back(i.texcoord_index);
    }
    r->shape_first.push_back(int32_t(r->idx.size() / 3));
    for (int m : s.mesh.material_ids) r->mat_ids.push_back(m);
    r->text += "shape\t" + s.name + "\n";
  }
  for (const auto& m : r->materials) {
    r->text += "material\t" + m.name + "\n";
    for (const auto& kv : m.unknown_parameter) r->text += "param\t" + kv.first + "\t" + kv.second + "\n";
  }
  return r;
}
int refio_obj_ok(void* h) { return static_cast<ObjResult*>(h)->ok ? 1 : 0; }
// which: 0 vertices, 1 normals, 2 texcoords (float); 3 corner indices, 4 shape_first, 5 material ids (int32)
size_t refio_obj_size(void* h, int which) {
  ObjResult* r = static_cast<ObjResult*>(h);
  switch (which) {
    case 0: return r->attrib.vertices.size();
    case 1: return r->attrib.normals.size();
    case 2: return r->attrib.texcoords.size();
    case 3: return r->idx.size();
    case 4: return r->shape_first.size();
    case 5: return r->mat_ids.size();
    case 6: return r->text.size();
  }
  return 0;
}
const void* refio_obj_data(void* h, int which) {
  ObjResult* r = static_cast<ObjResult*>(h);
  switch (which) {
    case 0: return r->attrib.vertices.data();
    case 1: return r->attrib.normals.data();
    case 2: return r->attrib.texcoords.data();
    case 3: return r->idx.data();
    case 4: return r->shape_first.data();
    case 5: return r->mat_ids.data();
    case 6: return r->text.data();
  }
  return nullptr;
}
void refio_obj_free(void* h) { delete static_cast<ObjResult*>(h); }

// tinyobj::ParseTextureNameAndOption as triangle-mesh-io.cc:121-135 uses it: returns the file name and colorspace
int refio_parse_texopt(const char* value, char* name_out, size_t name_cap, char* cs_out, size_t cs_cap) {
  std::string name;
  tinyobj::texture_option_t opt;
  const bool ok = tinyobj::ParseTextureNameAndOption(&name, &opt, value);
  snprintf(name_out, name_cap, "%s", name.c_str());
  snprintf(cs_out, cs_cap, "%s", opt.colorspace.c_str());
  return ok ? 1 : 0;
}

// io::LoadCurveMeshAsCubicBezierCurve(filepath, memory_saving_mode, &vertices_thickness, &indices)
// (curve-mesh-io.cc:32-119)
void* refio_hair_load(const char* filepath, int memory_saving_mode) {
  HairResult* r = new HairResult();
  r->ok = pbrlab::io::LoadCurveMeshAsCubicBezierCurve(filepath, memory_saving_mode != 0, &r->vt, &r->indices);
  return r;
}
int refio_hair_ok(void* h) { return static_cast<HairResult*>(h)->ok ? 1 : 0; }
size_t refio_hair_size(void* h, int which) {
  HairResult* r = static_cast<HairResult*>(h);
  return which == 0 ? r->vt.size() : r->indices.size();
}
const void* refio_hair_data(void* h, int which) {
  HairResult* r = static_cast<HairResult*>(h);
  return which == 0 ? static_cast<const void*>(r->vt.data()) : static_cast<const void*>(r->indices.data());
}
void refio_hair_free(void* h) { delete static_cast<HairResult*>(h); }

// io::LoadImageFromFile<float> (image-io.cc:98-152)
void* refio_image_load(const char* filename, const char* asset_path) {
  ImageResult* r = new ImageResult();
  r->ok = pbrlab::io::LoadImageFromFile(std::string(filename), std::string(asset_path), &r->px, &r->w, &r->h, &r->c);
  return r;
}
int refio_image_ok(void* h) { return static_cast<ImageResult*>(h)->ok ? 1 : 0; }
void refio_image_dims(void* h, size_t* w, size_t* hh, size_t* c) {
  ImageResult* r = static_cast<ImageResult*>(h);
  *w = r->w; *hh = r->h; *c = r->c;
}
const float* refio_image_data(void* h) { return static_cast<ImageResult*>(h)->px.data(); }
void refio_image_free(void* h) { delete static_cast<ImageResult*>(h); }

// pbrlab-cli.cc:47-57: color = rgba / count; LinerToSrgb; WritePNG (float -> x*256 clamp -> stb PNG)
int refio_cli_output(const char* filename, const char* dir, const float* rgba, const uint32_t* count, size_t width,
                     size_t height) {
  std::vector<float> color(width * height * 4);
  for (size_t i = 0; i < width * height; ++i) {
    color[i * 4 + 0] = rgba[i * 4 + 0] / float(count[i]);
    color[i * 4 + 1] = rgba[i * 4 + 1] / float(count[i]);
    color[i * 4 + 2] = rgba[i * 4 + 2] / float(count[i]);
    color[i * 4 + 3] = rgba[i * 4 + 3] / float(count[i]);
  }
  pbrlab::LinerToSrgb(color, width, height, 4, &color);
  return pbrlab::io::WritePNG(std::string(filename), std::string(dir), color, width, height, 4) ? 1 : 0;
}

// test-file generator: tinyexr's own writer.  planes: nchan planes of w*h floats, names: nchan strings separated by '\0'
// (must be in alphabetical order, as the format requires); half != 0 stores HALF channels; compression = TINYEXR_COMPRESSIONTYPE_*
int refio_save_exr(const char* filename, const float* planes, const char* names, int nchan, int w, int h, int half,
                   int compression, int line_order) {
  EXRHeader header;
  InitEXRHeader(&header);
  EXRImage image;
  InitEXRImage(&image);
  image.num_channels = nchan;
  std::vector<const float*> ptrs;
  for (int c = 0; c < nchan; c++) ptrs.push_back(planes + size_t(c) * w * h);
  image.images = reinterpret_cast<unsigned char**>(const_cast<float**>(ptrs.data()));
  image.width = w, image.height = h;
  header.num_channels = nchan;
  header.channels = static_cast<EXRChannelInfo*>(malloc(sizeof(EXRChannelInfo) * size_t(nchan)));
  header.pixel_types = static_cast<int*>(malloc(sizeof(int) * size_t(nchan)));
  header.requested_pixel_types = static_cast<int*>(malloc(sizeof(int) * size_t(nchan)));
  const char* nm = names;
  for (int c = 0; c < nchan; c++) {
    memset(&header.channels[c], 0, sizeof(EXRChannelInfo));
    strncpy(header.channels[c].name, nm, 255);
    nm += strlen(nm) + 1;
    header.pixel_types[c] = TINYEXR_PIXELTYPE_FLOAT;
    header.requested_pixel_types[c] = half ? TINYEXR_PIXELTYPE_HALF : TINYEXR_PIXELTYPE_FLOAT;
  }
  header.compression_type = compression;
  header.line_order = line_order;
  const char* err = nullptr;
  const int ret = SaveEXRImageToFile(&image, &header, filename, &err);
  free(header.channels), free(header.pixel_types), free(header.requested_pixel_types);
  return ret == TINYEXR_SUCCESS ? 1 : 0;
}

// test-file generator: stb_image_write's JPEG writer (baseline, YCbCr, 4:2:0 when quality <= 90 else 4:4:4)
int refio_write_jpg(const char* filename, const unsigned char* px, int w, int h, int comp, int quality) {
  return stbi_write_jpg(filename, w, h, comp, px, quality) ? 1 : 0;
}

int refio_write_png_u8(const char* filename, const char* dir, const unsigned char* px, size_t width, size_t height,
                       size_t channels) {
  std::vector<unsigned char> v(px, px + width * height * channels);
  return pbrlab::io::WritePNG(std::string(filename), std::string(dir), v, width, height, channels) ? 1 : 0;
}

}  // extern "C"
