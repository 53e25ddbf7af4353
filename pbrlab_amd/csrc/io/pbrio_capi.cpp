// pbrio_capi.cpp -- C ABI of libpbrhip_io (include/pbrhip_io.h) over the C++ readers of this directory.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "hair_reader.h"
#include "image_codec.h"
#include "pbrhip_io.h"
#include "scene_loader.h"

struct pbrio_obj {
  pbio::ObjScene scene;
  std::string text;
};
struct pbrio_curves {
  std::vector<float> vt;
  std::vector<uint32_t> indices;
};

namespace {
thread_local std::string g_err;
int fail(const std::string& msg) {
  g_err = msg;
  return PBRHIP_EINVAL;
}
template <typename T>
T* dup_buffer(const std::vector<T>& v) {
  T* p = static_cast<T*>(malloc(v.size() * sizeof(T) + 1));
  if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}
}  // namespace

extern "C" {

const char* pbrio_last_error(void) { return g_err.c_str(); }
void pbrio_free(void* p) { free(p); }

int pbrio_obj_load(const char* filename, pbrio_obj** out) {
  try {
  if (!filename || !out) return fail("pbrio_obj_load: NULL argument");
  pbrio_obj* o = new pbrio_obj();
  if (!pbio::LoadTriangleMeshFromObj(filename, &o->scene)) {
    g_err = "cannot load [" + std::string(filename) + "]: " + o->scene.parsed.err;
    delete o;
    return PBRHIP_EINVAL;
  }
  for (const pbio::ObjShape& s : o->scene.parsed.shapes) o->text += "shape\t" + s.name + "\n";
  for (const pbio::ObjMaterial& m : o->scene.parsed.materials) {
    o->text += "material\t" + m.name + "\n";
    for (const auto& kv : m.params) o->text += "param\t" + kv.first + "\t" + kv.second + "\n";
  }
  *out = o;
  return PBRHIP_OK;
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
void pbrio_obj_free(pbrio_obj* o) { delete o; }

size_t pbrio_obj_attribute(const pbrio_obj* o, int which, const float** data) {
  const std::vector<float>* v = which == 0 ? &o->scene.vertices_xyzw : which == 1 ? &o->scene.normals_xyzw : &o->scene.texcoords_uv;
  if (data) *data = v->data();
  return v->size();
}
uint32_t pbrio_obj_num_shapes(const pbrio_obj* o) { return uint32_t(o->scene.meshes.size()); }
const char* pbrio_obj_shape_name(const pbrio_obj* o, uint32_t s) {
  return s < o->scene.meshes.size() ? o->scene.meshes[s].name.c_str() : nullptr;
}
size_t pbrio_obj_shape_ids(const pbrio_obj* o, uint32_t s, int which, const uint32_t** data) {
  if (s >= o->scene.meshes.size()) return 0;
  const pbio::LoadedMesh& m = o->scene.meshes[s];
  const std::vector<uint32_t>* v = which == 0 ? &m.vertex_ids : which == 1 ? &m.normal_ids : which == 2 ? &m.texcoord_ids : &m.material_ids;
  if (data) *data = v->data();
  return v->size();
}
uint32_t pbrio_obj_num_materials(const pbrio_obj* o) { return uint32_t(o->scene.materials.size()); }
int pbrio_obj_material(const pbrio_obj* o, uint32_t i, pbrhip_principled_param* out, const char** name) {
  try {
  if (i >= o->scene.materials.size()) return fail("material index out of range");
  if (out) *out = o->scene.materials[i];
  if (name) *name = o->scene.material_names[i].c_str();
  return PBRHIP_OK;
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
uint32_t pbrio_obj_num_textures(const pbrio_obj* o) { return uint32_t(o->scene.textures.size()); }
int pbrio_obj_texture(const pbrio_obj* o, uint32_t i, const float** pixels, uint32_t* width, uint32_t* height,
                      uint32_t* channels, const char** name) {
  try {
  if (i >= o->scene.textures.size()) return fail("texture index out of range");
  const pbio::LoadedTexture& t = o->scene.textures[i];
  if (pixels) *pixels = t.pixels.data();
  if (width) *width = t.width;
  if (height) *height = t.height;
  if (channels) *channels = t.channels;
  if (name) *name = t.name.c_str();
  return PBRHIP_OK;
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
const char* pbrio_obj_text(const pbrio_obj* o) { return o->text.c_str(); }
const char* pbrio_obj_warnings(const pbrio_obj* o) { return o->scene.parsed.warn.c_str(); }

int pbrio_parse_texture_statement(const char* value, char* texname, size_t texname_cap, char* colorspace, size_t colorspace_cap) {
  try {
  std::string name, cs;
  const bool found = value && pbio::ParseTextureStatement(value, &name, &cs);
  if (texname && texname_cap) snprintf(texname, texname_cap, "%s", name.c_str());
  if (colorspace && colorspace_cap) snprintf(colorspace, colorspace_cap, "%s", cs.c_str());
  return found ? 1 : 0;
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}

int pbrio_curves_load(const char* filepath, int memory_saving_mode, pbrio_curves** out) {
  try {
  if (!filepath || !out) return fail("pbrio_curves_load: NULL argument");
  pbrio_curves* c = new pbrio_curves();
  const bool ok = pbio::LoadCurveMeshAsCubicBezierCurve(filepath, memory_saving_mode != 0, &c->vt, &c->indices);
  *out = c;  // like the reference, the strands converted before a failure stay available
  if (!ok) return fail("[" + std::string(filepath) + "] was not converted completely");
  return PBRHIP_OK;
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
void pbrio_curves_free(pbrio_curves* c) { delete c; }
size_t pbrio_curves_vertices(const pbrio_curves* c, const float** d) {
  if (d) *d = c->vt.data();
  return c->vt.size();
}
size_t pbrio_curves_indices(const pbrio_curves* c, const uint32_t** d) {
  if (d) *d = c->indices.data();
  return c->indices.size();
}

int pbrio_scene_add_obj(pbrhip_scene* s, const char* f) {
  try {
  if (!s || !f) return fail("pbrio_scene_add_obj: NULL argument");
  std::string err;
  return pbio::AddObjToScene(s, f, &err) ? PBRHIP_OK : fail(err);
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
int pbrio_scene_add_hair(pbrhip_scene* s, const char* f) {
  try {
  if (!s || !f) return fail("pbrio_scene_add_hair: NULL argument");
  std::string err;
  return pbio::AddHairToScene(s, f, &err) ? PBRHIP_OK : fail(err);
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
int pbrio_create_scene(int argc, const char* const* argv, pbrhip_scene* s) {
  try {
  if (!s || !argv) return fail("pbrio_create_scene: NULL argument");
  std::string err;
  return pbio::CreateScene(argc, argv, s, &err) ? PBRHIP_OK : fail(err);
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}

int pbrio_image_load(const char* filename, const char* asset_path, float** pixels, size_t* w, size_t* h, size_t* c) {
  try {
  if (!filename || !pixels || !w || !h || !c) return fail("pbrio_image_load: NULL argument");
  std::vector<float> px;
  if (!pbio::LoadImageFromFile(filename, asset_path ? asset_path : "", &px, w, h, c)) return fail("cannot load image [" + std::string(filename) + "]");
  *pixels = dup_buffer(px);
  return PBRHIP_OK;
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
int pbrio_write_png_f32(const char* filename, const char* asset_path, const float* px, size_t w, size_t h, size_t c) {
  try {
  if (!filename || !px) return fail("pbrio_write_png_f32: NULL argument");
  const std::vector<float> v(px, px + w * h * c);
  return pbio::WritePNG(filename, asset_path ? asset_path : "", v, w, h, c) ? PBRHIP_OK : fail("WritePNG failed");
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
int pbrio_write_png_u8(const char* filename, const char* asset_path, const uint8_t* px, size_t w, size_t h, size_t c) {
  try {
  if (!filename || !px) return fail("pbrio_write_png_u8: NULL argument");
  const std::vector<uint8_t> v(px, px + w * h * c);
  return pbio::WritePNG(filename, asset_path ? asset_path : "", v, w, h, c) ? PBRHIP_OK : fail("WritePNG failed");
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
int pbrio_png_decode(const uint8_t* file, size_t n, uint8_t** pixels, size_t* w, size_t* h, size_t* c) {
  try {
  if (!file || !pixels || !w || !h || !c) return fail("pbrio_png_decode: NULL argument");
  std::vector<uint8_t> px;
  std::string err;
  if (!pbio::DecodePng(file, n, &px, w, h, c, &err)) return fail(err);
  *pixels = dup_buffer(px);
  return PBRHIP_OK;
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
int pbrio_layer_to_srgb8(const float* rgba, const uint32_t* count, size_t w, size_t h, uint8_t* out) {
  try {
  if (!rgba || !count || !out) return fail("pbrio_layer_to_srgb8: NULL argument");
  std::vector<uint8_t> v;
  pbio::ResolveLayerToSrgb8(rgba, count, w, h, &v);
  memcpy(out, v.data(), v.size());
  return PBRHIP_OK;
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}
int pbrio_write_layer_png(const char* filename, const char* asset_path, const float* rgba, const uint32_t* count,
                          size_t w, size_t h) {
  try {
  if (!filename || !rgba || !count) return fail("pbrio_write_layer_png: NULL argument");
  std::vector<uint8_t> v;
  pbio::ResolveLayerToSrgb8(rgba, count, w, h, &v);
  return pbio::WritePNG(filename, asset_path ? asset_path : "", v, w, h, 4) ? PBRHIP_OK : fail("WritePNG failed");
  } catch (const std::bad_alloc&) {
    g_err = "out of memory";
    return PBRHIP_ENOMEM;
  } catch (const std::exception& e) {
    return fail(e.what());
  }
}

}  // extern "C"
