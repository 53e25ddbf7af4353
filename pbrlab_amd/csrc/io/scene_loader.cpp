// scene_loader.cpp -- see scene_loader.h
#include "scene_loader.h"

#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "hair_reader.h"
#include "image_codec.h"

namespace pbio {
namespace {

std::string parent_dir(const std::string& path) {  // fs::path(path).parent_path()
  const size_t slash = path.find_last_of('/');
  if (slash == std::string::npos) return "";
  if (slash == 0) return "/";
  return path.substr(0, slash);
}

std::string extension(const std::string& path) {  // fs::path(path).extension()
  const size_t slash = path.find_last_of('/');
  const size_t start = slash == std::string::npos ? 0 : slash + 1;
  const size_t dot = path.find_last_of('.');
  if (dot == std::string::npos || dot < start || dot == start) return "";
  return path.substr(dot);
}

// ConvertTinyObjMaterialFloat (:53-62): atof of the value text
void key_float(const ObjMaterial& m, const char* key, float* out) {
  auto it = m.params.find(key);
  if (it != m.params.end()) *out = static_cast<float>(std::atof(it->second.c_str()));
}

// ConvertTinyObjMaterialFloat3 (:39-51, :64-79): sscanf("%lf %lf %lf"); components that do not scan are 0
void key_float3(const ObjMaterial& m, const char* key, float* out) {
  auto it = m.params.find(key);
  if (it == m.params.end()) return;
  double x = 0.0, y = 0.0, z = 0.0;
  sscanf(it->second.c_str(), "%lf %lf %lf", &x, &y, &z);
  out[0] = static_cast<float>(x), out[1] = static_cast<float>(y), out[2] = static_cast<float>(z);
}

bool is_hdr_name(const std::string& name) {  // IsHdr (:108-116)
  std::string e = extension(name);
  for (char& c : e) c = char(tolower(c));
  return e == ".exr" || e == ".hdr";
}

// LoadTextureFromTinyObjMaterial + LoadTexture (:81-137)
void key_texture(const ObjMaterial& m, const char* key, const std::string& base_dir, uint32_t* tex_id,
                 std::vector<LoadedTexture>* textures) {
  auto it = m.params.find(key);
  if (it == m.params.end()) return;
  std::string file, colorspace;
  ParseTextureStatement(it->second.c_str(), &file, &colorspace);
  const bool degamma = (colorspace.empty() || colorspace == "sRGB") && !is_hdr_name(file);
  LoadedTexture t;
  size_t w = 0, h = 0, c = 0;
  if (!LoadImageFromFile(file, base_dir, &t.pixels, &w, &h, &c)) {
    *tex_id = uint32_t(-1);
    return;
  }
  if (degamma) SrgbToLiner(t.pixels, w, h, c, &t.pixels);
  t.width = uint32_t(w), t.height = uint32_t(h), t.channels = uint32_t(c), t.name = file;
  *tex_id = uint32_t(textures->size());
  textures->push_back(std::move(t));
  std::cout << "Loaded texture for " << key << " : " << file << std::endl;
}

const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

bool check(int rc, std::string* err) {
  if (rc == PBRHIP_OK) return true;
  *err = pbrhip_last_error();
  return false;
}

}  // namespace

pbrhip_principled_param DefaultPrincipledParam() {
  pbrhip_principled_param p = {};
  p.base_color[0] = p.base_color[1] = p.base_color[2] = 0.8f;
  p.subsurface = 0.0f;
  p.subsurface_radius[0] = p.subsurface_radius[1] = p.subsurface_radius[2] = 1.0f;
  p.subsurface_color[0] = 0.7f, p.subsurface_color[1] = 0.1f, p.subsurface_color[2] = 0.1f;
  p.metallic = 0.0f, p.specular = 0.5f, p.specular_tint = 0.0f, p.roughness = 0.5f;
  p.anisotropic = 0.0f, p.anisotropic_rotation = 0.0f, p.sheen = 0.0f, p.sheen_tint = 0.5f;
  p.clearcoat = 0.0f, p.clearcoat_roughness = 0.03f, p.ior = 1.45f, p.transmission = 0.0f, p.transmission_roughness = 0.0f;
  p.base_color_tex_id = p.subsurface_color_tex_id = uint32_t(-1);
  return p;
}

pbrhip_hair_param DefaultHairParam() {
  pbrhip_hair_param p = {};
  p.coloring_hair = 1;  // kMelanin
  p.base_color[0] = 0.18f, p.base_color[1] = 0.06f, p.base_color[2] = 0.02f;
  p.melanin = 0.5f, p.melanin_redness = 0.8f, p.melanin_randomize = 0.0f;
  p.roughness = 0.2f, p.azimuthal_roughness = 0.3f, p.ior = 1.55f, p.shift = 2.0f;
  for (int i = 0; i < 3; ++i) p.specular_tint[i] = p.second_specular_tint[i] = p.transmission_tint[i] = 1.0f;
  return p;
}

bool LoadTriangleMeshFromObj(const std::string& filename, ObjScene* out) {
  const std::string base_dir = parent_dir(filename);
  std::cerr << "base dir : " << base_dir << std::endl;
  ObjFile& f = out->parsed;
  const bool ok = ReadObj(filename, base_dir == "/" ? "" : base_dir, &f);
  if (!f.warn.empty()) std::cerr << "warning : " << f.warn << std::endl;
  if (!ok || !f.err.empty()) {
    std::cerr << "error";
    if (!f.err.empty()) std::cerr << " : " << f.err;
    std::cerr << std::endl;
  }
  if (!ok) return false;

  const size_t nv = f.vertices.size() / 3, nn = f.normals.size() / 3, nt = f.texcoords.size() / 2;
  out->vertices_xyzw.resize(nv * 4);
  for (size_t i = 0; i < nv; ++i) {
    for (int k = 0; k < 3; ++k) out->vertices_xyzw[i * 4 + k] = f.vertices[i * 3 + k];
    out->vertices_xyzw[i * 4 + 3] = 1.0f;
  }
  out->normals_xyzw.resize(nn * 4);
  for (size_t i = 0; i < nn; ++i) {
    for (int k = 0; k < 3; ++k) out->normals_xyzw[i * 4 + k] = f.normals[i * 3 + k];
    out->normals_xyzw[i * 4 + 3] = 1.0f;
  }
  out->texcoords_uv.resize(nt * 2);
  for (size_t i = 0; i < nt; ++i) {
    out->texcoords_uv[i * 2 + 0] = f.texcoords[i * 2 + 0];
    out->texcoords_uv[i * 2 + 1] = 1.f - f.texcoords[i * 2 + 1];
  }
  out->meshes.clear();
  for (const ObjShape& s : f.shapes) {
    LoadedMesh m;
    m.name = s.name;
    for (const ObjCorner& c : s.corners) {
      m.vertex_ids.push_back(uint32_t(c.v));
      m.normal_ids.push_back(uint32_t(c.vn));
      m.texcoord_ids.push_back(uint32_t(c.vt));
    }
    for (int id : s.material_ids) m.material_ids.push_back(uint32_t(id));
    out->meshes.push_back(std::move(m));
  }
  for (const ObjMaterial& m : f.materials) {  // ParseTinyObjMaterial (:139-212); textures load in this order
    pbrhip_principled_param p = DefaultPrincipledParam();
    key_float3(m, "base_color", p.base_color);
    key_texture(m, "map_base_color", base_dir, &p.base_color_tex_id, &out->textures);
    key_float(m, "subsurface", &p.subsurface);
    key_float3(m, "subsurface_radius", p.subsurface_radius);
    key_float3(m, "subsurface_color", p.subsurface_color);
    key_texture(m, "map_subsurface_color", base_dir, &p.subsurface_color_tex_id, &out->textures);
    key_float(m, "metallic", &p.metallic);
    key_float(m, "specular", &p.specular);
    key_float(m, "specular_tint", &p.specular_tint);
    key_float(m, "roughness", &p.roughness);
    key_float(m, "anisotropic", &p.anisotropic);
    key_float(m, "anisotropic_rotation", &p.anisotropic_rotation);
    key_float(m, "sheen", &p.sheen);
    key_float(m, "sheen_tint", &p.sheen_tint);
    key_float(m, "clearcoat", &p.clearcoat);
    key_float(m, "clearcoat_roughness", &p.clearcoat_roughness);
    key_float(m, "ior", &p.ior);
    key_float(m, "transmission", &p.transmission);
    key_float(m, "transmission_roughness", &p.transmission_roughness);
    out->materials.push_back(p);
    out->material_names.push_back(m.name);
  }
  return true;
}

// The elements of `src` (stride floats each) that `ids` references, in order of first use; `ids` is rewritten to index
// them.  uint32(-1) ("none") stays as it is.
static void compact_attribute(const std::vector<float>& src, size_t stride, std::vector<uint32_t>* ids, std::vector<float>* out) {
  std::vector<uint32_t> remap(src.size() / stride, 0xFFFFFFFFu);
  out->clear();
  for (uint32_t& id : *ids) {
    if (id == 0xFFFFFFFFu || id >= remap.size()) continue;  // (out-of-range ids are left for the library to report)
    if (remap[id] == 0xFFFFFFFFu) {
      remap[id] = uint32_t(out->size() / stride);
      out->insert(out->end(), src.begin() + size_t(id) * stride, src.begin() + size_t(id + 1) * stride);
    }
    id = remap[id];
  }
}

bool AddObjToScene(pbrhip_scene* scene, const std::string& obj_filename, std::string* err) {
  ObjScene o;
  if (!LoadTriangleMeshFromObj(obj_filename, &o)) {
    std::cerr << "Faild loading obj file [" << obj_filename << "]" << std::endl;
    *err = "failed loading obj file [" + obj_filename + "]";
    return false;
  }
  std::cerr << "Load obj file [" << obj_filename << "]" << std::endl;

  // pc-common.cc:116-139.  The reference adds the materials BEFORE the textures and rewrites the texture ids only in
  // its local copy afterwards, so the scene's materials keep ids that index this file's own texture list.  Same here:
  // for one .obj (or when only the first .obj has textures) the two numberings coincide.
  std::vector<uint32_t> material_ids;
  for (const pbrhip_principled_param& p : o.materials) {
    uint32_t id;
    if (!check(pbrhip_scene_add_principled_material(scene, &p, &id), err)) return false;
    material_ids.push_back(id);
  }
  for (const LoadedTexture& t : o.textures) {
    uint32_t id;
    if (!check(pbrhip_scene_add_texture(scene, t.pixels.data(), t.width, t.height, t.channels, &id), err)) return false;
  }

  std::cerr << "The Number of shapes is " << o.meshes.size() << " in [" << obj_filename << "]" << std::endl;
  for (LoadedMesh& m : o.meshes) {
    const uint32_t nf = uint32_t(m.vertex_ids.size() / 3);
    std::cerr << "  add shape [" << m.name << "]" << std::endl;
    std::cerr << "    num face : " << nf << std::endl;
    for (uint32_t f = 0; f < nf; ++f) {  // material_ids.at(...) throws in the reference when a face has no material
      const uint32_t local = m.material_ids[f];
      if (local >= material_ids.size()) {
        *err = "shape [" + m.name + "] in [" + obj_filename + "]: face " + std::to_string(f) +
               " has no material (no usemtl, or the material is not in the .mtl)";
        return false;
      }
      m.material_ids[f] = material_ids[local];
    }
    if (nf == 0) {  // a shape of only l/p statements: nothing to intersect
      std::cerr << "    (no faces: skipped)" << std::endl;
      continue;
    }
    // The reference's meshes share one Attribute through a shared_ptr (triangle-mesh-io.cc:246-262); the library copies
    // what it is given, so every shape hands over only the vertices / normals / texcoords it references (ids remapped):
    // memory stays linear in the file size however many shapes there are.
    std::vector<float> sv, sn, st;
    std::vector<uint32_t> vid = m.vertex_ids, nid = m.normal_ids, tid = m.texcoord_ids;
    compact_attribute(o.vertices_xyzw, 4, &vid, &sv);
    compact_attribute(o.normals_xyzw, 4, &nid, &sn);
    compact_attribute(o.texcoords_uv, 2, &tid, &st);
    uint32_t mesh, local_scene, geom, instance;
    if (!check(pbrhip_scene_add_triangle_mesh(scene, sv.data(), uint32_t(sv.size() / 4), sn.data(), uint32_t(sn.size() / 4),
                                              st.data(), uint32_t(st.size() / 2), vid.data(), nid.data(), tid.data(),
                                              m.material_ids.data(), nf, &mesh), err) ||
        !check(pbrhip_scene_create_local_scene(scene, &local_scene), err) ||
        !check(pbrhip_scene_add_mesh_to_local_scene(scene, local_scene, mesh, &geom), err) ||
        !check(pbrhip_scene_create_instance(scene, local_scene, kIdentity, &instance), err))
      return false;
    if (m.name.substr(0, 5) == "light") {  // pc-common.cc:172-186
      const float emission[3] = {3.0f, 3.0f, 3.0f};
      uint32_t light;
      if (!check(pbrhip_scene_add_area_light(scene, emission, &light), err)) return false;
      const std::vector<uint32_t> ids(nf, light);
      if (!check(pbrhip_scene_attach_light_ids(scene, instance, 0, ids.data(), nf), err)) return false;
    }
  }
  std::cerr << std::endl;
  return true;
}

bool AddHairToScene(pbrhip_scene* scene, const std::string& filepath, std::string* err) {
  std::vector<float> vt;
  std::vector<uint32_t> indices;
  // curve-mesh-io.cc:121-138 ignores the result of the load: whatever was converted becomes the mesh
  if (!LoadCurveMeshAsCubicBezierCurve(filepath, false, &vt, &indices))
    std::cerr << "warning : [" << filepath << "] was not converted completely" << std::endl;
  std::cerr << "Load curve file [" << filepath << "]" << std::endl;
  std::cerr << "  add shape [" << filepath << "]" << std::endl;
  std::cerr << "  num segments : " << indices.size() << std::endl;
  const pbrhip_hair_param hp = DefaultHairParam();
  uint32_t material;
  if (!check(pbrhip_scene_add_hair_material(scene, &hp, &material), err)) return false;
  if (indices.empty()) {
    *err = "no curve segments in [" + filepath + "]";
    return false;
  }
  const std::vector<uint32_t> mats(indices.size(), material);
  uint32_t mesh, local_scene, geom, instance;
  return check(pbrhip_scene_add_curve_mesh(scene, vt.data(), uint32_t(vt.size() / 4), indices.data(), mats.data(),
                                           uint32_t(indices.size()), &mesh), err) &&
         check(pbrhip_scene_create_local_scene(scene, &local_scene), err) &&
         check(pbrhip_scene_add_mesh_to_local_scene(scene, local_scene, mesh, &geom), err) &&
         check(pbrhip_scene_create_instance(scene, local_scene, kIdentity, &instance), err);
}

bool CreateScene(int argc, const char* const* argv, pbrhip_scene* scene, std::string* err) {
  if (argc < 2) {
    *err = "no scene file";
    return false;
  }
  for (int i = 1; i < argc; ++i) {
    const std::string path(argv[i]);
    const std::string ext = extension(path);
    if (ext == ".obj") {
      if (!AddObjToScene(scene, path, err)) return false;
    } else if (ext == ".hair") {
      if (!AddHairToScene(scene, path, err)) return false;
    }
  }
  if (!check(pbrhip_scene_commit(scene), err)) return false;
  float bmin[3], bmax[3];
  if (!check(pbrhip_scene_aabb(scene, bmin, bmax), err)) return false;
  printf("bmin: %f %f %f\n  bmax: %f %f %f\n", double(bmin[0]), double(bmin[1]), double(bmin[2]), double(bmax[0]),
         double(bmax[1]), double(bmax[2]));
  return true;
}

}  // namespace pbio
