// pbrhip.cpp -- C ABI (include/pbrhip.h): host scene store, commit (light tables, BVH, upload) and the
// wavefront render loop that drives kernels.hip.  Host C++ only; device code lives in kernels.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "scene_impl.h"

using namespace pb;

static_assert(sizeof(pbrhip_principled_param) == sizeof(PrincipledParam), "param layout");
static_assert(sizeof(pbrhip_hair_param) == sizeof(HairParam), "param layout");
static_assert(sizeof(pbrhip_hit) == sizeof(HookHit), "hit layout");
static_assert(sizeof(pbrhip_ray) == 32, "ray layout");
static_assert(sizeof(LightRec) == 80, "light record layout");

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int g_device = 0;

int pb::fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
int pb::current_device() { return g_device; }

extern "C" const char* pbrhip_last_error(void) { return g_err.c_str(); }
extern "C" uint32_t pbrhip_abi_version(void) { return PBRHIP_ABI_VERSION; }
extern "C" uint32_t pbrhip_math_mode(void) { return pb::kMathMode; }
extern "C" size_t pbrhip_sizeof_render_stats(void) { return sizeof(pbrhip_render_stats); }

extern "C" int pbrhip_device_count(int* count) {
  if (!count) return fail(PBRHIP_EINVAL, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(PBRHIP_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return PBRHIP_OK;
}
extern "C" int pbrhip_set_device(int device) {
  int n = 0;
  int rc = pbrhip_device_count(&n);
  if (rc) return rc;
  if (device < 0 || device >= n) return fail(PBRHIP_EINVAL, "device %d out of range (%d devices)", device, n);
  g_device = device;
  return PBRHIP_OK;
}

static uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* e = getenv(name);
  return e ? (uint32_t)strtoul(e, nullptr, 10) : dflt;
}

// ------------------------------------------------------------------ scene construction
extern "C" int pbrhip_scene_create(pbrhip_scene** out) {
  return guarded([&]() -> int {
  if (!out) return fail(PBRHIP_EINVAL, "out is NULL");
  int n = 0;
  int rc = pbrhip_device_count(&n);
  if (rc) return rc;
  if (n <= 0) return fail(PBRHIP_ENODEVICE, "no HIP device available: libpbrhip has no CPU fallback");
  HIPCHK(hipSetDevice(g_device));
  std::unique_ptr<pbrhip_scene> s(new pbrhip_scene());
  s->device = g_device;
  HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIPCHK(hipHostMalloc((void**)&s->h_counts, sizeof(uint32_t) * kCntNum * kMaxGroups, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&s->h_ring, sizeof(uint32_t) * 4 * kRingSlots * kMaxGroups, hipHostMallocDefault));
  memset(s->h_ring, 0, sizeof(uint32_t) * 4 * kRingSlots * kMaxGroups);
  HIPCHK(hipHostGetDevicePointer((void**)&s->d_ring, s->h_ring, 0));
  memset(&s->dscene, 0, sizeof(s->dscene));
  *out = s.release();
  return PBRHIP_OK;
  });
}

extern "C" int pbrhip_scene_destroy(pbrhip_scene* s) {
  return guarded([&]() -> int {
  if (!s) return PBRHIP_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
  for (hipStream_t g : s->group_streams) (void)hipStreamDestroy(g);
  if (s->h_counts) (void)hipHostFree(s->h_counts);
  if (s->h_ring) (void)hipHostFree(s->h_ring);
  hipStream_t st = s->stream;
  delete s;
  if (st) (void)hipStreamDestroy(st);
  return PBRHIP_OK;
  });
}

extern "C" int pbrhip_scene_add_triangle_mesh(pbrhip_scene* s, const float* vertices_xyzw, uint32_t num_vertices,
                                              const float* normals_xyzw, uint32_t num_normals,
                                              const float* texcoords_uv, uint32_t num_texcoords,
                                              const uint32_t* vertex_ids, const uint32_t* normal_ids,
                                              const uint32_t* texcoord_ids, const uint32_t* material_ids,
                                              uint32_t num_faces, uint32_t* mesh_id) {
  return guarded([&]() -> int {
  if (!s || !mesh_id || (!vertices_xyzw && num_vertices) || (!vertex_ids && num_faces))
    return fail(PBRHIP_EINVAL, "add_triangle_mesh: NULL argument");
  if (s->committed) return fail(PBRHIP_ESTATE, "scene already committed");
  for (size_t i = 0; i < (size_t)num_faces * 3; i++)
    if (vertex_ids[i] >= num_vertices) return fail(PBRHIP_EINVAL, "vertex id %u out of range", vertex_ids[i]);
  if (normal_ids)
    for (size_t i = 0; i < (size_t)num_faces * 3; i++)
      if (normal_ids[i] != kNone && normal_ids[i] >= num_normals)
        return fail(PBRHIP_EINVAL, "normal id %u out of range", normal_ids[i]);
  if (texcoord_ids)
    for (size_t i = 0; i < (size_t)num_faces * 3; i++)
      if (texcoord_ids[i] != kNone && texcoord_ids[i] >= num_texcoords)
        return fail(PBRHIP_EINVAL, "texcoord id %u out of range", texcoord_ids[i]);
  HostMesh m;
  m.kind = 0;
  m.nfaces = num_faces;
  m.vertices.assign(vertices_xyzw, vertices_xyzw + (size_t)num_vertices * 4);
  if (num_normals) m.normals.assign(normals_xyzw, normals_xyzw + (size_t)num_normals * 4);
  if (num_texcoords) m.texcoords.assign(texcoords_uv, texcoords_uv + (size_t)num_texcoords * 2);
  m.vid.assign(vertex_ids, vertex_ids + (size_t)num_faces * 3);
  // mesh/triangle-mesh.cc:33-55: missing id arrays become all -1
  if (normal_ids) m.nid.assign(normal_ids, normal_ids + (size_t)num_faces * 3);
  else m.nid.assign((size_t)num_faces * 3, kNone);
  if (texcoord_ids) m.tid.assign(texcoord_ids, texcoord_ids + (size_t)num_faces * 3);
  else m.tid.assign((size_t)num_faces * 3, kNone);
  if (material_ids) m.mat.assign(material_ids, material_ids + num_faces);
  else m.mat.assign(num_faces, kNone);
  *mesh_id = (uint32_t)s->meshes.size();
  s->meshes.push_back(std::move(m));
  return PBRHIP_OK;
  });
}

extern "C" int pbrhip_scene_add_curve_mesh(pbrhip_scene* s, const float* vertices_xyzr, uint32_t num_vertices,
                                           const uint32_t* indices, const uint32_t* material_ids,
                                           uint32_t num_segments, uint32_t* mesh_id) {
  return guarded([&]() -> int {
  if (!s || !mesh_id || (!vertices_xyzr && num_vertices) || (!indices && num_segments))
    return fail(PBRHIP_EINVAL, "add_curve_mesh: NULL argument");
  if (s->committed) return fail(PBRHIP_ESTATE, "scene already committed");
  for (uint32_t i = 0; i < num_segments; i++)
    if ((uint64_t)indices[i] + 4 > num_vertices) return fail(PBRHIP_EINVAL, "curve index %u out of range", indices[i]);
  HostMesh m;
  m.kind = 1;
  m.cverts.assign(vertices_xyzr, vertices_xyzr + (size_t)num_vertices * 4);
  m.cidx.assign(indices, indices + num_segments);
  if (material_ids) m.cmat.assign(material_ids, material_ids + num_segments);
  else m.cmat.assign(num_segments, kNone);
  *mesh_id = (uint32_t)s->meshes.size();
  s->meshes.push_back(std::move(m));
  return PBRHIP_OK;
  });
}

// texture ids are validated at commit (pc/pc-common.cc:116-139 adds materials first, textures after)
static int check_tex(const pbrhip_principled_param*) { return PBRHIP_OK; }
// Scene::AddTexture (scene.h:46-51) with Texture(pixels, width, height, channels) (texture.cc:10-21)
extern "C" int pbrhip_scene_add_texture(pbrhip_scene* s, const float* pixels, uint32_t width, uint32_t height,
                                        uint32_t channels, uint32_t* texture_id) {
  return guarded([&]() -> int {
  if (!s || !pixels || !texture_id) return fail(PBRHIP_EINVAL, "add_texture: NULL argument");
  if (width == 0 || height == 0 || channels == 0 || channels > 4) return fail(PBRHIP_EINVAL, "add_texture: bad shape");
  if (s->committed) return fail(PBRHIP_ESTATE, "scene already committed");
  size_t n = (size_t)width * height * channels;
  if (s->tex_pixels.size() + n >= (1ull << 32)) return fail(PBRHIP_EUNSUPPORTED, "texture pool exceeds 2^32 floats");
  TexDesc t = {(uint32_t)s->tex_pixels.size(), width, height, channels};
  s->tex_pixels.insert(s->tex_pixels.end(), pixels, pixels + n);
  *texture_id = (uint32_t)s->tex_descs.size();
  s->tex_descs.push_back(t);
  return PBRHIP_OK;
  });
}
extern "C" int pbrhip_scene_add_principled_material(pbrhip_scene* s, const pbrhip_principled_param* p, uint32_t* id) {
  return guarded([&]() -> int {
  if (!s || !p || !id) return fail(PBRHIP_EINVAL, "add_principled_material: NULL argument");
  if (int rc = check_tex(p)) return rc;
  HostMaterial m;
  m.kind = kMatPrincipled;
  memcpy(&m.pr, p, sizeof(m.pr));
  memset(&m.hr, 0, sizeof(m.hr));
  *id = (uint32_t)s->materials.size();
  s->materials.push_back(m);
  return PBRHIP_OK;
  });
}
extern "C" int pbrhip_scene_add_hair_material(pbrhip_scene* s, const pbrhip_hair_param* p, uint32_t* id) {
  return guarded([&]() -> int {
  if (!s || !p || !id) return fail(PBRHIP_EINVAL, "add_hair_material: NULL argument");
  HostMaterial m;
  m.kind = kMatHair;
  memset(&m.pr, 0, sizeof(m.pr));
  memcpy(&m.hr, p, sizeof(m.hr));
  *id = (uint32_t)s->materials.size();
  s->materials.push_back(m);
  return PBRHIP_OK;
  });
}
extern "C" int pbrhip_scene_add_area_light(pbrhip_scene* s, const float emission[3], uint32_t* id) {
  return guarded([&]() -> int {
  if (!s || !emission || !id) return fail(PBRHIP_EINVAL, "add_area_light: NULL argument");
  *id = (uint32_t)s->light_params.size();
  s->light_params.push_back(V3(emission[0], emission[1], emission[2]));
  return PBRHIP_OK;
  });
}
extern "C" int pbrhip_scene_create_local_scene(pbrhip_scene* s, uint32_t* id) {
  return guarded([&]() -> int {
  if (!s || !id) return fail(PBRHIP_EINVAL, "create_local_scene: NULL argument");
  *id = (uint32_t)s->locals.size();
  s->locals.emplace_back();
  return PBRHIP_OK;
  });
}
extern "C" int pbrhip_scene_add_mesh_to_local_scene(pbrhip_scene* s, uint32_t local_scene_id, uint32_t mesh_id,
                                                    uint32_t* geom_id) {
  return guarded([&]() -> int {
  if (!s || !geom_id) return fail(PBRHIP_EINVAL, "add_mesh_to_local_scene: NULL argument");
  if (local_scene_id >= s->locals.size() || mesh_id >= s->meshes.size())
    return fail(PBRHIP_EINVAL, "local scene %u / mesh %u out of range", local_scene_id, mesh_id);
  *geom_id = (uint32_t)s->locals[local_scene_id].size();
  s->locals[local_scene_id].push_back(mesh_id);
  return PBRHIP_OK;
  });
}
extern "C" int pbrhip_scene_create_instance(pbrhip_scene* s, uint32_t local_scene_id, const float* transform,
                                            uint32_t* instance_id) {
  return guarded([&]() -> int {
  if (!s || !instance_id) return fail(PBRHIP_EINVAL, "create_instance: NULL argument");
  if (local_scene_id >= s->locals.size()) return fail(PBRHIP_EINVAL, "local scene %u out of range", local_scene_id);
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  HostInstance in;
  in.local_scene = local_scene_id;
  memcpy(in.xf, transform ? transform : ident, sizeof(ident));
  in.identity = memcmp(in.xf, ident, sizeof(ident)) == 0;
  if (!in.identity) {
    // raytracer_impl.cc:49-84 hands the matrix to Embree (row-vector convention, v' = v * M, translation in the last
    // row); everything above the raytracer keeps working in the instance's local space (scene.cc:217,237 "TODO
    // transform").  A matrix that is not invertible has no such instance.
    const float* m = in.xf;
    for (int k = 0; k < 16; k++)
      if (!std::isfinite(m[k])) return fail(PBRHIP_EINVAL, "create_instance: the transform has a non-finite entry");
    const double det = (double)m[0] * ((double)m[5] * m[10] - (double)m[6] * m[9]) - (double)m[1] * ((double)m[4] * m[10] - (double)m[6] * m[8]) +
                       (double)m[2] * ((double)m[4] * m[9] - (double)m[5] * m[8]);
    if (!(det != 0.0)) return fail(PBRHIP_EINVAL, "create_instance: the transform is singular");
  }
  // scene.cc:119-143: material ids are copied from the meshes when the instance is created
  for (uint32_t mid : s->locals[local_scene_id]) {
    const HostMesh& m = s->meshes[mid];
    in.material_ids.push_back(m.kind == 0 ? m.mat : m.cmat);
    in.light_ids.emplace_back();
  }
  *instance_id = (uint32_t)s->instances.size();
  s->instances.push_back(std::move(in));
  return PBRHIP_OK;
  });
}
static const HostMesh* inst_mesh(const pbrhip_scene* s, uint32_t instance_id, uint32_t geom_id) {
  const HostInstance& in = s->instances[instance_id];
  return &s->meshes[s->locals[in.local_scene][geom_id]];
}
extern "C" int pbrhip_scene_attach_light_ids(pbrhip_scene* s, uint32_t instance_id, uint32_t geom_id,
                                             const uint32_t* ids, uint32_t n) {
  return guarded([&]() -> int {
  if (!s || (!ids && n)) return fail(PBRHIP_EINVAL, "attach_light_ids: NULL argument");
  if (instance_id >= s->instances.size() || geom_id >= s->instances[instance_id].light_ids.size())
    return fail(PBRHIP_EINVAL, "instance %u / geom %u out of range", instance_id, geom_id);
  if (n != 0 && n != inst_mesh(s, instance_id, geom_id)->num_prims()) return fail(PBRHIP_ESIZE, "light param error");
  for (uint32_t i = 0; i < n; i++)
    if (ids[i] != kNone && ids[i] >= s->light_params.size()) return fail(PBRHIP_EINVAL, "light id %u out of range", ids[i]);
  s->instances[instance_id].light_ids[geom_id].assign(ids, ids + n);
  return PBRHIP_OK;
  });
}
extern "C" int pbrhip_scene_attach_material_ids(pbrhip_scene* s, uint32_t instance_id, uint32_t geom_id,
                                                const uint32_t* ids, uint32_t n) {
  return guarded([&]() -> int {
  if (!s || (!ids && n)) return fail(PBRHIP_EINVAL, "attach_material_ids: NULL argument");
  if (instance_id >= s->instances.size() || geom_id >= s->instances[instance_id].material_ids.size())
    return fail(PBRHIP_EINVAL, "instance %u / geom %u out of range", instance_id, geom_id);
  if (n != inst_mesh(s, instance_id, geom_id)->num_prims()) return fail(PBRHIP_ESIZE, "material param error");
  s->instances[instance_id].material_ids[geom_id].assign(ids, ids + n);
  return PBRHIP_OK;
  });
}

// ------------------------------------------------------------------ commit
static V3 mesh_vertex(const HostMesh& m, uint32_t prim, int k) {
  const float* p = m.vertices.data() + (size_t)m.vid[prim * 3 + k] * 4;
  return V3(p[0], p[1], p[2]);
}
// What the raytracer sees of an instance (raytracer_impl.cc:61-81: the transform goes to Embree and nowhere else):
// v' = v * M with the translation row, in this order of operations (the checker uses the same expression).
static V3 xf_point(const float* m, V3 v) {
  return V3(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12], m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13],
            m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14]);
}
static V3 world_vertex(const HostInstance& in, const HostMesh& m, uint32_t prim, int k) {
  const V3 v = mesh_vertex(m, prim, k);
  return in.identity ? v : xf_point(in.xf, v);
}
// control points xyzr of curve `prim` as the raytracer sees them (the radius is not scaled)
static void world_curve(const HostInstance& in, const HostMesh& m, uint32_t prim, float out[16]) {
  const float* cps = m.cverts.data() + (size_t)m.cidx[prim] * 4;
  for (int c = 0; c < 4; c++) {
    V3 v(cps[4 * c], cps[4 * c + 1], cps[4 * c + 2]);
    if (!in.identity) v = xf_point(in.xf, v);
    out[4 * c] = v.x, out[4 * c + 1] = v.y, out[4 * c + 2] = v.z, out[4 * c + 3] = cps[4 * c + 3];
  }
}
// TriangleMesh::FetchFaceArea (mesh/triangle-mesh.cc:113-124)
static float face_area(const HostMesh& m, uint32_t prim) {
  V3 p0 = mesh_vertex(m, prim, 0), p1 = mesh_vertex(m, prim, 1), p2 = mesh_vertex(m, prim, 2);
  return length(cross(p1 - p0, p2 - p0)) * 0.5f;
}

// LightManager::RegisterInstanceMesh (light-manager.cc:79-184)
static void register_lights(pbrhip_scene* s, uint32_t instance_id) {
  HostInstance& in = s->instances[instance_id];
  size_t ng = in.light_ids.size();
  in.has_area_light.assign(ng, 0);
  in.area_lights.assign(ng, HostAreaLight());
  for (size_t g = 0; g < ng; g++) {
    const std::vector<uint32_t>& ids = in.light_ids[g];
    if (ids.empty()) continue;
    const HostMesh& m = *inst_mesh(s, instance_id, (uint32_t)g);
    if (m.kind != 0) continue;
    uint32_t nf = m.nfaces;
    bool have = false;
    for (uint32_t f = 0; f < nf; f++) have = have || ids[f] != kNone;
    if (!have) continue;
    HostAreaLight& a = in.area_lights[g];
    in.has_area_light[g] = 1;
    a.light_param_ids = ids;
    a.choose_prob.assign(nf, 0.f);
    for (uint32_t f = 0; f < nf; f++) {
      float intensity = 0.0f;
      if (ids[f] != kNone) intensity = spectrum_norm(s->light_params[ids[f]]);
      a.choose_prob[f] = intensity * face_area(m, f);
    }
    a.intensity_sum = std::accumulate(a.choose_prob.begin(), a.choose_prob.end(), 0.0f);
    const float sum = a.intensity_sum;
    for (float& v : a.choose_prob) v = v / sum;
    a.cdf = a.choose_prob;
    for (uint32_t f = 0; nf > 0 && f < nf - 1u; f++) a.cdf[f + 1u] += a.cdf[f];
    a.area_pdf.assign(nf, 0.f);
    for (uint32_t f = 0; f < nf; f++)
      if (ids[f] != kNone) a.area_pdf[f] = 1.0f / face_area(m, f);
  }
}
// LightManager::Commit (light-manager.cc:29-77)
static void commit_lights(pbrhip_scene* s) {
  s->lights.clear();
  double intensity_sum = 0.0;
  for (uint32_t i = 0; i < s->instances.size(); i++) {
    HostInstance& in = s->instances[i];
    for (uint32_t g = 0; g < in.area_lights.size(); g++) {
      if (!in.has_area_light[g]) continue;
      in.area_lights[g].global_id = (uint32_t)s->lights.size();
      HostLight L;
      L.choose_prob = in.area_lights[g].intensity_sum;
      L.instance_id = i, L.geom_id = g;
      intensity_sum += (double)L.choose_prob;
      s->lights.push_back(L);
    }
  }
  for (HostLight& L : s->lights) L.choose_prob = (float)((double)L.choose_prob / intensity_sum);
  s->light_cdf.resize(s->lights.size());
  for (size_t l = 0; l < s->lights.size(); l++) s->light_cdf[l] = s->lights[l].choose_prob;
  for (size_t l = 0; !s->light_cdf.empty() && l < s->light_cdf.size() - 1u; l++) s->light_cdf[l + 1u] += s->light_cdf[l];
}

static Material make_material(const HostMaterial& hm) {
  Material m;
  memset(&m, 0, sizeof(m));
  m.kind = hm.kind;
  m.bsdf = default_bsdf();
  if (hm.kind == kMatPrincipled) {
    m.bsdf = param_to_bsdf(hm.pr);
    m.param = hm.pr;
    medium_coefficients(m.bsdf, m.sss_sigt, m.sss_sigs, m.sss_wthr);  // (only read when the subsurface closure is picked)
    m.textured = (hm.pr.base_color_tex_id != kNone || hm.pr.subsurface_color_tex_id != kNone) ? 1u : 0u;
  } else {
    m.hair = hair_param_to_bsdf(hm.hr);
  }
  return m;
}

// End points of linear piece `sub` of a cubic Bezier (control points xyzr): B(sub/4) and B((sub+1)/4), evaluated with
// the arithmetic of the intersection contract (Bernstein weights, products summed left to right, single precision,
// no contraction) so that every back end tests the same segment.
static void bezier_point(const float* cp, float u, float out[4]) {
  const float s = 1.0f - u;
  const float b0 = s * s * s, b1 = 3.0f * u * s * s, b2 = 3.0f * u * u * s, b3 = u * u * u;
  for (int k = 0; k < 4; k++) out[k] = ((cp[k] * b0 + cp[4 + k] * b1) + cp[8 + k] * b2) + cp[12 + k] * b3;
}
static void curve_piece(const float* cp, uint32_t sub, float a[4], float b[4]) {
  bezier_point(cp, (float)sub * 0.25f, a);
  bezier_point(cp, (float)(sub + 1) * 0.25f, b);
}

extern "C" int pbrhip_scene_commit(pbrhip_scene* s) {
  return guarded([&]() -> int {
  if (!s) return fail(PBRHIP_EINVAL, "scene is NULL");
  HIPCHK(hipSetDevice(s->device));
  for (uint32_t i = 0; i < s->instances.size(); i++) register_lights(s, i);
  commit_lights(s);

  // flatten primitives in canonical (instance, geom, prim) order = gid
  std::vector<PrimRef> prims;
  for (uint32_t i = 0; i < s->instances.size(); i++)
    for (uint32_t g = 0; g < s->instances[i].material_ids.size(); g++) {
      const HostMesh& m = *inst_mesh(s, i, g);
      if (s->instances[i].material_ids[g].size() != m.num_prims())
        return fail(PBRHIP_ESIZE, "material param error (instance %u geom %u)", i, g);
      for (uint32_t p = 0; p < m.num_prims(); p++)
        for (uint32_t sub = 0; sub < (m.kind == 1 ? 4u : 1u); sub++) prims.push_back({i, g, p, (uint32_t)m.kind, sub});
    }
  uint32_t np = (uint32_t)prims.size();
  if (np >= (1u << 27)) return fail(PBRHIP_EUNSUPPORTED, "too many primitives (%u)", np);
  std::vector<float> lo(3 * (size_t)np), hi(3 * (size_t)np);
  std::vector<uint8_t> kinds(np);
  const float inf = std::numeric_limits<float>::infinity();
  // Scene bounds (rtcGetSceneBounds, raytracer_impl.cc:199-202; they place the camera): the union of the instances'
  // bounds.  An RTC_GEOMETRY_TYPE_INSTANCE (raytracer_impl.cc:61-81) reports the box of the transformed CORNERS of its local
  // scene's box -- larger than the box of the transformed geometry under rotation or shear; an instance whose matrix is
  // bit for bit the identity reports the local box.  Local box: triangles by their corners, curves by the hull of their
  // control points widened by the largest control radius.  (The tree below is built over the transformed primitives.)
  float bmin[3] = {inf, inf, inf}, bmax[3] = {-inf, -inf, -inf};
  for (uint32_t i = 0; i < s->instances.size(); i++) {
    const HostInstance& inst = s->instances[i];
    float ll[3] = {inf, inf, inf}, lh[3] = {-inf, -inf, -inf};
    bool any = false;
    for (uint32_t g = 0; g < inst.material_ids.size(); g++) {
      const HostMesh& m = *inst_mesh(s, i, g);
      for (uint32_t p = 0; p < m.num_prims(); p++) {
        any = true;
        if (m.kind == 0) {
          for (int c = 0; c < 3; c++) {
            const V3 v = mesh_vertex(m, p, c);
            const float a[3] = {v.x, v.y, v.z};
            for (int k = 0; k < 3; k++) ll[k] = fminf(ll[k], a[k]), lh[k] = fmaxf(lh[k], a[k]);
          }
        } else {
          float r = 0.f, cl[3] = {inf, inf, inf}, ch[3] = {-inf, -inf, -inf};
          for (int c = 0; c < 4; c++) {
            const float* cp = m.cverts.data() + ((size_t)m.cidx[p] + c) * 4;
            r = fmaxf(r, fabsf(cp[3]));
            for (int k = 0; k < 3; k++) cl[k] = fminf(cl[k], cp[k]), ch[k] = fmaxf(ch[k], cp[k]);
          }
          for (int k = 0; k < 3; k++) ll[k] = fminf(ll[k], cl[k] - r), lh[k] = fmaxf(lh[k], ch[k] + r);
        }
      }
    }
    if (!any) continue;
    if (inst.identity) {
      for (int k = 0; k < 3; k++) bmin[k] = fminf(bmin[k], ll[k]), bmax[k] = fmaxf(bmax[k], lh[k]);
    } else {
      for (int c = 0; c < 8; c++) {
        const V3 v = xf_point(inst.xf, V3((c & 1) ? lh[0] : ll[0], (c & 2) ? lh[1] : ll[1], (c & 4) ? lh[2] : ll[2]));
        const float a[3] = {v.x, v.y, v.z};
        for (int k = 0; k < 3; k++) bmin[k] = fminf(bmin[k], a[k]), bmax[k] = fmaxf(bmax[k], a[k]);
      }
    }
  }
  for (uint32_t g = 0; g < np; g++) {
    const PrimRef& pr = prims[g];
    const HostMesh& m = *inst_mesh(s, pr.instance_id, pr.geom_id);
    const HostInstance& inst = s->instances[pr.instance_id];
    float l[3] = {inf, inf, inf}, h[3] = {-inf, -inf, -inf};
    kinds[g] = (uint8_t)pr.kind;
    if (pr.kind == 0) {
      for (int c = 0; c < 3; c++) {
        V3 v = world_vertex(inst, m, pr.prim_id, c);
        float a[3] = {v.x, v.y, v.z};
        for (int k = 0; k < 3; k++) l[k] = std::min(l[k], a[k]), h[k] = std::max(h[k], a[k]);
      }
    } else {
      float cps[16];
      world_curve(inst, m, pr.prim_id, cps);
      // BVH box of this piece: its two end points widened by the larger end radius (the ribbon between them never
      // leaves that box, and a hit is reported at the depth of the axis point)
      float a[4], b[4];
      curve_piece(cps, pr.sub, a, b);
      const float r = std::max(fabsf(a[3]), fabsf(b[3]));
      for (int k = 0; k < 3; k++) l[k] = std::min(a[k], b[k]) - r, h[k] = std::max(a[k], b[k]) + r;
    }
    for (int k = 0; k < 3; k++) lo[3 * g + k] = l[k], hi[3 * g + k] = h[k];
  }
  memcpy(s->bmin, bmin, sizeof(bmin));
  memcpy(s->bmax, bmax, sizeof(bmax));

  FlatBvh bvh;
  bool gpu_built = false;
  uint32_t num_nodes = 0;
  int builder = s->bvh_builder;
  if (const char* e = getenv("PBRHIP_BVH")) builder = (strcmp(e, "gpu") == 0) ? PBRHIP_BVH_GPU_LBVH : PBRHIP_BVH_HOST_SAH;
  if (builder == PBRHIP_BVH_GPU_LBVH && np > 0) {
    HIPCHK(s->d_nodes.reserve(std::max<size_t>(np > 1 ? np - 1 : 1, 1) + np));  // nodes, then one 64-byte slot per primitive
    HIPCHK(build_bvh_gpu(s->stream, lo, hi, kinds, s->d_nodes.p, &bvh.slot_gid, &bvh.depth));
    if (bvh.depth > (uint32_t)kStackDepth) {
      // a Morton-order tree over badly distributed primitives can be deeper than the traversal stack: use the SAH tree
      fprintf(stderr, "pbrhip: GPU-built BVH is %u deep (stack %d): building on the host instead\n", bvh.depth, kStackDepth);
      bvh = FlatBvh();
    } else {
      gpu_built = true;
      num_nodes = np > 1 ? np - 1 : 1;
    }
  }
  if (!gpu_built) {
    build_bvh(lo, hi, kinds, &bvh);
    num_nodes = (uint32_t)bvh.nodes.size();
  }
  s->bvh_built_on_gpu = gpu_built;
  if (bvh.depth > (uint32_t)kStackDepth)
    return fail(PBRHIP_EOVERFLOW, "BVH depth %u exceeds the traversal stack (%d)", bvh.depth, kStackDepth);
  s->bvh_depth = bvh.depth;

  // light records: one per (light, prim), concatenated
  std::vector<LightHead> heads(s->lights.size());
  std::vector<LightRec> lrecs;
  std::vector<float> lprim_cdf;
  for (size_t l = 0; l < s->lights.size(); l++) {
    const HostLight& L = s->lights[l];
    const HostAreaLight& a = s->instances[L.instance_id].area_lights[L.geom_id];
    const HostMesh& m = *inst_mesh(s, L.instance_id, L.geom_id);
    heads[l].first = (uint32_t)lrecs.size();
    heads[l].count = m.nfaces;
    for (uint32_t f = 0; f < m.nfaces; f++) {
      LightRec r;
      memset(&r, 0, sizeof(r));
      V3 p0 = mesh_vertex(m, f, 0), p1 = mesh_vertex(m, f, 1), p2 = mesh_vertex(m, f, 2);
      V3 n = vnormalize(cross(p1 - p0, p2 - p1));  // CalcGeometryNormal (triangle-mesh.cc:181-184)
      r.p0[0] = p0.x, r.p0[1] = p0.y, r.p0[2] = p0.z;
      r.p1[0] = p1.x, r.p1[1] = p1.y, r.p1[2] = p1.z;
      r.p2[0] = p2.x, r.p2[1] = p2.y, r.p2[2] = p2.z;
      r.normal[0] = n.x, r.normal[1] = n.y, r.normal[2] = n.z;
      // light-manager.h:68-70,149-150: choose_light * choose_prim * prim_area_pdf, in that order
      r.pdf = L.choose_prob * a.choose_prob[f] * a.area_pdf[f];
      if (a.light_param_ids[f] != kNone) {
        V3 e = s->light_params[a.light_param_ids[f]];
        r.emission[0] = e.x, r.emission[1] = e.y, r.emission[2] = e.z;
      }
      lrecs.push_back(r);
      lprim_cdf.push_back(a.cdf[f]);
    }
  }

  // leaf-ordered slots (traversal geometry) + one 128-byte ShadeRec per slot (everything shading needs)
  uint32_t ns = (uint32_t)bvh.slot_gid.size();
  if (ns > kHitSlotMask) return fail(PBRHIP_EINVAL, "%u traversal primitives: at most %u are supported", ns, kHitSlotMask);
  std::vector<float4> slots(4 * (size_t)ns);
  std::vector<ShadeRec> shade(ns);
  for (uint32_t k = 0; k < ns; k++) {
    uint32_t g = bvh.slot_gid[k];
    const PrimRef& pr = prims[g];
    const HostInstance& in = s->instances[pr.instance_id];
    const HostMesh& m = *inst_mesh(s, pr.instance_id, pr.geom_id);
    uint32_t mat = in.material_ids[pr.geom_id][pr.prim_id];
    if (mat != kNone && (mat >= s->materials.size() || mat >= 0x00FFFFFFu)) return fail(PBRHIP_EINVAL, "material id %u out of range", mat);
    ShadeRec& sr = shade[k];
    memset(&sr, 0, sizeof(sr));
    uint32_t flags = 0, lightrec = kNone;
    if (mat == kNone) flags |= kSlotMatNone;
    else if (s->materials[mat].kind == kMatHair) flags |= kSlotMatHair;
    float4* sl = &slots[4 * (size_t)k];
    for (int c = 0; c < 4; c++) sl[c] = make_float4(0, 0, 0, 0);
    if (pr.kind == 0) {
      for (int c = 0; c < 3; c++) {
        // traversal: what Embree sees (the transformed triangle); shading: the mesh's own corners -- the geometric normal
        // Embree reports for an instance is in the instance's local space and pbrlab uses it as it is
        const V3 w = world_vertex(in, m, pr.prim_id, c);
        sl[c] = make_float4(w.x, w.y, w.z, 0.f);
      }
      {
        // the two normals every hit on this triangle would otherwise compute from its corners (ShadeRec, dscene.h): the kernels'
        // own functions, evaluated here
        const V3 v0 = mesh_vertex(m, pr.prim_id, 0), v1 = mesh_vertex(m, pr.prim_id, 1), v2 = mesh_vertex(m, pr.prim_id, 2);
        const V3 ng = normalize_raw(cross(v1 - v0, v2 - v0));
        const V3 nf = vnormalize(cross(v1 - v0, v2 - v1));  // CalcGeometryNormal, triangle-mesh.cc:181-184
        sr.ng[0] = ng.x, sr.ng[1] = ng.y, sr.ng[2] = ng.z;
        sr.ns_flat[0] = nf.x, sr.ns_flat[1] = nf.y, sr.ns_flat[2] = nf.z;
      }
      uint32_t a = m.nid[pr.prim_id * 3 + 0], b = m.nid[pr.prim_id * 3 + 1], c = m.nid[pr.prim_id * 3 + 2];
      if (a != kNone && b != kNone && c != kNone) {  // triangle-mesh.cc:81-84
        flags |= kSlotHasNormals;
        const uint32_t idx[3] = {a, b, c};
        for (int q = 0; q < 3; q++) {
          const float* n = m.normals.data() + (size_t)idx[q] * 4;
          sr.n[3 * q + 0] = n[0], sr.n[3 * q + 1] = n[1], sr.n[3 * q + 2] = n[2];
        }
      }
      if (in.has_area_light[pr.geom_id]) {
        const HostAreaLight& al = in.area_lights[pr.geom_id];
        if (al.light_param_ids[pr.prim_id] != kNone) lightrec = heads[al.global_id].first + pr.prim_id;
      }
      uint32_t ta = m.tid[pr.prim_id * 3 + 0], tb = m.tid[pr.prim_id * 3 + 1], tc = m.tid[pr.prim_id * 3 + 2];
      if (ta != kNone && tb != kNone && tc != kNone) {  // triangle-mesh.cc:130-133
        flags |= kSlotHasUV;
        const uint32_t idx[3] = {ta, tb, tc};
        for (int q = 0; q < 3; q++) {
          sr.uv[2 * q + 0] = m.texcoords[(size_t)idx[q] * 2 + 0];
          sr.uv[2 * q + 1] = m.texcoords[(size_t)idx[q] * 2 + 1];
        }
      }
    } else {
      flags |= kSlotIsCurve;
      const float* cps = m.cverts.data() + (size_t)m.cidx[pr.prim_id] * 4;  // local: the tangent (= Ng) shading uses
      float wcps[16];
      world_curve(in, m, pr.prim_id, wcps);
      float a[4], b[4];
      curve_piece(wcps, pr.sub, a, b);
      sl[0] = make_float4(a[0], a[1], a[2], a[3]);
      sl[1] = make_float4(b[0], b[1], b[2], b[3]);
      sl[2] = make_float4(__builtin_bit_cast(float, pr.sub), 0.f, 0.f, 0.f);
      // shading needs the cubic itself (tangent = dP/du at the hit): control points xyzr in words 8..23 of the record
      float* w = reinterpret_cast<float*>(&sr);
      for (int c = 0; c < 16; c++) w[8 + c] = cps[c];
    }
    sr.gid = g, sr.lightrec = lightrec;
    sr.matflags = (mat == kNone ? 0x00FFFFFFu : mat) | (flags << 24);
    const uint32_t route = ((flags & kSlotMatHair) ? kHitHair : 0u) | ((flags & kSlotMatNone) ? kHitNoMaterial : 0u) |
                           (lightrec != kNone ? kHitLight : 0u) |
                           ((flags & (kSlotHasNormals | kSlotHasUV | kSlotIsCurve)) ? kHitMore : 0u);
    sl[2].w = __builtin_bit_cast(float, route);  // travels with the hit record (Hit::slot)
    sr.instance_id = pr.instance_id, sr.geom_id = pr.geom_id, sr.prim_id = pr.prim_id;
  }
  std::vector<Material> mats(s->materials.size());
  s->has_hair = s->has_sss = s->has_textured = false;
  for (size_t i = 0; i < mats.size(); i++) {
    const HostMaterial& hm = s->materials[i];
    if (hm.kind == kMatPrincipled)
      for (uint32_t t : {hm.pr.base_color_tex_id, hm.pr.subsurface_color_tex_id})
        if (t != kNone && t >= s->tex_descs.size()) return fail(PBRHIP_EINVAL, "material %zu: texture id %u out of range", i, t);
    mats[i] = make_material(s->materials[i]);
    if (mats[i].textured) s->has_sss = s->has_textured = true;  // a subsurface_color / base_color map can switch the SSS closure on per hit
    s->has_hair = s->has_hair || mats[i].kind == kMatHair;
    s->has_sss = s->has_sss || (mats[i].kind == kMatPrincipled && mats[i].bsdf.enable_subsurface);
  }

  hipStream_t st = s->stream;
  // nodes and primitive slots share one allocation (both are 64-byte items: the traversal addresses either as
  // base + index * 64, with slot k at index num_nodes + k)
  static_assert(sizeof(BvhNode) == 64 && sizeof(float4) == 16, "node / slot footprint");
  if (!gpu_built) {
    HIPCHK(s->d_nodes.reserve((size_t)num_nodes + ns));
    if (num_nodes) HIPCHK(hipMemcpyAsync(s->d_nodes.p, bvh.nodes.data(), (size_t)num_nodes * sizeof(BvhNode), hipMemcpyHostToDevice, st));
  }
  if (ns) HIPCHK(hipMemcpyAsync(s->d_nodes.p + num_nodes, slots.data(), (size_t)ns * 64, hipMemcpyHostToDevice, st));
  // The Q tree of the traversal kernels (host-built trees; PBRHIP_WIDE=0 at commit: none): the binary tree collapsed to four
  // children per node with quantised boxes (64 B per node), followed by its own compact triangle slots and by the curve
  // pieces stored as chains of points (16 B per piece instead of a 64-byte slot): dscene.h::QNode.
  std::vector<QNode> wide;
  std::vector<float4> qtri, qpts;
  std::vector<uint32_t> qhit;
  const char* wide_env = getenv("PBRHIP_WIDE");
  static_assert(kMaxLeaf <= 2, "build_qtree expects at most two primitives per leaf of the binary tree");
  if (!gpu_built && num_nodes && !(wide_env && atoi(wide_env) == 0)) {
    // points: the cubics in canonical order; a cubic whose first point equals the last point of its predecessor (same mesh,
    // bit for bit: the segments of a strand) continues that chain, otherwise a new chain starts at the next multiple of 4
    std::vector<uint32_t> piece_point(np, kNone);
    {
      uint32_t prev_inst = kNone, prev_geom = kNone;
      for (uint32_t g = 0; g < np && !PB_CURVE_RECORDS; g++) {  // (curve records: no chains)
        const PrimRef& pr = prims[g];
        if (pr.kind != 1 || pr.sub != 0) continue;
        float wcps[16], pt[5][4];
        world_curve(s->instances[pr.instance_id], *inst_mesh(s, pr.instance_id, pr.geom_id), pr.prim_id, wcps);
        for (int j = 0; j < 5; j++) bezier_point(wcps, (float)j * 0.25f, pt[j]);
        const bool chained = !qpts.empty() && pr.instance_id == prev_inst && pr.geom_id == prev_geom &&
                             memcmp(&qpts.back(), pt[0], 16) == 0;
        if (!chained) {
          while (qpts.size() % 4) qpts.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
          qpts.push_back(make_float4(pt[0][0], pt[0][1], pt[0][2], pt[0][3]));
        }
        const uint32_t start = (uint32_t)qpts.size() - 1u;  // index of this cubic's first point (a multiple of 4)
        for (int j = 1; j < 5; j++) qpts.push_back(make_float4(pt[j][0], pt[j][1], pt[j][2], pt[j][3]));
        for (uint32_t sub = 0; sub < 4; sub++) piece_point[g + sub] = start + sub;  // (the four pieces of a cubic are consecutive gids)
        prev_inst = pr.instance_id, prev_geom = pr.geom_id;
      }
      for (int k = 0; k < 4; k++) qpts.push_back(make_float4(0.f, 0.f, 0.f, 0.f));  // (the last piece reads point p + 1)
    }
    // the complete hit code of every slot (slot | routing bits); curve pieces: per point
    std::vector<uint32_t> slot_code(ns, kNone);
    qhit.assign(qpts.size(), kNone);
    for (uint32_t k = 0; k < ns; k++) {
      const uint32_t g = bvh.slot_gid[k];
      slot_code[k] = k | __builtin_bit_cast(uint32_t, slots[4 * (size_t)k + 2].w);
      if (prims[g].kind != 0 && piece_point[g] != kNone) qhit[piece_point[g]] = slot_code[k];
    }
    // Triangle leaves.  Triangle-only scenes: one TriPair per leaf (dscene.h): its one or two triangles interleaved coordinate by
    // coordinate, 80 bytes, both tested at once on packed fp32.  Scenes with curves (their kernels have no registers to spare
    // and meet triangles rarely): 48 bytes per triangle -- three corners, the hit code in the third word's .w -- one after the other.
    bool tri_pairs = true;
    for (uint8_t kd : kinds) tri_pairs = tri_pairs && kd == 0;
    // the five words of the TriPair of a triangle leaf (slots first .. first + count - 1 of the binary tree)
    auto pair_words = [&](uint32_t first, uint32_t count, float4* o) {
      const float4* a = &slots[4 * (size_t)first];
      const float4* b = count == 2 ? &slots[4 * (size_t)(first + 1)] : a;  // (one triangle: stored twice, the copy is no candidate)
      const float ca = __builtin_bit_cast(float, slot_code[first]), cb = __builtin_bit_cast(float, count == 2 ? slot_code[first + 1] : kNone);
      o[0] = make_float4(a[0].x, b[0].x, a[0].y, b[0].y);
      o[1] = make_float4(a[0].z, b[0].z, a[1].x, b[1].x);
      o[2] = make_float4(a[1].y, b[1].y, a[1].z, b[1].z);
      o[3] = make_float4(a[2].x, b[2].x, a[2].y, b[2].y);
      o[4] = make_float4(a[2].z, b[2].z, ca, cb);
    };
    auto tri_pair = [&](uint32_t first, uint32_t count) -> uint32_t {
      if (!tri_pairs) {
        const uint32_t rec = (uint32_t)(qtri.size() / 3);
        for (uint32_t i = 0; i < count; i++) {
          for (int c = 0; c < 3; c++) qtri.push_back(slots[4 * (size_t)(first + i) + c]);
          qtri.back().w = __builtin_bit_cast(float, slot_code[first + i]);
        }
        return rec;
      }
      const uint32_t rec = (uint32_t)(qtri.size() / kTriPairWords);
      qtri.resize(qtri.size() + kTriPairWords);
      pair_words(first, count, &qtri[(size_t)rec * kTriPairWords]);
      return rec;
    };
    size_t leaves_one = 0, leaves_pair = 0, leaves_split = 0;  // curve leaves of one piece / of two pieces / binary leaves cut in two (PBRHIP_DEBUG)
    const bool curve_records = PB_CURVE_RECORDS != 0;  // (a build option: the traversal kernels are compiled for one leaf format)
    auto map_leaf = [&](uint32_t ref, const float* blo, const float* bhi, QChild* o) -> int {
      const uint32_t first = (ref & 0x3FFFFFFFu) >> 3, count = (ref & 7u) + 1u;
      if (!(ref & kCurveBit)) {
        o[0].ref = kLeafBit | (tri_pair(first, count) << 3) | (count - 1u);
        for (int a = 0; a < 3; a++) o[0].lo[a] = blo[a], o[0].hi[a] = bhi[a];
        return 1;
      }
      if (curve_records) {
        // a record (dscene.h): the end points of the leaf's one or two pieces, 64-byte aligned, tested in one turn
        while (qpts.size() % 4) qpts.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
        const uint32_t P = (uint32_t)qpts.size();
        uint32_t sub[2] = {0u, 0u};
        for (uint32_t i = 0; i < count; i++) {
          const float4* sl = &slots[4 * (size_t)(first + i)];  // (a piece's slot of the binary tree: its two end points, then its index in the cubic)
          qpts.push_back(sl[0]), qpts.push_back(sl[1]);
          sub[i] = __builtin_bit_cast(uint32_t, sl[2].x) & 3u;
        }
        qhit.resize(qpts.size(), kNone);
        qhit[P] = slot_code[first];
        if (count == 2) qhit[P + 2] = slot_code[first + 1];
        o[0].ref = kLeafBit | kCurveBit | ((P | sub[0]) << 3) | (count == 2 ? (kCurvePairBit | sub[1]) : 0u);
        for (int a = 0; a < 3; a++) o[0].lo[a] = blo[a], o[0].hi[a] = bhi[a];
        (count == 2 ? leaves_pair : leaves_one)++;
        return 1;
      }
      uint32_t p0 = piece_point[bvh.slot_gid[first]];
      if (count == 2) {
        uint32_t p1 = piece_point[bvh.slot_gid[first + 1]];
        if ((p0 > p1 ? p0 - p1 : p1 - p0) != 1u) {  // not neighbours in a chain: two leaves, each with its own (widened) box
          for (uint32_t i = 0; i < 2; i++) {
            const uint32_t g = bvh.slot_gid[first + i];
            o[i].ref = kLeafBit | kCurveBit | (piece_point[g] << 3);
            for (int a = 0; a < 3; a++) o[i].lo[a] = BvhNode::widen_lo(lo[3 * (size_t)g + a]), o[i].hi[a] = BvhNode::widen_hi(hi[3 * (size_t)g + a]);
          }
          leaves_split++;
          return 2;
        }
        p0 = std::min(p0, p1);
      }
      o[0].ref = kLeafBit | kCurveBit | (p0 << 3) | (count - 1u);
      for (int a = 0; a < 3; a++) o[0].lo[a] = blo[a], o[0].hi[a] = bhi[a];
      (count == 2 ? leaves_pair : leaves_one)++;
      return 1;
    };
    if (build_qtree(bvh.nodes, map_leaf, &wide) > (uint32_t)kStackDepth || qpts.size() >= (1u << 27)) wide.clear();
    while (qtri.size() % 4) qtri.push_back(make_float4(0.f, 0.f, 0.f, 0.f));  // (q_pt0 a multiple of 4: the low bits of a curve record's address are free)
    for (int k = 0; k < 4; k++) qpts.push_back(make_float4(0.f, 0.f, 0.f, 0.f));  // (the load site reads four words of a leaf)
    qhit.resize(qpts.size(), kNone);
    if (getenv("PBRHIP_DEBUG")) fprintf(stderr, "pbrhip: commit: curve leaves of the Q tree (counted over the collapse's visits): %zu of one piece, %zu of two pieces, %zu binary leaves cut in two\n", leaves_one, leaves_pair, leaves_split);
  }
  // Where the random walks' rays start (dscene.h::SssEntry): per instance, the cut of the Q tree around its bounds
  std::vector<SssEntry> sss_entries;
  if (!wide.empty() && env_u32("PBRHIP_SSS_ENTRY", 1u) != 0u) {  // (whatever the materials are now: pbrhip_scene_update_* can switch subsurface on later)
    const size_t ninst = s->instances.size();
    // (every foreign reference costs each walk ray a slab test: a deeper entry is only worth so many)
    const uint32_t max_foreign = std::min(env_u32("PBRHIP_SSS_FOREIGN", 3u), kSssMaxForeign);
    sss_entries.assign(ninst, SssEntry{});
    std::vector<float> ilo(3 * ninst, INFINITY), ihi(3 * ninst, -INFINITY);
    for (uint32_t g = 0; g < np; g++)
      for (int a = 0; a < 3; a++) {
        float& l = ilo[3 * (size_t)prims[g].instance_id + a];
        float& h = ihi[3 * (size_t)prims[g].instance_id + a];
        l = std::min(l, lo[3 * (size_t)g + a]), h = std::max(h, hi[3 * (size_t)g + a]);
      }
    struct CutRef {
      uint32_t ref;
      float lo[3], hi[3];
    };
    for (size_t i = 0; i < ninst; i++) {
      SssEntry& E = sss_entries[i];
      if (!(ilo[3 * i] <= ihi[3 * i])) continue;  // (no primitive)
      float ext = 0.f;
      for (int a = 0; a < 3; a++) ext = std::max(ext, ihi[3 * i + a] - ilo[3 * i + a]);
      if (!(ext > 0.f) || !std::isfinite(ext)) continue;
      const float m1 = 1e-4f * ext, m2 = 2e-3f * ext;  // the region rays may stay in / how far beyond it a primitive's box can matter
      float rlo[3], rhi[3], wlo[3], whi[3];
      for (int a = 0; a < 3; a++) rlo[a] = ilo[3 * i + a] - m1, rhi[a] = ihi[3 * i + a] + m1, wlo[a] = rlo[a] - m2, whi[a] = rhi[a] + m2;
      std::vector<CutRef> cut{{0u, {-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}}};
      for (bool changed = true; changed;) {
        changed = false;
        for (size_t c = 0; c < cut.size() && !changed; c++) {
          if (cut[c].ref & kLeafBit) continue;
          const QNode& nd = wide[cut[c].ref];
          const float org[3] = {nd.org[0], nd.org[1], nd.org[2]}, st3[3] = {nd.sx, nd.sy, nd.sz};
          const uint32_t ql[3] = {nd.qlo_x, nd.qlo_y, nd.qlo_z}, qh[3] = {nd.qhi_x, nd.qhi_y, nd.qhi_z};
          std::vector<CutRef> kids;
          int nchild = 0;
          for (int k = 0; k < 4; k++) {
            if (nd.c[k] == kEmptyChild) continue;
            nchild++;
            CutRef r;
            r.ref = nd.c[k];
            bool meets = true;
            for (int a = 0; a < 3; a++) {  // the box the traversal rebuilds for this child (dtrace.h::box_test4q): fma(q, s, org)
              r.lo[a] = fmaf((float)((ql[a] >> (8 * k)) & 255u), st3[a], org[a]);
              r.hi[a] = fmaf((float)((qh[a] >> (8 * k)) & 255u), st3[a], org[a]);
              meets = meets && r.lo[a] <= whi[a] && r.hi[a] >= wlo[a];
            }
            if (meets) kids.push_back(r);
          }
          // descend where that drops a child (or leads to an only child), while the cut stays small
          if (((int)kids.size() < nchild || kids.size() == 1) && cut.size() - 1 + kids.size() <= 1u + max_foreign) {
            cut.erase(cut.begin() + (ptrdiff_t)c);
            cut.insert(cut.end(), kids.begin(), kids.end());
            changed = true;
          }
        }
      }
      // the entry: the inner node of the cut that shares the most volume with the instance's bounds
      int best = -1;
      double best_v = -1.0;
      for (size_t c = 0; c < cut.size(); c++) {
        if (cut[c].ref & kLeafBit) continue;
        double v = 1.0;
        for (int a = 0; a < 3; a++) v *= std::max(0.0, (double)std::min(cut[c].hi[a], ihi[3 * i + a]) - (double)std::max(cut[c].lo[a], ilo[3 * i + a]));
        if (v > best_v) best_v = v, best = (int)c;
      }
      if (best < 0 || cut[(size_t)best].ref == 0u) continue;  // (the root, or leaves only: start at the root)
      E.entry = cut[(size_t)best].ref;
      for (int a = 0; a < 3; a++) E.lo[a] = rlo[a], E.hi[a] = rhi[a];
      for (size_t c = 0; c < cut.size(); c++) {
        if ((int)c == best) continue;
        auto& f = E.foreign[E.nforeign++];
        f.ref = cut[c].ref;
        for (int a = 0; a < 3; a++) f.lo[a] = cut[c].lo[a], f.hi[a] = cut[c].hi[a];
      }
      if (getenv("PBRHIP_DEBUG")) fprintf(stderr, "pbrhip: commit: instance %zu: random walks start at Q node %u with %u foreign references\n", i, E.entry, E.nforeign);
    }
  }
  if (sss_entries.empty()) s->d_sss_entries.release();
  else HIPCHK(s->d_sss_entries.upload(sss_entries, st));
  if (wide.empty()) s->d_wide.release(), s->d_qhit.release();
  if (getenv("PBRHIP_DEBUG")) fprintf(stderr, "pbrhip: commit: %u binary nodes, %zu wide nodes, %zu slots, %zu triangle leaves + %zu points in the Q tree\n", num_nodes, wide.size(), (size_t)ns, qtri.size() / kTriPairWords, qpts.size());
  if (!wide.empty()) {
    HIPCHK(s->d_wide.reserve(wide.size() * 4 + qtri.size() + qpts.size()));
    HIPCHK(hipMemcpyAsync(s->d_wide.p, wide.data(), wide.size() * sizeof(QNode), hipMemcpyHostToDevice, st));
    if (!qtri.empty()) HIPCHK(hipMemcpyAsync(s->d_wide.p + wide.size() * 4, qtri.data(), qtri.size() * 16, hipMemcpyHostToDevice, st));
    if (!qpts.empty()) HIPCHK(hipMemcpyAsync(s->d_wide.p + wide.size() * 4 + qtri.size(), qpts.data(), qpts.size() * 16, hipMemcpyHostToDevice, st));
    HIPCHK(s->d_qhit.upload(qhit, st));
  }
  HIPCHK(s->d_shade.upload(shade, st));
  HIPCHK(s->d_materials.upload(mats, st));
  HIPCHK(s->d_light_cdf.upload(s->light_cdf, st));
  HIPCHK(s->d_heads.upload(heads, st));
  HIPCHK(s->d_lprim_cdf.upload(lprim_cdf, st));
  HIPCHK(s->d_lrecs.upload(lrecs, st));
  // one box per light over all primitives of its mesh, packed two per node (an odd last one is stored twice)
  std::vector<BvhNode> light_boxes((heads.size() + 1) / 2);
  for (size_t l = 0; l < heads.size(); l++) {
    float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t f = heads[l].first; f < heads[l].first + heads[l].count; f++)
      for (const float* p : {lrecs[f].p0, lrecs[f].p1, lrecs[f].p2})
        for (int a = 0; a < 3; a++) lo3[a] = std::min(lo3[a], p[a]), hi3[a] = std::max(hi3[a], p[a]);
    BvhNode& nd = light_boxes[l / 2];
    if (l % 2 == 0) memset(&nd, 0, sizeof(nd)), nd.set_box(1, lo3, hi3);
    nd.set_box(int(l % 2), lo3, hi3);
  }
  HIPCHK(s->d_light_boxes.upload(light_boxes, st));
  HIPCHK(s->d_tex_pixels.upload(s->tex_pixels, st));
  HIPCHK(s->d_tex_descs.upload(s->tex_descs, st));
  HIPCHK(hipStreamSynchronize(st));
  DScene& d = s->dscene;
  d.nodes = s->d_nodes.p, d.slots = reinterpret_cast<const float4*>(s->d_nodes.p + num_nodes), d.shade = s->d_shade.p;
  d.materials = s->d_materials.p, d.light_cdf = s->d_light_cdf.p;
  d.light_heads = s->d_heads.p, d.lprim_cdf = s->d_lprim_cdf.p, d.lrecs = s->d_lrecs.p, d.light_boxes = s->d_light_boxes.p;
  d.num_nodes = num_nodes, d.num_slots = ns, d.num_lights = (uint32_t)s->lights.size(), d.num_lrecs = (uint32_t)lrecs.size();
  d.num_materials = (uint32_t)mats.size();
  d.tex_pixels = s->d_tex_pixels.p, d.textures = s->d_tex_descs.p, d.num_textures = (uint32_t)s->tex_descs.size();
  d.num_curves = 0;
  for (uint8_t kd : kinds) d.num_curves += kd ? 1u : 0u;
  d.wide = wide.empty() ? nullptr : s->d_wide.p, d.wide_nodes = (uint32_t)wide.size();
  d.q_tri0 = (uint32_t)wide.size() * 4u, d.q_pt0 = d.q_tri0 + (uint32_t)qtri.size(), d.q_hitcode = wide.empty() ? nullptr : s->d_qhit.p;
  d.top_nodes = gpu_built ? 0u : std::min<uint32_t>(num_nodes, (uint32_t)kTopNodes);
  d.wide_top_nodes = std::min<uint32_t>((uint32_t)wide.size(), (uint32_t)kTopNodes);
  // light sampling works on the meshes' local positions (light-manager.h:128-136 "TODO transform"), the raytracer on the
  // transformed ones: the doomed-path pretest against the light primitives (kernels.hip::misses_all_lights) is only the
  // traversal's own test when the two coincide
  d.sss_entries = sss_entries.empty() ? nullptr : s->d_sss_entries.p, d.num_sss_entries = (uint32_t)sss_entries.size();
  d.lights_transformed = 0;
  for (const HostLight& L : s->lights) d.lights_transformed |= s->instances[L.instance_id].identity ? 0u : 1u;
  s->committed = true;
  return PBRHIP_OK;
  });
}

extern "C" int pbrhip_scene_set_bvh_builder(pbrhip_scene* s, int builder) {
  return guarded([&]() -> int {
  if (!s) return fail(PBRHIP_EINVAL, "scene is NULL");
  if (builder != PBRHIP_BVH_HOST_SAH && builder != PBRHIP_BVH_GPU_LBVH) return fail(PBRHIP_EINVAL, "unknown BVH builder %d", builder);
  if (s->committed) return fail(PBRHIP_ESTATE, "scene already committed");
  s->bvh_builder = builder;
  return PBRHIP_OK;
  });
}

extern "C" int pbrhip_scene_aabb(const pbrhip_scene* s, float bmin[3], float bmax[3]) {
  return guarded([&]() -> int {
  if (!s || !bmin || !bmax) return fail(PBRHIP_EINVAL, "scene_aabb: NULL argument");
  if (!s->committed) return fail(PBRHIP_ESTATE, "scene not committed");
  memcpy(bmin, s->bmin, 12), memcpy(bmax, s->bmax, 12);
  return PBRHIP_OK;
  });
}
extern "C" int pbrhip_scene_info(const pbrhip_scene* s, uint64_t* num_nodes, uint64_t* num_slots, uint32_t* depth,
                                 uint64_t* device_bytes) {
  return guarded([&]() -> int {
  if (!s) return fail(PBRHIP_EINVAL, "scene is NULL");
  if (num_nodes) *num_nodes = s->dscene.num_nodes;
  if (num_slots) *num_slots = s->dscene.num_slots;
  if (depth) *depth = s->bvh_depth;
  if (device_bytes) *device_bytes = s->device_bytes();
  return PBRHIP_OK;
  });
}

static int update_material(pbrhip_scene* s, uint32_t id, const HostMaterial& hm) {
  // validate everything first: a rejected call leaves the host material, has_sss and the device copy as they were
  if (id >= s->materials.size()) return fail(PBRHIP_EINVAL, "material id %u out of range", id);
  if (s->materials[id].kind != hm.kind) return fail(PBRHIP_EINVAL, "material %u is of the other kind", id);
  if (s->committed && hm.kind == kMatPrincipled)
    for (uint32_t t : {hm.pr.base_color_tex_id, hm.pr.subsurface_color_tex_id})
      if (t != kNone && t >= s->tex_descs.size()) return fail(PBRHIP_EINVAL, "texture id %u out of range", t);
  if (s->committed) {
    HIPCHK(hipSetDevice(s->device));
    const Material m = make_material(hm);
    HIPCHK(hipMemcpyAsync(s->d_materials.p + id, &m, sizeof(m), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->has_sss = s->has_sss || (m.kind == kMatPrincipled && (m.bsdf.enable_subsurface || m.textured));
    s->has_textured = s->has_textured || (m.kind == kMatPrincipled && m.textured);
  }
  s->materials[id] = hm;
  return PBRHIP_OK;
}
extern "C" int pbrhip_scene_update_principled_material(pbrhip_scene* s, uint32_t id, const pbrhip_principled_param* p) {
  return guarded([&]() -> int {
  if (!s || !p) return fail(PBRHIP_EINVAL, "update_material: NULL argument");
  if (int rc = check_tex(p)) return rc;
  HostMaterial m;
  m.kind = kMatPrincipled;
  memcpy(&m.pr, p, sizeof(m.pr));
  memset(&m.hr, 0, sizeof(m.hr));
  return update_material(s, id, m);
  });
}
extern "C" int pbrhip_scene_update_hair_material(pbrhip_scene* s, uint32_t id, const pbrhip_hair_param* p) {
  return guarded([&]() -> int {
  if (!s || !p) return fail(PBRHIP_EINVAL, "update_material: NULL argument");
  HostMaterial m;
  m.kind = kMatHair;
  memset(&m.pr, 0, sizeof(m.pr));
  memcpy(&m.hr, p, sizeof(m.hr));
  return update_material(s, id, m);
  });
}

// ------------------------------------------------------------------ tiles (render-tile.cc:29-41)
extern "C" int pbrhip_create_tiles(uint32_t width, uint32_t height, uint32_t* out, uint32_t* num_tiles) {
  return guarded([&]() -> int {
  if (!num_tiles) return fail(PBRHIP_EINVAL, "num_tiles is NULL");
  const uint32_t kTile = 64;
  uint32_t n = 0;
  for (uint32_t i = 0; i < height; i += kTile)
    for (uint32_t j = 0; j < width; j += kTile) {
      if (out) {
        out[4 * n + 0] = j, out[4 * n + 1] = std::min(j + kTile, width);
        out[4 * n + 2] = i, out[4 * n + 3] = std::min(i + kTile, height);
      }
      n++;
    }
  *num_tiles = n;
  return PBRHIP_OK;
  });
}

// camera of RenderingTile (render.cc:132-158)
static Camera make_camera(const pbrhip_scene* s, uint32_t width, uint32_t height) {
  const float *bmin = s->bmin, *bmax = s->bmax;
  float hs, vs;
  if (bmax[0] - bmin[0] > bmax[1] - bmin[1]) {
    hs = bmax[0] - bmin[0];
    vs = hs * float(height) / float(width);
  } else {
    vs = bmax[1] - bmin[1];
    hs = vs * float(width) / float(height);
  }
  Camera c;
  c.org[0] = (bmax[0] + bmin[0]) * 0.5f;
  c.org[1] = (bmax[1] + bmin[1]) * 0.5f;
  c.org[2] = bmax[2] + hs * 0.5f * sqrtf(3.f);
  c.x_corner = (bmax[0] + bmin[0]) * 0.5f - hs * 0.5f;
  c.y_corner = (bmax[1] + bmin[1]) * 0.5f + vs * 0.5f;
  c.z_corner = bmax[2];
  c.dx = hs / float(width);
  c.dy = vs / float(height);
  return c;
}

// ------------------------------------------------------------------ render
static constexpr uint64_t kBytesPerPath = 64 + 32 + 2 * 16 + 5 * 16 + 16 + 7 * 4;  // ensure_paths(): rec, srec, L + hit, sss, sh_e, queues
namespace {
struct Timer {
  pbrhip_scene* s;
  bool on;
  hipStream_t stream;
  std::vector<hipEvent_t> events;  // owned by the scene's pool once collected
  size_t used = 0;
  struct Rec {
    size_t ev;
    double* acc;
  };
  std::vector<Rec> recs;
  // first / last event of every enqueued iteration of the group: the gaps between them are the time the stream sat empty
  std::vector<std::pair<size_t, size_t>> iters;
  double* idle_acc = nullptr;
  const double* only = nullptr;  // PBRHIP_RENDER_TIMING_TRACE: only the launches that report into this accumulator are timed
  bool skipped = false;
  void iteration_begins() {
    if (on && !only) iters.push_back({used, used});
  }
  void iteration_ends() {
    if (on && !only && !iters.empty() && used >= 1) iters.back().second = used - 1;
  }
  hipError_t begin(double* acc) {
    if (!on) return hipSuccess;
    skipped = only && acc != only;
    if (skipped) return hipSuccess;
    while (events.size() < used + 2) {
      hipEvent_t e;
      if (!s->events.empty()) {
        e = s->events.back();
        s->events.pop_back();
      } else {
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) return rc;
      }
      events.push_back(e);
    }
    recs.push_back({used, acc});
    return hipEventRecord(events[used], stream);
  }
  hipError_t end() {
    if (!on || skipped) return hipSuccess;
    hipError_t rc = hipEventRecord(events[used + 1], stream);
    used += 2;
    return rc;
  }
  // call after a stream sync
  hipError_t collect() {
    if (!on) return hipSuccess;
    for (const Rec& r : recs) {
      float ms = 0.f;
      hipError_t rc = hipEventElapsedTime(&ms, events[r.ev], events[r.ev + 1]);
      if (rc != hipSuccess) return rc;
      *r.acc += (double)ms;
    }
    recs.clear();
    for (size_t i = 1; i < iters.size() && idle_acc; i++) {
      float ms = 0.f;
      if (iters[i - 1].second >= iters[i].first || iters[i].first >= used) continue;
      hipError_t rc = hipEventElapsedTime(&ms, events[iters[i - 1].second], events[iters[i].first]);
      if (rc != hipSuccess) return rc;
      *idle_acc += (double)ms;
    }
    iters.clear();
    used = 0;
    for (hipEvent_t e : events) s->events.push_back(e);  // back to the scene's pool
    events.clear();
    return hipSuccess;
  }
};
}  // namespace


void pb::shard_pixels(uint32_t w, uint32_t h, uint32_t rank, uint32_t world, uint32_t block, std::vector<uint32_t>* out) {
  if (block == 0) block = 64;  // CreateTiles' tile (pbrhip_create_tiles enumerates the same blocks in the same order)
  out->clear();
  uint32_t t = 0;
  for (uint32_t by = 0; by < h; by += block)
    for (uint32_t bx = 0; bx < w; bx += block, t++) {
      if (t % world != rank) continue;  // interleaved block -> GPU map
      for (uint32_t y = by; y < std::min(by + block, h); y++)
        for (uint32_t x = bx; x < std::min(bx + block, w); x++) out->push_back(y * w + x);
    }
}

int pb::ensure_pixels(pbrhip_scene* s, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, uint32_t block) {
  if (block == 0) block = 64;
  uint32_t pt = 8u;
  if (const char* e = getenv("PBRHIP_PIXEL_TILE")) pt = (uint32_t)strtoul(e, nullptr, 10);
  const bool shuffle = env_u32("PBRHIP_PATCH_SHUFFLE", 1u) != 0u;
  pt |= shuffle ? 0x80000000u : 0u;  // (part of the cache key below)
  if (s->pk_w == w && s->pk_h == h && s->pk_rank == rank && s->pk_world == world && s->pk_block == block && s->pk_tile == pt && s->pix_index.p) return PBRHIP_OK;
  std::vector<uint32_t> pix;
  shard_pixels(w, h, rank, world, block, &pix);
  // The order the paths of a pass are laid out in (path j of a pass = pixel pix[j]): the rank's blocks in shard_pixels' order, and
  // inside a block sub-blocks of PBRHIP_PIXEL_TILE x PBRHIP_PIXEL_TILE pixels (default 8) instead of rows -- a wave's 64 camera
  // rays are an 8 x 8 patch, not a 64 x 1 strip: they share more of the tree, and so do their later bounces.  A permutation of
  // the list: every value is a function of (pixel, pass) alone, images do not change.
  HIPCHK(s->pix_index.upload(pix, s->stream));  // (the shard's own order: what the exchange packs and unpacks by, multi.cpp)
  const uint32_t pt_key = pt;
  pt &= 0x7FFFFFFFu;
  if (pt > 1u && pt < block) {
    std::vector<uint32_t> ordered;
    ordered.reserve(pix.size());
    std::vector<std::pair<uint32_t, uint32_t>> patches;  // (first entry, entries) of every patch in `ordered`
    uint32_t t = 0;
    for (uint32_t by = 0; by < h; by += block)
      for (uint32_t bx = 0; bx < w; bx += block, t++) {
        if (t % world != rank) continue;
        const uint32_t ey = std::min(by + block, h), ex = std::min(bx + block, w);
        for (uint32_t sy = by; sy < ey; sy += pt)
          for (uint32_t sx = bx; sx < ex; sx += pt) {
            const uint32_t at = (uint32_t)ordered.size();
            for (uint32_t y = sy; y < std::min(sy + pt, ey); y++)
              for (uint32_t x = sx; x < std::min(sx + pt, ex); x++) ordered.push_back(y * w + x);
            patches.push_back({at, (uint32_t)ordered.size() - at});
          }
      }
    // Round 6: the patches in a SCATTERED order (patch i of the list = patch i x K mod M of the image order, K ~ 0.38 M, coprime to
    // M).  k_trace's waves take rays from the queue in batches of up to 512 = eight patches, a wave takes only five or six batches
    // per launch, and in image order a batch is ONE image region: the batches over dense geometry cost several times the batches over
    // the walls, the waves that drew them found the queue empty up to 0.46 ms after the first wave had (per-wave timeline:
    // profiles/README.md), and every launch ended with the chip half empty for that long.  Scattered, a batch is eight regions and
    // consecutive batches are unrelated: the batches cost about the same.  A permutation: the image does not depend on it.
    if (shuffle && patches.size() > 2) {
      const uint64_t M = patches.size();
      uint64_t K = (uint64_t)((double)M * 0.381966) | 1ull;
      auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t r = a % b; a = b, b = r; } return a; };
      while (gcd(K, M) != 1) K += 2;
      std::vector<uint32_t> scattered;
      scattered.reserve(ordered.size());
      for (uint64_t i = 0; i < M; i++) {
        const auto& pch = patches[(size_t)((i * K) % M)];
        scattered.insert(scattered.end(), ordered.begin() + pch.first, ordered.begin() + pch.first + pch.second);
      }
      ordered.swap(scattered);
    }
    pix.swap(ordered);
  }
  HIPCHK(s->path_pix.upload(pix, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  s->pk_w = w, s->pk_h = h, s->pk_rank = rank, s->pk_world = world, s->pk_block = block, s->pk_tile = pt_key, s->pk_npix = (uint32_t)pix.size();
  return PBRHIP_OK;
}

static int ensure_groups(pbrhip_scene* s, uint32_t groups) {
  while (s->group_streams.size() + 1 < groups) {
    hipStream_t g;
    HIPCHK(hipStreamCreateWithFlags(&g, hipStreamNonBlocking));
    s->group_streams.push_back(g);
  }
  HIPCHK(s->counts.reserve(kCntNum * kMaxGroups));
  HIPCHK(s->spill.reserve((size_t)groups * kSpillWords));
  return PBRHIP_OK;
}

static int ensure_paths(pbrhip_scene* s, size_t n) {
  HIPCHK(s->rec.reserve(4 * n));   // ray_o | ray_d | thr | rng (kernels.h::PathState)
  HIPCHK(s->srec.reserve(2 * n));  // sh_d | sh_c
  HIPCHK(s->L.reserve(n));
  HIPCHK(s->hit.reserve(n));
  HIPCHK(s->ssrec.reserve(4 * n));  // sss_sigt | sss_sigs | sss_thr | sss_ez
  HIPCHK(s->sss_A.reserve(n));
  HIPCHK(s->sh_e.reserve(n));
  for (auto& b : s->q) HIPCHK(b.reserve(n));
  HIPCHK(s->counts.reserve(kCntNum * kMaxGroups));
  HIPCHK(s->stats.reserve(kStatNum));
  return PBRHIP_OK;
}

// PathState::pass_run of a group of `npass` passes: 1 for scenes of surfaces; scenes with curves: the largest power of two <= 64 that
// divides npass.  PBRHIP_PASS_RUN=R forces a run length (when it divides npass; 1 = off): A/B and tests.
static uint32_t pass_run_for(uint32_t npass, bool curves) {
  uint32_t want = curves ? 64u : 1u;
  if (const char* e = getenv("PBRHIP_PASS_RUN")) want = std::max(1u, (uint32_t)strtoul(e, nullptr, 10));
  uint32_t r = 1u;
  while (r * 2u <= want && npass % (r * 2u) == 0u) r *= 2u;
  return r;
}

// How the passes of a chunk are split into path groups (each group = its own queues, counters and HIP stream; path
// slots stay global and passes are accumulated in ascending order, so the image does not depend on the split: GPU test).
// The scheduler in render_impl starts groups in order while fewer than `window` of them are in their bulk phase (default:
// all at once).  PBRHIP_GROUPS="56,8" (passes per group, started one after the other: PBRHIP_WINDOW defaults to 1 then)
// and pbrhip_render_desc.num_streams = n (n equal groups at once) override the default plan; the pipelined plans that
// were tried (geometric sizes, big-then-small pairs) all lost to it, see profiles/README.md.
static std::vector<uint32_t> plan_groups(uint32_t np, uint32_t npix, uint32_t want_groups) {
  std::vector<uint32_t> g;
  if (const char* e = getenv("PBRHIP_GROUPS")) {  // explicit passes per group, e.g. "32,16,8,4,2,1,1" (the rest joins the last)
    uint32_t left = np;
    for (const char* p = e; *p && left;) {
      uint32_t v = (uint32_t)strtoul(p, (char**)&p, 10);
      if (*p == ',') p++;
      v = std::max(1u, std::min(v, left));
      g.push_back(v), left -= v;
    }
    if (left) {
      if (g.empty()) g.push_back(left);
      else g.back() += left;
    }
    return g;
  }
  if (want_groups) {  // pbrhip_render_desc.num_streams: that many equal groups
    const uint32_t ng = std::min(std::min(want_groups, (uint32_t)kMaxGroups * 2u), np);
    for (uint32_t k = 0; k < ng; k++) g.push_back((uint32_t)((uint64_t)np * (k + 1) / ng) - (uint32_t)((uint64_t)np * k / ng));
    return g;
  }
  // default (A/B on C2, scripts/sched_ab.py, profiles/README.md): two equal groups, both started at once, when the chunk
  // holds at least 96 Mi paths (the whole 132.7 M-path frame: 60.4 -> 58.9 ms; one group's launch-bound drains and its
  // k_tail overlap the other's bulk work); one group below that (a half / quarter / eighth of the frame: 32.6 / 19.7 /
  // 12.1 ms with one group against 32.2 / 19.9 / 12.8 with two -- every extra group adds its own latency-bound launches)
  // Round 4, after the shading kernels got faster: two groups also pay for a half and a quarter of the frame (27.1-27.2 / 16.0-16.4 ms
  // against 27.9-28.2 / 16.6-16.7 with one), not for an eighth (10.0-10.2 either way): the threshold was 24 Mi paths.
  // Round 5 (kernels 5-8 % faster, the launches' drains the same): an eighth of the frame (15.8 Mi paths) 9.50-9.57 ms with one group,
  // 9.13-9.15 with two, 9.19-9.33 with three: the threshold is 12 Mi paths.
  if ((uint64_t)np * npix >= (12ull << 20) && np >= 2) g.push_back(np / 2), g.push_back(np - np / 2);
  else g.push_back(np);
  return g;
}

int pb::render_impl(pbrhip_scene* s, const pbrhip_render_desc* d, const volatile unsigned char* cancel, float* d_rgba,
                    uint32_t* d_count, size_t* finish_pass, pbrhip_render_stats* stats) {
  auto t_begin = std::chrono::steady_clock::now();
  if (!s->committed) return fail(PBRHIP_ESTATE, "scene not committed");
  if (d->width == 0 || d->height == 0) return fail(PBRHIP_EINVAL, "empty image");
  if ((uint64_t)d->width * d->height >= (1ull << 32)) return fail(PBRHIP_EINVAL, "image too large");
  uint32_t world = d->tile_world ? d->tile_world : 1;
  if (d->tile_rank >= world) return fail(PBRHIP_EINVAL, "tile_rank %u >= tile_world %u", d->tile_rank, world);
  if (d->shard_block > 4096) return fail(PBRHIP_EINVAL, "shard_block %u is not a sensible block edge", d->shard_block);
  auto cancelled = [&]() { return cancel && __atomic_load_n(cancel, __ATOMIC_RELAXED) != 0; };
  auto publish = [&](size_t passes) {
    if (finish_pass) __atomic_store_n(finish_pass, passes, __ATOMIC_RELEASE);
  };
  HIPCHK(hipSetDevice(s->device));
  hipStream_t st = s->stream;
  const size_t npx_img = (size_t)d->width * d->height;
  if (!(d->flags & PBRHIP_RENDER_NO_CLEAR)) {  // PrepareRendering: layer->Resize + Clear (render.cc:99-100)
    HIPCHK(hipMemsetAsync(d_rgba, 0, npx_img * 4 * sizeof(float), st));
    HIPCHK(hipMemsetAsync(d_count, 0, npx_img * sizeof(uint32_t), st));
  }
  publish(0);
  pbrhip_render_stats S;
  memset(&S, 0, sizeof(S));
  if (int rc = ensure_pixels(s, d->width, d->height, d->tile_rank, world, d->shard_block)) return rc;
  const uint32_t npix = s->pk_npix;
  const bool want_stats = (d->flags & PBRHIP_RENDER_STATS) != 0;
  const bool trace_timing_only = (d->flags & PBRHIP_RENDER_TIMING) == 0 && (d->flags & PBRHIP_RENDER_TIMING_TRACE) != 0;
  const bool want_timing = (d->flags & (PBRHIP_RENDER_TIMING | PBRHIP_RENDER_TIMING_TRACE)) != 0;
  uint32_t done = 0;  // passes accumulated into the layer
  if (npix > 0 && d->num_sample > 0) {
    // default: as many paths in flight as 60 % of the free HBM holds (288 GB: a whole 1080p x 64 spp frame,
    // 132.7 M paths x 244 B, is one chunk) -- fewer, larger launches and one tail instead of many
    uint64_t max_paths = d->max_paths_in_flight;
    if (!max_paths) {
      size_t free_b = 0, total_b = 0;
      HIPCHK(hipMemGetInfo(&free_b, &total_b));
      // what this scene already holds for path state counts as available
      size_t have = (s->rec.n + s->srec.n + s->ssrec.n + s->L.n + s->hit.n + s->sh_e.n + s->sss_A.n) * 16;
      for (auto& b : s->q) have += b.n * 4;
      max_paths = std::min<uint64_t>(kMaxPathsInFlight,
                                     std::max<uint64_t>(1ull << 20, (uint64_t)((free_b + have) * 0.6) / kBytesPerPath));
    }
    if (max_paths > kMaxPathsInFlight) max_paths = kMaxPathsInFlight;
    if (getenv("PBRHIP_DEBUG")) {
      size_t fb = 0, tb = 0;
      (void)hipMemGetInfo(&fb, &tb);
      fprintf(stderr, "pbrhip: free %.1f GB total %.1f GB max_paths %llu npix %u\n", fb / 1e9, tb / 1e9, (unsigned long long)max_paths, npix);
    }
    if (npix > kMaxPathsInFlight) return fail(PBRHIP_EUNSUPPORTED, "more than 2^28 pixels per rank");
    uint32_t chunk_passes = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(d->num_sample, max_paths / npix));
    if ((uint64_t)chunk_passes * npix >= (1ull << 32)) chunk_passes = (uint32_t)(((1ull << 32) - 1) / npix);
    // A working set that is nearly large enough is used as it is: growing it means freeing and re-allocating every path-state
    // array (65 GB at the largest chunk: 1.3 s), and the chunk size does not change the image.  (An eighth of the C5 frame asks
    // for 258 passes = 267.5 M paths where the whole frame had allocated 265.4 M.)
    if (s->hit.n < (size_t)chunk_passes * npix && s->hit.n / npix >= 1 && (double)(s->hit.n / npix) >= 0.75 * chunk_passes)
      chunk_passes = (uint32_t)(s->hit.n / npix);
    if (int rc = ensure_paths(s, (size_t)chunk_passes * npix)) return rc;
    PathState P;
    P.pass_run = 1u;
    P.ray_o.base = s->rec.p, P.ray_d.base = s->rec.p + 1, P.thr.base = s->rec.p + 2, P.L = s->L.p, P.hit = s->hit.p;
    P.rng.base = reinterpret_cast<uint64_t*>(s->rec.p + 3);
    P.hold.base = reinterpret_cast<uint32_t*>(s->rec.p + 3) + 2;
    P.rng4.base = reinterpret_cast<uint4*>(s->rec.p + 3);
    P.sss_sigt.base = s->ssrec.p, P.sss_sigs.base = s->ssrec.p + 1, P.sss_thr.base = s->ssrec.p + 2;
    P.sss_ez.base = s->ssrec.p + 3, P.sss_A = s->sss_A.p;
    P.q_in = s->q[0].p, P.q_out = s->q[1].p, P.q_principled = s->q[2].p, P.q_hair = s->q[3].p, P.q_sss = s->q[4].p, P.q_shadow = s->q[5].p, P.q_shadow_in = s->q[6].p;
    P.sh_d.base = s->srec.p, P.sh_c.base = s->srec.p + 1, P.sh_e = s->sh_e.p;
    P.counts = s->counts.p, P.stats = want_stats ? s->stats.p : nullptr, P.spill = s->spill.p;
    P.first = 0u, P.direct = 0u, P.cam_org[0] = P.cam_org[1] = P.cam_org[2] = 0.f;
    P.heads = nullptr;
    P.susp_turns = 0u, P.susp_out = nullptr, P.susp_in = nullptr, P.shadow_first = 0u;
    P.no_medium = s->has_sss ? 0u : 1u;
    P.wave_log = nullptr, P.wave_log_launch = 0;
    // debugging aid: PBRHIP_WAVE_LOG=<file> with PBRHIP_RENDER_STATS dumps start / end / turns of every wave of every
    // k_trace launch (scripts/wave_log.py reads it)
    DevBuf<unsigned long long> wave_log;
    const char* wave_log_path = getenv("PBRHIP_WAVE_LOG");
    uint32_t wave_log_launches = 0;
    if (wave_log_path) {
      HIPCHK(wave_log.reserve((size_t)kWaveLogLaunches * kWaveLogWaves * 4));
      HIPCHK(hipMemsetAsync(wave_log.p, 0, sizeof(unsigned long long) * kWaveLogLaunches * kWaveLogWaves * 4, st));
      P.wave_log = wave_log.p;
    }
    HIPCHK(hipMemsetAsync(s->stats.p, 0, sizeof(unsigned long long) * kStatNum, st));
    const Camera cam = make_camera(s, d->width, d->height);
    const uint64_t rng_inc = (d->seed_seq << 1u) | 1u;  // pcg32_srandom (rng.h:30-36)
    const DScene& sc = s->dscene;

    uint32_t tail_paths = d->tail_paths == 0xFFFFFFFFu ? 0u : (d->tail_paths ? d->tail_paths : 262144u);
    if (const char* e = getenv("PBRHIP_TAIL_PATHS")) tail_paths = (uint32_t)strtoul(e, nullptr, 10);  // 0 = never
    uint32_t want_groups = d->num_streams;
    if (const char* e = getenv("PBRHIP_STREAMS")) want_groups = (uint32_t)atoi(e);
    // at most `window` groups in their bulk phase at a time; a group is in its bulk phase until it has handed its
    // remaining paths to k_tail (or, with PBRHIP_BULK_DIV = k, until fewer than 1/k of its paths are alive)
    const uint32_t window = std::max(1u, env_u32("PBRHIP_WINDOW", getenv("PBRHIP_GROUPS") ? 1u : (uint32_t)kMaxGroups));
    const uint32_t bulk_div = env_u32("PBRHIP_BULK_DIV", 0u);
    const bool trace_sched = getenv("PBRHIP_TRACE_SCHED") != nullptr;
    const bool sss_walk = env_u32("PBRHIP_SSS_WALK", 1u) != 0u;  // 0: one wavefront iteration per step of a random walk (A/B)
    // lanes (stream + counters + 117 MB of traversal spill area each) for the groups this call can have in flight: the first
    // chunk is the largest, so its plan has the most groups
    const uint32_t max_lanes = std::min<uint32_t>((uint32_t)kMaxGroups, (uint32_t)plan_groups(std::min(chunk_passes, d->num_sample), npix, want_groups).size());
    if (int rc = ensure_groups(s, std::max(1u, max_lanes))) return rc;
    // Resumable rays and the pipelined host loop (round 6; kernels.h::PathState::susp_turns, scene_impl.h::h_ring).
    // susp_turns: loop turns a k_trace wave keeps draining after the queue ran dry before it suspends its closest-hit rays (0: never);
    // pipe_depth: iterations of a group enqueued ahead of what the host has heard of (1: the old round trip per iteration).  An
    // iteration enqueued ahead sizes its launches by the last count the host saw (live paths only ever decrease: an upper bound).
    const uint32_t susp_turns = env_u32("PBRHIP_SUSP_TURNS", 24u);
    const bool direct_all = env_u32("PBRHIP_DIRECT", 1u) != 0u;  // scenes of principled surfaces only (no hair, no media): no k_classify on any bounce (C2 frame -3 %)
    const uint32_t pipe_depth = std::min(std::max(1u, env_u32("PBRHIP_PIPE_DEPTH", 2u)), kRingSlots - 1u);
    const uint32_t pipe_depth_small = std::min(std::max(1u, env_u32("PBRHIP_PIPE_DEPTH_SMALL", 8u)), kRingSlots - 1u);  // below 256 Ki live paths (renders without k_tail)
    // close to the hand-over to k_tail (live paths <= pipe_stop x tail_paths) nothing is enqueued ahead: the hand-over is decided on
    // exact counts (an iteration enqueued ahead would run as a full wavefront iteration on what k_tail finishes faster)
    const double pipe_stop = getenv("PBRHIP_PIPE_STOP") ? atof(getenv("PBRHIP_PIPE_STOP")) : 2.0;
    const uint32_t shadow_first = env_u32("PBRHIP_SHADOW_FIRST", 1u);
    HIPCHK(s->heads.reserve((size_t)kMaxGroups * kTraceHeads * kHeadStride));
    HIPCHK(s->susp.reserve((size_t)std::max(1u, max_lanes) * 2u * kSuspRecords * kSuspWords));  // (264 MB per group in flight)
    struct Group {
      PathState P;
      uint32_t n0, first_pass, npass, slot0;  // paths at the start, pass range, first path slot
      uint32_t n = 0, iters = 0;
      int lane = -1;  // stream / counter / spill slot while active
      bool started = false, finished = false;
      uint32_t enq = 0, seen = 0;     // iterations enqueued / heard of (ring stamps)
      uint32_t stamps[kRingSlots];    // stamp of enqueued iteration i at [i % kRingSlots]
      bool tail_enqueued = false;
      Timer tm;
    };

    bool stop = false;
    for (; done < d->num_sample && !stop;) {
      if (cancelled()) break;  // render.cc:217
      const uint32_t np = std::min(chunk_passes, d->num_sample - done);
      HIPCHK(hipStreamSynchronize(st));  // clears / the previous chunk's accumulates are done before groups start
      const std::vector<uint32_t> plan = plan_groups(np, npix, want_groups);
      const uint32_t ng = (uint32_t)plan.size();
      std::vector<Group> G(ng);
      for (uint32_t g = 0, p0 = 0; g < ng; p0 += plan[g], g++) {
        Group& gr = G[g];
        gr.P = P, gr.n0 = plan[g] * npix, gr.first_pass = d->first_pass + done + p0, gr.npass = plan[g], gr.slot0 = p0 * npix;
        const size_t off = gr.slot0;
        gr.P.q_in += off, gr.P.q_out += off, gr.P.q_principled += off, gr.P.q_hair += off, gr.P.q_sss += off, gr.P.q_shadow += off, gr.P.q_shadow_in += off;
        for (int k = 0; k < 3; k++) gr.P.cam_org[k] = cam.org[k];
        gr.P.cam = cam, gr.P.pix_index = s->path_pix.p, gr.P.npix = npix, gr.P.width = d->width, gr.P.first_pass = gr.first_pass;
        gr.P.slot0 = gr.slot0, gr.P.seed_seq = d->seed_seq;
        gr.P.pass_run = pass_run_for(gr.npass, sc.num_curves != 0);
        gr.P.shadow_first = shadow_first, gr.P.susp_turns = 0u;
        gr.tm = Timer{s, want_timing, nullptr};
        gr.tm.only = trace_timing_only ? &S.ms_trace_closest : nullptr;
        gr.tm.idle_acc = &S.ms_host_idle;
      }
      bool lane_busy[kMaxGroups] = {};
      auto lane_stream = [&](int lane) { return lane == 0 ? st : s->group_streams[lane - 1]; };
      auto ring_slot = [&](int lane, uint32_t i) { return (size_t)(lane * kRingSlots + i % kRingSlots) * 4u; };
      // every enqueued iteration ends with k_advance, which tells the host (ring) what is left
      auto advance = [&](Group& gr, hipStream_t gst) {
        const uint32_t stamp = ++s->ring_stamp ? s->ring_stamp : ++s->ring_stamp;  // (never 0: the rings start zeroed)
        gr.stamps[gr.enq % kRingSlots] = stamp;
        launch_advance(gst, gr.P, s->d_ring + ring_slot(gr.lane, gr.enq), stamp);
        gr.enq++;
      };
      // one iteration of group gr, its launches sized for at most n_upper live paths (and as many pending shadow rays)
      auto enqueue_iteration = [&](Group& gr, uint32_t n_upper, bool to_tail) -> int {
        hipStream_t gst = lane_stream(gr.lane);
        const uint32_t n = std::max(n_upper, 1u);
        gr.P.first = gr.iters++ == 0 ? 1u : 0u;
        // the launch's suspend records: written by this k_trace, read by the next (alternating halves of the lane's area)
        uint32_t* const susp_lane = s->susp.p + (size_t)gr.lane * 2u * kSuspRecords * kSuspWords;
        gr.P.susp_out = susp_lane + (size_t)(gr.iters & 1u) * kSuspRecords * kSuspWords;
        gr.P.susp_in = susp_lane + (size_t)((gr.iters & 1u) ^ 1u) * kSuspRecords * kSuspWords;
        gr.P.susp_turns = to_tail ? 0u : susp_turns;  // (k_tail takes every path to its end: the rays in front of it all finish)
        gr.tm.iteration_begins();
        HIPCHK(gr.tm.begin(&S.ms_trace_closest));
        gr.P.wave_log_launch = wave_log_launches++;
        launch_trace(gst, gr.P, sc, 2 * n, want_stats);  // this bounce's closest rays + last bounce's shadow rays
        HIPCHK(gr.tm.end());
        S.n_trace_closest++, S.iterations++;
        if (to_tail) {
          // few live paths: after this bounce's trace (and the pending shadow rays) every path is finished in one launch
          HIPCHK(gr.tm.begin(&S.ms_tail));
          launch_tail(gst, gr.P, sc, n, rng_inc, want_stats, s->has_sss, s->has_textured);
          HIPCHK(gr.tm.end());
          S.n_tail++;
          gr.tm.iteration_ends();
          advance(gr, gst);  // nothing was queued: both "in" counts become 0
          gr.tail_enqueued = true;
          return PBRHIP_OK;
        }
        // a first bounce in a scene without hair needs no routing -- every hit takes the principled shader --: the shading kernel
        // walks the group's paths itself (PathState::direct)
        // (media do not matter here: no path is inside one before its first shading)
        const bool direct = !s->has_hair && ((gr.P.first && env_u32("PBRHIP_FIRST_DIRECT", 1u) != 0u) || (!s->has_sss && direct_all));
        gr.P.direct = direct ? 1u : 0u;
        if (!direct) {
          HIPCHK(gr.tm.begin(&S.ms_surface));
          launch_classify(gst, gr.P, sc, n);
          HIPCHK(gr.tm.end());
          S.n_surface++;
        }
        HIPCHK(gr.tm.begin(&S.ms_shade_principled));
        launch_shade_principled(gst, gr.P, sc, n, rng_inc, s->has_sss, s->has_textured);
        HIPCHK(gr.tm.end());
        if (s->has_hair) {
          HIPCHK(gr.tm.begin(&S.ms_shade_hair));
          launch_shade_hair(gst, gr.P, sc, n, rng_inc);
          HIPCHK(gr.tm.end());
          S.n_shade_hair++;
        }
        if (s->has_sss) {
          HIPCHK(gr.tm.begin(&S.ms_sss_step));
          if (sss_walk) launch_sss_walk(gst, gr.P, sc, n, rng_inc, want_stats);  // every walk forward to its last event ...
          launch_sss_step(gst, gr.P, sc, n, rng_inc);                            // ... which the step kernel handles
          HIPCHK(gr.tm.end());
          S.n_sss_step++;
        }
        HIPCHK(gr.tm.begin(&S.ms_compact));
        launch_compact(gst, gr.P, n);
        HIPCHK(gr.tm.end());
        S.n_shade_principled++;
        gr.tm.iteration_ends();
        advance(gr, gst);
        std::swap(gr.P.q_in, gr.P.q_out);
        std::swap(gr.P.q_shadow, gr.P.q_shadow_in);
        return PBRHIP_OK;
      };
      // keeps group gr's stream fed: iterations are enqueued ahead of the counts the host has seen (gr.n = the last count heard:
      // an upper bound for every later iteration)
      auto feed = [&](Group& gr) -> int {
        if (gr.tail_enqueued) return PBRHIP_OK;
        const uint32_t depth = gr.n < (1u << 18) ? pipe_depth_small : pipe_depth;
        while (gr.enq - gr.seen < depth) {
          const bool exact = gr.enq == gr.seen;  // the host knows this iteration's input counts
          if (gr.n <= tail_paths) {
            if (int rc = enqueue_iteration(gr, gr.n, true)) return rc;
            break;
          }
          if (!exact && tail_paths && (double)gr.n <= pipe_stop * (double)tail_paths) break;
          if (int rc = enqueue_iteration(gr, gr.n, false)) return rc;
        }
        return PBRHIP_OK;
      };
      auto start = [&](Group& gr, int lane) -> int {
        gr.lane = lane, gr.started = true, gr.n = gr.n0, lane_busy[lane] = true;
        hipStream_t gst = lane_stream(lane);
        gr.tm.stream = gst;
        gr.P.counts = s->counts.p + lane * kCntNum;
        gr.P.spill = s->spill.p + (size_t)lane * kSpillWords;
        gr.P.heads = s->heads.p + (size_t)lane * kTraceHeads * kHeadStride;
        HIPCHK(hipMemsetAsync(gr.P.heads, 0, sizeof(uint32_t) * kTraceHeads * kHeadStride, gst));
        uint32_t* hc = s->h_counts + lane * kCntNum;
        memset(hc, 0, sizeof(uint32_t) * kCntNum);
        hc[kCntIn] = gr.n0;
        HIPCHK(hipMemcpyAsync(gr.P.counts, hc, sizeof(uint32_t) * kCntNum, hipMemcpyHostToDevice, gst));
        HIPCHK(gr.tm.begin(&S.ms_generate));
        launch_generate(gst, gr.P, gr.n0);
        HIPCHK(gr.tm.end());
        return feed(gr);
      };
      // Scheduler: poll the active groups' rings; a group whose oldest enqueued iteration has reported gets more work enqueued
      // behind what is still running; a finished group frees its lane; passes are accumulated (ascending, on the main stream) as
      // soon as every earlier group of the chunk is complete, and *finish_pass follows.  *cancel is read on every turn.
      uint32_t next_start = 0, acc_prefix = 0, active = 0, acc_passes = 0, idle_polls = 0;
      for (;;) {
        if (!stop && cancelled()) stop = true;
        // start groups while the window allows
        while (!stop && next_start < ng) {
          uint32_t bulk = 0;
          for (const Group& gr : G)
            if (gr.started && !gr.finished && gr.n > std::max<uint64_t>(tail_paths, bulk_div ? gr.n0 / bulk_div : 0u)) bulk++;
          int lane = -1;
          for (int l = 0; l < (int)std::max(1u, max_lanes); l++)
            if (!lane_busy[l]) {
              lane = l;
              break;
            }
          if (bulk >= window || lane < 0) break;
          if (int rc = start(G[next_start], lane)) return rc;
          next_start++, active++;
        }
        if (active == 0) break;
        bool progressed = false;
        for (Group& gr : G) {
          if (!gr.started || gr.finished) continue;
          // the oldest iteration the host has not heard of: has its k_advance written the stamp?
          while (gr.seen < gr.enq) {
            const volatile uint32_t* slot = s->h_ring + ring_slot(gr.lane, gr.seen);
            if (__atomic_load_n(&slot[3], __ATOMIC_ACQUIRE) != gr.stamps[gr.seen % kRingSlots]) break;
            progressed = true;
            if (slot[2]) return fail(PBRHIP_EOVERFLOW, "BVH traversal stack overflow");
            gr.n = std::max(slot[0], slot[1]);  // pending shadow rays need one more trace
            gr.seen++;
            if (trace_sched)
              fprintf(stderr, "sched %8.3f ms  group %d (passes %u)  iter %u of %u enqueued  live %u\n",
                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(),
                      (int)(&gr - G.data()), gr.npass, gr.seen, gr.enq, gr.n);
          }
          if (gr.seen == gr.enq && (gr.n == 0 || stop)) {  // complete -- or abandoned: a cancelled render drops what is in flight
            HIPCHK(hipStreamSynchronize(lane_stream(gr.lane)));  // (its last k_advance has written the stamp: the stream is about to be idle)
            HIPCHK(gr.tm.collect());
            gr.finished = gr.n == 0;
            if (!gr.finished) gr.started = false;
            lane_busy[gr.lane] = false, active--;
            progressed = true;
            continue;
          }
          if (gr.n != 0 && !stop)
            if (int rc = feed(gr)) return rc;
        }
        // accumulate the complete prefix of the chunk's groups
        while (acc_prefix < ng && G[acc_prefix].finished) {
          const Group& gr = G[acc_prefix];
          PathState PA = P;
          PA.L = P.L + gr.slot0, PA.pass_run = gr.P.pass_run;
          Timer tm{s, want_timing, st};
          tm.only = trace_timing_only ? &S.ms_trace_closest : nullptr;
          HIPCHK(tm.begin(&S.ms_accumulate));
          launch_accumulate(st, PA, s->path_pix.p, npix, gr.npass, d_rgba, d_count);
          HIPCHK(tm.end());
          HIPCHK(hipGetLastError());
          if (want_timing && !trace_timing_only) {  // (trace-only timing has nothing to collect here and must not stall the host)
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(tm.collect());
          }
          acc_passes += gr.npass, acc_prefix++;
          S.samples += (uint64_t)gr.npass * npix;
          publish((size_t)done + acc_passes);  // render.cc:224-231
        }
        if (!progressed) {
          std::this_thread::yield();
          // (now and then: a stream that failed, or went idle without its last stamp, must not leave this loop spinning)
          if ((++idle_polls & 1023u) == 0u)
            for (Group& gr : G) {
              if (!gr.started || gr.finished || gr.seen == gr.enq) continue;
              const hipError_t q = hipStreamQuery(lane_stream(gr.lane));
              if (q == hipErrorNotReady) continue;
              HIPCHK(q);
              const volatile uint32_t* slot = s->h_ring + ring_slot(gr.lane, gr.enq - 1u);
              if (__atomic_load_n(&slot[3], __ATOMIC_ACQUIRE) != gr.stamps[(gr.enq - 1u) % kRingSlots])
                return fail(PBRHIP_EHIP, "render: a path group's stream went idle without reporting its last iteration");
            }
        } else {
          idle_polls = 0;
        }
      }
      HIPCHK(hipStreamSynchronize(st));
      done += acc_passes;
      S.chunks++;
    }
    HIPCHK(hipStreamSynchronize(st));
    if (wave_log_path) {
      std::vector<unsigned long long> h((size_t)kWaveLogLaunches * kWaveLogWaves * 4);
      HIPCHK(hipMemcpy(h.data(), wave_log.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      if (FILE* f = fopen(wave_log_path, "wb")) {
        fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
        fclose(f);
      }
    }
    if (want_stats) {
      unsigned long long hs[kStatNum];
      HIPCHK(hipMemcpy(hs, s->stats.p, sizeof(hs), hipMemcpyDeviceToHost));
      S.closest_rays = hs[kStatClosestRays] - hs[kStatSuspended] - hs[kStatHeld], S.closest_nodes = hs[kStatClosestNodes];  // (a suspended ray is counted by the launch that suspends it and by the one that resumes it)
      S.suspended_rays = hs[kStatSuspended] + hs[kStatSuspendedShadow];
      S.closest_tris = hs[kStatClosestTris], S.closest_curves = hs[kStatClosestCurves];
      S.shadow_rays = hs[kStatShadowRays] - hs[kStatSuspendedShadow], S.shadow_nodes = hs[kStatShadowNodes];
      S.tail_closest_rays = hs[kStatTailClosestRays], S.tail_shadow_rays = hs[kStatTailShadowRays];
      S.pruned_rays = hs[kStatPrunedRays];
      S.shadow_tris = hs[kStatShadowTris], S.shadow_curves = hs[kStatShadowCurves];
      if (getenv("PBRHIP_PV_STATS"))
        fprintf(stderr, "pv closest: it node %llu tri %llu curve %llu refill %llu | lanes/iter node %.1f tri %.1f curve %.1f\n",
                hs[kStatPvItNode], hs[kStatPvItTri], hs[kStatPvItCurve], hs[kStatPvItRefill],
                hs[kStatPvLnNode] / (double)std::max<unsigned long long>(1, hs[kStatPvItNode]),
                hs[kStatPvLnTri] / (double)std::max<unsigned long long>(1, hs[kStatPvItTri]),
                hs[kStatPvLnCurve] / (double)std::max<unsigned long long>(1, hs[kStatPvItCurve]));
      if (getenv("PBRHIP_PV_STATS"))
        fprintf(stderr, "pv cycles per wave turn (shader clock, lane 0 of every wave, from one turn's start to the next's): node %.0f  triangle %.0f  curve %.0f  refill %.0f | share of the waves' time: %.3f %.3f %.3f %.3f\n",
                hs[kStatCycNode] / (double)std::max<unsigned long long>(1, hs[kStatPvItNode]), hs[kStatCycTri] / (double)std::max<unsigned long long>(1, hs[kStatPvItTri]),
                hs[kStatCycCurve] / (double)std::max<unsigned long long>(1, hs[kStatPvItCurve]), hs[kStatCycRefill] / (double)std::max<unsigned long long>(1, hs[kStatPvItRefill]),
                hs[kStatCycNode] / (double)std::max<unsigned long long>(1, hs[kStatCycNode] + hs[kStatCycTri] + hs[kStatCycCurve] + hs[kStatCycRefill]),
                hs[kStatCycTri] / (double)std::max<unsigned long long>(1, hs[kStatCycNode] + hs[kStatCycTri] + hs[kStatCycCurve] + hs[kStatCycRefill]),
                hs[kStatCycCurve] / (double)std::max<unsigned long long>(1, hs[kStatCycNode] + hs[kStatCycTri] + hs[kStatCycCurve] + hs[kStatCycRefill]),
                hs[kStatCycRefill] / (double)std::max<unsigned long long>(1, hs[kStatCycNode] + hs[kStatCycTri] + hs[kStatCycCurve] + hs[kStatCycRefill]));
      if (getenv("PBRHIP_PV_STATS")) {
        fprintf(stderr, "pv steps per closest-hit ray (<=16, 32, 64, 128, 256, 512, 1024, more):");
        for (int i = 0; i < 8; i++) fprintf(stderr, " %llu", hs[kStatStepHist0 + i]);
        fprintf(stderr, " | max %llu | most loop turns of one wave (whole render) %llu\n", hs[kStatMaxSteps], hs[kStatMaxWaveIters]);
        fprintf(stderr, "pv steps per shadow ray:");
        for (int i = 0; i < 8; i++) fprintf(stderr, " %llu", hs[kStatAnyHist0 + i]);
        fprintf(stderr, " | max %llu\n", hs[kStatAnyMaxSteps]);
        fprintf(stderr, "walk: nodes %llu prims %llu | wave turns: traversal %llu, step / refill %llu | cycles per traversal turn %.0f, per step / refill turn %.0f (share %.3f)\n", hs[kStatWalkNodes], hs[kStatWalkTris],
                hs[kStatWalkTurns], hs[kStatWalkSteps], hs[kStatWalkCycTrav] / (double)std::max<unsigned long long>(1, hs[kStatWalkTurns]),
                hs[kStatWalkCycStep] / (double)std::max<unsigned long long>(1, hs[kStatWalkSteps]),
                hs[kStatWalkCycStep] / (double)std::max<unsigned long long>(1, hs[kStatWalkCycStep] + hs[kStatWalkCycTrav]));
      }
    }
  } else {
    HIPCHK(hipStreamSynchronize(st));
    done = d->num_sample;
    publish(done);
  }
  S.passes_done = done;
  S.node_bytes = trace_uses_wide(s->dscene) ? sizeof(QNode) : sizeof(BvhNode);
  S.curve_bytes = trace_uses_wide(s->dscene) ? 32 : 64;
  S.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  if (stats) *stats = S;
  return PBRHIP_OK;
}

extern "C" int pbrhip_render_device(pbrhip_scene* s, const pbrhip_render_desc* d, const volatile unsigned char* cancel,
                                    float* d_rgba, uint32_t* d_count, size_t* finish_pass, pbrhip_render_stats* stats) {
  return guarded([&]() -> int {
    if (!s || !d || !d_rgba || !d_count) return fail(PBRHIP_EINVAL, "render: NULL argument");
    return render_impl(s, d, cancel, d_rgba, d_count, finish_pass, stats);
  });
}

extern "C" int pbrhip_render(pbrhip_scene* s, const pbrhip_render_desc* d, const volatile unsigned char* cancel,
                             float* rgba, uint32_t* count, size_t* finish_pass, pbrhip_render_stats* stats) {
  return guarded([&]() -> int {
    if (!s || !d || !rgba || !count) return fail(PBRHIP_EINVAL, "render: NULL argument");
    auto t_begin = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(s->device));
    size_t npx = (size_t)d->width * d->height;
    HIPCHK(s->own_rgba.reserve(npx * 4));
    HIPCHK(s->own_count.reserve(npx));
    if (d->flags & PBRHIP_RENDER_NO_CLEAR) {
      HIPCHK(hipMemcpyAsync(s->own_rgba.p, rgba, npx * 4 * sizeof(float), hipMemcpyHostToDevice, s->stream));
      HIPCHK(hipMemcpyAsync(s->own_count.p, count, npx * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    }
    int rc = render_impl(s, d, cancel, s->own_rgba.p, s->own_count.p, finish_pass, stats);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(rgba, s->own_rgba.p, npx * 4 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(count, s->own_count.p, npx * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (stats) stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return PBRHIP_OK;
  });
}

// ------------------------------------------------------------------ test hooks
extern "C" int pbrhip_texture_fetch(pbrhip_scene* s, uint32_t texture_id, const float* uv, size_t n, float* rgb) {
  return guarded([&]() -> int {
    if (!s || ((!uv || !rgb) && n)) return fail(PBRHIP_EINVAL, "texture_fetch: NULL argument");
    if (!s->committed) return fail(PBRHIP_ESTATE, "scene not committed");
    if (texture_id >= s->tex_descs.size()) return fail(PBRHIP_EINVAL, "texture_fetch: texture id %u out of range", texture_id);
    if (n > (1u << 24)) return fail(PBRHIP_EINVAL, "texture_fetch: too many coordinates");
    if (!n) return PBRHIP_OK;
    HIPCHK(hipSetDevice(s->device));
    DevBuf<float> d_uv, d_rgb;
    HIPCHK(d_uv.reserve(2 * n));
    HIPCHK(d_rgb.reserve(3 * n));
    HIPCHK(hipMemcpyAsync(d_uv.p, uv, 2 * n * sizeof(float), hipMemcpyHostToDevice, s->stream));
    launch_texture_fetch(s->stream, s->dscene, texture_id, d_uv.p, (uint32_t)n, d_rgb.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(rgb, d_rgb.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PBRHIP_OK;
  });
}

extern "C" int pbrhip_leaf_eval(uint32_t op, const float* in, size_t n, uint32_t in_words, float* out, uint32_t out_words) {
  return guarded([&]() -> int {
    if ((!in || !out) && n) return fail(PBRHIP_EINVAL, "leaf_eval: NULL argument");
    if (op > 10u || in_words == 0 || out_words == 0 || n > (1u << 24)) return fail(PBRHIP_EINVAL, "leaf_eval: bad operation or sizes");
    static const uint32_t need_in[11] = {4, 3, 2, 2, 2, 2, 2, 9, 8, 29, 30}, need_out[11] = {1, 1, 1, 1, 5, 3, 2, 2, 5, 4, 7};
    if (in_words < need_in[op] || out_words < need_out[op]) return fail(PBRHIP_EINVAL, "leaf_eval: operation %u needs %u words in, %u out", op, need_in[op], need_out[op]);
    if (!n) return PBRHIP_OK;
    DevBuf<float> d_in, d_out;
    HIPCHK(d_in.reserve(n * in_words));
    HIPCHK(d_out.reserve(n * out_words));
    HIPCHK(hipMemcpy(d_in.p, in, n * in_words * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(d_out.p, 0, n * out_words * sizeof(float)));
    launch_leaf_eval(nullptr, op, d_in.p, (uint32_t)n, in_words, d_out.p, out_words);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, d_out.p, n * out_words * sizeof(float), hipMemcpyDeviceToHost));
    return PBRHIP_OK;
  });
}

extern "C" int pbrhip_trace_closest(pbrhip_scene* s, const pbrhip_ray* rays, size_t n, pbrhip_hit* hits) {
  return guarded([&]() -> int {
  if (!s || (!rays && n) || (!hits && n)) return fail(PBRHIP_EINVAL, "trace_closest: NULL argument");
  if (!s->committed) return fail(PBRHIP_ESTATE, "scene not committed");
  if (n == 0) return PBRHIP_OK;
  if (n >= (1ull << 31)) return fail(PBRHIP_EINVAL, "too many rays");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(s->hook_rays.reserve(2 * n));
  HIPCHK(s->hook_hits.reserve(n));
  HIPCHK(s->counts.reserve(kCntNum * kMaxGroups));
  HIPCHK(hipMemsetAsync(s->counts.p, 0, sizeof(uint32_t) * kCntNum, s->stream));
  HIPCHK(hipMemcpyAsync(s->hook_rays.p, rays, n * sizeof(pbrhip_ray), hipMemcpyHostToDevice, s->stream));
  HIPCHK(s->spill.reserve(kSpillWords));
  launch_hook_closest(s->stream, s->dscene, s->hook_rays.p, (uint32_t)n, s->hook_hits.p, s->counts.p, s->spill.p,
                      getenv("PBRHIP_SIMPLE_TRAVERSAL") != nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(hits, s->hook_hits.p, n * sizeof(pbrhip_hit), hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipMemcpyAsync(s->h_counts, s->counts.p, sizeof(uint32_t) * kCntNum, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  if (s->h_counts[kCntOverflow]) return fail(PBRHIP_EOVERFLOW, "BVH traversal stack overflow");
  return PBRHIP_OK;
  });
}
extern "C" int pbrhip_trace_any(pbrhip_scene* s, const pbrhip_ray* rays, size_t n, uint8_t* occluded) {
  return guarded([&]() -> int {
  if (!s || (!rays && n) || (!occluded && n)) return fail(PBRHIP_EINVAL, "trace_any: NULL argument");
  if (!s->committed) return fail(PBRHIP_ESTATE, "scene not committed");
  if (n == 0) return PBRHIP_OK;
  if (n >= (1ull << 31)) return fail(PBRHIP_EINVAL, "too many rays");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(s->hook_rays.reserve(2 * n));
  HIPCHK(s->hook_occ.reserve(n));
  HIPCHK(s->counts.reserve(kCntNum * kMaxGroups));
  HIPCHK(hipMemsetAsync(s->counts.p, 0, sizeof(uint32_t) * kCntNum, s->stream));
  HIPCHK(hipMemcpyAsync(s->hook_rays.p, rays, n * sizeof(pbrhip_ray), hipMemcpyHostToDevice, s->stream));
  HIPCHK(s->spill.reserve(kSpillWords));
  launch_hook_any(s->stream, s->dscene, s->hook_rays.p, (uint32_t)n, s->hook_occ.p, s->counts.p, s->spill.p,
                  getenv("PBRHIP_SIMPLE_TRAVERSAL") != nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(occluded, s->hook_occ.p, n, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipMemcpyAsync(s->h_counts, s->counts.p, sizeof(uint32_t) * kCntNum, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  if (s->h_counts[kCntOverflow]) return fail(PBRHIP_EOVERFLOW, "BVH traversal stack overflow");
  return PBRHIP_OK;
  });
}
