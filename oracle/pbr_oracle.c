/*
 * pbr_oracle.c -- ORACLE (test infrastructure only; see pbr_oracle.h for the pin status).
 *
 * Plain-C restatement of pbrlab's path-tracing hot path.  Citations are reference file:line
 * (relative to /root/reference).  Written from reading the reference; no reference source is
 * included or copied.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off, baseline x86-64).
 */
#define _GNU_SOURCE
#include "pbr_oracle.h"

#include <pthread.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>

#include "orc_closures.h"
#include "orc_math.h"

int g_orc_math_mode = ORC_MATH_LIBM;
void orc_set_math_mode(int mode) { g_orc_math_mode = mode == ORC_MATH_GLIBCF ? ORC_MATH_GLIBCF : (mode ? ORC_MATH_F64R : ORC_MATH_LIBM); }
/* Is the host's libm the one include/pbr_glibcf.h restates?  Compares cosf / sinf / expf / logf of the host libm with the
 * restatement on every `stride`-th float bit pattern (stride 1: all 2^32 arguments of each function) and returns the number of
 * differing results (two NaNs count as equal).  tests/test_glibcf.py; the GPU tests use it to decide whether "equal to oracle[libm]"
 * may be asserted as bits. */
unsigned long long orc_glibcf_vs_libm(uint32_t stride, uint32_t first) {
  unsigned long long bad = 0;
  if (stride == 0) stride = 1;
  for (uint64_t i = first; i < (1ull << 32); i += stride) {
    const uint32_t u = (uint32_t)i;
    float x;
    memcpy(&x, &u, 4);
    const float a[4] = {cosf(x), sinf(x), expf(x), logf(x)};
    const float b[4] = {glibcf_cosf(x), glibcf_sinf(x), glibcf_expf(x), glibcf_logf(x)};
    for (int k = 0; k < 4; k++)
      if (orc_f2u(a[k]) != orc_f2u(b[k]) && !(a[k] != a[k] && b[k] != b[k])) bad++;
  }
  return bad;
}
int orc_get_math_mode(void) { return g_orc_math_mode; }

#define ORC_NONE 0xFFFFFFFFu

/* ================================================================ scene model (host side) */
typedef struct {
  int kind; /* 0 = triangle mesh, 1 = cubic Bezier curve mesh (mesh/mesh.h:23) */
  /* triangles: mesh/attribute.h + mesh/triangle-mesh.h */
  float *vertices, *normals, *uvs;
  uint32_t nverts, nnormals, nuvs;
  uint32_t *vid, *nid, *tid, *mat;
  uint32_t nfaces, nmat;
  /* curves */
  float* cverts;
  uint32_t ncverts;
  uint32_t *cidx, *cmat;
  uint32_t nseg;
} orc_mesh;

typedef struct {
  uint32_t* mesh_ids;
  uint32_t nmesh;
} orc_local_scene;

typedef struct {
  uint32_t* light_param_ids; /* per prim */
  float *choose_prob, *cdf, *area_pdf;
  uint32_t nprim;
  float intensity_sum;
  uint32_t global_id;
} orc_area_light;

typedef struct {
  uint32_t local_scene;
  float xf[16];
  int identity;
  uint32_t ngeom;
  uint32_t** material_ids;
  uint32_t* nmaterial_ids;
  uint32_t** light_ids;
  uint32_t* nlight_ids;
  orc_area_light** area_lights; /* per geom, NULL if none (light-manager.h:186) */
} orc_instance;

typedef struct {
  int kind; /* 0 principled, 1 hair (material-param.h:20-23) */
  orc_principled_param pr;
  orc_hair_param hr;
} orc_material;

typedef struct {
  float choose_prob;
  uint32_t instance_id, geom_id;
} orc_light;

typedef struct orc_texture {
  float* pixels;
  uint32_t width, height, channels;
} orc_texture;

/* flattened primitive reference used by the BVH */
typedef struct {
  uint32_t instance_id, geom_id, prim_id, kind;
} orc_primref;

typedef struct {
  float lo[3], hi[3];
  int32_t left, right; /* internal: child node ids; leaf: left = -1 - first, right = count */
} orc_node;

struct orc_scene {
  orc_mesh* meshes;
  uint32_t nmeshes;
  orc_local_scene* locals;
  uint32_t nlocals;
  orc_instance* instances;
  uint32_t ninstances;
  orc_material* materials;
  uint32_t nmaterials;
  struct orc_texture* textures; /* src/texture.h:13-44 */
  uint32_t ntextures;
  f3* light_params; /* AreaLightParameter.emission */
  uint32_t nlight_params;
  orc_light* lights;
  float* light_cdf;
  uint32_t nlights;
  /* raytracer back end */
  orc_primref* prims; /* canonical (instance, geom, prim) order = "gid" */
  uint32_t nprims;
  float* prim_geo;    /* per gid: tri = 9 floats world xyz*3 ; curve = 16 floats xyzr*4 (stride 16) */
  uint32_t* order;    /* BVH leaf order -> gid */
  orc_node* nodes;
  uint32_t nnodes, bvh_depth;
  float bmin[3], bmax[3];
  int committed;
};

static void* xrealloc(void* p, size_t n) {
  void* q = realloc(p, n ? n : 1);
  if (!q) {
    fprintf(stderr, "oracle: out of memory\n");
    abort();
  }
  return q;
}
static void* xdup(const void* p, size_t n) {
  void* q = xrealloc(NULL, n);
  if (n) memcpy(q, p, n);
  return q;
}
static uint32_t* xfill_u32(uint32_t n, uint32_t v) {
  uint32_t* q = (uint32_t*)xrealloc(NULL, sizeof(uint32_t) * (size_t)n);
  for (uint32_t i = 0; i < n; i++) q[i] = v;
  return q;
}

orc_scene* orc_scene_create(void) { return (orc_scene*)calloc(1, sizeof(orc_scene)); }

void orc_scene_destroy(orc_scene* s) {
  if (!s) return;
  for (uint32_t i = 0; i < s->nmeshes; i++) {
    orc_mesh* m = &s->meshes[i];
    free(m->vertices), free(m->normals), free(m->uvs), free(m->vid), free(m->nid), free(m->tid);
    free(m->mat), free(m->cverts), free(m->cidx), free(m->cmat);
  }
  for (uint32_t i = 0; i < s->nlocals; i++) free(s->locals[i].mesh_ids);
  for (uint32_t i = 0; i < s->ninstances; i++) {
    orc_instance* in = &s->instances[i];
    for (uint32_t g = 0; g < in->ngeom; g++) {
      free(in->material_ids[g]);
      free(in->light_ids[g]);
      if (in->area_lights && in->area_lights[g]) {
        orc_area_light* a = in->area_lights[g];
        free(a->light_param_ids), free(a->choose_prob), free(a->cdf), free(a->area_pdf), free(a);
      }
    }
    free(in->material_ids), free(in->nmaterial_ids), free(in->light_ids), free(in->nlight_ids);
    free(in->area_lights);
  }
  for (uint32_t i = 0; i < s->ntextures; i++) free(s->textures[i].pixels);
  free(s->textures);
  free(s->meshes), free(s->locals), free(s->instances), free(s->materials), free(s->light_params);
  free(s->lights), free(s->light_cdf), free(s->prims), free(s->prim_geo), free(s->order), free(s->nodes);
  free(s);
}

int orc_add_triangle_mesh(orc_scene* s, const float* vertices_xyzw, uint32_t num_vertices,
                          const float* normals_xyzw, uint32_t num_normals, const float* texcoords_uv,
                          uint32_t num_texcoords, const uint32_t* vertex_ids, const uint32_t* normal_ids,
                          const uint32_t* texcoord_ids, const uint32_t* material_ids, uint32_t num_faces) {
  s->meshes = (orc_mesh*)xrealloc(s->meshes, sizeof(orc_mesh) * (s->nmeshes + 1));
  orc_mesh* m = &s->meshes[s->nmeshes];
  memset(m, 0, sizeof(*m));
  m->kind = 0;
  m->vertices = (float*)xdup(vertices_xyzw, sizeof(float) * 4 * (size_t)num_vertices);
  m->nverts = num_vertices;
  m->normals = (float*)xdup(normals_xyzw, sizeof(float) * 4 * (size_t)num_normals);
  m->nnormals = num_normals;
  m->uvs = (float*)xdup(texcoords_uv, sizeof(float) * 2 * (size_t)num_texcoords);
  m->nuvs = num_texcoords;
  m->nfaces = num_faces;
  m->vid = (uint32_t*)xdup(vertex_ids, sizeof(uint32_t) * 3 * (size_t)num_faces);
  /* triangle-mesh.cc:33-55 : absent id arrays become all -1 */
  m->nid = normal_ids ? (uint32_t*)xdup(normal_ids, sizeof(uint32_t) * 3 * (size_t)num_faces)
                      : xfill_u32(num_faces * 3, ORC_NONE);
  m->tid = texcoord_ids ? (uint32_t*)xdup(texcoord_ids, sizeof(uint32_t) * 3 * (size_t)num_faces)
                        : xfill_u32(num_faces * 3, ORC_NONE);
  if (material_ids) {
    m->mat = (uint32_t*)xdup(material_ids, sizeof(uint32_t) * (size_t)num_faces);
    m->nmat = num_faces;
  } else {
    m->mat = xfill_u32(num_faces * 3, ORC_NONE); /* triangle-mesh.cc:53-55 (sic: 3 per face) */
    m->nmat = num_faces * 3;
  }
  return (int)s->nmeshes++;
}

int orc_add_curve_mesh(orc_scene* s, const float* vertices_xyzr, uint32_t num_vertices,
                       const uint32_t* indices, const uint32_t* material_ids, uint32_t num_segments) {
  s->meshes = (orc_mesh*)xrealloc(s->meshes, sizeof(orc_mesh) * (s->nmeshes + 1));
  orc_mesh* m = &s->meshes[s->nmeshes];
  memset(m, 0, sizeof(*m));
  m->kind = 1;
  m->cverts = (float*)xdup(vertices_xyzr, sizeof(float) * 4 * (size_t)num_vertices);
  m->ncverts = num_vertices;
  m->cidx = (uint32_t*)xdup(indices, sizeof(uint32_t) * (size_t)num_segments);
  m->cmat = material_ids ? (uint32_t*)xdup(material_ids, sizeof(uint32_t) * (size_t)num_segments)
                         : xfill_u32(num_segments, ORC_NONE);
  m->nseg = num_segments;
  return (int)s->nmeshes++;
}

int orc_add_principled(orc_scene* s, const orc_principled_param* p) {
  s->materials = (orc_material*)xrealloc(s->materials, sizeof(orc_material) * (s->nmaterials + 1));
  memset(&s->materials[s->nmaterials], 0, sizeof(orc_material));
  s->materials[s->nmaterials].kind = 0;
  s->materials[s->nmaterials].pr = *p;
  return (int)s->nmaterials++;
}
int orc_add_hair(orc_scene* s, const orc_hair_param* p) {
  s->materials = (orc_material*)xrealloc(s->materials, sizeof(orc_material) * (s->nmaterials + 1));
  memset(&s->materials[s->nmaterials], 0, sizeof(orc_material));
  s->materials[s->nmaterials].kind = 1;
  s->materials[s->nmaterials].hr = *p;
  return (int)s->nmaterials++;
}
/* Scene::AddTexture (scene.h:46-51) with Texture(pixels, width, height, channels) (texture.cc:10-21) */
int orc_add_texture(orc_scene* s, const float* pixels, uint32_t width, uint32_t height, uint32_t channels) {
  s->textures = (orc_texture*)xrealloc(s->textures, sizeof(orc_texture) * (s->ntextures + 1));
  orc_texture* t = &s->textures[s->ntextures];
  t->pixels = (float*)xdup(pixels, sizeof(float) * (size_t)width * height * channels);
  t->width = width, t->height = height, t->channels = channels;
  return (int)s->ntextures++;
}
/* BilinearFilter with clamp addressing (image-utils.cc:99-167) behind Texture::FetchFloat3 (texture.cc:43-68):
 * px = width*u (no half-texel offset), channels beyond the image's are 0 */
static void texture_fetch3(const orc_texture* t, float u, float v, float dst[3]) {
  float uu = orc_max(u, 0.0f);
  uu = orc_min(uu, 1.0f);
  float vv = orc_max(v, 0.0f);
  vv = orc_min(vv, 1.0f);
  const int width = (int)t->width, height = (int)t->height, stride = (int)t->channels;
  const float px = (float)t->width * uu;
  const float py = (float)t->height * vv;
  int x0 = (int)px, y0 = (int)py;
  x0 = x0 < width - 1 ? x0 : width - 1;   /* std::min(int(width) - 1, int(px)) */
  x0 = x0 > 0 ? x0 : 0;                   /* std::max(0, ...) */
  y0 = y0 < height - 1 ? y0 : height - 1;
  y0 = y0 > 0 ? y0 : 0;
  const int x1 = ((x0 + 1) >= width) ? (width - 1) : (x0 + 1);
  const int y1 = ((y0 + 1) >= height) ? (height - 1) : (y0 + 1);
  const float dx = px - (float)x0;
  const float dy = py - (float)y0;
  const float w0 = (1.0f - dx) * (1.0f - dy), w1 = (1.0f - dx) * dy, w2 = dx * (1.0f - dy), w3 = dx * dy;
  const int i00 = stride * (y0 * width + x0), i01 = stride * (y0 * width + x1);
  const int i10 = stride * (y1 * width + x0), i11 = stride * (y1 * width + x1);
  for (int i = 0; i < 3; i++) {
    if (i < stride)
      dst[i] = t->pixels[i00 + i] * w0 + t->pixels[i10 + i] * w1 + t->pixels[i01 + i] * w2 + t->pixels[i11 + i] * w3;
    else
      dst[i] = 0.f;
  }
}
void orc_kat_texture_fetch(const float* pixels, uint32_t width, uint32_t height, uint32_t channels, float u, float v,
                           float out[3]) {
  orc_texture t = {(float*)pixels, width, height, channels};
  texture_fetch3(&t, u, v, out);
}

int orc_add_area_light(orc_scene* s, const float emission[3]) {
  s->light_params = (f3*)xrealloc(s->light_params, sizeof(f3) * (s->nlight_params + 1));
  s->light_params[s->nlight_params] = f3_make(emission[0], emission[1], emission[2]);
  return (int)s->nlight_params++;
}
int orc_create_local_scene(orc_scene* s) {
  s->locals = (orc_local_scene*)xrealloc(s->locals, sizeof(orc_local_scene) * (s->nlocals + 1));
  memset(&s->locals[s->nlocals], 0, sizeof(orc_local_scene));
  return (int)s->nlocals++;
}
int orc_add_mesh_to_local_scene(orc_scene* s, uint32_t local_scene_id, uint32_t mesh_id) {
  if (local_scene_id >= s->nlocals || mesh_id >= s->nmeshes) return -1;
  orc_local_scene* l = &s->locals[local_scene_id];
  l->mesh_ids = (uint32_t*)xrealloc(l->mesh_ids, sizeof(uint32_t) * (l->nmesh + 1));
  l->mesh_ids[l->nmesh] = mesh_id;
  return (int)l->nmesh++;
}
static uint32_t mesh_num_prims(const orc_mesh* m) { return m->kind == 0 ? m->nfaces : m->nseg; }

/* scene.cc:106-155 : material ids are copied from the meshes at instancing time */
int orc_create_instance(orc_scene* s, uint32_t local_scene_id, const float transform[16]) {
  if (local_scene_id >= s->nlocals) return -1;
  s->instances = (orc_instance*)xrealloc(s->instances, sizeof(orc_instance) * (s->ninstances + 1));
  orc_instance* in = &s->instances[s->ninstances];
  memset(in, 0, sizeof(*in));
  in->local_scene = local_scene_id;
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(in->xf, transform ? transform : ident, sizeof(float) * 16);
  in->identity = memcmp(in->xf, ident, sizeof(ident)) == 0;
  const orc_local_scene* l = &s->locals[local_scene_id];
  in->ngeom = l->nmesh;
  in->material_ids = (uint32_t**)calloc(l->nmesh ? l->nmesh : 1, sizeof(uint32_t*));
  in->nmaterial_ids = (uint32_t*)calloc(l->nmesh ? l->nmesh : 1, sizeof(uint32_t));
  in->light_ids = (uint32_t**)calloc(l->nmesh ? l->nmesh : 1, sizeof(uint32_t*));
  in->nlight_ids = (uint32_t*)calloc(l->nmesh ? l->nmesh : 1, sizeof(uint32_t));
  for (uint32_t g = 0; g < l->nmesh; g++) {
    const orc_mesh* m = &s->meshes[l->mesh_ids[g]];
    if (m->kind == 0) {
      in->material_ids[g] = (uint32_t*)xdup(m->mat, sizeof(uint32_t) * m->nmat);
      in->nmaterial_ids[g] = m->nmat;
    } else {
      in->material_ids[g] = (uint32_t*)xdup(m->cmat, sizeof(uint32_t) * m->nseg);
      in->nmaterial_ids[g] = m->nseg;
    }
  }
  return (int)s->ninstances++;
}
/* scene.cc:64-72 */
int orc_attach_light_ids(orc_scene* s, uint32_t instance_id, uint32_t geom_id, const uint32_t* ids, uint32_t n) {
  if (instance_id >= s->ninstances) return -1;
  orc_instance* in = &s->instances[instance_id];
  if (geom_id >= in->ngeom) return -1;
  const orc_mesh* m = &s->meshes[s->locals[in->local_scene].mesh_ids[geom_id]];
  if (n != 0 && n != mesh_num_prims(m)) return -2;
  free(in->light_ids[geom_id]);
  in->light_ids[geom_id] = (uint32_t*)xdup(ids, sizeof(uint32_t) * n);
  in->nlight_ids[geom_id] = n;
  return 0;
}
/* scene.cc:74-94 */
int orc_attach_material_ids(orc_scene* s, uint32_t instance_id, uint32_t geom_id, const uint32_t* ids, uint32_t n) {
  if (instance_id >= s->ninstances) return -1;
  orc_instance* in = &s->instances[instance_id];
  if (geom_id >= in->ngeom) return -1;
  const orc_mesh* m = &s->meshes[s->locals[in->local_scene].mesh_ids[geom_id]];
  if (n != mesh_num_prims(m)) return -2;
  free(in->material_ids[geom_id]);
  in->material_ids[geom_id] = (uint32_t*)xdup(ids, sizeof(uint32_t) * n);
  in->nmaterial_ids[geom_id] = n;
  return 0;
}

static const orc_mesh* inst_mesh(const orc_scene* s, uint32_t instance_id, uint32_t geom_id) {
  const orc_instance* in = &s->instances[instance_id];
  return &s->meshes[s->locals[in->local_scene].mesh_ids[geom_id]];
}
static f3 mesh_vertex(const orc_mesh* m, uint32_t prim, int k) {
  const float* p = m->vertices + (size_t)m->vid[prim * 3 + k] * 4;
  return f3_make(p[0], p[1], p[2]);
}
/* mesh/triangle-mesh.cc:181-184 */
static f3 calc_geometry_normal(f3 p0, f3 p1, f3 p2) {
  return f3_normalize(f3_cross(f3_sub(p1, p0), f3_sub(p2, p1)));
}
/* mesh/triangle-mesh.cc:62-75 */
static f3 mesh_geometry_normal(const orc_mesh* m, uint32_t prim) {
  return calc_geometry_normal(mesh_vertex(m, prim, 0), mesh_vertex(m, prim, 1), mesh_vertex(m, prim, 2));
}
/* mesh/triangle-mesh.cc:113-124 */
static float mesh_face_area(const orc_mesh* m, uint32_t prim) {
  f3 p0 = mesh_vertex(m, prim, 0), p1 = mesh_vertex(m, prim, 1), p2 = mesh_vertex(m, prim, 2);
  return f3_length(f3_cross(f3_sub(p1, p0), f3_sub(p2, p0))) * 0.5f;
}
/* mesh/triangle-mesh.cc:77-101 */
static f3 mesh_shading_normal(const orc_mesh* m, uint32_t prim, float u, float v) {
  uint32_t a = m->nid[prim * 3 + 0], b = m->nid[prim * 3 + 1], c = m->nid[prim * 3 + 2];
  if (a == ORC_NONE || b == ORC_NONE || c == ORC_NONE) return mesh_geometry_normal(m, prim);
  const float *na = m->normals + (size_t)a * 4, *nb = m->normals + (size_t)b * 4, *nc = m->normals + (size_t)c * 4;
  return f3_normalize(f3_lerp3(f3_make(na[0], na[1], na[2]), f3_make(nb[0], nb[1], nb[2]),
                               f3_make(nc[0], nc[1], nc[2]), u, v));
}
/* mesh/triangle-mesh.cc:102-112 */
static f3 mesh_local_position(const orc_mesh* m, uint32_t prim, float u, float v) {
  return f3_lerp3(mesh_vertex(m, prim, 0), mesh_vertex(m, prim, 1), mesh_vertex(m, prim, 2), u, v);
}

/* ============================================================== light tables (A8) */
/* light-manager.cc:79-184 */
static void register_instance_lights(orc_scene* s, uint32_t instance_id) {
  orc_instance* in = &s->instances[instance_id];
  if (in->area_lights) {
    for (uint32_t g = 0; g < in->ngeom; g++) {
      orc_area_light* a = in->area_lights[g];
      if (a) free(a->light_param_ids), free(a->choose_prob), free(a->cdf), free(a->area_pdf), free(a);
    }
    free(in->area_lights);
  }
  in->area_lights = (orc_area_light**)calloc(in->ngeom ? in->ngeom : 1, sizeof(orc_area_light*));
  for (uint32_t g = 0; g < in->ngeom; g++) {
    const uint32_t* ids = in->light_ids[g];
    if (in->nlight_ids[g] == 0) continue;
    const orc_mesh* m = inst_mesh(s, instance_id, g);
    if (m->kind != 0) continue;
    uint32_t nf = m->nfaces;
    int have = 0;
    for (uint32_t f = 0; f < nf; f++)
      if (ids[f] != ORC_NONE) {
        have = 1;
        break;
      }
    if (!have) continue;
    orc_area_light* a = (orc_area_light*)calloc(1, sizeof(orc_area_light));
    a->nprim = nf;
    a->light_param_ids = (uint32_t*)xdup(ids, sizeof(uint32_t) * nf);
    a->choose_prob = (float*)calloc(nf, sizeof(float));
    a->cdf = (float*)calloc(nf, sizeof(float));
    a->area_pdf = (float*)calloc(nf, sizeof(float));
    for (uint32_t f = 0; f < nf; f++) {
      float intensity = 0.0f;
      if (ids[f] != ORC_NONE) intensity = orc_spectrum_norm(s->light_params[ids[f]]);
      a->choose_prob[f] = intensity * mesh_face_area(m, f);
    }
    float sum = 0.0f; /* std::accumulate(..., 0.0f) */
    for (uint32_t f = 0; f < nf; f++) sum = sum + a->choose_prob[f];
    a->intensity_sum = sum;
    for (uint32_t f = 0; f < nf; f++) a->choose_prob[f] = a->choose_prob[f] / sum;
    for (uint32_t f = 0; f < nf; f++) a->cdf[f] = a->choose_prob[f];
    for (uint32_t f = 0; nf > 0 && f < nf - 1u; f++) a->cdf[f + 1u] += a->cdf[f];
    for (uint32_t f = 0; f < nf; f++)
      if (ids[f] != ORC_NONE) a->area_pdf[f] = 1.0f / mesh_face_area(m, f);
    a->global_id = ORC_NONE;
    in->area_lights[g] = a;
  }
}
/* light-manager.cc:29-77 */
static void commit_lights(orc_scene* s) {
  free(s->lights), free(s->light_cdf);
  s->lights = NULL, s->light_cdf = NULL, s->nlights = 0;
  double intensity_sum = 0.0;
  for (uint32_t i = 0; i < s->ninstances; i++) {
    orc_instance* in = &s->instances[i];
    for (uint32_t g = 0; g < in->ngeom; g++) {
      orc_area_light* a = in->area_lights[g];
      if (!a) continue;
      a->global_id = s->nlights;
      s->lights = (orc_light*)xrealloc(s->lights, sizeof(orc_light) * (s->nlights + 1));
      s->lights[s->nlights].choose_prob = a->intensity_sum;
      s->lights[s->nlights].instance_id = i;
      s->lights[s->nlights].geom_id = g;
      intensity_sum += (double)a->intensity_sum;
      s->nlights++;
    }
  }
  for (uint32_t l = 0; l < s->nlights; l++)
    s->lights[l].choose_prob = (float)((double)s->lights[l].choose_prob / intensity_sum);
  s->light_cdf = (float*)calloc(s->nlights ? s->nlights : 1, sizeof(float));
  for (uint32_t l = 0; l < s->nlights; l++) s->light_cdf[l] = s->lights[l].choose_prob;
  for (uint32_t l = 0; s->nlights > 0 && l < s->nlights - 1u; l++) s->light_cdf[l + 1u] += s->light_cdf[l];
}

/* std::lower_bound on a float CDF, clamped to n-1 (Q10) */
static uint32_t cdf_lower_bound(const float* cdf, uint32_t n, float u) {
  uint32_t lo = 0, len = n;
  while (len > 0) {
    uint32_t half = len >> 1;
    if (cdf[lo + half] < u) {
      lo = lo + half + 1;
      len = len - half - 1;
    } else {
      len = half;
    }
  }
  return lo < n ? lo : n - 1;
}

/* ===================================================== raytracer back end (own BVH, A5/A6) */
/* v' = v * M with translation row (matrix.cc:218-222) */
static f3 xf_point(const float m[16], f3 v) {
  return f3_make(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12], m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13],
                 m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14]);
}

#define ORC_GEO_STRIDE 16

static void flatten_prims(orc_scene* s) {
  uint32_t n = 0;
  for (uint32_t i = 0; i < s->ninstances; i++)
    for (uint32_t g = 0; g < s->instances[i].ngeom; g++) n += mesh_num_prims(inst_mesh(s, i, g));
  s->nprims = n;
  s->prims = (orc_primref*)xrealloc(s->prims, sizeof(orc_primref) * n);
  s->prim_geo = (float*)xrealloc(s->prim_geo, sizeof(float) * ORC_GEO_STRIDE * (size_t)n);
  uint32_t k = 0;
  for (uint32_t i = 0; i < s->ninstances; i++) {
    const orc_instance* in = &s->instances[i];
    for (uint32_t g = 0; g < in->ngeom; g++) {
      const orc_mesh* m = inst_mesh(s, i, g);
      uint32_t np = mesh_num_prims(m);
      for (uint32_t p = 0; p < np; p++, k++) {
        s->prims[k].instance_id = i;
        s->prims[k].geom_id = g;
        s->prims[k].prim_id = p;
        s->prims[k].kind = (uint32_t)m->kind;
        float* geo = s->prim_geo + (size_t)k * ORC_GEO_STRIDE;
        memset(geo, 0, sizeof(float) * ORC_GEO_STRIDE);
        if (m->kind == 0) {
          for (int c = 0; c < 3; c++) {
            f3 v = mesh_vertex(m, p, c);
            if (!in->identity) v = xf_point(in->xf, v);
            geo[c * 3 + 0] = v.x, geo[c * 3 + 1] = v.y, geo[c * 3 + 2] = v.z;
          }
        } else {
          for (int c = 0; c < 4; c++) {
            const float* cp = m->cverts + ((size_t)m->cidx[p] + c) * 4;
            f3 v = f3_make(cp[0], cp[1], cp[2]);
            if (!in->identity) v = xf_point(in->xf, v);
            geo[c * 4 + 0] = v.x, geo[c * 4 + 1] = v.y, geo[c * 4 + 2] = v.z, geo[c * 4 + 3] = cp[3];
          }
        }
      }
    }
  }
}

static inline void bezier_eval(const float* cp, float u, float out[4]);

/* Box of one contract primitive: a triangle's corners; a linear piece of a ribbon = its two end points B(i/4), B((i+1)/4)
 * widened by the larger end radius.  This box is part of the intersection contract (hit_inside below). */
static inline void tri_box(const float* geo, float lo[3], float hi[3]) {
  for (int a = 0; a < 3; a++) {
    lo[a] = fminf(fminf(geo[a], geo[3 + a]), geo[6 + a]);
    hi[a] = fmaxf(fmaxf(geo[a], geo[3 + a]), geo[6 + a]);
  }
}
static inline void piece_box(const float a[4], const float b[4], float lo[3], float hi[3]) {
  float r = fmaxf(fabsf(a[3]), fabsf(b[3]));
  for (int k = 0; k < 3; k++) lo[k] = fminf(a[k], b[k]) - r, hi[k] = fmaxf(a[k], b[k]) + r;
}

/* hull != 0: the bounds Embree reports for the scene (rtcGetSceneBounds; they place the camera): for a curve the convex
 * hull of the control points widened by the largest control radius.  hull == 0: the box the tree is built over: the
 * union of the boxes of the curve's four contract primitives. */
static void prim_bounds(const orc_scene* s, uint32_t gid, int hull, float lo[3], float hi[3]) {
  const float* geo = s->prim_geo + (size_t)gid * ORC_GEO_STRIDE;
  for (int a = 0; a < 3; a++) lo[a] = INFINITY, hi[a] = -INFINITY;
  if (s->prims[gid].kind == 0) {
    for (int c = 0; c < 3; c++)
      for (int a = 0; a < 3; a++) {
        float v = geo[c * 3 + a];
        if (v < lo[a]) lo[a] = v;
        if (v > hi[a]) hi[a] = v;
      }
  } else if (hull) {
    float r = 0.0f;
    for (int c = 0; c < 4; c++) {
      float rc = fabsf(geo[c * 4 + 3]);
      if (rc > r) r = rc;
      for (int a = 0; a < 3; a++) {
        float v = geo[c * 4 + a];
        if (v < lo[a]) lo[a] = v;
        if (v > hi[a]) hi[a] = v;
      }
    }
    for (int a = 0; a < 3; a++) lo[a] -= r, hi[a] += r;
  } else {
    float p[5][4];
    for (int i = 0; i < 5; i++) bezier_eval(geo, (float)i * 0.25f, p[i]);
    for (int i = 0; i < 4; i++) {
      float l[3], h[3];
      piece_box(p[i], p[i + 1], l, h);
      for (int a = 0; a < 3; a++) {
        if (l[a] < lo[a]) lo[a] = l[a];
        if (h[a] > hi[a]) hi[a] = h[a];
      }
    }
  }
}

/* rtcGetSceneBounds (raytracer_impl.cc:199-202) of the global scene: the union over the instances of the instance's
 * bounds.  An instance (RTC_GEOMETRY_TYPE_INSTANCE, raytracer_impl.cc:61-81) reports the box of the transformed CORNERS of
 * its local scene's box, which under rotation or shear is larger than the box of the transformed geometry; an instance
 * whose matrix is bit for bit the identity reports the local box itself.  The local box: triangles by their corners,
 * curves by the hull of their control points widened by the largest control radius. */
static void scene_bounds(orc_scene* s) {
  for (int a = 0; a < 3; a++) s->bmin[a] = INFINITY, s->bmax[a] = -INFINITY;
  for (uint32_t i = 0; i < s->ninstances; i++) {
    const orc_instance* in = &s->instances[i];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int any = 0;
    for (uint32_t g = 0; g < in->ngeom; g++) {
      const orc_mesh* m = inst_mesh(s, i, g);
      uint32_t np = mesh_num_prims(m);
      for (uint32_t p = 0; p < np; p++) {
        any = 1;
        if (m->kind == 0) {
          for (int c = 0; c < 3; c++) {
            f3 v = mesh_vertex(m, p, c);
            const float w[3] = {v.x, v.y, v.z};
            for (int a = 0; a < 3; a++) lo[a] = fminf(lo[a], w[a]), hi[a] = fmaxf(hi[a], w[a]);
          }
        } else {
          float r = 0.0f, cl[3] = {INFINITY, INFINITY, INFINITY}, ch[3] = {-INFINITY, -INFINITY, -INFINITY};
          for (int c = 0; c < 4; c++) {
            const float* cp = m->cverts + ((size_t)m->cidx[p] + c) * 4;
            r = fmaxf(r, fabsf(cp[3]));
            for (int a = 0; a < 3; a++) cl[a] = fminf(cl[a], cp[a]), ch[a] = fmaxf(ch[a], cp[a]);
          }
          for (int a = 0; a < 3; a++) lo[a] = fminf(lo[a], cl[a] - r), hi[a] = fmaxf(hi[a], ch[a] + r);
        }
      }
    }
    if (!any) continue;
    if (in->identity) {
      for (int a = 0; a < 3; a++) s->bmin[a] = fminf(s->bmin[a], lo[a]), s->bmax[a] = fmaxf(s->bmax[a], hi[a]);
    } else {
      for (int c = 0; c < 8; c++) {
        f3 v = xf_point(in->xf, f3_make((c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]));
        const float w[3] = {v.x, v.y, v.z};
        for (int a = 0; a < 3; a++) s->bmin[a] = fminf(s->bmin[a], w[a]), s->bmax[a] = fmaxf(s->bmax[a], w[a]);
      }
    }
  }
}

typedef struct {
  orc_scene* s;
  float *plo, *phi, *pc; /* per-gid bounds and centroids */
  uint32_t node_cap, max_depth;
} orc_builder;

static float box_area(const float lo[3], const float hi[3]) {
  float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0f;
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}

#define ORC_BINS 12
#define ORC_LEAF_MAX 4

static uint32_t build_node(orc_builder* b, uint32_t first, uint32_t count, uint32_t depth) {
  orc_scene* s = b->s;
  if (s->nnodes >= b->node_cap) {
    b->node_cap = b->node_cap * 2 + 16;
    s->nodes = (orc_node*)xrealloc(s->nodes, sizeof(orc_node) * b->node_cap);
  }
  uint32_t id = s->nnodes++;
  if (depth > b->max_depth) b->max_depth = depth;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t i = first; i < first + count; i++) {
    uint32_t g = s->order[i];
    for (int a = 0; a < 3; a++) {
      if (b->plo[g * 3 + a] < lo[a]) lo[a] = b->plo[g * 3 + a];
      if (b->phi[g * 3 + a] > hi[a]) hi[a] = b->phi[g * 3 + a];
      if (b->pc[g * 3 + a] < clo[a]) clo[a] = b->pc[g * 3 + a];
      if (b->pc[g * 3 + a] > chi[a]) chi[a] = b->pc[g * 3 + a];
    }
  }
  memcpy(s->nodes[id].lo, lo, sizeof(lo));
  memcpy(s->nodes[id].hi, hi, sizeof(hi));
  if (count <= ORC_LEAF_MAX) {
    s->nodes[id].left = -1 - (int32_t)first;
    s->nodes[id].right = (int32_t)count;
    return id;
  }
  /* binned SAH over the centroid bounds; falls back to a median split */
  int best_axis = -1, best_bin = 0;
  float best_cost = INFINITY;
  if (depth < 40) {
    for (int a = 0; a < 3; a++) {
      float ext = chi[a] - clo[a];
      if (!(ext > 0.0f)) continue;
      uint32_t cnt[ORC_BINS];
      float blo[ORC_BINS][3], bhi[ORC_BINS][3];
      for (int k = 0; k < ORC_BINS; k++) {
        cnt[k] = 0;
        for (int c = 0; c < 3; c++) blo[k][c] = INFINITY, bhi[k][c] = -INFINITY;
      }
      float scale = (float)ORC_BINS / ext;
      for (uint32_t i = first; i < first + count; i++) {
        uint32_t g = s->order[i];
        int k = (int)((b->pc[g * 3 + a] - clo[a]) * scale);
        if (k >= ORC_BINS) k = ORC_BINS - 1;
        if (k < 0) k = 0;
        cnt[k]++;
        for (int c = 0; c < 3; c++) {
          if (b->plo[g * 3 + c] < blo[k][c]) blo[k][c] = b->plo[g * 3 + c];
          if (b->phi[g * 3 + c] > bhi[k][c]) bhi[k][c] = b->phi[g * 3 + c];
        }
      }
      float rarea[ORC_BINS];
      uint32_t rcnt[ORC_BINS];
      float alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
      uint32_t acc = 0;
      for (int k = ORC_BINS - 1; k > 0; k--) {
        acc += cnt[k];
        for (int c = 0; c < 3; c++) {
          if (blo[k][c] < alo[c]) alo[c] = blo[k][c];
          if (bhi[k][c] > ahi[c]) ahi[c] = bhi[k][c];
        }
        rarea[k] = box_area(alo, ahi);
        rcnt[k] = acc;
      }
      for (int c = 0; c < 3; c++) alo[c] = INFINITY, ahi[c] = -INFINITY;
      acc = 0;
      for (int k = 0; k < ORC_BINS - 1; k++) {
        acc += cnt[k];
        for (int c = 0; c < 3; c++) {
          if (blo[k][c] < alo[c]) alo[c] = blo[k][c];
          if (bhi[k][c] > ahi[c]) ahi[c] = bhi[k][c];
        }
        if (acc == 0 || rcnt[k + 1] == 0) continue;
        float cost = box_area(alo, ahi) * (float)acc + rarea[k + 1] * (float)rcnt[k + 1];
        if (cost < best_cost) {
          best_cost = cost;
          best_axis = a;
          best_bin = k; /* bins 0..k go left */
        }
      }
    }
  }
  uint32_t mid = first;
  if (best_axis >= 0) {
    float scale = (float)ORC_BINS / (chi[best_axis] - clo[best_axis]);
    uint32_t i = first, j = first + count;
    while (i < j) {
      uint32_t g = s->order[i];
      int k = (int)((b->pc[g * 3 + best_axis] - clo[best_axis]) * scale);
      if (k >= ORC_BINS) k = ORC_BINS - 1;
      if (k < 0) k = 0;
      if (k <= best_bin) {
        i++;
      } else {
        j--;
        s->order[i] = s->order[j];
        s->order[j] = g;
      }
    }
    mid = i;
  }
  if (mid == first || mid == first + count) {
    /* median split along the widest centroid axis (insertion-free: simple nth by sort of a copy) */
    int a = 0;
    if (chi[1] - clo[1] > chi[a] - clo[a]) a = 1;
    if (chi[2] - clo[2] > chi[a] - clo[a]) a = 2;
    /* shell sort on the sub-range (rarely reached) */
    for (uint32_t gap = count / 2; gap > 0; gap /= 2)
      for (uint32_t i = first + gap; i < first + count; i++) {
        uint32_t g = s->order[i];
        float key = b->pc[g * 3 + a];
        uint32_t j = i;
        while (j >= first + gap && b->pc[s->order[j - gap] * 3 + a] > key) {
          s->order[j] = s->order[j - gap];
          j -= gap;
        }
        s->order[j] = g;
      }
    mid = first + count / 2;
  }
  uint32_t l = build_node(b, first, mid - first, depth + 1);
  uint32_t r = build_node(b, mid, first + count - mid, depth + 1);
  s->nodes[id].left = (int32_t)l;
  s->nodes[id].right = (int32_t)r;
  return id;
}

static void build_bvh(orc_scene* s) {
  orc_builder b;
  memset(&b, 0, sizeof(b));
  b.s = s;
  uint32_t n = s->nprims;
  b.plo = (float*)xrealloc(NULL, sizeof(float) * 3 * (size_t)n);
  b.phi = (float*)xrealloc(NULL, sizeof(float) * 3 * (size_t)n);
  b.pc = (float*)xrealloc(NULL, sizeof(float) * 3 * (size_t)n);
  s->order = (uint32_t*)xrealloc(s->order, sizeof(uint32_t) * (size_t)n);
  for (uint32_t g = 0; g < n; g++) {
    prim_bounds(s, g, 0, b.plo + g * 3, b.phi + g * 3);
    for (int a = 0; a < 3; a++) b.pc[g * 3 + a] = 0.5f * (b.plo[g * 3 + a] + b.phi[g * 3 + a]);
    s->order[g] = g;
  }
  scene_bounds(s);
  s->nnodes = 0;
  b.node_cap = 0;
  if (n > 0) build_node(&b, 0, n, 1);
  s->bvh_depth = b.max_depth;
  free(b.plo), free(b.phi), free(b.pc);
}

/* scene.cc:96-104 */
int orc_commit(orc_scene* s) {
  for (uint32_t i = 0; i < s->ninstances; i++) register_instance_lights(s, i);
  commit_lights(s);
  flatten_prims(s);
  build_bvh(s);
  s->committed = 1;
  return 0;
}
void orc_scene_aabb(const orc_scene* s, float bmin[3], float bmax[3]) {
  memcpy(bmin, s->bmin, sizeof(float) * 3);
  memcpy(bmax, s->bmax, sizeof(float) * 3);
}
uint32_t orc_bvh_depth(const orc_scene* s) { return s->bvh_depth; }

/* ---------------------------------------------------------------- intersection kernels
 * Conventions (DESIGN.md "intersection contract"; Embree's are not reproducible here, F5):
 *  - a primitive hit is accepted for  tmin < t <= tmax  if, in addition, t lies inside the interval in which the ray
 *    crosses the primitive's OWN box (hit_inside): a numerically degenerate sliver can pass the triangle test with a
 *    meaningless distance far from the sliver; without this rule whether such a hit is ever seen depends on the order
 *    in which subtrees are visited (a nearer true hit culls the sliver's box), with it the accepted set is a property of
 *    (ray, primitive) alone;
 *  - closest hit = smallest accepted t, ties broken towards the smaller canonical primitive id
 *    (instance, geom, prim order), so the result does not depend on BVH shape or visit order: rtcIntersect1's
 *    "closest accepted hit" (raytracer_impl.cc:268-278);
 *  - any-hit = "some primitive has an accepted hit" (rtcOccluded1, raytracer_impl.cc:280-287).
 * Why the rule makes every tree agree with the brute-force loop: every box above a primitive contains that primitive's
 * box; the slab arithmetic below is monotone in the box (rounded subtraction, multiplication, min, max all are), the
 * validation box is strictly inside the stored one and both widen the interval with the same function -- so whenever
 * tmin < t <= tmax holds for a validated hit, the conservative test of every ancestor passes for [tmin, tmax].
 */
/* conservative slab test: the box is widened by 2^-16 relative in space (a ray parallel to an axis with its origin exactly
 * on a face of the tight box would give 0 * inf = NaN and be rejected) and the interval by 2^-16 relative before comparing */
static inline float box_lo(float v) { return v - (fabsf(v) * 1.52587890625e-05f + 1e-30f); }
static inline float box_hi(float v) { return v + (fabsf(v) * 1.52587890625e-05f + 1e-30f); }
/* the interval in which the ray is inside the box lo..hi (faces as given), widened by 2^-16 relative */
static inline void slab_interval(const float lo[3], const float hi[3], f3 o, f3 inv, float* ta, float* tb) {
  float t0 = (lo[0] - o.x) * inv.x, t1 = (hi[0] - o.x) * inv.x;
  float a = fminf(t0, t1), b = fmaxf(t0, t1);
  t0 = (lo[1] - o.y) * inv.y, t1 = (hi[1] - o.y) * inv.y;
  a = fmaxf(a, fminf(t0, t1)), b = fminf(b, fmaxf(t0, t1));
  t0 = (lo[2] - o.z) * inv.z, t1 = (hi[2] - o.z) * inv.z;
  a = fmaxf(a, fminf(t0, t1)), b = fminf(b, fmaxf(t0, t1));
  *ta = fmaf(-fabsf(a), 1.52587890625e-05f, a);
  *tb = fmaf(fabsf(b), 1.52587890625e-05f, b);
}
static inline int box_test(const float lo[3], const float hi[3], f3 o, f3 inv, float tmin, float tmax, float* tnear) {
  const float l[3] = {box_lo(lo[0]), box_lo(lo[1]), box_lo(lo[2])}, h[3] = {box_hi(hi[0]), box_hi(hi[1]), box_hi(hi[2])};
  float a, b;
  slab_interval(l, h, o, inv, &a, &b);
  *tnear = a;
  return a <= b && b >= tmin && a <= tmax;
}
/* The validation of the contract: lo / hi = the primitive's own box (tri_box / piece_box); it is widened by 2^-17
 * relative + 1e-31 (strictly inside what box_lo / box_hi store for any box that contains it, strictly outside the
 * geometry) and the hit distance has to lie in the ray's interval through it. */
static inline float vbox_lo(float v) { return fmaf(-fabsf(v), 7.62939453125e-06f, v) - 1e-31f; }
static inline float vbox_hi(float v) { return fmaf(fabsf(v), 7.62939453125e-06f, v) + 1e-31f; }
static inline int hit_inside(const float lo[3], const float hi[3], f3 o, f3 inv, float t) {
  const float l[3] = {vbox_lo(lo[0]), vbox_lo(lo[1]), vbox_lo(lo[2])}, h[3] = {vbox_hi(hi[0]), vbox_hi(hi[1]), vbox_hi(hi[2])};
  float a, b;
  slab_interval(l, h, o, inv, &a, &b);
  return a <= t && t <= b;
}

typedef struct {
  float t, u, v;
  uint32_t gid;
  f3 ng; /* unnormalised geometric normal (tri) or tangent dP/du (curve) */
} orc_isect;

static inline int tri_test(const float* geo, f3 o, f3 d, f3 invd, float tmin, float* t, float* u, float* v) {
  f3 v0 = f3_make(geo[0], geo[1], geo[2]), v1 = f3_make(geo[3], geo[4], geo[5]), v2 = f3_make(geo[6], geo[7], geo[8]);
  f3 e1 = f3_sub(v1, v0), e2 = f3_sub(v2, v0);
  f3 p = f3_cross(d, e2);
  float det = f3_dot(e1, p);
  if (!(det != 0.0f)) return 0;
  float inv = 1.0f / det;
  f3 s = f3_sub(o, v0);
  float uu = f3_dot(s, p) * inv;
  if (!(uu >= 0.0f && uu <= 1.0f)) return 0;
  f3 q = f3_cross(s, e1);
  float vv = f3_dot(d, q) * inv;
  if (!(vv >= 0.0f && uu + vv <= 1.0f)) return 0;
  float tt = f3_dot(e2, q) * inv;
  if (!(tt > tmin)) return 0;
  float lo[3], hi[3];
  tri_box(geo, lo, hi);
  if (!hit_inside(lo, hi, o, invd, tt)) return 0;
  *t = tt, *u = uu, *v = vv;
  return 1;
}

/* Pixar branchless ONB (shader-utils.h:44-50), reused for the ray frame of the ribbon test */
static inline void branchless_onb(f3 n, f3* x, f3* y) {
  float sign = copysignf(1.0f, n.z);
  float a = -1.0f / (sign + n.z);
  float b = n.x * n.y * a;
  *x = f3_make(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
  *y = f3_make(b, sign + n.y * n.y * a, -n.y);
}

static inline void bezier_eval(const float* cp, float u, float out[4]) {
  float s = 1.0f - u;
  float b0 = s * s * s, b1 = 3.0f * u * s * s, b2 = 3.0f * u * u * s, b3 = u * u * u;
  for (int c = 0; c < 4; c++) out[c] = ((cp[c] * b0 + cp[4 + c] * b1) + cp[8 + c] * b2) + cp[12 + c] * b3;
}
static inline f3 bezier_tangent(const float* cp, float u) {
  float s = 1.0f - u;
  float c0 = 3.0f * s * s, c1 = 6.0f * u * s, c2 = 3.0f * u * u;
  f3 p0 = f3_make(cp[0], cp[1], cp[2]), p1 = f3_make(cp[4], cp[5], cp[6]);
  f3 p2 = f3_make(cp[8], cp[9], cp[10]), p3 = f3_make(cp[12], cp[13], cp[14]);
  return f3_add(f3_add(f3_scale(f3_sub(p1, p0), c0), f3_scale(f3_sub(p2, p1), c1)), f3_scale(f3_sub(p3, p2), c2));
}

/* Ray-facing flat ribbon (RTC_GEOMETRY_TYPE_FLAT_BEZIER_CURVE semantics, raytracer_impl.cc:158-159):
 * 4 linear sub-segments per cubic, u = curve parameter, v in [-1,1] across the width. */
static inline int curve_test(const float* cp, f3 o, f3 d, f3 invd, float tmin, float tmax, float* t, float* u, float* v) {
  float inv_len = 1.0f / sqrtf(f3_dot(d, d));
  f3 dn = f3_scale(d, inv_len);
  f3 bx, by;
  branchless_onb(dn, &bx, &by);
  float px[5], py[5], pz[5], pr[5], pc[5][4];
  for (int i = 0; i < 5; i++) {
    float* c = pc[i];
    bezier_eval(cp, (float)i * 0.25f, c);
    f3 rel = f3_sub(f3_make(c[0], c[1], c[2]), o);
    px[i] = f3_dot(rel, bx), py[i] = f3_dot(rel, by), pz[i] = f3_dot(rel, dn), pr[i] = c[3];
  }
  int found = 0;
  float best = tmax;
  for (int i = 0; i < 4; i++) {
    float ex = px[i + 1] - px[i], ey = py[i + 1] - py[i];
    float len2 = ex * ex + ey * ey;
    if (!(len2 > 0.0f)) continue;
    float s = -(px[i] * ex + py[i] * ey) / len2;
    if (!(s >= 0.0f && s <= 1.0f)) continue;
    float dist = (ey * px[i] - ex * py[i]) / sqrtf(len2);
    float r = pr[i] + s * (pr[i + 1] - pr[i]);
    if (!(r > 0.0f && fabsf(dist) <= r)) continue;
    float tt = (pz[i] + s * (pz[i + 1] - pz[i])) * inv_len;
    if (!(tt > tmin)) continue;
    if (found ? !(tt < best) : !(tt <= best)) continue;
    float lo[3], hi[3];
    piece_box(pc[i], pc[i + 1], lo, hi); /* every linear piece is a primitive of the contract */
    if (!hit_inside(lo, hi, o, invd, tt)) continue;
    best = tt, found = 1;
    *t = tt, *u = ((float)i + s) * 0.25f, *v = dist / r;
  }
  return found;
}

typedef struct {
  uint64_t nodes, tris, curves;
} orc_trav_stats;

static inline int prim_test(const orc_scene* s, uint32_t gid, f3 o, f3 d, f3 inv, float tmin, float tmax, orc_isect* is,
                            orc_trav_stats* st) {
  const float* geo = s->prim_geo + (size_t)gid * ORC_GEO_STRIDE;
  float t, u, v;
  if (s->prims[gid].kind == 0) {
    if (st) st->tris++;
    if (!tri_test(geo, o, d, inv, tmin, &t, &u, &v)) return 0;
    if (!(t <= tmax)) return 0;
  } else {
    if (st) st->curves++;
    if (!curve_test(geo, o, d, inv, tmin, tmax, &t, &u, &v)) return 0;
  }
  is->t = t, is->u = u, is->v = v, is->gid = gid;
  return 1;
}

static int better_hit(float t, uint32_t gid, float best_t, uint32_t best_gid) {
  return (t < best_t) || (t == best_t && gid < best_gid);
}

/* Ng as Embree reports it for an instanced geometry: in the instance's LOCAL space (the ray is intersected in object
 * space; raytracer_impl.cc:221-233 normalises it and never transforms it).  prim_geo holds what the raytracer sees (the
 * transformed primitive), so the normal is taken from the mesh itself. */
static void fill_ng(const orc_scene* s, f3 d, orc_isect* is) {
  (void)d;
  const orc_primref* pr = &s->prims[is->gid];
  const orc_mesh* m = inst_mesh(s, pr->instance_id, pr->geom_id);
  if (pr->kind == 0) {
    f3 v0 = mesh_vertex(m, pr->prim_id, 0), v1 = mesh_vertex(m, pr->prim_id, 1), v2 = mesh_vertex(m, pr->prim_id, 2);
    is->ng = f3_cross(f3_sub(v1, v0), f3_sub(v2, v0));
  } else {
    is->ng = bezier_tangent(m->cverts + (size_t)m->cidx[pr->prim_id] * 4, is->u);
  }
}

static int closest_hit(const orc_scene* s, f3 o, f3 d, float tmin, float tmax, int brute, orc_isect* out,
                       orc_trav_stats* st) {
  float best_t = tmax;
  uint32_t best_gid = ORC_NONE;
  orc_isect best, cur;
  memset(&best, 0, sizeof(best));
  int found = 0;
  const f3 inv = f3_make(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  if (brute) {
    for (uint32_t g = 0; g < s->nprims; g++)
      if (prim_test(s, g, o, d, inv, tmin, best_t, &cur, st) && better_hit(cur.t, cur.gid, best_t, best_gid)) {
        best = cur, best_t = cur.t, best_gid = cur.gid, found = 1;
      }
  } else if (s->nnodes > 0) {
    uint32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
      const orc_node* n = &s->nodes[stack[--sp]];
      float tn;
      if (st) st->nodes++;
      if (!box_test(n->lo, n->hi, o, inv, tmin, best_t, &tn)) continue;
      if (n->left < 0) {
        uint32_t first = (uint32_t)(-1 - n->left), cnt = (uint32_t)n->right;
        for (uint32_t i = first; i < first + cnt; i++) {
          uint32_t g = s->order[i];
          if (prim_test(s, g, o, d, inv, tmin, best_t, &cur, st) && better_hit(cur.t, cur.gid, best_t, best_gid)) {
            best = cur, best_t = cur.t, best_gid = cur.gid, found = 1;
          }
        }
      } else {
        /* push far child first so the near one is popped next */
        const orc_node *l = &s->nodes[n->left], *r = &s->nodes[n->right];
        float tl, tr;
        int hl = box_test(l->lo, l->hi, o, inv, tmin, best_t, &tl);
        int hr = box_test(r->lo, r->hi, o, inv, tmin, best_t, &tr);
        if (hl && hr) {
          if (tl <= tr) {
            stack[sp++] = (uint32_t)n->right, stack[sp++] = (uint32_t)n->left;
          } else {
            stack[sp++] = (uint32_t)n->left, stack[sp++] = (uint32_t)n->right;
          }
        } else if (hl) {
          stack[sp++] = (uint32_t)n->left;
        } else if (hr) {
          stack[sp++] = (uint32_t)n->right;
        }
      }
    }
  }
  if (found) {
    fill_ng(s, d, &best);
    *out = best;
  }
  return found;
}

static int any_hit(const orc_scene* s, f3 o, f3 d, float tmin, float tmax, int brute, orc_trav_stats* st) {
  orc_isect cur;
  const f3 inv = f3_make(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  if (brute) {
    for (uint32_t g = 0; g < s->nprims; g++)
      if (prim_test(s, g, o, d, inv, tmin, tmax, &cur, st)) return 1;
    return 0;
  }
  if (s->nnodes == 0) return 0;
  uint32_t stack[128];
  int sp = 0;
  stack[sp++] = 0;
  while (sp > 0) {
    const orc_node* n = &s->nodes[stack[--sp]];
    float tn;
    if (st) st->nodes++;
    if (!box_test(n->lo, n->hi, o, inv, tmin, tmax, &tn)) continue;
    if (n->left < 0) {
      uint32_t first = (uint32_t)(-1 - n->left), cnt = (uint32_t)n->right;
      for (uint32_t i = first; i < first + cnt; i++)
        if (prim_test(s, s->order[i], o, d, inv, tmin, tmax, &cur, st)) return 1;
    } else {
      stack[sp++] = (uint32_t)n->right;
      stack[sp++] = (uint32_t)n->left;
    }
  }
  return 0;
}

/* ============================================================ per-thread tracing context */
typedef struct {
  const orc_scene* scene;
  orc_stats stats;
  orc_trav_stats trav;
  orc_hit* trace_hits; /* optional hit log (orc_sample_trace) */
  uint32_t trace_cap, trace_n;
} orc_ctx;

typedef struct {
  f3 dir, org;
  float min_t, max_t;
} orc_rayf;

/* Raytracer::FirstHitTrace1 + EmbreeRayToTraceResult (raytracer_impl.cc:221-278) */
static orc_hit trace_first_hit(orc_ctx* c, const orc_rayf* r) {
  orc_hit h;
  h.normal_g[0] = 1.0f, h.normal_g[1] = 0.0f, h.normal_g[2] = 0.0f;
  h.t = 1.0f, h.u = 0.0f, h.v = 0.0f;
  h.instance_id = h.geom_id = h.prim_id = ORC_NONE;
  orc_isect is;
  c->stats.closest_rays++;
  float tfar = orc_min(r->max_t, INFINITY);
  if (closest_hit(c->scene, r->org, r->dir, r->min_t, tfar, 0, &is, &c->trav)) {
    f3 n = f3_normalize_raw(is.ng);
    h.normal_g[0] = n.x, h.normal_g[1] = n.y, h.normal_g[2] = n.z;
    h.t = is.t, h.u = is.u, h.v = is.v;
    const orc_primref* p = &c->scene->prims[is.gid];
    h.instance_id = p->instance_id, h.geom_id = p->geom_id, h.prim_id = p->prim_id;
  }
  if (c->trace_hits && c->trace_n < c->trace_cap) c->trace_hits[c->trace_n] = h;
  if (c->trace_hits) c->trace_n++;
  return h;
}
/* Raytracer::AnyHit1 (raytracer_impl.cc:280-287) */
static int trace_any_hit(orc_ctx* c, const orc_rayf* r) {
  c->stats.shadow_rays++;
  return any_hit(c->scene, r->org, r->dir, r->min_t, orc_min(r->max_t, INFINITY), 0, &c->trav);
}

void orc_trace_closest(const orc_scene* s, const orc_ray* rays, size_t n, orc_hit* hits, int brute_force) {
  for (size_t i = 0; i < n; i++) {
    orc_hit h;
    h.normal_g[0] = 1.0f, h.normal_g[1] = 0.0f, h.normal_g[2] = 0.0f;
    h.t = 1.0f, h.u = 0.0f, h.v = 0.0f;
    h.instance_id = h.geom_id = h.prim_id = ORC_NONE;
    orc_isect is;
    f3 o = f3_make(rays[i].org[0], rays[i].org[1], rays[i].org[2]);
    f3 d = f3_make(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]);
    if (closest_hit(s, o, d, rays[i].tmin, orc_min(rays[i].tmax, INFINITY), brute_force, &is, NULL)) {
      f3 ng = f3_normalize_raw(is.ng);
      h.normal_g[0] = ng.x, h.normal_g[1] = ng.y, h.normal_g[2] = ng.z;
      h.t = is.t, h.u = is.u, h.v = is.v;
      const orc_primref* p = &s->prims[is.gid];
      h.instance_id = p->instance_id, h.geom_id = p->geom_id, h.prim_id = p->prim_id;
    }
    hits[i] = h;
  }
}
void orc_trace_any(const orc_scene* s, const orc_ray* rays, size_t n, uint8_t* occluded, int brute_force) {
  for (size_t i = 0; i < n; i++) {
    f3 o = f3_make(rays[i].org[0], rays[i].org[1], rays[i].org[2]);
    f3 d = f3_make(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]);
    occluded[i] = (uint8_t)any_hit(s, o, d, rays[i].tmin, orc_min(rays[i].tmax, INFINITY), brute_force, NULL);
  }
}

/* ================================================================== shading (L1) */
enum { ORC_FRONT = 0, ORC_BACK = 1, ORC_AMBIGUOUS = 2 };

/* shader-utils.h:18-41 */
typedef struct {
  float u, v;
  uint32_t instance_id, geom_id, prim_id;
  f3 global_position, normal_s, normal_g;
  float texcoord[2];
  int face_direction;
  const orc_material* material;
} orc_surface;

/* 3x3 part of the reference's row-vector 4x4 matrices: rows r0,r1,r2 ; MultV = v*M (+0 translation) */
typedef struct {
  f3 r0, r1, r2;
} orc_mat3;
/* shader-utils.h:66-90 */
static orc_mat3 global_to_local(f3 ex, f3 ey, f3 ez) {
  orc_mat3 m;
  m.r0 = f3_make(ex.x, ey.x, ez.x);
  m.r1 = f3_make(ex.y, ey.y, ez.y);
  m.r2 = f3_make(ex.z, ey.z, ez.z);
  return m;
}
/* shader-utils.h:92-114 */
static orc_mat3 local_to_global(f3 ex, f3 ey, f3 ez) {
  orc_mat3 m;
  m.r0 = ex, m.r1 = ey, m.r2 = ez;
  return m;
}
/* matrix.cc:218-222 (the "+ m[3][k]" term is the zero translation row) */
static f3 mult_v(f3 v, const orc_mat3* m) {
  return f3_make(m->r0.x * v.x + m->r1.x * v.y + m->r2.x * v.z + 0.0f,
                 m->r0.y * v.x + m->r1.y * v.y + m->r2.y * v.z + 0.0f,
                 m->r0.z * v.x + m->r1.z * v.y + m->r2.z * v.z + 0.0f);
}
static orc_mat3 transpose3(const orc_mat3* m) {
  orc_mat3 t;
  t.r0 = f3_make(m->r0.x, m->r1.x, m->r2.x);
  t.r1 = f3_make(m->r0.y, m->r1.y, m->r2.y);
  t.r2 = f3_make(m->r0.z, m->r1.z, m->r2.z);
  return t;
}

/* shader-utils.h:131-164 + scene.cc:186-249 */
static orc_surface trace_result_to_surface(const orc_scene* s, const orc_rayf* ray, const orc_hit* tr) {
  orc_surface si;
  memset(&si, 0, sizeof(si));
  si.u = tr->u, si.v = tr->v;
  si.instance_id = tr->instance_id, si.geom_id = tr->geom_id, si.prim_id = tr->prim_id;
  si.global_position = f3_add(ray->org, f3_scale(ray->dir, tr->t));
  const orc_mesh* m = inst_mesh(s, tr->instance_id, tr->geom_id);
  f3 ng = f3_make(tr->normal_g[0], tr->normal_g[1], tr->normal_g[2]);
  si.normal_s = (m->kind == 0) ? mesh_shading_normal(m, tr->prim_id, tr->u, tr->v) : ng;
  si.normal_g = ng;
  /* Scene::FetchMeshTexcoord (scene.cc:230-249) + TriangleMesh::FetchTexcoord (mesh/triangle-mesh.cc:126-156) */
  si.texcoord[0] = si.texcoord[1] = 0.f;
  if (m->kind == 0) {
    uint32_t a = m->tid[tr->prim_id * 3 + 0], b = m->tid[tr->prim_id * 3 + 1], c = m->tid[tr->prim_id * 3 + 2];
    if (a == ORC_NONE || b == ORC_NONE || c == ORC_NONE) {
      si.texcoord[0] = tr->u, si.texcoord[1] = tr->v;
    } else {
      const float *ta = m->uvs + (size_t)a * 2, *tb = m->uvs + (size_t)b * 2, *tc = m->uvs + (size_t)c * 2;
      for (int k = 0; k < 2; k++) si.texcoord[k] = (1.0f - tr->u - tr->v) * ta[k] + tr->u * tb[k] + tr->v * tc[k];
    }
  }
  float dg = f3_dot(ray->dir, si.normal_g), ds = f3_dot(ray->dir, si.normal_s);
  if (dg < 0.0f && ds < 0.0f)
    si.face_direction = ORC_FRONT;
  else if (dg > 0.0f && ds > 0.0f)
    si.face_direction = ORC_BACK;
  else
    si.face_direction = ORC_AMBIGUOUS;
  const orc_instance* in = &s->instances[tr->instance_id];
  uint32_t mid = in->material_ids[tr->geom_id][tr->prim_id];
  si.material = (mid == ORC_NONE) ? NULL : &s->materials[mid];
  return si;
}

/* light-manager.h:37-74 */
static int implicit_area_light(const orc_scene* s, uint32_t instance_id, uint32_t geom_id, uint32_t prim_id,
                               f3* emission, float* pdf) {
  const orc_area_light* a = s->instances[instance_id].area_lights[geom_id];
  if (!a || a->light_param_ids[prim_id] == ORC_NONE) return 0;
  *emission = s->light_params[a->light_param_ids[prim_id]];
  *pdf = s->lights[a->global_id].choose_prob * a->choose_prob[prim_id] * a->area_pdf[prim_id];
  return 1;
}

typedef struct {
  int valid; /* kAreaLight vs kLightNone */
  f3 position, normal, emission;
  float pdf;
} orc_light_sample;

/* light-manager.h:79-170 : 4 draws (0 if the scene has no lights) */
static orc_light_sample sample_all_light(const orc_scene* s, orc_rng* rng) {
  orc_light_sample r;
  memset(&r, 0, sizeof(r));
  if (s->nlights == 0) return r;
  float u0 = orc_rng_draw(rng);
  uint32_t li = cdf_lower_bound(s->light_cdf, s->nlights, u0);
  const orc_light* L = &s->lights[li];
  const orc_area_light* a = s->instances[L->instance_id].area_lights[L->geom_id];
  float u1 = orc_rng_draw(rng);
  uint32_t pi = cdf_lower_bound(a->cdf, a->nprim, u1);
  float u2 = orc_rng_draw(rng);
  float u3 = orc_rng_draw(rng);
  float bu, bv;
  orc_triangle_uniform_sampler(u2, u3, &bu, &bv);
  const orc_mesh* m = inst_mesh(s, L->instance_id, L->geom_id);
  r.valid = 1;
  r.position = mesh_local_position(m, pi, bu, bv);
  r.normal = mesh_geometry_normal(m, pi);
  r.emission = s->light_params[a->light_param_ids[pi]];
  r.pdf = L->choose_prob * a->choose_prob[pi] * a->area_pdf[pi];
  return r;
}

/* cycles-principled-shader.cc:20-45 */
typedef struct {
  int enable_diffuse;
  f3 diffuse_weight;
  int enable_subsurface;
  f3 subsurface_weight, subsurface_albedo, subsurface_radius;
  int enable_specular;
  f3 specular_weight;
  float alpha_x, alpha_y, ior;
  f3 specular_color;
  int enable_clearcoat;
  f3 clearcoat_weight;
  float clearcoat_alpha_x, clearcoat_alpha_y, clearcoat_ior;
  f3 clearcoat_color;
} orc_bsdf;

static void bsdf_default(orc_bsdf* b) {
  memset(b, 0, sizeof(*b));
  b->alpha_x = b->alpha_y = 1.f;
  b->ior = 1.5f;
  b->clearcoat_alpha_x = b->clearcoat_alpha_y = 1.f;
  b->clearcoat_ior = 1.5f;
}

/* what DirectIllumination's EvalFunc closes over */
typedef struct {
  int is_hair;
  const orc_bsdf* bsdf;
  const orc_hair_bsdf* hair;
} orc_eval;

/* cycles-principled-shader.cc:54-61 */
static f3 specular_color_fn(f3 omega_in, f3 omega_out, f3 specular_color, float ior) {
  f3 h = f3_normalize(f3_add(omega_in, omega_out));
  float f0 = orc_fresnel_dielectric_cos(1.0f, ior);
  float fh = (orc_fresnel_dielectric_cos(f3_dot(h, omega_out), ior) - f0) / (1.0f - f0);
  return f3_add(f3_scale(specular_color, 1.f - fh), f3_set1(fh));
}

typedef struct {
  float diffuse, subsurface, specular, clearcoat;
} orc_sample_weight;

/* cycles-principled-shader.cc:63-112 */
static orc_sample_weight closure_sample_weight(f3 omega_out, const orc_bsdf* b) {
  orc_sample_weight w;
  f3 refl = f3_make(-omega_out.x, -omega_out.y, omega_out.z);
  w.diffuse = b->enable_diffuse ? orc_rgb_to_y(b->diffuse_weight) : 0.f;
  w.subsurface = b->enable_subsurface ? orc_rgb_to_y(b->subsurface_weight) : 0.f;
  w.specular = b->enable_specular
                   ? orc_rgb_to_y(f3_mul(b->specular_weight, specular_color_fn(refl, omega_out, b->specular_color, b->ior)))
                   : 0.f;
  w.clearcoat = b->enable_clearcoat
                    ? orc_rgb_to_y(f3_mul(b->clearcoat_weight,
                                          specular_color_fn(refl, omega_out, b->clearcoat_color, b->clearcoat_ior)))
                    : 0.f;
  float sum = 0.0f;
  sum += w.diffuse;
  sum += w.subsurface;
  sum += w.specular;
  sum += w.clearcoat;
  w.diffuse /= sum;
  w.subsurface /= sum;
  w.specular /= sum;
  w.clearcoat /= sum;
  if (!isfinite(w.diffuse)) w.diffuse = 0.f;
  if (!isfinite(w.subsurface)) w.subsurface = 0.f;
  if (!isfinite(w.specular)) w.specular = 0.f;
  if (!isfinite(w.clearcoat)) w.clearcoat = 0.f;
  return w;
}

/* cycles-principled-shader.cc:114-155 */
static void eval_bsdf(f3 omega_in, f3 omega_out, const orc_bsdf* b, f3* bsdf_f, float* pdf) {
  orc_sample_weight w = closure_sample_weight(omega_out, b);
  *bsdf_f = f3_set1(0.0f);
  *pdf = 0.0f;
  if (b->enable_diffuse) {
    float p = 0.f;
    float f = orc_lambert_brdf_pdf(omega_in, &p);
    *bsdf_f = f3_add(*bsdf_f, f3_scale(b->diffuse_weight, f));
    *pdf += w.diffuse * p;
  }
  if (b->enable_specular) {
    float p = 0.f;
    float f = orc_ggx_bsdf_pdf(omega_in, omega_out, b->alpha_x, b->alpha_y, 2, &p);
    *bsdf_f = f3_add(*bsdf_f,
                     f3_scale(f3_mul(b->specular_weight, specular_color_fn(omega_in, omega_out, b->specular_color, b->ior)), f));
    *pdf += w.specular * p;
  }
  if (b->enable_clearcoat) {
    float p = 0.f;
    float f = orc_ggx_bsdf_pdf(omega_in, omega_out, b->clearcoat_alpha_x, b->clearcoat_alpha_y, 1, &p);
    *bsdf_f = f3_add(*bsdf_f, f3_scale(f3_mul(b->clearcoat_weight,
                                              specular_color_fn(omega_in, omega_out, b->clearcoat_color, b->clearcoat_ior)),
                                       f));
    *pdf += w.clearcoat * p;
  }
}

static void eval_dispatch(const orc_eval* e, f3 omega_in, f3 omega_out, f3* bsdf_f, float* pdf) {
  if (e->is_hair) {
    /* hair-shader.cc:191-197 */
    f3 fcos = orc_hair_eval(omega_in, omega_out, e->hair, pdf);
    *bsdf_f = f3_divs(fcos, fabsf(omega_in.x));
  } else {
    eval_bsdf(omega_in, omega_out, e->bsdf, bsdf_f, pdf);
  }
}

/* shader-utils.h:116-129 */
static int shadow_ray(orc_ctx* c, f3 pos, f3 dir, float dist) {
  orc_rayf r;
  r.org = pos, r.dir = dir;
  r.min_t = ORC_EPS;
  r.max_t = orc_max(r.min_t, dist - ORC_EPS);
  return trace_any_hit(c, &r);
}

/* shader-utils.h:166-212 */
static f3 direct_illumination(orc_ctx* c, f3 omega_out, const orc_surface* si, const orc_mat3* Rgl, f3 global_normal,
                              orc_rng* rng, const orc_eval* ev, int hemisphere) {
  orc_light_sample ls = sample_all_light(c->scene, rng);
  f3 contribute = f3_set1(0.f);
  if (ls.valid) {
    f3 pos = si->global_position;
    f3 dir_to_light = f3_normalize(f3_sub(ls.position, si->global_position));
    float dist = f3_length(f3_sub(pos, ls.position));
    float wl_dot_nl = -f3_dot(dir_to_light, ls.normal);
    float wl_dot_np = f3_dot(dir_to_light, global_normal);
    float pdf_sigma = fabsf(ls.pdf * dist * dist / (wl_dot_nl * wl_dot_np));
    if (((!hemisphere) || (wl_dot_nl > 0.0f && wl_dot_np > 0.0f)) && !shadow_ray(c, pos, dir_to_light, dist)) {
      f3 omega_l = mult_v(dir_to_light, Rgl);
      f3 bsdf_f = f3_set1(0.0f);
      float ret_pdf = 0.f;
      eval_dispatch(ev, omega_l, omega_out, &bsdf_f, &ret_pdf);
      float weight = orc_power_heuristic(pdf_sigma, ret_pdf);
      contribute = f3_divs(f3_scale(f3_mul(bsdf_f, ls.emission), weight), pdf_sigma);
    }
  }
  return contribute;
}

/* ---------------------------------------------------- random-walk SSS (random-walk-sss.h) */
/* :35-48 */
static float burley_fitting(float A) { return 1.9f - A + 3.5f * (A - 0.8f) * (A - 0.8f); }
static float burley_fitting5(float A) { return 1.85f - A + 7.0f * fabsf((A - 0.8f) * (A - 0.8f) * (A - 0.8f)); }
/* :50-72 */
static void bssrdf_burley_setup(f3 albedo, f3 radius, int scale_mfp, int mode, f3* radius_out) {
  f3 l = scale_mfp ? f3_scale(radius, 0.25f * (1.0f / ORC_PI)) : radius;
  f3 A = albedo, s;
  if (mode == 0)
    s = f3_make(burley_fitting(A.x), burley_fitting(A.y), burley_fitting(A.z));
  else
    s = f3_make(burley_fitting5(A.x), burley_fitting5(A.y), burley_fitting5(A.z));
  *radius_out = f3_div(l, s);
}
/* :74-104 */
static void bssrdf_setup(int burley_radius, int scale_mfp, int use_eq5, f3* weight, f3* albedo, f3* radius,
                         f3* diffuse_weight) {
  *diffuse_weight = f3_set1(0.0f);
  const float kMinRadius = 1e-8f;
  float kd[3] = {0, 0, 0}, w[3] = {weight->x, weight->y, weight->z}, r[3] = {radius->x, radius->y, radius->z};
  int channels = 3;
  for (int i = 0; i < 3; i++)
    if (r[i] < kMinRadius) {
      kd[i] = w[i];
      w[i] = 0.f;
      r[i] = 0.f;
      channels--;
    }
  *weight = f3_make(w[0], w[1], w[2]);
  *radius = f3_make(r[0], r[1], r[2]);
  if (channels < 3) *diffuse_weight = f3_make(kd[0], kd[1], kd[2]);
  if (channels > 0 && burley_radius) {
    f3 upd;
    bssrdf_burley_setup(*albedo, *radius, scale_mfp, use_eq5, &upd);
    *radius = upd;
  }
}
/* :111-122 */
static void scattering_from_albedo(float A, float d, float* sigma_t, float* sigma_s) {
  float a = 1.0f - orc_expf(A * (-5.09406f + A * (2.61188f - A * 4.31805f)));
  float s = 1.9f - A + 3.5f * orc_sqr(A - 0.8f);
  *sigma_t = 1.0f / orc_max(d * s, 1e-16f);
  *sigma_s = *sigma_t * a;
}
/* :141-172 */
static int sample_channel(f3 albedo, f3 throughput, float r, f3* pdf) {
  f3 w = f3_make(fabsf(throughput.x * albedo.x), fabsf(throughput.y * albedo.y), fabsf(throughput.z * albedo.z));
  float sum = w.x + w.y + w.z;
  if (sum > 0.0f)
    *pdf = f3_make(w.x / sum, w.y / sum, w.z / sum);
  else
    *pdf = f3_make(1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 3.0f);
  if (r < pdf->x) return 0;
  if (r < pdf->x + pdf->y) return 1;
  return 2;
}
/* :174-188 */
static float sample_scatter_distance(f3 throughput, f3 sigma_s, f3 sigma_t, float u0, float u1, f3* channel_pdf) {
  f3 albedo = orc_safe_divide_spectrum(sigma_s, sigma_t);
  int ch = sample_channel(albedo, throughput, u0, channel_pdf);
  return -orc_logf(1.0f - u1) / f3_get(sigma_t, ch);
}
/* :190-198 */
static f3 attenuate_transmission(f3 sigma_t, float distance) {
  return f3_make(orc_expf(-sigma_t.x * distance), orc_expf(-sigma_t.y * distance), orc_expf(-sigma_t.z * distance));
}

/* :227-405.  On success *si is the exit surface, *Rgl the exit frame. */
static int random_walk_subsurface(orc_ctx* c, f3 weight, f3 albedo, f3 radius, orc_rng* rng, orc_surface* si,
                                  orc_mat3* Rgl, f3* new_omega_out, f3* throughput_out) {
  if (si->face_direction != ORC_FRONT) return 0;
  orc_mat3 Rlg = transpose3(Rgl);
  f3 global_dir;
  {
    f3 tmp;
    float pdf = 0.0f;
    float u0 = orc_rng_draw(rng);
    float u1 = orc_rng_draw(rng);
    orc_lambert_sample(u0, u1, &tmp, &pdf);
    tmp = f3_neg(tmp);
    global_dir = mult_v(tmp, &Rlg);
    if (f3_dot(f3_neg(si->normal_g), global_dir) <= 0.0f) return 0;
  }
  f3 sigma_t, sigma_s, throughput;
  scattering_from_albedo(albedo.x, radius.x, &sigma_t.x, &sigma_s.x);
  scattering_from_albedo(albedo.y, radius.y, &sigma_t.y, &sigma_s.y);
  scattering_from_albedo(albedo.z, radius.z, &sigma_t.z, &sigma_s.z);
  throughput = orc_safe_divide_spectrum(weight, albedo);

  orc_rayf ray;
  ray.org = si->global_position;
  ray.dir = global_dir;
  ray.min_t = 1e-3f;
  ray.max_t = ORC_INF;

  const uint32_t max_bounces = 8192;
  orc_hit tr;
  memset(&tr, 0, sizeof(tr));
  int hit = 0;
  for (uint32_t bounce = 0; bounce <= max_bounces; ++bounce) {
    if (bounce > 0) {
      /* :296 UniformSampleSphere(rng.Draw(), rng.Draw()): g++ evaluates arguments right-to-left,
       * so the FIRST draw is u2 and the second u1. */
      float first = orc_rng_draw(rng);
      float second = orc_rng_draw(rng);
      f3 wi = f3_normalize(orc_uniform_sample_sphere(second, first));
      ray.dir = wi;
      ray.min_t = 0.f;
    }
    f3 channel_pdf;
    float d0 = orc_rng_draw(rng);
    float d1 = orc_rng_draw(rng);
    float t_scatter = sample_scatter_distance(throughput, sigma_s, sigma_t, d0, d1, &channel_pdf);
    ray.max_t = t_scatter;
    tr = trace_first_hit(c, &ray);
    c->stats.sss_steps++;
    hit = (tr.instance_id != ORC_NONE);
    float t = hit ? tr.t : t_scatter;
    f3 transmittance = attenuate_transmission(sigma_t, t);
    if (hit) {
      float pdf = f3_dot(channel_pdf, transmittance);
      throughput = f3_divs(f3_mul(throughput, transmittance), pdf);
      break;
    } else {
      float pdf = f3_dot(channel_pdf, f3_mul(sigma_t, transmittance));
      throughput = f3_divs(f3_mul(throughput, f3_mul(sigma_s, transmittance)), pdf);
    }
    {
      float p = orc_saturate(orc_spectrum_norm(throughput));
      float q = orc_rng_draw(rng);
      if (q >= p) break;
      throughput = f3_divs(throughput, p);
    }
    ray.org = f3_add(ray.org, f3_scale(ray.dir, t));
  }
  if (!hit) return 0;
  uint32_t prev_instance = si->instance_id;
  *si = trace_result_to_surface(c->scene, &ray, &tr);
  if (si->instance_id != prev_instance) return 0;
  if (si->face_direction != ORC_BACK) return 0;
  {
    f3 ez = si->normal_s, ex, ey;
    branchless_onb(ez, &ex, &ey);
    *Rgl = global_to_local(ex, ey, ez);
  }
  *new_omega_out = mult_v(ray.dir, Rgl);
  *throughput_out = throughput;
  return 1;
}

/* cycles-principled-shader.cc:169-242 */
static void sample_bsdf(orc_ctx* c, f3 omega_out, const orc_bsdf* b, orc_rng* rng, orc_surface* si, orc_mat3* Rgl,
                        f3* omega_in, f3* bsdf_f, f3* contribute, float* pdf) {
  *contribute = f3_set1(0.f);
  orc_sample_weight w = closure_sample_weight(omega_out, b);
  float select = orc_rng_draw(rng);
  if (select < w.diffuse) {
    float u0 = orc_rng_draw(rng);
    float u1 = orc_rng_draw(rng);
    float p = 0.f;
    orc_lambert_sample(u0, u1, omega_in, &p);
  } else if (select < w.diffuse + w.subsurface) {
    f3 new_omega_out, sss_thr;
    int ok = random_walk_subsurface(c, b->subsurface_weight, b->subsurface_albedo, b->subsurface_radius, rng, si, Rgl,
                                    &new_omega_out, &sss_thr);
    if (ok) {
      orc_bsdf nb;
      bsdf_default(&nb);
      nb.enable_diffuse = 1;
      nb.diffuse_weight = sss_thr;
      orc_eval ev = {0, &nb, NULL};
      *contribute = direct_illumination(c, new_omega_out, si, Rgl, si->normal_s, rng, &ev, 1);
      f3 dummy;
      sample_bsdf(c, new_omega_out, &nb, rng, si, Rgl, omega_in, bsdf_f, &dummy, pdf);
      return;
    }
    *omega_in = f3_set1(0.f);
    *bsdf_f = f3_set1(0.f);
    *pdf = 0.f;
    return;
  } else if (select < w.diffuse + w.subsurface + w.specular) {
    float u0 = orc_rng_draw(rng);
    float u1 = orc_rng_draw(rng);
    float p = 0.f;
    orc_ggx_sample(omega_out, b->alpha_x, b->alpha_y, u0, u1, 2, omega_in, &p);
  } else {
    float u0 = orc_rng_draw(rng);
    float u1 = orc_rng_draw(rng);
    float p = 0.f;
    orc_ggx_sample(omega_out, b->clearcoat_alpha_x, b->clearcoat_alpha_y, u0, u1, 1, omega_in, &p);
  }
  eval_bsdf(*omega_in, omega_out, b, bsdf_f, pdf);
}

/* cycles-principled-shader.cc:244-412; scene/uv may be NULL when the material has no texture */
static void param_to_bsdf(const orc_scene* scene, const float* uv, const orc_principled_param* mp, orc_bsdf* bsdf) {
  f3 weight = f3_set1(1.f);
  f3 base_color = f3_make(mp->base_color[0], mp->base_color[1], mp->base_color[2]);
  if (mp->base_color_tex_id != ORC_NONE) { /* :281-288 */
    float c[3];
    texture_fetch3(&scene->textures[mp->base_color_tex_id], uv[0], uv[1], c);
    base_color = f3_make(c[0], c[1], c[2]);
  }
  float subsurface = mp->subsurface;
  f3 subsurface_radius = f3_make(mp->subsurface_radius[0], mp->subsurface_radius[1], mp->subsurface_radius[2]);
  f3 subsurface_color = f3_make(mp->subsurface_color[0], mp->subsurface_color[1], mp->subsurface_color[2]);
  if (mp->subsurface_color_tex_id != ORC_NONE) { /* :292-301 */
    float c[3];
    texture_fetch3(&scene->textures[mp->subsurface_color_tex_id], uv[0], uv[1], c);
    subsurface_color = f3_make(c[0], c[1], c[2]);
  }
  const float cutoff = ORC_EPS;
  bsdf_default(bsdf);
  float diffuse_w = (1.0f - orc_saturate(mp->metallic)) * (1.0f - orc_saturate(mp->transmission));
  float final_transmission = orc_saturate(mp->transmission) * (1.0f - orc_saturate(mp->metallic));
  float specular_w = (1.0f - final_transmission);
  {
    f3 mixed = f3_add(f3_scale(subsurface_color, subsurface), f3_scale(base_color, 1.0f - subsurface));
    bsdf->enable_diffuse = 0;
    if (orc_average(mixed) > cutoff) {
      if (subsurface < cutoff && diffuse_w > cutoff) {
        bsdf->enable_diffuse = 1;
        bsdf->diffuse_weight = f3_scale(f3_mul(weight, base_color), diffuse_w);
      } else if (subsurface > cutoff) {
        bsdf->enable_subsurface = 1;
        bsdf->subsurface_weight = f3_scale(f3_mul(weight, mixed), diffuse_w);
        bsdf->subsurface_albedo = mixed;
        bsdf->subsurface_radius = f3_scale(subsurface_radius, subsurface);
        f3 add_diffuse = f3_set1(0.f);
        bssrdf_setup(1, 1, 1, &bsdf->subsurface_weight, &bsdf->subsurface_albedo, &bsdf->subsurface_radius, &add_diffuse);
        if (!orc_is_black(add_diffuse)) {
          bsdf->enable_diffuse = 1;
          bsdf->diffuse_weight = f3_add(bsdf->diffuse_weight, add_diffuse);
        }
      }
    }
  }
  bsdf->enable_specular = 0;
  if (specular_w > cutoff && (mp->specular > cutoff || mp->metallic > cutoff)) {
    bsdf->enable_specular = 1;
    bsdf->specular_weight = f3_scale(weight, specular_w);
    bsdf->ior = (2.0f / (1.0f - orc_safe_sqrtf(0.08f * mp->specular))) - 1.0f;
    float aspect = orc_safe_sqrtf(1.0f - mp->anisotropic * 0.9f);
    float r2 = mp->roughness * mp->roughness;
    bsdf->alpha_x = r2 / aspect;
    bsdf->alpha_y = r2 * aspect;
    float y = orc_rgb_to_y(base_color);
    f3 rho_tint = y > 0.0f ? f3_divs(base_color, y) : f3_set1(0.0f);
    f3 rho_specular = f3_lerp(f3_set1(1.0f), rho_tint, mp->specular_tint);
    bsdf->specular_color = f3_lerp(f3_scale(rho_specular, 0.08f * mp->specular), base_color, mp->metallic);
  }
  bsdf->enable_clearcoat = 0;
  if (mp->clearcoat > cutoff) {
    bsdf->enable_clearcoat = 1;
    bsdf->clearcoat_weight = f3_set1(0.25f * mp->clearcoat);
    bsdf->clearcoat_alpha_x = mp->clearcoat_roughness * mp->clearcoat_roughness;
    bsdf->clearcoat_alpha_y = mp->clearcoat_roughness * mp->clearcoat_roughness;
    bsdf->clearcoat_color = f3_set1(0.04f);
    bsdf->clearcoat_ior = 1.5f;
  }
}

/* cycles-principled-shader.cc:414-484 */
static void principled_shader(orc_ctx* c, f3 global_omega_out, orc_rng* rng, orc_surface* si, f3* global_omega_in,
                              f3* throughput, f3* contribute, float* pdf) {
  if (si->face_direction == ORC_AMBIGUOUS) {
    *global_omega_in = global_omega_out;
    *throughput = f3_set1(0.0f);
    *contribute = f3_set1(0.0f);
    *pdf = 0.0f;
    return;
  }
  f3 ez = (si->face_direction == ORC_FRONT) ? si->normal_s : f3_neg(si->normal_s);
  f3 ex, ey;
  branchless_onb(ez, &ex, &ey);
  orc_mat3 Rgl = global_to_local(ex, ey, ez);
  f3 omega_out = mult_v(global_omega_out, &Rgl);
  orc_bsdf bsdf;
  param_to_bsdf(c->scene, si->texcoord, &si->material->pr, &bsdf);
  *contribute = f3_set1(0.f);
  {
    orc_eval ev = {0, &bsdf, NULL};
    f3 d = direct_illumination(c, omega_out, si, &Rgl, ez, rng, &ev, 1);
    *contribute = f3_add(*contribute, d);
  }
  f3 omega_in = f3_set1(0.f), bsdf_f = f3_set1(0.f), contrib2 = f3_set1(0.f);
  float ret_pdf = 0.f;
  sample_bsdf(c, omega_out, &bsdf, rng, si, &Rgl, &omega_in, &bsdf_f, &contrib2, &ret_pdf);
  *contribute = f3_add(*contribute, contrib2);
  /* :467-469 rebuilds local->global from the ENTRY frame (ex,ey,ez), also after an SSS exit */
  orc_mat3 Rlg = local_to_global(ex, ey, ez);
  *global_omega_in = mult_v(omega_in, &Rlg);
  float cos_i = fabsf(omega_in.z);
  *throughput = f3_divs(f3_scale(bsdf_f, cos_i), ret_pdf);
  *pdf = ret_pdf;
  if (!orc_is_finite3(*throughput) || !isfinite(*pdf)) {
    *throughput = f3_set1(0.f);
    *pdf = 0.f;
  }
}

/* ------------------------------------------------------------ hair shader (hair-shader.cc) */
static float pow_n(float v, int n) { /* pbrlab_math.h:40-55 Pow<n> */
  if (n == 0) return 1.f;
  if (n == 1) return v;
  float h = pow_n(v, n / 2);
  return h * h * pow_n(v, n & 1);
}
/* :19-64 */
static void beta_m_to_v(float beta_m, float v[4]) {
  v[0] = orc_sqr(0.726f * beta_m + 0.812f * orc_sqr(beta_m) + 3.7f * pow_n(beta_m, 20));
  v[1] = 0.25f * v[0];
  v[2] = 4.0f * v[0];
  v[3] = v[2];
}
static float calc_s(float beta_n) {
  float b2 = orc_sqr(beta_n);
  return sqrtf(ORC_PI / 8.0f) * (0.265f * beta_n + 1.194f * b2 + 5.372f * pow_n(b2, 11));
}
static f3 sigma_a_from_rgb(f3 c, float beta_n) {
  float r[3], cc[3] = {c.x, c.y, c.z};
  for (int i = 0; i < 3; i++)
    r[i] = orc_sqr(orc_fast_log(cc[i]) / (5.969f - 0.215f * beta_n + 2.532f * orc_sqr(beta_n) - 10.73f * pow_n(beta_n, 3) +
                                          5.574f * pow_n(beta_n, 4) + 0.245f * pow_n(beta_n, 5)));
  return f3_make(r[0], r[1], r[2]);
}
static f3 sigma_a_from_melanin(float melanin, float redness) {
  const float random_value = 0.5f;
  float factor = 1.f + 2.f * (random_value - 0.5f);
  melanin = orc_clamp(melanin, 0.0f, 1.0f) * factor;
  redness = orc_clamp(redness, 0.0f, 1.0f);
  melanin = -orc_fast_log(orc_max(1.0f - melanin, 0.0001f));
  float eu = melanin * (1.0f - redness);
  float pheo = melanin * redness;
  return f3_make(orc_max(0.0f, eu * 0.506f + pheo * 0.343f), orc_max(0.0f, eu * 0.841f + pheo * 0.733f),
                 orc_max(0.0f, eu * 1.653f + pheo * 1.924f));
}
/* :100-151 */
static void hair_param_to_bsdf(const orc_hair_param* mp, float geom_v, orc_hair_bsdf* b) {
  if (mp->coloring_hair == 0)
    b->sigma_a = sigma_a_from_rgb(f3_make(mp->base_color[0], mp->base_color[1], mp->base_color[2]), mp->azimuthal_roughness);
  else
    b->sigma_a = sigma_a_from_melanin(mp->melanin, mp->melanin_redness);
  b->h = geom_v;
  beta_m_to_v(mp->roughness, b->v);
  b->s = calc_s(mp->azimuthal_roughness);
  b->eta = mp->ior;
  b->alpha = mp->shift * ORC_PI / 180.f;
  b->tints[0] = f3_make(mp->specular_tint[0], mp->specular_tint[1], mp->specular_tint[2]);
  b->tints[1] = f3_make(mp->transmission_tint[0], mp->transmission_tint[1], mp->transmission_tint[2]);
  b->tints[2] = f3_make(mp->second_specular_tint[0], mp->second_specular_tint[1], mp->second_specular_tint[2]);
  b->tints[3] = f3_set1(1.f);
  b->transparent_scale = 1.f;
}
/* :153-229 */
static void hair_shader(orc_ctx* c, f3 global_omega_out, orc_rng* rng, orc_surface* si, f3* global_omega_in,
                        f3* throughput, f3* contribute, float* pdf) {
  if (si->face_direction == ORC_AMBIGUOUS) {
    *global_omega_in = global_omega_out;
    *throughput = f3_set1(0.0f);
    *contribute = f3_set1(0.0f);
    *pdf = 0.0f;
    return;
  }
  f3 ex = si->normal_s;
  f3 ey = f3_normalize(f3_cross(f3_cross(global_omega_out, ex), ex));
  f3 ez = f3_cross(ex, ey);
  orc_mat3 Rgl = global_to_local(ex, ey, ez);
  f3 omega_out = mult_v(global_omega_out, &Rgl);
  orc_hair_bsdf hb;
  hair_param_to_bsdf(&si->material->hr, si->v, &hb);
  *contribute = f3_set1(0.f);
  {
    orc_eval ev = {1, NULL, &hb};
    f3 d = direct_illumination(c, omega_out, si, &Rgl, ex, rng, &ev, 0);
    *contribute = f3_add(*contribute, d);
  }
  f3 omega_in = f3_set1(0.f), fcos;
  float ret_pdf = 0.f;
  {
    float us[4];
    us[0] = orc_rng_draw(rng), us[1] = orc_rng_draw(rng), us[2] = orc_rng_draw(rng), us[3] = orc_rng_draw(rng);
    fcos = orc_hair_sample(omega_out, &hb, us, &omega_in, &ret_pdf);
  }
  orc_mat3 Rlg = local_to_global(ex, ey, ez);
  *global_omega_in = mult_v(omega_in, &Rlg);
  *throughput = f3_divs(fcos, ret_pdf);
  *pdf = ret_pdf;
  if (!orc_is_finite3(*throughput) || !isfinite(*pdf)) {
    *throughput = f3_set1(0.f);
    *pdf = 0.f;
  }
}

/* shader.cc:8-36 */
static void shader(orc_ctx* c, f3 global_omega_out, orc_rng* rng, orc_surface* si, f3* global_omega_in, f3* throughput,
                   f3* contribute, float* pdf) {
  if (si->material == NULL) {
    *global_omega_in = global_omega_out;
    *throughput = f3_set1(0.0f);
    *contribute = f3_set1(0.0f);
    *pdf = 0.0f;
    return;
  }
  if (si->material->kind == 0)
    principled_shader(c, global_omega_out, rng, si, global_omega_in, throughput, contribute, pdf);
  else
    hair_shader(c, global_omega_out, rng, si, global_omega_in, throughput, contribute, pdf);
}

/* ============================================================== integrator (render.cc:24-90) */
static f3 get_radiance(orc_ctx* c, const orc_rayf* input_ray, orc_rng* rng) {
  orc_rayf ray = *input_ray;
  f3 contribution = f3_set1(0.0f);
  f3 throughput = f3_set1(1.0f);
  float bsdf_sampling_pdf = 0.f;
  for (uint32_t depth = 0;; depth++) {
    if (orc_is_black(throughput)) break;
    orc_hit tr = trace_first_hit(c, &ray);
    if (tr.instance_id == ORC_NONE) break;
    orc_surface si = trace_result_to_surface(c->scene, &ray, &tr);
    if (si.face_direction == ORC_FRONT) {
      f3 emission = f3_set1(0.f);
      float pdf_area = 0.f;
      if (implicit_area_light(c->scene, tr.instance_id, tr.geom_id, tr.prim_id, &emission, &pdf_area)) {
        float a2s = fabsf((tr.t * tr.t) / f3_dot(si.normal_s, ray.dir));
        float weight = (depth == 0) ? 1.0f : orc_power_heuristic(bsdf_sampling_pdf, pdf_area * a2s);
        contribution = f3_add(contribution, f3_mul(f3_scale(emission, weight), throughput));
      }
    }
    float rr = orc_spectrum_norm(throughput);
    if (rr < orc_rng_draw(rng)) break;
    throughput = f3_mul(throughput, f3_set1(1.0f / rr));
    c->stats.bounces++;
    f3 next_dir, r_thr, d_contrib;
    float pdf;
    shader(c, f3_neg(ray.dir), rng, &si, &next_dir, &r_thr, &d_contrib, &pdf);
    contribution = f3_add(contribution, f3_mul(throughput, d_contrib));
    throughput = f3_mul(r_thr, throughput);
    bsdf_sampling_pdf = pdf;
    ray.org = si.global_position;
    ray.dir = next_dir;
    ray.min_t = 1e-3f;
    ray.max_t = ORC_INF;
  }
  return contribution;
}

/* camera of RenderingTile (render.cc:132-158) */
typedef struct {
  f3 org;
  float x_corner, y_corner, z_corner, dx, dy;
} orc_camera;

static orc_camera make_camera(const orc_scene* s, uint32_t width, uint32_t height) {
  const float *bmin = s->bmin, *bmax = s->bmax;
  float hs, vs;
  if (bmax[0] - bmin[0] > bmax[1] - bmin[1]) {
    hs = bmax[0] - bmin[0];
    vs = hs * (float)height / (float)width;
  } else {
    vs = bmax[1] - bmin[1];
    hs = vs * (float)width / (float)height;
  }
  orc_camera cam;
  cam.org = f3_make((bmax[0] + bmin[0]) * 0.5f, (bmax[1] + bmin[1]) * 0.5f, bmax[2] + hs * 0.5f * sqrtf(3.f));
  cam.x_corner = (bmax[0] + bmin[0]) * 0.5f - hs * 0.5f;
  cam.y_corner = (bmax[1] + bmin[1]) * 0.5f + vs * 0.5f;
  cam.z_corner = bmax[2];
  cam.dx = hs / (float)width;
  cam.dy = vs / (float)height;
  return cam;
}
/* render.cc:160-171 : two draws, x first */
static orc_rayf camera_ray(const orc_camera* cam, uint32_t x, uint32_t y, orc_rng* rng) {
  float jx = orc_rng_draw(rng);
  float jy = orc_rng_draw(rng);
  f3 target = f3_make(cam->x_corner + cam->dx * ((float)x + jx), cam->y_corner - cam->dy * ((float)y + jy), cam->z_corner);
  orc_rayf r;
  r.dir = f3_normalize_raw(f3_sub(target, cam->org));
  r.org = cam->org;
  r.min_t = 0.0f;
  r.max_t = ORC_INF;
  return r;
}

static void seed_sample(orc_rng* rng, uint32_t width, uint32_t x, uint32_t y, uint32_t pass, uint64_t seed_seq) {
  orc_rng_seed(rng, ((uint64_t)pass << 32) + ((uint64_t)y * width + x), seed_seq);
}

void orc_camera_ray(const orc_scene* s, uint32_t width, uint32_t height, uint32_t x, uint32_t y, uint32_t pass,
                    uint64_t seed_seq, orc_ray* out) {
  orc_camera cam = make_camera(s, width, height);
  orc_rng rng;
  seed_sample(&rng, width, x, y, pass, seed_seq);
  orc_rayf r = camera_ray(&cam, x, y, &rng);
  out->org[0] = r.org.x, out->org[1] = r.org.y, out->org[2] = r.org.z, out->tmin = r.min_t;
  out->dir[0] = r.dir.x, out->dir[1] = r.dir.y, out->dir[2] = r.dir.z, out->tmax = r.max_t;
}

uint32_t orc_sample_trace(const orc_scene* s, uint32_t width, uint32_t height, uint32_t x, uint32_t y, uint32_t pass,
                          uint64_t seed_seq, float radiance[3], uint64_t* draws, orc_hit* hits, uint32_t max_hits) {
  orc_ctx c;
  memset(&c, 0, sizeof(c));
  c.scene = s;
  c.trace_hits = hits;
  c.trace_cap = max_hits;
  orc_camera cam = make_camera(s, width, height);
  orc_rng rng;
  seed_sample(&rng, width, x, y, pass, seed_seq);
  orc_rayf r = camera_ray(&cam, x, y, &rng);
  f3 L = get_radiance(&c, &r, &rng);
  radiance[0] = L.x, radiance[1] = L.y, radiance[2] = L.z;
  if (draws) *draws = rng.draws;
  return c.trace_n;
}

/* render-tile.cc:29-41 */
void orc_create_tiles(uint32_t width, uint32_t height, uint32_t* out, uint32_t* num_tiles) {
  uint32_t n = 0;
  for (uint32_t i = 0; i < height; i += 64)
    for (uint32_t j = 0; j < width; j += 64) {
      if (out) {
        out[n * 4 + 0] = j;
        out[n * 4 + 1] = (j + 64 < width) ? j + 64 : width;
        out[n * 4 + 2] = i;
        out[n * 4 + 3] = (i + 64 < height) ? i + 64 : height;
      }
      n++;
    }
  *num_tiles = n;
}

typedef struct {
  const orc_scene* scene;
  uint32_t width, height, spp, first_pass, tile_rank, tile_world;
  uint64_t seed_seq;
  float* rgba;
  uint32_t* count;
  uint32_t* tiles;
  uint32_t ntiles;
  uint32_t job_mode;
  volatile uint64_t next_job;
  pthread_mutex_t mtx;
  pthread_mutex_t* tile_mtx; /* ORC_JOBS_TILE_PASS: RenderTile::mtx (render.cc:176) */
  orc_stats total;
} orc_job;

static void stats_add(orc_stats* a, const orc_stats* b, const orc_trav_stats* t) {
  a->samples += b->samples, a->closest_rays += b->closest_rays, a->shadow_rays += b->shadow_rays;
  a->bounces += b->bounces, a->sss_steps += b->sss_steps, a->rng_draws += b->rng_draws;
  a->nodes_visited += t->nodes, a->tris_tested += t->tris, a->curves_tested += t->curves;
}

/* the samples of passes [p0, p1) of the pixels [x0, x1) x [y0, y1): render.cc:160-183 */
static void render_rect(orc_job* job, orc_ctx* c, const orc_camera* cam, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1,
                        uint32_t p0, uint32_t p1, pthread_mutex_t* mtx) {
  for (uint32_t pass = p0; pass < p1; pass++)
    for (uint32_t y = y0; y < y1; y++)
      for (uint32_t x = x0; x < x1; x++) {
        orc_rng rng;
        seed_sample(&rng, job->width, x, y, pass, job->seed_seq);
        orc_rayf r = camera_ray(cam, x, y, &rng);
        f3 L = get_radiance(c, &r, &rng);
        size_t p = (size_t)y * job->width + x;
        /* render.cc:175-183 */
        if (mtx) pthread_mutex_lock(mtx);
        job->rgba[p * 4 + 0] += L.x;
        job->rgba[p * 4 + 1] += L.y;
        job->rgba[p * 4 + 2] += L.z;
        job->rgba[p * 4 + 3] += 1.0f;
        job->count[p]++;
        if (mtx) pthread_mutex_unlock(mtx);
        c->stats.samples++;
        c->stats.rng_draws += rng.draws;
      }
}

/* One worker of the pool (render.cc:210-233: every thread pulls job ids from one atomic counter).
 * ORC_JOBS_BLOCKS (the checker's mode): a job = one 16x16 block of a tile x ALL passes, so that each pixel's passes are
 *   accumulated in ascending order by a single thread (Q13: the image does not depend on the schedule or the thread count);
 *   blocks instead of whole tiles because the heaviest tile of a frame costs ~10x the mean and 510 tiles cannot keep 256
 *   threads busy (parallel efficiency 0.20).
 * ORC_JOBS_TILE_PASS (timing mode, the reference's own granularity): job id -> (tile = id % ntiles, pass = id / ntiles),
 *   pixels accumulated under the tile's mutex as in render.cc:175-183; the float sums then depend on the schedule, exactly
 *   like the reference's, so this mode is for the CPU baseline's clock, not for parity. */
static double g_last_busy = 0.0; /* sum over the workers of the last orc_render_jobs call: seconds from its start to the end of the worker's last job */
static double now_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
double orc_last_render_busy(void) { return g_last_busy; }

static void* render_worker(void* arg) {
  orc_job* job = (orc_job*)arg;
  orc_ctx c;
  memset(&c, 0, sizeof(c));
  c.scene = job->scene;
  const double t_start = now_seconds();
  double t_last = t_start;
  orc_camera cam = make_camera(job->scene, job->width, job->height);
  const uint64_t njobs = job->job_mode == ORC_JOBS_TILE_PASS ? (uint64_t)job->ntiles * job->spp : (uint64_t)job->ntiles * 16u;
  for (;;) {
    const uint64_t id = __sync_fetch_and_add(&job->next_job, 1ull);
    if (id >= njobs) break;
    if (job->job_mode == ORC_JOBS_TILE_PASS) {
      const uint32_t tile = (uint32_t)(id % job->ntiles), pass = job->first_pass + (uint32_t)(id / job->ntiles);
      if (tile % job->tile_world != job->tile_rank) continue;
      const uint32_t* tl = job->tiles + tile * 4;
      render_rect(job, &c, &cam, tl[0], tl[1], tl[2], tl[3], pass, pass + 1u, &job->tile_mtx[tile]);
    } else {
      const uint32_t tile = (uint32_t)(id / 16u), sub = (uint32_t)(id % 16u);
      if (tile % job->tile_world != job->tile_rank) continue;
      const uint32_t* tl = job->tiles + tile * 4;
      const uint32_t x0 = tl[0] + 16u * (sub % 4u), y0 = tl[2] + 16u * (sub / 4u);
      if (x0 >= tl[1] || y0 >= tl[3]) continue;
      render_rect(job, &c, &cam, x0, x0 + 16u < tl[1] ? x0 + 16u : tl[1], y0, y0 + 16u < tl[3] ? y0 + 16u : tl[3], job->first_pass,
                  job->first_pass + job->spp, NULL);
    }
    t_last = now_seconds();
  }
  pthread_mutex_lock(&job->mtx);
  g_last_busy += t_last - t_start;
  stats_add(&job->total, &c.stats, &c.trav);
  pthread_mutex_unlock(&job->mtx);
  return NULL;
}

void orc_render(const orc_scene* s, uint32_t width, uint32_t height, uint32_t spp, uint32_t first_pass, uint64_t seed_seq,
                uint32_t tile_rank, uint32_t tile_world, uint32_t num_threads, float* rgba, uint32_t* count,
                orc_stats* stats) {
  orc_render_jobs(s, width, height, spp, first_pass, seed_seq, tile_rank, tile_world, num_threads, ORC_JOBS_BLOCKS, rgba, count, stats);
}

void orc_render_jobs(const orc_scene* s, uint32_t width, uint32_t height, uint32_t spp, uint32_t first_pass, uint64_t seed_seq,
                     uint32_t tile_rank, uint32_t tile_world, uint32_t num_threads, uint32_t job_mode, float* rgba,
                     uint32_t* count, orc_stats* stats) {
  orc_job job;
  memset(&job, 0, sizeof(job));
  g_last_busy = 0.0;
  job.job_mode = job_mode;
  job.scene = s, job.width = width, job.height = height, job.spp = spp, job.first_pass = first_pass;
  job.tile_rank = tile_rank, job.tile_world = tile_world ? tile_world : 1, job.seed_seq = seed_seq;
  job.rgba = rgba, job.count = count;
  /* PrepareRendering: layer->Resize + Clear (render.cc:99-100) */
  memset(rgba, 0, sizeof(float) * 4 * (size_t)width * height);
  memset(count, 0, sizeof(uint32_t) * (size_t)width * height);
  orc_create_tiles(width, height, NULL, &job.ntiles);
  job.tiles = (uint32_t*)xrealloc(NULL, sizeof(uint32_t) * 4 * job.ntiles);
  orc_create_tiles(width, height, job.tiles, &job.ntiles);
  pthread_mutex_init(&job.mtx, NULL);
  if (job_mode == ORC_JOBS_TILE_PASS) {
    job.tile_mtx = (pthread_mutex_t*)xrealloc(NULL, sizeof(pthread_mutex_t) * job.ntiles);
    for (uint32_t i = 0; i < job.ntiles; i++) pthread_mutex_init(&job.tile_mtx[i], NULL);
  }
  if (num_threads < 1) num_threads = 1;
  if (num_threads == 1) {
    render_worker(&job);
  } else {
    pthread_t* th = (pthread_t*)xrealloc(NULL, sizeof(pthread_t) * num_threads);
    for (uint32_t i = 0; i < num_threads; i++) pthread_create(&th[i], NULL, render_worker, &job);
    for (uint32_t i = 0; i < num_threads; i++) pthread_join(th[i], NULL);
    free(th);
  }
  pthread_mutex_destroy(&job.mtx);
  if (job.tile_mtx) {
    for (uint32_t i = 0; i < job.ntiles; i++) pthread_mutex_destroy(&job.tile_mtx[i]);
    free(job.tile_mtx);
  }
  free(job.tiles);
  if (stats) *stats = job.total;
}

/* ================================================================= known-answer hooks */
void orc_kat_rng(uint64_t initstate, uint64_t initseq, uint32_t n, float* out) {
  orc_rng r;
  orc_rng_seed(&r, initstate, initseq);
  for (uint32_t i = 0; i < n; i++) out[i] = orc_rng_draw(&r);
}
float orc_kat_fastmath(int op, float x, float y2) {
  float s, c;
  switch (op) {
    case 0: return orc_fast_sin(x);
    case 1: return orc_fast_cos(x);
    case 2: return orc_fast_exp(x);
    case 3: return orc_fast_log(x);
    case 4: return orc_fast_atan2(x, y2);
    case 5: return orc_fast_asin(x);
    case 6: return orc_fast_exp2(x);
    case 7: return orc_fast_log2(x);
    case 8: orc_fast_sincos(x, &s, &c); return s;
    case 9: orc_fast_sincos(x, &s, &c); return c;
  }
  return 0.f;
}
/* the shared f64r functions over an array (tests/test_f64r.py): op 0 sin, 1 cos, 2 exp, 3 log */
void orc_kat_f64r(int op, const float* x, size_t n, float* out) {
  for (size_t i = 0; i < n; i++)
    out[i] = op == 0 ? f64r_sinf(x[i]) : (op == 1 ? f64r_cosf(x[i]) : (op == 2 ? f64r_expf(x[i]) : f64r_logf(x[i])));
}
float orc_kat_fresnel(float c, float eta) { return orc_fresnel_dielectric_cos(c, eta); }
float orc_kat_power_heuristic(float a, float b) { return orc_power_heuristic(a, b); }
void orc_kat_lambert_sample(float u0, float u1, float out[5]) {
  f3 wi;
  float pdf;
  float f = orc_lambert_sample(u0, u1, &wi, &pdf);
  out[0] = wi.x, out[1] = wi.y, out[2] = wi.z, out[3] = f, out[4] = pdf;
}
void orc_kat_ggx_eval(const float wi[3], const float wo[3], float ax, float ay, int distrib, float out[2]) {
  float pdf = 0.f;
  out[0] = orc_ggx_bsdf_pdf(f3_make(wi[0], wi[1], wi[2]), f3_make(wo[0], wo[1], wo[2]), ax, ay, distrib, &pdf);
  out[1] = pdf;
}
void orc_kat_ggx_sample(const float wo[3], float ax, float ay, float u0, float u1, int distrib, float out[5]) {
  f3 wi = f3_set1(0.f);
  float pdf = 0.f;
  float f = orc_ggx_sample(f3_make(wo[0], wo[1], wo[2]), ax, ay, u0, u1, distrib, &wi, &pdf);
  out[0] = wi.x, out[1] = wi.y, out[2] = wi.z, out[3] = f, out[4] = pdf;
}
static void hair_from_params(const float* p, orc_hair_bsdf* b) {
  b->h = p[0];
  for (int i = 0; i < 4; i++) b->v[i] = p[1 + i];
  b->s = p[5];
  b->sigma_a = f3_make(p[6], p[7], p[8]);
  b->eta = p[9];
  b->alpha = p[10];
  for (int i = 0; i < 4; i++) b->tints[i] = f3_make(p[11 + i * 3], p[12 + i * 3], p[13 + i * 3]);
  b->transparent_scale = p[23 - 1];
}
static void hair_to_params(const orc_hair_bsdf* b, float* p) {
  p[0] = b->h;
  for (int i = 0; i < 4; i++) p[1 + i] = b->v[i];
  p[5] = b->s;
  p[6] = b->sigma_a.x, p[7] = b->sigma_a.y, p[8] = b->sigma_a.z;
  p[9] = b->eta;
  p[10] = b->alpha;
  for (int i = 0; i < 4; i++) p[11 + i * 3] = b->tints[i].x, p[12 + i * 3] = b->tints[i].y, p[13 + i * 3] = b->tints[i].z;
  p[22] = b->transparent_scale;
}
void orc_kat_hair_eval(const float wi[3], const float wo[3], const float* params, float out[4]) {
  orc_hair_bsdf b;
  hair_from_params(params, &b);
  float pdf = 0.f;
  f3 f = orc_hair_eval(f3_make(wi[0], wi[1], wi[2]), f3_make(wo[0], wo[1], wo[2]), &b, &pdf);
  out[0] = f.x, out[1] = f.y, out[2] = f.z, out[3] = pdf;
}
void orc_kat_hair_sample(const float wo[3], const float* params, const float us[4], float out[7]) {
  orc_hair_bsdf b;
  hair_from_params(params, &b);
  f3 wi = f3_set1(0.f);
  float pdf = 0.f;
  f3 f = orc_hair_sample(f3_make(wo[0], wo[1], wo[2]), &b, us, &wi, &pdf);
  out[0] = wi.x, out[1] = wi.y, out[2] = wi.z, out[3] = f.x, out[4] = f.y, out[5] = f.z, out[6] = pdf;
}
void orc_kat_uniform_sphere(float u1, float u2, float out[3]) {
  f3 v = orc_uniform_sample_sphere(u1, u2);
  out[0] = v.x, out[1] = v.y, out[2] = v.z;
}
void orc_kat_triangle_sampler(float u1, float u2, float out[2]) { orc_triangle_uniform_sampler(u1, u2, &out[0], &out[1]); }

/* out: [0] en_diffuse [1..3] diffuse_w [4] en_sss [5..7] sss_w [8..10] albedo [11..13] radius
 *      [14] en_spec [15..17] spec_w [18] ax [19] ay [20] ior [21..23] spec_color
 *      [24] en_coat [25..27] coat_w [28] cax [29] cay [30] cior [31..33] coat_color */
void orc_kat_param_to_bsdf(const orc_principled_param* p, float out[34]) {
  orc_bsdf b;
  param_to_bsdf(NULL, NULL, p, &b);
  out[0] = (float)b.enable_diffuse, out[1] = b.diffuse_weight.x, out[2] = b.diffuse_weight.y, out[3] = b.diffuse_weight.z;
  out[4] = (float)b.enable_subsurface;
  out[5] = b.subsurface_weight.x, out[6] = b.subsurface_weight.y, out[7] = b.subsurface_weight.z;
  out[8] = b.subsurface_albedo.x, out[9] = b.subsurface_albedo.y, out[10] = b.subsurface_albedo.z;
  out[11] = b.subsurface_radius.x, out[12] = b.subsurface_radius.y, out[13] = b.subsurface_radius.z;
  out[14] = (float)b.enable_specular;
  out[15] = b.specular_weight.x, out[16] = b.specular_weight.y, out[17] = b.specular_weight.z;
  out[18] = b.alpha_x, out[19] = b.alpha_y, out[20] = b.ior;
  out[21] = b.specular_color.x, out[22] = b.specular_color.y, out[23] = b.specular_color.z;
  out[24] = (float)b.enable_clearcoat;
  out[25] = b.clearcoat_weight.x, out[26] = b.clearcoat_weight.y, out[27] = b.clearcoat_weight.z;
  out[28] = b.clearcoat_alpha_x, out[29] = b.clearcoat_alpha_y, out[30] = b.clearcoat_ior;
  out[31] = b.clearcoat_color.x, out[32] = b.clearcoat_color.y, out[33] = b.clearcoat_color.z;
}
void orc_kat_hair_param_to_bsdf(const orc_hair_param* p, float h, float out[23]) {
  orc_hair_bsdf b;
  hair_param_to_bsdf(p, h, &b);
  hair_to_params(&b, out);
}
uint32_t orc_light_table(const orc_scene* s, uint32_t li, uint32_t* instance_id, uint32_t* geom_id, float* choose_prob,
                         float* cdf, uint32_t* num_prims) {
  if (li < s->nlights) {
    const orc_light* L = &s->lights[li];
    *instance_id = L->instance_id, *geom_id = L->geom_id, *choose_prob = L->choose_prob, *cdf = s->light_cdf[li];
    *num_prims = s->instances[L->instance_id].area_lights[L->geom_id]->nprim;
  }
  return s->nlights;
}
void orc_light_prims(const orc_scene* s, uint32_t li, float* prim_prob, float* prim_cdf, float* prim_area_pdf) {
  const orc_light* L = &s->lights[li];
  const orc_area_light* a = s->instances[L->instance_id].area_lights[L->geom_id];
  memcpy(prim_prob, a->choose_prob, sizeof(float) * a->nprim);
  memcpy(prim_cdf, a->cdf, sizeof(float) * a->nprim);
  memcpy(prim_area_pdf, a->area_pdf, sizeof(float) * a->nprim);
}

/* src/curve-util.cc:7-199 : Catmull-Rom (tau = 0.5) -> cubic Bezier, root / in-between / end */
int orc_to_cubic_bezier(const float* cvs, const float* radii, uint32_t n, float* out) {
  if (n < 3) return -1;
  const float tau = 0.5f, tau3 = tau / 3.0f;
  uint32_t nseg = n - 1, k = 0;
#define CV(i, c) cvs[3 * (i) + (c)]
  for (int c = 0; c < 4; c++) { /* root: P0,P1,P2 = cv 0,1,2 */
    float p0 = c < 3 ? CV(0, c) : radii[0], p1 = c < 3 ? CV(1, c) : radii[1], p2 = c < 3 ? CV(2, c) : radii[2];
    out[k * 16 + 0 * 4 + c] = p0;
    out[k * 16 + 1 * 4 + c] = ((tau + 1.0f) / 3.0f) * p0 + (2.0f / 3.0f) * p1 - tau3 * p2;
    out[k * 16 + 2 * 4 + c] = tau3 * (p0 - p2) + p1;
    out[k * 16 + 3 * 4 + c] = p1;
  }
  k++;
  for (uint32_t s = 1; s + 1 < nseg; s++, k++) {
    uint32_t b = s - 1;
    for (int c = 0; c < 4; c++) {
      float p0 = c < 3 ? CV(b, c) : radii[b], p1 = c < 3 ? CV(b + 1, c) : radii[b + 1];
      float p2 = c < 3 ? CV(b + 2, c) : radii[b + 2], p3 = c < 3 ? CV(b + 3, c) : radii[b + 3];
      out[k * 16 + 0 * 4 + c] = p1;
      out[k * 16 + 1 * 4 + c] = tau3 * (p2 - p0) + p1;
      out[k * 16 + 2 * 4 + c] = tau3 * (p1 - p3) + p2;
      out[k * 16 + 3 * 4 + c] = p2;
    }
  }
  if (nseg > 1) {
    uint32_t b = nseg - 2;
    for (int c = 0; c < 4; c++) {
      float p0 = c < 3 ? CV(b, c) : radii[b], p1 = c < 3 ? CV(b + 1, c) : radii[b + 1], p2 = c < 3 ? CV(b + 2, c) : radii[b + 2];
      out[k * 16 + 0 * 4 + c] = p1;
      out[k * 16 + 1 * 4 + c] = tau3 * (p2 - p0) + p1;
      out[k * 16 + 2 * 4 + c] = (-tau3) * p0 + (2.0f / 3.0f) * p1 + ((tau + 1.0f) / 3.0f) * p2;
      out[k * 16 + 3 * 4 + c] = p2;
    }
    k++;
  }
#undef CV
  return (int)k;
}
