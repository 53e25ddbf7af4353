/*
 * pbr_oracle.h -- ORACLE C API (test infrastructure only).
 *
 * CPU restatement of pbrlab's per-pixel path-tracing integrator (src/render.cc GetRadiance loop,
 * src/shader, src/closure, src/light-manager, src/scene.cc read side) with its own BVH behind
 * the Raytracer facade (src/raytracer/raytracer.h:27-114; the reference's Embree back end is absent,
 * SURVEY.md F3/F5).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (pbrlab_amd/, include/pbrhip.h) never does.
 *
 * PARITY PIN STATUS: leaf closures / RNG / fast-math / sampling are pinned against the reference's
 * own headers compiled unmodified (oracle/ref_harness.cc -> oracle/_ref/) and against the
 * known-answer values recorded in SURVEY.md Appendix A.  The shader- and integrator-level logic
 * cannot be compiled from the reference without stand-ins for mpark/variant.hpp and Embree, so at
 * that level this oracle is a restatement by reading: **parity unpinned** there.
 */
#ifndef PBR_ORACLE_H_
#define PBR_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* src/material-param.h:24-49 (name dropped).  Layout == pbrhip_principled_param. */
typedef struct {
  float base_color[3];
  float subsurface;
  float subsurface_radius[3];
  float subsurface_color[3];
  float metallic, specular, specular_tint, roughness, anisotropic, anisotropic_rotation;
  float sheen, sheen_tint, clearcoat, clearcoat_roughness, ior, transmission, transmission_roughness;
  uint32_t base_color_tex_id, subsurface_color_tex_id;
} orc_principled_param;

/* src/material-param.h:51-72.  Layout == pbrhip_hair_param. */
typedef struct {
  uint32_t coloring_hair; /* 0 = kRGB, 1 = kMelanin */
  float base_color[3];
  float melanin, melanin_redness, melanin_randomize;
  float roughness, azimuthal_roughness, ior, shift;
  float specular_tint[3], second_specular_tint[3], transmission_tint[3];
} orc_hair_param;

typedef struct {
  float org[3];
  float tmin;
  float dir[3];
  float tmax;
} orc_ray;

/* src/raytracer/raytracer.h:9-17 TraceResult */
typedef struct {
  float normal_g[3];
  float t, u, v;
  uint32_t instance_id, geom_id, prim_id;
} orc_hit;

typedef struct {
  uint64_t samples, closest_rays, shadow_rays, nodes_visited, tris_tested, curves_tested;
  uint64_t bounces, sss_steps, rng_draws;
} orc_stats;

void orc_set_math_mode(int mode); /* 0 = libm (reference arithmetic), 1 = f64-rounded */
int orc_get_math_mode(void);

orc_scene* orc_scene_create(void);
void orc_scene_destroy(orc_scene*);

/* Scene::AddTriangleMesh (scene.h:19-24) + TriangleMesh ctor (mesh/triangle-mesh.cc:17-57).
 * normal_ids / texcoord_ids / material_ids may be NULL (-> all uint32(-1)).  Returns mesh id. */
int orc_add_triangle_mesh(orc_scene*, const float* vertices_xyzw, uint32_t num_vertices,
                          const float* normals_xyzw, uint32_t num_normals, const float* texcoords_uv,
                          uint32_t num_texcoords, const uint32_t* vertex_ids, const uint32_t* normal_ids,
                          const uint32_t* texcoord_ids, const uint32_t* material_ids, uint32_t num_faces);
/* Scene::AddCubicBezierCurveMesh (scene.h:26-32) */
int orc_add_curve_mesh(orc_scene*, const float* vertices_xyzr, uint32_t num_vertices,
                       const uint32_t* indices, const uint32_t* material_ids, uint32_t num_segments);
int orc_add_principled(orc_scene*, const orc_principled_param*);
int orc_add_hair(orc_scene*, const orc_hair_param*);
int orc_add_area_light(orc_scene*, const float emission[3]);
/* Scene::AddTexture (scene.h:46-51): float pixels, row-major, `channels` interleaved */
int orc_add_texture(orc_scene*, const float* pixels, uint32_t width, uint32_t height, uint32_t channels);
void orc_kat_texture_fetch(const float* pixels, uint32_t width, uint32_t height, uint32_t channels, float u, float v,
                           float out[3]);
int orc_create_local_scene(orc_scene*);
int orc_add_mesh_to_local_scene(orc_scene*, uint32_t local_scene_id, uint32_t mesh_id);
int orc_create_instance(orc_scene*, uint32_t local_scene_id, const float transform[16]);
int orc_attach_light_ids(orc_scene*, uint32_t instance_id, uint32_t geom_id, const uint32_t* ids, uint32_t n);
int orc_attach_material_ids(orc_scene*, uint32_t instance_id, uint32_t geom_id, const uint32_t* ids, uint32_t n);
int orc_commit(orc_scene*);
void orc_scene_aabb(const orc_scene*, float bmin[3], float bmax[3]);
uint32_t orc_bvh_depth(const orc_scene*);

/* Raytracer::FirstHitTrace1 / AnyHit1.  brute_force != 0 tests every primitive (no BVH). */
void orc_trace_closest(const orc_scene*, const orc_ray* rays, size_t n, orc_hit* hits, int brute_force);
void orc_trace_any(const orc_scene*, const orc_ray* rays, size_t n, uint8_t* occluded, int brute_force);

/* pbrlab::Render restated with per-(pixel,pass) seeding RNG((pass<<32)+y*W+x, seed_seq) and
 * ascending-pass accumulation.  Tiles (64x64 row-major, render-tile.cc:29-41)
 * with index % tile_world == tile_rank are rendered; other pixels stay zero.  rgba: W*H*4, count: W*H. */
void orc_render(const orc_scene*, uint32_t width, uint32_t height, uint32_t spp, uint32_t first_pass,
                uint64_t seed_seq, uint32_t tile_rank, uint32_t tile_world, uint32_t num_threads,
                float* rgba, uint32_t* count, orc_stats* stats);

/* The same with the job granularity of the worker pool chosen by the caller:
 *   ORC_JOBS_BLOCKS     a job = a 16x16 block of a tile x all passes (what orc_render does: per-pixel ascending-pass sums
 *                       by one thread -- the image is independent of the schedule; the checker's mode)
 *   ORC_JOBS_TILE_PASS  a job = (tile, pass), id -> (id % ntiles, id / ntiles), pixel sums under the tile's mutex: the
 *                       reference's own pool (render.cc:210-233, 175-183); float sums depend on the schedule like the
 *                       reference's, so it is used for the CPU baseline's clock only */
enum { ORC_JOBS_BLOCKS = 0, ORC_JOBS_TILE_PASS = 1 };
void orc_render_jobs(const orc_scene*, uint32_t width, uint32_t height, uint32_t spp, uint32_t first_pass,
                     uint64_t seed_seq, uint32_t tile_rank, uint32_t tile_world, uint32_t num_threads, uint32_t job_mode,
                     float* rgba, uint32_t* count, orc_stats* stats);

/* timing aid: the sum over the worker threads of the last orc_render / orc_render_jobs call of (end of the worker's last job -
 * start of the call), in seconds; divided by threads x wall time it is the pool's scheduling efficiency */
double orc_last_render_busy(void);

/* one sample, with a trace of the hit sequence: returns number of closest-hit records written
 * (<= max_hits; the path may be longer), radiance[3], draws consumed. */
uint32_t orc_sample_trace(const orc_scene*, uint32_t width, uint32_t height, uint32_t x, uint32_t y,
                          uint32_t pass, uint64_t seed_seq, float radiance[3], uint64_t* draws,
                          orc_hit* hits, uint32_t max_hits);
/* camera ray for (x,y,pass): render.cc:132-171 */
void orc_camera_ray(const orc_scene*, uint32_t width, uint32_t height, uint32_t x, uint32_t y,
                    uint32_t pass, uint64_t seed_seq, orc_ray* ray);

/* ---- known-answer hooks (leaf functions) ---- */
void orc_kat_rng(uint64_t initstate, uint64_t initseq, uint32_t n, float* out);
/* op: 0 sin 1 cos 2 exp 3 log 4 atan2(y=x,x=y2) 5 asin 6 exp2 7 log2 */
float orc_kat_fastmath(int op, float x, float y2);
float orc_kat_fresnel(float c, float eta);
float orc_kat_power_heuristic(float a, float b);
void orc_kat_lambert_sample(float u0, float u1, float out[5]);                       /* wi[3], f, pdf */
void orc_kat_ggx_eval(const float wi[3], const float wo[3], float ax, float ay, int distrib, float out[2]);
void orc_kat_ggx_sample(const float wo[3], float ax, float ay, float u0, float u1, int distrib, float out[5]);
/* hair: params = h, v0..v3, s, sigma_a[3], eta, alpha, tints[12], transparent_scale (23 floats) */
void orc_kat_hair_eval(const float wi[3], const float wo[3], const float* params, float out[4]);
void orc_kat_hair_sample(const float wo[3], const float* params, const float us[4], float out[7]);
void orc_kat_uniform_sphere(float u1, float u2, float out[3]);
void orc_kat_triangle_sampler(float u1, float u2, float out[2]);
/* ParamToBsdf (cycles-principled-shader.cc:244-412) dump: 34 floats, see pbr_oracle.c */
void orc_kat_param_to_bsdf(const orc_principled_param*, float out[34]);
/* hair ParamToBsdf (hair-shader.cc:100-151) with h: 23 floats in the orc_kat_hair_eval order */
void orc_kat_hair_param_to_bsdf(const orc_hair_param*, float h, float out[23]);
/* light tables: returns number of lights; per light (instance, geom, nprim, p_light, cdf_light) */
uint32_t orc_light_table(const orc_scene*, uint32_t light_index, uint32_t* instance_id, uint32_t* geom_id,
                         float* choose_prob, float* cdf, uint32_t* num_prims);
void orc_light_prims(const orc_scene*, uint32_t light_index, float* prim_prob, float* prim_cdf, float* prim_area_pdf);
void orc_create_tiles(uint32_t width, uint32_t height, uint32_t* out_sx_tx_sy_ty, uint32_t* num_tiles);
/* CyHair strand -> cubic Bezier (src/curve-util.cc:79-199): cvs xyz*n, radii n; out: 4*(n-1) xyzr */
int orc_to_cubic_bezier(const float* cvs, const float* radii, uint32_t n, float* out_xyzr);

#ifdef __cplusplus
}
#endif
#endif
