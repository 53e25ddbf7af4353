/*
 * pbrhip.h -- C ABI of the MI355X-native path-tracing core that sits behind pbrlab's
 * Scene / Render() / RenderLayer API (libpbrhip.so, built from pbrlab_amd/csrc).
 *
 * This is the drop-in boundary: scene construction keeps pbrlab's builder calls,
 * BVH build/flatten and tile dispatch stay on the host inside the library, and the per-sample hot
 * path -- camera ray -> GetRadiance bounce loop -> RenderLayer accumulation -- runs as HIP kernels.
 * Plain pointers and sizes only; host pointers in, host pointers out unless a function says "device".
 * Every entry point cites the reference interface it replaces (file:line under lighttransport/pbrlab).
 *
 * All functions return 0 on success or a negative PBRHIP_E* code; pbrhip_last_error() returns a
 * thread-local message.  The library fails loudly (PBRHIP_ENODEVICE) when no HIP device is usable:
 * there is no CPU fallback.
 */
#ifndef PBRHIP_H_
#define PBRHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBRHIP_OK 0
#define PBRHIP_EINVAL (-1)    /* bad argument / id out of range */
#define PBRHIP_ESIZE (-2)     /* std::runtime_error("... param error") cases of scene.cc:64-94 */
#define PBRHIP_ENODEVICE (-3) /* no usable HIP device */
#define PBRHIP_EHIP (-4)      /* a HIP runtime call failed */
#define PBRHIP_EUNSUPPORTED (-5)
#define PBRHIP_ESTATE (-6)    /* scene not committed / already committed */
#define PBRHIP_EOVERFLOW (-7) /* traversal stack overflow (BVH deeper than the kernel supports) */
#define PBRHIP_ENOMEM (-8)    /* host allocation failed (std::bad_alloc) */
#define PBRHIP_ECOMM (-9)     /* RCCL is unavailable or one of its calls failed */

#define PBRHIP_NONE 0xFFFFFFFFu

typedef struct pbrhip_scene pbrhip_scene;

/* pbrlab::CyclesPrincipledBsdfParameter (src/material-param.h:24-49), name dropped */
typedef struct {
  float base_color[3];
  float subsurface;
  float subsurface_radius[3];
  float subsurface_color[3];
  float metallic, specular, specular_tint, roughness, anisotropic, anisotropic_rotation;
  float sheen, sheen_tint, clearcoat, clearcoat_roughness, ior, transmission, transmission_roughness;
  uint32_t base_color_tex_id, subsurface_color_tex_id; /* PBRHIP_NONE or an id from pbrhip_scene_add_texture */
} pbrhip_principled_param;

/* pbrlab::HairBsdfParameter (src/material-param.h:51-72) */
typedef struct {
  uint32_t coloring_hair; /* 0 = kRGB, 1 = kMelanin */
  float base_color[3];
  float melanin, melanin_redness, melanin_randomize;
  float roughness, azimuthal_roughness, ior, shift;
  float specular_tint[3], second_specular_tint[3], transmission_tint[3];
} pbrhip_hair_param;

/* pbrlab::Ray (src/ray.h:9-14), packed as two float4 */
typedef struct {
  float org[3];
  float tmin;
  float dir[3];
  float tmax;
} pbrhip_ray;

/* pbrlab::TraceResult (src/raytracer/raytracer.h:9-17) */
typedef struct {
  float normal_g[3];
  float t, u, v;
  uint32_t instance_id, geom_id, prim_id;
} pbrhip_hit;

/* arguments of pbrlab::Render (src/render.h:14-17) plus what the reference hard-wires */
typedef struct {
  uint32_t width, height, num_sample; /* Render(scene, width, height, num_sample, ...) */
  uint32_t first_pass;                /* passes [first_pass, first_pass+num_sample) (progressive resume) */
  uint64_t seed_seq;                  /* PCG32 initseq; the reference uses 1234567890 (render.cc:215) */
  uint32_t tile_rank, tile_world;     /* 64x64 tiles (render.cc:107-108) with index % world == rank */
  uint32_t max_paths_in_flight;       /* 0 = default (half of the free HBM, <= 256 Mi): passes are rendered in chunks of this many paths */
  uint32_t flags;                     /* PBRHIP_RENDER_* */
  uint32_t num_streams;               /* the passes of a chunk are split into path groups, one HIP stream each (pbrhip.cpp::plan_groups).
                                         0 = default: two equal groups started at once when the chunk holds at least 96 Mi paths,
                                         otherwise one (pipelined plans were measured and lost); n > 0: n equal groups (at most 16),
                                         of which at most 8 are in flight at a time */
  uint32_t tail_paths;                /* once a group has at most this many live paths, the rest of every path runs in ONE launch
                                         (k_tail) instead of one set of launches per bounce; 0 = default 262144, 0xFFFFFFFF = never */
  uint32_t shard_block;               /* edge of the square pixel blocks dealt to ranks (block index % tile_world == tile_rank);
                                         0 = 64, the reference's tile (render-tile.cc:29-41).  Smaller blocks balance the load of
                                         many GPUs better; the image does not depend on it */
} pbrhip_render_desc;

#define PBRHIP_RENDER_STATS 1u   /* count BVH nodes / primitives visited (slower; for algorithmic bytes) */
#define PBRHIP_RENDER_TIMING 2u  /* time every kernel launch with HIP events on the render stream */
#define PBRHIP_RENDER_NO_CLEAR 4u /* keep the layer's current contents (Render() clears: render.cc:99-100) */
#define PBRHIP_RENDER_TIMING_TRACE 8u /* time only the k_trace launches (ms_trace_closest / n_trace_closest): a sixth of the events of
                                         PBRHIP_RENDER_TIMING, for measurements that must not slow the frame (bench.py's timed region) */

/* filled by pbrhip_render when stats != NULL */
typedef struct {
  uint64_t samples, iterations, chunks;
  uint64_t closest_rays, closest_nodes, closest_tris, closest_curves; /* PBRHIP_RENDER_STATS */
  uint64_t shadow_rays, shadow_nodes, shadow_tris, shadow_curves;
  /* PBRHIP_RENDER_TIMING: total ms and launch count per kernel */
  double ms_generate, ms_trace_closest /* = k_trace: closest + shadow rays */, ms_surface /* = k_classify */, ms_shade_principled, ms_shade_hair, ms_sss_step,
      ms_tail /* = k_tail */, ms_accumulate, ms_compact;
  uint64_t n_trace_closest, n_tail, n_surface, n_shade_principled, n_shade_hair, n_sss_step;
  double ms_total; /* wall time of the call measured on the host */
  uint64_t tail_closest_rays, tail_shadow_rays; /* PBRHIP_RENDER_STATS: rays traced inside k_tail (not part of the k_trace counts above) */
  uint64_t pruned_rays; /* PBRHIP_RENDER_STATS: closest-hit rays of the reference that were never traced: the path's next
                           Russian roulette was already known to fail and the ray cannot reach an area light (it misses every
                           light primitive, or the bounding box of every light), so nothing it could find changes the image (closest + tail_closest + pruned = the reference's count) */
  uint64_t passes_done; /* passes every pixel of this rank holds when the call returns (= num_sample unless cancelled) */
  uint64_t node_bytes;  /* footprint of one node of the tree k_trace walked: 64 (the Q tree: 4 children, quantised boxes; or the binary tree) */
  uint64_t curve_bytes; /* bytes fetched per curve-piece test: 32 (Q tree: two 16-byte points of a chain) or 64 (binary tree: one slot) */
  uint64_t suspended_rays; /* PBRHIP_RENDER_STATS: closest-hit rays a k_trace launch suspended at its drain and the next launch resumed
                              (counted ONCE in closest_rays; their node / primitive counts are complete: nothing is traversed twice) */
  double ms_host_idle;  /* PBRHIP_RENDER_TIMING: sum over the path groups of the time their stream sat empty between two bursts of
                           launches (the host's round trips; 0 when the next iteration was always enqueued in time) */
} pbrhip_render_stats;

/* Layout version of the structs of this header (pbrhip_render_desc, pbrhip_render_stats, pbrhip_*_param, pbrhip_hit): it
 * changes whenever one of them changes (3 -> 4: pbrhip_render_stats grew by curve_bytes; 5 -> 6: by suspended_rays, ms_host_idle).  The library writes whole structs
 * (n of them for pbrhip_render_multi), so a caller compiled against another version must not call it: check
 * pbrhip_abi_version() == PBRHIP_ABI_VERSION once after loading (include/pbrlab_hip.hpp and pbrlab_amd/api.py do). */
#define PBRHIP_ABI_VERSION 6u
uint32_t pbrhip_abi_version(void);
/* Which implementation of cos / sin / exp / log -- the reference's std::cos ... on float, sampler/sampling-utils.h:10-14,
 * closure/microfacet-ggx.h:55-118, shader/random-walk-sss.h:116,183,192-194 -- this build of the library computes with:
 * PBRHIP_MATH_GLIBCF: GNU libc's float functions restated bit for bit (include/pbr_glibcf.h; the default since round 5: the
 * reference's own arithmetic where its libm is glibc 2.35 as Ubuntu 22.04 builds it, x86-64 with FMA: tests/test_glibcf.py); PBRHIP_MATH_F64R: the double-precision value
 * rounded once (include/pbr_f64r.h; a library built with -DPBR_MATH_F64R). */
#define PBRHIP_MATH_F64R 1u
#define PBRHIP_MATH_GLIBCF 2u
uint32_t pbrhip_math_mode(void);
size_t pbrhip_sizeof_render_stats(void); /* == sizeof(pbrhip_render_stats) of the library's build */

const char* pbrhip_last_error(void);
int pbrhip_device_count(int* count);
/* selects the HIP device used by subsequently created scenes (one process per GPU: LOCAL_RANK) */
int pbrhip_set_device(int device);

/* ---- scene construction: pbrlab::Scene (src/scene.h:14-111) ---- */
int pbrhip_scene_create(pbrhip_scene** out); /* Scene::Scene, scene.cc:11 */
int pbrhip_scene_destroy(pbrhip_scene*);     /* Scene::~Scene, scene.cc:12 */

/* Scene::AddTriangleMesh (scene.h:19-24) with the TriangleMesh ctor (mesh/triangle-mesh.cc:17-57) and the
 * shared Attribute buffers (mesh/attribute.h:8-12).  The reference shares the buffers by pointer
 * (raytracer.h:34); this call copies them.  normal_ids / texcoord_ids / material_ids may be NULL
 * (ids default to uint32(-1)).  vertices: xyzw * num_vertices, normals: xyzw, texcoords: uv. */
int pbrhip_scene_add_triangle_mesh(pbrhip_scene*, const float* vertices_xyzw, uint32_t num_vertices,
                                   const float* normals_xyzw, uint32_t num_normals, const float* texcoords_uv,
                                   uint32_t num_texcoords, const uint32_t* vertex_ids, const uint32_t* normal_ids,
                                   const uint32_t* texcoord_ids, const uint32_t* material_ids, uint32_t num_faces,
                                   uint32_t* mesh_id);
/* Scene::AddCubicBezierCurveMesh (scene.h:26-32, mesh/cubic-bezier-curve-mesh.cc:7-15):
 * vertices xyz+radius, indices = first control point of each cubic segment */
int pbrhip_scene_add_curve_mesh(pbrhip_scene*, const float* vertices_xyzr, uint32_t num_vertices,
                                const uint32_t* indices, const uint32_t* material_ids, uint32_t num_segments,
                                uint32_t* mesh_id);
/* Scene::AddMaterialParam (scene.h:39-44) for the two alternatives of MaterialParameter */
int pbrhip_scene_add_principled_material(pbrhip_scene*, const pbrhip_principled_param*, uint32_t* material_id);
int pbrhip_scene_add_hair_material(pbrhip_scene*, const pbrhip_hair_param*, uint32_t* material_id);
/* Scene::AddTexture (scene.h:46-51) with pbrlab::Texture (src/texture.h:13-44): float pixels, row-major, `channels`
 * (1..4) interleaved; sampled bilinearly with clamp addressing like Texture::FetchFloat3 (texture.cc:43-68,
 * image-utils.cc:99-167).  Referenced by base_color_tex_id / subsurface_color_tex_id; ids are checked at commit. */
int pbrhip_scene_add_texture(pbrhip_scene*, const float* pixels, uint32_t width, uint32_t height, uint32_t channels,
                             uint32_t* texture_id);
/* Scene::AddLightParam (scene.h:34-37) with AreaLightParameter (light-param.h:20-23) */
int pbrhip_scene_add_area_light(pbrhip_scene*, const float emission[3], uint32_t* light_id);
/* Scene::CreateLocalScene (scene.cc:157-164), AddMeshToLocalScene (scene.cc:14-62), CreateInstance (scene.cc:106-155;
 * transform = float[4][4], row-vector convention v' = v*M with the translation in the last row, NULL = identity.  As in the
 * reference the transform only reaches the raytracer (raytracer_impl.cc:61-81 hands it to Embree): rays see the transformed
 * geometry -- triangles and curve control points transformed, curve radii as they are -- while geometric / shading normals,
 * texcoords and light sampling stay in the instance's local space (scene.cc:217,237 and light-manager.h:128-136 "TODO
 * transform").  A singular or non-finite matrix is refused (PBRHIP_EINVAL). */
int pbrhip_scene_create_local_scene(pbrhip_scene*, uint32_t* local_scene_id);
int pbrhip_scene_add_mesh_to_local_scene(pbrhip_scene*, uint32_t local_scene_id, uint32_t mesh_id, uint32_t* geom_id);
int pbrhip_scene_create_instance(pbrhip_scene*, uint32_t local_scene_id, const float* transform4x4,
                                 uint32_t* instance_id);
/* Scene::AttachLightParamIdsToInstance / AttachMaterialParamIdsToInstance (scene.cc:64-94), one geometry at a
 * time; a size mismatch returns PBRHIP_ESIZE where the reference throws std::runtime_error */
int pbrhip_scene_attach_light_ids(pbrhip_scene*, uint32_t instance_id, uint32_t geom_id, const uint32_t* light_ids,
                                  uint32_t n);
int pbrhip_scene_attach_material_ids(pbrhip_scene*, uint32_t instance_id, uint32_t geom_id,
                                     const uint32_t* material_ids, uint32_t n);
/* Scene::CommitScene (scene.cc:96-104): light tables (LightManager::RegisterInstanceMesh/Commit,
 * light-manager.cc:29-184), BVH build + flatten (replaces Embree's rtcCommitScene, raytracer_impl.cc:93-197),
 * upload to HBM, scene bounds (rtcGetSceneBounds, raytracer_impl.cc:199-211) */
int pbrhip_scene_commit(pbrhip_scene*);
/* Which builder pbrhip_scene_commit uses for the acceleration structure (the reference has one: Embree's, behind
 * rtcCommitScene, raytracer_impl.cc:136-147,181-192).  HOST_SAH (default): binned SAH on the host cores, the best tree.
 * GPU_LBVH: Morton-order linear BVH built on the GPU -- commit in milliseconds for edit -> re-render loops and
 * multi-million-primitive hair, at a higher traversal cost.  Rendered images do not depend on the choice.
 * Call before pbrhip_scene_commit.  (Environment override: PBRHIP_BVH=gpu|host.) */
#define PBRHIP_BVH_HOST_SAH 0
#define PBRHIP_BVH_GPU_LBVH 1
int pbrhip_scene_set_bvh_builder(pbrhip_scene*, int builder);
/* Scene::FetchSceneAABB (scene.cc:251-259) */
int pbrhip_scene_aabb(const pbrhip_scene*, float bmin[3], float bmax[3]);
/* Scene::FetchMeshMaterialParameters + EditQueue edits between renders (pc/pc-common.cc:57-84): replace one
 * material's parameters in place on a committed scene */
int pbrhip_scene_update_principled_material(pbrhip_scene*, uint32_t material_id, const pbrhip_principled_param*);
int pbrhip_scene_update_hair_material(pbrhip_scene*, uint32_t material_id, const pbrhip_hair_param*);
/* BVH facts for reports: node count, leaf slots, depth, device bytes */
int pbrhip_scene_info(const pbrhip_scene*, uint64_t* num_nodes, uint64_t* num_slots, uint32_t* depth,
                      uint64_t* device_bytes);

/* ---- the hot path ---- */
/* pbrlab::Render (src/render.h:14-17, render.cc:192-241).  Blocking.  rgba: width*height*4 floats (sum of
 * radiance, A = sample count), count: width*height (RenderLayer, render-layer.h:11-26).
 *
 * cancel: NULL or the address of a byte another thread may set non-zero while the call runs -- the object
 * representation of the reference's `const std::atomic_bool& cancel_render_flag` (a lock-free byte).  It is read live
 * at every host round trip of the render loop (once per wavefront iteration of a path group, i.e. every few
 * milliseconds at most; the reference polls before every tile job, render.cc:217).  A cancelled call drains what is in
 * flight, drops the passes that are not complete and returns PBRHIP_OK with a consistent layer: every pixel of this rank
 * holds exactly *finish_pass passes.
 * finish_pass: NULL or a size_t (the reference's std::atomic_size_t, render.cc:224-231) that is stored to (atomically,
 * monotone) while the call runs, each time a further group of passes is complete for every pixel.
 * Per-sample RNG: RNG((pass << 32) + y*width + x, seed_seq). */
int pbrhip_render(pbrhip_scene*, const pbrhip_render_desc*, const volatile unsigned char* cancel, float* rgba,
                  uint32_t* count, size_t* finish_pass, pbrhip_render_stats* stats);
/* Same, writing into DEVICE buffers owned by the caller (e.g. a torch tensor that is then exchanged over RCCL):
 * d_rgba / d_count are device pointers on the scene's device; nothing is copied to the host. */
int pbrhip_render_device(pbrhip_scene*, const pbrhip_render_desc*, const volatile unsigned char* cancel, float* d_rgba,
                         uint32_t* d_count, size_t* finish_pass, pbrhip_render_stats* stats);

/* ---- multi-GPU (new: the reference has one process and std::threads, render.cc:203-238) ----
 * Pixels are independent, so the only exchange is the RenderLayer at the end of a frame: rank r renders the pixel blocks
 * with index % world == r (pbrhip_render_desc.tile_rank / tile_world / shard_block) into a zeroed full-size layer, then the
 * layers are combined on one rank.  Disjoint blocks + zeros: the combined frame is bit-identical to the one-GPU frame. */

/* (1) one process, several GPUs: scenes[i] is a committed copy of the same scene on its own device (pbrhip_scene_replicate;
 * several scenes may share a device).  One host thread per scene renders rank desc->tile_rank * n + i of
 * desc->tile_world * n; the shards are then copied device-to-device (xGMI peer copies: only the blocks a rank rendered
 * travel) into scenes[0]'s layer and from there to the host.  cancel / finish_pass as in pbrhip_render, except that the
 * devices of a cancelled call stop at their own pass counts: *finish_pass = the passes complete on EVERY device (the
 * minimum), a device that was ahead keeps its further passes in its pixels, and `count` says per pixel how many it holds;
 * stats: NULL or n records. */
int pbrhip_render_multi(pbrhip_scene* const* scenes, uint32_t n, const pbrhip_render_desc*,
                        const volatile unsigned char* cancel, float* rgba, uint32_t* count, size_t* finish_pass,
                        pbrhip_render_stats* stats);
/* A committed scene's copy on another device (device memory is copied device-to-device: no second ingestion, no second
 * BVH build); what `pbrlab-hip-cli --gpus N` calls per extra GPU. */
int pbrhip_scene_replicate(const pbrhip_scene* src, int device, pbrhip_scene** out);

/* (2) one process per GPU (torch.distributed / MPI launchers): an RCCL communicator inside the library.
 * pbrhip_comm_unique_id: ncclGetUniqueId on one rank; the launcher hands the 128 bytes to every rank.
 * pbrhip_comm_create: ncclCommInitRank on the device selected with pbrhip_set_device (collective: every rank calls it). */
#define PBRHIP_COMM_ID_BYTES 128
typedef struct pbrhip_comm pbrhip_comm;
int pbrhip_comm_unique_id(unsigned char id[PBRHIP_COMM_ID_BYTES]);
int pbrhip_comm_create(pbrhip_comm** out, const unsigned char id[PBRHIP_COMM_ID_BYTES], int rank, int world);
int pbrhip_comm_destroy(pbrhip_comm*);
/* ncclReduce(sum) of rgba (f32) and count (u32) to `root`, in place, in one RCCL group: what SURVEY 8e specifies.  Works
 * for any layers (e.g. different passes of the same pixels on different ranks).  Blocking. */
int pbrhip_comm_reduce_layer(pbrhip_comm*, float* d_rgba, uint32_t* d_count, size_t num_pixels, int root);
/* The same result for layers rendered with (tile_rank, tile_world) = (rank, world) of the communicator and `desc`'s
 * width / height / shard_block: a reduce whose zero terms do not travel.  Every rank packs the pixels of its own blocks
 * (20 bytes per pixel) and sends them straight to `root` (ncclSend / ncclRecv in one group: xGMI is point-to-point, so
 * the N - 1 shards arrive over N - 1 different links at once), root scatters them into its layer.  Blocking. */
int pbrhip_comm_gather_layer(pbrhip_comm*, pbrhip_scene*, const pbrhip_render_desc* desc, float* d_rgba,
                             uint32_t* d_count, int root);

/* Raytracer::FirstHitTrace1 / AnyHit1 (src/raytracer/raytracer.h:95-111, raytracer_impl.cc:268-287) over an
 * array of rays: test hooks for hit-index parity */
int pbrhip_trace_closest(pbrhip_scene*, const pbrhip_ray* rays, size_t n, pbrhip_hit* hits);
int pbrhip_trace_any(pbrhip_scene*, const pbrhip_ray* rays, size_t n, uint8_t* occluded);

/* The device's leaf functions on arrays of inputs -- a test hook like the two above: what the kernels compute for the generator
 * (src/random/rng.h), the fast math of the hair BSDF (src/pbrlab_math.h:135-344), Fresnel and the MIS weight (closure-util.h,
 * sampling-utils.h), the Lambert / sphere / triangle samplers (closure/lambert.h, sampler/sampling-utils.h), GGX eval and sample
 * (closure/microfacet-ggx.h:164-286) and the hair BSDF (closure/energy-conserving-hair-bsdf.h:295-572), so that they can be compared
 * with outputs of the reference's own headers.  n items; item i reads in_words floats at in + i * in_words and writes out_words at
 * out + i * out_words (host memory; integers travel as their bits):
 *   RNG          in: initstate lo, hi, initseq lo, hi            out: out_words draws
 *   FASTMATH     in: function (0 sin 1 cos 2 exp 3 log 4 atan2(x, y) 5 asin 6 exp2 7 log2 8 / 9 sincos' sine / cosine), x, y   out: 1
 *   FRESNEL      in: cos, eta -> 1        MIS: sampled pdf, other pdf -> 1
 *   LAMBERT      in: u0, u1 -> wi[3], f, pdf     SPHERE: u1, u2 -> v[3]     TRIANGLE: u1, u2 -> a, b
 *   GGX_EVAL     in: wi[3], wo[3], ax, ay, distrib -> f, pdf        GGX_SAMPLE: wo[3], ax, ay, u0, u1, distrib -> wi[3], f, pdf
 *   HAIR_EVAL    in: wi[3], wo[3], params[23] -> f[3], pdf           HAIR_SAMPLE: wo[3], params[23], us[4] -> wi[3], f[3], pdf
 *   (params: h, v[4], s, sigma_a[3], eta, alpha, tints[4][3], transparent_scale) */
enum {
  PBRHIP_LEAF_RNG = 0, PBRHIP_LEAF_FASTMATH = 1, PBRHIP_LEAF_FRESNEL = 2, PBRHIP_LEAF_MIS = 3, PBRHIP_LEAF_LAMBERT = 4, PBRHIP_LEAF_SPHERE = 5,
  PBRHIP_LEAF_TRIANGLE = 6, PBRHIP_LEAF_GGX_EVAL = 7, PBRHIP_LEAF_GGX_SAMPLE = 8, PBRHIP_LEAF_HAIR_EVAL = 9, PBRHIP_LEAF_HAIR_SAMPLE = 10
};
int pbrhip_leaf_eval(uint32_t op, const float* in, size_t n, uint32_t in_words, float* out, uint32_t out_words);
/* Texture::FetchFloat3 (src/texture.cc:43-68 -> BilinearFilter, src/image-utils.cc:99-167) of texture `texture_id` of a committed
 * scene at n coordinates uv[2 * i], uv[2 * i + 1] -> rgb[3 * i ..]: the fetch of the textured shading kernels, as a test hook */
int pbrhip_texture_fetch(pbrhip_scene*, uint32_t texture_id, const float* uv, size_t n, float* rgb);

/* CreateTiles (src/render-tile.cc:29-41): out = sx,tx,sy,ty per tile (may be NULL to query the count) */
int pbrhip_create_tiles(uint32_t width, uint32_t height, uint32_t* out_sx_tx_sy_ty, uint32_t* num_tiles);

#ifdef __cplusplus
}
#endif
#endif /* PBRHIP_H_ */
