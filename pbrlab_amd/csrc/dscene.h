// dscene.h -- flat, device-resident scene layout (what pbrhip_scene_commit uploads).
//
// HBM layout (all arrays 16-byte aligned, read-only during a render):
//   nodes      BVH2, 64 B per node: both children's AABBs + two child references
//   slots      leaf-ordered primitives, 64 B per slot (4 x float4):
//                triangle: v0.xyz,_ | v1.xyz,_ | v2.xyz,_ | unused     (world space)
//                curve   : 4 cubic Bezier control points xyz + radius  (world space)
//   shade      one 128-byte line per slot with everything a hit needs for shading: triangle corners,
//              corner normals, canonical primitive id (gid), material id, light record, flags, TraceResult ids
//   materials  one closure record per material (ParamToBsdf hoisted to commit time unless textured)
//   textures   float pixel pool + descriptor table (bilinear, clamp addressing)
//   lights     per-light and per-light-primitive sampling tables (LightManager::Commit)
#pragma once
#include <math.h>

#include "dclosures.h"

namespace pb {

// child reference: internal node index, or leaf = 0x80000000 | kind<<30 | first_slot<<3 | (count-1)
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kCurveBit = 0x40000000u;
constexpr uint32_t kEmptyChild = 0xFFFFFFFFu;  // (a leaf reference that can never be produced)
#ifndef PB_MAX_LEAF
#define PB_MAX_LEAF 2
#endif
constexpr int kMaxLeaf = PB_MAX_LEAF;
#ifndef PB_STACK_DEPTH
#define PB_STACK_DEPTH 64
#endif
constexpr int kStackDepth = PB_STACK_DEPTH;  // traversal stack entries per ray (a tree that can need more is refused at commit)
constexpr int kSimpleLdsStack = 40;          // of which the one-ray-per-lane traversal (dtrace.h::traverse: k_tail, trace hooks) keeps this many in LDS
// The host builder numbers the first kTopNodes nodes breadth first (levels 0..5 of a full tree): DScene::top_nodes of
// them form the top of the tree and are staged in LDS by k_trace (4 KB per block).  0 for trees without that numbering.
#ifndef PB_TOP_NODES
#define PB_TOP_NODES 64
#endif
constexpr int kTopNodes = PB_TOP_NODES;
// ... and the Q tree's kernels.  0 since round 4: with the packed two-triangle leaf test the triangle kernel is at its register
// budget and the staging branch costs it a spill; same-box A/B on C2 (k_trace ms per frame, two runs each): no staging 33.9 / 33.6,
// 64 nodes 35.1 / 34.8, 128 nodes 34.9 / 35.0, 256 nodes (12 stack entries in LDS) 34.8 / 35.3 -- the top of the tree is served by
// the vector L1 either way (profiles/README.md).
#ifndef PB_TOP_NODES_WIDE
#define PB_TOP_NODES_WIDE 0
#endif
constexpr int kTopNodesWide = PB_TOP_NODES_WIDE;

// Both children's boxes, interleaved [axis][child] so that the two children's values of one bound sit in an aligned
// register pair after the 16-byte loads: the slab test runs on v_pk_add_f32 / v_pk_mul_f32 (two children per instruction).
struct alignas(16) BvhNode {
  float lo[3][2];
  float hi[3][2];
  uint32_t c0, c1;
  uint32_t pad[2];
  // Boxes are stored widened by 2^-16 relative (+ a denormal-sized absolute step) on every side.  A ray parallel to an
  // axis whose origin lies exactly ON a face of the tight box would otherwise produce 0 * inf = NaN in the slab test and
  // be rejected although it can hit the primitive; with the stored faces strictly outside the geometry that product is
  // -inf / +inf (no constraint), and a ray that is exactly on a stored face is too far from the geometry to hit it.
  static float widen_lo(float v) { return v - (fabsf(v) * 1.52587890625e-05f + 1e-30f); }
  static float widen_hi(float v) { return v + (fabsf(v) * 1.52587890625e-05f + 1e-30f); }
  void set_box(int child, const float* l, const float* h) {
    for (int a = 0; a < 3; a++) lo[a][child] = widen_lo(l[a]), hi[a][child] = widen_hi(h[a]);
  }
};
static_assert(sizeof(BvhNode) == 64, "node must be 64 B");

// The same tree with four children per node and quantised boxes (bvh_build.cpp::build_qtree: the children of a Q node are a
// frontier of <= 4 subtrees below a binary node, chosen by dynamic programming over the area of the quantised boxes): 64 B = ONE item.  Per axis the node has an origin `org` and a
// step `s` (extent / 253); child i's box is [fma(qlo[i], s, org), fma(qhi[i], s, org)] in single precision with 8-bit qlo / qhi (byte i of the
// word), rounded outwards: the builder evaluates that very expression and moves a bound out until the result encloses the
// binary tree's (already widened) box.  The traversal rebuilds the bounds the same way and then runs the binary tree's slab
// arithmetic on them, so the quantised test inherits its properties (monotone in the box, correct for axis-parallel rays).
// Child references: an inner child = its item index; a triangle leaf = kLeafBit | first << 3 | (count - 1) with `first` the
// index of the leaf's TriPair among the tree's own triangle records (DScene::q_tri0; 80 B each, below; scenes with curves: of
// its first 48-byte triangle slot); a curve leaf = kLeafBit | kCurveBit | first << 3 |
// (count - 1) with `first` a POINT index (DScene::q_pt0): the linear pieces of a strand are stored as a chain of points
// (xyz + radius, 16 B), piece p = points p, p + 1, piece-in-cubic index = p & 3 (every cubic starts at a multiple of 4).
// An unused child has reference kEmptyChild.
// Round 6 (PB_CURVE_RECORDS): a curve leaf is a RECORD instead of a stretch of a chain: four consecutive 16-byte words a0 a1 b0 b1 -- the
// end points of its one or two pieces, copied; 64-byte aligned: one item, like a node -- and both pieces are tested in ONE turn of the
// traversal.  Why: nine in ten of the two-piece leaves the SAH builder makes over hair are two pieces of NEIGHBOURING STRANDS, side by
// side, not neighbours in a chain; rounds 3-5 cut such a leaf into two leaves of one piece (two children of its node, two turns).
// Its reference is  kLeafBit | kCurveBit | (P | i_a) << 3 | (two pieces ? kCurvePairBit | i_b : 0),  P = the record's first point index
// (a multiple of 4), i_a / i_b = the pieces' indices in their cubics (what `p & 3` was for a chain piece); piece a is point P, piece b
// point P + 2 (q_hitcode has their codes).  A build option (-DPB_CURVE_RECORDS=0: the chains of rounds 3-5, one piece per turn).
constexpr uint32_t kCurvePairBit = 4u;
#ifndef PB_CURVE_RECORDS
#define PB_CURVE_RECORDS 1
#endif
#ifndef PB_CURVE_TWO
#define PB_CURVE_TWO 1  // curve records: 1 = both pieces of a leaf in ONE traversal turn (88-91 registers: five blocks per CU), 0 = one piece per turn (six blocks per CU)
#endif
struct alignas(16) QNode {
  float org[3], sx;
  float sy, sz;
  uint32_t qlo_x, qlo_y;
  uint32_t qlo_z, qhi_x, qhi_y, qhi_z;
  uint32_t c[4];
};
static_assert(sizeof(QNode) == 64, "quantised wide node must be 64 B");
// A triangle leaf of the Q tree of a TRIANGLE-ONLY scene (scenes with curves keep 48 bytes per triangle -- three corners, the
// hit code in the third word's .w -- one after the other: their kernels have no registers to spare for the packed test and
// meet triangles rarely): its one or two triangles (a, b) interleaved coordinate by coordinate, five 16-byte words
//   (v0x_a v0x_b v0y_a v0y_b) (v0z_a v0z_b v1x_a v1x_b) (v1y_a v1y_b v1z_a v1z_b) (v2x_a v2x_b v2y_a v2y_b) (v2z_a v2z_b code_a code_b)
// so that the two triangles' values of one coordinate sit in an aligned register pair after the loads and the intersection test
// runs on packed fp32 for both (dtrace.h::tri_test_pair).  code = the complete hit code (slot | routing bits); a leaf of ONE
// triangle stores it twice with code_b = kNone.  80 B per leaf instead of 48 B per triangle: -17 % for a pair, +67 % for a single.
constexpr uint32_t kTriPairWords = 5;

// during a traversal of the Q tree a curve hit is held as kQPointHit | point index; it becomes the hit code every other
// stage sees (slot | routing bits, below) through DScene::q_hitcode when the ray is delivered
constexpr uint32_t kQPointHit = 0x80000000u;

constexpr uint32_t kSlotHasNormals = 1u;
constexpr uint32_t kSlotIsCurve = 2u;
constexpr uint32_t kSlotMatHair = 4u;   // material kind, denormalised here so a hit can be routed without
constexpr uint32_t kSlotMatNone = 8u;   // touching the material table (shader.cc:11-17 for "none")
constexpr uint32_t kSlotHasUV = 16u;

// What shading needs of a primitive, 128 bytes, laid out by how often a hit needs it: the first 32 bytes serve every hit on a
// flat triangle (2 lane requests of 16 B -- the shading kernels are bound by the vector memory path's request rate, TD_TD_BUSY
// 0.90: profiles/README.md); corner normals / control points / texcoords only for the hits whose code says so (kHitMore).
// ng, ns_flat: the geometric normal normalize_raw(cross(v1 - v0, v2 - v0)) and the shading normal of a triangle without corner
// normals, vnormalize(cross(v1 - v0, v2 - v1)) (CalcGeometryNormal, triangle-mesh.cc:181-184), of the mesh's own (local) corners:
// evaluated at commit by the very functions the kernels used to call per hit (dmath.h, host and device: IEEE single precision,
// no contraction) -- 6 words instead of 9, and no cross product, square root and division per hit.
struct alignas(128) ShadeRec {
  float ng[3];
  uint32_t matflags;  // material (24 bits; 0xFFFFFF = none) | kSlot* flags << 24
  float ns_flat[3];
  uint32_t lightrec;
  float n[9];         // words 8..16: corner shading normals xyz (kSlotHasNormals).  Curve piece: words 8..23 hold the cubic's four control points xyzr
  uint32_t pad0;
  float uv[6];        // words 18..23: corner texcoords (kSlotHasUV), mesh/triangle-mesh.cc:126-156
  uint32_t gid, instance_id, geom_id, prim_id;  // words 24..27
  uint32_t pad[4];
};
static_assert(sizeof(ShadeRec) == 128, "one cache line per primitive");

// hit record code (Hit::slot, P.hit[].w): slot | routing bits; kNone = miss.  The routing bits are stored with the
// primitive's traversal data (.w of the third 16-byte word of its slot), so a hit can be routed (k_classify) without
// touching the primitive's ShadeRec line.
constexpr uint32_t kHitSlotMask = 0x07FFFFFFu;
constexpr uint32_t kHitMore = 1u << 27;        // shading needs words 8..23 of the ShadeRec (corner normals, texcoords, a curve's control points)
constexpr uint32_t kHitHair = 1u << 28;        // hair material (== kSlotMatHair)
constexpr uint32_t kHitLight = 1u << 29;       // the primitive is an area-light primitive (lightrec != kNone)
constexpr uint32_t kHitNoMaterial = 1u << 30;  // == kSlotMatNone

// closure set of one material == struct CyclesPrincipledBsdf (cycles-principled-shader.cc:20-45)
struct PrincipledBsdf {
  int enable_diffuse;
  V3 diffuse_weight;
  int enable_subsurface;
  V3 subsurface_weight, subsurface_albedo, subsurface_radius;
  int enable_specular;
  V3 specular_weight;
  float alpha_x, alpha_y, ior;
  V3 specular_color;
  int enable_clearcoat;
  V3 clearcoat_weight;
  float clearcoat_alpha_x, clearcoat_alpha_y, clearcoat_ior;
  V3 clearcoat_color;
};
PB_HD PrincipledBsdf default_bsdf() {
  PrincipledBsdf b;
  b.enable_diffuse = 0, b.diffuse_weight = V3(0.f);
  b.enable_subsurface = 0, b.subsurface_weight = V3(0.f), b.subsurface_albedo = V3(0.f), b.subsurface_radius = V3(0.f);
  b.enable_specular = 0, b.specular_weight = V3(0.f), b.alpha_x = 1.f, b.alpha_y = 1.f, b.ior = 1.5f;
  b.specular_color = V3(0.f);
  b.enable_clearcoat = 0, b.clearcoat_weight = V3(0.f), b.clearcoat_alpha_x = 1.f, b.clearcoat_alpha_y = 1.f;
  b.clearcoat_ior = 1.5f, b.clearcoat_color = V3(0.f);
  return b;
}

struct PrincipledParam {  // == pbrhip_principled_param == CyclesPrincipledBsdfParameter (material-param.h:24-49)
  float base_color[3];
  float subsurface;
  float subsurface_radius[3];
  float subsurface_color[3];
  float metallic, specular, specular_tint, roughness, anisotropic, anisotropic_rotation;
  float sheen, sheen_tint, clearcoat, clearcoat_roughness, ior, transmission, transmission_roughness;
  uint32_t base_color_tex_id, subsurface_color_tex_id;
};
struct HairParam {  // == pbrhip_hair_param == HairBsdfParameter (material-param.h:51-72)
  uint32_t coloring_hair;
  float base_color[3];
  float melanin, melanin_redness, melanin_randomize;
  float roughness, azimuthal_roughness, ior, shift;
  float specular_tint[3], second_specular_tint[3], transmission_tint[3];
};

enum : uint32_t { kMatPrincipled = 0, kMatHair = 1 };

struct alignas(16) Material {
  uint32_t kind;
  uint32_t textured;    // kMatPrincipled with map_base_color / map_subsurface_color: ParamToBsdf runs per hit
  uint32_t pad[2];
  PrincipledBsdf bsdf;  // kMatPrincipled without textures: ParamToBsdf hoisted to commit time
  HairBsdf hair;        // kMatHair: everything except h (= hit v), hair-shader.cc:100-151
  PrincipledParam param;  // raw parameters (used when textured)
  // kMatPrincipled without textures: the medium of the random walk, from the closure set at commit (dshade.h::medium_coefficients;
  // host and device evaluate the same f64r exp): sigma_t, sigma_s, first walk throughput -- nine consecutive words
  V3 sss_sigt, sss_sigs, sss_wthr;
  uint32_t pad2[3];
};
static_assert(sizeof(Material) % 16 == 0, "material records are read with 16-byte loads");

struct TexDesc {  // pbrlab::Texture (src/texture.h:13-44): float pixels, row-major, interleaved channels
  uint32_t offset, width, height, channels;
};

struct alignas(16) LightRec {  // one per (area light, primitive) ; light-manager.h:79-170
  float p0[3], pdf;            // pdf = P(light) * P(prim) * 1/area  (float, in that order)
  float p1[3], pad0;
  float p2[3], pad1;
  float normal[3], pad2;  // CalcGeometryNormal (triangle-mesh.cc:181-184)
  float emission[3], pad3;
};

struct LightHead {
  uint32_t first, count;  // range in LightRec / prim cdf
};

// Round 6: where the rays of a random walk start (k_sss_walk).  A walk's ray is a short segment inside its instance and two node visits
// long, the first of which is always the root.  Per instance, at commit: the CUT of the Q tree that covers everything a segment inside the
// instance's bounds can meet -- every reference (inner node or leaf) whose box meets the instance's bounds widened by a margin, found by
// descending from the root wherever that drops a child; one inner node of it (the one that holds the instance) is the ENTRY, the others
// -- primitives of OTHER instances inside the bounds: the floor under a statue; a hit on them ends the walk in the reference,
// random-walk-sss.h:371-384 -- are FOREIGN references with the boxes their parents present.  A ray whose end points lie inside `lo / hi`
// (the bounds widened by the smaller margin) starts at the entry, with the foreign references its interval meets (the tree's own
// conservative slab test on those very boxes) on its stack; any other ray starts at the root.  Sound because a hit is accepted only inside
// the ray's interval through the primitive's own box (dtrace.h): a primitive whose box is more than the margin away from every point of
// the segment cannot be hit -- and the closest of what is left is the closest (hits do not depend on the visiting order).
constexpr uint32_t kSssMaxForeign = 7;
struct alignas(16) SssEntry {
  float lo[3];
  uint32_t entry;     // index of the Q node the walk's rays start at; 0 with nforeign = 0: the root, always
  float hi[3];
  uint32_t nforeign;
  struct {
    float lo[3];
    uint32_t ref;
    float hi[3];
    uint32_t pad;
  } foreign[kSssMaxForeign];
};
static_assert(sizeof(SssEntry) == 16 * (2 + 2 * kSssMaxForeign), "SssEntry is read as 16-byte words");

struct DScene {
  const BvhNode* nodes;
  const float4* slots;
  const ShadeRec* shade;
  const Material* materials;
  const float* light_cdf;
  const LightHead* light_heads;
  const float* lprim_cdf;
  const LightRec* lrecs;
  const BvhNode* light_boxes;  // bounding boxes of the lights (emissive meshes), two per node, stored like BVH boxes
  const float* tex_pixels;
  const TexDesc* textures;
  uint32_t num_nodes, num_slots, num_lights, num_materials, num_curves, num_textures, num_lrecs;
  const float4* wide;           // Q tree (4-wide, quantised): wide_nodes x QNode, its triangle leaves (TriPair, 80 B), the points of the
                                // curve pieces (16 B); or null
  uint32_t wide_nodes;
  uint32_t q_tri0, q_pt0;       // 16-byte index of triangle leaf record 0 / point 0 in `wide`
  const uint32_t* q_hitcode;    // per point p: the hit code of piece p (slot | routing bits)
  uint32_t top_nodes;           // nodes 0 .. top_nodes-1 are the breadth-first top of the tree (<= kTopNodes; 0: not numbered that way)
  uint32_t wide_top_nodes;      // the same for the Q tree
  uint32_t lights_transformed;  // an emissive instance has a transform: lrecs / light_boxes are not what the raytracer sees
  const SssEntry* sss_entries;  // per instance (num_sss_entries), or null: where its random walks' rays start
  uint32_t num_sss_entries;
};

// camera of RenderingTile (render.cc:132-158), derived on the host from the scene AABB
struct Camera {
  float org[3];
  float x_corner, y_corner, z_corner, dx, dy;
};

}  // namespace pb
