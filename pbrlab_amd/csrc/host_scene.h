// host_scene.h -- host-side scene model kept by libpbrhip (the "scene upload / BVH flatten / tile
// dispatch stay in C++" part of the design).  Mirrors the read side of pbrlab's Scene
// (src/scene.h:93-110), LightManager (src/light-manager.h:172-193) and Raytracer facade.
#pragma once

#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "dshade.h"

namespace pb {

struct HostMesh {
  int kind = 0;  // 0 triangle mesh, 1 cubic Bezier curve mesh (mesh/mesh.h:23)
  // triangles (mesh/attribute.h, mesh/triangle-mesh.h)
  std::vector<float> vertices, normals, texcoords;  // xyzw, xyzw, uv
  std::vector<uint32_t> vid, nid, tid, mat;
  uint32_t nfaces = 0;
  // curves (mesh/cubic-bezier-curve-mesh.h)
  std::vector<float> cverts;  // xyz + radius
  std::vector<uint32_t> cidx, cmat;
  uint32_t num_prims() const { return kind == 0 ? nfaces : (uint32_t)cidx.size(); }
};

struct HostAreaLight {  // LightManager::AreaLight (light-manager.h:174-182)
  std::vector<uint32_t> light_param_ids;
  std::vector<float> choose_prob, cdf, area_pdf;
  float intensity_sum = 0.f;
  uint32_t global_id = kNone;
};

struct HostInstance {  // MeshInstance (mesh-instance.h:22-36)
  uint32_t local_scene = 0;
  float xf[16];
  bool identity = true;  // xf is bit-for-bit the identity: the raytracer sees the meshes as they are
  std::vector<std::vector<uint32_t>> material_ids, light_ids;
  std::vector<int> has_area_light;  // per geom
  std::vector<HostAreaLight> area_lights;
};

struct HostMaterial {
  uint32_t kind = kMatPrincipled;
  PrincipledParam pr;
  HairParam hr;
};

struct HostLight {  // LightManager::Light (light-manager.h:184-188)
  float choose_prob = 0.f;
  uint32_t instance_id = 0, geom_id = 0;
};

// canonical primitive reference; index in the flattened list = gid
// One traversal primitive.  A triangle is one; a cubic Bezier curve contributes FOUR, one per linear piece of its flat
// ribbon (sub = 0..3, curve parameter [sub/4, (sub+1)/4]): tight boxes instead of one box around the whole cubic, and a
// quarter of the intersection work per test.  The index of a PrimRef in canonical (instance, geom, prim, sub) order
// is the id that breaks ties between equal hit distances.
struct PrimRef {
  uint32_t instance_id, geom_id, prim_id, kind, sub;
};

struct FlatBvh {
  std::vector<BvhNode> nodes;
  std::vector<uint32_t> slot_gid;  // leaf order -> gid
  uint32_t depth = 0;
};

// Binned-SAH BVH2 over primitive boxes; leaves hold <= kMaxLeaf primitives of ONE kind.
// boxes: lo/hi per primitive (3 floats each); kinds: 0 triangle / 1 curve.
void build_bvh(const std::vector<float>& lo, const std::vector<float>& hi, const std::vector<uint8_t>& kinds,
               FlatBvh* out);

// The binary tree collapsed to four children per node with quantised boxes (dscene.h::QNode).  Which descendants of a
// binary node become the children of its Q node -- a frontier of at most four subtrees below it -- is chosen bottom-up by
// dynamic programming over the surface-area cost WITH THE BOX A CHILD REALLY PRESENTS: its box rounded outwards on the node's
// 8-bit grid (a thin primitive in a big node is a thick slab).
// map_leaf turns a leaf reference of the binary tree into one or two children of the Q tree (its references use the Q
// tree's own triangle slots and curve points; a curve leaf whose two pieces are not neighbours in a chain becomes two):
// it fills ref / lo / hi and returns the count.  PRECONDITION on lo / hi (QChild): they are boxes as the binary tree stores
// them, i.e. already widened with BvhNode::widen_lo / widen_hi -- for a curve leaf that map_leaf splits, the box of each piece
// (end points +- the larger radius) widened the same way; quantise_node only rounds outwards from there, and the
// traversal's exactness argument needs every stored box to contain the validation boxes below it.
// Returns the EXACT stack need of a near-first traversal of the Q tree: the maximum over root-to-leaf paths of the sum of
// (children - 1) of the nodes on the path (a node pushes all hit children but the nearest).
struct QChild {
  uint32_t ref;
  float lo[3], hi[3];
};
uint32_t build_qtree(const std::vector<BvhNode>& nodes, const std::function<int(uint32_t, const float*, const float*, QChild*)>& map_leaf,
                     std::vector<QNode>* out);

// The same tree format built on the GPU (bvh_gpu.hip: Morton-order linear BVH).  nodes_out: DEVICE array of
// max(n - 1, 1) nodes; order_out: slot -> primitive index; depth_out: traversal stack depth needed.
hipError_t build_bvh_gpu(hipStream_t st, const std::vector<float>& lo, const std::vector<float>& hi,
                         const std::vector<uint8_t>& kinds, BvhNode* nodes_out, std::vector<uint32_t>* order_out,
                         uint32_t* depth_out);

}  // namespace pb
