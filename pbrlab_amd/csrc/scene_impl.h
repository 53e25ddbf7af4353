// scene_impl.h -- what libpbrhip's translation units share behind the C ABI: the error helpers, the device-buffer
// holder and the scene object (host model + device scene + render working set).  Not part of the interface.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pbrhip.h"
#include "host_scene.h"
#include "kernels.h"

namespace pb {

// ------------------------------------------------------------------ errors
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int current_device();  // what pbrhip_set_device selected
#define HIPCHK(expr)                                                                                          \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) return pb::fail(PBRHIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                                \
  } while (0)

// Every extern "C" body runs inside this guard: an allocation failure or any other C++ exception becomes an error
// code instead of unwinding through a C / ctypes caller.
template <typename F>
static inline int guarded(F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(PBRHIP_ENOMEM, "out of host memory");
  } catch (const std::exception& e) {
    return fail(PBRHIP_EINVAL, "%s", e.what());
  } catch (...) {
    return fail(PBRHIP_EINVAL, "unknown C++ exception");
  }
}

// ------------------------------------------------------------------ device buffers
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr, n = 0;
  }
  hipError_t reserve(size_t count) {
    if (count <= n) return hipSuccess;
    release();
    hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  hipError_t upload(const std::vector<T>& h, hipStream_t s) {
    hipError_t e = reserve(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
  }
};

}  // namespace pb

struct pbrhip_scene {
  int device = 0;
  hipStream_t stream = nullptr;
  // host model
  std::vector<pb::HostMesh> meshes;
  std::vector<std::vector<uint32_t>> locals;
  std::vector<pb::HostInstance> instances;
  std::vector<pb::HostMaterial> materials;
  std::vector<pb::V3> light_params;
  std::vector<pb::TexDesc> tex_descs;  // Scene::AddTexture
  std::vector<float> tex_pixels;
  std::vector<pb::HostLight> lights;
  std::vector<float> light_cdf;
  bool committed = false, has_hair = false, has_sss = false, has_textured = false;  // has_sss: a material can enter a medium (or is textured)
  float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
  uint32_t bvh_depth = 0;
  int bvh_builder = PBRHIP_BVH_HOST_SAH;
  bool bvh_built_on_gpu = false;
  // device scene
  pb::DevBuf<pb::BvhNode> d_nodes;
  pb::DevBuf<float4> d_wide;  // the Q tree: quantised 4-wide nodes + its triangle slots + curve points (DScene::wide), host-built trees only
  pb::DevBuf<uint32_t> d_qhit;  // hit code per curve point of the Q tree (DScene::q_hitcode)
  pb::DevBuf<pb::ShadeRec> d_shade;
  pb::DevBuf<pb::Material> d_materials;
  pb::DevBuf<float> d_light_cdf, d_lprim_cdf, d_tex_pixels;
  pb::DevBuf<pb::TexDesc> d_tex_descs;
  pb::DevBuf<pb::LightHead> d_heads;
  pb::DevBuf<pb::LightRec> d_lrecs;
  pb::DevBuf<pb::BvhNode> d_light_boxes;
  pb::DevBuf<pb::SssEntry> d_sss_entries;  // DScene::sss_entries
  pb::DScene dscene;
  // render working set (grown on demand, reused across calls)
  pb::DevBuf<float4> rec, srec, ssrec, L, hit, sss_A, sh_e;  // path state (kernels.h::PathState): rec = 4 words of 16 B per path, srec = 2
  pb::DevBuf<uint32_t> q[7], counts, pix_index, path_pix, spill;  // pix_index: the rank's pixels in the shard's order (exchange); path_pix: in the order the paths are laid out in
  pb::DevBuf<unsigned long long> stats;
  pb::DevBuf<float> own_rgba;
  pb::DevBuf<uint32_t> own_count;
  pb::DevBuf<float4> hook_rays;
  pb::DevBuf<pb::HookHit> hook_hits;
  pb::DevBuf<uint8_t> hook_occ;
  uint32_t* h_counts = nullptr;            // pinned, kMaxGroups x kCntNum
  // Round 6: what the host learns about an iteration it enqueued -- written by the iteration's last kernel (k_advance) straight into
  // pinned host memory: kRingSlots slots of 4 words per group lane: live paths, pending shadow rays, overflow flag, stamp (a number
  // that is unique per scene and launch, written last).  The host polls the stamp: no copy, no stream query, and the NEXT iteration
  // is already enqueued behind this one (pbrhip.cpp::render_impl).
  uint32_t* h_ring = nullptr;              // pinned, kMaxGroups x kRingSlots x 4
  uint32_t* d_ring = nullptr;              // the same memory as the device addresses it
  uint32_t ring_stamp = 0;                 // last stamp handed out
  pb::DevBuf<uint32_t> heads;              // the heads of k_trace's ray queue: lanes x kTraceHeads x kHeadStride words
  pb::DevBuf<uint32_t> susp;               // suspend records of the resumable rays: lanes x 2 (written / read by alternate launches) x cap x kSuspWords
  std::vector<hipStream_t> group_streams;  // streams of path groups 1.. (group 0 uses `stream`)
  // pixel list cache key (ensure_pixels)
  uint32_t pk_w = 0, pk_h = 0, pk_rank = 0, pk_world = 0, pk_block = 0, pk_tile = 0, pk_npix = 0;
  std::vector<hipEvent_t> events;
  // layer exchange (multi.cpp): packed shard of this rank / staging for the shards of the others
  pb::DevBuf<float> xchg_send, xchg_recv;
  pb::DevBuf<uint32_t> xchg_pix;                  // pixel lists of the ranks whose shards arrive here, concatenated
  std::vector<size_t> xk_off, xk_cnt;             // per rank: first entry in xchg_pix / number of pixels
  uint32_t xk_key[7] = {0, 0, 0, 0, 0, 0, 0};     // w, h, world, block, first rank, end rank, skipped rank

  size_t device_bytes() const {
    return d_nodes.n * sizeof(pb::BvhNode) + d_wide.n * sizeof(float4) + d_qhit.n * 4 + d_shade.n * sizeof(pb::ShadeRec) + d_materials.n * sizeof(pb::Material) +
           d_lrecs.n * sizeof(pb::LightRec);
  }
};

namespace pb {
// pixel indices (y * w + x) of the blocks of rank `rank` in CreateTiles order (render-tile.cc:29-41 for block = 64)
void shard_pixels(uint32_t w, uint32_t h, uint32_t rank, uint32_t world, uint32_t block, std::vector<uint32_t>* out);
int ensure_pixels(pbrhip_scene* s, uint32_t w, uint32_t h, uint32_t rank, uint32_t world, uint32_t block);
// the body of pbrhip_render_device (device pointers on the scene's device)
int render_impl(pbrhip_scene* s, const pbrhip_render_desc* d, const volatile unsigned char* cancel, float* d_rgba,
                uint32_t* d_count, size_t* finish_pass, pbrhip_render_stats* stats);
}  // namespace pb
