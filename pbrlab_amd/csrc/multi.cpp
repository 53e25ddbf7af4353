// multi.cpp -- multi-GPU side of the C ABI (include/pbrhip.h, "multi-GPU"): the RenderLayer exchange.
//
// The reference has one process and a std::thread pool over tile jobs (render.cc:203-238); here a frame is split over
// GPUs by pixel blocks (block index % world == rank) and the only exchange is the framebuffer at the end.
//   * one process, several GPUs (pbrhip_render_multi): one host thread per device, shards moved with HIP peer copies
//     over xGMI, no communicator needed;
//   * one process per GPU (pbrhip_comm_*): RCCL, resolved at run time with dlopen so that single-GPU users of
//     libpbrhip.so do not need librccl (and so that a process that already carries an RCCL -- PyTorch -- shares it).
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: every call goes through the table below
#include <string.h>

#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>

#include "scene_impl.h"

using namespace pb;

// ------------------------------------------------------------------ RCCL, loaded on first use
namespace {
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};
Rccl g_rccl;
Rccl* rccl() {
  Rccl& r = g_rccl;
  static std::once_flag once;
  std::call_once(once, [&r]() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle) {
      const char* e = dlerror();
      r.error = std::string("cannot load librccl: ") + (e ? e : "?");
      return;
    }
    bool ok = true;
    auto sym = [&](const char* name) {
      void* p = dlsym(r.handle, name);
      if (!p) ok = false, r.error = std::string("librccl lacks ") + name;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.Reduce = (decltype(r.Reduce))sym("ncclReduce");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) r.handle = nullptr;
  });
  return r.handle ? &r : nullptr;
}
int rccl_missing() { return fail(PBRHIP_ECOMM, "RCCL unavailable: %s", g_rccl.error.c_str()); }
}  // namespace

#define NCCLCHK(R, expr)                                                                                          \
  do {                                                                                                            \
    ncclResult_t r_ = (expr);                                                                                     \
    if (r_ != ncclSuccess) return fail(PBRHIP_ECOMM, "%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

// Calls between GroupStart and GroupEnd: a failure is recorded and the group is ALWAYS closed before the function returns
// (the thread's group depth is shared with every other user of this librccl -- PyTorch included: a group left open would
// queue their later collectives and never launch them).
#define NCCLGRP(R, first, expr)                 \
  do {                                          \
    if ((first) == ncclSuccess) (first) = (expr); \
  } while (0)
#define NCCLGRP_END(R, first, what)                                                                                         \
  do {                                                                                                                      \
    ncclResult_t e_ = (R)->GroupEnd();                                                                                      \
    if ((first) == ncclSuccess) (first) = e_;                                                                               \
    if ((first) != ncclSuccess) return fail(PBRHIP_ECOMM, "%s failed: %s (%s:%d)", what, (R)->GetErrorString(first), __FILE__, __LINE__); \
  } while (0)

struct pbrhip_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
};

static_assert(PBRHIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" int pbrhip_comm_unique_id(unsigned char id[PBRHIP_COMM_ID_BYTES]) {
  return guarded([&]() -> int {
    if (!id) return fail(PBRHIP_EINVAL, "id is NULL");
    Rccl* R = rccl();
    if (!R) return rccl_missing();
    ncclUniqueId u;
    NCCLCHK(R, R->GetUniqueId(&u));
    memcpy(id, u.internal, PBRHIP_COMM_ID_BYTES);
    return PBRHIP_OK;
  });
}

extern "C" int pbrhip_comm_create(pbrhip_comm** out, const unsigned char id[PBRHIP_COMM_ID_BYTES], int rank, int world) {
  return guarded([&]() -> int {
    if (!out || !id) return fail(PBRHIP_EINVAL, "comm_create: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(PBRHIP_EINVAL, "comm_create: rank %d of %d", rank, world);
    Rccl* R = rccl();
    if (!R) return rccl_missing();
    std::unique_ptr<pbrhip_comm> c(new pbrhip_comm());
    c->rank = rank, c->world = world, c->device = current_device();
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    ncclUniqueId u;
    memcpy(u.internal, id, PBRHIP_COMM_ID_BYTES);
    NCCLCHK(R, R->CommInitRank(&c->comm, world, u, rank));
    *out = c.release();
    return PBRHIP_OK;
  });
}

extern "C" int pbrhip_comm_destroy(pbrhip_comm* c) {
  return guarded([&]() -> int {
    if (!c) return PBRHIP_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    Rccl* R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PBRHIP_OK;
  });
}

extern "C" int pbrhip_comm_reduce_layer(pbrhip_comm* c, float* d_rgba, uint32_t* d_count, size_t num_pixels, int root) {
  return guarded([&]() -> int {
    if (!c || !d_rgba || !d_count) return fail(PBRHIP_EINVAL, "reduce_layer: NULL argument");
    if (root < 0 || root >= c->world) return fail(PBRHIP_EINVAL, "reduce_layer: root %d of %d", root, c->world);
    Rccl* R = rccl();
    if (!R) return rccl_missing();
    HIPCHK(hipSetDevice(c->device));
    // one ncclReduce(sum) over rgba (w*h*4 f32) and one over count (u32), fused in one group
    NCCLCHK(R, R->GroupStart());
    ncclResult_t first = ncclSuccess;
    NCCLGRP(R, first, R->Reduce(d_rgba, d_rgba, num_pixels * 4, ncclFloat32, ncclSum, root, c->comm, c->stream));
    NCCLGRP(R, first, R->Reduce(d_count, d_count, num_pixels, ncclUint32, ncclSum, root, c->comm, c->stream));
    NCCLGRP_END(R, first, "ncclReduce of the layer");
    HIPCHK(hipStreamSynchronize(c->stream));
    return PBRHIP_OK;
  });
}

// ------------------------------------------------------------------ shards
namespace {
// words (floats) of the shard of npix pixels, padded so that consecutive shards in one buffer stay 16-byte aligned
size_t shard_words(size_t npix) { return (5 * npix + 3) & ~(size_t)3; }

// Pixel lists of the ranks [lo, hi) except `skip` of a `world`-rank job, concatenated in s->xchg_pix (cached: the lists
// only depend on the image size and the dealing); s->xk_off[r - lo] / xk_cnt[r - lo] = first entry / size of rank r's list
int ensure_peer_pixels(pbrhip_scene* s, hipStream_t st, uint32_t w, uint32_t h, uint32_t world, uint32_t block,
                       uint32_t lo, uint32_t hi, uint32_t skip) {
  if (block == 0) block = 64;
  const uint32_t key[7] = {w, h, world, block, lo, hi, skip};
  if (s->xchg_pix.p && memcmp(key, s->xk_key, sizeof(key)) == 0) return PBRHIP_OK;
  std::vector<uint32_t> all, one;
  s->xk_off.assign(hi - lo, 0), s->xk_cnt.assign(hi - lo, 0);
  for (uint32_t r = lo; r < hi; r++) {
    if (r == skip) continue;
    shard_pixels(w, h, r, world, block, &one);
    s->xk_off[r - lo] = all.size(), s->xk_cnt[r - lo] = one.size();
    all.insert(all.end(), one.begin(), one.end());
  }
  HIPCHK(s->xchg_pix.upload(all, st));
  HIPCHK(hipStreamSynchronize(st));
  memcpy(s->xk_key, key, sizeof(key));
  return PBRHIP_OK;
}
}  // namespace

extern "C" int pbrhip_comm_gather_layer(pbrhip_comm* c, pbrhip_scene* s, const pbrhip_render_desc* d, float* d_rgba,
                                        uint32_t* d_count, int root) {
  return guarded([&]() -> int {
    if (!c || !s || !d || !d_rgba || !d_count) return fail(PBRHIP_EINVAL, "gather_layer: NULL argument");
    if (root < 0 || root >= c->world) return fail(PBRHIP_EINVAL, "gather_layer: root %d of %d", root, c->world);
    const uint32_t world = d->tile_world ? d->tile_world : 1;
    if ((int)world != c->world || (int)d->tile_rank != c->rank)
      return fail(PBRHIP_EINVAL, "gather_layer: desc is rank %u of %u, the communicator rank %d of %d", d->tile_rank, world, c->rank, c->world);
    if (s->device != c->device) return fail(PBRHIP_EINVAL, "gather_layer: scene on device %d, communicator on %d", s->device, c->device);
    if (c->world == 1) return PBRHIP_OK;
    Rccl* R = rccl();
    if (!R) return rccl_missing();
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    if (c->rank != root) {
      if (int rc = ensure_pixels(s, d->width, d->height, d->tile_rank, world, d->shard_block)) return rc;
      const uint32_t npix = s->pk_npix;
      HIPCHK(s->xchg_send.reserve(shard_words(npix)));
      if (npix == 0) return PBRHIP_OK;  // (the root posts no receive for a rank without pixels either)
      launch_layer_pack(st, s->pix_index.p, npix, d_rgba, d_count, s->xchg_send.p);
      HIPCHK(hipGetLastError());
      NCCLCHK(R, R->Send(s->xchg_send.p, 5 * (size_t)npix, ncclFloat32, root, c->comm, st));
      HIPCHK(hipStreamSynchronize(st));
      return PBRHIP_OK;
    }
    if (int rc = ensure_peer_pixels(s, st, d->width, d->height, world, d->shard_block, 0, world, (uint32_t)root)) return rc;
    std::vector<size_t> woff(world, 0);
    size_t words = 0;
    for (uint32_t r = 0; r < world; r++) woff[r] = words, words += shard_words(s->xk_cnt[r]);
    HIPCHK(s->xchg_recv.reserve(words));
    // every shard arrives over its own xGMI link: post all receives in one group
    NCCLCHK(R, R->GroupStart());
    ncclResult_t first = ncclSuccess;
    for (uint32_t r = 0; r < world; r++)
      if ((int)r != root && s->xk_cnt[r]) NCCLGRP(R, first, R->Recv(s->xchg_recv.p + woff[r], 5 * s->xk_cnt[r], ncclFloat32, (int)r, c->comm, st));
    NCCLGRP_END(R, first, "ncclRecv of the shards");
    for (uint32_t r = 0; r < world; r++)
      if ((int)r != root)
        launch_layer_unpack_add(st, s->xchg_pix.p + s->xk_off[r], (uint32_t)s->xk_cnt[r], s->xchg_recv.p + woff[r], d_rgba, d_count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return PBRHIP_OK;
  });
}

// ------------------------------------------------------------------ one process, several devices
template <typename T>
static int copy_buf(DevBuf<T>& dst, int dst_dev, const DevBuf<T>& src, int src_dev) {
  HIPCHK(dst.reserve(src.n));
  if (src.n) HIPCHK(hipMemcpyPeer(dst.p, dst_dev, src.p, src_dev, src.n * sizeof(T)));
  return PBRHIP_OK;
}

extern "C" int pbrhip_scene_replicate(const pbrhip_scene* src, int device, pbrhip_scene** out) {
  return guarded([&]() -> int {
    if (!src || !out) return fail(PBRHIP_EINVAL, "scene_replicate: NULL argument");
    if (!src->committed) return fail(PBRHIP_ESTATE, "scene not committed");
    int ndev = 0;
    if (int rc = pbrhip_device_count(&ndev)) return rc;
    if (device < 0 || device >= ndev) return fail(PBRHIP_EINVAL, "device %d out of range (%d devices)", device, ndev);
    const int keep = current_device();
    if (int rc = pbrhip_set_device(device)) return rc;
    pbrhip_scene* s = nullptr;
    int rc = pbrhip_scene_create(&s);
    (void)pbrhip_set_device(keep);
    if (rc) return rc;
    std::unique_ptr<pbrhip_scene, int (*)(pbrhip_scene*)> guard(s, pbrhip_scene_destroy);
    // host side: what a committed scene still needs (material edits, bounds, flags); geometry stays with `src`
    s->materials = src->materials, s->light_params = src->light_params, s->tex_descs = src->tex_descs;
    s->lights = src->lights, s->light_cdf = src->light_cdf;
    s->has_hair = src->has_hair, s->has_sss = src->has_sss, s->has_textured = src->has_textured;
    memcpy(s->bmin, src->bmin, sizeof(s->bmin)), memcpy(s->bmax, src->bmax, sizeof(s->bmax));
    s->bvh_depth = src->bvh_depth, s->bvh_builder = src->bvh_builder, s->bvh_built_on_gpu = src->bvh_built_on_gpu;
    HIPCHK(hipSetDevice(device));
    if (device != src->device) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can) {
        hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      }
    }
    if (int r = copy_buf(s->d_nodes, device, src->d_nodes, src->device)) return r;
    if (int r = copy_buf(s->d_wide, device, src->d_wide, src->device)) return r;
    if (int r = copy_buf(s->d_qhit, device, src->d_qhit, src->device)) return r;
    if (int r = copy_buf(s->d_shade, device, src->d_shade, src->device)) return r;
    if (int r = copy_buf(s->d_materials, device, src->d_materials, src->device)) return r;
    if (int r = copy_buf(s->d_light_cdf, device, src->d_light_cdf, src->device)) return r;
    if (int r = copy_buf(s->d_lprim_cdf, device, src->d_lprim_cdf, src->device)) return r;
    if (int r = copy_buf(s->d_tex_pixels, device, src->d_tex_pixels, src->device)) return r;
    if (int r = copy_buf(s->d_tex_descs, device, src->d_tex_descs, src->device)) return r;
    if (int r = copy_buf(s->d_heads, device, src->d_heads, src->device)) return r;
    if (int r = copy_buf(s->d_lrecs, device, src->d_lrecs, src->device)) return r;
    if (int r = copy_buf(s->d_light_boxes, device, src->d_light_boxes, src->device)) return r;
    if (int r = copy_buf(s->d_sss_entries, device, src->d_sss_entries, src->device)) return r;
    HIPCHK(hipDeviceSynchronize());
    DScene& dd = s->dscene;
    dd = src->dscene;
    dd.nodes = s->d_nodes.p, dd.slots = reinterpret_cast<const float4*>(s->d_nodes.p + dd.num_nodes), dd.shade = s->d_shade.p;
    dd.wide = src->dscene.wide ? s->d_wide.p : nullptr;
    dd.q_hitcode = src->dscene.wide ? s->d_qhit.p : nullptr;
    dd.materials = s->d_materials.p, dd.light_cdf = s->d_light_cdf.p, dd.light_heads = s->d_heads.p;
    dd.lprim_cdf = s->d_lprim_cdf.p, dd.lrecs = s->d_lrecs.p, dd.light_boxes = s->d_light_boxes.p;
    dd.sss_entries = src->dscene.sss_entries ? s->d_sss_entries.p : nullptr;
    dd.tex_pixels = s->d_tex_pixels.p, dd.textures = s->d_tex_descs.p;
    s->committed = true;
    *out = guard.release();
    return PBRHIP_OK;
  });
}

extern "C" int pbrhip_render_multi(pbrhip_scene* const* scenes, uint32_t n, const pbrhip_render_desc* d,
                                   const volatile unsigned char* cancel, float* rgba, uint32_t* count,
                                   size_t* finish_pass, pbrhip_render_stats* stats) {
  return guarded([&]() -> int {
    if (!scenes || !n || !d || !rgba || !count) return fail(PBRHIP_EINVAL, "render_multi: NULL argument");
    for (uint32_t i = 0; i < n; i++)
      if (!scenes[i] || !scenes[i]->committed) return fail(PBRHIP_ESTATE, "render_multi: scene %u is not committed", i);
    for (uint32_t i = 0; i < n; i++)
      for (uint32_t j = 0; j < i; j++)
        if (scenes[i] == scenes[j]) return fail(PBRHIP_EINVAL, "render_multi: scene %u is listed twice (replicate it)", i);
    if (d->width == 0 || d->height == 0 || (uint64_t)d->width * d->height >= (1ull << 32)) return fail(PBRHIP_EINVAL, "bad image size");
    const uint32_t outer_world = d->tile_world ? d->tile_world : 1;
    if (d->tile_rank >= outer_world) return fail(PBRHIP_EINVAL, "tile_rank %u >= tile_world %u", d->tile_rank, outer_world);
    if ((uint64_t)outer_world * n >= (1ull << 32)) return fail(PBRHIP_EINVAL, "too many ranks");
    auto t_begin = std::chrono::steady_clock::now();
    const size_t npx = (size_t)d->width * d->height;
    const uint32_t world = outer_world * n;
    pbrhip_scene* root = scenes[0];

    // every device renders its blocks into its own full-size, cleared layer (NO_CLEAR: the caller's layer is the
    // starting point on the first device, zeros elsewhere)
    std::vector<int> rc(n, PBRHIP_OK);
    std::vector<std::string> msg(n);
    std::vector<size_t> fin(n, 0);
    std::vector<pbrhip_render_stats> st(n);
    std::atomic<uint32_t> running(n);
    auto worker = [&](uint32_t i) {
      pbrhip_scene* s = scenes[i];
      rc[i] = guarded([&]() -> int {
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(s->own_rgba.reserve(npx * 4));
        HIPCHK(s->own_count.reserve(npx));
        pbrhip_render_desc di = *d;
        di.tile_rank = d->tile_rank * n + i, di.tile_world = world;
        if (di.shard_block == 0 && n > 1) di.shard_block = 16;  // finer than the 64 x 64 tile: evens out the devices' load
        if (i == 0 && (d->flags & PBRHIP_RENDER_NO_CLEAR)) {
          HIPCHK(hipMemcpyAsync(s->own_rgba.p, rgba, npx * 4 * sizeof(float), hipMemcpyHostToDevice, s->stream));
          HIPCHK(hipMemcpyAsync(s->own_count.p, count, npx * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
        } else {
          di.flags &= ~PBRHIP_RENDER_NO_CLEAR;
        }
        if (int r = render_impl(s, &di, cancel, s->own_rgba.p, s->own_count.p, &fin[i], &st[i])) return r;
        if (i != 0) {  // this device's shard, packed for the trip to the first device
          HIPCHK(s->xchg_send.reserve(shard_words(s->pk_npix)));
          launch_layer_pack(s->stream, s->pix_index.p, s->pk_npix, s->own_rgba.p, s->own_count.p, s->xchg_send.p);
          HIPCHK(hipGetLastError());
          HIPCHK(hipStreamSynchronize(s->stream));
        }
        return PBRHIP_OK;
      });
      if (rc[i]) msg[i] = pbrhip_last_error();
      running.fetch_sub(1);
    };
    std::vector<std::thread> threads;
    for (uint32_t i = 1; i < n; i++) threads.emplace_back(worker, i);
    std::thread first(worker, 0u);
    // progress: the passes complete on every device (render.cc:224-231 counts passes, not tile jobs)
    size_t published = 0;
    if (finish_pass) __atomic_store_n(finish_pass, (size_t)0, __ATOMIC_RELEASE);
    while (running.load() != 0) {
      size_t m = (size_t)-1;
      for (uint32_t i = 0; i < n; i++) m = std::min(m, __atomic_load_n(&fin[i], __ATOMIC_ACQUIRE));
      if (finish_pass && m > published) __atomic_store_n(finish_pass, m, __ATOMIC_RELEASE), published = m;
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    first.join();
    for (std::thread& t : threads) t.join();
    for (uint32_t i = 0; i < n; i++)
      if (rc[i]) return fail(rc[i], "render_multi: rank %u (device %d): %s", i, scenes[i]->device, msg[i].c_str());
    size_t m = (size_t)-1;
    for (uint32_t i = 0; i < n; i++) m = std::min(m, fin[i]);
    if (finish_pass) __atomic_store_n(finish_pass, m, __ATOMIC_RELEASE);

    // exchange: shard i -> first device (peer copy over xGMI; a plain device copy when the scenes share a device)
    HIPCHK(hipSetDevice(root->device));
    hipStream_t rst = root->stream;
    if (n > 1) {
      std::vector<size_t> woff(n, 0);
      size_t words = 0;
      for (uint32_t i = 1; i < n; i++) woff[i] = words, words += shard_words(scenes[i]->pk_npix);
      HIPCHK(root->xchg_recv.reserve(words));
      const uint32_t base = d->tile_rank * n;
      if (int r = ensure_peer_pixels(root, rst, d->width, d->height, world, scenes[1]->pk_block, base, base + n, base)) return r;
      const std::vector<size_t>&poff = root->xk_off;
      for (uint32_t i = 1; i < n; i++) {
        const pbrhip_scene* s = scenes[i];
        if (root->xk_cnt[i] != s->pk_npix) return fail(PBRHIP_EINVAL, "render_multi: shard %u changed size", i);
        if (!s->pk_npix) continue;
        const size_t bytes = 5 * (size_t)s->pk_npix * sizeof(float);
        if (s->device == root->device) HIPCHK(hipMemcpyAsync(root->xchg_recv.p + woff[i], s->xchg_send.p, bytes, hipMemcpyDeviceToDevice, rst));
        else HIPCHK(hipMemcpyPeerAsync(root->xchg_recv.p + woff[i], root->device, s->xchg_send.p, s->device, bytes, rst));
        launch_layer_unpack_add(rst, root->xchg_pix.p + poff[i], s->pk_npix, root->xchg_recv.p + woff[i], root->own_rgba.p, root->own_count.p);
      }
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(rgba, root->own_rgba.p, npx * 4 * sizeof(float), hipMemcpyDeviceToHost, rst));
    HIPCHK(hipMemcpyAsync(count, root->own_count.p, npx * sizeof(uint32_t), hipMemcpyDeviceToHost, rst));
    HIPCHK(hipStreamSynchronize(rst));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (stats)
      for (uint32_t i = 0; i < n; i++) stats[i] = st[i], stats[i].ms_total = i == 0 ? ms : st[i].ms_total;
    return PBRHIP_OK;
  });
}
