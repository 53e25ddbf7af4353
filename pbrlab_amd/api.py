"""Host-side mirror of pbrlab's drop-in boundary over the C ABI of libpbrhip.so (include/pbrhip.h).

`Scene` keeps the builder method names of `pbrlab::Scene` (src/scene.h:19-91), `RenderLayer` the
fields of `pbrlab::RenderLayer` (src/render-layer.h:11-26) and `Render` the argument list of
`pbrlab::Render` (src/render.h:14-17).  Python is only the binding layer here: all work happens in
the HIP library; nothing in this module computes pixels and nothing falls back to the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib

NONE = 0xFFFFFFFF
fp = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)

RENDER_STATS, RENDER_TIMING, RENDER_NO_CLEAR, RENDER_TIMING_TRACE = 1, 2, 4, 8


class PrincipledParam(C.Structure):
    """pbrhip_principled_param == CyclesPrincipledBsdfParameter (src/material-param.h:24-49)."""
    _fields_ = [("base_color", C.c_float * 3), ("subsurface", C.c_float),
                ("subsurface_radius", C.c_float * 3), ("subsurface_color", C.c_float * 3),
                ("metallic", C.c_float), ("specular", C.c_float), ("specular_tint", C.c_float),
                ("roughness", C.c_float), ("anisotropic", C.c_float), ("anisotropic_rotation", C.c_float),
                ("sheen", C.c_float), ("sheen_tint", C.c_float), ("clearcoat", C.c_float),
                ("clearcoat_roughness", C.c_float), ("ior", C.c_float), ("transmission", C.c_float),
                ("transmission_roughness", C.c_float), ("base_color_tex_id", C.c_uint32),
                ("subsurface_color_tex_id", C.c_uint32)]


class HairParam(C.Structure):
    """pbrhip_hair_param == HairBsdfParameter (src/material-param.h:51-72)."""
    _fields_ = [("coloring_hair", C.c_uint32), ("base_color", C.c_float * 3), ("melanin", C.c_float),
                ("melanin_redness", C.c_float), ("melanin_randomize", C.c_float), ("roughness", C.c_float),
                ("azimuthal_roughness", C.c_float), ("ior", C.c_float), ("shift", C.c_float),
                ("specular_tint", C.c_float * 3), ("second_specular_tint", C.c_float * 3),
                ("transmission_tint", C.c_float * 3)]


class RenderDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("num_sample", C.c_uint32),
                ("first_pass", C.c_uint32), ("seed_seq", C.c_uint64), ("tile_rank", C.c_uint32),
                ("tile_world", C.c_uint32), ("max_paths_in_flight", C.c_uint32), ("flags", C.c_uint32),
                ("num_streams", C.c_uint32), ("tail_paths", C.c_uint32), ("shard_block", C.c_uint32)]


BVH_HOST_SAH, BVH_GPU_LBVH = 0, 1


class RenderStats(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in ("samples", "iterations", "chunks", "closest_rays", "closest_nodes",
                                           "closest_tris", "closest_curves", "shadow_rays", "shadow_nodes",
                                           "shadow_tris", "shadow_curves")] +
                [(n, C.c_double) for n in ("ms_generate", "ms_trace_closest", "ms_surface", "ms_shade_principled",
                                           "ms_shade_hair", "ms_sss_step", "ms_tail", "ms_accumulate", "ms_compact")] +
                [(n, C.c_uint64) for n in ("n_trace_closest", "n_tail", "n_surface", "n_shade_principled",
                                           "n_shade_hair", "n_sss_step")] + [("ms_total", C.c_double)] +
                [(n, C.c_uint64) for n in ("tail_closest_rays", "tail_shadow_rays", "pruned_rays", "passes_done", "node_bytes", "curve_bytes", "suspended_rays")] +
                [("ms_host_idle", C.c_double)])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


RAY_DT = np.dtype([("org", "<f4", 3), ("tmin", "<f4"), ("dir", "<f4", 3), ("tmax", "<f4")])
HIT_DT = np.dtype([("normal_g", "<f4", 3), ("t", "<f4"), ("u", "<f4"), ("v", "<f4"),
                   ("instance_id", "<u4"), ("geom_id", "<u4"), ("prim_id", "<u4")])


class PbrHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pbrhip error {code}: {msg}")
        self.code = code


def _chk(rc):
    if rc != 0:
        msg = _lib.lib().pbrhip_last_error()
        raise PbrHipError(rc, msg.decode() if msg else "")


def _ptr(a, t=fp):
    return None if a is None else a.ctypes.data_as(t)


def _fill(struct, d):
    for k, _ in struct._fields_:
        v = d[k]
        if isinstance(v, (tuple, list, np.ndarray)):
            setattr(struct, k, (C.c_float * 3)(*[float(x) for x in v]))
        else:
            setattr(struct, k, v)
    return struct


def make_principled(d):
    return _fill(PrincipledParam(), d)


def make_hair(d):
    return _fill(HairParam(), d)


LEAF_RNG, LEAF_FASTMATH, LEAF_FRESNEL, LEAF_MIS, LEAF_LAMBERT, LEAF_SPHERE, LEAF_TRIANGLE, LEAF_GGX_EVAL, LEAF_GGX_SAMPLE, LEAF_HAIR_EVAL, LEAF_HAIR_SAMPLE = range(11)


def leaf_eval(op, inputs, out_words):
    """pbrhip_leaf_eval (include/pbrhip.h): the device's leaf functions on an (n, in_words) array of float32 inputs (integers as their
    bits) -> (n, out_words) float32.  A test hook."""
    a = np.ascontiguousarray(inputs, np.float32)
    if a.ndim != 2:
        raise ValueError("inputs: (n, in_words)")
    out = np.zeros((a.shape[0], out_words), np.float32)
    _chk(_lib.lib().pbrhip_leaf_eval(C.c_uint32(op), C.c_void_p(a.ctypes.data), C.c_size_t(a.shape[0]), C.c_uint32(a.shape[1]),
                                     C.c_void_p(out.ctypes.data), C.c_uint32(out_words)))
    return out


def math_mode():
    """'glibcf' (glibc's float functions restated bit for bit: the default) or 'f64r' (correctly rounded): pbrhip_math_mode()"""
    return {1: "f64r", 2: "glibcf"}[int(_lib.lib().pbrhip_math_mode())]


def device_count():
    n = C.c_int(0)
    _lib.lib().pbrhip_device_count(C.byref(n))
    return n.value


def set_device(i):
    _chk(_lib.lib().pbrhip_set_device(int(i)))


def create_tiles(width, height):
    """CreateTiles (src/render-tile.cc:29-41): (n,4) array of sx,tx,sy,ty."""
    L = _lib.lib()
    n = C.c_uint32(0)
    _chk(L.pbrhip_create_tiles(width, height, None, C.byref(n)))
    out = np.zeros((n.value, 4), np.uint32)
    _chk(L.pbrhip_create_tiles(width, height, _ptr(out, u32p), C.byref(n)))
    return out


class RenderLayer:
    """pbrlab::RenderLayer (src/render-layer.h:11-26): rgba = sum of radiance (A = sample count), count."""

    def __init__(self, w=0, h=0):
        self.Resize(w, h)
        self.Clear()

    def Resize(self, w, h):
        self.width, self.height = int(w), int(h)
        self.rgba = np.zeros((self.height, self.width, 4), np.float32)
        self.count = np.zeros((self.height, self.width), np.uint32)

    def Clear(self):
        self.rgba[...] = 0
        self.count[...] = 0


class Scene:
    """pbrlab::Scene (src/scene.h:14-111) over libpbrhip."""

    def __init__(self):
        self.L = _lib.lib()
        h = C.c_void_p()
        _chk(self.L.pbrhip_scene_create(C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.pbrhip_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def AddTriangleMesh(self, vertices, normals, texcoords, vertex_ids, normal_ids=None, texcoord_ids=None,
                        material_ids=None):
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 4)
        n = np.ascontiguousarray(normals if normals is not None else np.zeros((0, 4)), np.float32).reshape(-1, 4)
        t = np.ascontiguousarray(texcoords if texcoords is not None else np.zeros((0, 2)), np.float32).reshape(-1, 2)
        vid = np.ascontiguousarray(vertex_ids, np.uint32).reshape(-1, 3)
        nid = None if normal_ids is None else np.ascontiguousarray(normal_ids, np.uint32).reshape(-1, 3)
        tid = None if texcoord_ids is None else np.ascontiguousarray(texcoord_ids, np.uint32).reshape(-1, 3)
        mid = None if material_ids is None else np.ascontiguousarray(material_ids, np.uint32).reshape(-1)
        out = C.c_uint32()
        _chk(self.L.pbrhip_scene_add_triangle_mesh(self.h, _ptr(v), len(v), _ptr(n), len(n), _ptr(t), len(t),
                                                   _ptr(vid, u32p), _ptr(nid, u32p), _ptr(tid, u32p),
                                                   _ptr(mid, u32p), len(vid), C.byref(out)))
        return out.value

    def AddCubicBezierCurveMesh(self, vertices_xyzr, indices, material_ids=None):
        v = np.ascontiguousarray(vertices_xyzr, np.float32).reshape(-1, 4)
        idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        mid = None if material_ids is None else np.ascontiguousarray(material_ids, np.uint32).reshape(-1)
        out = C.c_uint32()
        _chk(self.L.pbrhip_scene_add_curve_mesh(self.h, _ptr(v), len(v), _ptr(idx, u32p), _ptr(mid, u32p), len(idx),
                                                C.byref(out)))
        return out.value

    def AddMaterialParam(self, p):
        out = C.c_uint32()
        if isinstance(p, PrincipledParam):
            _chk(self.L.pbrhip_scene_add_principled_material(self.h, C.byref(p), C.byref(out)))
        elif isinstance(p, HairParam):
            _chk(self.L.pbrhip_scene_add_hair_material(self.h, C.byref(p), C.byref(out)))
        else:
            raise TypeError("material must be PrincipledParam or HairParam")
        return out.value

    def UpdateMaterialParam(self, material_id, p):
        """what EditQueue::EditAndPopAll does between renders (pc/pc-common.cc:57-84)."""
        if isinstance(p, PrincipledParam):
            _chk(self.L.pbrhip_scene_update_principled_material(self.h, material_id, C.byref(p)))
        else:
            _chk(self.L.pbrhip_scene_update_hair_material(self.h, material_id, C.byref(p)))

    def AddTexture(self, pixels):
        """Scene::AddTexture: pixels (H, W, C) float32."""
        px = np.ascontiguousarray(pixels, np.float32)
        if px.ndim == 2:
            px = px[..., None]
        out = C.c_uint32()
        _chk(self.L.pbrhip_scene_add_texture(self.h, _ptr(px), px.shape[1], px.shape[0], px.shape[2], C.byref(out)))
        return out.value

    def AddLightParam(self, emission):
        e = np.ascontiguousarray(emission, np.float32).reshape(3)
        out = C.c_uint32()
        _chk(self.L.pbrhip_scene_add_area_light(self.h, _ptr(e), C.byref(out)))
        return out.value

    def CreateLocalScene(self):
        out = C.c_uint32()
        _chk(self.L.pbrhip_scene_create_local_scene(self.h, C.byref(out)))
        return out.value

    def AddMeshToLocalScene(self, local_id, mesh_id):
        out = C.c_uint32()
        _chk(self.L.pbrhip_scene_add_mesh_to_local_scene(self.h, local_id, mesh_id, C.byref(out)))
        return out.value

    def CreateInstance(self, local_id, transform=None):
        t = None if transform is None else np.ascontiguousarray(transform, np.float32).reshape(16)
        out = C.c_uint32()
        _chk(self.L.pbrhip_scene_create_instance(self.h, local_id, _ptr(t), C.byref(out)))
        return out.value

    def AttachLightParamIdsToInstance(self, instance_id, ids_per_geom):
        for g, ids in enumerate(ids_per_geom):
            a = np.ascontiguousarray(ids, np.uint32).reshape(-1)
            _chk(self.L.pbrhip_scene_attach_light_ids(self.h, instance_id, g, _ptr(a, u32p), len(a)))

    def AttachMaterialParamIdsToInstance(self, instance_id, ids_per_geom):
        for g, ids in enumerate(ids_per_geom):
            a = np.ascontiguousarray(ids, np.uint32).reshape(-1)
            _chk(self.L.pbrhip_scene_attach_material_ids(self.h, instance_id, g, _ptr(a, u32p), len(a)))

    def CommitScene(self):
        _chk(self.L.pbrhip_scene_commit(self.h))

    def SetBvhBuilder(self, builder):
        """BVH_HOST_SAH (default) or BVH_GPU_LBVH; before CommitScene (pbrhip_scene_set_bvh_builder)"""
        _chk(self.L.pbrhip_scene_set_bvh_builder(self.h, int(builder)))

    def FetchSceneAABB(self):
        lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
        _chk(self.L.pbrhip_scene_aabb(self.h, _ptr(lo), _ptr(hi)))
        return lo, hi

    def info(self):
        nn, ns, nb = C.c_uint64(), C.c_uint64(), C.c_uint64()
        dp = C.c_uint32()
        _chk(self.L.pbrhip_scene_info(self.h, C.byref(nn), C.byref(ns), C.byref(dp), C.byref(nb)))
        return dict(num_nodes=nn.value, num_slots=ns.value, depth=dp.value, device_bytes=nb.value)

    # Raytracer::FirstHitTrace1 / AnyHit1 over ray arrays
    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, RAY_DT)
        hits = np.zeros(len(rays), HIT_DT)
        _chk(self.L.pbrhip_trace_closest(self.h, C.c_void_p(rays.ctypes.data), C.c_size_t(len(rays)),
                                         C.c_void_p(hits.ctypes.data)))
        return hits

    def texture_fetch(self, texture_id, uv):
        """pbrhip_texture_fetch: Texture::FetchFloat3 of a texture of this committed scene at (n, 2) coordinates -> (n, 3).  A test hook."""
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        rgb = np.zeros((len(uv), 3), np.float32)
        _chk(self.L.pbrhip_texture_fetch(self.h, C.c_uint32(texture_id), C.c_void_p(uv.ctypes.data), C.c_size_t(len(uv)), C.c_void_p(rgb.ctypes.data)))
        return rgb

    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, RAY_DT)
        occ = np.zeros(len(rays), np.uint8)
        _chk(self.L.pbrhip_trace_any(self.h, C.c_void_p(rays.ctypes.data), C.c_size_t(len(rays)),
                                     C.c_void_p(occ.ctypes.data)))
        return occ


def Render(scene, width, height, num_sample, cancel_render_flag=None, layer=None, finish_pass=None, *,
           first_pass=0, seed_seq=1234567890, tile_rank=0, tile_world=1, max_paths_in_flight=0, flags=0,
           device_out=None, num_streams=0, tail_paths=0, shard_block=0):
    """pbrlab::Render (src/render.h:14-17).  Resizes and clears `layer`, renders `num_sample` passes, and
    returns (True, stats) -- the reference always returns true (render.cc:240).

    cancel_render_flag: optional ctypes.c_ubyte (the byte of the reference's std::atomic_bool) another thread may set
    while the call runs; it is read at every host round trip of the render loop.
    finish_pass: optional ctypes.c_size_t, stored to while the call runs as groups of passes complete.
    device_out: optional (rgba_ptr, count_ptr) DEVICE pointers (ints); then the layer is not touched and
    nothing is copied to the host (used with torch tensors + RCCL reduce)."""
    L = _lib.lib()
    desc = RenderDesc(width, height, num_sample, first_pass, seed_seq, tile_rank, tile_world, max_paths_in_flight,
                      flags, num_streams, tail_paths, shard_block)
    st = RenderStats()
    fin = finish_pass if finish_pass is not None else C.c_size_t(0)
    cancel = C.byref(cancel_render_flag) if cancel_render_flag is not None else None
    if device_out is not None:
        _chk(L.pbrhip_render_device(scene.h, C.byref(desc), cancel, C.c_void_p(device_out[0]),
                                    C.c_void_p(device_out[1]), C.byref(fin), C.byref(st)))
        return True, st.as_dict()
    if layer is None:
        raise ValueError("layer is required")
    if not (flags & RENDER_NO_CLEAR):
        layer.Resize(width, height)  # PrepareRendering (render.cc:99-100)
    _chk(L.pbrhip_render(scene.h, C.byref(desc), cancel, _ptr(layer.rgba), _ptr(layer.count, u32p), C.byref(fin),
                         C.byref(st)))
    return True, st.as_dict()


def _desc(width, height, num_sample, first_pass=0, seed_seq=1234567890, tile_rank=0, tile_world=1,
          max_paths_in_flight=0, flags=0, num_streams=0, tail_paths=0, shard_block=0):
    return RenderDesc(width, height, num_sample, first_pass, seed_seq, tile_rank, tile_world, max_paths_in_flight,
                      flags, num_streams, tail_paths, shard_block)


def replicate(scene, device):
    """pbrhip_scene_replicate: a committed scene's copy on `device` (device-to-device, no second BVH build)."""
    out = Scene.__new__(Scene)
    out.L = _lib.lib()
    h = C.c_void_p()
    _chk(out.L.pbrhip_scene_replicate(scene.h, int(device), C.byref(h)))
    out.h = h
    return out


def RenderMulti(scenes, width, height, num_sample, cancel_render_flag=None, layer=None, finish_pass=None, **kw):
    """pbrhip_render_multi: one frame over several devices of this process (scenes[i] = the scene on device i);
    returns (True, [stats per device])."""
    L = _lib.lib()
    desc = _desc(width, height, num_sample, **kw)
    n = len(scenes)
    hs = (C.c_void_p * n)(*[s.h for s in scenes])
    st = (RenderStats * n)()
    fin = finish_pass if finish_pass is not None else C.c_size_t(0)
    cancel = C.byref(cancel_render_flag) if cancel_render_flag is not None else None
    if not (desc.flags & RENDER_NO_CLEAR):
        layer.Resize(width, height)
    _chk(L.pbrhip_render_multi(hs, n, C.byref(desc), cancel, _ptr(layer.rgba), _ptr(layer.count, u32p), C.byref(fin), st))
    return True, [x.as_dict() for x in st]


class Comm:
    """pbrhip_comm: the library's RCCL communicator for one-process-per-GPU jobs.  `unique_id()` on one rank, hand the
    bytes to every rank (e.g. torch.distributed.broadcast_object_list), then Comm(id, rank, world) on every rank."""

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        _chk(_lib.lib().pbrhip_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, uid, rank, world):
        self.L = _lib.lib()
        self.rank, self.world = int(rank), int(world)
        h = C.c_void_p()
        buf = (C.c_ubyte * 128)(*uid)
        _chk(self.L.pbrhip_comm_create(C.byref(h), buf, self.rank, self.world))
        self.h = h

    def reduce_layer(self, rgba_ptr, count_ptr, num_pixels, root=0):
        _chk(self.L.pbrhip_comm_reduce_layer(self.h, C.c_void_p(rgba_ptr), C.c_void_p(count_ptr), C.c_size_t(num_pixels), int(root)))

    def gather_layer(self, scene, width, height, rgba_ptr, count_ptr, shard_block=0, root=0):
        desc = _desc(width, height, 0, tile_rank=self.rank, tile_world=self.world, shard_block=shard_block)
        _chk(self.L.pbrhip_comm_gather_layer(self.h, scene.h, C.byref(desc), C.c_void_p(rgba_ptr), C.c_void_p(count_ptr), int(root)))

    def close(self):
        if getattr(self, "h", None):
            self.L.pbrhip_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def scene_from_desc(desc, bvh_builder=BVH_HOST_SAH):
    from . import scenes
    s = Scene()
    if bvh_builder != BVH_HOST_SAH:
        s.SetBvhBuilder(bvh_builder)
    scenes.build_scene(s, desc, make_principled, make_hair)
    return s
