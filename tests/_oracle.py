"""ctypes bindings for the ORACLE (oracle/libpbr_oracle.so) and, when built, the reference leaf
library (oracle/_ref/libref_leaf.so).  Test infrastructure only: nothing under pbrlab_amd/ imports
this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("PBR_ORACLE_SO") or os.path.join(ORACLE_DIR, "libpbr_oracle.so")  # PBR_ORACLE_SO: another build (bench.py)
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_leaf.so")

MATH_LIBM, MATH_F64R, MATH_GLIBCF = 0, 1, 2
# the arithmetic the HIP kernels compute with (pbrhip_math_mode(); tests/test_cabi_cpu.py checks the library says the same)
MATH_DEVICE = MATH_GLIBCF
def host_threads():
    """CPUs this process may really use: min(os.cpu_count(), scheduler affinity, the cgroup's CPU quota).  The GPU boxes report 256
    hardware threads and run the job under cpu.max = 16 CPUs: 256 oracle threads then take turns on 16 cores and finish LATER than
    32 do (scripts/cpu_scaling.py: 5.6 Msamples/s at 32 threads, 3.4 at 256)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pe = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(round(q / pe))))
        except (OSError, ValueError):
            pass
    return n


_libm_is_glibcf = None


def libm_is_glibcf():
    """True when the host libm's cosf / sinf / expf / logf are the functions include/pbr_glibcf.h restates (glibc 2.35 as built by Ubuntu 22.04, x86-64 with
    FMA: this container and the GPU boxes): 4 x 16.7 M sampled arguments, cached.  Then oracle[libm] == oracle[glibcf] == the GPU, bit for bit."""
    global _libm_is_glibcf
    if _libm_is_glibcf is None:
        _libm_is_glibcf = int(lib().orc_glibcf_vs_libm(257, 0)) == 0
    return _libm_is_glibcf


def oracle_threads():
    """worker threads for the oracle's pool: twice the usable CPUs where a quota is in force (the pool peaks there), else all of them"""
    n, hw = host_threads(), os.cpu_count() or 1
    return min(hw, 2 * n) if n < hw else n


JOBS_BLOCKS, JOBS_TILE_PASS = 0, 1  # orc_render_jobs: the worker pool's job granularity (pbr_oracle.h)


class PrincipledParam(C.Structure):
    """== orc_principled_param == pbrhip_principled_param (material-param.h:24-49)."""
    _fields_ = [("base_color", C.c_float * 3), ("subsurface", C.c_float),
                ("subsurface_radius", C.c_float * 3), ("subsurface_color", C.c_float * 3),
                ("metallic", C.c_float), ("specular", C.c_float), ("specular_tint", C.c_float),
                ("roughness", C.c_float), ("anisotropic", C.c_float), ("anisotropic_rotation", C.c_float),
                ("sheen", C.c_float), ("sheen_tint", C.c_float), ("clearcoat", C.c_float),
                ("clearcoat_roughness", C.c_float), ("ior", C.c_float), ("transmission", C.c_float),
                ("transmission_roughness", C.c_float), ("base_color_tex_id", C.c_uint32),
                ("subsurface_color_tex_id", C.c_uint32)]


class HairParam(C.Structure):
    _fields_ = [("coloring_hair", C.c_uint32), ("base_color", C.c_float * 3), ("melanin", C.c_float),
                ("melanin_redness", C.c_float), ("melanin_randomize", C.c_float), ("roughness", C.c_float),
                ("azimuthal_roughness", C.c_float), ("ior", C.c_float), ("shift", C.c_float),
                ("specular_tint", C.c_float * 3), ("second_specular_tint", C.c_float * 3),
                ("transmission_tint", C.c_float * 3)]


RAY_DT = np.dtype([("org", "<f4", 3), ("tmin", "<f4"), ("dir", "<f4", 3), ("tmax", "<f4")])
HIT_DT = np.dtype([("normal_g", "<f4", 3), ("t", "<f4"), ("u", "<f4"), ("v", "<f4"),
                   ("instance_id", "<u4"), ("geom_id", "<u4"), ("prim_id", "<u4")])


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "closest_rays", "shadow_rays", "nodes_visited",
                                          "tris_tested", "curves_tested", "bounces", "sss_steps", "rng_draws")]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


_lib = None
_ref = None
fp = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)


def _ptr(a, t=fp):
    return None if a is None else a.ctypes.data_as(t)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_add_triangle_mesh.argtypes = [C.c_void_p, fp, C.c_uint32, fp, C.c_uint32, fp, C.c_uint32, u32p, u32p,
                                            u32p, u32p, C.c_uint32]
        L.orc_add_curve_mesh.argtypes = [C.c_void_p, fp, C.c_uint32, u32p, u32p, C.c_uint32]
        L.orc_add_principled.argtypes = [C.c_void_p, C.POINTER(PrincipledParam)]
        L.orc_add_hair.argtypes = [C.c_void_p, C.POINTER(HairParam)]
        L.orc_add_area_light.argtypes = [C.c_void_p, fp]
        L.orc_add_texture.argtypes = [C.c_void_p, fp, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_kat_texture_fetch.argtypes = [fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, fp]
        L.orc_create_local_scene.argtypes = [C.c_void_p]
        L.orc_add_mesh_to_local_scene.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_create_instance.argtypes = [C.c_void_p, C.c_uint32, fp]
        L.orc_attach_light_ids.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32p, C.c_uint32]
        L.orc_attach_material_ids.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32p, C.c_uint32]
        L.orc_commit.argtypes = [C.c_void_p]
        L.orc_scene_aabb.argtypes = [C.c_void_p, fp, fp]
        L.orc_bvh_depth.argtypes = [C.c_void_p]
        L.orc_bvh_depth.restype = C.c_uint32
        L.orc_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.orc_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32,
                                 C.c_uint32, C.c_uint32, fp, u32p, C.POINTER(Stats)]
        L.orc_last_render_busy.restype = C.c_double
        L.orc_render_jobs.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32,
                                      C.c_uint32, C.c_uint32, C.c_uint32, fp, u32p, C.POINTER(Stats)]
        L.orc_sample_trace.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint64, fp, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint32]
        L.orc_sample_trace.restype = C.c_uint32
        L.orc_camera_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_uint64, C.c_void_p]
        L.orc_kat_rng.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, fp]
        L.orc_kat_fastmath.argtypes = [C.c_int, C.c_float, C.c_float]
        L.orc_kat_fastmath.restype = C.c_float
        L.orc_kat_fresnel.argtypes = [C.c_float, C.c_float]
        L.orc_kat_fresnel.restype = C.c_float
        L.orc_kat_power_heuristic.argtypes = [C.c_float, C.c_float]
        L.orc_kat_power_heuristic.restype = C.c_float
        L.orc_kat_lambert_sample.argtypes = [C.c_float, C.c_float, fp]
        L.orc_kat_ggx_eval.argtypes = [fp, fp, C.c_float, C.c_float, C.c_int, fp]
        L.orc_kat_ggx_sample.argtypes = [fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, fp]
        L.orc_kat_hair_eval.argtypes = [fp, fp, fp, fp]
        L.orc_kat_hair_sample.argtypes = [fp, fp, fp, fp]
        L.orc_kat_uniform_sphere.argtypes = [C.c_float, C.c_float, fp]
        L.orc_kat_triangle_sampler.argtypes = [C.c_float, C.c_float, fp]
        L.orc_kat_param_to_bsdf.argtypes = [C.POINTER(PrincipledParam), fp]
        L.orc_kat_hair_param_to_bsdf.argtypes = [C.POINTER(HairParam), C.c_float, fp]
        L.orc_light_table.argtypes = [C.c_void_p, C.c_uint32, u32p, u32p, fp, fp, u32p]
        L.orc_light_table.restype = C.c_uint32
        L.orc_light_prims.argtypes = [C.c_void_p, C.c_uint32, fp, fp, fp]
        L.orc_create_tiles.argtypes = [C.c_uint32, C.c_uint32, u32p, u32p]
        L.orc_to_cubic_bezier.argtypes = [fp, fp, C.c_uint32, fp]
        L.orc_set_math_mode.argtypes = [C.c_int]
        L.orc_glibcf_vs_libm.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_glibcf_vs_libm.restype = C.c_uint64
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.ref_rng.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, fp]
        R.ref_fastmath.argtypes = [C.c_int, C.c_float, C.c_float]
        R.ref_fastmath.restype = C.c_float
        R.ref_fresnel.argtypes = [C.c_float, C.c_float]
        R.ref_fresnel.restype = C.c_float
        R.ref_power_heuristic.argtypes = [C.c_float, C.c_float]
        R.ref_power_heuristic.restype = C.c_float
        R.ref_lambert_sample.argtypes = [C.c_float, C.c_float, fp]
        R.ref_ggx_eval.argtypes = [fp, fp, C.c_float, C.c_float, C.c_int, fp]
        R.ref_ggx_sample.argtypes = [fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, fp]
        R.ref_hair_eval.argtypes = [fp, fp, fp, fp]
        R.ref_hair_sample.argtypes = [fp, fp, fp, fp]
        R.ref_uniform_sphere.argtypes = [C.c_float, C.c_float, fp]
        R.ref_uniform_sphere_from_rng.argtypes = [C.c_uint64, C.c_uint64, fp]
        R.ref_triangle_sampler.argtypes = [C.c_float, C.c_float, fp]
        R.ref_cosine_hemisphere.argtypes = [C.c_float, C.c_float, fp]
        R.ref_mult_v.argtypes = [fp, fp, fp]
        R.ref_create_tiles.argtypes = [C.c_uint32, C.c_uint32, u32p, u32p]
        R.ref_to_cubic_bezier.argtypes = [fp, fp, C.c_uint32, fp]
        R.ref_triangle_fetch.argtypes = [fp, C.c_uint32, fp, C.c_uint32, u32p, u32p, C.c_uint32, C.c_uint32,
                                         C.c_float, C.c_float, C.c_int, fp]
        R.ref_texture_fetch.argtypes = [fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, fp]
        R.ref_linear_to_srgb.argtypes = [C.c_float]
        R.ref_linear_to_srgb.restype = C.c_float
        R.ref_srgb_to_linear.argtypes = [C.c_float]
        R.ref_srgb_to_linear.restype = C.c_float
        R.ref_spectrum_norm.argtypes = [fp]
        R.ref_spectrum_norm.restype = C.c_float
        R.ref_rgb_to_y.argtypes = [fp]
        R.ref_rgb_to_y.restype = C.c_float
        _ref = R
    return _ref


def f32(*v):
    return np.ascontiguousarray(np.array(v, dtype=np.float32).reshape(-1))


class OracleScene:
    """Mirrors pbrlab::Scene's builder methods (scene.h:19-91) over the oracle C API, with the same
    call sequence as pbrlab_amd.Scene so scene descriptions can be replayed on both."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_scene_create())

    def __del__(self):
        try:
            if self.h:
                self.L.orc_scene_destroy(self.h)
                self.h = None
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
        return self.L.orc_add_triangle_mesh(self.h, _ptr(v), len(v), _ptr(n), len(n), _ptr(t), len(t), _ptr(vid, u32p),
                                            _ptr(nid, u32p), _ptr(tid, u32p), _ptr(mid, u32p), len(vid))

    def AddCubicBezierCurveMesh(self, vertices_xyzr, indices, material_ids=None):
        v = np.ascontiguousarray(vertices_xyzr, np.float32).reshape(-1, 4)
        idx = np.ascontiguousarray(indices, np.uint32).reshape(-1)
        mid = None if material_ids is None else np.ascontiguousarray(material_ids, np.uint32).reshape(-1)
        return self.L.orc_add_curve_mesh(self.h, _ptr(v), len(v), _ptr(idx, u32p), _ptr(mid, u32p), len(idx))

    def AddMaterialParam(self, p):
        if isinstance(p, PrincipledParam):
            return self.L.orc_add_principled(self.h, C.byref(p))
        return self.L.orc_add_hair(self.h, C.byref(p))

    def AddTexture(self, pixels):
        px = np.ascontiguousarray(pixels, np.float32)
        if px.ndim == 2:
            px = px[..., None]
        return self.L.orc_add_texture(self.h, _ptr(px), px.shape[1], px.shape[0], px.shape[2])

    def AddLightParam(self, emission):
        e = f32(*emission)
        return self.L.orc_add_area_light(self.h, _ptr(e))

    def CreateLocalScene(self):
        return self.L.orc_create_local_scene(self.h)

    def AddMeshToLocalScene(self, local_id, mesh_id):
        return self.L.orc_add_mesh_to_local_scene(self.h, local_id, mesh_id)

    def CreateInstance(self, local_id, transform=None):
        t = None if transform is None else np.ascontiguousarray(transform, np.float32).reshape(16)
        return self.L.orc_create_instance(self.h, local_id, _ptr(t))

    def AttachLightParamIdsToInstance(self, instance_id, ids_per_geom):
        for g, ids in enumerate(ids_per_geom):
            a = np.ascontiguousarray(ids, np.uint32).reshape(-1)
            if self.L.orc_attach_light_ids(self.h, instance_id, g, _ptr(a, u32p), len(a)) != 0:
                raise RuntimeError("light param error")

    def AttachMaterialParamIdsToInstance(self, instance_id, ids_per_geom):
        for g, ids in enumerate(ids_per_geom):
            a = np.ascontiguousarray(ids, np.uint32).reshape(-1)
            if self.L.orc_attach_material_ids(self.h, instance_id, g, _ptr(a, u32p), len(a)) != 0:
                raise RuntimeError("material param error")

    def CommitScene(self):
        self.L.orc_commit(self.h)

    def FetchSceneAABB(self):
        lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self.L.orc_scene_aabb(self.h, _ptr(lo), _ptr(hi))
        return lo, hi

    # ---- tracing / rendering
    def trace_closest(self, rays, brute_force=False):
        rays = np.ascontiguousarray(rays, RAY_DT)
        hits = np.zeros(len(rays), HIT_DT)
        self.L.orc_trace_closest(self.h, rays.ctypes.data, len(rays), hits.ctypes.data, int(brute_force))
        return hits

    def trace_any(self, rays, brute_force=False):
        rays = np.ascontiguousarray(rays, RAY_DT)
        occ = np.zeros(len(rays), np.uint8)
        self.L.orc_trace_any(self.h, rays.ctypes.data, len(rays), occ.ctypes.data, int(brute_force))
        return occ

    def render(self, width, height, spp, first_pass=0, seed_seq=1234567890, tile_rank=0, tile_world=1, threads=1,
               math_mode=MATH_LIBM, job_mode=JOBS_BLOCKS):
        """job_mode: JOBS_BLOCKS (schedule-independent image: the checker) or JOBS_TILE_PASS (the reference's pool: timing)"""
        rgba = np.zeros((height, width, 4), np.float32)
        count = np.zeros((height, width), np.uint32)
        st = Stats()
        self.L.orc_set_math_mode(math_mode)
        try:
            self.L.orc_render_jobs(self.h, width, height, spp, first_pass, seed_seq, tile_rank, tile_world, threads, job_mode,
                                   _ptr(rgba), _ptr(count, u32p), C.byref(st))
        finally:
            self.L.orc_set_math_mode(MATH_LIBM)
        return rgba, count, {n: getattr(st, n) for n, _ in Stats._fields_}

    def sample_trace(self, width, height, x, y, p, seed_seq=1234567890, max_hits=64, math_mode=MATH_LIBM):
        rad = np.zeros(3, np.float32)
        draws = C.c_uint64(0)
        hits = np.zeros(max_hits, HIT_DT)
        self.L.orc_set_math_mode(math_mode)
        try:
            n = self.L.orc_sample_trace(self.h, width, height, x, y, p, seed_seq, _ptr(rad), C.byref(draws),
                                        hits.ctypes.data, max_hits)
        finally:
            self.L.orc_set_math_mode(MATH_LIBM)
        return rad, draws.value, n, hits[:min(n, max_hits)]

    def camera_rays(self, width, height, pixels, p=0, seed_seq=1234567890):
        out = np.zeros(len(pixels), RAY_DT)
        for i, (x, y) in enumerate(pixels):
            self.L.orc_camera_ray(self.h, width, height, int(x), int(y), p, seed_seq, out[i:i + 1].ctypes.data)
        return out


def make_principled(d):
    p = PrincipledParam()
    for k, _ in PrincipledParam._fields_:
        v = d[k]
        if isinstance(v, (tuple, list, np.ndarray)):
            setattr(p, k, (C.c_float * 3)(*[float(x) for x in v]))
        else:
            setattr(p, k, v)
    return p


def make_hair(d):
    p = HairParam()
    for k, _ in HairParam._fields_:
        v = d[k]
        if isinstance(v, (tuple, list, np.ndarray)):
            setattr(p, k, (C.c_float * 3)(*[float(x) for x in v]))
        else:
            setattr(p, k, v)
    return p


def oracle_scene_from_desc(desc):
    from pbrlab_amd import scenes
    s = OracleScene()
    scenes.build_scene(s, desc, make_principled, make_hair)
    return s


def to_srgb8(rgba, count):
    """pbrlab-cli output stage (pc/pbrlab-cli.cc:49-54, image-utils.cc:26-38, image-io.cc:206)."""
    img = rgba / np.maximum(count, 1)[..., None].astype(np.float32)
    rgb = img[..., :3]
    s = np.where(rgb <= 0.0031308, 12.92 * rgb, 1.055 * np.power(np.maximum(rgb, 0), 1 / 2.4) - 0.055)
    return np.clip(s * 256, 0, 255).astype(np.uint8)
