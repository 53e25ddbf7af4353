"""ctypes binding of libpbrhip_io (include/pbrhip_io.h): pbrlab's scene ingestion and image output.

    LoadTriangleMeshFromObj   src/io/triangle-mesh-io.cc:214-325
    LoadCurveMeshAsCubicBezierCurve   src/io/curve-mesh-io.cc:32-138
    CreateScene               pc/pc-common.cc:238-270
    LoadImageFromFile / WritePNG   src/io/image-io.cc:98-224
    write_layer_png           pc/pbrlab-cli.cc:47-57

No fallback: a missing library raises."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _lib
from .api import PrincipledParam, Scene

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpbrhip_io.so")
CLI_PATH = os.path.join(HERE, "pbrlab-hip-cli")
CSRC = os.path.join(HERE, "csrc", "io")

# every symbol include/pbrhip_io.h declares
EXPORTS = [
    "pbrio_last_error", "pbrio_free", "pbrio_obj_load", "pbrio_obj_free", "pbrio_obj_attribute", "pbrio_obj_num_shapes",
    "pbrio_obj_shape_name", "pbrio_obj_shape_ids", "pbrio_obj_num_materials", "pbrio_obj_material",
    "pbrio_obj_num_textures", "pbrio_obj_texture", "pbrio_obj_text", "pbrio_obj_warnings", "pbrio_parse_texture_statement", "pbrio_curves_load",
    "pbrio_curves_free", "pbrio_curves_vertices", "pbrio_curves_indices", "pbrio_scene_add_obj", "pbrio_scene_add_hair",
    "pbrio_create_scene", "pbrio_image_load", "pbrio_write_png_f32", "pbrio_write_png_u8", "pbrio_png_decode",
    "pbrio_layer_to_srgb8", "pbrio_write_layer_png",
]

fpp = C.POINTER(C.POINTER(C.c_float))
u32pp = C.POINTER(C.POINTER(C.c_uint32))


def build():
    _lib.build()
    subprocess.check_call(["make", "-s", "-C", CSRC, "-j4"])
    return LIB_PATH


_io = None


def lib():
    global _io
    if _io is None:
        _lib.lib()  # libpbrhip_io links against libpbrhip
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        L.pbrio_last_error.restype = C.c_char_p
        L.pbrio_free.argtypes = [C.c_void_p]
        L.pbrio_obj_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.pbrio_obj_free.argtypes = [C.c_void_p]
        L.pbrio_obj_attribute.restype = C.c_size_t
        L.pbrio_obj_attribute.argtypes = [C.c_void_p, C.c_int, fpp]
        L.pbrio_obj_num_shapes.restype = C.c_uint32
        L.pbrio_obj_num_shapes.argtypes = [C.c_void_p]
        L.pbrio_obj_shape_name.restype = C.c_char_p
        L.pbrio_obj_shape_name.argtypes = [C.c_void_p, C.c_uint32]
        L.pbrio_obj_shape_ids.restype = C.c_size_t
        L.pbrio_obj_shape_ids.argtypes = [C.c_void_p, C.c_uint32, C.c_int, u32pp]
        L.pbrio_obj_num_materials.restype = C.c_uint32
        L.pbrio_obj_num_materials.argtypes = [C.c_void_p]
        L.pbrio_obj_material.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(PrincipledParam), C.POINTER(C.c_char_p)]
        L.pbrio_obj_num_textures.restype = C.c_uint32
        L.pbrio_obj_num_textures.argtypes = [C.c_void_p]
        L.pbrio_obj_texture.argtypes = [C.c_void_p, C.c_uint32, fpp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_char_p)]
        L.pbrio_obj_text.restype = C.c_char_p
        L.pbrio_obj_text.argtypes = [C.c_void_p]
        L.pbrio_obj_warnings.restype = C.c_char_p
        L.pbrio_obj_warnings.argtypes = [C.c_void_p]
        L.pbrio_curves_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.pbrio_curves_free.argtypes = [C.c_void_p]
        L.pbrio_curves_vertices.restype = C.c_size_t
        L.pbrio_curves_vertices.argtypes = [C.c_void_p, fpp]
        L.pbrio_curves_indices.restype = C.c_size_t
        L.pbrio_curves_indices.argtypes = [C.c_void_p, u32pp]
        L.pbrio_scene_add_obj.argtypes = [C.c_void_p, C.c_char_p]
        L.pbrio_scene_add_hair.argtypes = [C.c_void_p, C.c_char_p]
        L.pbrio_create_scene.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_void_p]
        L.pbrio_image_load.argtypes = [C.c_char_p, C.c_char_p, fpp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_size_t)]
        L.pbrio_write_png_f32.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
        L.pbrio_write_png_u8.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
        L.pbrio_png_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.pbrio_layer_to_srgb8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.pbrio_write_layer_png.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        _io = L
    return _io


class PbrIoError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        msg = lib().pbrio_last_error()
        raise PbrIoError(msg.decode(errors="replace") if msg else f"pbrio error {rc}")


def _copy(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class ObjScene:
    """Result of io::LoadTriangleMeshFromObj: shared attributes, one mesh per shape, materials, textures."""

    def __init__(self, filename):
        L = lib()
        h = C.c_void_p()
        _chk(L.pbrio_obj_load(os.fsencode(filename), C.byref(h)))
        try:
            p = C.POINTER(C.c_float)()
            self.vertices = _copy(p, L.pbrio_obj_attribute(h, 0, C.byref(p)), np.float32).reshape(-1, 4)
            self.normals = _copy(p, L.pbrio_obj_attribute(h, 1, C.byref(p)), np.float32).reshape(-1, 4)
            self.texcoords = _copy(p, L.pbrio_obj_attribute(h, 2, C.byref(p)), np.float32).reshape(-1, 2)
            self.meshes = []
            q = C.POINTER(C.c_uint32)()
            for s in range(L.pbrio_obj_num_shapes(h)):
                m = {"name": L.pbrio_obj_shape_name(h, s).decode(errors="replace")}
                for which, key in enumerate(("vertex_ids", "normal_ids", "texcoord_ids", "material_ids")):
                    m[key] = _copy(q, L.pbrio_obj_shape_ids(h, s, which, C.byref(q)), np.uint32)
                self.meshes.append(m)
            self.materials, self.material_names = [], []
            for i in range(L.pbrio_obj_num_materials(h)):
                pp, name = PrincipledParam(), C.c_char_p()
                _chk(L.pbrio_obj_material(h, i, C.byref(pp), C.byref(name)))
                self.materials.append(pp)
                self.material_names.append(name.value.decode(errors="replace"))
            self.textures = []
            for i in range(L.pbrio_obj_num_textures(h)):
                w, hh, c, name = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_char_p()
                _chk(L.pbrio_obj_texture(h, i, C.byref(p), C.byref(w), C.byref(hh), C.byref(c), C.byref(name)))
                px = _copy(p, w.value * hh.value * c.value, np.float32).reshape(hh.value, w.value, c.value)
                self.textures.append({"pixels": px, "name": name.value.decode(errors="replace")})
            self.text = L.pbrio_obj_text(h).decode(errors="replace")
            self.warnings = L.pbrio_obj_warnings(h).decode(errors="replace")
        finally:
            L.pbrio_obj_free(h)


def parse_texture_statement(value):
    name, cs = C.create_string_buffer(4096), C.create_string_buffer(256)
    L = lib()
    L.pbrio_parse_texture_statement.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    found = L.pbrio_parse_texture_statement(value.encode(), name, 4096, cs, 256)
    return bool(found), name.value.decode(), cs.value.decode()


def LoadTriangleMeshFromObj(filename):
    return ObjScene(filename)


def LoadCurveMeshAsCubicBezierCurve(filepath, memory_saving_mode=False):
    """-> (ok, vertices_thickness[n,4], indices[m]); like the reference, a failed conversion keeps the strands done so far"""
    L = lib()
    h = C.c_void_p()
    rc = L.pbrio_curves_load(os.fsencode(filepath), int(bool(memory_saving_mode)), C.byref(h))
    if not h:
        _chk(rc)
    try:
        p, q = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
        vt = _copy(p, L.pbrio_curves_vertices(h, C.byref(p)), np.float32).reshape(-1, 4)
        idx = _copy(q, L.pbrio_curves_indices(h, C.byref(q)), np.uint32)
    finally:
        L.pbrio_curves_free(h)
    return rc == 0, vt, idx


def CreateScene(files, scene=None):
    """CreateScene(argc, argv, &scene): adds every .obj / .hair of `files` and commits (needs the GPU)."""
    scene = scene or Scene()
    argv = [b"pbrlab"] + [os.fsencode(f) for f in files]
    arr = (C.c_char_p * len(argv))(*argv)
    _chk(lib().pbrio_create_scene(len(argv), arr, scene.h))
    return scene


def add_obj(scene, filename):
    _chk(lib().pbrio_scene_add_obj(scene.h, os.fsencode(filename)))


def add_hair(scene, filename):
    _chk(lib().pbrio_scene_add_hair(scene.h, os.fsencode(filename)))


def LoadImageFromFile(filename, asset_path=""):
    p = C.POINTER(C.c_float)()
    w, h, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    _chk(lib().pbrio_image_load(os.fsencode(filename), os.fsencode(asset_path), C.byref(p), C.byref(w), C.byref(h), C.byref(c)))
    try:
        return _copy(p, w.value * h.value * c.value, np.float32).reshape(h.value, w.value, c.value)
    finally:
        lib().pbrio_free(p)


def WritePNG(filename, asset_path, pixels):
    px = np.ascontiguousarray(pixels)
    h, w = px.shape[:2]
    c = 1 if px.ndim == 2 else px.shape[2]
    if px.dtype == np.uint8:
        _chk(lib().pbrio_write_png_u8(os.fsencode(filename), os.fsencode(asset_path), px.ctypes.data, w, h, c))
    else:
        px = np.ascontiguousarray(px, np.float32)
        _chk(lib().pbrio_write_png_f32(os.fsencode(filename), os.fsencode(asset_path), px.ctypes.data, w, h, c))


def png_decode(data):
    p = C.POINTER(C.c_uint8)()
    w, h, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    _chk(lib().pbrio_png_decode(data, len(data), C.byref(p), C.byref(w), C.byref(h), C.byref(c)))
    try:
        return _copy(p, w.value * h.value * c.value, np.uint8).reshape(h.value, w.value, c.value)
    finally:
        lib().pbrio_free(p)


def layer_to_srgb8(rgba, count):
    rgba = np.ascontiguousarray(rgba, np.float32)
    count = np.ascontiguousarray(count, np.uint32)
    h, w = count.shape
    out = np.zeros((h, w, 4), np.uint8)
    _chk(lib().pbrio_layer_to_srgb8(rgba.ctypes.data, count.ctypes.data, w, h, out.ctypes.data))
    return out


def write_layer_png(filename, asset_path, rgba, count):
    rgba = np.ascontiguousarray(rgba, np.float32)
    count = np.ascontiguousarray(count, np.uint32)
    h, w = count.shape
    _chk(lib().pbrio_write_layer_png(os.fsencode(filename), os.fsencode(asset_path), rgba.ctypes.data, count.ctypes.data, w, h))
