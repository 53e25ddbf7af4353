"""ctypes access to oracle/_ref/libref_io.so: the reference's own host I/O code (vendored tinyobjloader, cyhair.cc,
curve-mesh-io.cc, image-io.cc + stb) compiled unmodified by oracle/Makefile.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libref_io.so")
_L = None


def available():
    return os.path.exists(PATH)


def lib():
    global _L
    if _L is None:
        L = C.CDLL(PATH)
        for n in ("refio_obj_load", "refio_hair_load", "refio_image_load"):
            getattr(L, n).restype = C.c_void_p
        L.refio_obj_load.argtypes = [C.c_char_p, C.c_char_p]
        L.refio_obj_size.restype = C.c_size_t
        L.refio_obj_size.argtypes = [C.c_void_p, C.c_int]
        L.refio_obj_data.restype = C.c_void_p
        L.refio_obj_data.argtypes = [C.c_void_p, C.c_int]
        L.refio_obj_ok.argtypes = [C.c_void_p]
        L.refio_obj_free.argtypes = [C.c_void_p]
        L.refio_hair_load.argtypes = [C.c_char_p, C.c_int]
        L.refio_hair_ok.argtypes = [C.c_void_p]
        L.refio_hair_size.restype = C.c_size_t
        L.refio_hair_size.argtypes = [C.c_void_p, C.c_int]
        L.refio_hair_data.restype = C.c_void_p
        L.refio_hair_data.argtypes = [C.c_void_p, C.c_int]
        L.refio_hair_free.argtypes = [C.c_void_p]
        L.refio_image_load.argtypes = [C.c_char_p, C.c_char_p]
        L.refio_image_ok.argtypes = [C.c_void_p]
        L.refio_image_dims.argtypes = [C.c_void_p] + [C.POINTER(C.c_size_t)] * 3
        L.refio_image_data.restype = C.c_void_p
        L.refio_image_data.argtypes = [C.c_void_p]
        L.refio_image_free.argtypes = [C.c_void_p]
        L.refio_cli_output.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        L.refio_write_png_u8.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
        L.refio_save_exr.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.refio_write_jpg.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.refio_parse_texopt.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        _L = L
    return _L


def _arr(h, which, dtype, fn_size, fn_data):
    n = fn_size(h, which)
    if n == 0:
        return np.zeros(0, dtype)
    p = fn_data(h, which)
    return np.frombuffer((C.c_char * (n * np.dtype(dtype).itemsize)).from_address(p), dtype=dtype).copy()


def obj_load(filename, base_dir):
    """tinyobj::LoadObj as triangle-mesh-io.cc calls it -> dict"""
    L = lib()
    h = L.refio_obj_load(os.fsencode(filename), os.fsencode(base_dir))
    try:
        out = {"ok": bool(L.refio_obj_ok(h))}
        for which, (key, dt) in enumerate([("vertices", np.float32), ("normals", np.float32), ("texcoords", np.float32),
                                           ("corners", np.int32), ("shape_first", np.int32), ("material_ids", np.int32)]):
            out[key] = _arr(h, which, dt, L.refio_obj_size, L.refio_obj_data)
        out["text"] = _arr(h, 6, np.uint8, L.refio_obj_size, L.refio_obj_data).tobytes().decode(errors="replace")
        return out
    finally:
        L.refio_obj_free(h)


def hair_load(path, memory_saving_mode=False):
    L = lib()
    h = L.refio_hair_load(os.fsencode(path), int(memory_saving_mode))
    try:
        return (bool(L.refio_hair_ok(h)), _arr(h, 0, np.float32, L.refio_hair_size, L.refio_hair_data).reshape(-1, 4),
                _arr(h, 1, np.uint32, L.refio_hair_size, L.refio_hair_data))
    finally:
        L.refio_hair_free(h)


def image_load(filename, asset_path=""):
    L = lib()
    h = L.refio_image_load(os.fsencode(filename), os.fsencode(asset_path))
    try:
        if not L.refio_image_ok(h):
            return None
        w, hh, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        L.refio_image_dims(h, C.byref(w), C.byref(hh), C.byref(c))
        n = w.value * hh.value * c.value
        p = L.refio_image_data(h)
        return np.frombuffer((C.c_char * (n * 4)).from_address(p), dtype=np.float32).copy().reshape(hh.value, w.value, c.value)
    finally:
        L.refio_image_free(h)


def cli_output(filename, directory, rgba, count):
    rgba = np.ascontiguousarray(rgba, np.float32)
    count = np.ascontiguousarray(count, np.uint32)
    h, w = count.shape
    return bool(lib().refio_cli_output(os.fsencode(filename), os.fsencode(directory), rgba.ctypes.data, count.ctypes.data, w, h))


def parse_texopt(value):
    name = C.create_string_buffer(1024)
    cs = C.create_string_buffer(256)
    ok = lib().refio_parse_texopt(value.encode(), name, 1024, cs, 256)
    return bool(ok), name.value.decode(), cs.value.decode()


def save_exr(path, planes, names, half=False, compression=3, line_order=0):
    """tinyexr's writer (test-file generator).  planes: (nchan, h, w) float32; names: alphabetical channel names"""
    planes = np.ascontiguousarray(planes, np.float32)
    nchan, h, w = planes.shape
    assert list(names) == sorted(names)
    blob = b"".join(n.encode() + b"\0" for n in names)
    return bool(lib().refio_save_exr(os.fsencode(path), planes.ctypes.data, blob, nchan, w, h, int(half), compression, line_order))


def save_exr_isolated(path, planes, names, half=False, compression=3, line_order=0):
    """save_exr in a child process: tinyexr's PIZ compressor crashes on some inputs (seen: one FLOAT channel of smooth data,
    38 x 34), which must not take the test run down.  Returns False when the child failed."""
    import pickle
    import subprocess
    import sys
    blob = pickle.dumps((path, np.ascontiguousarray(planes, np.float32), list(names), half, compression, line_order))
    code = ("import pickle, sys; sys.path.insert(0, %r); import _refio; a = pickle.loads(sys.stdin.buffer.read()); "
            "sys.exit(0 if _refio.save_exr(*a) else 1)") % os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", code], input=blob, capture_output=True)
    return r.returncode == 0 and os.path.exists(path)


def write_jpg(path, pixels, quality=90):
    """stb_image_write's JPEG writer (test-file generator).  pixels: (h, w, c) uint8"""
    px = np.ascontiguousarray(pixels, np.uint8)
    h, w, c = px.shape
    return bool(lib().refio_write_jpg(os.fsencode(path), px.ctypes.data, w, h, c, quality))
