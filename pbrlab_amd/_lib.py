"""Loader for libpbrhip.so (the HIP/C++ core).  There is no fallback: if the library is missing or
cannot be loaded this module raises, and every compute entry point of the library itself fails
with PBRHIP_ENODEVICE when no MI355X is visible."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBRHIP_LIB") or os.path.join(HERE, "libpbrhip.so")  # PBRHIP_LIB: A/B builds
CSRC = os.path.join(HERE, "csrc")

ABI_VERSION = 6  # PBRHIP_ABI_VERSION of include/pbrhip.h (struct layouts of this binding)

# every symbol include/pbrhip.h declares
EXPORTS = [
    "pbrhip_abi_version", "pbrhip_math_mode", "pbrhip_sizeof_render_stats", "pbrhip_last_error", "pbrhip_device_count", "pbrhip_set_device", "pbrhip_scene_create", "pbrhip_scene_destroy",
    "pbrhip_scene_add_triangle_mesh", "pbrhip_scene_add_curve_mesh", "pbrhip_scene_add_principled_material",
    "pbrhip_scene_add_hair_material", "pbrhip_scene_add_texture", "pbrhip_scene_add_area_light", "pbrhip_scene_create_local_scene",
    "pbrhip_scene_add_mesh_to_local_scene", "pbrhip_scene_create_instance", "pbrhip_scene_attach_light_ids",
    "pbrhip_scene_attach_material_ids", "pbrhip_scene_commit", "pbrhip_scene_set_bvh_builder", "pbrhip_scene_aabb",
    "pbrhip_scene_update_principled_material", "pbrhip_scene_update_hair_material", "pbrhip_scene_info",
    "pbrhip_render", "pbrhip_render_device", "pbrhip_trace_closest", "pbrhip_trace_any", "pbrhip_leaf_eval", "pbrhip_texture_fetch", "pbrhip_create_tiles",
    "pbrhip_render_multi", "pbrhip_scene_replicate", "pbrhip_comm_unique_id", "pbrhip_comm_create", "pbrhip_comm_destroy",
    "pbrhip_comm_reduce_layer", "pbrhip_comm_gather_layer",
]


def build(force=False):
    """Compile libpbrhip.so for gfx950 with hipcc (works without a GPU)."""
    args = ["make", "-s", "-C", CSRC, "-j4"]
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (hipcc, gfx950). pbrlab_amd has no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.pbrhip_last_error.restype = C.c_char_p
        _lib.pbrhip_abi_version.restype = C.c_uint32
        _lib.pbrhip_math_mode.restype = C.c_uint32
        _lib.pbrhip_sizeof_render_stats.restype = C.c_size_t
        if _lib.pbrhip_abi_version() != ABI_VERSION:
            v = _lib.pbrhip_abi_version()
            _lib = None
            raise RuntimeError(f"{LIB_PATH} has struct layout version {v}, this binding was written for {ABI_VERSION}: rebuild")
    return _lib
