"""pbrlab_amd -- MI355X-native path-tracing core behind pbrlab's Scene / Render() / RenderLayer API.

The product is libpbrhip.so (pbrlab_amd/csrc: host C++ + hand-written gfx950 HIP kernels, C ABI in
include/pbrhip.h).  This package is the Python binding used by the tests and bench.py."""
from .api import (Comm, HairParam, PbrHipError, PrincipledParam, Render, RenderLayer, RenderMulti, Scene,  # noqa: F401
                  create_tiles, device_count, make_hair, math_mode, make_principled, replicate, scene_from_desc, set_device)
