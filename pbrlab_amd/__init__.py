"""pbrlab_amd -- MI355X-native path-tracing core behind pbrlab's Scene / Render() / RenderLayer API.

The product is libpbrhip.so (pbrlab_amd/csrc: host C++ + hand-written gfx950 HIP kernels, C ABI in
include/pbrhip.h).  This package is the Python binding used by the tests and bench.py."""
from .api import (HairParam, PbrHipError, PrincipledParam, Render, RenderLayer, Scene, create_tiles,  # noqa: F401
                  device_count, make_hair, make_principled, scene_from_desc, set_device)
