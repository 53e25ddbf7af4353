"""CPU: the C-ABI library loads and exports every symbol include/pbrhip.h declares; no compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g
    from pbrlab_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib.lib()


def test_header_symbols_exported(L):
    from pbrlab_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "pbrhip.h")).read()
    declared = sorted(set(re.findall(r"\b(pbrhip_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layouts_match_header():
    from pbrlab_amd import _lib, api
    assert C.sizeof(api.PrincipledParam) == 25 * 4 and C.sizeof(api.HairParam) == 20 * 4
    assert C.sizeof(api.RenderDesc) == 56 and api.RAY_DT.itemsize == 32 and api.HIT_DT.itemsize == 36
    assert C.sizeof(api.RenderStats) == 11 * 8 + 9 * 8 + 6 * 8 + 8 + 7 * 8 + 8
    # the library's own idea of the layouts (a caller built against another header must be able to tell)
    L = _lib.lib()
    assert L.pbrhip_abi_version() == _lib.ABI_VERSION
    assert L.pbrhip_sizeof_render_stats() == C.sizeof(api.RenderStats)
    # which restatement of cos / sin / exp / log the kernels were compiled with (include/pbrhip.h PBRHIP_MATH_*): the default build
    # uses glibc 2.35's x86-64 FMA float functions (include/pbr_glibcf.h), the arithmetic the parity tests' oracle mode must match
    import pbrlab_amd as pa
    import _oracle as O
    assert L.pbrhip_math_mode() == 2 == O.MATH_DEVICE and pa.math_mode() == "glibcf"


def test_tiles_host_logic(L):
    import pbrlab_amd as pa
    t = pa.create_tiles(1920, 1080)
    assert t.shape == (510, 4) and tuple(t[0]) == (0, 64, 0, 64) and tuple(t[-1]) == (1856, 1920, 1024, 1080)
    assert pa.create_tiles(3840, 2160).shape[0] == 2040 and pa.create_tiles(256, 256).shape[0] == 16
    assert pa.create_tiles(1, 1).tolist() == [[0, 1, 0, 1]]
    area = ((t[:, 1] - t[:, 0]) * (t[:, 3] - t[:, 2])).sum()
    assert area == 1920 * 1080


def test_fails_loudly_without_device(L):
    import pbrlab_amd as pa
    if pa.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pa.PbrHipError) as e:
        pa.Scene()
    assert e.value.code == -3
    with pytest.raises(pa.PbrHipError):
        pa.set_device(0)


def test_product_does_not_import_oracle():
    """the product path must not route through oracle/: no reference to it anywhere under pbrlab_amd/ or include/"""
    for base in ("pbrlab_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".cpp", ".hip", ".hpp")) or f == "Makefile":
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "libpbr_oracle" not in txt and "_oracle" not in txt and "oracle/" not in txt, os.path.join(dp, f)
