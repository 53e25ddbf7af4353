"""The header-only C++ shim (include/pbrlab_hip.hpp) with pbrlab's names builds with plain g++ against
libpbrhip.so; on a GPU box it renders, on a CPU box Scene() throws (no fallback)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "shim_smoke")
EXE_TYPES = os.path.join(ROOT, "tests", "cpp", "shim_types")


def build():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "pbrlab_amd", "libpbrhip.so")):
        g.build()
    lib = os.path.join(ROOT, "pbrlab_amd")
    for src, exe in (("shim_smoke.cc", EXE), ("shim_types.cc", EXE_TYPES)):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", src), "-L" + lib, "-lpbrhip",
                               "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe])


def test_shim_compiles_and_fails_loudly_on_cpu():
    import pbrlab_amd as pa
    build()
    r = subprocess.run([EXE], capture_output=True, text=True)
    if pa.device_count() == 0:
        assert r.returncode == 3 and "no ROCm-capable device" in r.stderr or "no HIP device" in r.stderr, r.stderr
    else:
        assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_shim_renders_on_gpu():
    build()
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 0 and "shim ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_shim_value_types_and_material_edits():
    """tests/cpp/shim_types.cc: a caller written from INTEGRATION.md's API tables with pbrlab's own value types (TriangleMesh,
    Attribute, Texture, MaterialParameter, MeshPtr, float3, FetchMeshMaterialParameters) and the GUI-style material edit between
    two Render() calls.  (The reference's own callers are compiled unmodified by tests/test_reference_callers.py.)"""
    build()
    r = subprocess.run([EXE_TYPES], capture_output=True, text=True)
    assert r.returncode == 0 and "shim types ok" in r.stdout, (r.returncode, r.stdout + r.stderr)
