"""The header-only C++ shim (include/pbrlab_hip.hpp) with pbrlab's names builds with plain g++ against
libpbrhip.so; on a GPU box it renders, on a CPU box Scene() throws (no fallback)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "shim_smoke")


def build():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "pbrlab_amd", "libpbrhip.so")):
        g.build()
    lib = os.path.join(ROOT, "pbrlab_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_smoke.cc"), "-L" + lib, "-lpbrhip",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", EXE])


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
