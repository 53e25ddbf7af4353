"""The reference's own callers on this repository's drop-in.

pc/pbrlab-cli.cc (main) and pc/pc-common.cc (CreateScene, CreateSceneFromObj, CreateSceneFromCubicBezierCurve, EditQueue) are
compiled UNMODIFIED, where they lie under /root/reference, against include/pbrlab_hip.hpp + include/pbrlab_hip_io.hpp through
the one-line forwarding headers of tests/cpp/fwd (the reference's header names), and linked against libpbrhip / libpbrhip_io:
oracle/Makefile target `ref_cli` -> oracle/_ref/pbrlab-cli-ref (a built artefact: git-ignored, it travels to the GPU box with
the other built binaries).  Nothing of the reference's src/ is compiled and nothing of the reference is copied into the repo.

  CPU (-m "not gpu"): the two files compile and link (skipped where /root/reference is absent); without a device the binary
                      fails loudly -- Scene() throws, there is no CPU fallback.
  GPU (-m gpu):       the reference's main() renders its hard-wired 512 x 512 x 32 spp frame through this library; the PNG it
                      writes is, pixel for pixel, the one pbrlab-hip-cli writes for the same files (which tests/test_io_gpu.py
                      pins against the oracle)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
BIN = os.path.join(ROOT, "oracle", "_ref", "pbrlab-cli-ref")


def _tiny_scene(d):
    with open(os.path.join(d, "m.mtl"), "w") as f:
        f.write("newmtl white\nbase_color 0.7 0.7 0.7\nspecular 0\nnewmtl shiny\nbase_color 0.2 0.4 0.8\nspecular 1.0\nroughness 0.2\n")
    with open(os.path.join(d, "q.obj"), "w") as f:
        f.write("mtllib m.mtl\nv -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nv -0.3 -0.3 1\nv 0.3 -0.3 1\nv 0.3 0.3 1\nv -0.3 0.3 1\n"
                "v -0.5 -0.5 0.3\nv 0.5 -0.5 0.3\nv 0.0 0.5 0.5\n"
                "usemtl white\no floor\nf 1 2 3 4\no light_top\nf 8 7 6 5\nusemtl shiny\no wedge\nf 9 10 11\n")
    return ["q.obj"]


def test_reference_cli_sources_compile_unmodified_against_the_shim(tmp_path):
    if not os.path.isdir(os.path.join(REF, "pc")):
        pytest.skip("the reference tree is not present on this machine")
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "pbrlab_amd", "libpbrhip_io.so")):
        g.build()
    if os.path.exists(BIN):
        os.remove(BIN)
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_cli"], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(BIN), r.stdout + r.stderr
    # the include path of that build holds no directory of the reference's src/
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    recipe = mk[mk.index("ref_cli:"):mk.index("clean:")]
    assert "$(REF)/src" not in recipe and "pbrlab-cli.cc" in recipe and "pc-common.cc" in recipe
    import pbrlab_amd as pa
    if pa.device_count() == 0:
        files = _tiny_scene(str(tmp_path))
        run = subprocess.run([BIN] + files, cwd=str(tmp_path), capture_output=True, text=True)
        assert run.returncode != 0 and ("no ROCm-capable device" in run.stderr or "no HIP device" in run.stderr), run.stderr[-500:]
        assert not os.path.exists(os.path.join(str(tmp_path), "rgba.png"))


def test_forwarding_headers_are_one_liners():
    """tests/cpp/fwd: the reference's header names, each nothing but an include of this repository's shim (no reference text)"""
    fwd = os.path.join(ROOT, "tests", "cpp", "fwd")
    names = sorted(os.path.relpath(os.path.join(dp, f), fwd) for dp, _, fs in os.walk(fwd) for f in fs)
    assert names == ["image-utils.h", "io/curve-mesh-io.h", "io/image-io.h", "io/triangle-mesh-io.h", "material-param.h", "render-layer.h",
                     "render.h", "scene.h"]
    for n in names:
        code = [l for l in open(os.path.join(fwd, n)).read().splitlines() if l.strip() and not l.startswith("//")]
        assert code in (['#include "pbrlab_hip.hpp"'], ['#include "pbrlab_hip_io.hpp"']), (n, code)


def test_oracle_thread_count_follows_the_cpu_quota():
    """the GPU boxes report 256 hardware threads and give the job 16 CPUs (cgroup cpu.max): the oracle's pool is sized from what
    the process may use, not from os.cpu_count()"""
    import _oracle as O
    n, t = O.host_threads(), O.oracle_threads()
    assert 1 <= n <= (os.cpu_count() or 1) and n <= t <= (os.cpu_count() or 1) and t <= 2 * n


@pytest.mark.gpu
def test_reference_main_renders_through_this_library(tmp_path):
    import pbrlab_amd as pa
    from pbrlab_amd import io_api
    if pa.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run on an MI355X (there is no CPU fallback)")
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/pbrlab-cli-ref was not built (it is built where /root/reference exists and travels with the snapshot)")
    d_ref, d_own = os.path.join(str(tmp_path), "ref"), os.path.join(str(tmp_path), "own")
    os.makedirs(d_ref), os.makedirs(d_own)
    for d in (d_ref, d_own):
        files = _tiny_scene(d)
    r = subprocess.run([BIN] + files, cwd=d_ref, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "finish pass 32" in r.stdout and "bmin:" in r.stdout           # render.cc:229, pc-common.cc:264-267
    assert "add shape [light_top]" in r.stderr                             # pc-common.cc:146: the reference's own progress lines
    o = subprocess.run([io_api.CLI_PATH] + files, cwd=d_own, capture_output=True, text=True, timeout=600)
    assert o.returncode == 0, o.stderr[-2000:]
    a = io_api.png_decode(open(os.path.join(d_ref, "rgba.png"), "rb").read())
    b = io_api.png_decode(open(os.path.join(d_own, "rgba.png"), "rb").read())
    assert a.shape == (512, 512, 4) and np.array_equal(a, b)
    assert a[..., :3].std() > 5 and a[..., 3].min() == 255
