"""GPU (-m gpu): rows N1/N2 end to end -- files on disk -> pbrlab-hip-cli (C++ loader -> libpbrhip -> PNG) against the
oracle rendering exactly what the loader produced, resolved with the output stage pinned in tests/test_io_cpu.py.
GPU == oracle[f64r] bit for bit, so the PNGs must be identical byte for byte after decoding."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import _iofiles  # noqa: E402
import _oracle as O  # noqa: E402
from golden.make_golden import golden_scenes  # noqa: E402


@pytest.fixture(scope="module")
def pa():
    import pbrlab_amd as pa
    if pa.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run on an MI355X (there is no CPU fallback)")
    pa.set_device(0)
    return pa


def _write_scene_files(d, name, with_hair):
    from pbrlab_amd import io_api, scenes
    desc = golden_scenes()[name]
    tex_names = []
    for i, t in enumerate(desc.textures):
        tex_names.append("tex %d.png" % i if i == 1 else "tex%d.png" % i)       # one name with a blank
        io_api.WritePNG(tex_names[-1], d, np.ascontiguousarray(t, np.float32))
    files = [_iofiles.write_desc_as_obj(desc, os.path.join(d, name), tex_names)]
    if with_hair:
        cs = scenes.hair_strands(seed=3, n_strands=300, n_segments=5, head_radius=0.12, center=(0.0, 0.45, 0.1), length=0.2,
                                 thickness=0.003)
        # CyHair stores polylines: use the Bezier end points of consecutive segments as the strand points
        v = cs.vertices.reshape(300, 5, 4, 4)
        pts = np.concatenate([v[:, :, 0, :3], v[:, -1:, 3, :3]], axis=1)       # (300, 6, 3)
        th = np.concatenate([v[:, :, 0, 3], v[:, -1:, 3, 3]], axis=1)
        hair = os.path.join(d, "strands.hair")
        _iofiles.write_strands_as_cyhair(hair, list(pts), list(th))
        files.append(hair)
    return files


def _desc_from_files(files):
    from pbrlab_amd import io_api, scenes
    obj = io_api.ObjScene(files[0])
    curves = []
    for f in files[1:]:
        ok, vt, idx = io_api.LoadCurveMeshAsCubicBezierCurve(f)
        assert ok
        curves.append(scenes.CurveShape(f, vt, idx))
    return _iofiles.desc_from_obj(obj, curves)


@pytest.mark.parametrize("name,with_hair", [("textured", False), ("sss", True)])
def test_cli_end_to_end(pa, name, with_hair, tmp_path):
    from pbrlab_amd import io_api
    d = str(tmp_path)
    files = _write_scene_files(d, name, with_hair)
    W, H, SPP = 96, 64, 4
    out = os.path.join(d, "out.png")
    r = subprocess.run([io_api.CLI_PATH] + files + ["--width", str(W), "--height", str(H), "--spp", str(SPP), "--out", out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "finish pass %d" % SPP in r.stdout and "bmin:" in r.stdout        # render.cc:229, pc-common.cc:264-267
    got = io_api.png_decode(open(out, "rb").read())
    assert got.shape == (H, W, 4)

    # the oracle renders what the loader produced
    desc = _desc_from_files(files)
    so = O.oracle_scene_from_desc(desc)
    rgba, count, _ = so.render(W, H, SPP, threads=8, math_mode=O.MATH_DEVICE)
    want = io_api.layer_to_srgb8(rgba, count)
    ndiff = int((got != want).any(axis=2).sum())
    assert ndiff == 0, ndiff
    assert want[..., :3].std() > 10                                           # a real image, not a constant

    # same scene through the Python binding of CreateScene: the float layer is bit-identical too
    sg = io_api.CreateScene(files)
    lo, hi = so.FetchSceneAABB()
    glo, ghi = sg.FetchSceneAABB()
    assert np.array_equal(np.asarray(lo, np.float32), np.asarray(glo, np.float32))
    assert np.array_equal(np.asarray(hi, np.float32), np.asarray(ghi, np.float32))
    layer = pa.RenderLayer()
    pa.Render(sg, W, H, SPP, layer=layer)
    assert np.array_equal(layer.count, count)
    assert np.array_equal(layer.rgba.view(np.uint32), rgba.view(np.uint32))

    # --gpus 2 (two scene copies, interleaved tiles, host sum of disjoint layers) gives the same file contents
    out2 = os.path.join(d, "out2.png")
    r = subprocess.run([io_api.CLI_PATH] + files + ["--width", str(W), "--height", str(H), "--spp", str(SPP), "--out", out2,
                        "--gpus", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(io_api.png_decode(open(out2, "rb").read()), got)
    # --bvh gpu: the tree built on the GPU renders the same file contents
    out3 = os.path.join(d, "out3.png")
    r = subprocess.run([io_api.CLI_PATH] + files + ["--width", str(W), "--height", str(H), "--spp", str(SPP), "--out", out3,
                        "--bvh", "gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(io_api.png_decode(open(out3, "rb").read()), got)


def test_cli_defaults_and_errors(pa, tmp_path):
    """pbrlab-cli.cc:36-57: 512 x 512 x 32 spp into ./rgba.png; a face without material is an error (the reference
    throws from material_ids.at(-1), pc-common.cc:158-161)"""
    from pbrlab_amd import io_api
    d = str(tmp_path)
    with open(os.path.join(d, "m.mtl"), "w") as f:
        f.write("newmtl white\nbase_color 0.7 0.7 0.7\nspecular 0\n")
    with open(os.path.join(d, "q.obj"), "w") as f:
        f.write("mtllib m.mtl\nv -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nv -0.3 -0.3 1\nv 0.3 -0.3 1\nv 0.3 0.3 1\nv -0.3 0.3 1\n"
                "usemtl white\no floor\nf 1 2 3 4\no light_top\nf 8 7 6 5\n")
    r = subprocess.run([io_api.CLI_PATH, "q.obj"], cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    img = io_api.png_decode(open(os.path.join(d, "rgba.png"), "rb").read())
    assert img.shape == (512, 512, 4) and "finish pass 32" in r.stdout
    assert img[..., 3].min() == 255 and img[..., :3].max() > 0
    with open(os.path.join(d, "nomat.obj"), "w") as f:
        f.write("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    r = subprocess.run([io_api.CLI_PATH, "nomat.obj"], cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "no material" in r.stderr
    r = subprocess.run([io_api.CLI_PATH, "missing.obj"], cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
