"""Rows N1/N2 on the CPU: scene ingestion (OBJ/MTL, CyHair) and image I/O of libpbrhip_io against
  * committed golden fixtures = outputs of the REFERENCE's own loaders (tests/golden/make_io_golden.py), and
  * the reference's loaders themselves (oracle/_ref/libref_io.so) on seeded random files, when that library is present.
Bit-exact everywhere (floats compared as uint32)."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

import _iofiles
import _refio
from pbrlab_amd import io_api

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "io")
GOLD = np.load(os.path.join(GOLD_DIR, "io_golden.npz"))
OBJ_SEEDS = [0, 1, 3, 5, 7, 10, 11, 21, 35, 42]
needs_ref = pytest.mark.skipif(not _refio.available(), reason="oracle/_ref/libref_io.so not built (needs /root/reference)")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_obj(o, r):
    """ObjScene vs reference tinyobj result (dict of arrays + text)"""
    assert np.array_equal(bits(o.vertices[:, :3]).reshape(-1), bits(r["vertices"]))
    assert np.all(o.vertices[:, 3] == 1.0)
    assert np.array_equal(bits(o.normals[:, :3]).reshape(-1), bits(r["normals"]))
    tc = np.asarray(r["texcoords"], np.float32).reshape(-1, 2).copy()
    tc[:, 1] = np.float32(1.0) - tc[:, 1]           # triangle-mesh-io.cc:274-277
    assert np.array_equal(bits(o.texcoords), bits(tc))
    first = r["shape_first"]
    assert len(o.meshes) == len(first) - 1
    corners = np.asarray(r["corners"]).reshape(-1, 3)
    mi = 0
    for s, m in enumerate(o.meshes):
        c = corners[first[s]:first[s + 1]]
        assert np.array_equal(m["vertex_ids"].view(np.int32), c[:, 0])
        assert np.array_equal(m["normal_ids"].view(np.int32), c[:, 1])
        assert np.array_equal(m["texcoord_ids"].view(np.int32), c[:, 2])
        nf = len(c) // 3
        assert np.array_equal(m["material_ids"].view(np.int32), r["material_ids"][mi:mi + nf])
        mi += nf
    assert o.text == r["text"]


def test_io_library_exports():
    L = C.CDLL(io_api.LIB_PATH)
    header = open(os.path.join(os.path.dirname(GOLD_DIR), "..", "..", "include", "pbrhip_io.h")).read()
    import re
    declared = sorted(set(re.findall(r"\b(pbrio_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(io_api.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


@pytest.mark.parametrize("seed", OBJ_SEEDS)
def test_obj_golden(seed, capfd):
    o = io_api.ObjScene(os.path.join(GOLD_DIR, "case%d.obj" % seed))
    r = {k: GOLD["obj%d_%s" % (seed, k)] for k in ("vertices", "normals", "texcoords", "corners", "shape_first", "material_ids")}
    r["text"] = GOLD["obj%d_text" % seed].tobytes().decode()
    same_obj(o, r)


@needs_ref
def test_obj_fuzz_vs_reference(tmp_path, capfd):
    d = str(tmp_path)
    ntri = 0
    for seed in range(1000, 1120):
        f = _iofiles.write_obj_case(os.path.join(d, "c%d" % seed), seed, crlf=(seed % 4 == 1))
        r = _refio.obj_load(f, d)
        assert r["ok"]
        same_obj(io_api.ObjScene(f), r)
        ntri += len(r["corners"]) // 9
    assert ntri > 1500


def _write(path, text):
    with open(path, "w", newline="") as f:
        f.write(text)
    return path


EDGE_OBJS = {
    "zero_index": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n",                     # LoadObj fails
    "zero_vt": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/0 2/0 3/0\n",                  # vt index 0 -> -1, accepted
    "atoi_quirk": "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 1\nvt 0 1\nf 1/ 2/ 3/\n",
    "degenerate": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2\nf 1 2 3\n",
    "relative_oob": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -9\nf -1 -2 -3 -7\n",
    "quad_future_vertex": "v 0 0 0\nv 1 0 0\nv 1 1 0\nf 1 2 3 4\ng later\nv 0 1 0\nf 1 2 3 4\n",
    "usemtl_glued": "v 0 0 0\nv 1 0 0\nv 0 1 0\nusemtlFoo\nf 1 2 3\n",
    "vw_negative": "v 0 0 0\nvw 0 -1 0.5\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",
    "lines_only_object": "v 0 0 0\nv 1 0 0\nv 0 1 0\no wire\nl 1 2 3\no solid\nf 1 2 3\n",
    "lines_only_group": "v 0 0 0\nv 1 0 0\nv 0 1 0\ng wire\nl 1 2 3\ng solid\nf 1 2 3\n",
    "empty": "",
    "no_newline_at_end": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3",
    "numbers": "v 1e2 -1.5E-3 +.25\nv 1. .5e1 12abc\nv abc 1e 1e+\nv 0.123456789012345 123456789.123 1e-40\nf 1 2 3\nf 2 3 4\n",
    "concave_ngon": "v 0 0 0\nv 2 0 0\nv 2 2 0\nv 1 0.5 0\nv 0 2 0\nv -1 1 0\nf 1 2 3 4 5 6\nf 6 5 4 3 2 1\n",
    "collinear_ngon": "v 0 0 0\nv 1 0 0\nv 2 0 0\nv 3 0 0\nv 4 0 0\nv 4 1 1\nf 1 2 3 4 5 6\n",
    "degenerate_ngon": "v 0 0 0\nv 0 0 0\nv 0 0 0\nv 0 0 0\nv 0 0 0\nf 1 2 3 4 5\n",
}


@needs_ref
@pytest.mark.parametrize("name", sorted(EDGE_OBJS))
def test_obj_edge_cases_vs_reference(name, tmp_path, capfd):
    f = _write(os.path.join(str(tmp_path), name + ".obj"), EDGE_OBJS[name])
    r = _refio.obj_load(f, str(tmp_path))
    if not r["ok"]:
        with pytest.raises(io_api.PbrIoError):
            io_api.ObjScene(f)
    else:
        same_obj(io_api.ObjScene(f), r)


def test_obj_zero_index_fails_and_missing_file(tmp_path, capfd):
    f = _write(os.path.join(str(tmp_path), "z.obj"), EDGE_OBJS["zero_index"])
    with pytest.raises(io_api.PbrIoError):
        io_api.ObjScene(f)
    with pytest.raises(io_api.PbrIoError):
        io_api.ObjScene(os.path.join(str(tmp_path), "nope.obj"))


def test_material_conversion(tmp_path, capfd):
    """ParseTinyObjMaterial (triangle-mesh-io.cc:139-212): atof / sscanf("%lf %lf %lf") of the unknown-parameter text, defaults
    of CyclesPrincipledBsdfParameter for absent keys, first definition of a key wins, first `newmtl` of a name wins the
    name map.  (triangle-mesh-io.cc needs mpark/variant.hpp and cannot be compiled here: restated, not pinned.)"""
    d = str(tmp_path)
    _write(os.path.join(d, "m.mtl"),
           "newmtl A\nbase_color 0.25 0.5\nsubsurface 1e-1\nspecular 1.0\nspecular 0.0\nroughness\t0.125\nior  1.5abc\n"
           "Pr 0.9\nsheen_tint x\nsubsurface_radius 1 0.2 0.1 7\n\nnewmtl B\nnewmtl A\nbase_color 1 1 1\n")
    f = _write(os.path.join(d, "m.obj"), "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl A\nf 1 2 3\nusemtl B\nf 1 2 3\n")
    o = io_api.ObjScene(f)
    assert o.material_names == ["A", "B", "A"]
    a = o.materials[0]
    assert tuple(a.base_color) == (0.25, 0.5, 0.0)                 # the component that does not scan is 0
    assert a.subsurface == np.float32(0.1) and a.specular == 1.0   # first `specular` wins
    assert a.roughness == 0.125 and a.ior == 1.5                   # atof stops at "abc"
    assert a.sheen_tint == 0.0                                     # atof("x") = 0
    assert tuple(a.subsurface_radius) == (1.0, np.float32(0.2), np.float32(0.1))
    assert a.metallic == 0.0 and a.clearcoat_roughness == np.float32(0.03) and a.base_color_tex_id == 0xFFFFFFFF
    b = o.materials[1]
    assert tuple(b.base_color) == (np.float32(0.8),) * 3 and b.specular == 0.5 and b.ior == np.float32(1.45)
    assert tuple(b.subsurface_color) == (np.float32(0.7), np.float32(0.1), np.float32(0.1))
    assert list(o.meshes[0]["material_ids"]) == [0, 1]             # name map: first A
    assert "param\tior\t 1.5abc" in o.text                         # value = everything after the first blank


def test_texture_statements_golden():
    stmts = GOLD["texopt_in"].tobytes().decode().split("\n")
    want = GOLD["texopt_out"].tobytes().decode().split("\n")
    assert len(stmts) == len(want)
    for stmt, w in zip(stmts, want):
        found, name, cs = io_api.parse_texture_statement(stmt)
        assert "%d\t%s\t%s" % (int(found), name, cs) == w, stmt
    assert want[0] == "1\ttex.png\t" and want[1] == "1\ttex.png\tlinear" and want[2] == "1\tmy tex.png\tsRGB"


def test_obj_with_textures(tmp_path, capfd):
    d = str(tmp_path)
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(5, 7, 3))
    _iofiles.write_png(os.path.join(d, "albedo.png"), img, depth=8, color=2)
    _iofiles.write_png(os.path.join(d, "my sss.png"), img[:, :, :1], depth=8, color=0)
    _write(os.path.join(d, "t.mtl"), "newmtl A\nmap_base_color albedo.png\nmap_subsurface_color -colorspace linear my sss.png\n"
                                      "newmtl B\nmap_base_color missing.png\nnewmtl C\nmap_base_color -colorspace sRGB albedo.png\n")
    f = _write(os.path.join(d, "t.obj"), "mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0.25\nusemtl A\nf 1/1 2/1 3/1\n")
    o = io_api.ObjScene(f)
    assert [t["name"] for t in o.textures] == ["albedo.png", "my sss.png", "albedo.png"]
    assert (o.materials[0].base_color_tex_id, o.materials[0].subsurface_color_tex_id) == (0, 1)
    assert o.materials[1].base_color_tex_id == 0xFFFFFFFF          # failed load: untextured, like LoadTexture -> -1
    assert o.materials[2].base_color_tex_id == 2
    lin = (img / np.float32(255)).astype(np.float32)
    # default / sRGB colour space: de-gamma with powf (image-utils.cc:8-21)
    from _oracle import ref, have_ref
    px = o.textures[0]["pixels"]
    assert px.shape == (5, 7, 3)
    lo = lin <= np.float32(0.04045)
    assert np.array_equal(bits(px[lo]), bits(lin[lo] / np.float32(12.92)))
    approx = ((lin + 0.055) / 1.055) ** 2.4
    assert np.allclose(px[~lo], approx[~lo], rtol=2e-6)
    assert np.array_equal(bits(o.textures[1]["pixels"]), bits(lin[:, :, :1]))   # -colorspace linear: as stored / 255
    assert o.texcoords[0, 1] == np.float32(1.0) - np.float32(0.25)


@pytest.mark.parametrize("case", range(6))
@pytest.mark.parametrize("memory_saving", [0, 1])
def test_hair_golden(case, memory_saving, capfd):
    ok, v, idx = io_api.LoadCurveMeshAsCubicBezierCurve(os.path.join(GOLD_DIR, "strands%d.hair" % case), bool(memory_saving))
    assert ok == bool(GOLD["hair%d_%d_ok" % (case, memory_saving)][0])
    assert np.array_equal(bits(v), bits(GOLD["hair%d_%d_vertices" % (case, memory_saving)]))
    assert np.array_equal(idx, GOLD["hair%d_%d_indices" % (case, memory_saving)])


def test_hair_golden_covers_failure_and_success():
    oks = [bool(GOLD["hair%d_0_ok" % i][0]) for i in range(6)]
    assert any(oks) and not all(oks)          # strands with < 3 points abort the load (curve-util.cc:107-109)
    assert sum(len(GOLD["hair%d_0_indices" % i]) for i in range(6)) > 100


@needs_ref
def test_hair_fuzz_vs_reference(tmp_path, capfd):
    kws = [dict(), dict(segments=5), dict(thickness=False), dict(extras=True), dict(min_points=2), dict(segments=1)]
    for seed in range(200, 260):
        p = os.path.join(str(tmp_path), "h%d.hair" % seed)
        _iofiles.write_cyhair(p, seed, **kws[seed % 6])
        for ms in (False, True):
            rok, rv, ri = _refio.hair_load(p, ms)
            ok, v, i = io_api.LoadCurveMeshAsCubicBezierCurve(p, ms)
            assert ok == rok and np.array_equal(bits(v), bits(rv)) and np.array_equal(i, ri)


def test_hair_bad_files(tmp_path, capfd):
    d = str(tmp_path)
    with open(os.path.join(d, "bad.hair"), "wb") as f:
        f.write(b"NOPE" + bytes(124))
    ok, v, idx = io_api.LoadCurveMeshAsCubicBezierCurve(os.path.join(d, "bad.hair"))
    assert ok and len(v) == 0 and len(idx) == 0      # the reference ignores LoadCyHair's result: an empty mesh
    ok, v, idx = io_api.LoadCurveMeshAsCubicBezierCurve(os.path.join(d, "missing.hair"))
    assert ok and len(idx) == 0
    ok, _, _ = io_api.LoadCurveMeshAsCubicBezierCurve(os.path.join(d, "strands.abc"))
    assert not ok                                     # "unknown data type"


@pytest.mark.parametrize("i", range(13))
def test_png_decode_golden(i, capfd):
    got = io_api.LoadImageFromFile("tex%d.png" % i, GOLD_DIR)
    want = GOLD["png%d" % i]
    assert got.shape == want.shape and np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("i", range(3))
def test_hdr_decode_golden(i, capfd):
    got = io_api.LoadImageFromFile("env%d.hdr" % i, GOLD_DIR)
    want = GOLD["hdr%d" % i]
    assert got.shape == want.shape and np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("i", range(9))
def test_jpeg_decode_golden(i, capfd):
    """baseline JPEG: 4:4:4 / 4:2:0 from stb_image_write; grey, 4:2:2 + restarts, 4:4:0 non-interleaved, 4:1:1 + 16-bit
    tables, Adobe CMYK, YCCK 4:2:0, RGB component ids from the test encoder == stb_image's pixels"""
    got = io_api.LoadImageFromFile("photo%d.jpg" % i, GOLD_DIR)
    want = GOLD["jpg%d" % i]
    assert got.shape == want.shape and np.array_equal(bits(got), bits(want))


def _jpeg_plane(rng, h, w, k):
    yy, xx = np.mgrid[0:h, 0:w]
    return np.clip(128 + 90 * np.sin(xx / (3.0 + k)) * np.cos(yy / (4.0 + k)) + rng.normal(0, 8, (h, w)), 0, 255).astype(np.uint8)


@needs_ref
def test_jpeg_decode_fuzz_vs_reference(tmp_path, capfd):
    d = str(tmp_path)
    rng = np.random.default_rng(31)
    for case in range(40):                      # stb_image_write's files
        w, h, c = int(rng.integers(1, 90)), int(rng.integers(1, 70)), [3, 3, 4, 1][case % 4]
        img = np.stack([_jpeg_plane(rng, h, w, k) for k in range(c)], -1) if case % 3 else rng.integers(0, 256, size=(h, w, c)).astype(np.uint8)
        name = "s%d.jpg" % case
        assert _refio.write_jpg(os.path.join(d, name), img, [95, 90, 50, 10, 100][case % 5])
        want, got = _refio.image_load(name, d), io_api.LoadImageFromFile(name, d)
        assert want is not None and got.shape == want.shape and np.array_equal(bits(got), bits(want)), case
    variants = [("gray", [(1, 1)], {}), ("422", [(2, 1), (1, 1), (1, 1)], {}), ("440", [(1, 2), (1, 1), (1, 1)], {}),
                ("420rst", [(2, 2), (1, 1), (1, 1)], dict(restart=2)), ("411", [(4, 1), (1, 1), (1, 1)], {}),
                ("444rst1", [(1, 1)] * 3, dict(restart=1)), ("q16", [(2, 2), (1, 1), (1, 1)], dict(quant16=True, quant=np.arange(1, 65) * 3)),
                ("nonint", [(2, 1), (1, 1), (1, 1)], dict(interleaved=False)), ("nonint_rst", [(2, 2), (1, 1), (1, 1)], dict(interleaved=False, restart=3)),
                ("rgbids", [(1, 1)] * 3, dict(ids=[82, 71, 66])), ("adobe_rgb", [(1, 1)] * 3, dict(adobe=0, jfif=False)),
                ("adobe_rgb_jfif", [(1, 1)] * 3, dict(adobe=0)), ("cmyk", [(1, 1)] * 4, dict(adobe=0, jfif=False)),
                ("ycck", [(2, 2), (1, 1), (1, 1), (2, 2)], dict(adobe=2, jfif=False)), ("four", [(1, 1)] * 4, {}),
                ("sof1", [(1, 1)] * 3, dict(sof=0xC1)), ("fill", [(2, 2), (1, 1), (1, 1)], dict(fill_bytes=True)),
                ("mixed", [(2, 2), (2, 1), (1, 2)], {})]
    n = 0
    for (W, H) in [(1, 1), (7, 5), (33, 17), (45, 70)]:
        for name, samp, kw in variants:
            hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
            planes = [_jpeg_plane(rng, -(-H * s[1] // vmax), -(-W * s[0] // hmax), k) for k, s in enumerate(samp)]
            f = "v%d_%s.jpg" % (n, name)
            _iofiles.write_jpeg(os.path.join(d, f), planes, samp, **kw)
            want, got = _refio.image_load(f, d), io_api.LoadImageFromFile(f, d)
            assert want is not None and got.shape == want.shape and np.array_equal(bits(got), bits(want)), (W, H, name)
            n += 1


PROG_CASES = [("gray", [(1, 1)], "gray", {}), ("default420", [(2, 2), (1, 1), (1, 1)], "default3", {}), ("spectral444", [(1, 1)] * 3, "spectral3", {}),
              ("deep", [(1, 1)], "deep1", {}), ("dcsep422", [(2, 1), (1, 1), (1, 1)], "dc_separate3", {}), ("rst", [(2, 2), (1, 1), (1, 1)], "default3", dict(restart=3)),
              ("cmyk", [(1, 1)] * 4, "four", dict(adobe=0, jfif=False)), ("rst1", [(1, 1)], "deep1", dict(restart=1)),
              ("mixed", [(2, 2), (2, 1), (1, 2)], "default3", {})]


@pytest.mark.parametrize("i", range(3))
def test_progressive_jpeg_golden(i, capfd):
    """progressive JPEG (spectral selection + successive approximation, EOB runs, refinement passes) == stb_image's pixels"""
    got = io_api.LoadImageFromFile("prog%d.jpg" % i, GOLD_DIR)
    want = GOLD["pjpg%d" % i]
    assert got.shape == want.shape and np.array_equal(bits(got), bits(want))


@needs_ref
def test_progressive_jpeg_vs_reference(tmp_path, capfd):
    """progressive files from the test encoder, several scan scripts: this decoder == the reference's stb_image == the baseline
    file of the same coefficients (which pins the test encoder too)"""
    d = str(tmp_path)
    rng = np.random.default_rng(41)
    n = 0
    for (W, H) in [(1, 1), (9, 7), (33, 17), (47, 70)]:
        for name, samp, script, kw in PROG_CASES:
            hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
            planes = [_jpeg_plane(rng, -(-H * s[1] // vmax), -(-W * s[0] // hmax), k) for k, s in enumerate(samp)]
            if n % 3 == 0:
                planes = [rng.integers(0, 256, size=p.shape).astype(np.uint8) for p in planes]      # noise: many non-zero coefficients
            quant = [None, np.arange(1, 65), np.full(64, 2)][n % 3]
            fb, fp = "b%d_%s.jpg" % (n, name), "p%d_%s.jpg" % (n, name)
            bkw = {k: v for k, v in kw.items() if k != "restart"}
            _iofiles.write_jpeg(os.path.join(d, fb), planes, samp, quant=quant, **bkw)
            _iofiles.write_jpeg_progressive(os.path.join(d, fp), planes, samp, _iofiles.PROGRESSIVE_SCRIPTS[script], quant=quant, **kw)
            base, want, got = _refio.image_load(fb, d), _refio.image_load(fp, d), io_api.LoadImageFromFile(fp, d)
            assert want is not None and np.array_equal(bits(base), bits(want)), (W, H, name)
            assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (W, H, name)
            n += 1
    # mutations: same verdict and pixels as the reference wherever it still decodes
    data = bytearray(open(os.path.join(d, fp), "rb").read())
    same = total = 0
    for k in range(150):
        m = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(20, len(m)))] = int(rng.integers(0, 256))
        open(os.path.join(d, "mut.jpg"), "wb").write(bytes(m))
        want = _refio.image_load("mut.jpg", d)
        try:
            got = io_api.LoadImageFromFile("mut.jpg", d)
        except io_api.PbrIoError:
            got = None
        total += 1
        same += (want is None and got is None) or (want is not None and got is not None and got.shape == want.shape and np.array_equal(bits(got), bits(want)))
    assert same >= 0.95 * total, (same, total)


@pytest.mark.parametrize("i", range(9))
def test_exr_decode_golden(i, capfd):
    """scanline OpenEXR written by tinyexr (NONE/RLE/ZIPS/ZIP/PIZ, HALF/FLOAT, 1/3/4 channels, both line orders) == tinyexr's LoadEXR"""
    got = io_api.LoadImageFromFile("env%d.exr" % i, GOLD_DIR)
    want = GOLD["exr%d" % i]
    assert got.shape == want.shape and got.shape[2] == 4 and np.array_equal(bits(got), bits(want))


@needs_ref
def test_exr_decode_fuzz_vs_reference(tmp_path, capfd):
    d = str(tmp_path)
    rng = np.random.default_rng(21)
    nimg = 0
    for case in range(64):
        comp = case % 4
        half = (case // 4) % 2 == 1
        names = [["A", "B", "G", "R"], ["B", "G", "R"], ["Y"], ["B", "G", "R", "Z"], ["A", "B", "G"]][(case // 8) % 5]
        w, h = int(rng.integers(1, 70)), int(rng.integers(1, 50))
        planes = (rng.random((len(names), h, w)) * 10.0 ** rng.integers(-3, 3)).astype(np.float32)
        if case % 3 == 0:
            planes[:, : h // 2] = 0.5
        name = "t%d.exr" % case
        assert _refio.save_exr(os.path.join(d, name), planes, names, half, comp, (case // 32) % 2)
        want = _refio.image_load(name, d)
        if want is None:                       # e.g. no R channel: tinyexr refuses, so must this library
            with pytest.raises(io_api.PbrIoError):
                io_api.LoadImageFromFile(name, d)
            continue
        got = io_api.LoadImageFromFile(name, d)
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (case, comp, half, names)
        nimg += 1
    assert nimg >= 48


@needs_ref
def test_exr_piz_vs_reference(tmp_path, capfd):
    """PIZ-compressed scanline OpenEXR (bitmap + range table, wavelet, Huffman with run lengths) written by tinyexr == tinyexr's
    LoadEXR: noise, flat areas, smooth gradients, few distinct values; HALF and FLOAT; 1/3/4 channels; blocks of 32 lines"""
    d = str(tmp_path)
    rng = np.random.default_rng(23)
    nimg = 0
    for case in range(36):
        names = [["A", "B", "G", "R"], ["B", "G", "R"], ["Y"]][case % 3]
        half = bool(case % 2)
        w, h = [(1, 1), (2, 1), (1, 5), (33, 1)][case] if case < 4 else (int(rng.integers(1, 90)), int(rng.integers(1, 110)))
        kind = case % 4
        if kind == 0:
            planes = (rng.random((len(names), h, w)) * 10.0 ** rng.integers(-2, 3)).astype(np.float32)
        elif kind == 1:
            planes = np.full((len(names), h, w), 0.5, np.float32)
            planes[:, : h // 2] = 0.25
        elif kind == 2:
            yy, xx = np.mgrid[0:h, 0:w]
            planes = np.stack([np.sin(xx / 7.0 + k) * np.cos(yy / 5.0) + 1.5 for k in range(len(names))]).astype(np.float32)
        else:
            planes = rng.integers(0, 4, size=(len(names), h, w)).astype(np.float32) * 0.125
        name = "z%d.exr" % case
        if not _refio.save_exr_isolated(os.path.join(d, name), planes, names, half, 4, (case // 18) % 2):
            continue                              # tinyexr's own PIZ writer crashed on this input
        want = _refio.image_load(name, d)
        assert want is not None, case
        got = io_api.LoadImageFromFile(name, d)
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (case, names, half, w, h, kind)
        nimg += 1
    assert nimg >= 30


@needs_ref
def test_png_decode_fuzz_vs_reference(tmp_path, capfd):
    d = str(tmp_path)
    rng = np.random.default_rng(11)
    depths = {0: [1, 2, 4, 8, 16], 2: [8, 16], 3: [1, 2, 4, 8], 4: [8, 16], 6: [8, 16]}
    for case in range(100):
        color = [0, 2, 3, 4, 6][case % 5]
        depth = depths[color][(case // 5) % len(depths[color])]
        c = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        img = rng.integers(0, 1 << depth, size=(h, w, c))
        if case % 3 == 0:
            img = (np.add.outer(np.arange(h), np.arange(w))[:, :, None] * np.ones(c, int)) % (1 << depth)
        pal = trns = None
        if color == 3:
            pal = rng.integers(0, 256, size=(1 << depth, 3))
            if case % 2:
                trns = rng.integers(0, 256, size=int(rng.integers(1, (1 << depth) + 1))).astype(np.uint8).tobytes()
        elif color in (0, 2) and case % 4 == 1:
            trns = b"".join(int(k).to_bytes(2, "big") for k in img[0, 0])
        name = "t%d.png" % case
        _iofiles.write_png(os.path.join(d, name), img, depth=depth, color=color, interlace=(case // 7) % 2, palette=pal,
                           trns=trns, level=[0, 1, 6, 9][case % 4], seed=case)
        want = _refio.image_load(name, d)
        got = io_api.LoadImageFromFile(name, d)
        assert want is not None and got.shape == want.shape and np.array_equal(bits(got), bits(want)), (case, color, depth)


OTHER_FIXTURES = sorted(k[len("other_"):] for k in GOLD.files if k.startswith("other_"))


@pytest.mark.parametrize("name", OTHER_FIXTURES)
def test_other_formats_golden(name, capfd):
    """BMP / TGA / PNM / GIF / PSD fixtures (and a Radiance picture under a non-.hdr name) == the reference's stb_image pixels"""
    got = io_api.LoadImageFromFile(name, GOLD_DIR)
    want = GOLD["other_" + name]
    assert got.shape == want.shape and np.array_equal(bits(got), bits(want))


def test_other_formats_golden_is_complete():
    assert len(OTHER_FIXTURES) >= 21 and {n.rsplit(".", 1)[1] for n in OTHER_FIXTURES} >= {"bmp", "tga", "ppm", "pgm", "gif", "psd", "pic"}


def _other_format_cases(d, rng):
    """writes BMP / TGA / PNM / GIF / PSD files of every flavour stb_image reads into d; yields (file name, expected pixels or
    None).  Expected pixels are known from the generator for the lossless cases, so the generators are checked too."""
    def rgb(h, w, c=3):
        return rng.integers(0, 256, size=(h, w, c)).astype(np.uint8)
    n = 0
    for (w, h) in [(1, 1), (5, 3), (18, 7), (33, 12)]:
        # --- BMP
        img = rgb(h, w)
        for header in (12, 40, 56, 108, 124):
            for td in (False, True):
                if header == 12 and td:
                    continue
                name = "b%d_24_%d_%d.bmp" % (n, header, td); n += 1
                _iofiles.write_bmp(os.path.join(d, name), img, bpp=24, header=header, top_down=td)
                yield name, img
        img4 = rgb(h, w, 4)
        name = "b%d_32.bmp" % n; n += 1
        _iofiles.write_bmp(os.path.join(d, name), img4, bpp=32)
        yield name, img4
        img0 = img4.copy(); img0[..., 3] = 0                  # all-zero alpha bytes: opaque
        name = "b%d_32a0.bmp" % n; n += 1
        _iofiles.write_bmp(os.path.join(d, name), img0, bpp=32)
        e = img0.copy(); e[..., 3] = 255
        yield name, e
        name = "b%d_32v5.bmp" % n; n += 1                      # masks in a V5 header, standard layout
        packed = (img4[..., 3].astype(np.uint64) << 24) | (img4[..., 0].astype(np.uint64) << 16) | (img4[..., 1].astype(np.uint64) << 8) | img4[..., 2]
        _iofiles.write_bmp(os.path.join(d, name), packed, bpp=32, header=124, masks=(0xff0000, 0xff00, 0xff, 0xff000000), compress=3)
        yield name, img4
        for bpp in (1, 4, 8):
            ncol = 1 << bpp
            pal = rng.integers(0, 256, size=(ncol if bpp < 8 else 200, 3))
            idx = rng.integers(0, len(pal), size=(h, w))
            for header, gap in ((40, 0), (108, 8)):
                name = "b%d_p%d_%d.bmp" % (n, bpp, header); n += 1
                _iofiles.write_bmp(os.path.join(d, name), idx, bpp=bpp, header=header, palette=pal, gap=gap)
                yield name, pal[idx].astype(np.uint8)
        # OS/2 header: stb derives the palette size from the data offset with a formula that is 12 bytes short, i.e. it reads 4
        # entries fewer than the file has (and none at all, leaving the colours uninitialised, below 5 entries): only an 8-bit
        # file with a full palette whose last four entries are unused decodes to something defined
        pal = rng.integers(0, 256, size=(256, 3))
        idx = rng.integers(0, 252, size=(h, w))
        name = "b%d_p8_os2.bmp" % n; n += 1
        _iofiles.write_bmp(os.path.join(d, name), idx, bpp=8, header=12, palette=pal)
        yield name, pal[idx].astype(np.uint8)
        words = rng.integers(0, 1 << 16, size=(h, w))
        name = "b%d_555.bmp" % n; n += 1
        _iofiles.write_bmp(os.path.join(d, name), words, bpp=16)
        yield name, None
        for masks in ((0xF800, 0x07E0, 0x001F), (0x0F00, 0x00F0, 0x000F, 0xF000), (0x7C00, 0x03E0, 0x001F, 0x8000)):
            for header in (40, 56, 108):
                if header == 40 and len(masks) == 4:
                    continue
                name = "b%d_bf16_%d.bmp" % (n, header); n += 1
                _iofiles.write_bmp(os.path.join(d, name), words, bpp=16, header=header, masks=masks)
                yield name, None
        words32 = rng.integers(0, 1 << 32, size=(h, w), dtype=np.uint64)
        for masks in ((0x3FC00000, 0x000FF000, 0x000003FC), (0xFF, 0xFF00, 0xFF0000, 0xFF000000), (0x00E00000, 0x00001C00, 0x00000003, 0x80000000)):
            name = "b%d_bf32.bmp" % n; n += 1
            _iofiles.write_bmp(os.path.join(d, name), words32, bpp=32, header=108, masks=masks)
            yield name, None
        # --- TGA
        for td in (False, True):
            for rle in (False, True):
                for c in (3, 4):
                    im = rgb(h, w, c)
                    if rle:
                        im[:, : w // 2] = im[0, 0]
                    name = "t%d_rgb%d_%d%d.tga" % (n, c, td, rle); n += 1
                    _iofiles.write_tga(os.path.join(d, name), im, rle=rle, top_down=td, rng=rng, id_bytes=b"id" if td else b"")
                    yield name, im
                g = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
                if rle:
                    g[h // 2:] = 7
                name = "t%d_grey_%d%d.tga" % (n, td, rle); n += 1
                _iofiles.write_tga(os.path.join(d, name), g, kind="grey", rle=rle, top_down=td, rng=rng)
                yield name, g[..., None]
                ga = rgb(h, w, 2)
                name = "t%d_ga_%d%d.tga" % (n, td, rle); n += 1
                _iofiles.write_tga(os.path.join(d, name), ga, kind="grey_alpha", rle=rle, top_down=td, rng=rng)
                yield name, ga
                w16 = rng.integers(0, 1 << 16, size=(h, w))
                name = "t%d_16_%d%d.tga" % (n, td, rle); n += 1
                _iofiles.write_tga(os.path.join(d, name), w16, kind="rgb16", rle=rle, top_down=td, rng=rng)
                yield name, None
                for pal_bits in (24, 32, 16, 8):
                    npal = int(rng.integers(2, 256))
                    pal = rng.integers(0, 1 << 16, size=npal) if pal_bits == 16 else rng.integers(0, 256, size=(npal, {24: 3, 32: 4, 8: 1}[pal_bits]))
                    idx = rng.integers(0, npal + (3 if pal_bits == 24 else 0), size=(h, w))    # a few indices past the map -> entry 0
                    name = "t%d_idx%d_%d%d.tga" % (n, pal_bits, td, rle); n += 1
                    _iofiles.write_tga(os.path.join(d, name), idx, kind="indexed", palette=pal, pal_bits=pal_bits, rle=rle, top_down=td,
                                       index16=(pal_bits == 32), rng=rng)
                    yield name, (None if pal_bits == 16 else np.asarray(pal, np.uint8)[np.where(idx >= npal, 0, idx)].reshape(h, w, -1))
        # --- PNM
        for comments in (False, True):
            im = rgb(h, w)
            name = "n%d.ppm" % n; n += 1
            _iofiles.write_pnm(os.path.join(d, name), im, comments=comments)
            yield name, im
            name = "n%d.pgm" % n; n += 1
            _iofiles.write_pnm(os.path.join(d, name), im[..., 0], comments=comments)
            yield name, im[..., :1]
        # --- GIF
        for k in range(6):
            ncol = [2, 4, 16, 200, 256, 7][k]
            pal = rng.integers(0, 256, size=(ncol, 3))
            idx = rng.integers(0, ncol, size=(h, w))
            if k % 2:
                idx[:, : w // 2] = idx[0, 0]
            kw = [dict(), dict(interlace=True), dict(transparent=int(idx[0, 0])), dict(canvas=(w + 3, h + 2), origin=(2, 1), bg_index=1),
                  dict(local_palette=rng.integers(0, 256, size=(ncol, 3)), comment=True, version=b"87a"),
                  dict(interlace=True, transparent=1, canvas=(w + 1, h + 1), bg_index=2, clear_every=5, block=17)][k]
            name = "g%d.gif" % n; n += 1
            _iofiles.write_gif(os.path.join(d, name), idx, pal, **kw)
            exp = None
            if k in (0, 1):
                exp = np.concatenate([pal[idx], np.full((h, w, 1), 255)], -1).astype(np.uint8)
            yield name, exp
        # --- PSD
        for nch in (3, 4, 5):
            for depth, rle in ((8, False), (8, True), (16, False)):
                pl = rng.integers(0, 256 if depth == 8 else 65536, size=(nch, h, w))
                if rle:
                    pl[:, :, : w // 2] = 9
                name = "p%d_%d_%d%d.psd" % (n, nch, depth, rle); n += 1
                _iofiles.write_psd(os.path.join(d, name), pl, depth=depth, rle=rle, seed=n)
                exp = None
                if nch == 3:
                    p8 = (pl >> 8) if depth == 16 else pl
                    exp = np.concatenate([np.moveaxis(p8, 0, -1), np.full((h, w, 1), 255)], -1).astype(np.uint8)
                yield name, exp


@needs_ref
def test_bmp_tga_pnm_gif_psd_vs_reference(tmp_path, capfd):
    """every flavour of the other stb_image formats: this library's pixels == the reference's stb_image's, and == what the
    generator put in where that is known"""
    d = str(tmp_path)
    rng = np.random.default_rng(77)
    count = {}
    for name, exp in _other_format_cases(d, rng):
        want = _refio.image_load(name, d)
        assert want is not None, name
        got = io_api.LoadImageFromFile(name, d)
        assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), name
        if exp is not None:
            assert got.shape == exp.shape and np.array_equal((got * 255 + 0.5).astype(np.uint8), exp), name
        count[name.rsplit(".", 1)[1]] = count.get(name.rsplit(".", 1)[1], 0) + 1
    assert min(count[k] for k in ("bmp", "tga", "ppm", "pgm", "gif", "psd")) >= 8, count
    # a Radiance picture under a name that does not end in .hdr goes through stb's 8-bit path (gamma 2.2)
    img = (rng.random((6, 9, 3)) * 3).astype(np.float32)
    img[0, 0] = (0, 1e-5, 700.0)
    _iofiles.write_hdr(os.path.join(d, "sky.pic"), img, rle=True)
    want, got = _refio.image_load("sky.pic", d), io_api.LoadImageFromFile("sky.pic", d)
    assert want.shape == (6, 9, 3) and np.array_equal(bits(got), bits(want))


def _declared_pixels(m):
    """width * height a (mutated) BMP / TGA / PNM / GIF / PSD header announces: the fuzz skips files that announce huge images
    (both decoders would spend seconds filling them with zeros)"""
    m = bytes(m) + bytes(32)
    if m[:2] == b"BM":
        if m[14] == 12:
            w, h = struct.unpack_from("<HH", m, 18)
        else:
            w, h = struct.unpack_from("<ii", m, 18)
    elif m[:4] == b"GIF8":
        w, h = struct.unpack_from("<HH", m, 6)
    elif m[:4] == b"8BPS":
        h, w = struct.unpack_from(">ii", m, 14)
    elif m[:1] == b"P":
        import re
        nums = re.findall(rb"\d+", m[2:64])
        w, h = (int(nums[0]), int(nums[1])) if len(nums) >= 2 else (0, 0)
    else:
        w, h = struct.unpack_from("<HH", m, 12)
    return abs(w) * abs(h)


@needs_ref
def test_other_formats_mutation_fuzz_vs_reference(tmp_path, capfd):
    """byte mutations of valid files: wherever the reference's stb_image still returns an image whose content does not depend on
    bytes it never had (truncated files are skipped), this library returns the same one; neither crashes"""
    d = str(tmp_path)
    rng = np.random.default_rng(78)
    names = [n for n, _ in _other_format_cases(d, rng)]
    checked = agree = 0
    for name in names[::3]:
        data = bytearray(open(os.path.join(d, name), "rb").read())
        for k in range(6):
            m = bytearray(data)
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(0, min(len(m), 64 if k < 4 else len(m))))] = int(rng.integers(0, 256))
            if _declared_pixels(m) > (1 << 18):
                continue
            f = "m_%d_%s" % (k, name)
            open(os.path.join(d, f), "wb").write(bytes(m))
            want = _refio.image_load(f, d)
            try:
                got = io_api.LoadImageFromFile(f, d)
            except io_api.PbrIoError:
                got = None
            checked += 1
            if want is not None and got is not None and got.shape == want.shape and np.array_equal(bits(got), bits(want)):
                agree += 1
            elif want is None and got is None:
                agree += 1
    assert checked > 300 and agree >= 0.9 * checked, (checked, agree)      # the rest: files stb reads past their end


def test_unsupported_image_formats_fail_loudly(tmp_path, capfd):
    d = str(tmp_path)
    rng = np.random.default_rng(0)
    _iofiles.write_jpeg(os.path.join(d, "a.jpg"), [rng.integers(0, 256, size=(8, 8)).astype(np.uint8)], [(1, 1)], sof=0xC2)  # SOF2 over a baseline scan
    with open(os.path.join(d, "a.pic"), "wb") as f:
        f.write(b"\x53\x80\xF6\x34" + bytes(84) + b"PICT" + bytes(32))
    with open(os.path.join(d, "a.bmp"), "wb") as f:
        f.write(b"BM" + bytes(12) + struct.pack("<IiiHHI", 40, 4, 4, 1, 8, 1) + bytes(64))       # BI_RLE8
    with open(os.path.join(d, "a.exr"), "wb") as f:
        f.write(b"\x76\x2f\x31\x01\x02\x02\x00\x00" + bytes(64))     # tiled flag set
    with open(os.path.join(d, "trunc.png"), "wb") as f:
        f.write(open(os.path.join(GOLD_DIR, "tex4.png"), "rb").read()[:60])
    for name in ("a.jpg", "a.pic", "a.bmp", "a.exr", "trunc.png", "missing.png"):
        with pytest.raises(io_api.PbrIoError):
            io_api.LoadImageFromFile(name, d)
    err = capfd.readouterr().err
    assert "progressive JPEG: DC and AC in one scan" in err and "Softimage PIC" in err and "BMP: RLE" in err and "tiled OpenEXR" in err


def _png_decode_python(data):
    """independent decoder for 8-bit non-interlaced PNGs (python zlib): checks the files this library WRITES"""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, = struct.unpack(">I", data[pos:pos + 4])
        t = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == (zlib.crc32(t + body) & 0xFFFFFFFF), "chunk CRC"
        if t == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif t == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, color, _, _, inter = hdr
    assert depth == 8 and inter == 0
    c = {0: 1, 2: 3, 4: 2, 6: 4}[color]
    raw = zlib.decompress(idat)       # verifies the Adler-32 trailer too
    rb = w * c
    out = np.zeros((h, rb), np.uint8)
    prev = np.zeros(rb, np.int32)
    for y in range(h):
        f = raw[y * (rb + 1)]
        cur = np.frombuffer(raw[y * (rb + 1) + 1:(y + 1) * (rb + 1)], np.uint8).astype(np.int32)
        rec = np.zeros(rb, np.int32)
        for x in range(rb):
            a = rec[x - c] if x >= c else 0
            b = prev[x]
            cc = prev[x - c] if x >= c else 0
            if f == 0:
                p = 0
            elif f == 1:
                p = a
            elif f == 2:
                p = b
            elif f == 3:
                p = (a + b) >> 1
            else:
                pa, pb, pc = abs(b - cc), abs(a - cc), abs(a + b - 2 * cc)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
            rec[x] = (cur[x] + p) & 255
        out[y] = rec
        prev = rec
    return out.reshape(h, w, c)


@pytest.mark.parametrize("channels", [1, 2, 3, 4])
def test_png_write_round_trip(channels, tmp_path, capfd):
    rng = np.random.default_rng(channels)
    for (h, w) in [(1, 1), (7, 13), (64, 48)]:
        img = rng.integers(0, 256, size=(h, w, channels)).astype(np.uint8)
        if h > 1:
            img[h // 2:] = (np.add.outer(np.arange(h - h // 2), np.arange(w))[:, :, None] % 256).astype(np.uint8)   # compressible part
        io_api.WritePNG("o.png", str(tmp_path), img)
        data = open(os.path.join(str(tmp_path), "o.png"), "rb").read()
        assert np.array_equal(_png_decode_python(data), img)
        assert np.array_equal(io_api.png_decode(data), img)
    with pytest.raises(io_api.PbrIoError):
        io_api.WritePNG("o.jpg", str(tmp_path), img)     # image-io.cc:180-185: only ".png"


def test_png_writer_compresses(tmp_path, capfd):
    img = np.zeros((256, 256, 4), np.uint8)
    img[..., 0] = np.arange(256)[None, :]
    io_api.WritePNG("flat.png", str(tmp_path), img)
    assert os.path.getsize(os.path.join(str(tmp_path), "flat.png")) < img.size // 20


def test_cli_output_stage_golden(tmp_path, capfd):
    """pbrlab-cli.cc:47-57: rgba/count (count 0 -> NaN -> 255), LinerTosRGB with powf, byte(clamp(x*256, 0, 255))"""
    rgba, count = GOLD["cli_rgba"], GOLD["cli_count"]
    want = GOLD["cli_png_pixels"]
    got = io_api.layer_to_srgb8(rgba, count)
    assert np.array_equal(got, want)
    assert tuple(got[0, 0]) == (255, 255, 255, 255) and tuple(got[1, 1]) == (255, 255, 255, 255) and tuple(got[2, 2][:3]) == (0, 0, 0)
    io_api.write_layer_png("rgba.png", str(tmp_path), rgba, count)
    data = open(os.path.join(str(tmp_path), "rgba.png"), "rb").read()
    assert np.array_equal(_png_decode_python(data), want)
    # the reference's own file decodes to the same pixels with this library's reader
    assert np.array_equal(io_api.png_decode(open(os.path.join(GOLD_DIR, "cli_ref.png"), "rb").read()), want)


@needs_ref
def test_reference_reads_our_png(tmp_path, capfd):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(33, 21, 4)).astype(np.uint8)
    io_api.WritePNG("mine.png", str(tmp_path), img)
    back = _refio.image_load("mine.png", str(tmp_path))      # stb_image
    assert back is not None and np.array_equal((back * 255 + 0.5).astype(np.uint8), img)


def test_cli_binary_usage(capfd):
    import subprocess
    assert os.path.exists(io_api.CLI_PATH)
    r = subprocess.run([io_api.CLI_PATH], capture_output=True, text=True)
    assert r.returncode != 0 and "not specified obj filename" in r.stderr      # pbrlab-cli.cc:24-27
