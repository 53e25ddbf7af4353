"""Generates tests/golden/io/: small OBJ/MTL/CyHair/PNG/HDR inputs and, in io_golden.npz, what the REFERENCE's own
loaders return for them (oracle/_ref/libref_io.so = the reference's vendored tinyobjloader, cyhair.cc,
curve-mesh-io.cc, image-io.cc + stb compiled unmodified).  Needs /root/reference (run oracle/Makefile first).

    python tests/golden/make_io_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _iofiles  # noqa: E402
import _refio  # noqa: E402

OUT = os.path.join(HERE, "io")
OBJ_SEEDS = [0, 1, 3, 5, 7, 10, 11, 21, 35, 42]
HAIR_CASES = [dict(), dict(segments=5), dict(thickness=False), dict(extras=True), dict(min_points=2), dict(segments=1)]
PNG_CASES = [  # (color, depth, interlace, trns)
    (0, 1, 0, False), (0, 4, 1, True), (0, 8, 0, False), (0, 16, 0, True), (2, 8, 0, False), (2, 8, 1, True),
    (2, 16, 0, False), (3, 2, 0, False), (3, 8, 1, True), (4, 8, 0, False), (4, 16, 1, False), (6, 8, 0, False),
    (6, 16, 0, False),
]


JPEG_CASES = [(33, 17, [(1, 1)], {}), (45, 30, [(2, 1), (1, 1), (1, 1)], dict(restart=2)),
              (31, 29, [(1, 2), (1, 1), (1, 1)], dict(interleaved=False)), (40, 24, [(4, 1), (1, 1), (1, 1)], dict(quant16=True)),
              (20, 20, [(1, 1)] * 4, dict(adobe=0, jfif=False)), (26, 22, [(2, 2), (1, 1), (1, 1), (2, 2)], dict(adobe=2, jfif=False)),
              (19, 23, [(1, 1)] * 3, dict(ids=[82, 71, 66]))]
EXR_CASES = [(["A", "B", "G", "R"], False, 3, 0), (["B", "G", "R"], True, 3, 0), (["Y"], True, 2, 0), (["A", "B", "G", "R"], True, 1, 0),
             (["B", "G", "R"], False, 0, 0), (["A", "B", "G", "R"], True, 3, 1),
             (["B", "G", "R"], True, 4, 0), (["A", "B", "G", "R"], False, 4, 0), (["Y"], True, 4, 1)]           # PIZ


def write_other_format_fixtures(out):
    rng = np.random.default_rng(2024)
    h, w = 9, 14
    rgb = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    rgba = rng.integers(0, 256, size=(h, w, 4)).astype(np.uint8)
    rgb[:, :5] = rgb[0, 0]
    names = []

    def add(name, fn, *a, **k):
        fn(os.path.join(out, name), *a, **k)
        names.append(name)
    add("o_24.bmp", _iofiles.write_bmp, rgb)
    add("o_32_topdown.bmp", _iofiles.write_bmp, rgba, bpp=32, top_down=True)
    add("o_pal4.bmp", _iofiles.write_bmp, rng.integers(0, 16, size=(h, w)), bpp=4, palette=rng.integers(0, 256, size=(16, 3)))
    add("o_pal1_v4.bmp", _iofiles.write_bmp, rng.integers(0, 2, size=(h, w)), bpp=1, header=108, palette=rng.integers(0, 256, size=(2, 3)), gap=4)
    add("o_565.bmp", _iofiles.write_bmp, rng.integers(0, 1 << 16, size=(h, w)), bpp=16, masks=(0xF800, 0x07E0, 0x001F))
    add("o_4444_v5.bmp", _iofiles.write_bmp, rng.integers(0, 1 << 16, size=(h, w)), bpp=16, header=124, masks=(0x0F00, 0x00F0, 0x000F, 0xF000))
    add("o_rgb_rle.tga", _iofiles.write_tga, rgb, rle=True, rng=rng)
    add("o_rgba_topdown.tga", _iofiles.write_tga, rgba, top_down=True, id_bytes=b"fixture")
    add("o_555.tga", _iofiles.write_tga, rng.integers(0, 1 << 16, size=(h, w)), kind="rgb16")
    add("o_grey_alpha_rle.tga", _iofiles.write_tga, rgba[..., :2], kind="grey_alpha", rle=True, rng=rng)
    add("o_indexed.tga", _iofiles.write_tga, rng.integers(0, 40, size=(h, w)), kind="indexed", palette=rng.integers(0, 256, size=(37, 3)), rle=True, rng=rng)
    add("o_indexed16.tga", _iofiles.write_tga, rng.integers(0, 300, size=(h, w)), kind="indexed", palette=rng.integers(0, 1 << 16, size=300), pal_bits=16, index16=True)
    add("o.ppm", _iofiles.write_pnm, rgb, comments=True)
    add("o.pgm", _iofiles.write_pnm, rgb[..., 1])
    idx = rng.integers(0, 16, size=(h, w))
    idx[:, :6] = 3
    add("o_plain.gif", _iofiles.write_gif, idx, rng.integers(0, 256, size=(16, 3)))
    add("o_interlaced_transparent.gif", _iofiles.write_gif, idx, rng.integers(0, 256, size=(16, 3)), interlace=True, transparent=3, canvas=(w + 2, h + 3),
        origin=(1, 2), bg_index=5, comment=True)
    add("o_local_palette.gif", _iofiles.write_gif, idx, None, local_palette=rng.integers(0, 256, size=(16, 3)), version=b"87a", clear_every=7, block=19)
    add("o_rgb16.psd", _iofiles.write_psd, rng.integers(0, 65536, size=(3, h, w)), depth=16)
    pl = rng.integers(0, 256, size=(4, h, w))
    pl[:, :, :7] = 200
    add("o_rgba_rle.psd", _iofiles.write_psd, pl, rle=True, seed=5)
    add("o_5ch.psd", _iofiles.write_psd, rng.integers(0, 256, size=(5, h, w)))
    img = (rng.random((5, 8, 3)) * 4).astype(np.float32)
    img[0, 0] = (0, 1e-5, 700.0)
    add("o_radiance.pic", _iofiles.write_hdr, img, rle=True)
    return names


def main():
    os.makedirs(OUT, exist_ok=True)
    g = {}
    for seed in OBJ_SEEDS:
        f = _iofiles.write_obj_case(os.path.join(OUT, "case%d" % seed), seed, crlf=(seed % 4 == 1))
        r = _refio.obj_load(f, OUT)
        assert r["ok"]
        for k in ("vertices", "normals", "texcoords", "corners", "shape_first", "material_ids"):
            g["obj%d_%s" % (seed, k)] = r[k]
        g["obj%d_text" % seed] = np.frombuffer(r["text"].encode(), np.uint8)
    for i, kw in enumerate(HAIR_CASES):
        p = os.path.join(OUT, "strands%d.hair" % i)
        _iofiles.write_cyhair(p, 100 + i, **kw)
        for ms in (0, 1):
            ok, v, idx = _refio.hair_load(p, bool(ms))
            g["hair%d_%d_ok" % (i, ms)] = np.asarray([ok])
            g["hair%d_%d_vertices" % (i, ms)] = v
            g["hair%d_%d_indices" % (i, ms)] = idx
    rng = np.random.default_rng(7)
    for i, (color, depth, interlace, trns) in enumerate(PNG_CASES):
        c = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
        w, h = int(rng.integers(3, 24)), int(rng.integers(3, 24))
        img = rng.integers(0, 1 << depth, size=(h, w, c))
        if i % 2 == 0:
            img = (np.add.outer(np.arange(h), np.arange(w))[:, :, None] * np.ones(c, int)) % (1 << depth)
        pal = t = None
        if color == 3:
            pal = rng.integers(0, 256, size=(1 << depth, 3))
            if trns:
                t = rng.integers(0, 256, size=(1 << depth) // 2).astype(np.uint8).tobytes()
        elif trns:
            t = b"".join(int(k).to_bytes(2, "big") for k in img[0, 0])
        name = "tex%d.png" % i
        _iofiles.write_png(os.path.join(OUT, name), img, depth=depth, color=color, interlace=interlace, palette=pal, trns=t,
                           level=[0, 1, 6, 9][i % 4], seed=i)
        g["png%d" % i] = _refio.image_load(name, OUT)
    for i, (w, h, rle) in enumerate([(5, 4, True), (20, 6, True), (16, 5, False)]):
        img = rng.random((h, w, 3)).astype(np.float32) * 10.0 ** rng.integers(-2, 3)
        img[:, : w // 2] = 0.25
        name = "env%d.hdr" % i
        _iofiles.write_hdr(os.path.join(OUT, name), img, rle=rle)
        g["hdr%d" % i] = _refio.image_load(name, OUT)
    # OpenEXR (written by tinyexr itself): compression NONE/RLE/ZIPS/ZIP, HALF/FLOAT, RGBA / RGB / one channel, both line orders
    for i, (names, half, comp, lo) in enumerate(EXR_CASES):
        w, h = int(rng.integers(5, 40)), int(rng.integers(17, 40))
        if comp == 4:
            h += 30                                # more than one 32-line block
        planes = (rng.random((len(names), h, w)) * 10.0 ** rng.integers(-2, 3)).astype(np.float32)
        planes[:, : h // 2] = 0.5
        if half:
            planes[0, 0, 0], planes[-1, -1, -1] = 1e-7, 70000.0     # half denormal / overflow to inf
        name = "env%d.exr" % i
        assert _refio.save_exr(os.path.join(OUT, name), planes, names, half, comp, lo)
        g["exr%d" % i] = _refio.image_load(name, OUT)
        assert g["exr%d" % i] is not None
    # JPEG: two files from stb_image_write (4:4:4, 4:2:0) and the variants only the test encoder can make
    def plane(h, w, k):
        yy, xx = np.mgrid[0:h, 0:w]
        return np.clip(128 + 90 * np.sin(xx / (3.0 + k)) * np.cos(yy / (4.0 + k)) + rng.normal(0, 8, (h, w)), 0, 255).astype(np.uint8)
    img = np.stack([plane(37, 53, k) for k in range(3)], -1)
    assert _refio.write_jpg(os.path.join(OUT, "photo0.jpg"), img, 95) and _refio.write_jpg(os.path.join(OUT, "photo1.jpg"), img, 60)
    for i, (W, H, samp, kw) in enumerate(JPEG_CASES):
        hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
        planes = [plane(-(-H * s[1] // vmax), -(-W * s[0] // hmax), k) for k, s in enumerate(samp)]
        _iofiles.write_jpeg(os.path.join(OUT, "photo%d.jpg" % (i + 2)), planes, samp, **kw)
    for i in range(len(JPEG_CASES) + 2):
        g["jpg%d" % i] = _refio.image_load("photo%d.jpg" % i, OUT)
        assert g["jpg%d" % i] is not None
    # progressive JPEG (test encoder; the same coefficients as a baseline file)
    for i, (W, H, samp, script, kw) in enumerate([(37, 29, [(2, 2), (1, 1), (1, 1)], "default3", {}), (23, 40, [(1, 1)], "deep1", dict(restart=2)),
                                                  (30, 21, [(1, 1)] * 3, "spectral3", {})]):
        hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
        planes = [plane(-(-H * s[1] // vmax), -(-W * s[0] // hmax), k) for k, s in enumerate(samp)]
        _iofiles.write_jpeg_progressive(os.path.join(OUT, "prog%d.jpg" % i), planes, samp, _iofiles.PROGRESSIVE_SCRIPTS[script], **kw)
        g["pjpg%d" % i] = _refio.image_load("prog%d.jpg" % i, OUT)
        assert g["pjpg%d" % i] is not None
    # the other stb_image formats: BMP, TGA, PNM, GIF, PSD, and a Radiance picture under a non-.hdr name
    for name in write_other_format_fixtures(OUT):
        g["other_" + name] = _refio.image_load(name, OUT)
        assert g["other_" + name] is not None, name
    # output stage of pbrlab-cli (rgba/count -> sRGB -> byte(x*256) -> stb PNG), decoded back by stb
    rgba = (rng.random((19, 23, 4)) * 40).astype(np.float32)
    count = np.full((19, 23), 32, np.uint32)
    count[0, 0] = 0
    rgba[1, 1] = np.nan
    rgba[2, 2] = -1.0
    rgba[3, 3] = 1e9
    assert _refio.cli_output("cli_ref.png", OUT, rgba, count)
    g["cli_rgba"], g["cli_count"] = rgba, count
    g["cli_png_pixels"] = (_refio.image_load("cli_ref.png", OUT) * 255 + 0.5).astype(np.uint8)
    # texture statements
    stmts = ["tex.png", "-colorspace linear tex.png", "-colorspace sRGB my tex.png", "-clamp on -s 1 2 3 tex.png",
             "-bm 0.5 -blendu off -mm 0 1 -o 1 1 1 -t 0 0 0 -texres 512 -imfchan r -type sphere -boost 2 a b.png",
             "-colorspace", "  spaced.png", "-clamp on"]
    res = [_refio.parse_texopt(s) for s in stmts]
    g["texopt_in"] = np.frombuffer("\n".join(stmts).encode(), np.uint8)
    g["texopt_out"] = np.frombuffer("\n".join("%d\t%s\t%s" % (int(a), b, c) for a, b, c in res).encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "io_golden.npz"), **g)
    print("wrote", len(g), "arrays,", sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)), "bytes in", OUT)


if __name__ == "__main__":
    main()
