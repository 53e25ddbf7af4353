"""ad-hoc fuzz of libpbrhip_io against oracle/_ref/libref_io.so (needs both built): python scripts/io_fuzz.py 2>/dev/null"""
import collections, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _iofiles, _refio
from pbrlab_amd import io_api
d = tempfile.mkdtemp()
st = collections.Counter()
for seed in range(300):
    p = os.path.join(d, "case%d" % seed)
    f = _iofiles.write_obj_case(p, seed, crlf=(seed % 4 == 1))
    for line in open(f, newline="").read().replace("\r\n", "\n").replace("\r", "\n").split("\n"):
        t = line.split()
        if t and t[0] == 'f': st["deg%d" % min(len(t) - 1, 5)] += 1
    r = _refio.obj_load(f, d)
    st["ok" if r["ok"] else "fail"] += 1
    st["tris"] += len(r["corners"]) // 9
print(dict(st))
bad = 0
for seed in range(60):
    hp = os.path.join(d, "h%d.hair" % seed)
    kw = [dict(), dict(segments=5), dict(thickness=False), dict(extras=True), dict(min_points=2), dict(segments=1)][seed % 6]
    _iofiles.write_cyhair(hp, seed, **kw)
    for ms in (False, True):
        rok, rv, ri = _refio.hair_load(hp, ms)
        ok, v, i = io_api.LoadCurveMeshAsCubicBezierCurve(hp, ms)
        same = (ok == rok) and np.array_equal(v.view(np.uint32), rv.view(np.uint32)) and np.array_equal(i, ri)
        if not same:
            bad += 1; print("hair mismatch", seed, ms, ok, rok, v.shape, rv.shape)
print("hair bad", bad)
bad = 0; n = 0
rng = np.random.default_rng(0)
DEPTHS = {0: [1, 2, 4, 8, 16], 2: [8, 16], 3: [1, 2, 4, 8], 4: [8, 16], 6: [8, 16]}
for case in range(120):
    color = [0, 2, 3, 4, 6][case % 5]
    depth = DEPTHS[color][(case // 5) % len(DEPTHS[color])]
    c = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    img = rng.integers(0, 1 << depth, size=(h, w, c))
    if case % 3 == 0:
        img = (np.add.outer(np.arange(h), np.arange(w))[:, :, None] * np.ones(c, int)) % (1 << depth)
    pal = trns = None
    if color == 3:
        pal = rng.integers(0, 256, size=(1 << depth, 3))
        if case % 2: trns = rng.integers(0, 256, size=int(rng.integers(1, (1 << depth) + 1))).astype(np.uint8).tobytes()
    elif color in (0, 2) and case % 4 == 1:
        trns = b"".join(int(k).to_bytes(2, "big") for k in img[0, 0])
    fp = os.path.join(d, "t%d.png" % case)
    _iofiles.write_png(fp, img, depth=depth, color=color, interlace=(case // 7) % 2, palette=pal, trns=trns, level=[0, 1, 6, 9][case % 4], seed=case)
    ref = _refio.image_load("t%d.png" % case, d)
    try:
        got = io_api.LoadImageFromFile("t%d.png" % case, d)
    except Exception as e:
        got = None; print("load fail", case, e)
    n += 1
    if ref is None or got is None or ref.shape != got.shape or not np.array_equal(ref.view(np.uint32), got.view(np.uint32)):
        bad += 1; print("png mismatch", case, color, depth, None if ref is None else ref.shape, None if got is None else got.shape)
print("png bad", bad, "of", n)
bad = 0
for case in range(12):
    w, h = [(5, 4), (8, 3), (33, 7), (64, 16)][case % 4]
    img = rng.random((h, w, 3)).astype(np.float32) * 10.0 ** rng.integers(-3, 3)
    if case % 3 == 0: img[:, : w // 2] = 0.25
    fp = os.path.join(d, "t%d.hdr" % case)
    _iofiles.write_hdr(fp, img, rle=(case % 2 == 0))
    ref = _refio.image_load("t%d.hdr" % case, d); got = io_api.LoadImageFromFile("t%d.hdr" % case, d)
    if ref is None or ref.shape != got.shape or not np.array_equal(ref.view(np.uint32), got.view(np.uint32)):
        bad += 1; print("hdr mismatch", case)
print("hdr bad", bad)
rgba = rng.random((37, 53, 4)).astype(np.float32) * 40; count = np.full((37, 53), 32, np.uint32); count[0, 0] = 0; rgba[1, 1] = np.nan; rgba[2, 2] = -1
assert _refio.cli_output("ref.png", d, rgba, count)
io_api.write_layer_png("mine.png", d, rgba, count)
a = io_api.png_decode(open(os.path.join(d, "ref.png"), "rb").read()); b = io_api.png_decode(open(os.path.join(d, "mine.png"), "rb").read())
ref_via_stb = _refio.image_load("mine.png", d)
print("cli png equal:", np.array_equal(a, b), "stb reads mine:", ref_via_stb is not None and np.array_equal((ref_via_stb * 255 + 0.5).astype(np.uint8), b), os.path.getsize(os.path.join(d, "ref.png")), os.path.getsize(os.path.join(d, "mine.png")))
