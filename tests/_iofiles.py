"""Deterministic generators of small OBJ / MTL / CyHair / PNG / HDR inputs for the ingestion tests (rows N1, N2).
Test infrastructure only."""
import os
import struct
import zlib

import numpy as np

NUM_FORMATS = ["%.6f", "%g", "%.9f", "%.3e", "%.12f", "%d.", "%+.4f", "%.7E"]


def _num(rng, x):
    f = NUM_FORMATS[rng.integers(len(NUM_FORMATS))]
    if f == "%d.":
        return "%d." % int(x * 4)
    s = f % x
    if rng.random() < 0.15 and s.startswith("0."):
        s = s[1:]          # ".5"
    if rng.random() < 0.1 and s.startswith("-0."):
        s = "-" + s[2:]    # "-.5"
    return s


def write_obj_case(path, seed, *, crlf=False, mtl=True, polygons=True, oddities=True):
    """Writes <path>.obj (+ <path>.mtl, optionally <path>_b.mtl) and returns the obj file name."""
    rng = np.random.default_rng(seed)
    nl = "\r\n" if crlf else "\n"
    base = os.path.basename(path)
    lines = ["# generated case %d" % seed]
    mats = ["Floor", "Monkey", "Lucy", "light_mat", "dup"]
    if mtl:
        names = base + ".mtl"
        if oddities and seed % 3 == 0:
            names = "missing_file.mtl " + base + ".mtl"
        if oddities and seed % 5 == 0:
            names = base + ".mtl " + base + "_b.mtl"
        lines.append("mtllib " + names)
    nv = 0
    nvn = 0
    nvt = 0
    shapes = int(rng.integers(1, 5))
    for s in range(shapes):
        kind = rng.integers(4)
        if kind == 0:
            lines.append("o " + ("light_%d" % s if rng.random() < 0.3 else "object %d  " % s))
        elif kind == 1:
            lines.append("g " + " ".join(["grp%d" % s] + (["extra", "names"] if rng.random() < 0.3 else [])))
        elif kind == 2 and oddities:
            lines.append("g")
        # a patch of vertices
        n = int(rng.integers(4, 12))
        P = rng.normal(size=(n, 3)) * (10.0 ** rng.integers(-2, 3))
        for p in P:
            sep = "\t" if (oddities and rng.random() < 0.1) else " "
            col = ""
            if oddities and rng.random() < 0.1:
                col = " 0.5 0.25 1"
            lines.append(("  " if oddities and rng.random() < 0.1 else "") + "v" + sep + sep.join(_num(rng, x) for x in p) + col)
        nv += n
        m = int(rng.integers(0, 5))
        for q in rng.normal(size=(m, 3)):
            lines.append("vn " + " ".join(_num(rng, x) for x in q))
        nvn += m
        k = int(rng.integers(0, 5))
        for q in rng.random(size=(k, 2)):
            lines.append("vt " + " ".join(_num(rng, x) for x in q) + (" 0" if rng.random() < 0.2 else ""))
        nvt += k
        if oddities and rng.random() < 0.2:
            lines.append("s %s" % ("off" if rng.random() < 0.5 else "1"))
        nfaces = int(rng.integers(1, 8))
        for f in range(nfaces):
            if mtl and rng.random() < 0.4:
                name = mats[rng.integers(len(mats))]
                if oddities and rng.random() < 0.1:
                    name = "not_there"
                lines.append(("usemtl" if not (oddities and rng.random() < 0.05) else "usemtl\t") + " " + name)
            deg = 3
            r = rng.random()
            if polygons and r < 0.35:
                deg = 4
            elif polygons and r < 0.55:
                deg = int(rng.integers(5, 9))
            if deg <= 4 or rng.random() < 0.5:
                ids = rng.choice(nv, size=min(deg, nv), replace=False)
            else:  # a planar, mostly convex ring among the last n vertices plus noise -> concave cases too
                ids = (nv - n) + rng.permutation(n)[:min(deg, n)]
            style = rng.integers(5)
            toks = []
            for i in ids:
                vi = int(i) + 1
                if style == 4 or (oddities and rng.random() < 0.1):
                    vi = int(i) - nv  # relative
                t = str(vi)
                if style == 1 and nvt:
                    t += "/%d" % (rng.integers(nvt) + 1)
                elif style == 2 and nvn:
                    t += "//%d" % (rng.integers(nvn) + 1)
                elif style == 3 and nvn and nvt:
                    t += "/%d/%d" % (rng.integers(nvt) + 1 if rng.random() < 0.8 else -int(rng.integers(1, nvt + 1)),
                                     rng.integers(nvn) + 1)
                toks.append(t)
            lines.append("f " + ("  " if oddities and rng.random() < 0.1 else " ").join(toks) + (" " if oddities and rng.random() < 0.1 else ""))
        if oddities and rng.random() < 0.15:
            lines.append("l 1 2 3")
        if oddities and rng.random() < 0.1:
            lines.append("p 1")
    if oddities and seed % 7 == 0:
        lines.append("usemtl Floor")   # usemtl on the last line
    text = nl.join(lines) + (nl if seed % 2 == 0 else "")
    if oddities and seed % 11 == 0:
        text = text.replace(nl, "\r", 2)  # a few lone CRs
    with open(path + ".obj", "w", newline="") as f:
        f.write(text)
    if mtl:
        with open(path + ".mtl", "w", newline="") as f:
            f.write(_mtl_text(rng, nl, oddities))
        if oddities and seed % 5 == 0:
            with open(path + "_b.mtl", "w", newline="") as f:
                f.write("newmtl other" + nl + "base_color 1 0 0" + nl)
    return path + ".obj"


def _mtl_text(rng, nl, oddities):
    L = ["# Blender MTL File", ""]

    def fl():
        return "%.6f" % rng.random() if rng.random() < 0.7 else "%g" % (rng.random() * 2)

    def block(name):
        L.append("newmtl " + name)
        L.append("Ns 0.000000")
        L.append("Ka 0.0 0.0 0.0")
        keys = ["base_color", "subsurface", "subsurface_radius", "subsurface_color", "metallic", "specular", "specular_tint",
                "roughness", "anisotropic", "anisotropic_rotation", "sheen", "sheen_tint", "clearcoat", "clearcoat_roughness",
                "ior", "transmission", "transmission_roughness"]
        for k in keys:
            if rng.random() < 0.5:
                continue
            sep = "\t" if (oddities and rng.random() < 0.1) else " "
            if k in ("base_color", "subsurface_radius", "subsurface_color"):
                vals = [fl() for _ in range(3 if rng.random() < 0.9 else 2)]
                L.append(k + sep + " ".join(vals) + ("   " if oddities and rng.random() < 0.2 else ""))
            else:
                L.append(k + sep + fl())
            if oddities and rng.random() < 0.15:   # second definition: the first must win
                L.append(k + " 0.123")
        L.append("Ks 0.5 0.5 0.5")
        L.append("Ke 0 0 0")
        L.append("Ni 1.45")
        L.append("d 1.0")
        if oddities and rng.random() < 0.3:
            L.append("Pr 0.3")
            L.append("aniso 0.1")
            L.append("custom_key  some value with spaces")
            L.append("norm bump.png")
        L.append("illum 2")
        L.append("")

    for name in ["Floor", "Monkey", "Lucy", "light_mat", "dup", "dup"]:
        block(name)
    return nl.join(L)


def write_cyhair(path, seed, *, segments="array", thickness=True, extras=False, min_points=3):
    """Small CyHair file; returns the strand list [(points[n,3], thickness[n])]."""
    rng = np.random.default_rng(seed)
    ns = int(rng.integers(3, 12))
    if segments == "array":
        segs = rng.integers(min_points - 1, 9, size=ns).astype(np.uint16)
    else:
        segs = np.full(ns, int(segments), np.uint16)
    total = int((segs.astype(np.int64) + 1).sum())
    pts = rng.normal(size=(total, 3)).astype(np.float32)
    th = (rng.random(total).astype(np.float32) * 0.05 + 0.001)
    flags = 0x2 | (0x1 if segments == "array" else 0) | (0x4 if thickness else 0) | (0x18 if extras else 0)
    hdr = struct.pack("<4sIIIIff3f88s", b"HAIR", ns, total, flags, 0 if segments == "array" else int(segments), 0.0125, 1.0,
                      0.5, 0.5, 0.5, b"generated")
    assert len(hdr) == 128
    with open(path, "wb") as f:
        f.write(hdr)
        if segments == "array":
            f.write(segs.tobytes())
        f.write(pts.tobytes())
        if thickness:
            f.write(th.tobytes())
        if extras:
            f.write(rng.random(total).astype(np.float32).tobytes())
            f.write(rng.random((total, 3)).astype(np.float32).tobytes())
    return ns, total


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def write_png(path, img, *, depth=8, color=None, interlace=0, palette=None, trns=None, filters="mixed", level=6, seed=0):
    """Independent PNG encoder (python zlib) for decoder tests.  img: [h,w] or [h,w,c] integer samples at `depth` bits."""
    rng = np.random.default_rng(seed)
    img = np.asarray(img)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    if color is None:
        color = {1: 0, 2: 4, 3: 2, 4: 6}[c]

    def pack_rows(sub):
        hh, ww, _ = sub.shape
        if depth == 16:
            rows = sub.astype(">u2").tobytes()
            rb = ww * c * 2
            rows = [rows[i * rb:(i + 1) * rb] for i in range(hh)]
        elif depth == 8:
            rows = [sub[i].astype(np.uint8).tobytes() for i in range(hh)]
        else:
            rows = []
            for i in range(hh):
                bits = np.unpackbits(sub[i].astype(np.uint8).reshape(-1, 1), axis=1)[:, 8 - depth:].reshape(-1)
                rows.append(np.packbits(bits).tobytes())
        bpp = max(1, (c * depth + 7) // 8)
        out = bytearray()
        prev = bytes(len(rows[0])) if rows else b""
        for r in rows:
            f = int(rng.integers(5)) if filters == "mixed" else int(filters)
            cur = bytearray(r)
            enc = bytearray(len(cur))
            for x in range(len(cur)):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                cc = prev[x - bpp] if x >= bpp else 0
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
                enc[x] = (cur[x] - p) & 255
            out.append(f)
            out += enc
            prev = bytes(cur)
        return bytes(out)

    if interlace:
        raw = b""
        for (x0, y0, dx, dy) in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
            sub = img[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += pack_rows(sub)
    else:
        raw = pack_rows(img)
    data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, interlace))
    if palette is not None:
        data += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        data += _chunk(b"tRNS", bytes(trns))
    data += _chunk(b"tEXt", b"Comment\x00generated")
    z = zlib.compress(raw, level)
    half = len(z) // 2
    data += _chunk(b"IDAT", z[:half]) + _chunk(b"IDAT", z[half:]) + _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(data)


def write_hdr(path, img, rle=True):
    """Radiance RGBE writer (float [h,w,3] -> shared exponent), new-style RLE scanlines when rle."""
    img = np.asarray(img, np.float32)
    h, w, _ = img.shape
    m = img.max(axis=2)
    e = np.zeros_like(m, dtype=np.int32)
    nz = m > 1e-32
    mant, ex = np.frexp(m[nz])
    e[nz] = ex + 128
    scale = np.zeros_like(m)
    scale[nz] = mant * 256.0 / m[nz]
    rgbe = np.zeros((h, w, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(nz, e, 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %d +X %d\n" % (h, w))
        if not rle or w < 8 or w >= 32768:
            f.write(rgbe.tobytes())
        else:
            for y in range(h):
                f.write(bytes([2, 2, w >> 8, w & 255]))
                for k in range(4):
                    row = rgbe[y, :, k]
                    i = 0
                    while i < w:
                        run = 1
                        while i + run < w and run < 127 and row[i + run] == row[i]:
                            run += 1
                        if run >= 3:
                            f.write(bytes([128 + run, int(row[i])]))
                            i += run
                        else:
                            j = i
                            while j < w and j - i < 128:
                                if j + 2 < w and row[j] == row[j + 1] == row[j + 2]:
                                    break
                                j += 1
                            j = max(j, i + 1)
                            f.write(bytes([j - i]) + row[i:j].tobytes())
                            i = j
    return rgbe


# ---------------------------------------------------------------------------------------------------------------
# SceneDesc <-> files (end-to-end tests of the CLI path)
def write_desc_as_obj(desc, path, texture_names=None):
    """Dump a pbrlab_amd.scenes.SceneDesc (triangle shapes only) as <path>.obj + <path>.mtl the way Blender exports for
    pbrlab: one `o` per shape, `usemtl` runs, the PBR/SSS extension keys.  Textures must already be files
    (texture_names[i] for desc.textures[i])."""
    base = os.path.basename(path)
    with open(path + ".mtl", "w") as f:
        for i, m in enumerate(desc.materials):
            f.write("newmtl %s\n" % m.get("name", "mat%d" % i).replace(" ", "_"))
            for k in ("base_color", "subsurface_radius", "subsurface_color"):
                f.write("%s %s\n" % (k, " ".join("%.9g" % x for x in m[k])))
            for k in ("subsurface", "metallic", "specular", "specular_tint", "roughness", "anisotropic", "anisotropic_rotation",
                      "sheen", "sheen_tint", "clearcoat", "clearcoat_roughness", "ior", "transmission", "transmission_roughness"):
                f.write("%s %.9g\n" % (k, m[k]))
            for k, key in (("base_color_tex_id", "map_base_color"), ("subsurface_color_tex_id", "map_subsurface_color")):
                if m[k] != 0xFFFFFFFF:
                    f.write("%s -colorspace linear %s\n" % (key, texture_names[m[k]]))
            f.write("\n")
    names = [m.get("name", "mat%d" % i).replace(" ", "_") for i, m in enumerate(desc.materials)]
    with open(path + ".obj", "w") as f:
        f.write("mtllib %s.mtl\n" % base)
        for v in desc.vertices:
            f.write("v %.9g %.9g %.9g\n" % tuple(v[:3]))
        for n in desc.normals:
            f.write("vn %.9g %.9g %.9g\n" % tuple(n[:3]))
        if desc.texcoords is not None:
            for t in desc.texcoords:
                f.write("vt %.9g %.9g\n" % (t[0], 1.0 - float(t[1])))
        for sh in desc.shapes:
            f.write("o %s\n" % sh.name)
            cur = None
            for fi in range(len(sh.vertex_ids)):
                if sh.material_ids[fi] != cur:
                    cur = sh.material_ids[fi]
                    f.write("usemtl %s\n" % names[cur])
                toks = []
                for c in range(3):
                    t = "%d" % (sh.vertex_ids[fi][c] + 1)
                    has_t = sh.texcoord_ids is not None
                    has_n = sh.normal_ids is not None
                    if has_t or has_n:
                        t += "/" + ("%d" % (sh.texcoord_ids[fi][c] + 1) if has_t else "")
                    if has_n:
                        t += "/%d" % (sh.normal_ids[fi][c] + 1)
                    toks.append(t)
                f.write("f " + " ".join(toks) + "\n")
    return path + ".obj"


def desc_from_obj(obj, curves=()):
    """pbrlab_amd.io_api.ObjScene -> SceneDesc holding exactly what the loader produced."""
    from pbrlab_amd import scenes
    mats = []
    for p, name in zip(obj.materials, obj.material_names):
        d = {"kind": "principled", "name": name}
        for k, _ in p._fields_:
            v = getattr(p, k)
            d[k] = tuple(float(x) for x in v) if hasattr(v, "__len__") else v
        mats.append(d)
    shapes = []
    for m in obj.meshes:
        nf = len(m["vertex_ids"]) // 3
        shapes.append(scenes.Shape(m["name"], m["vertex_ids"].reshape(nf, 3), m["normal_ids"].reshape(nf, 3),
                                   m["material_ids"], m["texcoord_ids"].reshape(nf, 3)))
    return scenes.SceneDesc(obj.vertices, obj.normals, mats, shapes, list(curves), texcoords=obj.texcoords,
                            textures=[t["pixels"] for t in obj.textures])


def write_strands_as_cyhair(path, strands, thickness):
    """strands: list of (n_i, 3) float32 arrays (y-up), thickness: list of (n_i,) arrays"""
    segs = np.asarray([len(s) - 1 for s in strands], np.uint16)
    pts = np.concatenate(strands).astype(np.float32)
    th = np.concatenate(thickness).astype(np.float32)
    hdr = struct.pack("<4sIIIIff3f88s", b"HAIR", len(strands), len(pts), 0x1 | 0x2 | 0x4, 0, 0.01, 1.0, 0.5, 0.5, 0.5, b"")
    with open(path, "wb") as f:
        f.write(hdr + segs.tobytes() + pts.tobytes() + th.tobytes())
