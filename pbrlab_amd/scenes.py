"""Deterministic procedural stand-ins for pbrlab's demo assets.

The reference's geometry (`data/cornellbox_suzanne_lucy.obj`, cemyuksel's `wCurly.hair`) is not in
the checkout, so the BASELINE configs are restated on generated scenes that carry the reference's
material set (`data/cornellbox_suzanne_lucy.mtl` after its first-occurrence-wins parse, SURVEY.md
Appendix B).  A scene is a plain description (`SceneDesc`) that is *replayed* through the builder
methods of pbrlab's `Scene` (src/scene.h:19-91) by `build_scene`, the way `pc/pc-common.cc:100-237`
drives them: one shared attribute buffer per OBJ, one TriangleMesh + local scene + identity instance
per shape, shapes whose name starts with "light" become area lights with emission (3,3,3).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

NONE = np.uint32(0xFFFFFFFF)

# --------------------------------------------------------------------------------------------------
# materials: src/material-param.h:24-49 defaults + the demo .mtl (Appendix B)
PRINCIPLED_DEFAULTS = dict(
    base_color=(0.8, 0.8, 0.8), subsurface=0.0, subsurface_radius=(1.0, 1.0, 1.0),
    subsurface_color=(0.7, 0.1, 0.1), metallic=0.0, specular=0.5, specular_tint=0.0, roughness=0.5,
    anisotropic=0.0, anisotropic_rotation=0.0, sheen=0.0, sheen_tint=0.5, clearcoat=0.0,
    clearcoat_roughness=0.03, ior=1.45, transmission=0.0, transmission_roughness=0.0,
    base_color_tex_id=0xFFFFFFFF, subsurface_color_tex_id=0xFFFFFFFF)

# src/material-param.h:51-72
HAIR_DEFAULTS = dict(
    coloring_hair=1, base_color=(0.18, 0.06, 0.02), melanin=0.5, melanin_redness=0.8,
    melanin_randomize=0.0, roughness=0.2, azimuthal_roughness=0.3, ior=1.55, shift=2.0,
    specular_tint=(1.0, 1.0, 1.0), second_specular_tint=(1.0, 1.0, 1.0), transmission_tint=(1.0, 1.0, 1.0))


def demo_materials(variant: str):
    """The 8 materials of the demo .mtl, in file order.  variant: 'lambert' (C1: specular 0 everywhere,
    no subsurface), 'ggx' (C2: Lucy's subsurface forced to 0), 'sss' (C3: the .mtl as parsed)."""
    assert variant in ("lambert", "ggx", "sss")

    def m(name, **kw):
        d = dict(PRINCIPLED_DEFAULTS)
        d.update(kw)
        d["name"] = name
        d["kind"] = "principled"
        return d

    mats = [
        m("Floor", base_color=(0.8, 0.8, 0.8), specular=0.0),
        m("Light", base_color=(0.0, 0.0, 0.0), specular=0.0),
        m("Monkey", base_color=(0.8, 0.5, 0.2), specular=1.0, roughness=0.01),
        m("Lucy", base_color=(1.0, 0.8, 0.8), subsurface=1.0, subsurface_radius=(1.0, 0.2, 0.1),
          subsurface_color=(1.0, 0.8, 0.8), specular=1.0, roughness=0.2),
        m("Reflective", base_color=(0.2, 0.2, 0.8), specular=0.0),
        m("Wall_Green", base_color=(0.023333, 0.4096, 0.047991), specular=0.0),
        m("Wall_Red", base_color=(0.4096, 0.050353, 0.037544), specular=0.0),
        m("Wall_White", base_color=(0.8, 0.8, 0.8), specular=0.0),
    ]
    if variant == "lambert":
        for d in mats:
            d["specular"] = 0.0
            d["subsurface"] = 0.0
    elif variant == "ggx":
        mats[3]["subsurface"] = 0.0
    return mats


# --------------------------------------------------------------------------------------------------
@dataclass
class Shape:
    name: str
    vertex_ids: np.ndarray            # (F,3) u32 into the shared attribute
    normal_ids: Optional[np.ndarray]  # (F,3) u32 or None
    material_ids: np.ndarray          # (F,) u32 (scene-global material ids)
    texcoord_ids: Optional[np.ndarray] = None  # (F,3) u32 into SceneDesc.texcoords or None
    transform: Optional[np.ndarray] = None     # (4,4) instance transform, row-vector convention v' = v M (None = identity)


@dataclass
class CurveShape:
    name: str
    vertices: np.ndarray   # (4*S, 4) xyz + radius, 4 control points per segment (not shared)
    indices: np.ndarray    # (S,) u32 first control point
    material: dict = field(default_factory=lambda: dict(HAIR_DEFAULTS, kind="hair", name="hair"))
    transform: Optional[np.ndarray] = None     # (4,4) instance transform (None = identity)


@dataclass
class SceneDesc:
    vertices: np.ndarray               # (V,4) xyzw, w = 1
    normals: np.ndarray                # (N,4)
    materials: List[dict]
    shapes: List[Shape]
    curves: List[CurveShape] = field(default_factory=list)
    light_emission: tuple = (3.0, 3.0, 3.0)   # pc/pc-common.cc:174
    texcoords: Optional[np.ndarray] = None     # (T,2) uv, shared attribute (mesh/attribute.h:11)
    textures: List[np.ndarray] = field(default_factory=list)  # (H,W,C) float32; material *_tex_id index this list

    def num_triangles(self):
        return int(sum(len(s.vertex_ids) for s in self.shapes))

    def num_segments(self):
        return int(sum(len(c.indices) for c in self.curves))


class _Builder:
    def __init__(self):
        self.v, self.n, self.shapes = [], [], []
        self.nv = self.nn = 0

    def add(self, name, verts, faces, material_id, normals=None):
        verts = np.asarray(verts, np.float32).reshape(-1, 3)
        faces = np.asarray(faces, np.uint32).reshape(-1, 3)
        v4 = np.concatenate([verts, np.ones((len(verts), 1), np.float32)], 1)
        self.v.append(v4)
        nid = None
        if normals is not None:
            normals = np.asarray(normals, np.float32).reshape(-1, 3)
            self.n.append(np.concatenate([normals, np.ones((len(normals), 1), np.float32)], 1))
            nid = (faces + np.uint32(self.nn)).astype(np.uint32)
            self.nn += len(normals)
        self.shapes.append(Shape(name, (faces + np.uint32(self.nv)).astype(np.uint32), nid,
                                 np.full(len(faces), material_id, np.uint32)))
        self.nv += len(verts)

    def finish(self, materials, curves=()):
        normals = np.concatenate(self.n) if self.n else np.zeros((0, 4), np.float32)
        return SceneDesc(np.concatenate(self.v), normals, materials, self.shapes, list(curves))


def _quad(p0, p1, p2, p3):
    """two triangles (p0,p1,p2), (p0,p2,p3); geometric normal = (p1-p0)x(p2-p0)."""
    return np.array([p0, p1, p2, p3], np.float32), np.array([[0, 1, 2], [0, 2, 3]], np.uint32)


def _icosphere(subdiv):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
         (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11),
         (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = np.array(v, np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array(f, np.int64)
    for _ in range(subdiv):
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        es = np.sort(edges, axis=1)
        uniq, inv = np.unique(es, axis=0, return_inverse=True)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid])
        nf = len(f)
        a, b, c = base + inv[:nf], base + inv[nf:2 * nf], base + inv[2 * nf:]
        f = np.concatenate([np.stack([f[:, 0], a, c], 1), np.stack([f[:, 1], b, a], 1),
                            np.stack([f[:, 2], c, b], 1), np.stack([a, b, c], 1)])
    return v, f


def _value_noise(p, seed):
    """3-octave lattice value noise on points p (N,3), deterministic in seed."""
    rng = np.random.RandomState(seed)
    table = rng.rand(32, 32, 32)
    out = np.zeros(len(p))
    amp, freq = 1.0, 2.0
    for _ in range(3):
        q = p * freq + 11.5
        i = np.floor(q).astype(np.int64)
        fr = q - i
        fr = fr * fr * (3 - 2 * fr)
        acc = 0
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    w = (fr[:, 0] if dx else 1 - fr[:, 0]) * (fr[:, 1] if dy else 1 - fr[:, 1]) * \
                        (fr[:, 2] if dz else 1 - fr[:, 2])
                    acc = acc + w * table[(i[:, 0] + dx) % 32, (i[:, 1] + dy) % 32, (i[:, 2] + dz) % 32]
        out += amp * (acc - 0.5)
        amp *= 0.5
        freq *= 2.0
    return out


def _smooth_normals(v, f):
    n = np.zeros_like(v)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    for k in range(3):
        np.add.at(n, f[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1
    return n / ln


def _torus_knot(nu, nv, p=2, q=3, tube=0.16):
    """(p,q) torus-knot tube: nu steps along the knot, nv around the tube -> 2*nu*nv triangles."""
    u = np.arange(nu) / nu * 2 * np.pi

    def curve(t):
        r = 0.7 + 0.3 * np.cos(q * t)
        return np.stack([r * np.cos(p * t), 0.45 * np.sin(q * t), r * np.sin(p * t)], 1)

    c = curve(u)
    eps = 1e-4
    tan = curve(u + eps) - curve(u - eps)
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    b = np.cross(tan, up)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    n = np.cross(b, tan)
    ang = np.arange(nv) / nv * 2 * np.pi
    # bumpy tube so the stand-in has statue-like surface detail
    rad = tube * (1.0 + 0.15 * np.sin(7 * u)[:, None] * np.cos(3 * ang)[None, :])
    ring = (np.cos(ang)[None, :, None] * n[:, None, :] + np.sin(ang)[None, :, None] * b[:, None, :]) * rad[:, :, None]
    v = (c[:, None, :] + ring).reshape(-1, 3)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    a = (i * nv + j).ravel()
    bb = (((i + 1) % nu) * nv + j).ravel()
    cc = (((i + 1) % nu) * nv + (j + 1) % nv).ravel()
    d = (i * nv + (j + 1) % nv).ravel()
    f = np.concatenate([np.stack([a, bb, cc], 1), np.stack([a, cc, d], 1)])
    return v, f


def _fix_winding_outward(v, f, center):
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    fc = (v[f[:, 0]] + v[f[:, 1]] + v[f[:, 2]]) / 3.0 - center
    flip = (fn * fc).sum(1) < 0
    f = f.copy()
    f[flip] = f[flip][:, [0, 2, 1]]
    return f


def cornell_scene(variant="ggx", seed=1, monkey_subdiv=5, lucy_nu=2048, lucy_nv=128):
    """S-cornell.  Box [-1,1]^3 open towards +z, 0.6x0.6 `light` quad at y=0.99,
    Suzanne stand-in (noise-displaced icosphere, material Monkey), Lucy stand-in (bumpy torus-knot
    tube, material Lucy), small box (material Reflective).  Material ids follow the .mtl order."""
    mats = demo_materials(variant)
    FLOOR, LIGHT, MONKEY, LUCY, REFL, GREEN, RED, WHITE = range(8)
    b = _Builder()
    # walls: inward-facing (p1-p0)x(p2-p0)
    b.add("floor", *_quad((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1)), FLOOR)
    b.add("ceiling", *_quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)), WHITE)
    b.add("back", *_quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)), WHITE)
    b.add("left", *_quad((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1)), RED)
    b.add("right", *_quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)), GREEN)
    # light: faces down (-y)
    b.add("light", *_quad((-0.3, 0.99, -0.3), (0.3, 0.99, -0.3), (0.3, 0.99, 0.3), (-0.3, 0.99, 0.3)), LIGHT)
    # Suzanne stand-in
    v, f = _icosphere(monkey_subdiv)
    disp = 1.0 + 0.35 * _value_noise(v, seed)
    v = v * disp[:, None]
    n = _smooth_normals(v, f)
    v = v * 0.36 + np.array([-0.45, -0.58, 0.15])
    b.add("monkey", v, f, MONKEY, n)
    # Lucy stand-in (upright knot)
    v, f = _torus_knot(lucy_nu, lucy_nv)
    f = f[:, [0, 2, 1]]
    v = v[:, [0, 2, 1]] * np.array([0.42, 0.5, 0.42])
    f = _fix_winding_outward_tube(v, f)
    n = _smooth_normals(v, f)
    v = v + np.array([0.42, -0.36, -0.25])
    b.add("lucy", v, f, LUCY, n)
    # reflective box
    c, h = np.array([0.05, -0.85, 0.55]), np.array([0.18, 0.15, 0.18])
    corners = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1],
                        [-1, 1, 1]], np.float64) * h + c
    faces = np.array([[0, 3, 2], [0, 2, 1], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7], [2, 7, 6],
                      [1, 2, 6], [1, 6, 5], [0, 4, 7], [0, 7, 3]], np.uint32)
    b.add("box", corners, faces, REFL)
    return b.finish(mats)


def _fix_winding_outward_tube(v, f):
    """orient tube triangles consistently: keep the majority orientation w.r.t. smooth offset."""
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    # tube faces: test against the direction from the tube's local ring centre, approximated by the
    # mean of the two ring centres the face touches; cheap proxy = face centre minus 16-ring average.
    fc = (v[f[:, 0]] + v[f[:, 1]] + v[f[:, 2]]) / 3.0
    # ring centres via moving average over the vertex array is overkill; use the global trick:
    # flip everything if the signed volume is negative.
    vol = (fc * fn).sum() / 6.0
    if vol < 0:
        f = f[:, [0, 2, 1]]
    return f


def to_cubic_bezier(cvs, radii):
    """Catmull-Rom (tau=0.5) strand -> cubic Bezier control points, 4 per segment, xyz+radius.
    Restates ToCubicBezierCurve (src/curve-util.cc:79-199): root / in-between / end formulas applied
    to positions and thickness alike.  cvs (n,3) float32, radii (n,) float32, n >= 3."""
    p = np.concatenate([np.asarray(cvs, np.float32), np.asarray(radii, np.float32)[:, None]], 1)
    n = len(p)
    assert n >= 3
    tau = np.float32(0.5)
    tau3 = np.float32(tau / np.float32(3.0))
    a = np.float32((tau + np.float32(1.0)) / np.float32(3.0))
    c23 = np.float32(np.float32(2.0) / np.float32(3.0))
    nseg = n - 1
    out = np.zeros((nseg, 4, 4), np.float32)
    # root
    out[0, 0] = p[0]
    out[0, 1] = a * p[0] + c23 * p[1] - tau3 * p[2]
    out[0, 2] = tau3 * (p[0] - p[2]) + p[1]
    out[0, 3] = p[1]
    # in-between
    for s in range(1, nseg - 1):
        P0, P1, P2, P3 = p[s - 1], p[s], p[s + 1], p[s + 2]
        out[s, 0] = P1
        out[s, 1] = tau3 * (P2 - P0) + P1
        out[s, 2] = tau3 * (P1 - P3) + P2
        out[s, 3] = P2
    # end
    if nseg > 1:
        P0, P1, P2 = p[nseg - 2], p[nseg - 1], p[nseg]
        out[nseg - 1, 0] = P1
        out[nseg - 1, 1] = tau3 * (P2 - P0) + P1
        out[nseg - 1, 2] = (-tau3) * P0 + c23 * P1 + a * P2
        out[nseg - 1, 3] = P2
    return out.reshape(-1, 4)


def _strands_to_bezier(pts, radius):
    """vectorised to_cubic_bezier for an array of strands pts (S, n, 3) with constant radius."""
    S, n, _ = pts.shape
    p = np.concatenate([pts.astype(np.float32), np.full((S, n, 1), radius, np.float32)], 2)
    tau = np.float32(0.5)
    tau3 = np.float32(tau / np.float32(3.0))
    a = np.float32((tau + np.float32(1.0)) / np.float32(3.0))
    c23 = np.float32(np.float32(2.0) / np.float32(3.0))
    nseg = n - 1
    out = np.zeros((S, nseg, 4, 4), np.float32)
    out[:, 0, 0] = p[:, 0]
    out[:, 0, 1] = a * p[:, 0] + c23 * p[:, 1] - tau3 * p[:, 2]
    out[:, 0, 2] = tau3 * (p[:, 0] - p[:, 2]) + p[:, 1]
    out[:, 0, 3] = p[:, 1]
    if nseg > 2:
        P0, P1, P2, P3 = p[:, 0:nseg - 2], p[:, 1:nseg - 1], p[:, 2:nseg], p[:, 3:nseg + 1]
        out[:, 1:nseg - 1, 0] = P1
        out[:, 1:nseg - 1, 1] = tau3 * (P2 - P0) + P1
        out[:, 1:nseg - 1, 2] = tau3 * (P1 - P3) + P2
        out[:, 1:nseg - 1, 3] = P2
    if nseg > 1:
        P0, P1, P2 = p[:, nseg - 2], p[:, nseg - 1], p[:, nseg]
        out[:, nseg - 1, 0] = P1
        out[:, nseg - 1, 1] = tau3 * (P2 - P0) + P1
        out[:, nseg - 1, 2] = (-tau3) * P0 + c23 * P1 + a * P2
        out[:, nseg - 1, 3] = P2
    return out.reshape(-1, 4)


def hair_strands(seed=1, n_strands=50000, n_segments=24, head_radius=0.5, center=(0.0, 0.0, 0.0),
                 length=0.6, thickness=0.004):
    """S-hair: helical 'curly' strands rooted on the upper part of a head-sized sphere."""
    rng = np.random.RandomState(seed)
    z = rng.uniform(-0.2, 1.0, n_strands)          # root height on the unit sphere (y-up)
    phi = rng.uniform(0, 2 * np.pi, n_strands)
    r = np.sqrt(np.maximum(0.0, 1 - z * z))
    nrm = np.stack([r * np.cos(phi), z, r * np.sin(phi)], 1)
    root = nrm * head_radius
    t = np.linspace(0.0, 1.0, n_segments + 1)[None, :, None]
    # helix frame
    up = np.array([0.0, 1.0, 0.0])
    tx = np.cross(nrm, up)
    ln = np.linalg.norm(tx, axis=1, keepdims=True)
    tx = np.where(ln > 1e-6, tx / np.maximum(ln, 1e-6), np.array([1.0, 0.0, 0.0]))
    ty = np.cross(nrm, tx)
    turns = rng.uniform(2.0, 4.0, n_strands)[:, None, None]
    amp = rng.uniform(0.02, 0.05, n_strands)[:, None, None]
    ph0 = rng.uniform(0, 2 * np.pi, n_strands)[:, None, None]
    ang = ph0 + 2 * np.pi * turns * t
    grav = np.array([0.0, -1.0, 0.0])
    pts = (root[:, None, :] + nrm[:, None, :] * (length * t) * (1 - 0.5 * t) + grav * (0.5 * length * t * t)
           + amp * t * (np.cos(ang) * tx[:, None, :] + np.sin(ang) * ty[:, None, :]))
    pts = pts + np.asarray(center)
    verts = _strands_to_bezier(pts, thickness)
    idx = (np.arange(n_strands * n_segments, dtype=np.uint32) * 4).astype(np.uint32)
    return CurveShape("hair", verts, idx)


def hair_scene(seed=1, n_strands=50000, n_segments=24, head_subdiv=5, with_light=True):
    """S-hair + head mesh (C4): head = icosphere (material Wall_White-like diffuse), a `light` quad and
    a floor so that NEE has an emitter."""
    mats = demo_materials("lambert")
    FLOOR, LIGHT, WHITE = 0, 1, 7
    b = _Builder()
    v, f = _icosphere(head_subdiv)
    n = _smooth_normals(v, f)
    b.add("head", v * 0.5, f, WHITE, n)
    b.add("floor", *_quad((-1.6, -1.2, 1.6), (1.6, -1.2, 1.6), (1.6, -1.2, -1.6), (-1.6, -1.2, -1.6)), FLOOR)
    b.add("back", *_quad((-1.6, -1.2, -1.6), (1.6, -1.2, -1.6), (1.6, 1.6, -1.6), (-1.6, 1.6, -1.6)), WHITE)
    if with_light:
        b.add("light", *_quad((-0.8, 1.55, -0.8), (0.8, 1.55, -0.8), (0.8, 1.55, 0.8), (-0.8, 1.55, 0.8)), LIGHT)
    hair = hair_strands(seed, n_strands, n_segments, head_radius=0.5)
    return b.finish(mats, [hair])


def cornell_hair_scene(variant="sss", seed=1, n_strands=50000, n_segments=24, **kw):
    """C5: S-cornell + S-hair (a tuft of strands on the Suzanne stand-in's side of the box)."""
    d = cornell_scene(variant, seed, **kw)
    hair = hair_strands(seed, n_strands, n_segments, head_radius=0.22, center=(-0.05, 0.15, -0.35), length=0.35,
                        thickness=0.002)
    d.curves.append(hair)
    return d


def textured_cornell_scene(seed=1, **kw):
    """S-cornell with the two texture slots pbrlab reads (`map_base_color`, `map_subsurface_color`,
    cycles-principled-shader.cc:281-301): a checker/gradient base-colour map on the floor (with explicit
    texcoords), a 1-channel map on the back wall (no texcoords: falls back to the barycentrics (u,v),
    mesh/triangle-mesh.cc:130-133) and a subsurface-colour map on the Lucy stand-in."""
    d = cornell_scene("sss", seed, **kw)
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:32, 0:48]
    checker = (((xx // 6) + (yy // 4)) % 2).astype(np.float32)
    tex0 = np.stack([0.15 + 0.7 * checker, 0.2 + 0.6 * (xx / 47.0), 0.25 + 0.5 * (yy / 31.0)], -1).astype(np.float32)
    tex1 = (0.2 + 0.6 * rng.rand(9, 7)).astype(np.float32)[..., None]              # 1 channel: g,b read as 0
    tex2 = np.stack([0.9 * np.ones((16, 16)), 0.3 + 0.6 * rng.rand(16, 16), 0.3 + 0.6 * rng.rand(16, 16),
                     np.ones((16, 16))], -1).astype(np.float32)                        # 4 channels: alpha ignored
    d.textures = [tex0, tex1, tex2]
    d.materials[0] = dict(d.materials[0], base_color_tex_id=0)                       # Floor
    d.materials[7] = dict(d.materials[7], base_color_tex_id=1)                       # Wall_White (ceiling, back)
    d.materials[3] = dict(d.materials[3], subsurface_color_tex_id=2)                 # Lucy
    # explicit texcoords for the floor quad only (uv outside [0,1] exercises the clamp)
    d.texcoords = np.array([[-0.1, -0.1], [1.1, -0.1], [1.1, 1.1], [-0.1, 1.1]], np.float32)
    d.shapes[0].texcoord_ids = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    # Lucy: planar-ish texcoords from the vertex positions
    lucy = next(s for s in d.shapes if s.name == "lucy")
    base = len(d.texcoords)
    vids = np.unique(lucy.vertex_ids)
    remap = np.zeros(int(vids.max()) + 1, np.uint32)
    remap[vids] = np.arange(len(vids), dtype=np.uint32) + base
    uv = d.vertices[vids][:, [0, 1]] * np.float32(0.9) + np.float32(0.45)
    d.texcoords = np.concatenate([d.texcoords, uv.astype(np.float32)])
    lucy.texcoord_ids = remap[lucy.vertex_ids]
    return d


# --------------------------------------------------------------------------------------------------
def build_scene(scene, desc: SceneDesc, make_principled, make_hair):
    """Replay `desc` through a Scene-like object exposing pbrlab's builder names (src/scene.h:19-91),
    in the order pc/pc-common.cc:100-237 uses.  `make_principled(dict)` / `make_hair(dict)` build the
    back end's POD material parameter.  Returns the list of instance ids."""
    mat_ids = []
    tex_ids = [scene.AddTexture(t) for t in desc.textures]   # Scene::AddTexture; ids remapped like FixTextureId
    for m in desc.materials:
        if m.get("kind", "principled") == "principled":
            m = dict(m)
            for k in ("base_color_tex_id", "subsurface_color_tex_id"):
                if m[k] != 0xFFFFFFFF:
                    m[k] = tex_ids[m[k]]
            p = make_principled(m)
        else:
            p = make_hair(m)
        mat_ids.append(scene.AddMaterialParam(p))
    mat_ids = np.asarray(mat_ids, np.uint32)
    instances = []
    for sh in desc.shapes:
        mesh = scene.AddTriangleMesh(desc.vertices, desc.normals, desc.texcoords, sh.vertex_ids, sh.normal_ids,
                                     sh.texcoord_ids, mat_ids[sh.material_ids])
        ls = scene.CreateLocalScene()
        scene.AddMeshToLocalScene(ls, mesh)
        inst = scene.CreateInstance(ls, sh.transform)
        instances.append(inst)
        if sh.name[:5] == "light":
            lid = scene.AddLightParam(desc.light_emission)
            scene.AttachLightParamIdsToInstance(inst, [np.full(len(sh.vertex_ids), lid, np.uint32)])
    for cs in desc.curves:
        mid = scene.AddMaterialParam(make_hair(cs.material))
        mesh = scene.AddCubicBezierCurveMesh(cs.vertices, cs.indices, np.full(len(cs.indices), mid, np.uint32))
        ls = scene.CreateLocalScene()
        scene.AddMeshToLocalScene(ls, mesh)
        instances.append(scene.CreateInstance(ls, cs.transform))
    scene.CommitScene()
    return instances


def instance_matrix(rotate_deg=(0.0, 0.0, 0.0), scale=(1.0, 1.0, 1.0), translate=(0.0, 0.0, 0.0)):
    """4x4 for Scene::CreateInstance in pbrlab's row-vector convention (v' = v M, translation in the last row):
    scale, then rotate about x, y, z, then translate."""
    rx, ry, rz = [np.deg2rad(a) for a in rotate_deg]
    S = np.diag([scale[0], scale[1], scale[2]])
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, sx], [0, -sx, cx]])
    Ry = np.array([[cy, 0, -sy], [0, 1, 0], [sy, 0, cy]])
    Rz = np.array([[cz, sz, 0], [-sz, cz, 0], [0, 0, 1]])
    M = np.eye(4)
    M[:3, :3] = S @ Rx @ Ry @ Rz
    M[3, :3] = translate
    return M.astype(np.float32)


def random_rays(desc_aabb, n, seed=0):
    """rays with origins inside the scene box and uniformly random unit directions (test helper)."""
    lo, hi = [np.asarray(a, np.float64) for a in desc_aabb]
    rng = np.random.RandomState(seed)
    o = lo + (hi - lo) * rng.rand(n, 3)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(n, dtype=[("org", "<f4", 3), ("tmin", "<f4"), ("dir", "<f4", 3), ("tmax", "<f4")])
    rays["org"] = o
    rays["dir"] = d
    rays["tmin"] = 1e-3
    rays["tmax"] = 1.844e18
    return rays
