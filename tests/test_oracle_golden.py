"""CPU: the oracle against the committed fixtures (tests/golden, made by make_golden.py).
ref_leaf_kats.npz holds outputs of the REFERENCE's own leaf functions; oracle_images.npz pins the
oracle's integrator-level behaviour (oracle outputs, not reference outputs)."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O
from golden.make_golden import golden_scenes, scene_digest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P = O._ptr


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def eq(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(bits(np.nan_to_num(a)), bits(np.nan_to_num(b)))


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(G, "ref_leaf_kats.npz"))


@pytest.fixture(params=["libm", "glibcf"], autouse=False)
def leaf_math(request):
    """The reference's leaf functions were compiled against this image's libm (glibc 2.35) when the vectors were made: the checker has
    to reproduce them with the host libm (where that is this glibc) AND with include/pbr_glibcf.h -- the arithmetic the HIP kernels
    compute with -- on any host: the device's cos / sin / exp / log sit under the very closures the reference's headers define."""
    L = O.lib()
    if request.param == "libm" and not O.libm_is_glibcf():
        pytest.skip("the vectors were made with glibc 2.35's libm; this host has another one")
    L.orc_set_math_mode(O.MATH_GLIBCF if request.param == "glibcf" else O.MATH_LIBM)
    yield request.param
    L.orc_set_math_mode(O.MATH_LIBM)


@pytest.fixture(scope="module")
def img():
    return np.load(os.path.join(G, "oracle_images.npz"))


def test_reference_leaf_vectors(kat, leaf_math):
    L = O.lib()
    for s, want in zip(kat["rng_seeds"], kat["rng_draws"]):
        a = np.zeros(16, np.float32)
        L.orc_kat_rng(int(s), 1234567890, 16, P(a))
        assert eq(a, want)
    for op in (0, 1, 2, 3, 5, 6, 7):
        got = [L.orc_kat_fastmath(op, x, 0) for x in kat[f"fm{op}_x"]]
        assert eq(got, kat[f"fm{op}_y"]), op
    assert eq([L.orc_kat_fastmath(4, y, x) for y, x in kat["fm4_yx"]], kat["fm4_r"])
    u = kat["u2"]
    n = len(u)
    lam, sph, tri = np.zeros((n, 5), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32)
    for i in range(n):
        L.orc_kat_lambert_sample(u[i, 0], u[i, 1], P(lam[i]))
        L.orc_kat_uniform_sphere(u[i, 0], u[i, 1], P(sph[i]))
        L.orc_kat_triangle_sampler(u[i, 0], u[i, 1], P(tri[i]))
    assert eq(lam, kat["lambert"]) and eq(sph, kat["sphere"]) and eq(tri, kat["triangle"])
    assert eq(lam[:, :3], kat["cos_hemi"])
    assert eq([L.orc_kat_fresnel(c, e) for c, e in kat["fresnel_in"]], kat["fresnel"])
    assert eq([L.orc_kat_power_heuristic(a, b) for a, b in kat["mis_in"]], kat["mis"])


def test_reference_ggx_vectors(kat, leaf_math):
    L = O.lib()
    wo, wi, al, u = kat["ggx_wo"], kat["ggx_wi"], kat["ggx_alpha"], kat["u2"]
    for distrib in (1, 2):
        ev, sm = np.zeros((len(wo), 2), np.float32), np.zeros((len(wo), 5), np.float32)
        for i in range(len(wo)):
            L.orc_kat_ggx_eval(P(np.ascontiguousarray(wi[i])), P(np.ascontiguousarray(wo[i])), al[i, 0], al[i, 1], distrib, P(ev[i]))
            L.orc_kat_ggx_sample(P(np.ascontiguousarray(wo[i])), al[i, 0], al[i, 1], u[i, 0], u[i, 1], distrib, P(sm[i]))
        assert eq(ev, kat[f"ggx_eval{distrib}"]) and eq(sm, kat[f"ggx_sample{distrib}"])


def test_reference_hair_vectors(kat):
    L = O.lib()
    n = len(kat["hair_wo"])
    ev, sm = np.zeros((n, 4), np.float32), np.zeros((n, 7), np.float32)
    for i in range(n):
        p, wo, wi, us = (np.ascontiguousarray(kat[k][i]) for k in ("hair_params", "hair_wo", "hair_wi", "hair_us"))
        L.orc_kat_hair_eval(P(wi), P(wo), P(p), P(ev[i]))
        L.orc_kat_hair_sample(P(wo), P(p), P(us), P(sm[i]))
    assert eq(ev, kat["hair_eval"]) and eq(sm, kat["hair_sample"])


def test_reference_misc_vectors(kat, leaf_math):
    L = O.lib()
    for i, want in enumerate(kat["sphere_from_rng"]):   # first draw -> u2 (random-walk-sss.h:296 under g++)
        d, got = np.zeros(2, np.float32), np.zeros(3, np.float32)
        L.orc_kat_rng(i, 1234567890, 2, P(d))
        L.orc_kat_uniform_sphere(d[1], d[0], P(got))
        assert eq(got, want)
    nt = C.c_uint32()
    tiles = np.zeros((510, 4), np.uint32)
    L.orc_create_tiles(1920, 1080, P(tiles, O.u32p), C.byref(nt))
    assert nt.value == 510 and np.array_equal(tiles, kat["tiles_1920x1080"])
    assert tuple(tiles[-1]) == (1856, 1920, 1024, 1080)   # last row is 56 px tall
    out = np.zeros(8 * 16, np.float32)
    assert L.orc_to_cubic_bezier(P(np.ascontiguousarray(kat["bezier_cvs"])), P(np.ascontiguousarray(kat["bezier_radii"])), 9, P(out)) == 8
    assert eq(out.reshape(-1, 4), kat["bezier_out"])


def test_reference_texture_vectors(kat):
    L = O.lib()
    for c in (1, 2, 3, 4):
        px = np.ascontiguousarray(kat[f"tex{c}_pixels"])
        got = np.zeros_like(kat[f"tex{c}_rgb"])
        for i, (u, v) in enumerate(kat[f"tex{c}_uv"]):
            L.orc_kat_texture_fetch(P(px), 7, 5, c, u, v, P(got[i]))
        assert eq(got, kat[f"tex{c}_rgb"]), c
        if c < 3:
            assert not got[:, c:].any()      # channels the image lacks read as 0 (texture.cc:52-58)


@pytest.mark.parametrize("name", ["lambert", "ggx", "sss", "hair", "textured"])
def test_oracle_images_reproduce(img, name):
    desc = golden_scenes()[name]
    assert bytes(img[f"{name}_digest"]).hex() == scene_digest(desc), "scene generator drifted"
    so = O.oracle_scene_from_desc(desc)
    lo, hi = so.FetchSceneAABB()
    assert eq(np.stack([lo, hi]), img[f"{name}_aabb"])
    # the "libm" fixtures were made with this image's libm (glibc 2.35, x86-64 FMA variants): mode glibcf (that libm restated in
    # portable C, the device's arithmetic) must reproduce them on ANY host, mode libm on this image
    modes = [(O.MATH_GLIBCF, "libm"), (O.MATH_F64R, "f64r")] + ([(O.MATH_LIBM, "libm")] if O.libm_is_glibcf() else [])
    for mode, tag in modes:
        rgba, cnt, st = so.render(64, 64, 4, threads=4, math_mode=mode)
        assert (cnt == 4).all()
        assert eq(rgba, img[f"{name}_{tag}_rgba"]), (name, tag)
        assert [st["closest_rays"], st["shadow_rays"], st["rng_draws"]] == list(img[f"{name}_{tag}_rays"])
    xyp, dh = img[f"{name}_trace_xyp"], img[f"{name}_trace_draws_hits"]
    for i in range(0, len(xyp), 5):
        x, y, p = (int(v) for v in xyp[i])
        rad, draws, nh, hits = so.sample_trace(64, 64, x, y, p, max_hits=8)
        assert (draws, nh) == tuple(int(v) for v in dh[i]) and eq(rad, img[f"{name}_trace_radiance"][i])
        ids = img[f"{name}_trace_ids"][i]
        for k in range(min(nh, 8)):
            assert tuple(ids[k]) == (hits[k]["instance_id"], hits[k]["geom_id"], hits[k]["prim_id"])


def test_libm_vs_f64r_tolerance(img):
    """The reference calls libm's float functions (here glibc 2.35's, which the device restates bit for bit); oracle mode f64r
    uses the double result rounded once -- what another, correctly rounded libm would give.  The two differ in the last ulp of a
    few calls; on these images the effect must stay far below the 1e-4 relative-L2 bar of BASELINE.json (the divergent-pixel
    count is reported): the sensitivity of the frames to the libm underneath the reference."""
    for name in ("lambert", "ggx", "sss", "hair", "textured"):
        a, b = img[f"{name}_libm_rgba"][..., :3], img[f"{name}_f64r_rgba"][..., :3]
        rel = np.linalg.norm(a - b) / np.linalg.norm(a)
        px_rel = np.abs(a - b).max(axis=2) / np.maximum(np.abs(a).max(axis=2), 1e-12)
        divergent = int((px_rel > 1e-4).sum())
        print(f"{name}: image rel L2 {rel:.2e}, pixels over 1e-4: {divergent} / {px_rel.size}")
        assert rel < 1e-4, (name, rel)
