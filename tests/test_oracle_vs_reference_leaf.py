"""Pins the oracle's leaf functions against the REFERENCE's own headers compiled unmodified
(oracle/ref_harness.cc -> oracle/_ref/libref_leaf.so, built by oracle/Makefile when /root/reference
is present) and against the known-answer values recorded in SURVEY.md Appendix A / §8a-A4.
Bit-exact: both sides are g++/gcc on baseline x86-64 without FMA contraction."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O

needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/libref_leaf.so not built")
P = O._ptr


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.array_equal(bits(a), bits(b)) or (np.isnan(a) == np.isnan(b)).all() and np.array_equal(
        bits(np.nan_to_num(a)), bits(np.nan_to_num(b)))


def unit(rng, n, hemi=False):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
    if hemi:
        v[:, 2] = np.abs(v[:, 2])
    return np.ascontiguousarray(v, np.float32)


# ---------------------------------------------------------------- SURVEY.md known answers (A4, App. A)
def test_pcg32_known_answers():
    L = O.lib()
    a = np.zeros(3, np.float32)
    L.orc_kat_rng(0, 1234567890, 2, P(a))
    assert abs(a[0] - 0.654345155) < 1e-9 and abs(a[1] - 0.380264282) < 1e-9
    # SURVEY §8a-A4 lists RNG(7) draws as printed by a right-to-left printf; as a set they must agree
    L.orc_kat_rng(7, 1234567890, 3, P(a))
    assert sorted(np.round(a, 6)) == sorted(np.round(np.float32([0.610895991, 0.742815733, 0.592899442]), 6))


def test_scalar_known_answers():
    L = O.lib()
    assert abs(L.orc_kat_fresnel(0.5, 1.5) - 0.0891867131) < 1e-9
    assert abs(L.orc_kat_power_heuristic(2, 3) - 0.307692289) < 1e-8
    want = {0: (1, 0, 0.841470957), 1: (1, 0, 0.540302277), 2: (1, 0, 2.71827602), 3: (2, 0, 0.693147182),
            4: (1, 2, 0.46364972), 5: (0.5, 0, 0.523616552)}
    for op, (x, y, v) in want.items():
        assert abs(L.orc_kat_fastmath(op, x, y) - v) < 2e-7, op
    out = np.zeros(5, np.float32)
    L.orc_kat_lambert_sample(0.37, 0.71, P(out))
    assert np.allclose(out, [-0.576809645, 0.614239872, 0.538516521, 0.318309873, 0.17141512], rtol=0, atol=2e-7)


def test_hair_known_answer():
    """SURVEY App. A hair KAT (values printed to 9 digits; wo normalised in float there, so 1e-5)."""
    L = O.lib()
    wo = np.float32([.2, .5, .84])
    wo = wo / np.float32(np.sqrt(np.float32((wo * wo).sum(dtype=np.float32))))
    params = np.float32([.3, .0216, .0054, .0864, .0864, .25, .4, .7, 1.6, 1.55, np.float32(2.0) * np.float32(
        3.141592653589793) / np.float32(180.0)] + [1.0] * 12 + [1.0])
    us = np.float32([.1, .4, .6, .8])
    out = np.zeros(7, np.float32)
    L.orc_kat_hair_sample(P(wo), P(params), P(us), P(out))
    assert np.allclose(out[:3], [-0.287775218, 0.686873317, 0.667375863], atol=2e-5)
    assert np.allclose(out[3:6], [0.053030733, 0.0484344214, 0.0464757569], rtol=2e-4)
    assert abs(out[6] - 0.16235429) < 1e-4


# ---------------------------------------------------------------- bit-exact against the reference headers
@needs_ref
def test_rng_bit_exact():
    L, R = O.lib(), O.ref()
    for seed in [0, 1, 7, 12345, (5 << 32) + 99, 2 ** 63 + 17, 2 ** 64 - 1, 42]:
        a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
        L.orc_kat_rng(seed, 1234567890, 16, P(a))
        R.ref_rng(seed, 1234567890, 16, P(b))
        assert np.array_equal(bits(a), bits(b))
        assert (a >= 0).all() and (a < 1).all()


@needs_ref
def test_fastmath_bit_exact():
    L, R = O.lib(), O.ref()
    rng = np.random.RandomState(1)
    xs = {0: rng.uniform(-20, 20, 2000), 1: rng.uniform(-20, 20, 2000), 2: rng.uniform(-90, 90, 2000),
          3: np.exp(rng.uniform(-80, 80, 2000)), 5: rng.uniform(-1.2, 1.2, 2000), 6: rng.uniform(-130, 130, 2000),
          7: np.exp(rng.uniform(-80, 80, 2000)), 8: rng.uniform(-20, 20, 1000), 9: rng.uniform(-20, 20, 1000)}
    for op, arr in xs.items():
        for x in np.float32(arr):
            a, b = L.orc_kat_fastmath(op, x, 0), R.ref_fastmath(op, x, 0)
            assert same_bits([a], [b]), (op, x, a, b)
    for y, x in np.float32(rng.uniform(-3, 3, (2000, 2))):
        assert same_bits([L.orc_kat_fastmath(4, y, x)], [R.ref_fastmath(4, y, x)])
    for v in np.float32([0.0, -0.0, 1.0, -1.0, 1e-30, 3e38, np.inf]):
        for op in (0, 1, 2, 3, 5):
            assert same_bits([L.orc_kat_fastmath(op, v, 0)], [R.ref_fastmath(op, v, 0)]), (op, v)


@needs_ref
def test_sampling_and_fresnel_bit_exact():
    L, R = O.lib(), O.ref()
    rng = np.random.RandomState(2)
    for u1, u2 in np.float32(rng.rand(1000, 2)):
        a, b = np.zeros(5, np.float32), np.zeros(5, np.float32)
        L.orc_kat_lambert_sample(u1, u2, P(a)), R.ref_lambert_sample(u1, u2, P(b))
        assert same_bits(a, b)
        a3, b3 = np.zeros(3, np.float32), np.zeros(3, np.float32)
        L.orc_kat_uniform_sphere(u1, u2, P(a3)), R.ref_uniform_sphere(u1, u2, P(b3))
        assert same_bits(a3, b3)
        a2, b2 = np.zeros(2, np.float32), np.zeros(2, np.float32)
        L.orc_kat_triangle_sampler(u1, u2, P(a2)), R.ref_triangle_sampler(u1, u2, P(b2))
        assert same_bits(a2, b2)
    for c, eta in np.float32(np.stack([rng.uniform(-1, 1, 2000), rng.uniform(0.0, 3, 2000)], 1)):
        assert same_bits([L.orc_kat_fresnel(c, eta)], [R.ref_fresnel(c, eta)])
    for a_, b_ in np.float32(np.exp(rng.uniform(-10, 10, (2000, 2)))):
        assert same_bits([L.orc_kat_power_heuristic(a_, b_)], [R.ref_power_heuristic(a_, b_)])
    assert L.orc_kat_power_heuristic(2.0, 2.0) == 0.5 == R.ref_power_heuristic(2.0, 2.0)


@needs_ref
def test_sss_direction_argument_order():
    """random-walk-sss.h:296 `UniformSampleSphere(rng.Draw(), rng.Draw())`: with g++ the FIRST draw
    lands in u2.  The oracle hard-codes that order; check it against the compiled line."""
    L, R = O.lib(), O.ref()
    for seed in range(20):
        d = np.zeros(2, np.float32)
        L.orc_kat_rng(seed, 1234567890, 2, P(d))
        want, got = np.zeros(3, np.float32), np.zeros(3, np.float32)
        R.ref_uniform_sphere_from_rng(seed, 1234567890, P(want))
        L.orc_kat_uniform_sphere(d[1], d[0], P(got))   # u1 = second draw, u2 = first draw
        assert same_bits(got, want)


@needs_ref
@pytest.mark.parametrize("distrib", [1, 2])
def test_ggx_bit_exact(distrib):
    L, R = O.lib(), O.ref()
    rng = np.random.RandomState(3 + distrib)
    n = 1500
    wo, wi = unit(rng, n, hemi=True), unit(rng, n)
    alphas = np.float32(np.exp(rng.uniform(np.log(1e-4), 0.0, (n, 2))))
    alphas[::3, 1] = alphas[::3, 0]          # isotropic third
    alphas[1::50] = 1.0
    us = np.float32(rng.rand(n, 2))
    for i in range(n):
        ax, ay = alphas[i]
        a, b = np.zeros(2, np.float32), np.zeros(2, np.float32)
        L.orc_kat_ggx_eval(P(wi[i]), P(wo[i]), ax, ay, distrib, P(a))
        R.ref_ggx_eval(P(wi[i]), P(wo[i]), ax, ay, distrib, P(b))
        assert same_bits(a, b), (i, a, b)
        a5, b5 = np.zeros(5, np.float32), np.zeros(5, np.float32)
        L.orc_kat_ggx_sample(P(wo[i]), ax, ay, us[i, 0], us[i, 1], distrib, P(a5))
        R.ref_ggx_sample(P(wo[i]), ax, ay, us[i, 0], us[i, 1], distrib, P(b5))
        assert same_bits(a5, b5), (i, a5, b5)
    # below-horizon wo: nothing is written (caller's zeros survive)
    wo_b = np.float32([0.3, 0.1, -0.9])
    a5, b5 = np.zeros(5, np.float32), np.zeros(5, np.float32)
    L.orc_kat_ggx_sample(P(wo_b), 0.2, 0.2, 0.3, 0.6, 2, P(a5)), R.ref_ggx_sample(P(wo_b), 0.2, 0.2, 0.3, 0.6, 2, P(b5))
    assert same_bits(a5, b5) and not a5.any()


def _hair_params(rng):
    beta_m, beta_n = rng.uniform(0.05, 1.0), rng.uniform(0.05, 1.0)
    v0 = (0.726 * beta_m + 0.812 * beta_m ** 2 + 3.7 * beta_m ** 20) ** 2
    s = np.sqrt(np.pi / 8) * (0.265 * beta_n + 1.194 * beta_n ** 2 + 5.372 * beta_n ** 22)
    h = rng.uniform(-1, 1)
    p = [h, v0, 0.25 * v0, 4 * v0, 4 * v0, s, *rng.uniform(0.05, 3, 3), rng.uniform(1.2, 1.8), np.radians(
        rng.uniform(0, 10)), *rng.uniform(0.3, 1, 9), 1, 1, 1, 1.0]
    return np.ascontiguousarray(np.float32(p))


@needs_ref
def test_hair_bsdf_bit_exact():
    L, R = O.lib(), O.ref()
    rng = np.random.RandomState(5)
    for i in range(1500):
        params = _hair_params(rng)
        if i % 97 == 0:
            params[0] = np.float32(1.0 if i % 2 else -1.0)     # Q12: |h| == 1
        wo, wi = unit(rng, 1)[0], unit(rng, 1)[0]
        us = np.ascontiguousarray(np.float32(rng.rand(4)))
        a, b = np.zeros(4, np.float32), np.zeros(4, np.float32)
        L.orc_kat_hair_eval(P(wi), P(wo), P(params), P(a)), R.ref_hair_eval(P(wi), P(wo), P(params), P(b))
        assert same_bits(a, b), (i, a, b)
        a7, b7 = np.zeros(7, np.float32), np.zeros(7, np.float32)
        L.orc_kat_hair_sample(P(wo), P(params), P(us), P(a7)), R.ref_hair_sample(P(wo), P(params), P(us), P(b7))
        assert same_bits(a7, b7), (i, a7, b7)


@needs_ref
def test_tiles_bit_exact():
    L, R = O.lib(), O.ref()
    for w, h in [(256, 256), (1920, 1080), (3840, 2160), (1, 1), (64, 64), (65, 63), (130, 7)]:
        na, nb = C.c_uint32(), C.c_uint32()
        L.orc_create_tiles(w, h, None, C.byref(na)), R.ref_create_tiles(w, h, None, C.byref(nb))
        assert na.value == nb.value == ((w + 63) // 64) * ((h + 63) // 64)
        a, b = np.zeros(na.value * 4, np.uint32), np.zeros(nb.value * 4, np.uint32)
        L.orc_create_tiles(w, h, P(a, O.u32p), C.byref(na)), R.ref_create_tiles(w, h, P(b, O.u32p), C.byref(nb))
        assert np.array_equal(a, b)


@needs_ref
def test_cubic_bezier_conversion_bit_exact():
    L, R = O.lib(), O.ref()
    from pbrlab_amd import scenes
    rng = np.random.RandomState(6)
    for n in [3, 4, 5, 9, 25]:
        cvs = np.ascontiguousarray(np.float32(rng.normal(size=(n, 3))))
        rad = np.ascontiguousarray(np.float32(rng.uniform(0.001, 0.05, n)))
        a, b = np.zeros((n - 1) * 16, np.float32), np.zeros((n - 1) * 16, np.float32)
        ka, kb = L.orc_to_cubic_bezier(P(cvs), P(rad), n, P(a)), R.ref_to_cubic_bezier(P(cvs), P(rad), n, P(b))
        assert ka == kb == n - 1 and same_bits(a, b)
        # the product-side numpy restatement used by the scene generators
        assert same_bits(scenes.to_cubic_bezier(cvs, rad).ravel(), b)
    # strands with fewer than 3 points abort the load (curve-util.cc:108-110)
    two = np.zeros(6, np.float32)
    assert L.orc_to_cubic_bezier(P(two), P(two), 2, P(two)) < 0 and R.ref_to_cubic_bezier(P(two), P(two), 2, P(two)) < 0


@needs_ref
def test_mult_v_bit_exact():
    R = O.ref()
    rng = np.random.RandomState(7)
    rows = np.ascontiguousarray(np.float32(rng.normal(size=9)))
    v = np.ascontiguousarray(np.float32([-0.0, 0.5, -0.25]))
    out = np.zeros(3, np.float32)
    R.ref_mult_v(P(v), P(rows), P(out))
    want = [np.float32(np.float32(np.float32(rows[0 + k] * v[0]) + np.float32(rows[3 + k] * v[1])) + np.float32(
        rows[6 + k] * v[2])) + np.float32(0) for k in range(3)]
    assert same_bits(out, want)


@needs_ref
def test_texture_fetch_bit_exact():
    """Texture::FetchFloat3 -> BilinearFilter (texture.cc:43-68, image-utils.cc:99-167), reference compiled unmodified"""
    L, R = O.lib(), O.ref()
    rng = np.random.RandomState(8)
    for (w, h, c) in [(4, 4, 3), (7, 3, 4), (1, 1, 3), (16, 9, 1), (5, 5, 2)]:
        px = np.ascontiguousarray(rng.rand(h, w, c).astype(np.float32))
        uvs = np.float32(np.concatenate([rng.uniform(-0.3, 1.3, (400, 2)), [[0, 0], [1, 1], [0.999999, 0.5], [1, 0], [0.5, 1]]]))
        for u, v in uvs:
            a, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
            L.orc_kat_texture_fetch(P(px), w, h, c, u, v, P(a)), R.ref_texture_fetch(P(px), w, h, c, u, v, P(b))
            assert same_bits(a, b), (w, h, c, u, v, a, b)
