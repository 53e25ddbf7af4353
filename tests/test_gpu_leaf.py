"""GPU (-m gpu): the DEVICE's leaf functions against outputs of the REFERENCE's own leaf code -- directly, no checker in between.

tests/golden/ref_leaf_kats.npz holds what the reference's headers (random/rng.h, pbrlab_math.h, sampler/sampling-utils.h,
closure/{lambert, closure-util, microfacet-ggx, energy-conserving-hair-bsdf}.h), compiled unmodified against this image's libm
(oracle/ref_harness.cc -> oracle/_ref/libref_leaf.so, tests/golden/make_golden.py), return on seeded random inputs.
pbrhip_leaf_eval (include/pbrhip.h, a test hook) evaluates what the HIP kernels compute -- dmath.h, dclosures.h, with the
transcendental functions of include/pbr_glibcf.h -- on the same inputs on the GPU: every output bit must agree."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def pa():
    import pbrlab_amd as pa
    if pa.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run on an MI355X (there is no CPU fallback)")
    pa.set_device(0)
    return pa


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(G, "ref_leaf_kats.npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(got, want):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32).reshape(np.shape(got))
    nan = np.isnan(want)
    return np.array_equal(np.isnan(got), nan) and np.array_equal(bits(np.where(nan, 0, got)), bits(np.where(nan, 0, want)))


def as_bits(v, n):
    return np.full(n, v, np.uint32).view(np.float32)


def test_generator_and_fast_math(pa, kat):
    A = pa.api
    seeds = kat["rng_seeds"].astype(np.uint64)
    seq = np.uint64(1234567890)
    inp = np.zeros((len(seeds), 4), np.uint32)
    inp[:, 0], inp[:, 1] = (seeds & np.uint64(0xFFFFFFFF)).astype(np.uint32), (seeds >> np.uint64(32)).astype(np.uint32)
    inp[:, 2], inp[:, 3] = np.uint32(seq & np.uint64(0xFFFFFFFF)), np.uint32(seq >> np.uint64(32))
    assert same(A.leaf_eval(A.LEAF_RNG, inp.view(np.float32), 16), kat["rng_draws"])
    for op in (0, 1, 2, 3, 5, 6, 7):
        x = kat[f"fm{op}_x"]
        inp = np.stack([as_bits(op, len(x)), x, np.zeros_like(x)], axis=1)
        assert same(A.leaf_eval(A.LEAF_FASTMATH, inp, 1)[:, 0], kat[f"fm{op}_y"]), op
    yx = kat["fm4_yx"]
    inp = np.stack([as_bits(4, len(yx)), yx[:, 0], yx[:, 1]], axis=1)
    assert same(A.leaf_eval(A.LEAF_FASTMATH, inp, 1)[:, 0], kat["fm4_r"])


def test_samplers_fresnel_mis(pa, kat):
    A = pa.api
    u = kat["u2"]
    lam = A.leaf_eval(A.LEAF_LAMBERT, u, 5)
    assert same(lam, kat["lambert"]) and same(lam[:, :3], kat["cos_hemi"])
    assert same(A.leaf_eval(A.LEAF_SPHERE, u, 3), kat["sphere"])
    assert same(A.leaf_eval(A.LEAF_TRIANGLE, u, 2), kat["triangle"])
    assert same(A.leaf_eval(A.LEAF_FRESNEL, kat["fresnel_in"], 1)[:, 0], kat["fresnel"])
    assert same(A.leaf_eval(A.LEAF_MIS, kat["mis_in"], 1)[:, 0], kat["mis"])
    # UniformSampleSphere fed by the generator the way random-walk-sss.h:296 is compiled by g++ (second draw first)
    n = len(kat["sphere_from_rng"])
    inp = np.zeros((n, 4), np.uint32)
    inp[:, 0], inp[:, 2] = np.arange(n, dtype=np.uint32), np.uint32(1234567890)
    d = A.leaf_eval(A.LEAF_RNG, inp.view(np.float32), 2)
    assert same(A.leaf_eval(A.LEAF_SPHERE, d[:, ::-1], 3), kat["sphere_from_rng"])


def test_ggx(pa, kat):
    A = pa.api
    wo, wi, al, u = kat["ggx_wo"], kat["ggx_wi"], kat["ggx_alpha"], kat["u2"]
    for distrib in (1, 2):
        db = as_bits(distrib, len(wo))[:, None]
        assert same(A.leaf_eval(A.LEAF_GGX_EVAL, np.hstack([wi, wo, al, db]), 2), kat[f"ggx_eval{distrib}"]), distrib
        assert same(A.leaf_eval(A.LEAF_GGX_SAMPLE, np.hstack([wo, al, u, db]), 5), kat[f"ggx_sample{distrib}"]), distrib


def test_hair_bsdf(pa, kat):
    A = pa.api
    p, wo, wi, us = kat["hair_params"][:, :23], kat["hair_wo"], kat["hair_wi"], kat["hair_us"]
    assert same(A.leaf_eval(A.LEAF_HAIR_EVAL, np.hstack([wi, wo, p]), 4), kat["hair_eval"])
    assert same(A.leaf_eval(A.LEAF_HAIR_SAMPLE, np.hstack([wo, p, us]), 7), kat["hair_sample"])


def test_leaf_eval_refuses_short_items(pa):
    A = pa.api
    with pytest.raises(Exception):
        A.leaf_eval(A.LEAF_GGX_EVAL, np.zeros((4, 5), np.float32), 2)
    with pytest.raises(Exception):
        A.leaf_eval(11, np.zeros((4, 5), np.float32), 2)


def test_texture_fetch(pa, kat):
    """Texture::FetchFloat3 (texture.cc:43-68, image-utils.cc:99-167: bilinear, clamp addressing, channels the image lacks read as 0)
    as the textured shading kernel runs it, on textures of 1 / 2 / 3 / 4 channels, against the reference's own code compiled unmodified"""
    import copy
    from pbrlab_amd import scenes
    desc = copy.copy(scenes.textured_cornell_scene(monkey_subdiv=1, lucy_nu=16, lucy_nv=6))
    first = len(desc.textures)
    desc.textures = list(desc.textures) + [np.ascontiguousarray(kat[f"tex{c}_pixels"]) for c in (1, 2, 3, 4)]
    sg = pa.scene_from_desc(desc)
    for k, c in enumerate((1, 2, 3, 4)):
        got = sg.texture_fetch(first + k, kat[f"tex{c}_uv"])
        assert same(got, kat[f"tex{c}_rgb"]), c
        if c < 3:
            assert not got[:, c:].any()
    with pytest.raises(Exception):
        sg.texture_fetch(first + 4, kat["tex1_uv"])
