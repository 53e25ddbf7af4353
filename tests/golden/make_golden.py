#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container, where
/root/reference exists; the fixtures then travel to the GPU box).

  ref_leaf_kats.npz   inputs + outputs of the REFERENCE's own leaf functions (RNG, fast-math, sampling,
                      Fresnel, MIS, Lambert, GGX, hair BSDF, tiles, Catmull-Rom->Bezier), produced by
                      oracle/_ref/libref_leaf.so = the reference headers compiled unmodified.
  oracle_images.npz   small renders + per-path traces from the ORACLE (the C restatement), both math
                      modes.  These pin the integrator-level behaviour of the oracle against drift; they
                      are not reference outputs (the reference integrator cannot be built here without
                      stand-ins for mpark/variant.hpp and Embree -- DESIGN.md §oracle).
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
from pbrlab_amd import scenes  # noqa: E402

P = O._ptr


def golden_scenes():
    return {
        "lambert": scenes.cornell_scene("lambert", monkey_subdiv=2, lucy_nu=64, lucy_nv=12),
        "ggx": scenes.cornell_scene("ggx", monkey_subdiv=2, lucy_nu=64, lucy_nv=12),
        "sss": scenes.cornell_scene("sss", monkey_subdiv=2, lucy_nu=64, lucy_nv=12),
        "hair": scenes.hair_scene(n_strands=500, n_segments=6, head_subdiv=2),
        "textured": scenes.textured_cornell_scene(monkey_subdiv=2, lucy_nu=64, lucy_nv=12),
    }


def scene_digest(desc):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(desc.vertices).tobytes())
    h.update(np.ascontiguousarray(desc.normals).tobytes())
    for s in desc.shapes:
        h.update(np.ascontiguousarray(s.vertex_ids).tobytes())
    for c in desc.curves:
        h.update(np.ascontiguousarray(c.vertices).tobytes())
    for t in desc.textures:
        h.update(np.ascontiguousarray(t).tobytes())
    if desc.texcoords is not None:
        h.update(np.ascontiguousarray(desc.texcoords).tobytes())
    return h.hexdigest()


def unit(rng, n, hemi=False):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
    if hemi:
        v[:, 2] = np.abs(v[:, 2])
    return np.ascontiguousarray(v, np.float32)


def hair_params(rng):
    beta_m, beta_n = rng.uniform(0.05, 1.0), rng.uniform(0.05, 1.0)
    v0 = (0.726 * beta_m + 0.812 * beta_m ** 2 + 3.7 * beta_m ** 20) ** 2
    s = np.sqrt(np.pi / 8) * (0.265 * beta_n + 1.194 * beta_n ** 2 + 5.372 * beta_n ** 22)
    p = [rng.uniform(-1, 1), v0, 0.25 * v0, 4 * v0, 4 * v0, s, *rng.uniform(0.05, 3, 3), rng.uniform(1.2, 1.8),
         np.radians(rng.uniform(0, 10)), *rng.uniform(0.3, 1, 9), 1, 1, 1, 1.0]
    return np.float32(p)


def make_ref_leaf():
    R = O.ref()
    rng = np.random.RandomState(2024)
    out = {}
    seeds = np.array([0, 1, 7, 12345, (5 << 32) + 99, 2 ** 63 + 17, 2 ** 64 - 1, 42], np.uint64)
    draws = np.zeros((len(seeds), 16), np.float32)
    for i, s in enumerate(seeds):
        R.ref_rng(int(s), 1234567890, 16, P(draws[i]))
    out["rng_seeds"], out["rng_draws"] = seeds, draws
    n = 256
    fm_x = {0: rng.uniform(-20, 20, n), 1: rng.uniform(-20, 20, n), 2: rng.uniform(-90, 90, n),
            3: np.exp(rng.uniform(-80, 80, n)), 5: rng.uniform(-1.2, 1.2, n), 6: rng.uniform(-130, 130, n),
            7: np.exp(rng.uniform(-80, 80, n))}
    for op, xs in fm_x.items():
        xs = np.float32(xs)
        out[f"fm{op}_x"] = xs
        out[f"fm{op}_y"] = np.float32([R.ref_fastmath(op, x, 0) for x in xs])
    yx = np.float32(rng.uniform(-3, 3, (n, 2)))
    out["fm4_yx"], out["fm4_r"] = yx, np.float32([R.ref_fastmath(4, y, x) for y, x in yx])
    u = np.float32(rng.rand(n, 2))
    out["u2"] = u
    lam, sph, tri, cosh = (np.zeros((n, k), np.float32) for k in (5, 3, 2, 3))
    for i in range(n):
        R.ref_lambert_sample(u[i, 0], u[i, 1], P(lam[i]))
        R.ref_uniform_sphere(u[i, 0], u[i, 1], P(sph[i]))
        R.ref_triangle_sampler(u[i, 0], u[i, 1], P(tri[i]))
        R.ref_cosine_hemisphere(u[i, 0], u[i, 1], P(cosh[i]))
    out["lambert"], out["sphere"], out["triangle"], out["cos_hemi"] = lam, sph, tri, cosh
    ce = np.float32(np.stack([rng.uniform(-1, 1, n), rng.uniform(0.0, 3, n)], 1))
    out["fresnel_in"], out["fresnel"] = ce, np.float32([R.ref_fresnel(c, e) for c, e in ce])
    ab = np.float32(np.exp(rng.uniform(-10, 10, (n, 2))))
    out["mis_in"], out["mis"] = ab, np.float32([R.ref_power_heuristic(a, b) for a, b in ab])
    wo, wi = unit(rng, n, hemi=True), unit(rng, n)
    alphas = np.float32(np.exp(rng.uniform(np.log(1e-4), 0.0, (n, 2))))
    alphas[::3, 1] = alphas[::3, 0]
    out["ggx_wo"], out["ggx_wi"], out["ggx_alpha"] = wo, wi, alphas
    for distrib in (1, 2):
        ev, sm = np.zeros((n, 2), np.float32), np.zeros((n, 5), np.float32)
        for i in range(n):
            R.ref_ggx_eval(P(wi[i]), P(wo[i]), alphas[i, 0], alphas[i, 1], distrib, P(ev[i]))
            R.ref_ggx_sample(P(wo[i]), alphas[i, 0], alphas[i, 1], u[i, 0], u[i, 1], distrib, P(sm[i]))
        out[f"ggx_eval{distrib}"], out[f"ggx_sample{distrib}"] = ev, sm
    hp = np.stack([hair_params(rng) for _ in range(n)])
    hwo, hwi, hus = unit(rng, n), unit(rng, n), np.float32(rng.rand(n, 4))
    hev, hsm = np.zeros((n, 4), np.float32), np.zeros((n, 7), np.float32)
    for i in range(n):
        R.ref_hair_eval(P(hwi[i]), P(hwo[i]), P(hp[i]), P(hev[i]))
        R.ref_hair_sample(P(hwo[i]), P(hp[i]), P(np.ascontiguousarray(hus[i])), P(hsm[i]))
    out.update(hair_params=hp, hair_wo=hwo, hair_wi=hwi, hair_us=hus, hair_eval=hev, hair_sample=hsm)
    # g++ argument order of UniformSampleSphere(rng.Draw(), rng.Draw())
    so = np.zeros((8, 3), np.float32)
    for i in range(8):
        R.ref_uniform_sphere_from_rng(i, 1234567890, P(so[i]))
    out["sphere_from_rng"] = so
    nt = C.c_uint32()
    tiles = np.zeros((510, 4), np.uint32)
    R.ref_create_tiles(1920, 1080, P(tiles, O.u32p), C.byref(nt))
    out["tiles_1920x1080"] = tiles[:nt.value]
    cvs, rad = np.float32(rng.normal(size=(9, 3))), np.float32(rng.uniform(0.001, 0.05, 9))
    bez = np.zeros(8 * 16, np.float32)
    R.ref_to_cubic_bezier(P(cvs), P(rad), 9, P(bez))
    out.update(bezier_cvs=cvs, bezier_radii=rad, bezier_out=bez.reshape(-1, 4))
    # Texture::FetchFloat3 (bilinear, clamp) on images with 1..4 channels, uv partly outside [0,1]
    for c in (1, 2, 3, 4):
        px = np.ascontiguousarray(rng.rand(5, 7, c).astype(np.float32))
        uvs = np.float32(np.concatenate([rng.uniform(-0.2, 1.2, (60, 2)), [[0, 0], [1, 1], [0.999999, 0.5], [1, 0]]]))
        res = np.zeros((len(uvs), 3), np.float32)
        for i, (u, v) in enumerate(uvs):
            R.ref_texture_fetch(P(px), 7, 5, c, u, v, P(res[i]))
        out[f"tex{c}_pixels"], out[f"tex{c}_uv"], out[f"tex{c}_rgb"] = px, uvs, res
    xs = np.float32(np.concatenate([np.linspace(0, 1.5, 200), [0.0031308, 0.04045, 1.0]]))
    out["srgb_x"] = xs
    out["srgb_oetf"] = np.float32([R.ref_linear_to_srgb(float(x)) for x in xs])
    out["srgb_eotf"] = np.float32([R.ref_srgb_to_linear(float(x)) for x in xs])
    np.savez_compressed(os.path.join(HERE, "ref_leaf_kats.npz"), **out)


def make_oracle_images():
    out = {}
    for name, desc in golden_scenes().items():
        so = O.oracle_scene_from_desc(desc)
        out[f"{name}_digest"] = np.frombuffer(bytes.fromhex(scene_digest(desc)), np.uint8)
        lo, hi = so.FetchSceneAABB()
        out[f"{name}_aabb"] = np.stack([lo, hi])
        for mode, tag in ((O.MATH_LIBM, "libm"), (O.MATH_F64R, "f64r")):
            rgba, cnt, st = so.render(64, 64, 4, threads=8, math_mode=mode)
            out[f"{name}_{tag}_rgba"] = rgba
            out[f"{name}_{tag}_rays"] = np.array([st["closest_rays"], st["shadow_rays"], st["rng_draws"]], np.uint64)
        # per-path traces (libm): pixel grid 8x8, pass 0..1
        recs = []
        for y in range(4, 64, 8):
            for x in range(4, 64, 8):
                for p in range(2):
                    rad, draws, nh, hits = so.sample_trace(64, 64, x, y, p, max_hits=8)
                    ids = np.full((8, 3), 0xFFFFFFFF, np.uint32)
                    for k in range(min(nh, 8)):
                        ids[k] = (hits[k]["instance_id"], hits[k]["geom_id"], hits[k]["prim_id"])
                    recs.append((x, y, p, draws, nh, rad.copy(), ids))
        out[f"{name}_trace_xyp"] = np.array([r[:3] for r in recs], np.uint32)
        out[f"{name}_trace_draws_hits"] = np.array([r[3:5] for r in recs], np.uint64)
        out[f"{name}_trace_radiance"] = np.stack([r[5] for r in recs])
        out[f"{name}_trace_ids"] = np.stack([r[6] for r in recs])
    np.savez_compressed(os.path.join(HERE, "oracle_images.npz"), **out)


if __name__ == "__main__":
    O.build_oracle()
    if O.have_ref():
        make_ref_leaf()
    else:
        print("oracle/_ref not built: ref_leaf_kats.npz not regenerated")
    make_oracle_images()
    print("fixtures written to", HERE)
