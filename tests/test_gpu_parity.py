"""GPU (-m gpu): parity of the HIP path (through the C ABI of libpbrhip.so) with the oracle.

Bars (BASELINE.json north_star): hit / tile indices bit-exact; radiance within 1e-4 relative L2.
Measured: against the oracle in the device's math mode (glibcf: glibc 2.35's x86-64 FMA float functions restated,
include/pbr_glibcf.h, = this image's libm, tests/test_glibcf.py) the images are bit-identical (0 pixels differ on every
case below), and so they are against the committed libm fixtures."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import _oracle as O  # noqa: E402
from golden.make_golden import golden_scenes  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_L2_TOL = 1e-4       # north_star tolerance for radiance
MAX_DIVERGENT = 1e-3    # fraction of pixels allowed to differ at all vs the oracle (measured: 0)


@pytest.fixture(scope="module")
def pa():
    import pbrlab_amd as pa
    if pa.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run on an MI355X (there is no CPU fallback)")
    pa.set_device(0)
    return pa


@pytest.fixture(scope="module")
def pairs(pa):
    out = {}
    for name, desc in golden_scenes().items():
        out[name] = (desc, pa.scene_from_desc(desc), O.oracle_scene_from_desc(desc))
    return out


def assert_hits_equal(a, b):
    for f in ("instance_id", "geom_id", "prim_id"):
        assert np.array_equal(a[f], b[f]), f
    for f in ("t", "u", "v", "normal_g"):
        assert np.array_equal(np.ascontiguousarray(a[f]).view(np.uint32), np.ascontiguousarray(b[f]).view(np.uint32)), f


def image_check(gpu_rgba, ref_rgba, exact_fraction=MAX_DIVERGENT):
    d = np.abs(gpu_rgba - ref_rgba).max(axis=2) > 0
    rel = np.linalg.norm(gpu_rgba[..., :3] - ref_rgba[..., :3]) / max(np.linalg.norm(ref_rgba[..., :3]), 1e-30)
    assert rel < REL_L2_TOL, rel
    assert d.mean() <= exact_fraction, (int(d.sum()), d.size)
    return int(d.sum()), rel


@pytest.mark.parametrize("name", ["lambert", "ggx", "sss", "hair", "textured"])
def test_trace_hooks_bit_exact(pa, pairs, name):
    from pbrlab_amd import scenes
    desc, sg, so = pairs[name]
    lo, hi = so.FetchSceneAABB()
    glo, ghi = sg.FetchSceneAABB()
    assert np.array_equal(lo, glo) and np.array_equal(hi, ghi)
    rays = scenes.random_rays((lo, hi), 50000, seed=21)
    assert_hits_equal(sg.trace_closest(rays), so.trace_closest(rays))
    assert_hits_equal(sg.trace_closest(rays[:1500]), so.trace_closest(rays[:1500], brute_force=True))
    for tmax in (0.05, 0.4, 1.5):
        sr = rays.copy()
        sr["tmax"] = tmax
        assert np.array_equal(sg.trace_any(sr), so.trace_any(sr))
        assert_hits_equal(sg.trace_closest(sr), so.trace_closest(sr))      # SSS-style bounded closest hit
    cam = so.camera_rays(64, 64, [(x, y) for y in range(64) for x in range(0, 64, 3)])
    assert_hits_equal(sg.trace_closest(cam), so.trace_closest(cam))


def test_trace_edge_cases(pa, pairs):
    desc, sg, so = pairs["lambert"]
    r = np.zeros(6, O.RAY_DT)
    r["org"] = [[0, 0, 0.9]] * 3 + [[0, 0, 3], [0, 0, 0], [0.25, 0.99, 0.25]]
    r["dir"] = [[0, 0, -1]] * 3 + [[0, 0, 1], [0, 1, 0], [1, 0, 0]]
    r["tmin"] = [0, 0, 1.9, 0, 0, 0]
    r["tmax"] = [1.844e18, 1.9, 1.844e18, 1.844e18, np.inf, 1.844e18]
    hg, ho = sg.trace_closest(r), so.trace_closest(r)
    assert_hits_equal(hg, ho)
    assert hg["instance_id"][1] == 2 and hg["instance_id"][2] == 0xFFFFFFFF and hg["instance_id"][3] == 0xFFFFFFFF
    assert hg["instance_id"][4] == 5                       # straight up: the light quad 1 cm under the ceiling
    assert len(sg.trace_closest(r[:0])) == 0 and len(sg.trace_any(r[:0])) == 0
    z = np.zeros(2, O.RAY_DT)                              # zero direction: no hit, no crash
    z["tmax"] = 1.0
    assert_hits_equal(sg.trace_closest(z), so.trace_closest(z))


# tail: 0xFFFFFFFF = every bounce goes through the wavefront kernels; 0 = default (these small frames switch to
# k_tail right after the first trace); 3000 = a switch in the middle of the render
@pytest.mark.parametrize("tail", [0xFFFFFFFF, 0, 3000])
@pytest.mark.parametrize("name", ["lambert", "ggx", "sss", "hair", "textured"])
def test_render_matches_oracle_and_fixture(pa, pairs, name, tail):
    desc, sg, so = pairs[name]
    layer = pa.RenderLayer()
    ok, st = pa.Render(sg, 64, 64, 4, layer=layer, flags=pa.api.RENDER_STATS, tail_paths=tail)
    assert (st["n_tail"] == 0) == (tail == 0xFFFFFFFF)
    assert ok is True and (layer.count == 4).all() and np.array_equal(layer.rgba[..., 3], np.full((64, 64), 4, np.float32))
    rgba, cnt, ost = so.render(64, 64, 4, threads=4, math_mode=O.MATH_DEVICE)
    ndiff, rel = image_check(layer.rgba, rgba)
    assert (st["closest_rays"] + st["tail_closest_rays"] + st["pruned_rays"], st["shadow_rays"] + st["tail_shadow_rays"]) == (ost["closest_rays"], ost["shadow_rays"])
    fx = np.load(os.path.join(G, "oracle_images.npz"))
    # the committed fixtures: the oracle with the reference's libm arithmetic (made on this image: glibc 2.35, whose float functions
    # the device restates, include/pbr_glibcf.h) -- bit for bit -- and with correctly rounded functions -- tolerance only
    image_check(layer.rgba, fx[f"{name}_libm_rgba"])
    rel_cr = np.linalg.norm(layer.rgba[..., :3] - fx[f"{name}_f64r_rgba"][..., :3]) / np.linalg.norm(fx[f"{name}_f64r_rgba"][..., :3])
    assert rel_cr < REL_L2_TOL
    print(f"{name}: {ndiff} px differ vs oracle[glibcf] and the libm fixture, rel L2 {rel:.1e}; vs the correctly rounded fixture rel L2 {rel_cr:.1e}")


def test_render_odd_size_and_larger(pa, pairs):
    desc, sg, so = pairs["sss"]
    for (w, h, spp) in [(130, 70, 3), (1, 1, 5), (65, 63, 2), (200, 150, 6)]:
        layer = pa.RenderLayer()
        pa.Render(sg, w, h, spp, layer=layer, tail_paths=0xFFFFFFFF if w == 130 else 0)
        rgba, cnt, _ = so.render(w, h, spp, threads=8, math_mode=O.MATH_DEVICE)
        assert np.array_equal(layer.count, cnt)
        image_check(layer.rgba, rgba)


@pytest.mark.parametrize("tile", ["1", "3", "8", "16", "100"])
def test_path_order_inside_a_block_does_not_matter(pa, pairs, tile, monkeypatch):
    """The paths of a pass are laid out block by block and, inside a block, in patches of PBRHIP_PIXEL_TILE x PBRHIP_PIXEL_TILE
    pixels (default 8: a wave's camera rays are an 8 x 8 patch; 1 or >= the block: rows).  Every value is a function of
    (pixel, pass) alone: the image is the oracle's whatever the order, at sizes that are no multiples of anything, alone and
    sharded (the exchange keeps the shard's own pixel order)."""
    desc, sg, so = pairs["ggx"]
    monkeypatch.setenv("PBRHIP_PIXEL_TILE", tile)
    for (w, h, spp) in [(131, 77, 3), (7, 5, 2)]:
        layer = pa.RenderLayer()
        pa.Render(sg, w, h, spp, layer=layer)
        rgba, cnt, _ = so.render(w, h, spp, threads=8, math_mode=O.MATH_DEVICE)
        assert np.array_equal(layer.count, cnt) and layer.rgba.tobytes() == rgba.tobytes(), (tile, w, h)
        acc, cacc = np.zeros_like(layer.rgba), np.zeros_like(layer.count)
        for r in range(3):
            part = pa.RenderLayer()
            pa.Render(sg, w, h, spp, layer=part, tile_rank=r, tile_world=3, shard_block=16)
            acc += part.rgba
            cacc += part.count
        assert acc.tobytes() == rgba.tobytes() and np.array_equal(cacc, cnt), (tile, w, h)


@pytest.mark.parametrize("turns", ["0", "1", "3", "24"])
@pytest.mark.parametrize("name", ["lambert", "ggx", "sss", "hair", "textured"])
def test_resumable_rays_are_exact(pa, pairs, name, turns, monkeypatch):
    """Round 6: a k_trace wave that has found the ray queue empty suspends the rays it still traces after PBRHIP_SUSP_TURNS more loop turns
    (closest-hit rays everywhere; shadow rays in scenes without media, whose paths are then HELD until the ray is delivered) and the
    next launch resumes them where they stopped (kernels.h::PathState::susp_turns).  With 1 or 3 turns nearly every ray of these small
    frames is suspended, many of them several times; a suspended ray loses nothing and its hit does not change, so the frame is the
    oracle's bit for bit and the ray counts are the reference's -- with the wavefront kernels alone and with a hand-over to k_tail in the
    middle, with the host one iteration ahead or three, shadow rays first or last in the launch."""
    desc, sg, so = pairs[name]
    monkeypatch.setenv("PBRHIP_SUSP_TURNS", turns)
    rgba, cnt, ost = so.render(96, 80, 5, threads=8, math_mode=O.MATH_DEVICE)
    seen = 0
    for tail, depth, shadow_first in [(0xFFFFFFFF, "2", "1"), (2500, "3", "0"), (0xFFFFFFFF, "1", "0")]:
        monkeypatch.setenv("PBRHIP_PIPE_DEPTH", depth)
        monkeypatch.setenv("PBRHIP_SHADOW_FIRST", shadow_first)
        layer = pa.RenderLayer()
        ok, st = pa.Render(sg, 96, 80, 5, layer=layer, flags=pa.api.RENDER_STATS, tail_paths=tail)
        assert np.array_equal(layer.count, cnt) and layer.rgba.tobytes() == rgba.tobytes(), (name, turns, tail, depth)
        assert (st["closest_rays"] + st["tail_closest_rays"] + st["pruned_rays"], st["shadow_rays"] + st["tail_shadow_rays"]) == (ost["closest_rays"], ost["shadow_rays"])
        seen += st["suspended_rays"]
    assert (seen == 0) == (turns == "0"), (name, turns, seen)
    if turns in ("1", "3"):
        assert seen > 1000, (name, turns, seen)     # (the case really exercises the path)


def test_patch_order_does_not_matter(pa, pairs, monkeypatch):
    """Round 6: the 8 x 8 pixel patches of a pass are laid out in a scattered order (PBRHIP_PATCH_SHUFFLE: a batch of k_trace's ray queue is
    eight image regions instead of one).  A permutation of the work: the image is the oracle's either way, alone and sharded.  (The
    curve leaves of the Q tree are 64-byte records of one or two arbitrary pieces since round 6 -- a build option, PB_CURVE_RECORDS --:
    every hair / curve test of this file runs on them, and test_wide_and_binary_trees_agree compares them with the binary tree.)"""
    for name in ("hair", "ggx"):
        desc, sg, so = pairs[name]
        rgba, cnt, _ = so.render(131, 77, 3, threads=8, math_mode=O.MATH_DEVICE)
        for shuffle in ("1", "0"):
            monkeypatch.setenv("PBRHIP_PATCH_SHUFFLE", shuffle)
            for tail, world in ((0xFFFFFFFF, 1), (0, 1), (0, 3)):
                acc, cacc = np.zeros_like(rgba), np.zeros_like(cnt)
                for r in range(world):
                    part = pa.RenderLayer()
                    pa.Render(sg, 131, 77, 3, layer=part, tail_paths=tail, tile_rank=r, tile_world=world, shard_block=16 if world > 1 else 0)
                    acc += part.rgba
                    cacc += part.count
                assert np.array_equal(cacc, cnt) and acc.tobytes() == rgba.tobytes(), (name, shuffle, tail, world)


def test_timing_flags_do_not_change_the_image(pa, pairs):
    """PBRHIP_RENDER_TIMING times every launch, PBRHIP_RENDER_TIMING_TRACE only the k_trace launches (bench.py's timed region): the same
    image either way, and the trace-only flag reports nothing but k_trace."""
    desc, sg, so = pairs["ggx"]
    ref = pa.RenderLayer()
    pa.Render(sg, 96, 64, 3, layer=ref, tail_paths=0xFFFFFFFF)
    for flag in (pa.api.RENDER_TIMING, pa.api.RENDER_TIMING_TRACE):
        layer = pa.RenderLayer()
        ok, st = pa.Render(sg, 96, 64, 3, layer=layer, flags=flag, tail_paths=0xFFFFFFFF)
        assert layer.rgba.tobytes() == ref.rgba.tobytes() and np.array_equal(layer.count, ref.count)
        assert st["ms_trace_closest"] > 0.0 and st["n_trace_closest"] > 0
        assert (st["ms_shade_principled"] > 0.0) == (flag == pa.api.RENDER_TIMING), (flag, st["ms_shade_principled"])


def test_shading_without_classify_is_exact(pa, pairs, monkeypatch):
    """Round 6: in a scene of principled surfaces only (no hair, no medium) k_classify is skipped on EVERY bounce -- k_shade_principled
    reads the trace queue itself and applies the drop rule (miss, known-to-fail roulette, no material) -- PBRHIP_DIRECT (default 1).
    With and without it the image is the oracle's; with suspended closest-hit and shadow rays (held paths) in the queue as well."""
    desc, sg, so = pairs["ggx"]
    rgba, cnt, _ = so.render(131, 77, 3, threads=8, math_mode=O.MATH_DEVICE)
    for direct in ("1", "0"):
        monkeypatch.setenv("PBRHIP_DIRECT", direct)
        for turns in ("24", "1"):
            monkeypatch.setenv("PBRHIP_SUSP_TURNS", turns)
            for tail in (0xFFFFFFFF, 0):
                layer = pa.RenderLayer()
                pa.Render(sg, 131, 77, 3, layer=layer, tail_paths=tail)
                assert np.array_equal(layer.count, cnt) and layer.rgba.tobytes() == rgba.tobytes(), (direct, turns, tail)


@pytest.mark.parametrize("run", ["1", "2", "4", "8", "64"])
@pytest.mark.parametrize("name", ["hair", "ggx"])
def test_pass_runs_do_not_matter(pa, pairs, name, run, monkeypatch):
    """PathState::pass_run (round 5): runs of R passes of one pixel adjacent in the path order (scenes with curves: the largest power of
    two <= 64 that divides a group's passes; PBRHIP_PASS_RUN forces R).  A permutation of the paths: the image is the oracle's for any
    R, with pass counts R does and does not divide, in one group, two groups, chunked, and resumed from a later pass."""
    desc, sg, so = pairs[name]
    monkeypatch.setenv("PBRHIP_PASS_RUN", run)
    for (w, h, spp, first) in [(61, 37, 8, 0), (33, 20, 12, 5), (16, 9, 7, 0)]:
        rgba, cnt, _ = so.render(w, h, spp, first_pass=first, threads=8, math_mode=O.MATH_DEVICE)
        for kw in ({}, {"num_streams": 2}, {"max_paths_in_flight": w * h * 4}):
            layer = pa.RenderLayer()
            pa.Render(sg, w, h, spp, layer=layer, first_pass=first, **kw)
            assert np.array_equal(layer.count, cnt) and layer.rgba.tobytes() == rgba.tobytes(), (run, w, h, spp, kw)


def test_chunking_and_progressive_are_exact(pa, pairs):
    """results must not depend on how passes are chunked (max_paths_in_flight) and a resumed render
    (first_pass + NO_CLEAR) equals the one-shot render bit for bit"""
    desc, sg, so = pairs["ggx"]
    a, b, c = pa.RenderLayer(), pa.RenderLayer(), pa.RenderLayer()
    pa.Render(sg, 96, 80, 6, layer=a)
    pa.Render(sg, 96, 80, 6, layer=b, max_paths_in_flight=96 * 80 * 2)
    assert a.rgba.tobytes() == b.rgba.tobytes()
    pa.Render(sg, 96, 80, 4, layer=c)
    pa.Render(sg, 96, 80, 2, layer=c, first_pass=4, flags=pa.api.RENDER_NO_CLEAR)
    assert a.rgba.tobytes() == c.rgba.tobytes() and (c.count == 6).all()
    pa.Render(sg, 96, 80, 6, layer=c)                       # Render() clears the layer (render.cc:99-100)
    assert a.rgba.tobytes() == c.rgba.tobytes()


@pytest.mark.parametrize("name", ["ggx", "sss", "hair"])
def test_concurrent_path_groups_are_exact(pa, pairs, name):
    """pbrhip_render_desc.num_streams: the passes of a chunk split into independent groups on their own HIP streams
    (their tails and drains overlap); path slots stay global, so the image is bit-identical for any group count, with
    and without the tail kernel, also when there are fewer passes than groups"""
    desc, sg, so = pairs[name]
    ref = pa.RenderLayer()
    pa.Render(sg, 96, 80, 5, layer=ref, num_streams=1)
    for streams, tail in ((2, 0), (3, 0), (8, 0), (2, 0xFFFFFFFF), (3, 64)):
        lay = pa.RenderLayer()
        ok, st = pa.Render(sg, 96, 80, 5, layer=lay, num_streams=streams, tail_paths=tail)
        assert ok is True and (lay.count == 5).all()
        assert lay.rgba.tobytes() == ref.rgba.tobytes(), (streams, tail)
    lay = pa.RenderLayer()
    pa.Render(sg, 96, 80, 5, layer=lay, num_streams=2, max_paths_in_flight=96 * 80 * 3)   # chunks of 3 + 2 passes
    assert lay.rgba.tobytes() == ref.rgba.tobytes()


def test_tile_sharding_matches_single(pa, pairs):
    desc, sg, so = pairs["sss"]
    full = pa.RenderLayer()
    pa.Render(sg, 200, 136, 3, layer=full)
    acc, cacc = np.zeros_like(full.rgba), np.zeros_like(full.count)
    tiles = pa.create_tiles(200, 136)
    for r in range(3):
        part = pa.RenderLayer()
        pa.Render(sg, 200, 136, 3, layer=part, tile_rank=r, tile_world=3)
        mask = np.zeros((136, 200), bool)
        for sx, tx, sy, ty in tiles[r::3]:
            mask[sy:ty, sx:tx] = True
        assert (part.count[mask] == 3).all() and not part.count[~mask].any() and not part.rgba[~mask].any()
        acc += part.rgba
        cacc += part.count
    assert acc.tobytes() == full.rgba.tobytes() and np.array_equal(cacc, full.count)


@pytest.mark.parametrize("block", [8, 16, 64, 100])
def test_shard_block_sizes(pa, pairs, block):
    """pbrhip_render_desc.shard_block: the pixel blocks dealt to ranks can be smaller (or larger) than the reference's 64 x 64
    tile; every rank's layer is disjoint from the others' and their sum is the one-rank frame, bit for bit"""
    desc, sg, so = pairs["ggx"]
    full = pa.RenderLayer()
    pa.Render(sg, 200, 136, 3, layer=full)
    for world in (2, 3, 8):
        acc = np.zeros_like(full.rgba)
        cnt = np.zeros_like(full.count)
        for r in range(world):
            p = pa.RenderLayer()
            pa.Render(sg, 200, 136, 3, layer=p, tile_rank=r, tile_world=world, shard_block=block)
            assert not (cnt.astype(bool) & p.count.astype(bool)).any()
            acc += p.rgba
            cnt += p.count
        assert acc.tobytes() == full.rgba.tobytes() and (cnt == 3).all(), (block, world)


def test_material_update_and_errors(pa, pairs):
    from pbrlab_amd import scenes
    desc = scenes.cornell_scene("lambert", monkey_subdiv=1, lucy_nu=16, lucy_nv=6)
    sg = pa.scene_from_desc(desc)
    a, b = pa.RenderLayer(), pa.RenderLayer()
    pa.Render(sg, 48, 48, 2, layer=a)
    m = dict(desc.materials[6])           # Wall_Red -> blue
    m["base_color"] = (0.05, 0.05, 0.6)
    sg.UpdateMaterialParam(6, pa.make_principled(m))
    pa.Render(sg, 48, 48, 2, layer=b)
    assert b.rgba[24, 2, 2] > b.rgba[24, 2, 0] and a.rgba[24, 2, 0] > a.rgba[24, 2, 2]
    d2 = scenes.cornell_scene("lambert", monkey_subdiv=1, lucy_nu=16, lucy_nv=6)
    d2.materials[6]["base_color"] = (0.05, 0.05, 0.6)
    s2 = O.oracle_scene_from_desc(d2)
    rgba, _, _ = s2.render(48, 48, 2, math_mode=O.MATH_DEVICE)
    image_check(b.rgba, rgba)
    with pytest.raises(pa.PbrHipError) as e:
        sg.AttachMaterialParamIdsToInstance(0, [np.zeros(5, np.uint32)])
    assert e.value.code == -2                       # "material param error" (scene.cc:86-88)
    with pytest.raises(pa.PbrHipError):
        sg.CreateInstance(0, np.diag([0.0, 1, 1, 1]).astype(np.float32))   # singular (any invertible transform is accepted)
    s3 = pa.Scene()
    with pytest.raises(pa.PbrHipError):
        pa.Render(s3, 8, 8, 1, layer=pa.RenderLayer())   # not committed
    s3.CommitScene()                                      # empty scene renders black
    lay = pa.RenderLayer()
    pa.Render(s3, 8, 8, 2, layer=lay)
    assert not lay.rgba[..., :3].any() and (lay.count == 2).all()


def test_material_update_switches_on_subsurface(pa):
    """A scene committed without any subsurface material runs the PLAIN shading kernel (medium entry compiled out); a material
    edit that switches subsurface scattering on must take the general kernel from then on: the image equals the oracle's for
    the edited scene, and differs from the image before the edit."""
    from pbrlab_amd import scenes
    desc = scenes.cornell_scene("ggx", monkey_subdiv=1, lucy_nu=24, lucy_nv=8)
    assert not any(m.get("subsurface", 0.0) > 0 for m in desc.materials if m["kind"] == "principled")
    sg = pa.scene_from_desc(desc)
    a, b = pa.RenderLayer(), pa.RenderLayer()
    pa.Render(sg, 64, 48, 3, layer=a)
    idx = next(i for i, m in enumerate(desc.materials) if m["name"] == "Lucy")
    m = dict(desc.materials[idx], subsurface=1.0, subsurface_radius=(1.0, 0.2, 0.1), subsurface_color=(1.0, 0.8, 0.8))
    sg.UpdateMaterialParam(idx, pa.make_principled(m))
    pa.Render(sg, 64, 48, 3, layer=b)
    assert a.rgba.tobytes() != b.rgba.tobytes()
    d2 = scenes.cornell_scene("ggx", monkey_subdiv=1, lucy_nu=24, lucy_nv=8)
    d2.materials[idx].update(subsurface=1.0, subsurface_radius=(1.0, 0.2, 0.1), subsurface_color=(1.0, 0.8, 0.8))
    rgba, cnt, _ = O.oracle_scene_from_desc(d2).render(64, 48, 3, threads=4, math_mode=O.MATH_DEVICE)
    assert b.rgba.tobytes() == rgba.tobytes()


def test_texture_errors(pa):
    from pbrlab_amd import scenes
    s = pa.Scene()
    s.AddTexture(np.ones((2, 2, 3), np.float32))
    s.AddMaterialParam(pa.make_principled(dict(scenes.PRINCIPLED_DEFAULTS, base_color_tex_id=7)))   # no such texture
    with pytest.raises(pa.PbrHipError) as e:
        s.CommitScene()                                             # ids are checked at commit
    assert e.value.code == -1
    s = pa.Scene()
    with pytest.raises(pa.PbrHipError):
        s.AddTexture(np.zeros((4, 4, 5), np.float32))               # more than 4 channels


def test_cancel_and_finish_pass(pa, pairs):
    desc, sg, so = pairs["lambert"]
    fin, cancel, lay = C.c_size_t(99), C.c_ubyte(1), pa.RenderLayer()
    pa.Render(sg, 32, 32, 4, cancel_render_flag=cancel, layer=lay, finish_pass=fin)
    assert fin.value == 0 and not lay.count.any()
    cancel.value = 0
    pa.Render(sg, 32, 32, 4, cancel_render_flag=cancel, layer=lay, finish_pass=fin)
    assert fin.value == 4 and (lay.count == 4).all()


def test_cancel_from_another_thread_mid_render(pa):
    """render.cc:217: the flag is set by another thread while Render() runs.  The library reads it at every host round
    trip, drains what is in flight and returns the passes that are complete: count is uniform and equals finish_pass,
    and those passes are the very ones an uncancelled render of that many passes produces."""
    import threading
    import time
    from pbrlab_amd import scenes
    desc = scenes.cornell_scene("sss", monkey_subdiv=2, lucy_nu=64, lucy_nv=12)
    sg = pa.scene_from_desc(desc)
    W, H, SPP = 640, 360, 4096            # ~1 G samples of the random-walk scene: seconds of work
    warm = pa.RenderLayer()
    pa.Render(sg, W, H, 8, layer=warm)    # allocate the working set outside the timed part
    fin, cancel, lay = C.c_size_t(0), C.c_ubyte(0), pa.RenderLayer()
    seen, t_set = [], [0.0]

    def killer():
        time.sleep(0.25)
        seen.append(fin.value)            # progress is visible while the call runs
        t_set[0] = time.perf_counter()
        cancel.value = 1
    th = threading.Thread(target=killer)
    th.start()
    ok, st = pa.Render(sg, W, H, SPP, cancel_render_flag=cancel, layer=lay, finish_pass=fin, max_paths_in_flight=W * H * 16)
    t_ret = time.perf_counter()
    th.join()
    assert ok is True and 0 < fin.value < SPP and st["passes_done"] == fin.value
    assert seen[0] <= fin.value
    assert (lay.count == fin.value).all() and np.array_equal(lay.rgba[..., 3], np.full((H, W), fin.value, np.float32))
    assert t_ret - t_set[0] < 0.25, t_ret - t_set[0]        # a wavefront iteration, not the rest of the frame
    ref = pa.RenderLayer()
    pa.Render(sg, W, H, int(fin.value), layer=ref)                      # (one chunk: the image does not depend on the chunking)
    assert ref.rgba.tobytes() == lay.rgba.tobytes()
    print(f"cancelled after {fin.value} of {SPP} passes, returned {1e3 * (t_ret - t_set[0]):.1f} ms after the flag")


@pytest.mark.parametrize("name", ["ggx", "sss", "hair"])
def test_group_schedules_are_exact(pa, pairs, name, monkeypatch):
    """pbrhip.cpp::plan_groups: however the passes of a chunk are split into path groups and however many of them are in
    their bulk phase at once, the image is the same, bit for bit"""
    desc, sg, so = pairs[name]
    ref = pa.RenderLayer()
    pa.Render(sg, 96, 80, 11, layer=ref, num_streams=1)
    for plan, window, minp in (("6,3,1,1", "2", None), ("1,1,1,1,1,1,1,1,1,1,1", "3", None), ("5,6", "1", None),
                               ("8", "2", None), (None, "2", "4096"), (None, "1", "1")):
        if plan is None:
            monkeypatch.delenv("PBRHIP_GROUPS", raising=False)
        else:
            monkeypatch.setenv("PBRHIP_GROUPS", plan)
        monkeypatch.setenv("PBRHIP_WINDOW", window)
        if minp is not None:
            monkeypatch.setenv("PBRHIP_GROUP_MIN_PATHS", minp)
        for tail in (0, 0xFFFFFFFF, 500):
            lay = pa.RenderLayer()
            fin = C.c_size_t(0)
            pa.Render(sg, 96, 80, 11, layer=lay, tail_paths=tail, finish_pass=fin)
            assert fin.value == 11 and lay.rgba.tobytes() == ref.rgba.tobytes(), (plan, window, minp, tail)


@pytest.mark.parametrize("name", ["sss", "hair"])
def test_render_multi_matches_single(pa, pairs, name):
    """pbrhip_render_multi: ranks of one process, rank g on device g % ndev, shards gathered device-to-device inside the
    library; equals the one-rank frame bit for bit (src/render.cc:203-238: the image does not depend on the workers)"""
    desc, sg, so = pairs[name]
    full = pa.RenderLayer()
    pa.Render(sg, 200, 136, 3, layer=full)
    ndev = pa.device_count()
    for n in (2, 3, 8):
        reps = [sg] + [pa.replicate(sg, g % ndev) for g in range(1, n)]
        lay = pa.RenderLayer()
        fin = C.c_size_t(0)
        ok, sts = pa.RenderMulti(reps, 200, 136, 3, layer=lay, finish_pass=fin)
        assert ok is True and fin.value == 3 and len(sts) == n
        assert lay.rgba.tobytes() == full.rgba.tobytes() and np.array_equal(lay.count, full.count), n
        assert sum(s["samples"] for s in sts) == 200 * 136 * 3
        lay2 = pa.RenderLayer()
        pa.RenderMulti(reps, 200, 136, 3, layer=lay2, shard_block=64)      # the reference's tiles as the dealing unit
        assert lay2.rgba.tobytes() == full.rgba.tobytes()
        for r in reps[1:]:
            r.close()
    # a replica is a full scene: it renders and traces on its own
    rep = pa.replicate(sg, 0)
    lay = pa.RenderLayer()
    pa.Render(rep, 200, 136, 3, layer=lay)
    assert lay.rgba.tobytes() == full.rgba.tobytes()
    assert np.array_equal(rep.FetchSceneAABB()[0], sg.FetchSceneAABB()[0])


def test_library_communicator_world1(pa, pairs):
    """pbrhip_comm_* (RCCL inside the library): with one rank the reduce is the identity and the gather a no-op; this
    checks that librccl loads, a communicator initialises and a collective runs on this box (the multi-rank exchange is
    covered by the in-process path above and by the gloo tests of the same pixel lists)"""
    import torch
    desc, sg, so = pairs["ggx"]
    W, H = 96, 80
    rgba = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    cnt = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    pa.Render(sg, W, H, 2, device_out=(rgba.data_ptr(), cnt.data_ptr()))
    torch.cuda.synchronize()
    before = rgba.clone()
    comm = pa.Comm(pa.Comm.unique_id(), 0, 1)
    comm.reduce_layer(rgba.data_ptr(), cnt.data_ptr(), W * H, root=0)
    comm.gather_layer(sg, W, H, rgba.data_ptr(), cnt.data_ptr(), shard_block=16, root=0)
    torch.cuda.synchronize()
    assert torch.equal(rgba, before) and int(cnt.min()) == 2
    comm.close()


@pytest.mark.parametrize("config", ["c2", "c3", "c4"])
def test_full_size_properties(pa, config):
    """BASELINE configs C2 (GGX), C3 (random-walk SSS), C4 (hair + head) at full geometry and resolution, low spp:
    size-independent properties + spot parity with the oracle."""
    from pbrlab_amd import scenes
    desc = {"c2": lambda: scenes.cornell_scene("ggx", seed=1), "c3": lambda: scenes.cornell_scene("sss", seed=1),
            "c4": lambda: scenes.hair_scene(seed=1)}[config]()
    sg = pa.scene_from_desc(desc)
    W, H, SPP = 1920, 1080, 2
    a, b = pa.RenderLayer(), pa.RenderLayer()
    pa.Render(sg, W, H, SPP, layer=a)
    pa.Render(sg, W, H, SPP, layer=b, max_paths_in_flight=W * H)
    assert (a.count == SPP).all() and np.isfinite(a.rgba).all() and (a.rgba[..., :3] >= 0).all()
    assert a.rgba.tobytes() == b.rgba.tobytes()                      # deterministic, chunk-independent
    acc = np.zeros_like(a.rgba)
    for r in range(2):
        p = pa.RenderLayer()
        pa.Render(sg, W, H, SPP, layer=p, tile_rank=r, tile_world=2)
        acc += p.rgba
    assert acc.tobytes() == a.rgba.tobytes()                         # sharded sum == whole
    c = pa.RenderLayer()
    pa.Render(sg, W, H, SPP, layer=c, tail_paths=0xFFFFFFFF)          # every bounce through the wavefront kernels
    assert c.rgba.tobytes() == a.rgba.tobytes()
    pa.Render(sg, W, H, SPP, layer=c, num_streams=2)                  # two concurrent path groups (the default for big chunks)
    assert c.rgba.tobytes() == a.rgba.tobytes()
    # spot parity with the oracle at full size: 3 tiles' worth of pixels through per-sample traces
    so = O.oracle_scene_from_desc(desc)
    rng = np.random.RandomState(0)
    for _ in range(60):
        x, y = int(rng.randint(W)), int(rng.randint(H))
        tot = np.zeros(3, np.float32)
        for p in range(SPP):
            rad, _, _, _ = so.sample_trace(W, H, x, y, p, math_mode=O.MATH_DEVICE)
            tot = tot + rad
        assert np.array_equal(tot.view(np.uint32), a.rgba[y, x, :3].view(np.uint32)), (x, y)


@pytest.mark.parametrize("config", ["c2", "c3", "c4"])
def test_whole_frames_on_the_benchmark_scenes(pa, config, monkeypatch):
    """The full-geometry benchmark scenes (545 k triangles; Lucy with random-walk SSS; 4.8 M hair pieces) rendered whole at a
    quarter of the resolution, 4 spp: every pixel of the GPU frame equals the oracle's, bit for bit (relative L2 = 0)."""
    from pbrlab_amd import scenes
    desc = {"c2": lambda: scenes.cornell_scene("ggx", seed=1), "c3": lambda: scenes.cornell_scene("sss", seed=1),
            "c4": lambda: scenes.hair_scene(seed=1)}[config]()
    sg = pa.scene_from_desc(desc)
    so = O.oracle_scene_from_desc(desc)
    W, H, SPP = 480, 270, 4
    a = pa.RenderLayer()
    pa.Render(sg, W, H, SPP, layer=a)
    rgba, cnt, _ = so.render(W, H, SPP, threads=O.oracle_threads(), math_mode=O.MATH_DEVICE)
    assert np.array_equal(a.count, cnt)
    ndiff, rel = image_check(a.rgba, rgba)
    assert ndiff == 0 and rel == 0.0, (ndiff, rel)
    assert a.rgba[..., :3].max() > 0
    # host-built trees are walked through the Q tree (quantised 4-wide nodes, curve pieces as chains of points) by default:
    # the binary tree must give the same frame
    monkeypatch.setenv("PBRHIP_WIDE", "0")
    b = pa.RenderLayer()
    _, st = pa.Render(sg, W, H, SPP, layer=b, flags=pa.api.RENDER_STATS)
    assert st["node_bytes"] == 64 and st["curve_bytes"] == 64
    nodes_binary = st["closest_nodes"] / max(st["closest_rays"], 1)
    monkeypatch.delenv("PBRHIP_WIDE")
    _, st = pa.Render(sg, W, H, SPP, layer=a, flags=pa.api.RENDER_STATS)
    assert st["node_bytes"] == 64 and st["curve_bytes"] == 32
    assert st["closest_nodes"] / max(st["closest_rays"], 1) < 0.62 * nodes_binary       # about half the node visits per ray
    assert np.array_equal(a.rgba.view(np.uint32), b.rgba.view(np.uint32)) and np.array_equal(a.count, b.count)


def test_headline_configuration_spot_parity(pa):
    """The frame bench.py times -- C2 at 1920 x 1080 x 64 spp with the library's defaults (one chunk, two path groups, tail kernel,
    doomed-path pruning) -- against the oracle: 48 random pixels, every one of their 64 samples traced by the oracle and summed
    in pass order, bit for bit; plus the frame-level invariants the bench asserts."""
    from pbrlab_amd import scenes
    desc = scenes.cornell_scene("ggx", seed=1)
    sg = pa.scene_from_desc(desc)
    W, H, SPP = 1920, 1080, 64
    a = pa.RenderLayer()
    ok, st = pa.Render(sg, W, H, SPP, layer=a)
    assert ok is True and (a.count == SPP).all() and np.isfinite(a.rgba).all() and np.array_equal(a.rgba[..., 3], np.full((H, W), SPP, np.float32))
    so = O.oracle_scene_from_desc(desc)
    rng = np.random.RandomState(7)
    for _ in range(48):
        x, y = int(rng.randint(W)), int(rng.randint(H))
        tot = np.zeros(3, np.float32)
        for p in range(SPP):
            rad, _, _, _ = so.sample_trace(W, H, x, y, p, math_mode=O.MATH_DEVICE)
            tot = tot + rad
        assert np.array_equal(tot.view(np.uint32), a.rgba[y, x, :3].view(np.uint32)), (x, y)


@pytest.mark.parametrize("name", ["lambert", "sss", "hair", "textured"])
def test_gpu_built_bvh_gives_identical_results(pa, pairs, name):
    """Row N3: the acceleration structure built on the GPU (Morton-order linear BVH).  Hits, and therefore images, do not
    depend on the tree: everything must equal the oracle (and the host-SAH scene) bit for bit."""
    from pbrlab_amd import scenes
    desc, sg, so = pairs[name]
    s2 = pa.scene_from_desc(desc, bvh_builder=pa.api.BVH_GPU_LBVH)
    i1, i2 = sg.info(), s2.info()
    assert i2["num_slots"] == i1["num_slots"] and i2["num_nodes"] == max(i2["num_slots"] - 1, 1)
    assert 1 <= i2["depth"] <= 40
    lo, hi = so.FetchSceneAABB()
    glo, ghi = s2.FetchSceneAABB()
    assert np.array_equal(np.asarray(lo, np.float32), glo) and np.array_equal(np.asarray(hi, np.float32), ghi)
    rays = scenes.random_rays((lo, hi), 20000, seed=5)
    assert_hits_equal(s2.trace_closest(rays), so.trace_closest(rays))
    assert np.array_equal(s2.trace_any(rays), so.trace_any(rays))
    for tail in (0, 0xFFFFFFFF):
        layer = pa.RenderLayer()
        pa.Render(s2, 64, 64, 4, layer=layer, tail_paths=tail)
        rgba, cnt, _ = so.render(64, 64, 4, threads=4, math_mode=O.MATH_DEVICE)
        assert np.array_equal(layer.count, cnt)
        assert np.array_equal(layer.rgba.view(np.uint32), rgba.view(np.uint32))


@pytest.mark.parametrize("name", ["lambert", "ggx", "sss", "textured", "hair"])
def test_wide_and_binary_trees_agree(pa, pairs, name, monkeypatch):
    """Host-built trees are traversed through the Q tree (k_trace, k_sss_walk, k_tail, the trace hooks: quantised 4-wide nodes,
    compact triangle slots, curve pieces as chains of points); PBRHIP_WIDE=0 selects the binary tree at every launch.  Hits
    and images must not depend on the choice (and both equal the oracle)."""
    from pbrlab_amd import scenes
    desc, sg, so = pairs[name]
    lo, hi = so.FetchSceneAABB()
    rays = scenes.random_rays((lo, hi), 30000, seed=11)
    want_hits, want_any = so.trace_closest(rays), so.trace_any(rays)
    rgba, cnt, _ = so.render(96, 64, 4, threads=4, math_mode=O.MATH_DEVICE)
    for wide in ("1", "0"):
        monkeypatch.setenv("PBRHIP_WIDE", wide)
        for simple in (False, True):
            if simple:
                monkeypatch.setenv("PBRHIP_SIMPLE_TRAVERSAL", "1")
            else:
                monkeypatch.delenv("PBRHIP_SIMPLE_TRAVERSAL", raising=False)
            assert_hits_equal(sg.trace_closest(rays), want_hits)
            assert np.array_equal(sg.trace_any(rays), want_any)
        monkeypatch.delenv("PBRHIP_SIMPLE_TRAVERSAL", raising=False)
        for tail in (0, 0xFFFFFFFF, 2000):
            layer = pa.RenderLayer()
            pa.Render(sg, 96, 64, 4, layer=layer, tail_paths=tail)
            assert np.array_equal(layer.count, cnt)
            assert np.array_equal(layer.rgba.view(np.uint32), rgba.view(np.uint32)), (wide, tail)


def test_gpu_bvh_builder_errors_and_tiny_scenes(pa):
    s = pa.Scene()
    with pytest.raises(pa.PbrHipError):
        s.SetBvhBuilder(7)
    # one triangle, two triangles: degenerate hierarchies
    for ntri in (1, 2, 3):
        s = pa.Scene()
        s.SetBvhBuilder(pa.api.BVH_GPU_LBVH)
        v = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [0, 1, 0, 1], [1, 1, 0.5, 1], [2, 0, 1, 1]], np.float32)
        f = np.array([[0, 1, 2], [1, 3, 2], [1, 4, 3]], np.uint32)[:ntri]
        m = s.AddMaterialParam(pa.make_principled(dict(base_color=(0.8, 0.8, 0.8), subsurface=0.0, subsurface_radius=(1, 1, 1), subsurface_color=(0.7, 0.1, 0.1), metallic=0.0, specular=0.5, specular_tint=0.0, roughness=0.5, anisotropic=0.0, anisotropic_rotation=0.0, sheen=0.0, sheen_tint=0.5, clearcoat=0.0, clearcoat_roughness=0.03, ior=1.45, transmission=0.0, transmission_roughness=0.0, base_color_tex_id=0xFFFFFFFF, subsurface_color_tex_id=0xFFFFFFFF)))
        mesh = s.AddTriangleMesh(v, None, None, f, None, None, np.full(ntri, m, np.uint32))
        ls = s.CreateLocalScene()
        s.AddMeshToLocalScene(ls, mesh)
        s.CreateInstance(ls, None)
        s.CommitScene()
        rays = np.zeros(ntri, pa.api.RAY_DT)
        for k in range(ntri):
            c = v[f[k], :3].mean(axis=0)
            rays[k]["org"] = c + np.array([0, 0, 5], np.float32)
            rays[k]["dir"] = (0, 0, -1)
            rays[k]["tmin"], rays[k]["tmax"] = 0.0, 100.0
        h = s.trace_closest(rays)
        assert list(h["prim_id"]) == list(range(ntri))
        with pytest.raises(pa.PbrHipError):
            s.SetBvhBuilder(pa.api.BVH_HOST_SAH)     # after commit


def _random_material_scene(seed, with_hair):
    """the small Cornell scene with EVERY parameter of every material drawn at random (closure branches the demo .mtl never
    reaches: clearcoat, anisotropy + rotation, metallic, tints, sheen, partial subsurface, odd ior / transmission values)"""
    from pbrlab_amd import scenes
    rng = np.random.RandomState(seed)
    desc = scenes.cornell_hair_scene("sss", n_strands=200, n_segments=5, monkey_subdiv=2, lucy_nu=64, lucy_nv=12) if with_hair \
        else scenes.cornell_scene("sss", monkey_subdiv=2, lucy_nu=64, lucy_nv=12)
    for i, m in enumerate(desc.materials):
        if m.get("kind", "principled") != "principled" or m["name"] == "Light":
            continue
        m = dict(m)
        u = lambda: float(rng.rand())                                     # noqa: E731
        pick = lambda p: u() if rng.rand() < p else 0.0                   # noqa: E731
        m.update(base_color=(u(), u(), u()), subsurface=pick(0.35), subsurface_radius=(0.05 + u(), 0.05 + u(), 0.05 + u()),
                 subsurface_color=(u(), u(), u()), metallic=pick(0.5), specular=pick(0.7), specular_tint=pick(0.5),
                 roughness=float(rng.choice([0.0, 0.01, 0.2, 0.5, 1.0, u()])), anisotropic=pick(0.5), anisotropic_rotation=pick(0.5),
                 sheen=pick(0.5), sheen_tint=u(), clearcoat=pick(0.6), clearcoat_roughness=float(rng.choice([0.0, 0.03, 0.3, u()])),
                 ior=float(rng.choice([1.0, 1.45, 2.5, 1.0 + u()])), transmission=pick(0.3), transmission_roughness=u())
        desc.materials[i] = m
    for c in desc.curves:
        h = dict(c.material)
        h.update(coloring_hair=int(rng.randint(2)), base_color=(float(rng.rand()), float(rng.rand()), float(rng.rand())),
                 melanin=float(rng.rand()), melanin_redness=float(rng.rand()), melanin_randomize=float(rng.rand()),
                 roughness=float(0.05 + 0.9 * rng.rand()), azimuthal_roughness=float(0.05 + 0.9 * rng.rand()),
                 ior=float(1.2 + rng.rand()), shift=float(rng.rand() * 5))
        c.material = h
    return desc


@pytest.mark.parametrize("seed", range(8))
def test_random_materials_parity(pa, seed):
    desc = _random_material_scene(100 + seed, with_hair=(seed % 2 == 1))
    sg, so = pa.scene_from_desc(desc), O.oracle_scene_from_desc(desc)
    rgba, cnt, _ = so.render(64, 48, 6, threads=4, math_mode=O.MATH_DEVICE)
    assert np.isfinite(rgba).all() and rgba[..., :3].max() > 0
    for tail in (0, 0xFFFFFFFF):
        layer = pa.RenderLayer()
        pa.Render(sg, 64, 48, 6, layer=layer, tail_paths=tail)
        assert np.array_equal(layer.count, cnt)
        nd = int((layer.rgba.view(np.uint32) != rgba.view(np.uint32)).any(axis=2).sum())
        assert nd == 0, (seed, tail, nd)


@pytest.mark.parametrize("extra_slivers", [0, 30])
@pytest.mark.parametrize("seed", range(6))
def test_triangle_soup_and_ties(pa, seed, extra_slivers):
    """Random triangle soups with exact duplicates, coplanar overlaps, slivers, degenerate (zero-area) triangles and shared
    edges (tests/_soups.py): equal hit distances must resolve to the smaller canonical id and an accepted hit must be a
    property of (ray, primitive) alone, whatever the tree looks like -- oracle BVH, oracle brute force (EVERY ray, incl. ray
    6858 of seed 1 on which a brute-force loop and the near-first traversals disagreed until round 3), GPU SAH tree, its
    4-wide collapse, the binary tree walked instead (PBRHIP_WIDE=0) and the GPU LBVH tree must agree bit for bit."""
    import _soups
    desc, so, rays = _soups.triangle_soup(seed, extra_slivers)
    hb = so.trace_closest(rays, brute_force=True)
    assert_hits_equal(so.trace_closest(rays), hb)
    assert (hb["instance_id"] != 0xFFFFFFFF).sum() > 1500
    short = rays.copy()
    short["tmax"] = np.where(np.isfinite(hb["t"]) & (hb["instance_id"] != 0xFFFFFFFF), hb["t"], 1.0)    # tmax == hit distance: accepted (t <= tmax)
    ob = so.trace_any(short, brute_force=True)
    assert np.array_equal(so.trace_any(short), ob)
    for builder in (pa.api.BVH_HOST_SAH, pa.api.BVH_GPU_LBVH):
        sg = pa.scene_from_desc(desc, bvh_builder=builder)
        for wide in ("1", "0") if builder == pa.api.BVH_HOST_SAH else ("1",):
            os.environ["PBRHIP_WIDE"] = wide
            try:
                assert_hits_equal(sg.trace_closest(rays), hb)
                assert np.array_equal(sg.trace_any(short), ob)
                os.environ["PBRHIP_SIMPLE_TRAVERSAL"] = "1"      # the one-ray-per-lane traversal (second implementation)
                assert_hits_equal(sg.trace_closest(rays), hb)
                assert np.array_equal(sg.trace_any(short), ob)
            finally:
                os.environ.pop("PBRHIP_WIDE", None)
                os.environ.pop("PBRHIP_SIMPLE_TRAVERSAL", None)


@pytest.mark.parametrize("which", ["PBRHIP_QUAD"])
def test_alternative_traversals_bit_exact(pa, pairs, which):
    """Alternative traversals behind environment switches (read per launch): hits do not depend on the visiting order, so soups (ties,
    slivers, tmax == hit distance) and whole renders are bit-identical.
    PBRHIP_QUAD: one ray per quad of lanes (dtrace_quad.h) -- k_tail runs it by default once a wave has at most eight paths left
    (the tail_paths=3000 renders below and every small render of this suite); here the hooks and every k_trace launch
    (PBRHIP_QUAD_RAYS) run it as well.  (Round 4's two-rays-per-lane and wave-pooled traversals were removed in round 5, round 5's
    8-wide O tree -- measured equal on triangles, 11 % slower on hair: profiles/README.md -- in round 6.)"""
    import _soups
    desc, so, rays = _soups.triangle_soup(2, 30)
    hb = so.trace_closest(rays, brute_force=True)
    short = rays.copy()
    short["tmax"] = np.where(np.isfinite(hb["t"]) & (hb["instance_id"] != 0xFFFFFFFF), hb["t"], 1.0)
    ob = so.trace_any(short, brute_force=True)
    sg = pa.scene_from_desc(desc)
    os.environ[which] = "1"
    if which == "PBRHIP_QUAD":
        os.environ["PBRHIP_QUAD_RAYS"] = str(1 << 30)
    try:
        assert_hits_equal(sg.trace_closest(rays), hb)
        assert np.array_equal(sg.trace_any(short), ob)
        for name in ("ggx", "sss"):
            _, g, o = pairs[name]
            rgba, cnt, _ = o.render(96, 64, 4, threads=4, math_mode=O.MATH_DEVICE)
            for tail in (0xFFFFFFFF, 3000):
                layer = pa.RenderLayer()
                pa.Render(g, 96, 64, 4, layer=layer, tail_paths=tail)
                assert np.array_equal(layer.count, cnt) and layer.rgba.tobytes() == rgba.tobytes(), (which, name, tail)
    finally:
        os.environ.pop(which, None)
        os.environ.pop("PBRHIP_QUAD_RAYS", None)


@pytest.mark.parametrize("seed", range(4))
def test_curve_soup(pa, seed):
    """Random cubic ribbons incl. duplicates, degenerate (all control points equal), zero-radius and straight ones, with a
    few triangles mixed in; random rays and axis-aligned rays aimed at control points.  Oracle brute force (whole-curve
    test) == oracle tree == GPU trees over the per-piece primitives (SAH and LBVH)."""
    from pbrlab_amd import scenes
    rng = np.random.RandomState(900 + seed)
    nc = 300
    cp = (rng.rand(nc, 4, 4).astype(np.float32) * 2 - 1)
    cp[:, 1:, :3] = cp[:, :1, :3] + np.cumsum((rng.rand(nc, 3, 3).astype(np.float32) - 0.5) * np.float32(0.3), axis=1)
    cp[..., 3] = np.float32(0.004) + rng.rand(nc, 4).astype(np.float32) * np.float32(0.03)
    cp[:20] = cp[20:40]                                     # duplicates
    cp[40:50, 1:] = cp[40:50, :1]                           # degenerate: a point
    cp[50:60, :, 3] = 0.0                                   # zero radius
    cp[60:70, 1:, :3] = cp[60:70, :1, :3] + np.arange(1, 4, dtype=np.float32)[None, :, None] * np.float32(0.1)   # straight
    cp[70:90, :, :3] = np.round(cp[70:90, :, :3] * 8) / 8   # grid-aligned control points
    curve = scenes.CurveShape("c", cp.reshape(-1, 4), (np.arange(nc, dtype=np.uint32) * 4))
    tv = np.array([[-1, -1, -0.9, 1], [1, -1, -0.9, 1], [1, 1, -0.9, 1], [-1, 1, -0.9, 1]], np.float32)
    mat = dict(scenes.PRINCIPLED_DEFAULTS, kind="principled", name="m")
    desc = scenes.SceneDesc(tv, np.zeros((0, 4), np.float32), [mat],
                            [scenes.Shape("back", np.array([[0, 1, 2], [0, 2, 3]], np.uint32), None, np.zeros(2, np.uint32))], [curve])
    so = O.oracle_scene_from_desc(desc)
    lo, hi = so.FetchSceneAABB()
    rays = scenes.random_rays((lo, hi), 5000, seed=seed)
    extra = np.zeros(1500, O.RAY_DT)
    tgt = cp.reshape(-1, 4)[rng.randint(nc * 4, size=1500), :3]
    axis = rng.randint(3, size=1500)
    dirs = np.zeros((1500, 3), np.float32)
    dirs[np.arange(1500), axis] = rng.choice([-1.0, 1.0], size=1500)
    extra["org"], extra["dir"] = tgt - dirs * np.float32(2.5), dirs
    extra["tmin"], extra["tmax"] = 0.0, 1e30
    rays = np.concatenate([rays, extra])
    hb = so.trace_closest(rays, brute_force=True)
    assert_hits_equal(so.trace_closest(rays), hb)
    assert ((hb["instance_id"] == 1)).sum() > 300          # the curve instance is hit often
    for builder in (pa.api.BVH_HOST_SAH, pa.api.BVH_GPU_LBVH):
        sg = pa.scene_from_desc(desc, bvh_builder=builder)
        assert_hits_equal(sg.trace_closest(rays), hb)
        short = rays.copy()
        short["tmax"] = np.where(hb["instance_id"] != 0xFFFFFFFF, hb["t"], 1.0)
        assert np.array_equal(sg.trace_any(short), so.trace_any(short, brute_force=True))


def test_curves_only_scene_on_both_trees(pa):
    """a scene without a single triangle (the Q tree then has no triangle slots, only chains of points): hits equal the
    oracle's brute force on both trees and both traversals; the frame (no emitter: black, but every path is traced) too"""
    from pbrlab_amd import scenes
    d = scenes.hair_scene(n_strands=300, n_segments=6, head_subdiv=1)
    desc = scenes.SceneDesc(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), d.materials, [], d.curves)
    so, sg = O.oracle_scene_from_desc(desc), pa.scene_from_desc(desc)
    lo, hi = so.FetchSceneAABB()
    glo, ghi = sg.FetchSceneAABB()
    assert np.array_equal(lo, glo) and np.array_equal(hi, ghi)
    rays = scenes.random_rays((lo, hi), 20000, seed=3)
    hb = so.trace_closest(rays, brute_force=True)
    assert (hb["instance_id"] != 0xFFFFFFFF).sum() > 2000
    for wide in ("1", "0"):
        os.environ["PBRHIP_WIDE"] = wide
        try:
            for simple in (False, True):
                if simple:
                    os.environ["PBRHIP_SIMPLE_TRAVERSAL"] = "1"
                assert_hits_equal(sg.trace_closest(rays), hb)
                assert np.array_equal(sg.trace_any(rays), so.trace_any(rays, brute_force=True))
            os.environ.pop("PBRHIP_SIMPLE_TRAVERSAL", None)
            lay = pa.RenderLayer()
            ok, st = pa.Render(sg, 64, 48, 3, layer=lay, flags=pa.api.RENDER_STATS)
            rgba, cnt, ost = so.render(64, 48, 3, threads=4, math_mode=O.MATH_DEVICE)
            assert lay.rgba.tobytes() == rgba.tobytes() and np.array_equal(lay.count, cnt)
            assert st["closest_rays"] + st["tail_closest_rays"] + st["pruned_rays"] == ost["closest_rays"] and st["closest_tris"] == 0
        finally:
            os.environ.pop("PBRHIP_WIDE", None)
            os.environ.pop("PBRHIP_SIMPLE_TRAVERSAL", None)


def _mini_scene(shapes_spec, materials):
    """shapes_spec: list of (name, verts (n,3), faces (m,3), material index)"""
    from pbrlab_amd import scenes
    vs, shapes, base = [], [], 0
    for name, v, f, mi in shapes_spec:
        v = np.asarray(v, np.float32)
        f = np.asarray(f, np.uint32)
        vs.append(np.concatenate([v, np.ones((len(v), 1), np.float32)], 1))
        shapes.append(scenes.Shape(name, f + np.uint32(base), None, np.full(len(f), mi, np.uint32)))
        base += len(v)
    return scenes.SceneDesc(np.concatenate(vs), np.zeros((0, 4), np.float32), materials, shapes)


@pytest.mark.parametrize("case", ["no_light", "light_only", "degenerate_light", "huge_emission_tiny_light", "black_materials"])
def test_degenerate_scenes_render_parity(pa, case):
    """scenes at the edges of the light tables and of the throughput arithmetic: no emitter at all, nothing but an emitter, an
    emitter with a zero-area triangle, a pin-point emitter (huge pdf), all-black materials -- image == oracle, bit for bit"""
    from pbrlab_amd import scenes
    m = lambda **kw: dict(scenes.PRINCIPLED_DEFAULTS, kind="principled", name="m", **kw)     # noqa: E731
    floor = ("floor", [[-1, -1, 1], [1, -1, 1], [1, -1, -1], [-1, -1, -1]], [[0, 1, 2], [0, 2, 3]], 0)
    back = ("back", [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1]], [[0, 1, 2], [0, 2, 3]], 0)
    light = ("light", [[-0.3, 0.9, -0.3], [0.3, 0.9, -0.3], [0.3, 0.9, 0.3], [-0.3, 0.9, 0.3]], [[0, 1, 2], [0, 2, 3]], 1)
    mats = [m(base_color=(0.7, 0.6, 0.5), specular=0.5, roughness=0.3), m(base_color=(0, 0, 0), specular=0.0)]
    if case == "no_light":
        desc = _mini_scene([floor, back, ("lamp", light[1], light[2], 1)], mats)          # not named light*: nothing emits
    elif case == "light_only":
        desc = _mini_scene([light], mats)
    elif case == "degenerate_light":
        lv = light[1] + [[0.3, 0.9, 0.3]]
        desc = _mini_scene([floor, back, ("light", lv, [[0, 1, 2], [0, 2, 3], [2, 4, 2]], 1)], mats)   # third triangle has zero area
    elif case == "huge_emission_tiny_light":
        e = 1e-4
        desc = _mini_scene([floor, back, ("light", [[-e, 0.9, -e], [e, 0.9, -e], [e, 0.9, e], [-e, 0.9, e]], light[2], 1)], mats)
        desc.light_emission = (3e6, 2e6, 1e6)
    else:
        desc = _mini_scene([floor, back, light], [m(base_color=(0, 0, 0), specular=0.0), mats[1]])
    so = O.oracle_scene_from_desc(desc)
    sg = pa.scene_from_desc(desc)
    rgba, cnt, _ = so.render(48, 40, 5, threads=4, math_mode=O.MATH_DEVICE)
    for tail in (0, 0xFFFFFFFF):
        layer = pa.RenderLayer()
        pa.Render(sg, 48, 40, 5, layer=layer, tail_paths=tail)
        assert np.array_equal(layer.count, cnt)
        assert layer.rgba.tobytes() == rgba.tobytes() or np.array_equal(np.isnan(layer.rgba), np.isnan(rgba)) and \
            np.array_equal(np.nan_to_num(layer.rgba).view(np.uint32), np.nan_to_num(rgba).view(np.uint32)), case
    if case == "no_light":
        assert not rgba[..., :3].any()


@pytest.mark.parametrize("foreign", ["0", "1", "3", "7"])
def test_random_walks_start_below_the_root(pa, foreign, monkeypatch):
    """Round 6 (dscene.h::SssEntry): the rays of a random walk start at the Q node that holds the walk's instance, with the references
    of OTHER instances' primitives inside its bounds on their stack.  Scene: a subdivided subsurface blob whose bounds are cut by a wall
    (a slab through the blob: a walk that meets it ends, random-walk-sss.h:371-384), which stands on the floor and touches a second,
    diffuse blob; the cap on foreign references (PBRHIP_SSS_FOREIGN, read at commit) moves the entry up and down the tree.  Image and ray
    counts are the oracle's whatever the entry."""
    from pbrlab_amd import scenes
    monkeypatch.setenv("PBRHIP_SSS_FOREIGN", foreign)
    m = lambda **kw: dict(scenes.PRINCIPLED_DEFAULTS, kind="principled", name="m", **kw)     # noqa: E731
    mats = [m(base_color=(0.8, 0.8, 0.8)), m(base_color=(0, 0, 0)), m(base_color=(0.9, 0.6, 0.4), subsurface=1.0, subsurface_radius=(0.5, 0.3, 0.2), subsurface_color=(0.9, 0.7, 0.5)),
            m(base_color=(0.3, 0.5, 0.8), specular=0.6, roughness=0.2)]
    v, f = scenes._icosphere(4)
    quad = [[0, 1, 2], [0, 2, 3]]
    vg, fg = [], []                                                                                          # a floor of 8 x 8 quads: the tree has something to drop
    for i in range(8):
        for j in range(8):
            x0, z0 = -2 + 0.5 * i, -2 + 0.5 * j
            fg += [[len(vg), len(vg) + 1, len(vg) + 2], [len(vg), len(vg) + 2, len(vg) + 3]]
            vg += [[x0, -0.5, z0 + 0.5], [x0 + 0.5, -0.5, z0 + 0.5], [x0 + 0.5, -0.5, z0], [x0, -0.5, z0]]
    spec = [("floor", vg, fg, 0),
            ("back", [[-2, -0.5, -2], [2, -0.5, -2], [2, 2, -2], [-2, 2, -2]], quad, 0),
            ("light", [[-0.5, 1.8, -0.5], [0.5, 1.8, -0.5], [0.5, 1.8, 0.5], [-0.5, 1.8, 0.5]], quad, 1),
            ("blob", v * 0.5 + np.array([0.9, 0.0, 0.9]), f, 2),                                            # rests on the floor (y = -0.5)
            ("wall", [[1.05, -0.5, 0.5], [1.05, -0.5, 1.3], [1.05, 0.4, 1.3], [1.05, 0.4, 0.5]], quad, 3),  # a small wall that cuts through the blob
            ("other", v * 0.3 + np.array([0.15, -0.2, 1.0]), f, 3)]                                         # touches the blob's bounds
    desc = _mini_scene(spec, mats)
    so = O.oracle_scene_from_desc(desc)
    sg = pa.scene_from_desc(desc)
    rgba, cnt, ost = so.render(64, 48, 6, threads=8, math_mode=O.MATH_DEVICE)
    for tail in (0xFFFFFFFF, 0):
        layer = pa.RenderLayer()
        ok, st = pa.Render(sg, 64, 48, 6, layer=layer, flags=pa.api.RENDER_STATS, tail_paths=tail)
        assert np.array_equal(layer.count, cnt) and layer.rgba.tobytes() == rgba.tobytes(), (foreign, tail)
        assert (st["closest_rays"] + st["tail_closest_rays"] + st["pruned_rays"], st["shadow_rays"] + st["tail_shadow_rays"]) == (ost["closest_rays"], ost["shadow_rays"])
    assert rgba[..., :3].any()


@pytest.mark.parametrize("nlight,nshapes", [(1, 1), (2, 1), (8, 1), (9, 1), (40, 1), (16, 8), (9, 9), (27, 9)])
def test_doomed_path_pruning_around_the_light_limits(pa, nlight, nshapes):
    """Paths whose next Russian roulette is known to fail are ended in shading when their ray cannot reach a light: with <= 8
    light primitives each one is tested exactly; with more primitives but <= 8 lights (emissive meshes) the lights' bounding
    boxes are tested; with more than 8 lights nothing is pruned.  In every case the image and the ray total (traced + pruned)
    are the oracle's.  The emitter is a fan of thin triangles under the ceiling, split over `nshapes` meshes."""
    from pbrlab_amd import scenes
    m = lambda **kw: dict(scenes.PRINCIPLED_DEFAULTS, kind="principled", name="m", **kw)     # noqa: E731
    mats = [m(base_color=(0.75, 0.7, 0.65), specular=0.4, roughness=0.4), m(base_color=(0.1, 0.1, 0.1), specular=0.0)]
    box = [("floor", [[-1, -1, 1], [1, -1, 1], [1, -1, -1], [-1, -1, -1]], [[0, 1, 2], [0, 2, 3]], 0),
           ("back", [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1]], [[0, 1, 2], [0, 2, 3]], 0),
           ("left", [[-1, -1, 1], [-1, -1, -1], [-1, 1, -1], [-1, 1, 1]], [[0, 1, 2], [0, 2, 3]], 0),
           ("top", [[-1, 1, -1], [1, 1, -1], [1, 1, 1], [-1, 1, 1]], [[0, 1, 2], [0, 2, 3]], 0)]
    xs = np.linspace(-0.5, 0.5, nlight + 1)
    lv = [[0.0, 0.95, -0.4]] + [[float(x), 0.95, 0.3] for x in xs]
    per = nlight // nshapes
    lights = [("light%d" % k, lv, [[0, j + 2, j + 1] for j in range(k * per, (k + 1) * per)], 1) for k in range(nshapes)]
    desc = _mini_scene(box + lights, mats)
    so = O.oracle_scene_from_desc(desc)
    sg = pa.scene_from_desc(desc)
    W, H, SPP = 56, 40, 6
    rgba, cnt, ost = so.render(W, H, SPP, threads=4, math_mode=O.MATH_DEVICE)
    for tail in (0, 0xFFFFFFFF, 300):
        layer = pa.RenderLayer()
        ok, st = pa.Render(sg, W, H, SPP, layer=layer, flags=pa.api.RENDER_STATS, tail_paths=tail)
        assert layer.rgba.tobytes() == rgba.tobytes(), (nlight, nshapes, tail)
        assert st["closest_rays"] + st["tail_closest_rays"] + st["pruned_rays"] == ost["closest_rays"]
        assert st["shadow_rays"] + st["tail_shadow_rays"] == ost["shadow_rays"]
        assert (st["pruned_rays"] > 0) == (nlight <= 8 or nshapes <= 8), (nlight, nshapes, st["pruned_rays"])
    assert rgba[..., :3].max() > 0


def transformed_scene(kind):
    from pbrlab_amd import scenes
    if kind == "hair":
        d = scenes.hair_scene(n_strands=400, n_segments=6, head_subdiv=2)
        for s in d.shapes:
            if s.name == "head":
                s.transform = scenes.instance_matrix((10, -25, 5), (1.0, 1.0, 1.0), (0.05, 0.02, -0.1))
        d.curves[0].transform = scenes.instance_matrix((10, -25, 5), (1.0, 1.0, 1.0), (0.05, 0.02, -0.1))
        return d
    d = scenes.cornell_scene(kind, monkey_subdiv=2, lucy_nu=64, lucy_nv=12)
    xf = {"monkey": scenes.instance_matrix((20, 35, -10), (1.2, 0.8, 1.1), (0.1, -0.05, 0.2)),
          "lucy": scenes.instance_matrix((0, 90, 0), (0.9, 0.9, 0.9), (-0.15, 0.0, 0.1)),
          "box": scenes.instance_matrix((0, 0, 45), (1.0, 2.0, 1.0), (0.0, 0.1, 0.0)),
          "light": scenes.instance_matrix((0, 15, 0), (1.0, 1.0, 1.0), (0.05, 0.0, 0.1))}
    for s in d.shapes:
        if s.name in xf:
            s.transform = xf[s.name]
    return d


@pytest.mark.parametrize("kind", ["ggx", "sss", "hair"])
def test_instance_transforms(pa, kind):
    """Scene::CreateInstance with a transform (scene.cc:106-155, raytracer_impl.cc:49-84): rotated / scaled / translated
    instances, also of the emissive mesh.  The raytracer sees the transformed geometry; normals, texcoords and light
    sampling stay local as in the reference ("TODO transform").  Hits and images equal the oracle's, bit for bit, for
    both BVH builders; a transform that is the identity changes nothing; singular matrices are refused."""
    from pbrlab_amd import scenes
    desc = transformed_scene(kind)
    so = O.oracle_scene_from_desc(desc)
    lo, hi = so.FetchSceneAABB()
    rays = scenes.random_rays((lo, hi), 30000, seed=4)
    ho = so.trace_closest(rays)
    assert_hits_equal(ho[:1200], so.trace_closest(rays[:1200], brute_force=True))
    rgba, cnt, ost = so.render(72, 56, 3, threads=4, math_mode=O.MATH_DEVICE)
    for builder in (pa.api.BVH_HOST_SAH, pa.api.BVH_GPU_LBVH):
        sg = pa.scene_from_desc(desc, bvh_builder=builder)
        glo, ghi = sg.FetchSceneAABB()
        assert np.array_equal(lo, glo) and np.array_equal(hi, ghi)
        assert_hits_equal(sg.trace_closest(rays), ho)
        sr = rays.copy()
        sr["tmax"] = 0.7
        assert np.array_equal(sg.trace_any(sr), so.trace_any(sr))
        for tail in (0, 0xFFFFFFFF):
            lay = pa.RenderLayer()
            ok, st = pa.Render(sg, 72, 56, 3, layer=lay, flags=pa.api.RENDER_STATS, tail_paths=tail)
            assert lay.rgba.tobytes() == rgba.tobytes(), (builder, tail)
            assert (st["closest_rays"] + st["tail_closest_rays"] + st["pruned_rays"], st["shadow_rays"] + st["tail_shadow_rays"]) == (ost["closest_rays"], ost["shadow_rays"])
    # the image really depends on the transforms ...
    plain = transformed_scene(kind)
    for s in plain.shapes:
        s.transform = None
    for c in plain.curves:
        c.transform = None
    a, b = pa.RenderLayer(), pa.RenderLayer()
    pa.Render(pa.scene_from_desc(plain), 72, 56, 3, layer=a)
    assert a.rgba.tobytes() != rgba.tobytes()
    # ... and an explicit identity is the same scene as no transform
    for s in plain.shapes:
        s.transform = np.eye(4, dtype=np.float32)
    pa.Render(pa.scene_from_desc(plain), 72, 56, 3, layer=b)
    assert a.rgba.tobytes() == b.rgba.tobytes()


def test_instance_transform_errors(pa):
    s = pa.Scene()
    v = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [0, 1, 0, 1]], np.float32)
    m = s.AddTriangleMesh(v, None, None, np.array([[0, 1, 2]], np.uint32))
    ls = s.CreateLocalScene()
    s.AddMeshToLocalScene(ls, m)
    sing = np.eye(4, dtype=np.float32)
    sing[1, 1] = 0.0
    for bad in (sing, np.full((4, 4), np.nan, np.float32), np.zeros((4, 4), np.float32)):
        with pytest.raises(pa.PbrHipError) as e:
            s.CreateInstance(ls, bad)
        assert e.value.code == -1
    assert s.CreateInstance(ls, np.diag([2, 2, 2, 1]).astype(np.float32)) == 0      # the failed calls created nothing


def test_multi_geometry_instances_and_duplicate_instances(pa):
    """A local scene with two meshes (geometry ids 0 and 1) instantiated TWICE (exact duplicates: every hit is a tie that the
    smaller instance id must win), lights attached to one geometry of one instance, materials overridden per instance."""
    from pbrlab_amd import scenes
    mats = [dict(scenes.PRINCIPLED_DEFAULTS, kind="principled", name="a", base_color=(0.8, 0.3, 0.2), specular=0.0),
            dict(scenes.PRINCIPLED_DEFAULTS, kind="principled", name="b", base_color=(0.2, 0.7, 0.9), specular=1.0, roughness=0.2)]
    quad = lambda z, s=1.0: np.array([[-s, -s, z, 1], [s, -s, z, 1], [s, s, z, 1], [-s, s, z, 1]], np.float32)   # noqa: E731
    faces = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)

    def build(S, mk):
        m0, m1 = S.AddMaterialParam(mk(mats[0])), S.AddMaterialParam(mk(mats[1]))
        wall = S.AddTriangleMesh(quad(-1.0), None, None, faces, None, None, np.array([m0, m0], np.uint32))
        lamp = S.AddTriangleMesh(quad(0.5, 0.25), None, None, faces[:, ::-1], None, None, np.array([m1, m1], np.uint32))
        ls = S.CreateLocalScene()
        assert S.AddMeshToLocalScene(ls, wall) == 0 and S.AddMeshToLocalScene(ls, lamp) == 1
        i0, i1 = S.CreateInstance(ls, None), S.CreateInstance(ls, None)
        lid = S.AddLightParam((5.0, 4.0, 3.0))
        S.AttachLightParamIdsToInstance(i0, [np.full(2, 0xFFFFFFFF, np.uint32), np.full(2, lid, np.uint32)])
        S.AttachMaterialParamIdsToInstance(i1, [np.array([m1, m1], np.uint32), np.array([m0, m0], np.uint32)])
        S.CommitScene()
        return S

    sg = build(pa.Scene(), pa.make_principled)
    so = build(O.OracleScene(), O.make_principled)
    lo, hi = so.FetchSceneAABB()
    rays = scenes.random_rays((lo, hi), 4000, seed=2)
    hg, ho = sg.trace_closest(rays), so.trace_closest(rays, brute_force=True)
    assert_hits_equal(hg, ho)
    hit = hg["instance_id"] != 0xFFFFFFFF
    assert hit.sum() > 500 and (hg["instance_id"][hit] == 0).all() and set(hg["geom_id"][hit]) == {0, 1}
    rgba, cnt, _ = so.render(40, 40, 6, threads=4, math_mode=O.MATH_DEVICE)
    layer = pa.RenderLayer()
    pa.Render(sg, 40, 40, 6, layer=layer)
    assert np.array_equal(layer.count, cnt) and layer.rgba.tobytes() == rgba.tobytes() and rgba[..., :3].max() > 0


def test_seeding_corners(pa, pairs):
    """RNG((pass << 32) + y*W + x, seed_seq): very large pass indices, other stream selectors, ranks that own no tile"""
    desc, sg, so = pairs["ggx"]
    for first_pass, seed_seq in [(1 << 20, 1234567890), (0xFFFFFFF0, 1234567890), (3, 1), (0, 0xFFFFFFFFFFFFFFFF), (7, 0x0123456789ABCDEF)]:
        rgba, cnt, _ = so.render(40, 24, 5, first_pass=first_pass, seed_seq=seed_seq, threads=4, math_mode=O.MATH_DEVICE)
        layer = pa.RenderLayer()
        pa.Render(sg, 40, 24, 5, layer=layer, first_pass=first_pass, seed_seq=seed_seq)
        assert np.array_equal(layer.count, cnt) and layer.rgba.tobytes() == rgba.tobytes(), (first_pass, seed_seq)
    a = pa.RenderLayer()
    pa.Render(sg, 40, 24, 5, layer=a)
    b = pa.RenderLayer()
    pa.Render(sg, 40, 24, 5, layer=b, first_pass=1)
    assert a.rgba.tobytes() != b.rgba.tobytes()                                   # different passes, different samples
    empty = pa.RenderLayer()
    ok, st = pa.Render(sg, 40, 24, 5, layer=empty, tile_rank=5, tile_world=9)     # one 64x64 tile only: rank 5 owns nothing
    assert ok and not empty.rgba.any() and not empty.count.any() and st["samples"] == 0
