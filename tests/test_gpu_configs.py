"""GPU (-m gpu): every BASELINE.json configuration at its own geometry and resolution.

  C1  S-cornell Lambert-only, 256 x 256 x 4        whole frame vs oracle and vs the committed fixture
  C2  S-cornell GGX, 1920 x 1080 (64 spp)          whole frame at 1 spp and at 64 spp vs oracle[glibcf] and oracle[libm] (bit-exact);
                                                   the 64-spp headline frame: tests/test_gpu_parity.py::test_headline_configuration_spot_parity
  C3  S-cornell SSS, 1920 x 1080 (256 spp)         the same at 1 spp + spot pixels over passes 200..255
  C4  S-hair + head, 1920 x 1080 (128 spp)         the same at 1 spp
  C5  S-cornell SSS + S-hair, 3840 x 2160 (1024)   size-independent properties + spot parity + whole frame at 1 spp

The whole-frame checks run the oracle (oracle/, the CPU restatement) on all host cores at full resolution and 1 spp: a few
seconds each.  Bars (BASELINE.json north_star: bit-exact integers, 1e-4 relative L2 on radiance): the device computes cos / sin /
exp / log with include/pbr_glibcf.h, glibc 2.35's x86-64 FMA float functions restated operation by operation, so against
oracle[glibcf] every pixel is bit-identical, and on a host whose libm IS that glibc (this image: O.libm_is_glibcf(), pinned by
tests/test_glibcf.py) the same holds against oracle[libm], the reference's own arithmetic -- 0 differing pixels is asserted there;
on any other host libm the 1e-4 bar is asserted as a tolerance and the measured values are printed."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import _oracle as O  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REL_L2_TOL = 1e-4
THREADS = O.oracle_threads()


@pytest.fixture(scope="module")
def pa():
    import pbrlab_amd as pa
    if pa.device_count() < 1:
        pytest.fail("no HIP device: the GPU tests must run on an MI355X (there is no CPU fallback)")
    pa.set_device(0)
    return pa


def config_desc(name):
    from pbrlab_amd import scenes
    return {"c1": lambda: scenes.cornell_scene("lambert", seed=1), "c2": lambda: scenes.cornell_scene("ggx", seed=1),
            "c3": lambda: scenes.cornell_scene("sss", seed=1), "c4": lambda: scenes.hair_scene(seed=1),
            "c5": lambda: scenes.cornell_hair_scene("sss", seed=1)}[name]()


def rel_l2(a, b):
    return float(np.linalg.norm(a[..., :3].astype(np.float64) - b[..., :3]) / max(np.linalg.norm(b[..., :3].astype(np.float64)), 1e-30))


def spot_parity(so, layer, W, H, passes, n, seed):
    """n random pixels: every sample traced by the oracle, summed in pass order, compared as bits"""
    rng = np.random.RandomState(seed)
    for _ in range(n):
        x, y = int(rng.randint(W)), int(rng.randint(H))
        tot = np.zeros(3, np.float32)
        for p in passes:
            rad, _, _, _ = so.sample_trace(W, H, x, y, p, math_mode=O.MATH_DEVICE)
            tot = tot + rad
        assert np.array_equal(tot.view(np.uint32), layer.rgba[y, x, :3].view(np.uint32)), (x, y)


def test_c1_whole_frame_vs_oracle_and_fixture(pa):
    """BASELINE configs[0]: 256 x 256, 4 spp, Lambert-only closures (one sample of this frame runs 10 825 bounces: Russian
    roulette with an unclamped probability never ends a path whose throughput stays at 1)."""
    desc = config_desc("c1")
    sg, so = pa.scene_from_desc(desc), O.oracle_scene_from_desc(desc)
    lay = pa.RenderLayer()
    ok, st = pa.Render(sg, 256, 256, 4, layer=lay, flags=pa.api.RENDER_STATS)
    assert ok is True and (lay.count == 4).all()
    rgba, cnt, ost = so.render(256, 256, 4, threads=THREADS, math_mode=O.MATH_DEVICE)
    assert lay.rgba.tobytes() == rgba.tobytes() and np.array_equal(lay.count, cnt)
    assert (st["closest_rays"] + st["tail_closest_rays"] + st["pruned_rays"], st["shadow_rays"] + st["tail_shadow_rays"]) == (ost["closest_rays"], ost["shadow_rays"])
    fx = np.load(os.path.join(G, "c1_oracle.npz"))
    # the committed fixture: the oracle's frame with libm arithmetic (made on this image: glibc 2.35, x86-64 FMA variants), stored
    # as its difference from the frame with correctly rounded functions
    libm = fx["f64r_rgb"].copy().reshape(-1, 3)
    libm[fx["libm_idx"]] = fx["libm_val"]
    libm = libm.reshape(256, 256, 3)
    assert (ost["closest_rays"], ost["shadow_rays"]) == tuple(int(v) for v in fx["libm_rays"])
    nd = int((lay.rgba[..., :3] != libm).any(axis=2).sum())
    r, r64 = rel_l2(lay.rgba, libm), rel_l2(lay.rgba, fx["f64r_rgb"])
    print(f"c1: 0 of 65536 pixels differ from oracle[glibcf]; {nd} differ from the libm fixture (rel L2 {r:.2e}); vs the correctly rounded fixture rel L2 {r64:.2e}")
    assert nd == 0 and r64 < REL_L2_TOL, (nd, r64)


@pytest.mark.parametrize("config", ["c2", "c3", "c4", "c5"])
def test_full_size_whole_frame_both_math_modes(pa, config):
    """the configuration's own scene and resolution, 1 spp, the whole frame: bit-identical to oracle[glibcf] and -- where the host
    libm is the glibc the device restates -- to oracle[libm], the reference's own arithmetic; on any other libm within 1e-4"""
    desc = config_desc(config)
    W, H = (3840, 2160) if config == "c5" else (1920, 1080)
    sg, so = pa.scene_from_desc(desc), O.oracle_scene_from_desc(desc)
    lay = pa.RenderLayer()
    ok, st = pa.Render(sg, W, H, 1, layer=lay, flags=pa.api.RENDER_STATS)
    assert ok is True and (lay.count == 1).all() and np.isfinite(lay.rgba).all()
    rgba, cnt, ost = so.render(W, H, 1, threads=THREADS, math_mode=O.MATH_DEVICE)
    ndiff = int((lay.rgba != rgba).any(axis=2).sum())
    assert ndiff == 0, ndiff
    assert (st["closest_rays"] + st["tail_closest_rays"] + st["pruned_rays"], st["shadow_rays"] + st["tail_shadow_rays"]) == (ost["closest_rays"], ost["shadow_rays"])
    libm, _, _ = so.render(W, H, 1, threads=THREADS, math_mode=O.MATH_LIBM)
    r = rel_l2(lay.rgba, libm)
    d = np.abs(lay.rgba[..., :3] - libm[..., :3]).max(axis=2)
    nl = int((d > 0).sum())
    if O.libm_is_glibcf():
        print(f"{config}: {W}x{H}x1: 0 pixels differ from oracle[glibcf]; {nl} of {W * H} differ from oracle[libm] (host libm = glibc 2.35 FMA variants)")
        assert nl == 0, nl
        return
    # (another libm) A last-ulp difference in cos / sin / exp / log can flip a discrete decision of ONE sample (a hit on the
    # other side of an edge, a Russian-roulette draw): that sample then differs by O(1); everything else differs by rounding noise.
    flipped = d > 1e-3 * np.maximum(libm[..., :3].max(axis=2), 0.05)
    nflip = int(flipped.sum())
    keep = ~flipped
    r_rest = float(np.linalg.norm((lay.rgba[..., :3] - libm[..., :3])[keep].astype(np.float64)) / np.linalg.norm(libm[..., :3][keep].astype(np.float64)))
    spp_cfg = {"c2": 64, "c3": 256, "c4": 128, "c5": 1024}[config]
    print(f"{config}: {W}x{H}x1: 0 pixels differ from oracle[glibcf]; vs oracle[libm] (NOT glibc 2.35 FMA): rel L2 {r:.2e} at 1 spp (x 1/sqrt({spp_cfg}) = "
          f"{r / np.sqrt(spp_cfg):.1e} at the configuration's spp), {nl} of {W * H} pixels differ at all, {nflip} samples flipped, "
          f"rel L2 without them {r_rest:.1e}")
    assert nflip <= max(8, 2e-5 * W * H), nflip
    assert r_rest < REL_L2_TOL, r_rest
    # (the bar at the configuration's own sample count is MEASURED by test_libm_tolerance_at_the_configurations_own_spp)


@pytest.mark.parametrize("config", ["c2", "c3", "c4", "c5"])
def test_libm_tolerance_at_the_configurations_own_spp(pa, config):
    """The bar against the reference's own arithmetic (oracle[libm]), MEASURED AT THE CONFIGURATION'S OWN SAMPLE COUNT, and the
    sample count reached is asserted, not scaled down (a slower host must fail here, not silently lower it):
      C2 / C3 / C4  the whole 1920 x 1080 frame at 64 / 256 / 128 spp
      C5            the 3840 x 2160 frame at its 1024 spp on every 64th of the reference's 64 x 64 tiles (tile i % 64 == 0: 32 tiles
                    spread over the frame, 131 k pixels, 134 M samples; the seeds are the full frame's, (pass << 32) + y * 3840 + x) --
                    the whole frame is 8.5 G samples, hours of oracle time
    Where the host libm is the glibc the device restates (this image), the frames must be IDENTICAL: 0 differing pixels, all five
    configurations (round 5; with correctly rounded device functions C5 measured 1.1e-3 here, above the bar).  On any other libm:
    plain relative L2 < 1e-4 for C2 / C3 / C4, and for C5 the robust figures (relative L2 without the pixels holding a flipped
    sample, their number, a ceiling on the plain figure).
    The oracle runs with 16 x 16-pixel jobs (schedule-independent image, tests/_oracle.py JOBS_BLOCKS).  PBR_TOL_SECONDS (default
    420) is a guard: if one calibration pass says the oracle would need longer than that, the test FAILS with the figures."""
    import time
    limit = float(os.environ.get("PBR_TOL_SECONDS", "420"))
    desc = config_desc(config)
    W, H = (3840, 2160) if config == "c5" else (1920, 1080)
    spp = {"c2": 64, "c3": 256, "c4": 128, "c5": 1024}[config]
    world = 64 if config == "c5" else 1   # C5: tiles i % 64 == 0 of the full frame
    sg, so = pa.scene_from_desc(desc), O.oracle_scene_from_desc(desc)
    t0 = time.time()
    so.render(W, H, 1, tile_rank=0, tile_world=world, threads=THREADS, math_mode=O.MATH_LIBM)   # calibration: one pass
    per_pass = max(time.time() - t0, 1e-3)
    if per_pass * spp > limit:
        pytest.fail(f"{config}: the oracle needs {per_pass:.2f} s per pass on {THREADS} threads = {per_pass * spp:.0f} s for the configuration's "
                    f"{spp} spp (limit PBR_TOL_SECONDS = {limit:.0f} s): the bar cannot be measured at the configuration's spp on this host")
    lay = pa.RenderLayer()
    ok, _ = pa.Render(sg, W, H, spp, layer=lay, tile_rank=0, tile_world=world)
    t0 = time.time()
    libm, cnt, _ = so.render(W, H, spp, tile_rank=0, tile_world=world, threads=THREADS, math_mode=O.MATH_LIBM)
    dt = time.time() - t0
    assert ok is True and np.array_equal(cnt, lay.count)
    mine = cnt > 0
    npx = int(mine.sum())
    assert (cnt[mine] == spp).all() and (npx == W * H if world == 1 else 0 < npx < W * H)
    r = rel_l2(lay.rgba[mine], libm[mine])
    d = np.abs(lay.rgba[..., :3] - libm[..., :3]).max(axis=2)
    nl = int((d > 0).sum())
    # pixels that hold a FLIPPED sample: a last-ulp difference between two roundings of cos / sin / exp / log changed a
    # discrete decision of one sample (an edge, a Russian-roulette draw) and that sample differs by O(its radiance): the pixel's sum
    # then differs by more than 1e-3 of itself (rounding noise is ~1e-6 of it)
    flipped = d > 1e-3 * np.maximum(libm[..., :3].max(axis=2), 0.05 * spp)
    nflip = int(flipped.sum())
    keep = mine & ~flipped
    r_rest = rel_l2(lay.rgba[keep], libm[keep])
    import platform
    print(f"host libm: {' '.join(platform.libc_ver())} on {platform.machine()}; the device restates it bit for bit: {bool(O.libm_is_glibcf())}")
    print(f"{config}: {W}x{H} x {spp} spp = the configuration's spp" + (f" on tiles i % {world} == 0" if world > 1 else "") +
          f" ({npx} pixels, {npx * spp / 1e6:.1f} M samples): rel L2 vs oracle[libm] {r:.2e} (bar 1e-4), {nl} pixels differ, {nflip} hold a flipped sample, "
          f"rel L2 without those {r_rest:.1e}; oracle {npx * spp / dt / 1e6:.2f} Msamples/s on {THREADS} threads, {dt:.0f} s")
    if O.libm_is_glibcf():
        assert nl == 0 and r == 0.0, (nl, r)   # the device's functions ARE this libm's: the reference's arithmetic, bit for bit
        return
    assert r_rest < REL_L2_TOL and nflip <= 5e-3 * npx, (r_rest, nflip)
    if config != "c5":
        assert r < REL_L2_TOL, r
    else:
        # (another libm) C5 at its own 1024 spp does not meet the plain 1e-4 bar between two DIFFERENT roundings of cos / sin / exp /
        # log -- measured 1.1e-3 between glibc's and the correctly rounded ones: the frame is dark (mean radiance 0.011) and Russian
        # roulette with an unclamped survival probability (render.cc:66-68, Q1) lets rare hair / subsurface paths carry a large
        # throughput, so the ~200 pixels of 130 048 that hold a flipped sample of such a path dominate the norm (without the 100
        # largest differences: 7.0e-5; without 1 000: 1.3e-6).  Any two libm builds differ from each other in the same way.
        # that is the 1e-4 contract NOT MET against this host's libm, not a weaker bar to assert -- report it as such
        if not r < REL_L2_TOL:
            pytest.xfail(f"C5 at 1024 spp against a libm the device does not restate ({' '.join(platform.libc_ver())}): plain rel L2 {r:.2e} >= 1e-4 "
                         f"(NOT MET; without the {nflip} pixels holding a flipped sample: {r_rest:.1e})")


def test_c3_high_pass_indices(pa):
    """C3 (random-walk SSS, 256 spp): passes 200..255 of the frame -- the RNG seeds (pass << 32) + pixel of the last
    passes -- rendered as a resumed frame (first_pass) and checked sample by sample on random pixels"""
    desc = config_desc("c3")
    sg, so = pa.scene_from_desc(desc), O.oracle_scene_from_desc(desc)
    W, H = 1920, 1080
    lay = pa.RenderLayer()
    fin = C.c_size_t(0)
    ok, st = pa.Render(sg, W, H, 56, layer=lay, first_pass=200, finish_pass=fin)
    assert ok is True and fin.value == 56 and (lay.count == 56).all() and np.isfinite(lay.rgba).all()
    spot_parity(so, lay, W, H, range(200, 256), 24, seed=3)


def test_c5_configuration(pa):
    """BASELINE configs[4]: S-cornell (SSS) + S-hair at 3840 x 2160 (4.8 M curve pieces + 545 k triangles, 8.3 M pixels;
    the 1024 spp of the configuration are 8.5 G samples -- here 2 spp): chunk, shard, block and group independence on the
    full-size frame, the in-library multi-device render, and spot parity with the oracle"""
    desc = config_desc("c5")
    sg = pa.scene_from_desc(desc)
    W, H, SPP = 3840, 2160, 2
    a, b = pa.RenderLayer(), pa.RenderLayer()
    pa.Render(sg, W, H, SPP, layer=a)
    assert (a.count == SPP).all() and np.isfinite(a.rgba).all() and (a.rgba[..., :3] >= 0).all()
    pa.Render(sg, W, H, SPP, layer=b, max_paths_in_flight=W * H)                 # two chunks
    assert a.rgba.tobytes() == b.rgba.tobytes()
    pa.Render(sg, W, H, SPP, layer=b, num_streams=2, tail_paths=0xFFFFFFFF)       # two groups, no tail kernel
    assert a.rgba.tobytes() == b.rgba.tobytes()
    acc = np.zeros_like(a.rgba)
    for r in range(8):                                                           # the 8-GPU dealing of the configuration
        p = pa.RenderLayer()
        pa.Render(sg, W, H, SPP, layer=p, tile_rank=r, tile_world=8, shard_block=16)
        acc += p.rgba
    assert acc.tobytes() == a.rgba.tobytes()
    ndev = pa.device_count()
    reps = [sg] + [pa.replicate(sg, g % ndev) for g in range(1, 4)]
    pa.RenderMulti(reps, W, H, SPP, layer=b)                                     # 4 ranks of this process, shards gathered in the library
    assert a.rgba.tobytes() == b.rgba.tobytes() and np.array_equal(a.count, b.count)
    for r in reps[1:]:
        r.close()
    so = O.oracle_scene_from_desc(desc)
    spot_parity(so, a, W, H, range(SPP), 48, seed=5)
