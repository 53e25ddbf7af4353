"""per-wave timeline of every k_trace launch of one render (PBRHIP_WAVE_LOG): when do the waves of a launch finish?
usage: python scripts/wave_log.py [world] (env VARIANT, SPP)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrlab_amd as pa
from pbrlab_amd import scenes
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variant = os.environ.get("VARIANT", "ggx")
desc = scenes.hair_scene(seed=1) if variant == "hair" else scenes.cornell_scene(variant, seed=1)
s = pa.scene_from_desc(desc)
layer = pa.RenderLayer()
os.environ["PBRHIP_STREAMS"] = "1"
pa.Render(s, 1920, 1080, int(os.environ.get("SPP", "64")), layer=layer, tile_world=world, shard_block=16 if world > 1 else 0)
path = "/tmp/wave_log.bin"
os.environ["PBRHIP_WAVE_LOG"] = path
ok, st = pa.Render(s, 1920, 1080, int(os.environ.get("SPP", "64")), layer=layer, tile_world=world, shard_block=16 if world > 1 else 0, flags=pa.api.RENDER_STATS if os.environ.get("STATS") else 0)
a = np.fromfile(path, dtype=np.uint64).reshape(64, 8192, 4)
for L in range(64):
    w = a[L]
    used = w[:, 1] != 0
    if not used.any():
        continue
    t0 = w[used, 0].min()
    start = (w[used, 0] - t0) / 100.0      # us (100 MHz)
    end = (w[used, 1] - t0) / 100.0
    turns = w[used, 3]
    refills = w[used, 2] & np.uint64(0xFFFFFFFF)
    rticks = (w[used, 2] >> np.uint64(32)) / 100.0  # us spent in refills (STATS=1)
    q = np.percentile(end, [10, 50, 90, 99, 100])
    if not os.environ.get("STATS"):   # word 2 = when the wave found the queue empty
        ex = w[used, 2]
        has = ex != 0
        exu = (ex[has] - t0) / 100.0
        qe = np.percentile(exu, [1, 10, 50, 90, 99, 100]) if has.any() else [0] * 6
        after = end[has] - exu
        qa = np.percentile(after, [10, 50, 90, 99, 100]) if has.any() else [0] * 5
        mean_end = end.mean()
        print(f"launch {L:2d}: waves {used.sum():5d} | end p10 {q[0]:7.1f} p50 {q[1]:7.1f} p90 {q[2]:7.1f} p99 {q[3]:7.1f} max {q[4]:7.1f} us, mean {mean_end:7.1f} = {100 * mean_end / q[4]:.0f} % of max | "
              f"queue found empty at p1 {qe[0]:7.1f} p10 {qe[1]:7.1f} p50 {qe[2]:7.1f} p90 {qe[3]:7.1f} p99 {qe[4]:7.1f} max {qe[5]:7.1f} | end - empty: p10 {qa[0]:6.1f} p50 {qa[1]:6.1f} p90 {qa[2]:6.1f} p99 {qa[3]:6.1f} max {qa[4]:6.1f} us")
        continue
    print(f"launch {L:2d}: waves {used.sum():5d}  start p50 {np.median(start):6.1f} max {start.max():6.1f} us | end p10 {q[0]:7.1f} p50 {q[1]:7.1f} p90 {q[2]:7.1f} "
          f"p99 {q[3]:7.1f} max {q[4]:7.1f} us | turns p50 {int(np.median(turns)):5d} max {int(turns.max()):5d} | us/turn p50 {np.median((end - start) / np.maximum(turns, 1)):.2f} | refills p50 {int(np.median(refills))} in {np.median(rticks):.1f} us = {100 * rticks.sum() / np.maximum((end - start).sum(), 1e-9):.0f} % of the waves' time")
