"""hook-level timing of small ray batches (random rays through the C2 scene): wall time of pbrhip_trace_closest / any"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrlab_amd as pa
from pbrlab_amd import scenes
desc = scenes.cornell_scene("ggx", seed=1)
s = pa.scene_from_desc(desc)
lo, hi = s.FetchSceneAABB()
for n in (64, 1000, 10000, 100000, 400000):
    rays = scenes.random_rays((lo, hi), n, seed=3)
    best = 1e9
    for rep in range(7):
        t = time.perf_counter(); h = s.trace_closest(rays); best = min(best, time.perf_counter() - t)
    besta = 1e9
    for rep in range(7):
        t = time.perf_counter(); o = s.trace_any(rays); besta = min(besta, time.perf_counter() - t)
    print(f"n {n:7d}: closest {best*1e6:8.1f} us   any {besta*1e6:8.1f} us   hits {(h['instance_id'] != 0xFFFFFFFF).mean():.3f} occluded {o.mean():.3f}")
