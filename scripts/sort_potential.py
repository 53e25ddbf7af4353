"""How much would ray reordering buy the traversal kernel?  The same 8 M incoherent rays (random origins in the C2 scene, random
directions) through pbrhip_trace_closest / pbrhip_trace_any, as they come and sorted by (direction octant, Morton code of the
origin) -- an upper bound for what a partition in k_compact could do (a full sort, for free)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrlab_amd as pa
from pbrlab_amd import scenes
desc = scenes.cornell_scene("ggx", seed=1)
s = pa.scene_from_desc(desc)
lo, hi = s.FetchSceneAABB()
n = 8 << 20
rays = scenes.random_rays((lo, hi), n, seed=5)


def part1by2(x):
    x = x.astype(np.uint64) & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x


q = np.clip(((rays["org"] - lo) / (hi - lo) * 1024).astype(np.int64), 0, 1023)
morton = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)
octant = ((rays["dir"][:, 0] < 0).astype(np.uint64) | ((rays["dir"][:, 1] < 0).astype(np.uint64) << 1) | ((rays["dir"][:, 2] < 0).astype(np.uint64) << 2))
orders = {"as generated": np.arange(n), "octant only": np.argsort(octant, kind="stable"), "octant + morton(origin, 10 bits)": np.argsort((octant << 30) | morton, kind="stable"),
          "morton only": np.argsort(morton, kind="stable")}
for name, idx in orders.items():
    r = np.ascontiguousarray(rays[idx])
    best = besta = 1e9
    for rep in range(4):
        t = time.perf_counter(); h = s.trace_closest(r); best = min(best, time.perf_counter() - t)
        t = time.perf_counter(); o = s.trace_any(r); besta = min(besta, time.perf_counter() - t)
    print(f"{name:36s} closest {best*1e3:7.2f} ms   any {besta*1e3:7.2f} ms   (incl. {n*32/1e6:.0f} MB H2D + results D2H)")
# copies only: an empty-ish scene would do; here: rays with tmax 0 (no traversal beyond the root)
z = rays.copy(); z["tmax"] = 0.0
t = time.perf_counter(); s.trace_any(z); print(f"copies + root only: {(time.perf_counter()-t)*1e3:.2f} ms")
