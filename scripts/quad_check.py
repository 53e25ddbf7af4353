"""PBRHIP_QUAD=1 (one ray per quad of lanes, dtrace_quad.h) against the default hooks: bit-exact on soups and on incoherent rays
of the C2 scene; run under rocprofv3 --kernel-trace --stats to read the hook kernels' durations for N rays (env N, default 4096)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pbrlab_amd as pa
from pbrlab_amd import scenes

def both(s, rays, short):
    os.environ.pop("PBRHIP_QUAD", None)
    a, b = s.trace_closest(rays), s.trace_any(short)
    os.environ["PBRHIP_QUAD"] = "1"
    c, d = s.trace_closest(rays), s.trace_any(short)
    os.environ.pop("PBRHIP_QUAD", None)
    return a.tobytes() == c.tobytes(), np.array_equal(b, d), a

import _soups
for seed in range(3):
    desc, so, rays = _soups.triangle_soup(seed, 30)
    s = pa.scene_from_desc(desc)
    h = s.trace_closest(rays)
    short = rays.copy()
    short["tmax"] = np.where(np.isfinite(h["t"]) & (h["instance_id"] != 0xFFFFFFFF), h["t"], 1.0)
    print("soup", seed, both(s, rays, short)[:2], len(rays))

s = pa.scene_from_desc(scenes.cornell_scene("ggx", seed=1))
n = int(os.environ.get("N", "4096"))
rng = np.random.RandomState(5)
rays = np.zeros(n, dtype=rays.dtype)
rays["org"] = (rng.rand(n, 3) * 1.6 - 0.8).astype(np.float32)
d = rng.randn(n, 3).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays["dir"] = d
rays["tmin"] = 1e-4
rays["tmax"] = np.inf
ok1, ok2, h = both(s, rays, rays)
print("c2 scene", n, "rays:", ok1, ok2, "hit fraction", float(np.mean(h["instance_id"] != 0xFFFFFFFF)))
for _ in range(3):
    both(s, rays, rays)
