import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrlab_amd as pa
from pbrlab_amd import scenes, api
s = pa.scene_from_desc(scenes.cornell_scene("ggx", seed=1))
for n in (64, 1024, 16384):
    rng = np.random.RandomState(5)
    rays = np.zeros(n, dtype=api.RAY_DT)
    rays["org"] = (rng.rand(n, 3) * 1.6 - 0.8).astype(np.float32)
    d = rng.randn(n, 3).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["dir"] = d; rays["tmin"] = 1e-4; rays["tmax"] = np.inf
    os.environ["PBRHIP_QUAD"] = "1"
    for rep in range(3):
        h = s.trace_closest(rays)
    m = h["instance_id"] != 0xFFFFFFFF
    u = h["u"][m].view(np.uint32); ticks = h["v"][m] * 10.0
    nodes, leaves, pops = u & 1023, (u >> 10) & 1023, (u >> 20) & 1023
    steps = nodes + leaves
    print(f"n={n}: hit rays {m.sum()}: nodes mean {nodes.mean():.1f} max {nodes.max()}, leaves mean {leaves.mean():.1f} max {leaves.max()}, pops mean {pops.mean():.1f}; "
          f"ns per ray mean {ticks.mean():.0f} max {ticks.max():.0f}; ns per step (nodes+leaves) mean {(ticks / steps).mean():.0f}, of the longest ray {ticks.max() / steps[np.argmax(ticks)]:.0f}")
