"""Traversal micro-benchmark through the C-ABI hooks: ONE dispatch of the production traversal kernel over a
large array of incoherent rays in the C2 scene (cheap to profile with rocprofv3 --pmc)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrlab_amd as pa
from pbrlab_amd import scenes
n = int(os.environ.get("NRAYS", str(16 << 20)))
desc = scenes.hair_scene(seed=1) if os.environ.get("VARIANT") == "hair" else scenes.cornell_scene("ggx", seed=1)
s = pa.scene_from_desc(desc)
lo, hi = s.FetchSceneAABB()
rays = scenes.random_rays((lo, hi), n, seed=1)
for rep in range(int(os.environ.get("REPS", "3"))):
    t = time.time(); h = s.trace_closest(rays); dt = time.time() - t
    print(f"closest: {n/dt/1e6:.1f} Mrays/s incl. copies ({dt*1e3:.1f} ms), hits {(h['instance_id'] != 0xFFFFFFFF).mean():.3f}")
