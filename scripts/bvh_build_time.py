"""commit (BVH build + upload) time and render time, host binned-SAH vs GPU linear BVH, on the C2 and C4 scenes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrlab_amd as pa
from pbrlab_amd import scenes, api
for name in ("c2", "c4"):
    desc = scenes.cornell_scene("ggx", seed=1) if name == "c2" else scenes.hair_scene(seed=1)
    for builder, tag in ((api.BVH_HOST_SAH, "host SAH"), (api.BVH_GPU_LBVH, "GPU LBVH")):
        best = 1e9
        for rep in range(2):
            s = pa.Scene()
            s.SetBvhBuilder(builder)
            import numpy as np
            t0 = time.perf_counter()
            real_commit = s.CommitScene
            dt = [0.0]
            def timed():
                t = time.perf_counter(); real_commit(); dt[0] = time.perf_counter() - t
            s.CommitScene = timed
            scenes.build_scene(s, desc, pa.make_principled, pa.make_hair)
            best = min(best, dt[0])
        info = s.info()
        layer = pa.RenderLayer()
        spp = 16
        pa.Render(s, 1920, 1080, spp, layer=layer)
        ok, st = pa.Render(s, 1920, 1080, spp, layer=layer)
        print(f"{name} {tag}: commit {best*1e3:.0f} ms, nodes {info['num_nodes']}, depth {info['depth']}, render {spp} spp {st['ms_total']:.1f} ms")
