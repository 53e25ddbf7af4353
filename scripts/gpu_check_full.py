"""Full-size-geometry parity check (one-off, GPU box): the C3 (545 k triangles, SSS) and C5 (+ 50 k strands = 4.8 M curve
pieces) scenes at a small frame, GPU vs oracle[f64r], host-SAH and GPU-LBVH trees, wavefront and tail modes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
import pbrlab_amd as pa
from pbrlab_amd import scenes
W, H, SPP = 160, 90, 8
for name, desc in (("c3 scene", scenes.cornell_scene("sss", seed=1)), ("c5 scene", scenes.cornell_hair_scene("sss", seed=1))):
    t = time.time(); so = O.oracle_scene_from_desc(desc); print(name, "oracle commit %.1f s" % (time.time() - t), flush=True)
    t = time.time(); rgba, cnt, ost = so.render(W, H, SPP, threads=os.cpu_count(), math_mode=O.MATH_DEVICE); print("  oracle render %.1f s" % (time.time() - t), flush=True)
    lo, hi = so.FetchSceneAABB()
    rays = scenes.random_rays((lo, hi), 200000, seed=9)
    ho = so.trace_closest(rays)
    for builder in (pa.api.BVH_HOST_SAH, pa.api.BVH_GPU_LBVH):
        sg = pa.scene_from_desc(desc, bvh_builder=builder)
        hg = sg.trace_closest(rays)
        ids = all(np.array_equal(hg[f], ho[f]) for f in ("instance_id", "geom_id", "prim_id"))
        tuv = all(np.array_equal(np.ascontiguousarray(hg[f]).view(np.uint32), np.ascontiguousarray(ho[f]).view(np.uint32)) for f in ("t", "u", "v", "normal_g"))
        for tail in (0, 0xFFFFFFFF, 20000):
            layer = pa.RenderLayer()
            pa.Render(sg, W, H, SPP, layer=layer, tail_paths=tail)
            nd = int((layer.rgba.view(np.uint32) != rgba.view(np.uint32)).any(axis=2).sum())
            print(f"  builder {builder} tail {tail}: hit ids equal {ids}, t/u/v/Ng bits equal {tuv}, pixels differing {nd} / {W*H}, count equal {np.array_equal(layer.count, cnt)}", flush=True)
