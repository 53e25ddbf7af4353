"""reproduces tests/test_gpu_parity.py::test_triangle_soup_and_ties[seed] and prints the rays that differ"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import _oracle as O
import pbrlab_amd as pa
from pbrlab_amd import scenes
for seed in range(6):
    rng = np.random.RandomState(500 + seed)
    n = 400
    v = (rng.rand(n, 3, 3).astype(np.float32) * 2 - 1)
    v[:, 1:] = v[:, :1] + (v[:, 1:] - v[:, :1]) * np.float32(0.3)
    v[:60] = np.round(v[:60] * 4) / 4
    v[60:100, :, 2] = np.float32(0.25)
    v[100:120] = v[60:80]
    v[120:130, 2] = v[120:130, 1]
    v[130:140, 2] = v[130:140, 0] + (v[130:140, 1] - v[130:140, 0]) * np.float32(1.000001)
    verts = np.concatenate([v.reshape(-1, 3), np.ones((n * 3, 1), np.float32)], 1)
    faces = np.arange(n * 3, dtype=np.uint32).reshape(n, 3)
    mat = dict(scenes.PRINCIPLED_DEFAULTS, kind="principled", name="m")
    desc = scenes.SceneDesc(verts, np.zeros((0, 4), np.float32), [mat],
                            [scenes.Shape("a", faces[:250], None, np.zeros(250, np.uint32)),
                             scenes.Shape("b", faces[250:], None, np.zeros(n - 250, np.uint32))])
    so = O.oracle_scene_from_desc(desc)
    lo, hi = so.FetchSceneAABB()
    rays = scenes.random_rays((lo, hi), 6000, seed=seed)
    extra = np.zeros(2000, O.RAY_DT)
    tgt = verts[rng.randint(len(verts), size=2000), :3]
    org = np.array([0.3, -0.2, 3.0], np.float32)
    extra["org"] = org
    extra["dir"] = tgt - org
    extra[:500]["org"] = tgt[:500] + np.array([0, 0, 2], np.float32)
    extra[:500]["dir"] = (0, 0, -1)
    extra["tmin"], extra["tmax"] = 0.0, 1e30
    rays = np.concatenate([rays, extra])
    ho = so.trace_closest(rays)
    for builder in (pa.api.BVH_HOST_SAH, pa.api.BVH_GPU_LBVH):
        sg = pa.scene_from_desc(desc, bvh_builder=builder)
        for rep in range(3):
            hg = sg.trace_closest(rays)
            bad = np.nonzero((hg["prim_id"] != ho["prim_id"]) | (hg["instance_id"] != ho["instance_id"]) | (hg["t"].view(np.uint32) != ho["t"].view(np.uint32)))[0]
            print("seed", seed, "builder", builder, "rep", rep, "depth", sg.info()["depth"], "bad", len(bad))
            for i in bad[:5]:
                print("   ray", i, "gpu", hg[i]["instance_id"], hg[i]["prim_id"], hg[i]["t"], "oracle", ho[i]["instance_id"], ho[i]["prim_id"], ho[i]["t"])
