"""First-contact GPU check: trace hooks and small renders against the oracle (debug aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import _oracle as O
import pbrlab_amd as pa
from pbrlab_amd import scenes

def cmp_hits(a, b, name):
    ids_ok = (a["instance_id"] == b["instance_id"]) & (a["geom_id"] == b["geom_id"]) & (a["prim_id"] == b["prim_id"])
    bits = lambda x: np.ascontiguousarray(x).view(np.uint32)
    tuv_ok = (bits(a["t"]) == bits(b["t"])) & (bits(a["u"]) == bits(b["u"])) & (bits(a["v"]) == bits(b["v"]))
    ng_ok = (bits(a["normal_g"]) == bits(b["normal_g"])).all(axis=1)
    print(f"{name}: n={len(a)} hits={(a['instance_id']!=0xFFFFFFFF).sum()} ids_mismatch={(~ids_ok).sum()} tuv_mismatch={(~tuv_ok).sum()} ng_mismatch={(~ng_ok).sum()}")
    bad = np.nonzero(~(ids_ok & tuv_ok))[0][:5]
    for i in bad: print("   ", i, a[i], b[i])

def run(name, desc, w=64, h=64, spp=4):
    print("=====", name, "tris", desc.num_triangles(), "segs", desc.num_segments())
    so = O.oracle_scene_from_desc(desc)
    t = time.time(); sg = pa.scene_from_desc(desc); print("gpu commit s", time.time() - t, sg.info())
    print("aabb", so.FetchSceneAABB(), sg.FetchSceneAABB())
    lo, hi = so.FetchSceneAABB()
    rays = scenes.random_rays((lo, hi), 20000, seed=3)
    hg = sg.trace_closest(rays); ho = so.trace_closest(rays)
    cmp_hits(hg, ho, "closest gpu vs oracle-bvh")
    hb = so.trace_closest(rays[:2000], brute_force=True)
    cmp_hits(hg[:2000], hb, "closest gpu vs oracle-brute")
    sr = rays.copy(); sr["tmax"] = 0.7
    og = sg.trace_any(sr); oo = so.trace_any(sr)
    print("any-hit mismatches", (og != oo).sum(), "occluded", og.sum())
    cam = so.camera_rays(w, h, [(x, y) for y in range(0, h, 7) for x in range(0, w, 7)])
    cmp_hits(sg.trace_closest(cam), so.trace_closest(cam), "camera rays")
    layer = pa.RenderLayer()
    t = time.time(); ok, st = pa.Render(sg, w, h, spp, layer=layer, flags=pa.api.RENDER_STATS | pa.api.RENDER_TIMING); tg = time.time() - t
    t = time.time(); rgba, cnt, ost = so.render(w, h, spp, threads=8, math_mode=O.MATH_F64R); to = time.time() - t
    print("gpu s", tg, "oracle s", to)
    print("gpu stats", {k: v for k, v in st.items() if v})
    print("orc stats", ost)
    d = layer.rgba - rgba
    nz = np.abs(d).max(axis=2) > 0
    rel = np.linalg.norm(d[..., :3]) / max(np.linalg.norm(rgba[..., :3]), 1e-30)
    print(f"render: pixels differing {nz.sum()} / {w*h}; rel L2 {rel:.3e}; count equal {np.array_equal(layer.count, cnt)}; mean gpu {layer.rgba[...,:3].mean():.6f} orc {rgba[...,:3].mean():.6f}")
    if nz.sum():
        ys, xs = np.nonzero(nz)
        for y, x in list(zip(ys, xs))[:5]:
            print("   px", x, y, layer.rgba[y, x], rgba[y, x])
    return layer, rgba

if __name__ == "__main__":
    print("devices", pa.device_count())
    which = sys.argv[1:] or ["lambert", "ggx", "sss", "hair"]
    if "lambert" in which: run("lambert", scenes.cornell_scene("lambert", monkey_subdiv=2, lucy_nu=64, lucy_nv=12))
    if "ggx" in which: run("ggx", scenes.cornell_scene("ggx", monkey_subdiv=2, lucy_nu=64, lucy_nv=12))
    if "sss" in which: run("sss", scenes.cornell_scene("sss", monkey_subdiv=2, lucy_nu=64, lucy_nv=12))
    if "hair" in which: run("hair", scenes.hair_scene(n_strands=500, n_segments=6, head_subdiv=2))
