"""per-ray traversal statistics + kernel times of one render on the O tree (8-wide, default) and on the Q tree (PBRHIP_WIDE8=0); VARIANT=ggx|sss, SPP"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrlab_amd as pa
from pbrlab_amd import scenes
import numpy as np
variant = os.environ.get("VARIANT", "ggx")
spp = int(os.environ.get("SPP", "8"))
desc = scenes.hair_scene(seed=1) if variant == "hair" else (scenes.cornell_hair_scene("sss", seed=1) if variant == "c5" else scenes.cornell_scene(variant, seed=1))
os.environ["PBRHIP_DEBUG"] = "1"
s = pa.scene_from_desc(desc)
os.environ.pop("PBRHIP_DEBUG")
W, H = (3840, 2160) if variant == "c5" else (1920, 1080)
imgs = {}
for wide8 in ("1", "0"):
    os.environ["PBRHIP_WIDE8"] = wide8
    layer = pa.RenderLayer()
    pa.Render(s, W, H, spp, layer=layer)
    imgs[wide8] = layer.rgba.copy()
    best = None
    for _ in range(3):
        ok, tm = pa.Render(s, W, H, spp, layer=layer, flags=pa.api.RENDER_TIMING, num_streams=1)
        if best is None or tm["ms_trace_closest"] < best["ms_trace_closest"]:
            best = tm
    tm = best
    ok, st = pa.Render(s, W, H, spp, layer=layer, flags=pa.api.RENDER_STATS, num_streams=1)
    c, sh = st["closest_rays"], st["shadow_rays"]
    print(f"WIDE8={wide8} {variant} {spp} spp: frame {tm['ms_total']:.2f} ms, k_trace {tm['ms_trace_closest']:.2f} ms in {tm['n_trace_closest']} launches, "
          f"tail {tm['ms_tail']:.2f}, walk+step {tm['ms_sss_step']:.2f}, shade {tm['ms_shade_principled']:.2f}; node bytes {st['node_bytes']}")
    print("  closest: nodes/ray %.2f tris/ray %.2f curves/ray %.2f" % (st["closest_nodes"] / c, st["closest_tris"] / c, st["closest_curves"] / c))
    print("  shadow : nodes/ray %.2f tris/ray %.2f curves/ray %.2f" % (st["shadow_nodes"] / max(sh, 1), st["shadow_tris"] / max(sh, 1), st["shadow_curves"] / max(sh, 1)), flush=True)
print("images identical:", imgs["1"].tobytes() == imgs["0"].tobytes())
