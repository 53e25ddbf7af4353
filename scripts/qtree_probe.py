"""per-ray traversal statistics + kernel time of one render on the Q tree and on the binary tree (PBRHIP_WIDE=0)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrlab_amd as pa
from pbrlab_amd import scenes
variant = os.environ.get("VARIANT", "ggx")
spp = int(os.environ.get("SPP", "8"))
desc = scenes.hair_scene(seed=1) if variant == "hair" else (scenes.cornell_hair_scene("sss", seed=1) if variant == "c5" else scenes.cornell_scene(variant, seed=1))
os.environ["PBRHIP_DEBUG"] = "1"
s = pa.scene_from_desc(desc)
W, H = (3840, 2160) if variant == "c5" else (1920, 1080)
for wide in ("1", "0"):
    os.environ["PBRHIP_WIDE"] = wide
    layer = pa.RenderLayer()
    pa.Render(s, W, H, spp, layer=layer)
    ok, tm = pa.Render(s, W, H, spp, layer=layer, flags=pa.api.RENDER_TIMING, num_streams=1)
    ok, st = pa.Render(s, W, H, spp, layer=layer, flags=pa.api.RENDER_STATS, num_streams=1)
    c, sh = st["closest_rays"], st["shadow_rays"]
    print(f"WIDE={wide} {variant} {spp} spp: frame {tm['ms_total']:.1f} ms, k_trace {tm['ms_trace_closest']:.1f} ms in {tm['n_trace_closest']} launches, "
          f"tail {tm['ms_tail']:.1f}, walk+step {tm['ms_sss_step']:.1f}")
    print("  closest: nodes/ray %.2f tris/ray %.2f curves/ray %.2f" % (st["closest_nodes"] / c, st["closest_tris"] / c, st["closest_curves"] / c))
    print("  shadow : nodes/ray %.2f tris/ray %.2f curves/ray %.2f" % (st["shadow_nodes"] / max(sh, 1), st["shadow_tris"] / max(sh, 1), st["shadow_curves"] / max(sh, 1)), flush=True)
