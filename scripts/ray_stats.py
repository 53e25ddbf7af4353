"""per-ray traversal statistics of one 1920x1080 x SPP render (PBRHIP_RENDER_STATS): nodes / primitive tests per ray, pruned rays"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrlab_amd as pa
from pbrlab_amd import scenes
variant = os.environ.get("VARIANT", "ggx")
desc = scenes.hair_scene(seed=1) if variant == "hair" else scenes.cornell_scene(variant, seed=1)
s = pa.scene_from_desc(desc)
layer = pa.RenderLayer()
ok, st = pa.Render(s, 1920, 1080, int(os.environ.get("SPP", "8")), layer=layer, flags=pa.api.RENDER_STATS)
c, sh = st["closest_rays"], st["shadow_rays"]
print("samples", st["samples"], "closest", c, "shadow", sh, "tail closest/shadow", st["tail_closest_rays"], st["tail_shadow_rays"],
      "pruned", st["pruned_rays"])
print("closest: nodes/ray %.1f tris/ray %.2f curves/ray %.2f" % (st["closest_nodes"] / c, st["closest_tris"] / c, st["closest_curves"] / c))
print("shadow : nodes/ray %.1f tris/ray %.2f curves/ray %.2f" % (st["shadow_nodes"] / max(sh, 1), st["shadow_tris"] / max(sh, 1), st["shadow_curves"] / max(sh, 1)))
