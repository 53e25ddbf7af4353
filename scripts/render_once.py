"""one small render of the C2 scene (for PMC passes): 1920x1080 x SPP (env, default 1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrlab_amd as pa
from pbrlab_amd import scenes
variant = os.environ.get("VARIANT", "ggx")
desc = scenes.hair_scene(seed=1) if variant == "hair" else scenes.cornell_scene(variant, seed=1)
s = pa.scene_from_desc(desc)
layer = pa.RenderLayer()
ok, st = pa.Render(s, 1920, 1080, int(os.environ.get("SPP", "1")), layer=layer, tile_world=int(os.environ.get("WORLD", "1")))
print(st["ms_total"], st["iterations"])
