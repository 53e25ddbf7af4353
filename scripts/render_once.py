"""renders of the C2 scene (for PMC passes / tuning): 1920x1080 x SPP (env, default 1); REPS repeats, prints the best"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrlab_amd as pa
from pbrlab_amd import scenes
variant = os.environ.get("VARIANT", "ggx")
desc = scenes.hair_scene(seed=1) if variant == "hair" else scenes.cornell_scene(variant, seed=1)
s = pa.scene_from_desc(desc)
layer = pa.RenderLayer()
best = None
for _ in range(int(os.environ.get("REPS", "1"))):
    ok, st = pa.Render(s, 1920, 1080, int(os.environ.get("SPP", "1")), layer=layer, tile_world=int(os.environ.get("WORLD", "1")))
    if best is None or st["ms_total"] < best["ms_total"]:
        best = st
print(round(best["ms_total"], 2), best["iterations"])
