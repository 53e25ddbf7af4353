"""where the lanes of k_shade_principled go: one statistics render with a -DPB_SHADE_PROBE build (PBRHIP_LIB=build/probe/libpbrhip.so)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PBRHIP_PV_STATS"] = "1"
import pbrlab_amd as pa
from pbrlab_amd import scenes
variant = os.environ.get("VARIANT", "ggx")
spp = int(os.environ.get("SPP", "8"))
s = pa.scene_from_desc(scenes.cornell_scene(variant, seed=1))
layer = pa.RenderLayer()
pa.Render(s, 1920, 1080, spp, layer=layer)
ok, st = pa.Render(s, 1920, 1080, spp, layer=layer, flags=pa.api.RENDER_STATS, num_streams=1, tail_paths=0xFFFFFFFF)
