"""frame time of a benchmark frame under environment settings: python scripts/frame_ab.py "NAME=VAL,..." "NAME=VAL" ... (VARIANT, SPP, WORLD -- render rank 0's share of WORLD ranks -- from the environment)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrlab_amd as pa
from pbrlab_amd import scenes
variant = os.environ.get("VARIANT", "ggx")
spp = int(os.environ.get("SPP", "64"))
world = int(os.environ.get("WORLD", "1"))
desc = scenes.hair_scene(seed=1) if variant == "hair" else scenes.cornell_scene(variant, seed=1)
s = pa.scene_from_desc(desc)
layer = pa.RenderLayer()
pa.Render(s, 1920, 1080, spp, layer=layer, tile_world=world)
ref = layer.rgba.copy()
for rep in range(int(os.environ.get("REPS", "3"))):
    for cfg in sys.argv[1:]:
        kv = [x.split("=", 1) for x in cfg.split(",") if "=" in x]
        for k, v in kv: os.environ[k] = v
        best = min(pa.Render(s, 1920, 1080, spp, layer=layer, tile_world=world)[1]["ms_total"] for _ in range(4))
        assert layer.rgba.tobytes() == ref.tobytes(), cfg
        for k, v in kv: os.environ.pop(k, None)
        print(f"{variant} {spp} spp 1/{world} [{cfg or 'default'}]: {best:.2f} ms", flush=True)
