"""per-kernel split of the C5 scene (S-cornell SSS + S-hair, 3840x2160) at SPP (env, default 64), one path group"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbrlab_amd as pa
from pbrlab_amd import scenes, api
desc = scenes.cornell_hair_scene("sss", seed=1)
s = pa.scene_from_desc(desc)
W, H, SPP = 3840, 2160, int(os.environ.get("SPP", "64"))
rgba = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"); cnt = torch.zeros((H, W), dtype=torch.int32, device="cuda")
api.Render(s, W, H, 8, device_out=(rgba.data_ptr(), cnt.data_ptr()))
_, st = api.Render(s, W, H, SPP, device_out=(rgba.data_ptr(), cnt.data_ptr()), flags=api.RENDER_TIMING, num_streams=1)
print({k: round(v, 1) for k, v in st.items() if k.startswith("ms_")}, "iterations", st["iterations"], "chunks", st["chunks"])
_, st = api.Render(s, W, H, SPP, device_out=(rgba.data_ptr(), cnt.data_ptr()), flags=api.RENDER_STATS, num_streams=1)
c, sh = st["closest_rays"], st["shadow_rays"]
print("rays closest %d shadow %d tail %d/%d pruned %d | closest: nodes/ray %.1f tris %.2f curves %.2f | shadow: nodes %.1f tris %.2f curves %.2f" % (
    c, sh, st["tail_closest_rays"], st["tail_shadow_rays"], st["pruned_rays"], st["closest_nodes"] / c, st["closest_tris"] / c, st["closest_curves"] / c,
    st["shadow_nodes"] / sh, st["shadow_tris"] / sh, st["shadow_curves"] / sh))
