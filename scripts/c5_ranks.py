import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbrlab_amd as pa
from pbrlab_amd import scenes, api
desc = scenes.cornell_hair_scene("sss", seed=1)
s = pa.scene_from_desc(desc)
W, H = 3840, 2160
SPP = int(os.environ.get("SPP", "256"))
rgba = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"); cnt = torch.zeros((H, W), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for block in (64, 32, 16):
    ts = []
    for rank in range(8):
        t = time.perf_counter()
        _, st = api.Render(s, W, H, SPP, tile_rank=rank, tile_world=8, device_out=(rgba.data_ptr(), cnt.data_ptr()), shard_block=block)
        ts.append((time.perf_counter() - t) * 1e3)
    print(f"block {block}: ranks {' '.join('%.0f' % t for t in ts)} ms | max {max(ts):.0f} mean {sum(ts)/8:.0f}", flush=True)
t = time.perf_counter()
api.Render(s, W, H, SPP, device_out=(rgba.data_ptr(), cnt.data_ptr()))
print(f"whole frame {1e3*(time.perf_counter()-t):.0f} ms")
