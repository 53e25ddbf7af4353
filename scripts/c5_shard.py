"""BASELINE configs[4] (3840x2160 x 1024 spp, S-cornell SSS + S-hair) on ONE GPU: the whole frame vs rank 0's share of an
8-rank run (16 x 16 pixel blocks i % 8 == 0, as bench.py --gpus 8 deals them) -- the single-GPU proxy for strong scaling of the configuration the multi-GPU target names."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbrlab_amd as pa
from pbrlab_amd import scenes, api
desc = scenes.cornell_hair_scene("sss", seed=1)
s = pa.scene_from_desc(desc)
W, H = 3840, 2160
SPP = int(os.environ.get("SPP", "1024"))
rgba = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"); cnt = torch.zeros((H, W), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
# warm-up: pixel lists built and the working set allocated at its final size (a chunk of the eighth holds 258 passes = 267.5 M
# paths, one of the whole frame 32 passes = 265.4 M: the first render that needs the larger one re-allocates 65 GB of path state)
api.Render(s, W, H, min(SPP, 300), tile_rank=0, tile_world=8, device_out=(rgba.data_ptr(), cnt.data_ptr()), shard_block=16)
api.Render(s, W, H, 32, device_out=(rgba.data_ptr(), cnt.data_ptr()))
res = {}
for world in (8, 1):
    t = time.perf_counter()
    _, st = api.Render(s, W, H, SPP, tile_rank=0, tile_world=world, device_out=(rgba.data_ptr(), cnt.data_ptr()), shard_block=16 if world > 1 else 0)
    dt = time.perf_counter() - t
    res[world] = dt
    print(f"world {world}: {dt*1e3:.0f} ms, {st['samples']/dt/1e6:.0f} Msamples/s on this GPU, chunks {st['chunks']}, iterations {st['iterations']}", flush=True)
print(f"strong-scaling proxy at 8 ranks: {res[1]/res[8]:.2f}x")
