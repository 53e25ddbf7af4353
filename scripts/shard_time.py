"""time of one rank's share of the C2 frame at world sizes 1,2,4,8 (proxy for multi-GPU strong scaling on one GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbrlab_amd as pa
from pbrlab_amd import scenes, api
desc = scenes.cornell_scene("ggx", seed=1)
s = pa.scene_from_desc(desc)
W, H, SPP = 1920, 1080, 64
rgba = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"); cnt = torch.zeros((H, W), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for world in (1, 2, 4, 8):
    ts = []
    for rank in range(min(world, 8)):
        for rep in range(2):
            t = time.perf_counter()
            _, st = api.Render(s, W, H, SPP, tile_rank=rank, tile_world=world, device_out=(rgba.data_ptr(), cnt.data_ptr()))
            dt = time.perf_counter() - t
        ts.append(dt * 1e3)
    print(f"world {world}: per-rank ms {[round(x,1) for x in ts]} -> max {max(ts):.1f} ms, speedup vs 1: {base/max(ts):.2f}x, iterations {st['iterations']}" if world > 1 else f"world 1: {ts[0]:.1f} ms, iterations {st['iterations']}")
    if world == 1: base = ts[0]
_, st = api.Render(s, W, H, SPP, tile_rank=0, tile_world=8, device_out=(rgba.data_ptr(), cnt.data_ptr()), flags=api.RENDER_TIMING)
print({k: round(v, 2) for k, v in st.items() if k.startswith("ms_")})
# weak-scaling proxy (what bench.py --gpus N runs per rank): rank 0's tiles of an N-rank job at 64*N spp
for world in (2, 4, 8):
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        _, st = api.Render(s, W, H, SPP * world, tile_rank=0, tile_world=world, device_out=(rgba.data_ptr(), cnt.data_ptr()))
        best = min(best, (time.perf_counter() - t) * 1e3)
    print(f"weak proxy world {world}: rank 0 renders its tiles at {SPP * world} spp in {best:.1f} ms (1-GPU frame {base:.1f} ms)")
# load balance of the weak-scaling job for different shard blocks: per-rank ms of the 8-rank job at 512 spp
for block in (64, 32, 16, 8):
    ts = []
    for rank in range(8):
        best = 1e9
        for rep in range(2):
            t = time.perf_counter()
            api.Render(s, W, H, SPP * 8, tile_rank=rank, tile_world=8, device_out=(rgba.data_ptr(), cnt.data_ptr()), shard_block=block)
            best = min(best, (time.perf_counter() - t) * 1e3)
        ts.append(best)
    print(f"shard_block {block}: per-rank ms {[round(x,1) for x in ts]} max {max(ts):.1f} mean {sum(ts)/8:.1f} (imbalance {max(ts)/(sum(ts)/8)-1:.1%})")
