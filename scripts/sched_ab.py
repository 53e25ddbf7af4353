"""A/B of the path-group schedule (pbrhip.cpp::plan_groups) on one GPU: the whole C2 frame (world 1) and rank 0's share
of an 8-rank job (16 x 16 blocks), best of REPS renders each.  Configurations = environment overrides the library reads
per render.  usage: python scripts/sched_ab.py [variant]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbrlab_amd as pa
from pbrlab_amd import scenes, api

variant = sys.argv[1] if len(sys.argv) > 1 else "ggx"
spp = {"ggx": 64, "sss": 256}.get(variant, 128)
desc = scenes.hair_scene(seed=1) if variant == "hair" else scenes.cornell_scene(variant, seed=1)
s = pa.scene_from_desc(desc)
W, H = 1920, 1080
rgba = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"); cnt = torch.zeros((H, W), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
REPS = int(os.environ.get("REPS", "3"))
KEYS = ("PBRHIP_WIDE", "PBRHIP_WIDE_WALK", "PBRHIP_RAYS_PER_WAVE", "PBRHIP_GROUPS", "PBRHIP_WINDOW", "PBRHIP_BULK_DIV", "PBRHIP_GROUP_MIN_PATHS", "PBRHIP_STREAMS", "PBRHIP_TAIL_PATHS", "PBRHIP_QUAD_RAYS", "PBRHIP_PIXEL_TILE", "PBRHIP_TRACE_BLOCKS_SMALL", "PBRHIP_TRACE_BLOCKS", "PBRHIP_RETIRE",
        "PBRHIP_SUSP_TURNS", "PBRHIP_PIPE_DEPTH", "PBRHIP_PIPE_DEPTH_SMALL", "PBRHIP_PIPE_STOP", "PBRHIP_SHADOW_FIRST", "PBRHIP_PATCH_SHUFFLE", "PBRHIP_DIRECT")
CONFIGS = [
    {"PBRHIP_STREAMS": "1"},
    {"PBRHIP_STREAMS": "2"},                                 # round 1's default for big chunks
    {"PBRHIP_GROUPS": "62,2"},
    {"PBRHIP_GROUPS": "60,4"},
    {"PBRHIP_GROUPS": "56,8"},
    {"PBRHIP_GROUPS": "48,16"},
    {"PBRHIP_GROUPS": "32,32"},
    {"PBRHIP_GROUPS": "48,12,4"},
    {"PBRHIP_GROUPS": "56,6,2"},
    {"PBRHIP_GROUPS": "60,3,1"},
    {"PBRHIP_GROUPS": "56,8", "PBRHIP_BULK_DIV": "8"},
    {"PBRHIP_GROUPS": "56,8", "PBRHIP_BULK_DIV": "3"},
    {"PBRHIP_GROUPS": "56,8", "PBRHIP_TAIL_PATHS": "1048576"},
    {"PBRHIP_GROUPS": "56,8", "PBRHIP_TAIL_PATHS": "65536"},
    {"PBRHIP_STREAMS": "1", "PBRHIP_TAIL_PATHS": "1048576"},
    {"PBRHIP_STREAMS": "1", "PBRHIP_TAIL_PATHS": "65536"},
]
if os.environ.get("SCHED_CONFIGS"):
    import json
    CONFIGS = json.loads(os.environ["SCHED_CONFIGS"])


SUMS = {}


def best(world, rank=0, **kw):
    t_best, st_best = 1e9, None
    for _ in range(REPS):
        t = time.perf_counter()
        _, st = api.Render(s, W, H, spp, tile_rank=rank, tile_world=world, device_out=(rgba.data_ptr(), cnt.data_ptr()),
                           shard_block=16 if world > 1 else 0, **kw)
        dt = (time.perf_counter() - t) * 1e3
        if dt < t_best:
            t_best, st_best = dt, st
    # the frame must not depend on the schedule: a checksum of its bits, compared with the first configuration's
    chk = (int(rgba.view(torch.int32).to(torch.int64).sum().item()), int(cnt.to(torch.int64).sum().item()))
    if SUMS.setdefault(world, chk) != chk:
        print(f"!! world {world}: the frame differs from the first configuration's ({chk} vs {SUMS[world]})", flush=True)
    return t_best, st_best


best(1)
for cfg in CONFIGS:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(cfg)
    t1, st1 = best(1)
    t8, st8 = best(8)
    t4, _ = best(4)
    t2, _ = best(2)
    _, sts = api.Render(s, W, H, spp, tile_rank=0, tile_world=8, device_out=(rgba.data_ptr(), cnt.data_ptr()), shard_block=16, flags=api.RENDER_STATS)
    print(f"{str(cfg):90s} world1 {t1:6.2f} ms ({st1['iterations']:3d} it)  1/2 {t2:6.2f} ({t1 / t2:.2f}x)  1/4 {t4:6.2f} ({t1 / t4:.2f}x)  "
          f"1/8 {t8:6.2f} ms ({st8['iterations']:3d} it, {t1 / t8:.2f}x; {sts['suspended_rays']} of {sts['closest_rays']} closest-hit rays suspended once)", flush=True)

# where the time of rank 0's eighth goes (one group), and its timeline
for k in KEYS:
    os.environ.pop(k, None)
os.environ["PBRHIP_STREAMS"] = "1"
_, st = api.Render(s, W, H, spp, tile_rank=0, tile_world=8, device_out=(rgba.data_ptr(), cnt.data_ptr()), shard_block=16, flags=api.RENDER_TIMING)
print("1/8, one group, per-kernel ms:", {k: round(v, 2) for k, v in st.items() if k.startswith("ms_")})
os.environ["PBRHIP_TRACE_SCHED"] = "1"
api.Render(s, W, H, spp, tile_rank=0, tile_world=8, device_out=(rgba.data_ptr(), cnt.data_ptr()), shard_block=16)
os.environ["PBRHIP_GROUPS"] = "56,8"
os.environ.pop("PBRHIP_STREAMS")
api.Render(s, W, H, spp, tile_rank=0, tile_world=8, device_out=(rgba.data_ptr(), cnt.data_ptr()), shard_block=16)
