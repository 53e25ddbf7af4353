"""CPU, world_size 2 over gloo: the multi-GPU exchange step (one framebuffer reduce to rank 0) on
per-rank tile shards.  The shards here come from the oracle (no GPU in this container); the GPU
version of the same check is tests/test_gpu_parity.py::test_tile_sharding_matches_single."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import _oracle as O
from pbrlab_amd import scenes
from pbrlab_amd.dist import reduce_layer, tiles_of_rank
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
d = scenes.cornell_scene("ggx", monkey_subdiv=1, lucy_nu=32, lucy_nv=8)
so = O.oracle_scene_from_desc(d)
W, H, SPP = 130, 70, 2
rgba, cnt, _ = so.render(W, H, SPP, tile_rank=rank, tile_world=world)
mine = tiles_of_rank(W, H, rank, world)
mask = np.zeros((H, W), bool)
for sx, tx, sy, ty in mine: mask[sy:ty, sx:tx] = True
assert (cnt[mask] == SPP).all() and (cnt[~mask] == 0).all() and not rgba[~mask].any()
t_rgba, t_cnt = torch.from_numpy(rgba), torch.from_numpy(cnt.astype(np.int32))
reduce_layer(t_rgba, t_cnt, dst=0)
if rank == 0:
    full, fcnt, _ = so.render(W, H, SPP)
    assert t_rgba.numpy().tobytes() == full.tobytes(), "reduced frame differs from the single-rank frame"
    assert np.array_equal(t_cnt.numpy(), fcnt.astype(np.int32))
    print("DIST_OK")
dist.destroy_process_group()
"""


def test_gloo_world2_reduce(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0]


WORKER_GATHER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import _oracle as O
from pbrlab_amd import scenes
from pbrlab_amd.dist import reduce_layer, gather_layer, shard_pixels
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
d = scenes.cornell_scene("ggx", monkey_subdiv=1, lucy_nu=32, lucy_nv=8)
so = O.oracle_scene_from_desc(d)
W, H, SPP = 150, 90, 2
full, fcnt, _ = so.render(W, H, SPP)
for block in (16, 64, 100):
    # this rank's layer: its blocks of the frame, zeros elsewhere (what pbrhip_render writes for tile_rank / shard_block)
    pix = shard_pixels(W, H, rank, world, block)
    rgba, cnt = np.zeros_like(full), np.zeros_like(fcnt)
    rgba.reshape(-1, 4)[pix] = full.reshape(-1, 4)[pix]
    cnt.reshape(-1)[pix] = fcnt.reshape(-1)[pix]
    assert len(np.unique(pix)) == len(pix)
    for how in (gather_layer, reduce_layer):
        t_rgba, t_cnt = torch.from_numpy(rgba.copy()), torch.from_numpy(cnt.astype(np.int32))
        if how is gather_layer: how(t_rgba, t_cnt, block=block, dst=0)
        else: how(t_rgba, t_cnt, dst=0)
        if rank == 0:
            assert t_rgba.numpy().tobytes() == full.tobytes(), (block, how.__name__)
            assert np.array_equal(t_cnt.numpy(), fcnt.astype(np.int32))
# the ranks' pixel lists partition the frame
all_pix = np.concatenate([shard_pixels(W, H, r, world, 16) for r in range(world)])
assert np.array_equal(np.sort(all_pix), np.arange(W * H))
if rank == 0: print("DIST_OK")
dist.destroy_process_group()
"""


def run_world(tmp_path, text, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(text.format(root=ROOT))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0]


def test_gloo_world2_gather_and_reduce(tmp_path):
    """the two exchanges of the N > 1 path (shard gather = what pbrhip_comm_gather_layer does; full-layer reduce) on the
    block -> rank map the library uses, for several block sizes: both land the one-rank frame on rank 0, bit for bit"""
    run_world(tmp_path, WORKER_GATHER, 2)


def test_gloo_world3_gather_and_reduce(tmp_path):
    run_world(tmp_path, WORKER_GATHER, 3)


def test_shard_pixels_matches_tiles():
    """block 64 = CreateTiles' tiles in CreateTiles' order (render-tile.cc:29-41)"""
    from pbrlab_amd.dist import shard_pixels, tiles_of_rank
    W, H = 200, 136
    for world in (1, 3):
        for r in range(world):
            want = np.concatenate([(np.arange(sy, ty)[:, None] * W + np.arange(sx, tx)[None, :]).reshape(-1)
                                   for sx, tx, sy, ty in tiles_of_rank(W, H, r, world).astype(np.int64)])
            assert np.array_equal(shard_pixels(W, H, r, world, 64), want)


def test_tile_rank_map():
    from pbrlab_amd.dist import tiles_of_rank
    import pbrlab_amd as pa
    all_tiles = pa.create_tiles(1920, 1080)
    got = [tiles_of_rank(1920, 1080, r, 8) for r in range(8)]
    assert sum(len(g) for g in got) == len(all_tiles) == 510
    assert np.array_equal(got[3], all_tiles[3::8])
    assert [len(g) for g in got] == [64, 64, 64, 64, 64, 64, 63, 63]


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` with no launcher (WORLD_SIZE unset) must start N rank processes itself -- torch.distributed.run as a
    child, before anything touches the GPU -- and hand on rank 0's line (it used to end in SystemExit).  Here on the
    CPU: the dry run shows the command, and the self-test runs it for real with two gloo ranks."""
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "2", "--warmup", "1"], env=dict(env, BENCH_SPAWN_DRY_RUN="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["spawn"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd
    assert cmd[cmd.index(bench):] == [bench, "--gpus", "8", "--steps", "2", "--warmup", "1"]
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--spawn-selftest"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["selftest"] == 2 and out["sum"] == 3
    # the per-rank fields an N > 1 bench line carries (bench.rank_diagnostics, gathered over the same process group): rank r reports a
    # render of 10 (r + 1) ms and an exchange of 1 + r ms
    d = out["diagnostics"]
    assert d["per_rank_render_ms"] == [10.0, 20.0] and d["exchange_ms"] == 2.0 and d["exchange_ms_rank0"] == 1.0
    assert abs(d["imbalance"] - 20.0 / 15.0) < 1e-12 and d["slowest_rank"] == 1
