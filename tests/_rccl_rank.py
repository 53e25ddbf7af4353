"""One rank of the multi-rank RCCL test (tests/test_gpu_multirank.py starts `world` of these, each a fresh process on its
own device): renders its pixel blocks of a small frame into device buffers, then runs the library's two exchanges --
pbrhip_comm_gather_layer (ncclSend / ncclRecv of the packed shards) and pbrhip_comm_reduce_layer (ncclReduce of the
layers) -- and rank 0 writes both frames.

usage: _rccl_rank.py <rank> <world> <id file> <out .npz>      (rank 0 creates the id file)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import pbrlab_amd as pa  # noqa: E402
from pbrlab_amd import api, scenes  # noqa: E402

W, H, SPP, BLOCK = 200, 120, 3, 16


def main():
    rank, world, id_file, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    torch.cuda.set_device(rank)
    pa.set_device(rank)
    dev = torch.device("cuda", rank)
    if rank == 0:
        uid = pa.Comm.unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 120:
                raise SystemExit("no communicator id after 120 s")
            time.sleep(0.05)
        uid = open(id_file, "rb").read()
    comm = pa.Comm(uid, rank, world)          # ncclCommInitRank: collective over the ranks
    scene = pa.scene_from_desc(scenes.cornell_hair_scene("sss", n_strands=200, n_segments=5, monkey_subdiv=2, lucy_nu=64, lucy_nv=12))
    frames = {}
    for exchange in ("gather", "reduce"):
        rgba = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        count = torch.zeros((H, W), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ptrs = (rgba.data_ptr(), count.data_ptr())
        api.Render(scene, W, H, SPP, tile_rank=rank, tile_world=world, device_out=ptrs, shard_block=BLOCK)
        if exchange == "gather":
            comm.gather_layer(scene, W, H, ptrs[0], ptrs[1], shard_block=BLOCK, root=0)
        else:
            comm.reduce_layer(ptrs[0], ptrs[1], W * H, root=0)
        torch.cuda.synchronize()
        frames[exchange + "_rgba"] = rgba.cpu().numpy()
        frames[exchange + "_count"] = count.cpu().numpy()
    comm.close()
    if rank == 0:
        np.savez(out, **frames)
    print(f"rank {rank} of {world} done", flush=True)


if __name__ == "__main__":
    main()
