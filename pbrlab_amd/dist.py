"""Multi-GPU plumbing on the Python side: the block -> rank map and the RenderLayer exchange over torch.distributed
(backend "nccl" is RCCL over xGMI on the GPU node, "gloo" in the CPU tests).  SURVEY.md §8e.

Pixels are independent, so there is no exchange while rendering: rank r renders the pixel blocks with
index % world == r (interleaved: object blocks cost far more than wall blocks) into a zero-initialised full-size
framebuffer and the layers are combined on rank 0.  Shards are disjoint and everything else is +0.0f, so the result is
exact: the N-GPU image is bit-identical to the 1-GPU image.

The product path for the exchange is inside libpbrhip (pbrhip_comm_gather_layer / pbrhip_comm_reduce_layer, RCCL called
from C++; `api.Comm`).  This module holds the same two exchanges written against torch.distributed: `bench.py --exchange
torch` uses reduce_layer, and the world-size-N gloo tests run both here on the CPU with the very pixel lists the library
uses (shard_pixels == pbrhip.cpp::shard_pixels)."""
import numpy as np

from . import api


def tiles_of_rank(width, height, rank, world):
    """CreateTiles order (src/render-tile.cc:29-41), every world-th tile starting at rank."""
    return api.create_tiles(width, height)[rank::world]


def shard_pixels(width, height, rank, world, block=64):
    """Pixel indices y * width + x of the block x block pixel blocks dealt to `rank` (row-major block order, block i ->
    rank i % world; block 64 = the reference's tiles), block by block, rows inside a block: the order in which the
    library lists a rank's pixels (pbrhip.cpp::shard_pixels) and packs its shard."""
    block = block or 64
    out = []
    t = 0
    for by in range(0, height, block):
        for bx in range(0, width, block):
            if t % world == rank:
                ys = np.arange(by, min(by + block, height), dtype=np.int64)[:, None]
                xs = np.arange(bx, min(bx + block, width), dtype=np.int64)[None, :]
                out.append((ys * width + xs).reshape(-1))
            t += 1
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def reduce_layer(rgba, count, dst=0):
    """sum-reduce the RenderLayer tensors to rank `dst` (in place on dst): SURVEY 8e's collective."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(rgba, dst=dst, op=dist.ReduceOp.SUM)
        dist.reduce(count, dst=dst, op=dist.ReduceOp.SUM)
    return rgba, count


def gather_layer(rgba, count, block=64, dst=0):
    """The same result as reduce_layer for layers that are zero outside the rank's own blocks: every rank sends only
    the pixels of its blocks (20 bytes per pixel) to `dst`, which adds them into its layer (what
    pbrhip_comm_gather_layer does with ncclSend / ncclRecv)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return rgba, count
    rank, world = dist.get_rank(), dist.get_world_size()
    H, W = count.shape
    flat_rgba, flat_count = rgba.view(-1, 4), count.view(-1)
    if rank != dst:
        pix = torch.from_numpy(shard_pixels(W, H, rank, world, block)).to(rgba.device)
        dist.send(flat_rgba[pix].contiguous(), dst=dst)
        dist.send(flat_count[pix].contiguous(), dst=dst)
        return rgba, count
    for r in range(world):
        if r == dst:
            continue
        pix = torch.from_numpy(shard_pixels(W, H, r, world, block)).to(rgba.device)
        buf = torch.empty((len(pix), 4), dtype=rgba.dtype, device=rgba.device)
        cbuf = torch.empty((len(pix),), dtype=count.dtype, device=count.device)
        dist.recv(buf, src=r)
        dist.recv(cbuf, src=r)
        flat_rgba[pix] += buf
        flat_count[pix] += cbuf
    return rgba, count
