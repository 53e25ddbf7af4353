"""Multi-GPU plumbing: tile -> rank map and the one framebuffer reduce (torch.distributed; backend
"nccl" is RCCL over xGMI on the GPU node, "gloo" in the CPU tests).  SURVEY.md §8e.

Pixels are independent, so there is no exchange while rendering: rank r renders the 64x64 tiles with
index % world == r (interleaved: object tiles cost far more than wall tiles) into a zero-initialised
full-size framebuffer; one reduce(sum) of rgba (W*H*4 f32) and count (W*H i32) to rank 0 then yields
the frame.  Shards are disjoint and everything else is +0.0f, so the sum is exact: the N-GPU image is
bit-identical to the 1-GPU image."""
import numpy as np

from . import api


def tiles_of_rank(width, height, rank, world):
    """CreateTiles order (src/render-tile.cc:29-41), every world-th tile starting at rank."""
    return api.create_tiles(width, height)[rank::world]


def reduce_layer(rgba, count, dst=0):
    """sum-reduce the RenderLayer tensors to rank `dst` (in place on dst)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(rgba, dst=dst, op=dist.ReduceOp.SUM)
        dist.reduce(count, dst=dst, op=dist.ReduceOp.SUM)
    return rgba, count
