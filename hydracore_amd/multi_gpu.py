"""Image-plane tile partition across the GPUs of one node + the single RCCL exchange of the float4 accumulator
(SURVEY.md 8e).  One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

Every rank renders only the tiles t with t % world == rank (the same rule hydra_hip_set_tile_partition applies on the
device), into a zero-initialised full-frame accumulator, so the frames of different ranks have disjoint supports and
one sum-reduce to rank 0 assembles the image exactly (x + 0 = x): the N-GPU image is bit-identical to the 1-GPU image.
Reference precedent: N processes adding whole frames into one shared-memory image (hydra_drv/GPUOCLLayerOther.cpp:365-429).
"""
import numpy as np


def tile_owner_mask(width, height, rank, world, tile=64):
    """boolean [height, width] mask of the pixels rank owns"""
    ys, xs = np.mgrid[0:height, 0:width]
    tiles_x = (width + tile - 1) // tile
    t = (ys // tile) * tiles_x + (xs // tile)
    return (t % world) == rank if world > 1 else np.ones((height, width), bool)


def reduce_accumulator(accum, dst=0):
    """sum-reduce the full-frame float4 accumulator (torch tensor, any device) to rank dst; no-op for world 1"""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return accum
    dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def all_reduce_scalar(value, device):
    """sum a python number over ranks (ray counters, times use MAX separately)"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_reduce_max(value, device):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
