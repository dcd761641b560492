"""Image-plane tile partition across the GPUs of one node + the single RCCL exchange of the float4 accumulator
(SURVEY.md 8e).  One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

Tiles are ordered along the Morton curve of their tile coordinates and dealt round-robin: rank i % world owns the i-th
tile (the same rule hydra_hip_set_tile_partition applies on the device).  Every rank renders only its tiles into a
zero-initialised full-frame accumulator, so the frames of different ranks have disjoint supports and
one sum-reduce to rank 0 assembles the image exactly (x + 0 = x): the N-GPU image is bit-identical to the 1-GPU image.
Reference precedent: N processes adding whole frames into one shared-memory image (hydra_drv/GPUOCLLayerOther.cpp:365-429).
"""
import numpy as np


def _morton2(x, y):
    def spread(v):
        v = v.astype(np.uint32) & 0xffff
        v = (v | (v << 8)) & 0x00ff00ff
        v = (v | (v << 4)) & 0x0f0f0f0f
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return spread(x) | (spread(y) << 1)


def tile_owner_table(width, height, world, tile=64):
    """int32 [tilesY, tilesX]: tiles ordered along the Morton curve of (tx, ty), the i-th going to rank i % world
    (the rule of hydra_hip_set_tile_partition, restated in numpy; tests compare it with hydra_hip_tile_owners)"""
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    ys, xs = np.mgrid[0:ty, 0:tx]
    code = _morton2(xs.ravel(), ys.ravel())
    order = np.argsort(code, kind="stable")
    owner = np.empty(tx * ty, np.int32)
    owner[order] = np.arange(tx * ty, dtype=np.int32) % max(world, 1)
    return owner.reshape(ty, tx)


def tile_owner_mask(width, height, rank, world, tile=64):
    """boolean [height, width] mask of the pixels rank owns"""
    if world <= 1:
        return np.ones((height, width), bool)
    owner = tile_owner_table(width, height, world, tile)
    ys, xs = np.mgrid[0:height, 0:width]
    return owner[ys // tile, xs // tile] == rank


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


def native_comm_init(core, rank, world, device):
    """Bootstrap the HIP layer's own RCCL communicator (hydra_hip_comm_*): rank 0 draws the ncclUniqueId through the C-ABI and
    the 128 bytes travel over the torch.distributed group that already exists (any other channel would do: a C++ host uses a
    socket or a file).  Collective: every rank must call it."""
    import torch
    import torch.distributed as dist
    ident = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        ident.copy_(torch.from_numpy(core.comm_unique_id()).to(device))
    dist.broadcast(ident, src=0)
    core.comm_init(ident.cpu().numpy(), rank, world)
