"""CPU, world_size 2 over gloo: tile partition + one sum-reduce of the float4 accumulator reproduces the 1-rank
frame bit for bit (the N>1 path of bench.py with the oracle standing in for the device)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, scene_path


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hydracore_amd import HostScene
    from hydracore_amd.multi_gpu import all_reduce_scalar, reduce_accumulator, tile_owner_mask
    from oracle_lib import Oracle
    sc = HostScene(scene_path("test_224"), 64, 48, trace_depth=3, enable_dof=0, use_hip=False)
    orc = Oracle(sc.buffers())
    img, rays, _ = orc.render(2, seed=777, sum_mode=True, rank=rank, world=world, tile=16, threads=2)
    mask = tile_owner_mask(64, 48, rank, world, 16)
    assert (img[~mask] == 0).all() and (img[mask][:, :3].sum() > 0)
    acc = torch.from_numpy(img)
    reduce_accumulator(acc, dst=0)
    total_rays = all_reduce_scalar(rays, "cpu")
    if rank == 0:
        np.save(out_path, acc.numpy())
        np.save(out_path + ".rays.npy", np.array([total_rays]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_partition_matches_single_rank(built, tmp_path):
    from hydracore_amd import HostScene
    from oracle_lib import Oracle
    out = str(tmp_path / "acc.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    sc = HostScene(scene_path("test_224"), 64, 48, trace_depth=3, enable_dof=0, use_hip=False)
    full, rays, _ = Oracle(sc.buffers()).render(2, seed=777, sum_mode=True)
    got = np.load(out)
    assert (got == full).all()
    assert int(np.load(out + ".rays.npy")[0]) == rays
