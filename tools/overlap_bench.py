#!/usr/bin/env python3
"""Experiment (GPU box): do two layers on their own HIP streams, each tracing half the image with half the resident
blocks, overlap the latency-bound k_bounce of one with the VALU-bound traversal of the other?  Compares one layer tracing
the whole frame with two layers (tile partition 0/2 and 1/2) whose passes are enqueued back to back.
   HYDRA_HIP_PRIVATE_STREAM=1 python tools/overlap_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from conftest import scene_path
    from hydracore_amd import HostScene
    w, h, spp = 1920, 1080, 64

    def make(rank, world, blocks_trace, blocks_shade):
        sc = HostScene(scene_path("test_224"), w, h, trace_depth=8, enable_dof=0, use_hip=True)
        core = sc.hip()
        core.set_option("samples_in_flight", spp)
        core.set_tile_partition(rank, world, 64)
        core.set_option("trace_blocks_per_cu", blocks_trace)
        core.set_option("shade_blocks_per_cu", blocks_shade)
        sc.draw(1, spp)
        return sc, core

    def run(cores, reps=3):
        for c in cores:
            c.trace_pass(spp)
        for c in cores:
            c.finish()
        for c in cores:
            c.reset_perf_counters()
        t0 = time.perf_counter()
        for _ in range(reps):
            for c in cores:
                c.trace_pass(spp)
        for c in cores:
            c.finish()
        dt = time.perf_counter() - t0
        rays = sum(int(c.rays_stat().extensionRays + c.rays_stat().shadowRays) for c in cores)
        return rays / dt / 1e6, dt / reps * 1e3

    single = make(0, 1, 12, 16)
    print("one layer, whole frame:            %7.0f Mrays/s  %7.2f ms per %d spp" % (*run([single[1]]), spp), flush=True)
    single[0].close()
    for bt, bs in ((12, 16), (6, 8), (5, 6), (8, 4)):
        a, b = make(0, 2, bt, bs), make(1, 2, bt, bs)
        print("two layers, blocks/CU trace %2d shade %2d: %7.0f Mrays/s  %7.2f ms per %d spp" % (bt, bs, *run([a[1], b[1]]), spp), flush=True)
        a[0].close(); b[0].close()


if __name__ == "__main__":
    main()
