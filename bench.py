#!/usr/bin/env python3
"""bench.py -- Mrays/s of the MI355X wavefront path tracer on BASELINE.json configs[1]:
Cornell-box-style test scene (tests/golden/scenes/test_224 = the reference's hydra_app/tests/test_224), 1920x1080,
8 bounces, PT integrator; 256 spp = 4 steps x 64 spp by default.

A "step" = one pass of the hot path: `--spp-per-step` samples for every pixel this rank owns, all in flight at once
(ray generation, then per bounce: closest-hit traversal, the fused hit/emission/light-sample/BSDF kernel with
compaction, shadow traversal; finally accumulate).  With N > 1
GPUs the image plane is tile-partitioned (weak work per step is fixed per pixel, the frame is split => "strong"
scaling of one frame) and the float4 accumulator is reduced once over RCCL at the end of the timed region.

Prints ONE JSON line (rank 0).  Rays are counted exactly on the device (every extension and shadow ray traced).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def pmc_traffic(workload_key):
    """HBM bytes per k_trace launch from the committed rocprofv3 --pmc passes (profiles/*/pmc_traffic.json), or None"""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload_key") == workload_key:
            best = d.get("k_trace_hbm_bytes_per_launch")
    return best


HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def traversal_bytes(counters):
    """Algorithmic bytes of the traversal kernels: per ray 36 B in (pos, dir, flags/count) + 16 B out,
    128 B per quad visited, 128 B per instance quad entered, 16 B per leaf header, 48 B per triangle tested.
    counters: uint64 [depth, 2 (closest|shadow), 5 (rays, quads, insts, leaves, tris)] -> bytes [depth, 2]"""
    c = counters.astype("float64")
    return c[..., 0] * (36 + 16) + c[..., 1] * 128 + c[..., 2] * 128 + c[..., 3] * 16 + c[..., 4] * 48


def cpu_baseline(scene, depth, budget_s=15.0):
    """the CPU oracle (kind "port") on a bounded sample of the same workload: a 320x180 frame of the same scene and
    depth, as many spp as fit the budget, OpenMP over all host cores"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hydracore_amd import HostScene
    from oracle_lib import Oracle
    w, h = 320, 180
    sc = HostScene(scene, w, h, trace_depth=depth, enable_dof=0, use_hip=False)
    orc = Oracle(sc.buffers())
    gens = orc.init_generators(777)
    img, _, gens = orc.render(1, gens=gens)          # warm-up pass (page-in, thread pool)
    spp, rays, done = 0, 0, 1
    t0 = time.time()
    while time.time() - t0 < budget_s and spp < 4096:
        img, r, gens = orc.render(8, gens=gens, image=img, spp_done=done)
        spp += 8
        done += 8
        rays += r
    dt = time.time() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": orc.max_threads(), "kind": "port",
            "sample": "%dx%d, %d bounces, %d spp of the same scene (%d rays, %.1f s); CPU oracle, own BVH4 walk (the stock "
                      "reference CPU layer traces through Embree 2.17, absent here)" % (w, h, depth, spp, rays, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=64, help="samples per pixel per step, all in flight at once (64 x 1080p = 133 M paths, 30 GB of path state)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--trace-depth", type=int, default=8)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--scene", default=os.path.join(ROOT, "tests", "golden", "scenes", "test_224"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal of N ranks on one GPU)")
    ap.add_argument("--device", type=int, default=-1, help="force this HIP device for every rank (rehearsal only)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from hydracore_amd import HostScene
    from hydracore_amd.multi_gpu import all_reduce_max, all_reduce_scalar, reduce_accumulator

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    dev_id = args.device if args.device >= 0 else local_rank
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    w, h, depth = args.width, args.height, args.trace_depth
    sc = HostScene(args.scene, w, h, trace_depth=depth, enable_dof=0, use_hip=True, device=dev_id, seed=777)
    if sc.unsupported():
        raise SystemExit("bench.py: scene uses features outside the HIP layer's subset:\n" + sc.log())
    core = sc.hip()
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)       # SetExternalImageAccumulator: reduced over RCCL
    core.set_external_accumulator(accum.data_ptr(), accum.numel() * 4)
    core.set_tile_partition(rank, world, args.tile)
    core.set_option("samples_in_flight", min(args.spp_per_step, 64))   # one sub-pass per step; same K on every rank
    sc.draw(passes=1, spp=args.spp_per_step)   # first Draw: camera matrices, globals, InitPathTracing(seed), one pass of the step size
    max_depth = depth + 1

    # algorithmic work of one step (counting kernel variants, outside the timed region): the same number of samples per
    # pixel through the same generator streams as a timed step, so rays/quads/triangles per step agree to ~0.1 %
    core.enable_traversal_counters(True)
    core.trace_pass(args.spp_per_step)
    core.finish()
    counters = core.traversal_counters(max_depth)
    core.enable_traversal_counters(False)
    bytes_per_step = traversal_bytes(counters)                    # [depth, 2]

    for _ in range(args.warmup):
        core.trace_pass(args.spp_per_step)
    core.clear()
    core.enable_stage_timing(True)
    core.reset_perf_counters()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        core.trace_pass(args.spp_per_step)
    if world > 1 and args.backend != "nccl":                       # rehearsal: gloo reduces host memory
        host = accum.cpu()
        reduce_accumulator(host, dst=0)
        accum.copy_(host)
    else:
        reduce_accumulator(accum, dst=0)                          # the one RCCL exchange of the frame
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    st = core.rays_stat()
    rays_local = int(st.extensionRays + st.shadowRays)
    cdev = dev if (world == 1 or args.backend == "nccl") else "cpu"
    rays_total = all_reduce_scalar(rays_local, cdev)
    t_max = all_reduce_max(elapsed, cdev)
    spp_total = args.steps * args.spp_per_step
    trace_bytes = float(bytes_per_step[:, 0].sum()) * args.steps  # closest-hit launches of this rank in the timed region
    trace_s = st.traversalTimeMs * 1e-3
    achieved = trace_bytes / trace_s / 1e9 if trace_s > 0 else 0.0

    if rank == 0:
        img = accum.cpu().numpy() / float(spp_total)
        assert np.isfinite(img).all()
        result = {
            "metric": "Mrays/s at 1080p, 8 bounces, 256 spp; 1/2/4/8 GPUs + % HBM roofline",
            "value": rays_total / t_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: Cornell-box-style test scene (reference hydra_app/tests/test_224, 25.6k-tri teapot), "
                                   "%dx%d, %d bounces, %d spp, PT (MIS) integrator" % (w, h, depth, spp_total),
                       "spp_per_step": args.spp_per_step, "samples_in_flight": core.samples_in_flight(), "tile": args.tile, "partition": "image tiles, t %% %d" % world,
                       "rays": int(rays_total), "mean_radiance": float(img[..., :3].mean())},
            "roofline": {"bound": "hbm", "kernel": "k_trace_dyn<false,false> (closest-hit BVH4 traversal, persistent form)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic("%s|%dx%d|d%d|spp%d" % (os.path.basename(args.scene), w, h, depth, args.spp_per_step)),
                         "launches": int(st.traceLaunches), "avg_launch_ms": st.traversalTimeMs / max(int(st.traceLaunches), 1),
                         "algorithmic_bytes_per_launch": trace_bytes / max(int(st.traceLaunches), 1)},
            "stage_ms": {"raygen": st.raygenTimeMs, "trace": st.traversalTimeMs, "bounce_hit_light_bsdf": st.evalHitMs, "shadow": st.shadowTimeMs,
                         "shade_split_form_only": st.shadeTimeMs, "accumulate": st.accumTimeMs, "pass_total": st.passTimeMs},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.scene, depth)
        print(json.dumps(result))
    sc.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
