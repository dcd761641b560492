#!/usr/bin/env python3
"""bench.py -- Mrays/s of the MI355X wavefront path tracer on BASELINE.json configs[1]:
Cornell-box-style test scene (tests/golden/scenes/test_224 = the reference's hydra_app/tests/test_224), 1920x1080,
8 bounces, PT integrator; 256 spp = 4 steps x 64 spp by default on one GPU.

A "step" = one pass of the hot path: `--spp-per-step` samples for every pixel this rank owns, all in flight at once
(ray generation, then per bounce: closest-hit traversal, the fused hit/emission/light-sample/BSDF kernel with
compaction, shadow traversal; finally accumulate).  With N > 1 GPUs the image plane is tile-partitioned, every rank
traces its 1/N of the pixels, and the float4 accumulator is reduced once over RCCL at the end of the timed region.
The default step is 64 x N samples per pixel, i.e. the same 133 M paths in flight on every GPU whatever N is: per-GPU work
per step is fixed and the frame gets N times the samples ("weak" scaling; `--spp-per-step 64` on N GPUs gives the
strong-scaling split of the one-GPU frame instead, with 1/N of the paths in flight per GPU).

Prints ONE JSON line (rank 0).  Rays are counted exactly on the device (every extension and shadow ray traced).
On one GPU the same line also carries, under "extra_configs", BASELINE configs[2] (the generated 249k-triangle atrium, closed hall,
1920x1080, 8 bounces, 4 x 64 spp) and configs[4] (MMLT on test_42 at 1920x1080, 1 M chains), each with its own live PMC rooflines;
the top-level fields stay those of configs[1].  `--no-extra` skips them.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
L2_PEAK_GBS = 34500.0     # aggregate L2 bandwidth, same guide, "L2 (per XCD)"
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0   # wave64 VALU instructions per ns: 1024 SIMD-32 units, 2 cycles per wave-instruction, 2.4 GHz max clock
L2_REQUEST_BYTES = 128    # one TCC request = one 128-byte line (calibrated on k_accumulate's known stream, profiles/r01/pmc_summary.csv)

PMC_GROUPS = ["FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum",
              "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS",
              # active lanes per VALU instruction (rocprofiler's VALUUtilization = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)), a run of its own
              "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"]
LANE_GROUP = len(PMC_GROUPS) - 1


def kernel_family(name):
    """the three kernel families of a bounce; the counting variants (<.., true>) are not timed and not profiled"""
    n = name.replace(" ", "")
    if n.startswith("void"):
        n = n[4:]
    # k_trace_dyn<ANYHIT, COUNT, TOPTRIS, ALPHA>: the first two arguments name the family, the others are variants of it
    if n.startswith("k_trace_dyn<false,false,") or n.startswith("k_trace_dyn<false,false>"):
        return "closest"
    if n.startswith("k_trace_dyn<true,false,") or n.startswith("k_trace_dyn<true,false>"):
        return "shadow"
    if n.startswith("k_bounce<"):
        return "bounce"
    for fam in ("k_mmlt_step", "k_mmlt_connect_end", "k_mmlt_connect_begin", "k_mmlt_mutate", "k_mmlt_accept", "k_mmlt_begin"):
        if n.startswith(fam + "<") or n.startswith(fam + "("):
            return fam[2:]
    return None


def live_pmc(child_args, out_dir, budget_s=420.0):
    """HBM / L2 / VALU counters of the timed launches, measured now: one `rocprofv3 --pmc` child run per counter group (the guide's
    rule: never mix groups, no trace domains), each running ONE bench-shaped step (tools/pmc_child.py <child_args>).  Returns
    {family: {counter: average per launch, "launches": n, "lane_fraction_per_launch": [...]}} or raises."""
    import csv
    import glob
    import shutil
    import subprocess
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    acc, lanes = {}, {}
    t0 = time.time()
    env = dict(os.environ, TMPDIR="/tmp")
    for gi, group in enumerate(PMC_GROUPS):
        left = budget_s - (time.time() - t0)
        if left < 30:
            raise RuntimeError("PMC passes ran out of their time budget")
        d = os.path.join(out_dir, "pass%d" % gi)
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--pmc"] + group.split() + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "pmc_child.py")] + [str(x) for x in child_args]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=min(left, 240))
        with open(os.path.join(out_dir, "pass%d.log" % gi), "w") as f:
            f.write(" ".join(cmd) + "\n" + r.stdout[-4000:] + r.stderr[-4000:])
        if r.returncode != 0:
            raise RuntimeError("rocprofv3 --pmc %s failed (exit %d): %s" % (group, r.returncode, (r.stderr or r.stdout)[-300:]))
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            raise RuntimeError("rocprofv3 --pmc %s wrote no counter_collection.csv" % group)
        for fn in files:
            with open(fn, newline="") as fh:
                for row in csv.DictReader(fh):
                    fam = kernel_family(row["Kernel_Name"])
                    if fam is None:
                        continue
                    if gi == LANE_GROUP:      # per launch, in dispatch order: launch k of a traversal family is bounce k
                        lanes.setdefault(fam, {}).setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
                        if row["Counter_Name"] == "SQ_ACTIVE_INST_VALU":
                            continue         # the average of this counter comes from the SQ group's run
                    e = acc.setdefault(fam, {}).setdefault(row["Counter_Name"], [0, 0.0])
                    e[0] += 1
                    e[1] += float(row["Counter_Value"])
    out = {}
    for fam, cs in acc.items():
        out[fam] = {c: tot / n for c, (n, tot) in cs.items()}
        out[fam]["launches"] = max(n for n, _ in cs.values())
        per = [v for _, v in sorted(lanes.get(fam, {}).items())]
        tc, ai = sum(v.get("SQ_THREAD_CYCLES_VALU", 0.0) for v in per), sum(v.get("SQ_ACTIVE_INST_VALU", 0.0) for v in per)
        if ai > 0:
            out[fam]["lane_fraction"] = tc / (64.0 * ai)
            out[fam]["lane_fraction_per_launch"] = [v.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * v["SQ_ACTIVE_INST_VALU"]) if v.get("SQ_ACTIVE_INST_VALU", 0.0) > 0 else None for v in per]
    return out


def committed_pmc(workload_key):
    """the same record from a committed run (profiles/*/pmc_live_<key>.json), newest round first, or None"""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_live_*.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload_key") == workload_key:
            best = (d, os.path.relpath(path, ROOT))
    return best


def roofline_entry(kernel, launches, total_ms, pass_ms, algo_bytes_total, pmc, algo_is_hbm):
    """one kernel family against the three ceilings it could hit: HBM (bytes that reached memory, PMC), L2 (requests x 128 B, PMC),
    VALU issue (wave-instructions, PMC).  `bound` names the ceiling it is closest to; algorithmic bytes stay next to it."""
    t = total_ms * 1e-3 / max(launches, 1)            # seconds per launch
    e = {"kernel": kernel, "launches": int(launches), "avg_launch_ms": total_ms / max(launches, 1), "time_share_of_pass": total_ms / pass_ms if pass_ms > 0 else 0.0,
         "algorithmic_bytes_per_launch": algo_bytes_total / max(launches, 1),
         "achieved_algorithmic_GBs": algo_bytes_total / max(launches, 1) / t / 1e9 if t > 0 else 0.0}
    fr = {}
    if pmc and t > 0:
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # gfx950: FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane loads at 64 B (guide, HBM section): x 2; WRITE_SIZE is exact; both in KB
            # The x 2 is calibrated for wide coalesced streams only (k_accumulate here, profiles/r01/pmc_summary.csv; k_bounce's state streams: raw
            # FETCH_SIZE would put its traffic at 0.6 of the bytes it cannot avoid moving); for the scattered 16-byte loads of the traversal
            # kernels it is uncalibrated, so `frac` is an UPPER bound and `frac_lower` (raw FETCH_SIZE) the lower one.
            hb = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
            hb_lo = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
            e["hbm"] = {"bytes_per_launch": hb, "GBs": hb / t / 1e9, "peak_GBs": HBM_PEAK_GBS, "frac": hb / t / 1e9 / HBM_PEAK_GBS,
                        "bytes_per_launch_lower": hb_lo, "frac_lower": hb_lo / t / 1e9 / HBM_PEAK_GBS,
                        "traffic_over_algorithmic": hb / e["algorithmic_bytes_per_launch"] if e["algorithmic_bytes_per_launch"] > 0 else None}
            fr["hbm"] = e["hbm"]["frac"]
        if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
            req = pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]
            e["l2"] = {"bytes_per_launch": req * L2_REQUEST_BYTES, "GBs": req * L2_REQUEST_BYTES / t / 1e9, "peak_GBs": L2_PEAK_GBS,
                       "frac": req * L2_REQUEST_BYTES / t / 1e9 / L2_PEAK_GBS, "hit_rate": pmc["TCC_HIT_sum"] / req if req > 0 else None}
            fr["l2"] = e["l2"]["frac"]
        if "SQ_INSTS_VALU" in pmc:
            gi = pmc["SQ_INSTS_VALU"] / t / 1e9
            e["valu"] = {"wave_insts_per_launch": pmc["SQ_INSTS_VALU"], "Ginst_s": gi, "peak_Ginst_s": VALU_PEAK_GINST, "frac": gi / VALU_PEAK_GINST}
            if pmc.get("SQ_WAVE_CYCLES", 0) > 0:
                e["valu"]["wave_cycles_waiting_on_memory"] = pmc.get("SQ_WAIT_ANY", 0.0) / pmc["SQ_WAVE_CYCLES"]
                e["valu"]["wave_cycles_issue_stalled"] = pmc.get("SQ_WAIT_INST_ANY", 0.0) / pmc["SQ_WAVE_CYCLES"]
            if "lane_fraction" in pmc:      # active lanes per VALU wave-instruction / 64: what SIMT divergence and partly filled waves leave of the issue rate
                e["valu"]["active_lane_fraction"] = pmc["lane_fraction"]
                e["valu"]["active_lane_fraction_per_launch"] = pmc.get("lane_fraction_per_launch")
                e["valu"]["useful_frac"] = e["valu"]["frac"] * pmc["lane_fraction"]
            fr["valu"] = e["valu"]["frac"]
    if fr:
        e["bound"] = max(fr, key=fr.get)
    else:
        e["bound"] = "hbm" if algo_is_hbm else "unmeasured"
    return e


def traversal_bytes(counters):
    """Algorithmic bytes of the traversal kernels: per ray 36 B in (pos, dir, flags/count) + 16 B out,
    128 B per quad visited, 128 B per instance quad entered, 16 B per leaf header, 48 B per triangle tested.
    counters: uint64 [depth, 2 (closest|shadow), 5 (rays, quads, insts, leaves, tris)] -> bytes [depth, 2]"""
    c = counters.astype("float64")
    return c[..., 0] * (36 + 16) + c[..., 1] * 128 + c[..., 2] * 128 + c[..., 3] * 16 + c[..., 4] * 48


def cpu_baseline(scene, depth, budget_s=20.0):
    """the CPU oracle (kind "port") on a bounded sample of the same workload: a 640x360 frame of the same scene and depth
    (3 600 dynamically scheduled 64-pixel chunks per pass, so that every host thread stays loaded), as many spp as fit the
    budget, OpenMP over all host cores.  Timed in its -O3 build with the traversal's visit counters compiled out
    (oracle/liboracle_fast.so: same source, never used as the checker)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hydracore_amd import HostScene
    from oracle_lib import Oracle
    w, h = 640, 360
    sc = HostScene(scene, w, h, trace_depth=depth, enable_dof=0, use_hip=False)
    orc = Oracle(sc.buffers(), fast=True)
    gens = orc.init_generators(777)
    img, _, gens = orc.render(2, gens=gens)          # warm-up passes (page-in, thread pool)
    spp, rays, done = 0, 0, 2
    t0 = time.time()
    while time.time() - t0 < budget_s and spp < 4096:
        img, r, gens = orc.render(4, gens=gens, image=img, spp_done=done)
        spp += 4
        done += 4
        rays += r
    dt = time.time() - t0
    threads = orc.max_threads()
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "per_core": rays / dt / 1e6 / max(threads, 1), "kind": "port",
            "sample": "%dx%d, %d bounces, %d spp of the same scene (%d rays, %.1f s) on %d OpenMP threads = %.3f Mrays/s per thread; CPU oracle at -O3 with its "
                      "visit counters compiled out, own BVH4 walk (the stock reference CPU layer traces through Embree 2.17, absent here)" % (w, h, depth, spp, rays, dt, threads, rays / dt / 1e6 / max(threads, 1))}


ATRIUM_NAMES = ("atrium250k", "atrium250k_sky", "atrium250k_glass", "atrium250k_nmap", "atrium250k_cutouts")


def resolve_scene(name, rank=0, world=1, barrier=None):
    """(scene directory, workload description) for a scene library path or one of the generated atrium names"""
    if name in ATRIUM_NAMES:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import scene_path
        if rank == 0:
            scene_path(name)                           # generate once; the other ranks wait at the barrier
        if barrier is not None:
            barrier()
        what = ("open roof + constant sky light 0.5 next to the roof light" if name.endswith("_sky") else
                "closed hall, roof light only; glass pots, rough-glass arches, layered glass bands, thin-glass curtains" if name.endswith("_glass") else
                "closed hall, roof light only; normal-mapped floor, walls and columns" if name.endswith("_nmap") else
                "closed hall, roof light only; 60 instanced plants of alpha-tested cards" if name.endswith("_cutouts") else "closed hall, roof light only")
        return scene_path(name), "configs[2]: generated Sponza-class atrium (tools/make_atrium.py, 249k triangles, 173 instances, textured; %s)" % what
    base = os.path.basename(name.rstrip("/"))
    if base == "test_42":
        return name, "the north star's own case: reference hydra_app/tests/test_42 (its teapot chunk is a missing blob: box + light)"
    return name, "configs[1]: Cornell-box-style test scene (reference hydra_app/tests/test_224, 25.6k-tri teapot)"


def measure_pt(scene_dir, w, h, depth, spp_per_step, steps, warmup, dev_id, rank=0, world=1, tile=64, dist_ctx=None, pmc=True, pmc_out="", exchange_mode="native", backend="nccl"):
    """the path tracer on one scene: counting pass, warm-up, `steps` timed steps, the per-family rooflines.  Returns (numbers dict, HostScene-free)."""
    import numpy as np
    import torch
    from hydracore_amd import HostScene
    from hydracore_amd.multi_gpu import all_reduce_max, all_reduce_scalar, native_comm_init, reduce_accumulator
    dist = dist_ctx
    dev = torch.device("cuda", dev_id)
    sc = HostScene(scene_dir, w, h, trace_depth=depth, enable_dof=0, use_hip=True, device=dev_id, seed=777)
    if sc.unsupported():
        raise SystemExit("bench.py: scene uses features outside the HIP layer's subset:\n" + sc.log())
    core = sc.hip()
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)       # SetExternalImageAccumulator: reduced over RCCL
    core.set_external_accumulator(accum.data_ptr(), accum.numel() * 4)
    core.set_tile_partition(rank, world, tile)
    core.set_option("samples_in_flight", min(spp_per_step, 512))  # one sub-pass per step; same K on every rank
    sc.draw(passes=1, spp=spp_per_step)   # first Draw: camera matrices, globals, InitPathTracing(seed), one pass of the step size
    max_depth = depth + 1

    # algorithmic work of one step (counting kernel variants, outside the timed region): the same number of samples per
    # pixel through the same generator streams as a timed step, so rays/quads/triangles per step agree to ~0.1 %
    core.enable_traversal_counters(True)
    core.trace_pass(spp_per_step)
    core.finish()
    counters = core.traversal_counters(max_depth)
    core.enable_traversal_counters(False)
    bytes_per_step = traversal_bytes(counters)                    # [depth, 2]

    for _ in range(warmup):
        core.trace_pass(spp_per_step)
    # N > 1: rehearse the layer's own RCCL exchange on the warm-up frame and check it, bit for bit, against torch.distributed.reduce
    # of the same frame; every rank then takes the same decision
    exchange = "none (one rank)"
    if world > 1:
        exchange = "torch.distributed.reduce(SUM) of the zero-padded full frame"
        if exchange_mode == "native" and backend == "nccl":
            ok = torch.ones(1, device=dev)
            why = ""
            try:
                native_comm_init(core, rank, world, dev)
            except Exception as e:          # noqa: BLE001 -- any failure means: use the torch path
                ok.zero_()
                why = str(e)[:160]
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() > 0:
                core.finish()
                expected = accum.clone()
                reduce_accumulator(expected, dst=0)
                try:
                    core.comm_gather_frame(0)   # its ranks agree on what they are about to send before any of them posts a send or a receive: all fail together or none does
                    core.finish()
                except Exception as e:          # noqa: BLE001
                    ok.zero_()
                    why = str(e)[:160]
                torch.cuda.synchronize()
                if rank == 0 and ok.item() > 0 and not torch.equal(accum, expected):
                    ok.zero_()
                    why = "gathered frame differs from the reduced frame"
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() > 0:
                exchange = "hydra_hip_comm_gather_frame: RCCL send/recv of every rank's own tiles (1/%d of the frame each) to rank 0" % world
            elif rank == 0:
                exchange += " (native gather not used: %s)" % (why or "another rank failed to initialise or run it")
    use_native = exchange.startswith("hydra_hip_comm")
    core.clear()
    core.enable_stage_timing(True)
    core.reset_perf_counters()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        core.trace_pass(spp_per_step)
    core.finish()                                                  # the frame is complete before it is exchanged (inside the timed region)
    if world > 1 and backend != "nccl":                            # rehearsal: gloo reduces host memory
        host = accum.cpu()
        reduce_accumulator(host, dst=0)
        accum.copy_(host)
    elif use_native:
        core.comm_gather_frame(0)                                  # the one RCCL exchange of the frame, on the layer's stream
        core.finish()
    elif world > 1:
        reduce_accumulator(accum, dst=0)                          # the one RCCL exchange of the frame
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    st = core.rays_stat()
    rays_local = int(st.extensionRays + st.shadowRays)
    cdev = dev if (world == 1 or backend == "nccl") else "cpu"
    rays_total = all_reduce_scalar(rays_local, cdev)
    t_max = all_reduce_max(elapsed, cdev)
    spp_total = steps * spp_per_step
    # algorithmic bytes of the timed launches of this rank.  Traversal: the per-ray figure from the counting pass.
    # Bounce kernel: the path state it must move: 108 B in per path (pos, dir, throughput, radiance, pending estimate float4s, generator 8 B, visibility 4 B,
    # hit 16 B), 120 B out per survivor (5 float4 + generator + shadow origin/direction), 24 B per terminated path (contribution + generator).
    paths_in = counters[:, 0, 0].astype("float64")
    survivors = np.concatenate([paths_in[1:], [0.0]])
    bounce_bytes_step = float((paths_in * 108 + survivors * 120 + (paths_in - survivors) * 24).sum())
    per_bounce_ms = core.stage_times_per_bounce(max_depth)        # [depth, 3] over the timed steps of this rank
    out = None
    if rank == 0:
        img = accum.cpu().numpy() / float(spp_total)
        assert np.isfinite(img).all()
        out = {"value": rays_total / t_max / 1e6, "unit": "Mrays/s", "ms_per_step": 1e3 * t_max / steps, "steps": steps, "warmup": warmup,
               "spp_per_step": spp_per_step, "spp": spp_total, "samples_in_flight": core.samples_in_flight(), "exchange": exchange,
               "rays": int(rays_total), "mean_radiance": float(img[..., :3].mean()),
               "stage_ms": {"raygen": st.raygenTimeMs, "trace": st.traversalTimeMs, "bounce_hit_light_bsdf": st.evalHitMs, "shadow": st.shadowTimeMs,
                            "shade_split_form_only": st.shadeTimeMs, "accumulate": st.accumTimeMs, "pass_total": st.passTimeMs}}
        key = "%s|%dx%d|d%d|spp%d|w%d" % (os.path.basename(scene_dir.rstrip("/")), w, h, depth, spp_per_step, world)
        counters_pmc, pmc_source = None, None
        if world == 1 and pmc:
            out_dir = os.path.join(pmc_out or os.path.join(ROOT, "gpurun_out", "pmc_live"), key.replace("|", "_"))
            try:
                os.makedirs(out_dir, exist_ok=True)
            except OSError:
                import tempfile
                out_dir = tempfile.mkdtemp(prefix="pmc_live_")
            try:
                counters_pmc = live_pmc(["--scene", scene_dir, "--width", w, "--height", h, "--depth", depth, "--spp", spp_per_step, "--rank", rank, "--world", world,
                                         "--tile", tile, "--device", dev_id], out_dir)
                pmc_source = "live: rocprofv3 --pmc child runs of tools/pmc_child.py, one counter group per run (%s)" % "; ".join(PMC_GROUPS)
                with open(os.path.join(out_dir, "pmc_live_%s.json" % key.replace("|", "_")), "w") as f:
                    json.dump({"workload_key": key, "source": pmc_source, "per_launch": counters_pmc}, f, indent=1)
            except Exception as e:   # profiler missing / refused / timed out: say so and fall back to the committed record
                pmc_source = "live PMC failed (%s)" % str(e)[:200]
        if counters_pmc is None:
            got = committed_pmc(key)
            if got is not None:
                counters_pmc = got[0]["per_launch"]
                pmc_source = (pmc_source + "; " if pmc_source else "") + "committed record " + got[1]
        lanes_c = ((counters_pmc or {}).get("closest") or {}).get("lane_fraction_per_launch") or []
        lanes_s = ((counters_pmc or {}).get("shadow") or {}).get("lane_fraction_per_launch") or []

        # traversal-only rate per bounce class: ray counts and algorithmic bytes of one step from the counting
        # pass, times from the timed steps; this rank's share of the frame.  active_lane_fraction: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) of
        # the class's launches in the PMC child's step (launch k of a family = bounce k), weighted by their ray counts
        def cls(lo, hi, kind):
            rays = float(counters[lo:hi, kind, 0].sum()) * steps
            ms = float(per_bounce_ms[lo:hi, 2 if kind else 0].sum())
            r = {"rays": int(rays), "ms": ms, "Mrays/s": rays / ms / 1e3 if ms > 0 else 0.0,
                 "algorithmic_GB/s": float(bytes_per_step[lo:hi, kind].sum()) * steps / ms / 1e6 if ms > 0 else 0.0}
            lf = lanes_s if kind else lanes_c
            pairs = [(float(counters[k, kind, 0]), lf[k]) for k in range(lo, min(hi, len(lf))) if lf[k] is not None and counters[k, kind, 0] > 0]
            if pairs:
                r["active_lane_fraction"] = sum(a * b for a, b in pairs) / sum(a for a, _ in pairs)
            return r
        out["traversal_per_bounce_class"] = {
            "closest_primary": cls(0, 1, 0), "closest_bounce_1": cls(1, 2, 0), "closest_bounce_2plus": cls(2, max_depth, 0),
            "shadow_bounce_0": cls(0, 1, 1), "shadow_bounce_1plus": cls(1, max_depth, 1)}
        out["bounce_kernel_ms_per_bounce"] = [float(x) for x in per_bounce_ms[:, 1]]
        # ---- roofline: each kernel family of a bounce against the HBM, L2 and VALU-issue ceilings, counters measured in this run
        pass_ms = st.passTimeMs
        fams = {
            "closest": roofline_entry("k_trace_dyn<false,false> (closest-hit BVH4 traversal + Moeller-Trumbore, persistent form)", st.traceLaunches, st.traversalTimeMs, pass_ms,
                                      float(bytes_per_step[:, 0].sum()) * steps, (counters_pmc or {}).get("closest"), False),
            "shadow": roofline_entry("k_trace_dyn<true,false> (any-hit shadow traversal, persistent form)", st.shadowLaunches, st.shadowTimeMs, pass_ms,
                                     float(bytes_per_step[:, 1].sum()) * steps, (counters_pmc or {}).get("shadow"), False),
            "bounce": roofline_entry("k_bounce (hit, emission/MIS, light sample, next-event estimate, BSDF sample, compaction)", st.traceLaunches, st.evalHitMs, pass_ms,
                                     bounce_bytes_step * steps, (counters_pmc or {}).get("bounce"), True),
        }
        out["roofline"] = dominant_roofline(fams, pmc_source)
        out["roofline_kernels"] = fams
    sc.close()
    del accum
    torch.cuda.empty_cache()
    return out


def dominant_roofline(fams, pmc_source):
    dom = max(fams, key=lambda k: fams[k]["avg_launch_ms"] * fams[k]["launches"])
    d = fams[dom]
    b = d["bound"]
    if b in ("hbm", "l2"):
        achieved, peak, unit = (d[b]["GBs"] if b in d else d["achieved_algorithmic_GBs"]), (HBM_PEAK_GBS if b == "hbm" else L2_PEAK_GBS), "GB/s"
    elif b == "valu":
        achieved, peak, unit = d["valu"]["Ginst_s"], VALU_PEAK_GINST, "G wave-instructions/s"
    else:
        achieved, peak, unit = d["achieved_algorithmic_GBs"], HBM_PEAK_GBS, "GB/s"
    return {"kernel": d["kernel"], "bound": b, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
            "traffic": d["hbm"]["bytes_per_launch"] if "hbm" in d else None, "counters": pmc_source,
            "launches": d["launches"], "avg_launch_ms": d["avg_launch_ms"], "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
            "achieved_algorithmic_GBs": d["achieved_algorithmic_GBs"],
            "hbm_frac_range": [d["hbm"]["frac_lower"], d["hbm"]["frac"]] if "hbm" in d else None,
            "active_lane_fraction": d.get("valu", {}).get("active_lane_fraction"),
            "note": "dominant kernel family by time in the timed region; frac = achieved / peak of the ceiling it is closest to (HBM bytes, L2 requests x 128 B "
                    "and VALU wave-instructions from PMC counters); hbm_frac_range = [raw FETCH_SIZE, FETCH_SIZE x 2 (the guide's factor for 16-byte-per-lane streams; "
                    "uncalibrated for scattered loads, hence an upper bound)]; active_lane_fraction = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); "
                    "achieved_algorithmic_GBs is the SURVEY 8d byte model over the same launches, which for traversal counts bytes served from LDS/L1/L2 and is not a fraction of HBM"}


def measure_mmlt(scene_dir, w, h, dev_id, chains=1 << 20, mutations=64, warmup=8, max_depth=6, first_bounce=3, pmc=True, pmc_out=""):
    """BASELINE configs[4]: IntegratorMMLT on test_42 at 1920x1080 -- `mutations` steps of `chains` Markov chains (one mutation = MutatePrimarySpace +
    F: two sub-paths through the traversal kernels and one connection + accept / contribute), HIP events around the timed steps, and the stage
    kernels' own PMC rooflines from a child run of the same shape"""
    import torch
    from hydracore_amd import HostScene
    sc = HostScene(scene_dir, w, h, trace_depth=8, enable_dof=0, use_hip=True, device=dev_id, seed=777)
    core = sc.hip()
    core.set_option("samples_in_flight", 1)
    sc.draw(passes=1, spp=1)      # the caller's Draw pushes the camera into the globals header (IHWLayer::SetCamMatrices), as the reference's does
    t0 = time.perf_counter()
    core.mmlt_begin(chains, seed=777, first_bounce=first_bounce, max_depth=max_depth, estimate_passes=1)
    core.finish()
    t_begin = time.perf_counter() - t0
    core.mmlt_pass(warmup)
    core.finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    core.mmlt_pass(mutations)
    core.finish()
    dt = time.perf_counter() - t0
    img, info = core.mmlt_image(w, h)
    core.mmlt_end()
    sc.close()
    out = {"workload": "configs[4]: MMLT (IntegratorMMLT equivalent) on test_42, %dx%d, %d chains, paths of %d..%d segments, %d timed mutations per chain" % (w, h, chains, first_bounce, max_depth, mutations),
           "metric": "M mutations/s (one mutation = MutatePrimarySpace + F + accept/contribute)", "value": chains * mutations / dt / 1e6, "unit": "M mutations/s",
           "ms_per_mutation_step": 1e3 * dt / mutations, "begin_s": t_begin, "acceptance": info["acceptance"], "avg_brightness": info["avg_brightness"],
           "mean_indirect_radiance": float(img[..., :3].mean())}
    if pmc:
        key = "mmlt_%s_%dx%d_c%d_d%d" % (os.path.basename(scene_dir.rstrip("/")), w, h, chains, max_depth)
        out_dir = os.path.join(pmc_out or os.path.join(ROOT, "gpurun_out", "pmc_live"), key)
        try:
            os.makedirs(out_dir, exist_ok=True)
            steps_child = 8
            c = live_pmc(["--mode", "mmlt", "--scene", scene_dir, "--width", w, "--height", h, "--chains", chains, "--mutations", steps_child, "--max-depth", max_depth,
                          "--first-bounce", first_bounce, "--device", dev_id], out_dir, budget_s=240.0)
            with open(os.path.join(out_dir, "pmc_live_%s.json" % key), "w") as f:
                json.dump({"workload_key": key, "per_launch": c}, f, indent=1)
            # what a chain's mutation must move through the two kernels whose traffic is fixed by the layout: the primary-sample vector (12 + 10 d floats)
            # read and written by k_mmlt_mutate; k_mmlt_accept reads both vectors' heads, F's 8 outputs and writes the chain record + 2 atomics of 16 B
            xbytes = (12 + 10 * max_depth) * 4
            algo = {"mmlt_mutate": 2.0 * xbytes * chains, "mmlt_accept": (2.0 * xbytes + 32 + 11 * 4 * 2 + 32) * chains}
            fams = {}
            # per-launch durations are not timed one by one inside the run: the PMC child's SQ_WAVE_CYCLES give each family's share, scaled to the timed step
            tot_cycles = sum(v.get("SQ_WAVE_CYCLES", 0.0) * v.get("launches", 0) for v in c.values())
            for fam, v in c.items():
                share = v.get("SQ_WAVE_CYCLES", 0.0) * v.get("launches", 0) / tot_cycles if tot_cycles > 0 else 0.0
                launches_per_step = v.get("launches", 0) / float(steps_child + 1)
                fams[fam] = {"launches_per_mutation_step": launches_per_step, "wave_cycle_share": share,
                             "hbm_bytes_per_mutation_step": [(v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 * launches_per_step, (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 * launches_per_step],
                             "l2_hit_rate": v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]) if v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0) > 0 else None,
                             "valu_wave_insts_per_mutation_step": v.get("SQ_INSTS_VALU", 0.0) * launches_per_step,
                             "wave_cycles_waiting_on_memory": v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES", 0) > 0 else None,
                             "active_lane_fraction": v.get("lane_fraction"),
                             "algorithmic_bytes_per_mutation_step": algo.get(fam)}
            out["roofline_kernels"] = fams
            out["roofline_note"] = ("families: the MMLT stage kernels and the path tracer's traversal kernels they run between; HBM bytes as [raw FETCH_SIZE, x 2] + WRITE_SIZE; "
                                    "wave_cycle_share = the family's SQ_WAVE_CYCLES over all families' in the PMC child's %d mutation steps" % steps_child)
            tot_hbm = sum(f["hbm_bytes_per_mutation_step"][1] for f in fams.values())
            out["hbm_frac_upper_all_kernels"] = tot_hbm / (dt / mutations) / 1e9 / HBM_PEAK_GBS
            tot_valu = sum(f["valu_wave_insts_per_mutation_step"] for f in fams.values())
            out["valu_frac_all_kernels"] = tot_valu / (dt / mutations) / 1e9 / VALU_PEAK_GINST
        except Exception as e:   # noqa: BLE001
            out["roofline_kernels"] = "live PMC failed (%s)" % str(e)[:200]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=0, help="samples per pixel per step, all in flight at once; default 64 x number of GPUs "
                    "(64 x 1080p = 133 M paths and 30 GB of path state per GPU)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--trace-depth", type=int, default=8)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--scene", default=os.path.join(ROOT, "tests", "golden", "scenes", "test_224"),
                    help="scene library directory, or 'atrium250k' / 'atrium250k_sky' / 'atrium250k_glass' / 'atrium250k_nmap' / 'atrium250k_cutouts' = BASELINE configs[2]/[3] and its glass, normal-map and cut-out variants (generated by tools/make_atrium.py on first use)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_configs (configs[2] atrium250k and configs[4] MMLT on test_42), which the one-GPU run otherwise appends")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc child runs (roofline then uses the committed profiles/*/pmc_live_*.json of this workload, if any)")
    ap.add_argument("--pmc-out", default="", help="directory for the PMC child runs' output (default: gpurun_out/pmc_live if writable, else a temp dir)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal of N ranks on one GPU)")
    ap.add_argument("--exchange", default="native", choices=["native", "torch"],
                    help="how the N > 1 ranks exchange the accumulator: native = the HIP layer's own RCCL gather of owned tiles (hydra_hip_comm_gather_frame, "
                         "rehearsed and checked against the torch path before the timed region, which is used instead if the check fails); torch = torch.distributed.reduce")
    ap.add_argument("--device", type=int, default=-1, help="force this HIP device for every rank (rehearsal only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

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

    if args.spp_per_step <= 0:
        args.spp_per_step = 64 * world
    weak = (args.spp_per_step == 64 * world)
    w, h, depth = args.width, args.height, args.trace_depth
    scene_arg = args.scene
    scene_dir, workload = resolve_scene(args.scene, rank, world, dist.barrier if world > 1 else None)
    m = measure_pt(scene_dir, w, h, depth, args.spp_per_step, args.steps, args.warmup, dev_id, rank, world, args.tile, dist if world > 1 else None,
                   pmc=not args.no_pmc, pmc_out=args.pmc_out, exchange_mode=args.exchange, backend=args.backend)
    if rank == 0:
        result = {
            "metric": "Mrays/s at 1080p, 8 bounces, 256 spp; 1/2/4/8 GPUs + % HBM roofline",
            "value": m["value"],
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload + ", %dx%d, %d bounces, PT (MIS) integrator; this run: %d steps x %d spp" % (w, h, depth, args.steps, args.spp_per_step),
                       "spp_metric": 256, "spp_this_run": m["spp"],
                       "spp_per_step": args.spp_per_step, "samples_in_flight": m["samples_in_flight"], "tile": args.tile, "exchange": m["exchange"], "partition": "image tiles in Morton order, i-th tile -> rank i %% %d" % world,
                       "rays": m["rays"], "mean_radiance": m["mean_radiance"]},
            "stage_ms": m["stage_ms"],
            "traversal_per_bounce_class": m["traversal_per_bounce_class"],
            "bounce_kernel_ms_per_bounce": m["bounce_kernel_ms_per_bounce"],
            "roofline": m["roofline"],
            "roofline_kernels": m["roofline_kernels"],
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(scene_dir, depth)
        # ---- the other single-GPU configurations of BASELINE.json, each measured the same way (own counting pass, own timed steps, own live PMC runs)
        if world == 1 and not args.no_extra and scene_arg == ap.get_default("scene") and (w, h, depth) == (1920, 1080, 8):
            extra = {}
            try:
                sd, wl = resolve_scene("atrium250k")
                a = measure_pt(sd, w, h, depth, 64, 4, 1, dev_id, pmc=not args.no_pmc, pmc_out=args.pmc_out)
                a["workload"] = wl + ", %dx%d, %d bounces, 4 steps x 64 spp (configs[2] names 1024 spp: the rate does not depend on the step count)" % (w, h, depth)
                a["metric"] = "Mrays/s"
                extra["configs[2]"] = a
            except (Exception, SystemExit) as e:   # noqa: BLE001 -- the headline line must survive a failure here
                extra["configs[2]"] = {"error": str(e)[:300]}
            try:
                extra["configs[4]"] = measure_mmlt(os.path.join(ROOT, "tests", "golden", "scenes", "test_42"), w, h, dev_id, pmc=not args.no_pmc, pmc_out=args.pmc_out)
            except (Exception, SystemExit) as e:   # noqa: BLE001
                extra["configs[4]"] = {"error": str(e)[:300]}
            result["extra_configs"] = extra
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
