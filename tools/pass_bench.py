#!/usr/bin/env python3
"""Whole-pass tuning sweep on the GPU box: times `--spp` samples per option combination with per-stage HIP events.
   python tools/pass_bench.py --sweep shade_waves=3,4,5 --sweep static_blocks_per_cu=8,16,32"""
import argparse
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="test_224")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--sweep", action="append", default=[])
    ap.add_argument("--per-bounce", action="store_true", help="also print the bounce kernel's time per bounce (ms)")
    ap.add_argument("--in-flight", type=int, default=0, help="samples per pixel in flight (option samples_in_flight; 0 = the layer's default for the resolution)")
    args = ap.parse_args()
    from conftest import scene_path
    from hydracore_amd import HostScene
    sc = HostScene(scene_path(args.scene), args.width, args.height, trace_depth=args.depth, enable_dof=0, use_hip=True)
    core = sc.hip()
    if args.in_flight > 0:
        core.set_option("samples_in_flight", args.in_flight)
    sc.draw(1, 1)
    core.enable_stage_timing(True)
    names = [s.split("=")[0] for s in args.sweep]
    values = [[int(v) for v in s.split("=")[1].split(",")] for s in args.sweep]
    print("%-40s %8s %8s %8s %8s %8s %8s %9s" % ("options", "trace", "hit", "shadow", "shade", "other", "total", "Mrays/s"))
    for combo in itertools.product(*values) if values else [()]:
        for n, v in zip(names, combo):
            core.set_option(n, v)
        core.trace_pass(2)
        core.reset_perf_counters()
        core.trace_pass(args.spp)
        st = core.rays_stat()
        other = st.passTimeMs - st.traversalTimeMs - st.evalHitMs - st.shadowTimeMs - st.shadeTimeMs
        rays = st.extensionRays + st.shadowRays
        print("%-40s %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f %9.0f" % (" ".join("%s=%d" % (n, v) for n, v in zip(names, combo)), st.traversalTimeMs, st.evalHitMs,
                                                                 st.shadowTimeMs, st.shadeTimeMs, other, st.passTimeMs, rays / st.passTimeMs / 1e3), flush=True)
        if args.per_bounce:
            pb = core.stage_times_per_bounce(args.depth + 1)
            print("    bounce kernel per bounce (ms): " + " ".join("%.2f" % x for x in pb[:, 1]) + "   closest: " + " ".join("%.2f" % x for x in pb[:, 0]), flush=True)


if __name__ == "__main__":
    main()
