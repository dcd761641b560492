#!/usr/bin/env python3
"""One bench-shaped step under `rocprofv3 --pmc` (started by bench.py as a child process, one counter group per run, or by
tools/pmc_passes.sh by hand).  No torch: the scene front end and the HIP layer through ctypes only, internal accumulator.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 tools/pmc_child.py --scene <dir> --spp 64

Runs `--steps` passes of `--spp` samples per pixel, all in flight at once -- the launches bench.py times.
`--mode mmlt`: mmlt_begin + (1 + `--mutations`) mutation steps of `--chains` Markov chains instead (BASELINE configs[4]).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", required=True, help="scene library directory")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--mode", default="pt", choices=["pt", "mmlt"])
    ap.add_argument("--chains", type=int, default=1 << 20)
    ap.add_argument("--mutations", type=int, default=8)
    ap.add_argument("--max-depth", type=int, default=6)
    ap.add_argument("--first-bounce", type=int, default=3)
    args = ap.parse_args()
    from hydracore_amd import HostScene
    if args.mode == "mmlt":
        sc = HostScene(args.scene, args.width, args.height, trace_depth=args.depth, enable_dof=0, use_hip=True, device=args.device, seed=777)
        core = sc.hip()
        core.set_option("samples_in_flight", 1)
        sc.draw(passes=1, spp=1)
        core.mmlt_begin(args.chains, seed=777, first_bounce=args.first_bounce, max_depth=args.max_depth, estimate_passes=1)
        core.mmlt_pass(1 + args.mutations)
        core.finish()
        print("pmc_child: %d chains x %d mutations" % (args.chains, 1 + args.mutations))
        core.mmlt_end()
        sc.close()
        return
    sc = HostScene(args.scene, args.width, args.height, trace_depth=args.depth, enable_dof=0, use_hip=True, device=args.device, seed=777)
    core = sc.hip()
    core.set_tile_partition(args.rank, args.world, args.tile)
    core.set_option("samples_in_flight", min(args.spp, 512))
    sc.draw(passes=1, spp=args.spp)
    for _ in range(args.steps - 1):
        core.trace_pass(args.spp)
    core.finish()
    st = core.rays_stat()
    print("pmc_child: %d rays" % int(st.extensionRays + st.shadowRays))
    sc.close()


if __name__ == "__main__":
    main()
