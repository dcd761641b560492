#!/usr/bin/env python3
"""Throughput of the IntegratorMMLT path (BASELINE.json configs[4]: MMLT on test_42, 1920x1080): mutations per second over all chains.

    python3 tools/mmlt_bench.py --scene tests/golden/scenes/test_42 --chains 1048576 --passes 8

One mutation = MutatePrimarySpace + F (two sub-paths through the traversal kernels, one connection) + accept / contribute.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default=os.path.join(ROOT, "tests", "golden", "scenes", "test_42"))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8, help="front-end trace depth setting")
    ap.add_argument("--max-depth", type=int, default=6, help="longest path (segments) the chains sample")
    ap.add_argument("--first-bounce", type=int, default=3)
    ap.add_argument("--chains", type=int, default=1 << 20)
    ap.add_argument("--passes", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    from hydracore_amd import HostScene
    sc = HostScene(args.scene, args.width, args.height, trace_depth=args.depth, enable_dof=0, use_hip=True, device=0, seed=777)
    core = sc.hip()
    sc.draw(passes=1, spp=1)      # the caller's Draw pushes the camera into the globals header (IHWLayer::SetCamMatrices), as the reference's does
    t0 = time.time()
    core.mmlt_begin(args.chains, seed=777, first_bounce=args.first_bounce, max_depth=args.max_depth, estimate_passes=1)
    core.finish()
    t_begin = time.time() - t0
    core.mmlt_pass(args.warmup)
    core.finish()
    t0 = time.time()
    core.mmlt_pass(args.passes)
    core.finish()
    dt = time.time() - t0
    img, info = core.mmlt_image(args.width, args.height)
    print(json.dumps({"metric": "mmlt_mutations_per_second", "value": args.chains * args.passes / dt, "unit": "mutations/s", "chains": args.chains,
                      "ms_per_pass": 1e3 * dt / args.passes, "begin_s": t_begin, "max_depth": args.max_depth, "first_bounce": args.first_bounce,
                      "acceptance": info["acceptance"], "avg_brightness": info["avg_brightness"], "workload": "%s %dx%d" % (os.path.basename(args.scene), args.width, args.height)}))
    core.mmlt_end()
    sc.close()


if __name__ == "__main__":
    main()
