#!/usr/bin/env python3
"""Per-ray averages of the traversal work by bounce (GPU box): quads, instance entries, leaves, triangles per closest-hit and shadow ray.
   python tools/trav_counters.py --scene atrium250k"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="atrium250k")
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--spp", type=int, default=4)
    args = ap.parse_args()
    from conftest import scene_path
    from hydracore_amd import HostScene
    sc = HostScene(scene_path(args.scene), args.width, args.height, trace_depth=args.depth, enable_dof=0, use_hip=True)
    core = sc.hip()
    sc.draw(1, 1)
    core.enable_traversal_counters(True)
    core.trace_pass(args.spp)
    c = core.traversal_counters(args.depth + 1).astype(float)
    print("%-8s %12s %8s %8s %8s %8s   | %12s %8s %8s %8s %8s" % ("bounce", "closest rays", "quads", "insts", "leaves", "tris", "shadow rays", "quads", "insts", "leaves", "tris"))
    for d in range(args.depth + 1):
        row = []
        for k in range(2):
            r = max(c[d, k, 0], 1.0)
            row += [c[d, k, 0], c[d, k, 1] / r, c[d, k, 2] / r, c[d, k, 3] / r, c[d, k, 4] / r]
        print("%-8d %12.0f %8.2f %8.2f %8.2f %8.2f   | %12.0f %8.2f %8.2f %8.2f %8.2f" % tuple([d] + row))
    print("out-of-range fetches:", core.traversal_oob())


if __name__ == "__main__":
    main()
