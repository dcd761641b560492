#!/usr/bin/env python3
"""Traversal micro-benchmark (GPU box): replays the live ray sets of bounce 0 / 1 / 3 (closest hit) and the shadow rays
of bounce 0 / 2 of one 1080p sample through the traversal kernels under different tuning options.  The ray sets come
from the CPU oracle (test infrastructure), so this script lives outside the product package."""
import argparse
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default=os.path.join(ROOT, "tests", "golden", "scenes", "test_224"))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--reorder", action="store_true", help="also time each ray set shuffled / sorted by origin cell / by direction octant + origin cell")
    ap.add_argument("--configs", default="0:0:16,1:0:12,1:24:12,1:32:12,1:40:12,1:48:12,1:56:12,1:40:8,1:40:16,1:40:24")
    args = ap.parse_args()
    from hydracore_amd import HipCore, HostScene
    from oracle_lib import Oracle
    sc = HostScene(args.scene, args.width, args.height, trace_depth=args.depth, enable_dof=0, use_hip=False)
    b = sc.buffers()
    orc = Oracle(b)
    sets = {}
    t0 = time.time()
    for bounce in (0, 1, 3):
        sets["closest b%d" % bounce] = orc.collect_rays(777, bounce, shadow=False) + (False,)
    for bounce in (0, 2):
        sets["shadow  b%d" % bounce] = orc.collect_rays(777, bounce, shadow=True) + (True,)
    print("ray sets collected in %.1f s: %s" % (time.time() - t0, {k: len(v[0]) for k, v in sets.items()}), flush=True)
    if args.reorder:
        def morton(p, bits):
            lo, hi = p.min(axis=0), p.max(axis=0)
            q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-20) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
            key = np.zeros(len(p), np.int64)
            for bit in range(bits):
                for ax in range(3):
                    key |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
            return key
        rng = np.random.default_rng(1)
        more = {}
        for name, (pos, dr, tf, shadow) in sets.items():
            if name.endswith("b0"):
                continue
            octant = (dr[:, 0] < 0).astype(np.int64) | ((dr[:, 1] < 0).astype(np.int64) << 1) | ((dr[:, 2] < 0).astype(np.int64) << 2)
            orders = {"shuf": rng.permutation(len(pos)), "mort10": np.argsort(morton(pos[:, :3], 10), kind="stable"),
                      "mort4": np.argsort(morton(pos[:, :3], 4), kind="stable"),
                      "oct+m4": np.argsort((octant << 12) | morton(pos[:, :3], 4), kind="stable"),
                      "m3+oct": np.argsort((morton(pos[:, :3], 3) << 3) | octant, kind="stable")}
            # block-local variant: sort inside consecutive groups of 256 rays only (what a k_hit block could do in LDS)
            g = np.arange(len(pos)) // 256
            orders["blk256 oct"] = np.lexsort((octant, g))
            for oname, o in orders.items():
                more[name[:7] + name[-2:] + " " + oname] = (pos[o], dr[o], tf[o] if hasattr(tf, "__len__") else tf, shadow)
        sets.update(more)
    core = HipCore(args.width, args.height)
    core.upload_scene(b)
    # algorithmic bytes per set from the per-ray counters (closest only)
    byts = {}
    for name, (pos, dr, tf, shadow) in sets.items():
        if not shadow:
            _, cnt, leaves = orc.trace(pos, dr, counters=True)
            byts[name] = 52.0 * len(pos) + 128.0 * cnt[:, 0].sum() + 128.0 * cnt[:, 1].sum() + 16.0 * leaves.sum() + 48.0 * cnt[:, 2].sum()
    print("%-12s %8s | %s" % ("config", "", " | ".join("%-22s" % k for k in sets)))
    for cfg in args.configs.split(","):
        mode, mina, bpc = [int(x) for x in cfg.split(":")]
        core.set_option("trace_mode", mode)
        core.set_option("trace_min_active", mina)
        core.set_option("trace_blocks_per_cu", bpc)
        cells = []
        for name, (pos, dr, tf, shadow) in sets.items():
            p = pos.copy()
            if shadow:
                p[:, 3] = tf
            ms = core.bench_trace(p, dr, iters=args.iters, shadow=shadow)
            cell = "%6.3f ms %6.0f Mr/s" % (ms, len(pos) / ms / 1e3)
            if name in byts:
                cell += " %4.0fGB/s" % (byts[name] / ms / 1e6)
            cells.append(cell)
        if args.reorder:
            for name, cell in zip(sets, cells):
                print("%-12s %-24s %s" % (cfg, name, cell), flush=True)
        else:
            print("%-12s %8s | %s" % (cfg, "", " | ".join(cells)), flush=True)


if __name__ == "__main__":
    main()
