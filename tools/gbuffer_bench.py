#!/usr/bin/env python3
"""IHWLayer::EvalGBuffer at 1080p on the GPU box: 64 primary rays per pixel + the cluster vote, wall time of the call (device work + the
read-back of the two float4 layers).   python tools/gbuffer_bench.py [--scene atrium250k]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="test_224")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    from conftest import scene_path
    from hydracore_amd import HostScene
    sc = HostScene(scene_path(args.scene), args.width, args.height, trace_depth=8, enable_dof=0, use_hip=True)
    sc.draw(1, 1)
    core = sc.hip()
    core.eval_gbuffer(args.width, args.height)
    t = time.time()
    for _ in range(args.iters):
        d1, d2 = core.eval_gbuffer(args.width, args.height)
    dt = (time.time() - t) / args.iters
    rays = args.width * args.height * 64
    cov = (d1[..., 2].view(np.uint32) >> 24).astype(np.float32) / 255.0
    dev = core.get_option("last_gbuffer_device_us") * 1e-6
    print("%s %dx%d: %.1f ms per call, of which %.1f ms on the device (%d M primary rays: %.0f Mrays/s with the vote; the rest is the read-back of 2 x %d MB into pageable host memory); mean coverage %.3f, pixels with a hit %.3f"
          % (args.scene, args.width, args.height, dt * 1e3, dev * 1e3, rays // 1000000, rays / dev / 1e6, args.width * args.height * 16 // 1000000, cov.mean(), (d2[..., 2].view(np.int32) >= 0).mean()))


if __name__ == "__main__":
    main()
