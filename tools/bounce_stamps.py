#!/usr/bin/env python3
"""Where a wave of k_bounce spends its cycles: needs the experiment build
     make LIBDIR=hydracore_amd/lib_stamps EXTRA_DEFS=-DHK_EXP_BOUNCE_STAMPS
     HYDRA_AMD_LIB_DIR=$PWD/hydracore_amd/lib_stamps python tools/bounce_stamps.py [--scene atrium250k]
   (s_memtime at the phase boundaries, summed over all waves of all k_bounce launches of 64 samples per pixel at 1080p)."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="test_224")
    ap.add_argument("--depth", type=int, default=8)
    args = ap.parse_args()
    from conftest import scene_path
    from hydracore_amd import HostScene
    from hydracore_amd import capi
    sc = HostScene(scene_path(args.scene), 1920, 1080, trace_depth=args.depth, enable_dof=0, use_hip=True)
    core = sc.hip()
    sc.draw(1, 1)
    lib = ctypes.CDLL(os.path.join(capi.lib_dir(), "libhydra_hip.so"))
    buf = (ctypes.c_ulonglong * 16)()
    core.trace_pass(64)
    assert lib.hydra_hip_debug_bounce_stamps(buf, 1) == 0
    core.trace_pass(64)
    assert lib.hydra_hip_debug_bounce_stamps(buf, 1) == 0
    names = ["stage tables + queue", "class sort (3 barriers)", "state load + surface (hit, triangle, emission)", "compaction atomic", "light pick + sample",
             "material eval (direct light)", "BSDF sample (next bounce)", "stores"]
    total = float(sum(buf[:8]))
    print("%s: %d waves, %.0f cycles per wave" % (args.scene, buf[8], total / max(1, buf[8])))
    for k in range(8):
        print("  %-48s %5.1f %%" % (names[k], 100.0 * buf[k] / total))


if __name__ == "__main__":
    main()
