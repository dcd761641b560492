#!/usr/bin/env python3
"""Run the host-emulated device code (libemu_device_asan.so) on the parity inputs and compare with the oracle.
Must be started with LD_PRELOAD=<libasan.so> ASAN_OPTIONS=detect_leaks=0 (tests/test_emu_device.py does that)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from conftest import host_scene, make_oracle, random_rays
    from oracle_lib import OrcScene
    lib = C.CDLL(os.path.join(HERE, "libemu_device_asan.so"))
    vp, i32 = C.c_void_p, C.c_int
    lib.emu_trace.argtypes = [C.POINTER(OrcScene), i32, vp, vp, vp, vp, i32, vp, vp]
    lib.emu_path_trace.argtypes = [C.POINTER(OrcScene), i32, vp, vp, vp, vp]
    lib.emu_eye_rays.argtypes = [C.POINTER(OrcScene), i32, i32, i32, vp, vp, vp, vp]

    def p(a):
        return a.ctypes.data_as(C.c_void_p)
    worst = 0.0
    hall = dict(center=(0.0, 4.0, 0.0), radius=3.0, spread=9.0)
    for name, ww, hh, depth, dof, rk in (("test_224", 96, 96, 4, 0, {}), ("test_42", 96, 96, 4, 1, {}), ("atrium_sky_small", 96, 54, 5, 0, hall), ("atrium_skytex_small", 96, 54, 5, 0, hall), ("atrium_lights_small", 96, 54, 5, 0, hall), ("atrium_glass_small", 96, 54, 8, 0, hall), ("atrium_ggx_small", 96, 54, 8, 0, hall)):
        _, b = host_scene(name, ww, hh, depth, dof)
        orc = make_oracle(b)
        w, h = b["width"], b["height"]
        # closest hit + counters + shadow
        pos4, dir4 = random_rays(20000, 21, **rk)
        hits = np.empty(len(pos4), np.dtype([("t", np.float32), ("primId", np.int32), ("instId", np.int32), ("geomId", np.int32)]))
        cnt = np.empty((len(pos4), 4), np.uint32)
        lib.emu_trace(C.byref(orc.s), len(pos4), p(pos4), p(dir4), p(hits), p(cnt), 0, None, None)
        ref, rcnt, rleaves = orc.trace(pos4, dir4, counters=True)
        assert (hits == ref).all(), "closest hit differs"
        assert (cnt[:, :3] == rcnt).all() and (cnt[:, 3] == rleaves).all(), "visit counters differ"
        tfar = np.random.default_rng(2).uniform(0.2, 25.0, len(pos4)).astype(np.float32)
        vis = np.empty(len(pos4), np.float32)
        lib.emu_trace(C.byref(orc.s), len(pos4), p(pos4), p(dir4), None, None, 1, p(tfar), p(vis))
        assert (vis == orc.shadow_trace(pos4, dir4, tfar)).all(), "shadow differs"
        # whole paths
        n = w * h
        ys, xs = np.divmod(np.arange(n), w)
        offs = np.random.default_rng(5).uniform(-1, 1, (n, 4)).astype(np.float32)
        xy = np.stack([xs, ys], 1).astype(np.int32)
        epos, edir = np.empty((n, 4), np.float32), np.empty((n, 4), np.float32)
        lib.emu_eye_rays(C.byref(orc.s), n, w, h, p(xy), p(offs), p(epos), p(edir))
        rpos, rdir = orc.make_eye_rays(xy, offs)
        assert np.abs(epos - rpos).max() < 2e-6 and np.abs(edir - rdir).max() < 2e-6, "eye rays differ"
        gens = orc.init_generators(4242)
        g = gens.copy()
        col = np.empty((n, 4), np.float32)
        lib.emu_path_trace(C.byref(orc.s), n, p(rpos), p(rdir), p(g), p(col))
        rcol, rg = orc.path_trace(rpos, rdir, gens)
        same = (g == rg).all(axis=1)
        err = np.abs(col[:, :3] - rcol[:, :3]) / np.maximum(np.abs(rcol[:, :3]), 1.0)
        worst = max(worst, float(err.max()))
        assert same.mean() > 0.999, "RNG draw counts differ on %.3f%% of the paths" % (100 * (1 - same.mean()))
        assert (err.max(axis=1) > 1e-4).mean() < 0.001, "radiance differs"
        print("%s: %d rays + %d paths ok, worst rel err %.3g, identical draws %.4f" % (name, len(pos4), n, err.max(), same.mean()))
    print("EMU_OK worst %.3g" % worst)


if __name__ == "__main__":
    main()
