#!/usr/bin/env python3
"""Generate the golden vectors that pin the CPU oracle: outputs of the REFERENCE's own inline functions
(hydra_drv/c*.h, compiled for gfx950 by oracle/build_ref.sh into oracle/_ref/ref_driver.hsaco) on seeded inputs.

Needs a GPU (run on the GPU box):   python tests/golden/make_golden.py gpurun_out/golden
then copy gpurun_out/golden/*.npz into tests/golden/ and commit them.  Inputs are stored next to the outputs so the
CPU test (tests/test_golden_ref.py) needs neither the GPU nor the reference.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = {"test_224": dict(w=96, h=96, depth=4, dof=0), "test_42": dict(w=96, h=96, depth=4, dof=1)}


def scene_inputs(b, seed):
    from conftest import random_rays
    w, h = b["width"], b["height"]
    rng = np.random.default_rng(seed)
    n_eye = 4096
    xy = np.stack([rng.integers(0, w, n_eye), rng.integers(0, h, n_eye)], 1).astype(np.int32)
    offs = rng.uniform(-1, 1, (n_eye, 4)).astype(np.float32)
    pos4, dir4 = random_rays(16384, seed + 1)
    ys, xs = np.divmod(np.arange(w * h), w)
    pxy = np.stack([xs, ys], 1).astype(np.int32)
    poffs = rng.uniform(-1, 1, (w * h, 4)).astype(np.float32)
    return xy, offs, pos4, dir4, pxy, poffs


def main(out_dir):
    from hydracore_amd import HostScene
    from oracle_lib import Oracle
    from ref_ocl import RefModule, RefScene
    os.makedirs(out_dir, exist_ok=True)
    mod = RefModule("ref_driver.hsaco")
    first = True
    for name, cfg in SCENES.items():
        sc = HostScene(os.path.join(HERE, "scenes", name), cfg["w"], cfg["h"], trace_depth=cfg["depth"], enable_dof=cfg["dof"], use_hip=False)
        b = sc.buffers()
        ref = RefScene(mod, b)
        if first:
            seeds = np.array([0, 1, 7, 777, 123456, 2147483647, 5, 6, 13], np.int32)
            out, st = ref.random(seeds, 64)
            np.savez_compressed(os.path.join(out_dir, "rng.npz"), seeds=seeds, out=out, state=st)
            first = False
        xy, offs, pos4, dir4, pxy, poffs = scene_inputs(b, 1234)
        epos, edir = ref.make_eye_rays(xy, offs)
        hits = ref.trace(pos4, dir4)
        surf = ref.eval_surface(pos4, dir4, hits)
        ppos, pdir = ref.make_eye_rays(pxy, poffs)
        gens = Oracle(b).init_generators(4242)          # RandomGenInit(seed + i): integer-exact, checked by rng.npz
        col, gens_out = ref.path_trace(ppos, pdir, gens)
        np.savez_compressed(os.path.join(out_dir, "ref_%s.npz" % name), width=cfg["w"], height=cfg["h"], depth=cfg["depth"], dof=cfg["dof"],
                            eye_xy=xy, eye_offs=offs, eye_pos=epos, eye_dir=edir,
                            ray_pos=pos4, ray_dir=dir4, hits=hits, surf=surf,
                            path_xy=pxy, path_offs=poffs, path_pos=ppos, path_dir=pdir, path_gens=gens, path_color=col, path_gens_out=gens_out)
        print(name, "hit fraction", float((hits["primId"] != -1).mean()), "mean radiance", float(col[:, :3].mean()))
    mod.close()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden"))
