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


def deep_chain_case(lib, p):
    """A degenerate tree deeper than the 80-entry traversal stack (the reference tests `top < 80` once and then
    pushes up to three links, ctrace.h:964-985).  Quad q = {inner child -> quad q + 1 (entered first), three leaves}
    (quad 1: one leaf): level 1 pushes one link and every further level three, so the test at level 27 sees top == 79 and the
    three pushes land on entries 79, 80 and 81.  The device code (HkStackT, 80 + 2 entries) must
    stay inside its arrays under ASan and visit exactly what the oracle visits."""
    from oracle_lib import OrcScene, load
    levels = 120
    nodes = np.zeros(((levels + 2) * 4, 8), np.float32)
    ni = nodes.view(np.int32)
    ni[:, 3] = -1
    ni[:, 7] = -1                                                    # invalid child: both words 0xFFFFFFFF
    tris = []

    def leaf(z):
        at = len(tris)
        tris.append([np.int32(at + 1).view(np.float32), np.int32(1).view(np.float32), np.int32(-1).view(np.float32), np.int32(-1).view(np.float32)])
        tris.append([-2.0, -2.0, z, np.int32(len(tris)).view(np.float32)])          # w = primId
        tris.append([2.0, -2.0, z, np.int32(0).view(np.float32)])
        tris.append([0.0, 2.0, z, np.int32(0).view(np.float32)])
        return at
    for q in range(1, levels + 1):
        zs = (0.5 + 0.001 * q, 0.7 + 0.001 * (levels - q), 0.9 + 0.0005 * q)
        if q < levels:
            nodes[4 * q, :3] = (-3, -3, -1.0); nodes[4 * q, 4:7] = (3, 3, 1.0); ni[4 * q, 3] = q + 1; ni[4 * q, 7] = 0
        for k, z in enumerate(zs[:1] if q == 1 else zs, 1):
            nodes[4 * q + k, :3] = (-3, -3, z - 1e-3); nodes[4 * q + k, 4:7] = (3, 3, z + 1e-3)
            ni[4 * q + k, 3] = np.int32(np.uint32(0x80000000 | leaf(z)).astype(np.int64) - (1 << 32)); ni[4 * q + k, 7] = 0
    tris = np.array(tris, np.float32)
    rng = np.random.default_rng(9)
    n = 2000
    pos4 = np.zeros((n, 4), np.float32); dir4 = np.zeros((n, 4), np.float32)
    pos4[:, :2] = rng.uniform(-1.5, 1.5, (n, 2)); pos4[:, 2] = -10.0
    d = np.stack([rng.uniform(-0.02, 0.02, n), rng.uniform(-0.02, 0.02, n), np.ones(n)], 1)
    dir4[:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True)
    s = OrcScene()
    s.bvh, s.tris, s.haveInst = nodes.ctypes.data, tris.ctypes.data, 0
    dt = np.dtype([("t", np.float32), ("primId", np.int32), ("instId", np.int32), ("geomId", np.int32)])
    hits, cnt = np.empty(n, dt), np.empty((n, 4), np.uint32)
    lib.emu_trace(C.byref(s), n, p(pos4), p(dir4), p(hits), p(cnt), 0, None, None)
    ref, rcnt, rleaves = np.empty(n, dt), np.empty((n, 3), np.uint32), np.empty(n, np.uint32)
    load().orc_trace(C.byref(s), n, p(pos4), p(dir4), p(ref), p(rcnt), p(rleaves))
    assert (hits == ref).all(), "deep chain: closest hit differs"
    assert (cnt[:, :3] == rcnt).all() and (cnt[:, 3] == rleaves).all(), "deep chain: visit counters differ"
    assert cnt[:, 0].max() == levels and cnt[:, 3].max() > 80 and (hits["primId"] != -1).mean() > 0.5
    print("deep chain (%d levels, stack overflows): %d rays ok, up to %d quads per ray" % (levels, n, cnt[:, 0].max()))


def main():
    from conftest import host_scene, make_oracle, random_rays
    from oracle_lib import OrcScene
    lib = C.CDLL(os.path.join(HERE, "libemu_device_asan.so"))
    vp, i32 = C.c_void_p, C.c_int
    lib.emu_trace.argtypes = [C.POINTER(OrcScene), i32, vp, vp, vp, vp, i32, vp, vp]
    lib.emu_path_trace.argtypes = [C.POINTER(OrcScene), i32, vp, vp, vp, vp]
    lib.emu_eye_rays.argtypes = [C.POINTER(OrcScene), i32, i32, i32, vp, vp, vp, vp]
    lib.emu_mmlt_run.argtypes = [C.POINTER(OrcScene), i32, vp, vp, i32, i32, vp, vp, vp, i32, vp]
    lib.emu_mmlt_f.argtypes = [C.POINTER(OrcScene), i32, vp, vp, i32, vp]
    lib.emu_gbuffer.argtypes = [C.POINTER(OrcScene), i32, i32, vp, vp, vp]
    lib.emu_bidir.argtypes = [C.POINTER(OrcScene), i32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp]

    def p(a):
        return a.ctypes.data_as(C.c_void_p)
    worst = 0.0
    hall = dict(center=(0.0, 4.0, 0.0), radius=3.0, spread=9.0)
    for name, ww, hh, depth, dof, rk in (("test_224", 96, 96, 4, 0, {}), ("test_42", 96, 96, 4, 1, {}), ("atrium_sky_small", 96, 54, 5, 0, hall), ("atrium_skytex_small", 96, 54, 5, 0, hall), ("atrium_lights_small", 96, 54, 5, 0, hall), ("atrium_glass_small", 96, 54, 8, 0, hall), ("atrium_ggx_small", 96, 54, 8, 0, hall), ("atrium_nmap_small", 96, 54, 5, 0, hall), ("atrium_transl_small", 96, 54, 5, 0, hall), ("atrium_aniso_small", 96, 54, 5, 0, hall), ("atrium_tubes_small", 96, 54, 5, 0, hall), ("atrium_portal_small", 96, 54, 5, 0, hall), ("atrium_ies_small", 96, 54, 5, 0, hall)):
        if os.environ.get("EMU_ONLY") and name not in os.environ["EMU_ONLY"].split(","):
            continue
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
        acnt = np.empty((len(pos4), 4), np.uint32)
        lib.emu_trace(C.byref(orc.s), len(pos4), p(pos4), p(dir4), None, p(acnt), 1, p(tfar), p(vis))
        assert (vis == orc.shadow_trace(pos4, dir4, tfar)).all(), "shadow differs"
        rvis, racnt = orc.shadow_trace_anyhit(pos4, dir4, tfar, counters=True)   # the early-out walk of ctrace.h:1065-1294
        assert (vis == rvis).all() and (acnt == racnt).all(), "any-hit walk differs from the oracle's"
        # row f3 building blocks (hk_bidir.h) against the oracle's, seeded inputs
        brng = np.random.default_rng(31)
        nb = 4096
        ids = brng.integers(0, int(b["globals"][238]), nb).astype(np.int32)
        r4, ct = brng.uniform(0, 1, (nb, 4)).astype(np.float32), brng.uniform(-0.25, 1, nb).astype(np.float32)
        surf = orc.eval_surface(pos4[:nb], dir4[:nb], ref[:nb])
        bp, bn = np.zeros((nb, 4), np.float32), np.zeros((nb, 4), np.float32)
        bp[:, :3], bn[:, :3] = surf[:, 0:3], surf[:, 3:6]
        bn[np.abs(bn[:, :3]).sum(axis=1) == 0, 1] = 1.0            # rays that missed: any unit normal
        dk, vals, r2 = brng.uniform(-1, 1, (nb, 2)).astype(np.float32), brng.uniform(0, 1, nb).astype(np.float32), brng.uniform(0, 1, (nb, 2)).astype(np.float32)
        fwd, pdf, cam, mut = np.empty((nb, 16), np.float32), np.empty((nb, 4), np.float32), np.empty((nb, 8), np.float32), np.empty(nb, np.float32)
        lib.emu_bidir(C.byref(orc.s), nb, p(ids), p(r4), p(ct), p(bp), p(bn), p(dk), p(vals), p(r2), 64.0, 1024.0, p(fwd), p(pdf), p(cam), p(mut))
        for got, want, what in ((fwd, orc.light_sample_forward(ids, r4), "LightSampleForward"), (pdf, orc.light_pdf_fwd(ids, ct), "lightPdfFwd"),
                                (cam, orc.camera_connect(bp, bn, dk), "camera connection"), (mut, orc.mutate_kelemen(vals, r2, 64.0, 1024.0), "MutateKelemen")):
            bad = ~np.isclose(got, want, rtol=2e-6, atol=2e-6, equal_nan=True)   # a sky dome takes the area-light branch, as in the reference: NaN on both sides
            assert not bad.any(), "%s differs from the oracle on %s: %d values, columns %s, e.g. %s vs %s" % (what, name, bad.sum(), np.unique(np.nonzero(bad.reshape(len(got), -1))[1]), got[bad][:4], want[bad][:4])
        # IntegratorMMLT::F through the wavefront stage functions against the oracle's restatement: every split of d = 1..5
        mrng = np.random.default_rng(41)
        md = np.repeat(np.arange(1, 6), 600).astype(np.int32)
        mxv = mrng.uniform(0, 1, (len(md), 12 + 10 * 5)).astype(np.float32)
        mout = np.empty((len(md), 8), np.float32)
        lib.emu_mmlt_f(C.byref(orc.s), len(md), p(md), p(mxv), mxv.shape[1], p(mout))
        mref = orc.mmlt_f(md, mxv)
        assert (mout[:, 3:6] == mref[:, 3:6]).all(), "MMLT F: pixel or split differs from the oracle on " + name
        mbad = ~np.isclose(mout, mref, rtol=2e-5, atol=1e-7)
        assert not mbad.any(), "MMLT F differs from the oracle on %s: %d values, rows %s" % (name, mbad.sum(), np.nonzero(mbad.any(axis=1))[0][:8])
        assert (mref[:, 7] > 0).mean() > 0.05, "MMLT F is zero nearly everywhere on " + name
        # ... and the Markov chains (InitialSamplePS, MutatePrimarySpace, accept / contribute) against the oracle's, 5 steps of 512 chains
        cd = mrng.integers(2, 5, 512).astype(np.int32)
        g_o = orc.mmlt_chain_gens(len(cd), 31)
        x_o = orc.mmlt_fresh(g_o, cd, 4)
        g_e, x_e = g_o.copy(), x_o.copy()
        img_o, ch_o, acc_o = orc.mmlt_run(cd, g_o, x_o, 5)
        img_e, ch_e, acc_e = np.zeros((h, w, 4), np.float32), np.zeros((len(cd), 6), np.float32), np.zeros(len(cd), np.int32)
        lib.emu_mmlt_run(C.byref(orc.s), len(cd), p(g_e), p(cd), 5, w, p(img_e), p(ch_e), p(x_e), x_e.shape[1], p(acc_e))
        assert (g_e == g_o).all() and (acc_e == acc_o).all() and np.allclose(x_e, x_o, atol=1e-7), "chains differ from the oracle's on " + name
        assert np.allclose(ch_e, ch_o, rtol=2e-5, atol=1e-7) and np.allclose(img_e, img_o, rtol=1e-4, atol=1e-6), "chain contributions differ on " + name
        # IHWLayer::EvalGBuffer through hk_gbuffer.h against the oracle's gbufferEval, whole frame
        if name in ("test_42", "atrium_transl_small"):
            from test_golden_ref import check_gbuffer
            e1, e2, eraw = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32), np.zeros((h, w, 14), np.float32)
            lib.emu_gbuffer(C.byref(orc.s), w, h, p(e1), p(e2), p(eraw))
            check_gbuffer((e1, e2, eraw), orc.gbuffer(), frac=0.005)
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
    deep_chain_case(lib, p)
    print("EMU_OK worst %.3g" % worst)


if __name__ == "__main__":
    main()
