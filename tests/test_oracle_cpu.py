"""CPU: properties of the oracle itself (the reference ships no golden vectors for this path; function-level vectors
produced by the reference's own OpenCL build live in tests/golden/ and are checked in test_golden_ref.py)."""
import numpy as np

from conftest import make_oracle, random_rays

M32 = 0xFFFFFFFF


def py_random_gen(seed, draws):
    """independent re-implementation of crandom.h:20-63 with python ints"""
    s = seed & M32
    x = (s * ((s * s * 15731 + 74323) & M32) + 871483) & M32
    y = (s * ((s * s * 13734 + 37828) & M32) + 234234) & M32

    def nxt():
        nonlocal x, y
        v = (x * 17 + y * 13123) & M32
        x = ((v << 13) ^ v) & M32
        y = (y ^ (v << 7)) & M32
        return v
    for _ in range(seed % 7 if seed >= 0 else 0):
        nxt()
    out = []
    for _ in range(draws):
        v = nxt()
        vals = []
        for a, b_, c in ((15731, 74323, 871483), (13734, 37828, 234234), (11687, 26461, 137589), (15707, 789221, 1376312589)):
            t = (v * ((v * v * a + b_) & M32) + c) & M32
            vals.append(np.float32(t) * np.float32(1.0 / 4294967296.0))
        out.append(vals)
    return np.array(out, np.float32), (x, y)


def test_random_gen_bit_exact(t42_small):
    _, b = t42_small
    orc = make_oracle(b)
    seeds = [0, 1, 7, 777, 123456, 2147483647]
    out, st = orc.random(seeds, 64)
    for i, sd in enumerate(seeds):
        ref, (x, y) = py_random_gen(sd, 64)
        assert (out[i].view(np.uint32) == ref.view(np.uint32)).all()
        assert (int(st[i, 0]), int(st[i, 1])) == (x, y)
    assert out.min() >= 0.0 and out.max() <= 1.0


def test_eye_rays_are_unit_and_cover_the_frustum(t224_small):
    _, b = t224_small
    orc = make_oracle(b)
    w, h = b["width"], b["height"]
    xy = np.array([[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1], [w // 2, h // 2]], np.int32)
    pos, dr = orc.make_eye_rays(xy, np.zeros((5, 4), np.float32))
    np.testing.assert_allclose(pos[:, :3], [[0, 0, 14]] * 5, atol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(dr[:, :3], axis=1), 1.0, atol=1e-6)
    assert dr[4, 2] < -0.999                                          # centre pixel looks down -z
    assert dr[0, 0] < 0 and dr[0, 1] < 0 and dr[3, 0] > 0 and dr[3, 1] > 0   # row 0 is the bottom of the image
    half = np.tan(np.deg2rad(22.5))
    assert abs(abs(dr[0, 1] / dr[0, 2]) - half * (1 - 1.0 / h)) < 2e-3


def test_surface_is_on_the_ray_and_frames_are_orthonormal(t224_small):
    _, b = t224_small
    orc = make_oracle(b)
    pos4, dir4 = random_rays(4000, 3)
    hits = orc.trace(pos4, dir4)
    surf = orc.eval_surface(pos4, dir4, hits)
    m = hits["primId"] != -1
    assert m.mean() > 0.5
    p = pos4[m, :3] + hits["t"][m, None] * dir4[m, :3]
    np.testing.assert_allclose(surf[m, 0:3], p, atol=2e-4)
    np.testing.assert_allclose(surf[m, 18], hits["t"][m], rtol=1e-4, atol=1e-4)
    for k in (3, 6, 9, 12):
        np.testing.assert_allclose(np.linalg.norm(surf[m, k:k + 3], axis=1), 1.0, atol=1e-4)
    assert ((surf[m, 6:9] * dir4[m, :3]).sum(1) <= 0.03).all()       # flat normal faces the ray (0.025 threshold)
    mat = surf[m, 17].view(np.int32)
    assert set(np.unique(mat)) <= {1, 6, 7, 8, 9, 10}
    assert (surf[~m, 17].view(np.int32) == -1).all()


def test_shadow_trace_is_closest_hit_in_range(t224_small):
    _, b = t224_small
    orc = make_oracle(b)
    pos4, dir4 = random_rays(3000, 5)
    hits = orc.trace(pos4, dir4)
    tfar = np.random.default_rng(1).uniform(0.5, 20.0, len(pos4)).astype(np.float32)
    vis = orc.shadow_trace(pos4, dir4, tfar)
    expect = np.where((hits["primId"] != -1) & (hits["t"] > 0) & (hits["t"] < tfar), 0.0, 1.0)
    assert (vis == expect).all()


def test_render_is_deterministic_and_tile_partition_is_exact(t224_small):
    _, b = t224_small
    orc = make_oracle(b)
    full, rays, _ = orc.render(2, seed=777, sum_mode=True)
    again, rays2, _ = orc.render(2, seed=777, sum_mode=True, threads=1)
    assert rays == rays2 and (full == again).all()                   # independent of the thread count
    parts = [orc.render(2, seed=777, sum_mode=True, rank=r, world=3, tile=16) for r in range(3)]
    assert sum(p[1] for p in parts) == rays
    assert (sum(p[0] for p in parts) == full).all()                  # disjoint supports: bit-identical to the 1-rank frame
    mean_img, _, _ = orc.render(2, seed=777, sum_mode=False)
    np.testing.assert_allclose(mean_img, full / 2.0, rtol=1e-6, atol=1e-7)
    assert full[..., :3].mean() > 0.05 and np.isfinite(full).all()


def test_path_trace_equals_render_pass(t224_small):
    _, b = t224_small
    orc = make_oracle(b)
    w, h = b["width"], b["height"]
    gens = orc.init_generators(777)
    img, _, gens_after = orc.render(1, seed=777, sum_mode=True)
    # replay by hand: lens draw, eye ray, PathTrace with the same per-pixel generator
    g = gens.copy()
    offs = np.empty((w * h, 4), np.float32)
    import ctypes as C
    tmp = np.empty(4, np.float32)
    for i in range(w * h):
        orc.lib.orc_rnd_float4(g[i].ctypes.data_as(C.c_void_p), tmp.ctypes.data_as(C.c_void_p))
        offs[i] = np.float32(-1.0) + np.float32(2.0) * tmp
    ys, xs = np.divmod(np.arange(w * h), w)
    pos, dr = orc.make_eye_rays(np.stack([xs, ys], 1).astype(np.int32), offs)
    col, g2 = orc.path_trace(pos, dr, g)
    assert (col[:, :3].reshape(h, w, 3) == img[..., :3]).all()
    assert (g2 == gens_after).all()


def test_energy_is_bounded_in_closed_box(t42_small):
    """test_42 box: radiance leaving towards the camera cannot exceed the emitter radiance"""
    _, b = t42_small
    orc = make_oracle(b)
    img, _, _ = orc.render(4, seed=1)
    assert img[..., :3].max() <= 31.4 * 1.0001 and img[..., :3].min() >= 0.0


def test_mmlt_chains_converge_to_the_path_tracer(built):
    """row f3 property: the image the Markov chains of IntegratorMMLT build (paths of 2..4 segments, every split, MIS over the splits,
    scaled by the separately estimated average brightness, CPUExp_Integrators_MMLT.cpp:358-461, 548-552) is the path tracer's image of
    the same path lengths, PT(trace depth 4) - PT(trace depth 1)"""
    from conftest import host_scene, make_oracle

    def pt(depth, spp=192):
        _, b = host_scene("test_42", 96, 96, depth, 0)          # the front end stores trace depth = depth + 1
        return make_oracle(b).render(spp, seed=777)[0][..., :3]
    ref = pt(3) - pt(0)
    _, b = host_scene("test_42", 96, 96, 3, 0)
    orc = make_oracle(b)
    rng = np.random.default_rng(1)
    avg = np.zeros(5)
    for d in (2, 3, 4):       # DoPassEstimateAvgBrightness
        x = rng.uniform(0, 1, (100000, 12 + 10 * d)).astype(np.float32)
        avg[d] = orc.mmlt_f(np.full(len(x), d, np.int32), x)[:, 7].mean() * (d + 1)
    assert abs(avg.sum() - (0.33334 * ref.sum(axis=2)).mean()) < 0.03 * avg.sum()      # sum over path lengths of E[F (d + 1)] = mean radiance
    n = 4096
    depth = rng.choice(5, size=n, p=avg / avg.sum()).astype(np.int32)
    gens = orc.mmlt_chain_gens(n, 1234)
    x = orc.mmlt_fresh(gens, depth, 4)
    img, ch, acc = orc.mmlt_run(depth, gens, x, 400)
    assert 0.5 < acc.mean() / 400 < 0.95
    ind = img[..., :3] * (avg.sum() / (0.33334 * img[..., :3].sum(axis=2).mean()))

    def down(a, f=8):
        return a.reshape(96 // f, f, 96 // f, f, 3).mean(axis=(1, 3))
    a, r = down(ind), down(ref)
    assert np.corrcoef(a.ravel(), r.ravel())[0, 1] > 0.99
    assert np.abs(a - r).sum() / r.sum() < 0.08
