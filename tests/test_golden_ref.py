"""CPU: the oracle against golden vectors produced by the REFERENCE's own code (tests/golden/make_golden.py ran the
reference's inline functions, compiled from /root/reference for gfx950, on the GPU box).  This is what pins the oracle.

Tolerances: integer results exact.  The reference binary is an OpenCL build (device libm, LiteMath replaced by the
OpenCL built-ins dot/cross/normalize/length), so floats that go through those built-ins or through sin/cos/pow may
differ in the last bits from the oracle's glibc build: stated per assert."""
import os

import numpy as np
import pytest

from conftest import ROOT, host_scene, make_oracle

GOLD = os.path.join(ROOT, "tests", "golden")


def load(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.skip("%s not generated yet (tests/golden/make_golden.py needs the GPU box)" % name)
    return np.load(path)


def test_rng_matches_reference_bit_for_bit(t42_small):
    g = load("rng.npz")
    orc = make_oracle(t42_small[1])
    out, st = orc.random(g["seeds"], 64)
    assert (out.view(np.uint32) == g["out"].view(np.uint32)).all()
    assert (st == g["state"]).all()


@pytest.mark.parametrize("name", ["test_224", "test_42", "atrium_small", "atrium_sky_small"])
def test_oracle_matches_reference_functions(name, built):
    g = load("ref_%s.npz" % name)
    _, b = host_scene(name, int(g["width"]), int(g["height"]), int(g["depth"]), int(g["dof"]))
    orc = make_oracle(b)
    # P1 MakeRandEyeRay
    pos, dr = orc.make_eye_rays(g["eye_xy"], g["eye_offs"])
    np.testing.assert_allclose(pos[:, :3], g["eye_pos"][:, :3], atol=2e-6)
    np.testing.assert_allclose(dr[:, :3], g["eye_dir"][:, :3], atol=2e-6)
    # T1 BVH4InstTraverse + Moeller-Trumbore: same triangle, same distance
    hits = orc.trace(g["ray_pos"], g["ray_dir"])
    ref = g["hits"]
    same = (hits["primId"] == ref["primId"]) & (hits["instId"] == ref["instId"]) & (hits["geomId"] == ref["geomId"])
    assert same.mean() >= 0.9999, same.mean()          # OpenCL dot/cross may round differently on an edge-on triangle
    m = same & (ref["primId"] != -1)
    np.testing.assert_allclose(hits["t"][m], ref["t"][m], rtol=3e-6)
    # H1 surfaceEvalLS + instance transform
    surf = orc.eval_surface(g["ray_pos"], g["ray_dir"], ref)
    rs = g["surf"]
    assert (surf[:, 17].view(np.int32) == rs[:, 17].view(np.int32)).all()
    assert (surf[:, 20] == rs[:, 20]).mean() > 0.9995
    ok = surf[:, 20] == rs[:, 20]
    np.testing.assert_allclose(surf[ok, :17], rs[ok, :17], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(surf[ok, 18:20], rs[ok, 18:20], rtol=1e-4, atol=1e-6)
    # whole paths through every shading function (emission, light sampling, materialEval, BxDF sampling, flags)
    col, gens = orc.path_trace(g["path_pos"], g["path_dir"], g["path_gens"])
    rc, rg = g["path_color"], g["path_gens_out"]
    same_draws = (gens == rg).all(axis=1)
    assert same_draws.mean() > 0.995, same_draws.mean()
    assert (col[same_draws, 3] == rc[same_draws, 3]).all()
    err = np.abs(col[:, :3] - rc[:, :3])
    tol = 2e-4 * np.maximum(np.abs(rc[:, :3]), 1.0)
    bad = (err > tol).any(axis=1)
    assert bad.mean() < 0.005, bad.mean()
    assert abs(col[:, :3].mean() - rc[:, :3].mean()) < 2e-3 * rc[:, :3].mean()
