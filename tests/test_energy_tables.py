"""The layer's own multi-scattering energy tables (hydracore_amd/csrc/hydra_bake.hip, baked on the device; tests/golden/energy_tables.npz is
such a bake) against the data the reference ships: bakeBrdfEnergy/MSTablesGGX2017.cpp (EssGgx2017Table, 64 x 64) and MSTablesTransp.cpp
(EssTranspGgx, 64^3), read here AS TEXT -- numeric literals between the braces -- where the reference tree is present (this container only);
nothing of them is copied into the repository.

What agreement to expect: the reference's tables are Monte-Carlo estimates (bakeBrdfEnergy/bakeBrdf.cpp:321-437: Sobol points over the whole
table, ~1 M landing in a GGX cell, ~16 k in a transparency cell; a running mean that counts its zero start as a sample), this build integrates
the same integrand over each cell with 16 384 / 2 048 fixed points.  Measured: GGX mean |difference| 4e-4, largest 2.2e-3; transparency mean
5e-4, 99th percentile 4e-3, largest 1e-2 (the reference's own noise); the bounds below leave a factor of 2-3."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT

REF_DIR = "/root/reference/bakeBrdfEnergy"
FIX = os.path.join(ROOT, "tests", "golden", "energy_tables.npz")


def literals(path, count):
    text = open(path).read()
    body = text[text.index("{") + 1: text.index("}")]
    v = np.array([int(t) for t in re.findall(r"\d+", body)], np.int64)
    assert v.size == count, (path, v.size)
    return v


@pytest.mark.skipif(not os.path.exists(FIX), reason="tests/golden/energy_tables.npz not generated yet (tests/golden/make_golden.py energy, on the GPU box)")
def test_tables_are_well_formed():
    t = np.load(FIX)
    ggx, transp = t["ggx"].astype(np.float64) / 65535.0, t["transp"].astype(np.float64) / 65535.0
    assert ggx.shape == (64, 64) and transp.shape == (64, 64, 64)
    assert ggx.min() > 0.3 and ggx.max() <= 1.0
    assert ggx[0].min() > 0.8                              # a smooth surface loses nothing to masking, except at grazing angles
    assert (np.diff(ggx[8:, 32:], axis=0) < 2e-3).all()    # energy falls as the surface gets rougher
    assert transp.max() <= 1.0 and transp.mean() > 0.3


@pytest.mark.skipif(not (os.path.exists(REF_DIR) and os.path.exists(FIX)), reason="needs the reference tree (read as text) and the baked fixture")
def test_own_bake_against_the_reference_s_literals():
    t = np.load(FIX)
    ref2 = literals(os.path.join(REF_DIR, "MSTablesGGX2017.cpp"), 4096).reshape(64, 64) / 65535.0
    ref3 = literals(os.path.join(REF_DIR, "MSTablesTransp.cpp"), 262144).reshape(64, 64, 64) / 65535.0
    got2, got3 = t["ggx"] / 65535.0, t["transp"] / 65535.0
    d2 = np.abs(got2 - ref2)
    assert d2.mean() < 1e-3 and np.percentile(d2, 99) < 3e-3 and d2.max() < 6e-3, (d2.mean(), np.percentile(d2, 99), d2.max())
    d3 = np.abs(got3 - ref3)
    assert d3.mean() < 1.5e-3 and np.percentile(d3, 99) < 1e-2 and d3.max() < 3e-2, (d3.mean(), np.percentile(d3, 99), d3.max())
    assert abs(got3.mean() - ref3.mean()) < 2e-3
    # what the shading reads is 1 + colour (1 - Ess) / Ess: bound the compensation factor too, where it matters (rough surfaces)
    c2, r2 = (1 - got2) / np.maximum(got2, 1e-6), (1 - ref2) / np.maximum(ref2, 1e-6)
    assert np.abs(c2 - r2)[16:].mean() < 4e-3
