"""Shared fixtures.  `gpu` marks tests that need a real MI355X; everything else runs on CPU in this container."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def scene_path(name):
    return os.path.join(SCENES, name)


@pytest.fixture(scope="session")
def built():
    """The three shared objects; built on demand so a fresh checkout can run the CPU suite."""
    import __graft_entry__ as ge
    ge.build(ref=False)
    return True


_cache = {}


def host_scene(name, w, h, depth, dof=0):
    """host-blob layer (no device) + the numpy copies of its buffers, cached per configuration"""
    from hydracore_amd import HostScene
    key = (name, w, h, depth, dof)
    if key not in _cache:
        sc = HostScene(scene_path(name), w, h, trace_depth=depth, enable_dof=dof, use_hip=False)
        _cache[key] = (sc, sc.buffers())
    return _cache[key]


@pytest.fixture(scope="session")
def t224_small(built):
    return host_scene("test_224", 96, 96, 4)


@pytest.fixture(scope="session")
def t42_small(built):
    return host_scene("test_42", 96, 96, 4)


def make_oracle(buffers):
    from oracle_lib import Oracle
    return Oracle(buffers)


def random_rays(n, seed, center=(0.0, 0.0, 0.0), radius=9.0):
    """rays from random points on a sphere around the scene towards random points near its centre"""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    org = np.asarray(center) + radius * d
    tgt = np.asarray(center) + rng.uniform(-3.5, 3.5, size=(n, 3))
    dr = tgt - org
    dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    pos4 = np.zeros((n, 4), np.float32)
    dir4 = np.zeros((n, 4), np.float32)
    pos4[:, :3] = org
    dir4[:, :3] = dr
    return pos4, dir4
