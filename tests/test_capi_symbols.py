"""CPU: the C-ABI library loads without a GPU and exports every symbol include/hydra_hip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "hydra_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hydra_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = header_symbols()
    for must in ("hydra_hip_create", "hydra_hip_upload_globals", "hydra_hip_upload_bvh", "hydra_hip_trace_pass",
                 "hydra_hip_get_hdr_image", "hydra_hip_stage_trace"):
        assert must in syms


def test_library_exports_every_declared_symbol(built):
    from hydracore_amd import C_ABI_SYMBOLS, load_hip_library
    lib = load_hip_library()
    syms = header_symbols()
    assert sorted(C_ABI_SYMBOLS) == syms, "python binding list and header drifted apart"
    for s in syms:
        assert hasattr(lib, s), s


def test_create_fails_loudly_without_device(built):
    """No GPU in the CPU container: create must fail with ENODEV and a message, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hydracore_amd import HipCore, HydraError
    with pytest.raises(HydraError) as e:
        HipCore(64, 64)
    assert "no HIP device" in str(e.value) or "hipGetDeviceCount" in str(e.value) or "device" in str(e.value)


def test_host_library_links_no_oracle(built):
    """the product libraries must not depend on the oracle"""
    import subprocess
    from hydracore_amd import lib_dir
    for name in ("libhydra_hip.so", "libhydra_host.so"):
        out = subprocess.run(["ldd", os.path.join(lib_dir(), name)], capture_output=True, text=True).stdout
        assert "oracle" not in out
        syms = subprocess.run(["nm", "-D", os.path.join(lib_dir(), name)], capture_output=True, text=True).stdout
        assert "orc_" not in syms
