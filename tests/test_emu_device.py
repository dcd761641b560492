"""CPU: the product's HIP device functions compiled for the host (-DHK_HOST_EMU) under AddressSanitizer + UBSan, run
on the parity inputs and compared with the oracle (bit-identical on CPU: same libm).  GPU sanitizers are not available
on the pool, so this is where out-of-bounds reads / UB in the device code are hunted."""
import os
import subprocess
import sys

from conftest import ROOT

EMU = os.path.join(ROOT, "tests", "emu")


def test_device_code_is_clean_under_asan_ubsan_and_matches_oracle(built):
    so = os.path.join(EMU, "libemu_device_asan.so")
    srcs = [os.path.join(EMU, "emu_device.cpp"), os.path.join(EMU, "emu_integrator.h")] + \
           [os.path.join(ROOT, "hydracore_amd", "csrc", f) for f in ("hk_common.h", "hk_trace.h", "hk_shading.h", "hk_bidir.h", "hk_gbuffer.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off",
                               "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I/opt/rocm/include",
                               "-fPIC", "-shared", "emu_device.cpp", "-o", so], cwd=EMU)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, os.path.join(EMU, "run_emu.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "EMU_OK" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
