"""hydracore_amd -- MI355X-native wavefront path-tracing core behind HydraCore's IHWLayer boundary.

The product is two in-tree shared objects built by ``make`` / ``__graft_entry__.build()``:

* ``lib/libhydra_hip.so``  -- hand-written gfx950 HIP kernels behind the C-ABI of ``include/hydra_hip.h``
* ``lib/libhydra_host.so`` -- C++ host layer: ``HipHWLayer`` (IHWLayer-shaped adapter), scene front end, BVH4 builder

This package only binds them with ctypes for the test / bench harness.  There is no CPU fallback: if the
libraries are missing the import fails, and without a HIP device ``HipCore``/``HostScene(use_hip=True)`` raise.
"""
from .capi import (HipCore, HostScene, HydraError, LiteHit, RaysStat, lib_dir, load_hip_library,
                   load_host_library, C_ABI_SYMBOLS)

__all__ = ["HipCore", "HostScene", "HydraError", "LiteHit", "RaysStat", "lib_dir", "load_hip_library",
           "load_host_library", "C_ABI_SYMBOLS"]
