"""ctypes bindings: include/hydra_hip.h (the C-ABI boundary) and the host layer's small C surface."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_dir():
    """in-tree build directory; HYDRA_AMD_LIB_DIR selects another build of the same two libraries (tuning A/B runs)"""
    return os.environ.get("HYDRA_AMD_LIB_DIR") or os.path.join(_HERE, "lib")


class HydraError(RuntimeError):
    pass


class LiteHit(C.Structure):          # HydraLiteHit, include/hydra_layouts.h
    _fields_ = [("t", C.c_float), ("primId", C.c_int32), ("instId", C.c_int32), ("geomId", C.c_int32)]


class RaysStat(C.Structure):         # HydraRaysStat
    _fields_ = [("raysPerSec", C.c_float), ("traversalTimeMs", C.c_float), ("samLightTimeMs", C.c_float),
                ("shadowTimeMs", C.c_float), ("shadeTimeMs", C.c_float), ("bounceTimeMs", C.c_float),
                ("evalHitMs", C.c_float), ("nextBounceMs", C.c_float), ("raygenTimeMs", C.c_float),
                ("accumTimeMs", C.c_float), ("passTimeMs", C.c_float), ("traceTimePerCent", C.c_int32),
                ("extensionRays", C.c_uint64), ("shadowRays", C.c_uint64), ("samples", C.c_uint64),
                ("traceLaunches", C.c_uint64), ("shadowLaunches", C.c_uint64)]


class StatePlan(C.Structure):       # HydraStatePlan, include/hydra_hip.h
    _fields_ = [("samples_in_flight", C.c_int32), ("pad_", C.c_int32), ("owned_pixels", C.c_int64), ("paths", C.c_int64),
                ("segments", C.c_int64), ("segment_capacity", C.c_int64), ("path_state_bytes", C.c_int64),
                ("generator_bytes", C.c_int64), ("contrib_bytes", C.c_int64), ("owned_map_bytes", C.c_int64), ("total_bytes", C.c_int64)]


LITE_HIT_DTYPE = np.dtype([("t", np.float32), ("primId", np.int32), ("instId", np.int32), ("geomId", np.int32)])

# every entry point include/hydra_hip.h declares (checked by tests/test_capi_symbols.py against the header text)
C_ABI_SYMBOLS = [
    "hydra_hip_create", "hydra_hip_destroy", "hydra_hip_last_error", "hydra_hip_device_name", "hydra_hip_resize",
    "hydra_hip_available_memory", "hydra_hip_finish", "hydra_hip_upload_globals", "hydra_hip_update_globals_header",
    "hydra_hip_upload_storage", "hydra_hip_upload_bvh", "hydra_hip_set_bvh_trees_num", "hydra_hip_upload_instances",
    "hydra_hip_upload_remap_lists", "hydra_hip_set_tile_partition", "hydra_hip_tile_owners", "hydra_hip_plan_render_state", "hydra_hip_set_external_accumulator",
    "hydra_hip_init_path_tracing", "hydra_hip_clear_accumulated_color", "hydra_hip_trace_pass", "hydra_hip_set_spp",
    "hydra_hip_get_spp", "hydra_hip_get_hdr_image", "hydra_hip_get_ldr_image", "hydra_hip_get_accumulator", "hydra_hip_get_rays_stat",
    "hydra_hip_reset_perf_counters", "hydra_hip_enable_stage_timing", "hydra_hip_get_stage_times_per_bounce", "hydra_hip_set_option", "hydra_hip_get_option", "hydra_hip_enable_traversal_counters",
    "hydra_hip_get_traversal_counters", "hydra_hip_get_traversal_oob", "hydra_hip_stage_trace_totals", "hydra_hip_stage_make_eye_rays",
    "hydra_hip_stage_trace", "hydra_hip_stage_shadow_trace", "hydra_hip_stage_eval_surface",
    "hydra_hip_stage_shade_point", "hydra_hip_stage_bounce", "hydra_hip_stage_path_trace", "hydra_hip_stage_random", "hydra_hip_bench_trace",
    "hydra_hip_comm_unique_id", "hydra_hip_comm_init", "hydra_hip_comm_gather_frame", "hydra_hip_comm_reduce_frame", "hydra_hip_comm_destroy",
    "hydra_hip_stage_pack_unpack", "hydra_hip_stage_light_sample_forward", "hydra_hip_stage_light_pdf_fwd", "hydra_hip_stage_camera_connect",
    "hydra_hip_stage_mutate_kelemen", "hydra_hip_stage_mmlt_f", "hydra_hip_mmlt_begin", "hydra_hip_mmlt_pass", "hydra_hip_mmlt_get_image", "hydra_hip_mmlt_reset_image",
    "hydra_hip_mmlt_get_state", "hydra_hip_mmlt_end", "hydra_hip_sbdpt_pass", "hydra_hip_sbdpt_get_image", "hydra_hip_eval_gbuffer", "hydra_hip_normal_map_from_displacement", "hydra_hip_image_last_error", "hydra_hip_bake_energy_tables", "hydra_hip_bake_last_error", "hydra_hip_bvh_build_mesh", "hydra_hip_bvh_build_mesh_ex", "hydra_hip_bvh_last_error",
    "hydra_hip_proctex_compile", "hydra_hip_proctex_check", "hydra_hip_stage_proctex", "hydra_hip_stage_set_proctex", "hydra_hip_stage_environment", "hydra_hip_stage_mmlt_accept",
]

_hip = None
_host = None


def load_hip_library():
    """dlopen libhydra_hip.so; raises (never falls back) when it has not been built."""
    global _hip
    if _hip is not None:
        return _hip
    path = os.path.join(lib_dir(), "libhydra_hip.so")
    if not os.path.exists(path):
        raise HydraError("libhydra_hip.so is missing: run `make` (or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, i32, f32p, u32p, i32p = C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
    sz = C.c_size_t
    sig = {
        "hydra_hip_create": ([i32, i32, i32, i32, C.POINTER(vp)], i32),
        "hydra_hip_destroy": ([vp], i32),
        "hydra_hip_last_error": ([vp], C.c_char_p),
        "hydra_hip_device_name": ([vp, C.c_char_p, i32], i32),
        "hydra_hip_resize": ([vp, i32, i32], i32),
        "hydra_hip_available_memory": ([vp, C.POINTER(sz), C.POINTER(sz)], i32),
        "hydra_hip_finish": ([vp], i32),
        "hydra_hip_upload_globals": ([vp, vp, sz], i32),
        "hydra_hip_update_globals_header": ([vp, vp, sz], i32),
        "hydra_hip_upload_storage": ([vp, i32, vp, sz], i32),
        "hydra_hip_upload_bvh": ([vp, i32, vp, i32, vp, i32, vp, i32, i32], i32),
        "hydra_hip_set_bvh_trees_num": ([vp, i32], i32),
        "hydra_hip_upload_instances": ([vp, vp, vp, i32], i32),
        "hydra_hip_upload_remap_lists": ([vp, vp, i32, vp, i32, vp, i32], i32),
        "hydra_hip_set_tile_partition": ([vp, i32, i32, i32], i32),
        "hydra_hip_set_external_accumulator": ([vp, vp, sz], i32),
        "hydra_hip_tile_owners": ([i32, i32, i32, i32, vp, i32], i32),
        "hydra_hip_plan_render_state": ([i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(StatePlan), C.c_char_p, i32], i32),
        "hydra_hip_init_path_tracing": ([vp, i32], i32),
        "hydra_hip_clear_accumulated_color": ([vp], i32),
        "hydra_hip_trace_pass": ([vp, i32], i32),
        "hydra_hip_set_spp": ([vp, C.c_float], i32),
        "hydra_hip_get_spp": ([vp], C.c_float),
        "hydra_hip_get_hdr_image": ([vp, vp, i32, i32], i32),
        "hydra_hip_get_ldr_image": ([vp, vp, i32, i32], i32),
        "hydra_hip_get_accumulator": ([vp, vp, i32, i32], i32),
        "hydra_hip_get_rays_stat": ([vp, C.POINTER(RaysStat)], i32),
        "hydra_hip_reset_perf_counters": ([vp], i32),
        "hydra_hip_enable_stage_timing": ([vp, i32], i32),
        "hydra_hip_get_stage_times_per_bounce": ([vp, vp, i32], i32),
        "hydra_hip_set_option": ([vp, C.c_char_p, i32], i32),
        "hydra_hip_get_option": ([vp, C.c_char_p, C.POINTER(i32)], i32),
        "hydra_hip_enable_traversal_counters": ([vp, i32], i32),
        "hydra_hip_get_traversal_counters": ([vp, vp, i32], i32),
        "hydra_hip_get_traversal_oob": ([vp, C.POINTER(C.c_uint64)], i32),
        "hydra_hip_stage_trace_totals": ([vp, i32, vp, vp, vp, vp], i32),
        "hydra_hip_stage_make_eye_rays": ([vp, i32, vp, vp, vp, vp], i32),
        "hydra_hip_stage_trace": ([vp, i32, vp, vp, vp, vp], i32),
        "hydra_hip_stage_shadow_trace": ([vp, i32, vp, vp, vp, vp], i32),
        "hydra_hip_stage_eval_surface": ([vp, i32, vp, vp, vp, vp], i32),
        "hydra_hip_stage_shade_point": ([vp, i32, vp, vp, vp, vp, vp, vp], i32),
        "hydra_hip_proctex_compile": ([vp, C.c_char_p, C.c_size_t], i32),
        "hydra_hip_stage_environment": ([vp, i32, vp, vp, vp], i32),
        "hydra_hip_stage_mmlt_accept": ([vp, i32, vp, vp, vp, C.c_float, vp], i32),
        "hydra_hip_proctex_check": ([C.c_char_p, C.c_size_t], i32),
        "hydra_hip_stage_proctex": ([vp, i32, i32, vp, vp, vp, vp, vp], i32),
        "hydra_hip_stage_set_proctex": ([vp, i32, i32, vp, vp], i32),
        "hydra_hip_stage_bounce": ([vp, i32, i32, i32, vp, vp, vp, vp, vp, vp], i32),
        "hydra_hip_stage_path_trace": ([vp, i32, vp, vp, vp, vp], i32),
        "hydra_hip_stage_random": ([vp, i32, vp, i32, vp, vp], i32),
        "hydra_hip_bench_trace": ([vp, i32, vp, vp, i32, i32, f32p], i32),
        "hydra_hip_comm_unique_id": ([vp, vp], i32),
        "hydra_hip_comm_init": ([vp, vp, i32, i32], i32),
        "hydra_hip_comm_gather_frame": ([vp, i32], i32),
        "hydra_hip_comm_reduce_frame": ([vp, i32], i32),
        "hydra_hip_comm_destroy": ([vp], i32),
        "hydra_hip_stage_pack_unpack": ([vp, vp, i32, i32], i32),
        "hydra_hip_stage_light_sample_forward": ([vp, i32, vp, vp, vp], i32),
        "hydra_hip_stage_light_pdf_fwd": ([vp, i32, vp, vp, vp], i32),
        "hydra_hip_stage_camera_connect": ([vp, i32, vp, vp, vp, vp], i32),
        "hydra_hip_stage_mutate_kelemen": ([vp, i32, vp, vp, C.c_float, C.c_float, vp], i32),
        "hydra_hip_stage_mmlt_f": ([vp, i32, vp, vp, i32, vp], i32),
        "hydra_hip_mmlt_begin": ([vp, i32, i32, i32, i32, i32], i32),
        "hydra_hip_mmlt_pass": ([vp, i32], i32),
        "hydra_hip_mmlt_get_image": ([vp, vp, i32, i32, vp], i32),
        "hydra_hip_mmlt_reset_image": ([vp], i32),
        "hydra_hip_mmlt_get_state": ([vp, vp, vp, vp, vp], i32),
        "hydra_hip_mmlt_end": ([vp], i32),
        "hydra_hip_sbdpt_pass": ([vp, i32], i32),
        "hydra_hip_sbdpt_get_image": ([vp, vp, i32, i32, vp], i32),
        "hydra_hip_eval_gbuffer": ([vp, vp, vp, i32, i32, vp, i32, vp], i32),
        "hydra_hip_bvh_build_mesh": ([i32, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp], i32),
        "hydra_hip_bvh_build_mesh_ex": ([i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp], i32),
        "hydra_hip_bvh_last_error": ([], C.c_char_p),
        "hydra_hip_normal_map_from_displacement": ([i32, i32, i32, vp, C.c_float, i32, C.c_float, vp, vp], i32),
        "hydra_hip_bake_energy_tables": ([i32, vp, vp, vp], i32),
        "hydra_hip_bake_last_error": ([], C.c_char_p),
        "hydra_hip_image_last_error": ([], C.c_char_p),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _hip = lib
    return lib


def plan_render_state(width, height, rank=0, world=1, tile=64, samples_in_flight=0, queue_segments=32, fused_bounce=1):
    """hydra_hip_plan_render_state: sizes of one rank's render state (host arithmetic, no device); raises HydraError on a limit"""
    lib = load_hip_library()
    plan, err = StatePlan(), C.create_string_buffer(256)
    rc = lib.hydra_hip_plan_render_state(width, height, rank, world, tile, samples_in_flight, queue_segments, fused_bounce, C.byref(plan), err, 256)
    if rc != 0:
        raise HydraError("plan_render_state: %s" % err.value.decode())
    return {k: getattr(plan, k) for k, _ in StatePlan._fields_ if k != "pad_"}


def tile_owners(width, height, world, tile=64):
    """int32 [tilesY, tilesX]: the rank that owns each tile (hydra_hip_tile_owners: Morton order, round-robin)"""
    lib = load_hip_library()
    tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
    out = np.zeros((ty, tx), np.int32)
    if lib.hydra_hip_tile_owners(width, height, world, tile, out.ctypes.data_as(C.c_void_p), tx * ty) != 0:
        raise HydraError("tile_owners: bad arguments")
    return out


def load_host_library():
    global _host
    if _host is not None:
        return _host
    load_hip_library()   # libhydra_host.so links against it
    path = os.path.join(lib_dir(), "libhydra_host.so")
    if not os.path.exists(path):
        raise HydraError("libhydra_host.so is missing: run `make` (or __graft_entry__.build())")
    lib = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int
    lib.hydra_host_open_scene.argtypes = [C.c_char_p, i32, i32, i32, i32, i32, i32, i32, C.c_char_p, i32]
    lib.hydra_host_open_scene.restype = vp
    lib.hydra_host_close_scene.argtypes = [vp]
    lib.hydra_host_close_scene.restype = None
    lib.hydra_host_have_inst_tree.argtypes = [vp, i32]
    lib.hydra_host_have_inst_tree.restype = i32
    for n in ("hydra_host_width", "hydra_host_height", "hydra_host_unsupported", "hydra_host_have_inst", "hydra_host_trees_num"):
        getattr(lib, n).argtypes = [vp]
        getattr(lib, n).restype = i32
    for n in ("hydra_host_log", "hydra_host_last_error", "hydra_host_proctex_program"):
        getattr(lib, n).argtypes = [vp]
        getattr(lib, n).restype = C.c_char_p
    lib.hydra_host_get_buffer.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.hydra_host_get_buffer.restype = i32
    lib.hydra_host_hip_handle.argtypes = [vp]
    lib.hydra_host_hip_handle.restype = vp
    lib.hydra_host_set_render_method.argtypes = [vp, C.c_char_p]
    lib.hydra_host_set_render_method.restype = i32
    lib.hydra_host_draw.argtypes = [vp, i32, i32]
    lib.hydra_host_draw.restype = i32
    lib.hydra_host_get_hdr.argtypes = [vp, vp, i32, i32]
    lib.hydra_host_get_hdr.restype = i32
    lib.hydra_host_get_spp.argtypes = [vp]
    lib.hydra_host_get_spp.restype = C.c_float
    lib.hydra_host_shared_image_open.argtypes = [vp, vp, i32, i32, i32]
    lib.hydra_host_shared_image_open.restype = vp
    lib.hydra_host_shared_image_stat.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(i32)]
    lib.hydra_host_shared_image_stat.restype = i32
    lib.hydra_host_shared_image_close.argtypes = [vp, vp]
    lib.hydra_host_shared_image_close.restype = None
    lib.hydra_host_eval_gbuffer.argtypes = [vp, vp, i32, i32, i32, vp, i32, C.POINTER(i32)]
    lib.hydra_host_eval_gbuffer.restype = i32
    lib.hydra_host_bvh_stats.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.hydra_host_bvh_stats.restype = i32
    _host = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f4(a, n):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.shape == (n, 4), a.shape
    return a


BUILD_NODE_DTYPE = np.dtype([("boxMin", np.float32, 3), ("first", np.int32), ("boxMax", np.float32, 3), ("count", np.int32), ("child", np.int32, 4)])


def bvh_build_mesh(vert4f, indices, leaf_max=2, device=0, method="ploc", radius=128):
    """GPU build of one mesh (hydra_hip_bvh_build_mesh_ex; method "ploc" or "lbvh"): -> (nodes (BUILD_NODE_DTYPE), triangle order, device ms)"""
    lib = load_hip_library()
    v, idx = np.ascontiguousarray(vert4f, np.float32).reshape(-1, 4), np.ascontiguousarray(indices, np.int32).ravel()
    tri = idx.size // 3
    nodes, order = np.zeros(2 * tri, BUILD_NODE_DTYPE), np.zeros(tri, np.int32)
    nn, npr, ms = C.c_int32(0), C.c_int32(0), C.c_float(0)
    rc = lib.hydra_hip_bvh_build_mesh_ex(device, _ptr(v), len(v), _ptr(idx), idx.size, leaf_max, {"lbvh": 0, "ploc": 1}[method], radius, _ptr(nodes), C.byref(nn), _ptr(order), C.byref(npr), C.byref(ms))
    if rc != 0:
        raise HydraError("bvh_build_mesh failed (%d): %s" % (rc, lib.hydra_hip_bvh_last_error().decode()))
    return nodes[:nn.value].copy(), order[:npr.value].copy(), ms.value


def normal_map_from_displacement(rgba, bump_amt, inv_height, smooth_lvl, device=0):
    """IHWLayer::NormalMapFromDisplacement on the device: uint8 [h, w, 4] height map -> (uint8 [h, w, 4] normal map, device ms)"""
    lib = load_hip_library()
    a = np.ascontiguousarray(rgba, np.uint8)
    out, ms = np.zeros_like(a), C.c_float(0)
    rc = lib.hydra_hip_normal_map_from_displacement(device, a.shape[1], a.shape[0], _ptr(a), float(bump_amt), int(bool(inv_height)), float(smooth_lvl), _ptr(out), C.byref(ms))
    if rc != 0:
        raise HydraError("normal_map_from_displacement failed (%d): %s" % (rc, lib.hydra_hip_image_last_error().decode()))
    return out, ms.value


def proctex_check(text):
    """does this procedural-texture program build for gfx950?  (hydra_hip_proctex_check: hiprtc only, no device) -> the build log; raises HydraError with the compiler's messages"""
    lib = load_hip_library()
    b = text.encode() if isinstance(text, str) else bytes(text)
    rc = lib.hydra_hip_proctex_check(b, len(b))
    msg = lib.hydra_hip_last_error(None).decode()
    if rc != 0:
        raise HydraError(msg)
    return msg


def bake_energy_tables(device=0):
    """the two multi-scattering energy tables of the globals header baked on the device (csrc/hydra_bake.hip):
    (uint16 [64, 64] roughness x dot(N,V), uint16 [64, 64, 64] ior x roughness x dot(N,V), device ms of the bake: 0 when it was cached)"""
    lib = load_hip_library()
    ggx, transp, ms = np.zeros((64, 64), np.uint16), np.zeros((64, 64, 64), np.uint16), C.c_float(0)
    rc = lib.hydra_hip_bake_energy_tables(device, _ptr(ggx), _ptr(transp), C.byref(ms))
    if rc != 0:
        raise HydraError("bake_energy_tables failed (%d): %s" % (rc, lib.hydra_hip_bake_last_error().decode()))
    return ggx, transp, ms.value


class HipCore:
    """Thin object wrapper over a hydra_hip_handle.  Either created directly (then the caller uploads the scene
    buffers) or borrowed from a HostScene that owns a HipHWLayer."""

    def __init__(self, width=0, height=0, device=0, flags=0, _borrowed=None, _owner=None):
        self.lib = load_hip_library()
        self._owner = _owner
        if _borrowed is not None:
            self.h, self.owned = C.c_void_p(_borrowed), False
            return
        h = C.c_void_p()
        rc = self.lib.hydra_hip_create(width, height, flags, device, C.byref(h))
        if rc != 0:
            raise HydraError("hydra_hip_create: %s" % self.lib.hydra_hip_last_error(None).decode())
        self.h, self.owned = h, True

    def close(self):
        if self.owned and self.h:
            self.lib.hydra_hip_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise HydraError("%s failed (%d): %s" % (what, rc, self.lib.hydra_hip_last_error(self.h).decode()))

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._ck(self.lib.hydra_hip_device_name(self.h, buf, 256), "device_name")
        return buf.value.decode()

    # ---- scene upload from host buffers (dict produced by HostScene.buffers())
    def upload_globals(self, g):
        """the [EngineGlobals | tables | lights] blob alone (the storages stay)"""
        g = np.ascontiguousarray(g, dtype=np.int32)
        self._ck(self.lib.hydra_hip_upload_globals(self.h, _ptr(g), g.size), "upload_globals")

    def upload_scene(self, b):
        L = self.lib
        g = np.ascontiguousarray(b["globals"], dtype=np.int32)
        for kind, key in enumerate(("textures", "textures_aux", "geom", "materials", "pdfs")):
            a = np.ascontiguousarray(b[key])
            self._ck(L.hydra_hip_upload_storage(self.h, kind, _ptr(a) if a.size else None, a.nbytes), "upload_storage")
        self._ck(L.hydra_hip_upload_globals(self.h, _ptr(g), g.size), "upload_globals")
        trees = int(b.get("trees_num", 1))
        for t in range(trees):
            sfx = "" if t == 0 else str(t)
            nodes, tris = np.ascontiguousarray(b["bvh_nodes" + sfx]), np.ascontiguousarray(b["bvh_tris" + sfx])
            alpha = np.ascontiguousarray(b.get("bvh_alpha" + sfx, np.zeros(0, np.uint32)), dtype=np.uint32)
            self._ck(L.hydra_hip_upload_bvh(self.h, t, _ptr(nodes), nodes.nbytes // 32, _ptr(tris), tris.nbytes // 16,
                                            _ptr(alpha) if alpha.size else None, alpha.size // 2, int(b["have_inst" + sfx])), "upload_bvh")
        self._ck(L.hydra_hip_set_bvh_trees_num(self.h, trees), "set_bvh_trees_num")
        im, il = np.ascontiguousarray(b["inst_matrices"], dtype=np.float32), np.ascontiguousarray(b["inst_light_id"], dtype=np.int32)
        if il.size < im.size // 16:
            il = np.concatenate([il, -np.ones(im.size // 16 - il.size, np.int32)])
        self._ck(L.hydra_hip_upload_instances(self.h, _ptr(im), _ptr(il), im.size // 16), "upload_instances")

    # ---- rendering
    def resize(self, width, height):
        """IHWLayer::ResizeScreen: a new frame size; the render state, a running MMLT run and the exchange's pixel lists restart"""
        self._ck(self.lib.hydra_hip_resize(self.h, width, height), "resize")
        self.width, self.height = width, height

    def set_tile_partition(self, rank, world, tile=64):
        self._ck(self.lib.hydra_hip_set_tile_partition(self.h, rank, world, tile), "set_tile_partition")

    def set_external_accumulator(self, dev_ptr, nbytes):
        self._ck(self.lib.hydra_hip_set_external_accumulator(self.h, C.c_void_p(dev_ptr), nbytes), "set_external_accumulator")

    def init_path_tracing(self, seed):
        self._ck(self.lib.hydra_hip_init_path_tracing(self.h, seed), "init_path_tracing")

    def clear(self):
        self._ck(self.lib.hydra_hip_clear_accumulated_color(self.h), "clear_accumulated_color")

    def trace_pass(self, spp=1):
        self._ck(self.lib.hydra_hip_trace_pass(self.h, spp), "trace_pass")

    def finish(self):
        self._ck(self.lib.hydra_hip_finish(self.h), "finish")

    def spp(self):
        return float(self.lib.hydra_hip_get_spp(self.h))

    def set_spp(self, v):
        self._ck(self.lib.hydra_hip_set_spp(self.h, float(v)), "set_spp")

    def hdr_image(self, w, h):
        out = np.empty((h, w, 4), np.float32)
        self._ck(self.lib.hydra_hip_get_hdr_image(self.h, _ptr(out), w, h), "get_hdr_image")
        return out

    def accumulator(self, w, h):
        """the float4 sums themselves (hydra_hip_get_accumulator)"""
        out = np.empty((h, w, 4), np.float32)
        self._ck(self.lib.hydra_hip_get_accumulator(self.h, _ptr(out), w, h), "get_accumulator")
        return out

    def ldr_image(self, w, h):
        out = np.empty((h, w), np.uint32)
        self._ck(self.lib.hydra_hip_get_ldr_image(self.h, _ptr(out), w, h), "get_ldr_image")
        return out

    def rays_stat(self):
        st = RaysStat()
        self._ck(self.lib.hydra_hip_get_rays_stat(self.h, C.byref(st)), "get_rays_stat")
        return st

    def reset_perf_counters(self):
        self._ck(self.lib.hydra_hip_reset_perf_counters(self.h), "reset_perf_counters")

    def enable_stage_timing(self, on=True):
        self._ck(self.lib.hydra_hip_enable_stage_timing(self.h, 1 if on else 0), "enable_stage_timing")

    def stage_times_per_bounce(self, max_depth):
        """float32 [max_depth, 3 (closest-hit traversal | bounce kernels | shadow traversal)] in ms since the last reset"""
        out = np.zeros((max_depth, 3), np.float32)
        self._ck(self.lib.hydra_hip_get_stage_times_per_bounce(self.h, _ptr(out), max_depth), "get_stage_times_per_bounce")
        return out

    def set_option(self, name, value):
        self._ck(self.lib.hydra_hip_set_option(self.h, name.encode(), int(value)), "set_option(%s)" % name)

    def get_option(self, name):
        v = C.c_int32(0)
        self._ck(self.lib.hydra_hip_get_option(self.h, name.encode(), C.byref(v)), "get_option(%s)" % name)
        return int(v.value)

    def samples_in_flight(self):
        return self.get_option("samples_in_flight")

    def enable_traversal_counters(self, on=True):
        self._ck(self.lib.hydra_hip_enable_traversal_counters(self.h, 1 if on else 0), "enable_traversal_counters")

    def traversal_counters(self, max_depth):
        """uint64 [max_depth, 2 (closest|shadow), 5 (rays, quads, insts, leaves, tris)]"""
        out = np.zeros((max_depth, 2, 5), np.uint64)
        self._ck(self.lib.hydra_hip_get_traversal_counters(self.h, _ptr(out), max_depth), "get_traversal_counters")
        return out

    def traversal_oob(self):
        """fetches the counting traversal kernels would have made out of range since enable_traversal_counters (must be 0)"""
        v = C.c_uint64(0)
        self._ck(self.lib.hydra_hip_get_traversal_oob(self.h, C.byref(v)), "get_traversal_oob")
        return int(v.value)

    # ---- stage entry points
    def stage_trace_totals(self, pos4, dir4, tfar=None):
        """uint64 [6]: rays, quads, instance quads, leaves, triangles, out-of-range fetches of the persistent counting kernels"""
        n = pos4.shape[0]
        pos4, dir4 = _f4(pos4, n), _f4(dir4, n)
        tf = np.ascontiguousarray(tfar, dtype=np.float32) if tfar is not None else None
        out = np.zeros(6, np.uint64)
        self._ck(self.lib.hydra_hip_stage_trace_totals(self.h, n, _ptr(pos4), _ptr(dir4), _ptr(tf) if tf is not None else None, _ptr(out)), "stage_trace_totals")
        return out

    def stage_random(self, seeds, draws):
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        n = seeds.size
        out, st = np.empty((n, draws, 4), np.float32), np.empty((n, 2), np.uint32)
        self._ck(self.lib.hydra_hip_stage_random(self.h, n, _ptr(seeds), draws, _ptr(out), _ptr(st)), "stage_random")
        return out, st

    def stage_make_eye_rays(self, xy, offs4):
        xy = np.ascontiguousarray(xy, dtype=np.int32)
        n = xy.shape[0]
        offs4 = _f4(offs4, n)
        pos, dr = np.empty((n, 4), np.float32), np.empty((n, 4), np.float32)
        self._ck(self.lib.hydra_hip_stage_make_eye_rays(self.h, n, _ptr(xy), _ptr(offs4), _ptr(pos), _ptr(dr)), "stage_make_eye_rays")
        return pos, dr

    def stage_trace(self, pos4, dir4, counters=False):
        n = pos4.shape[0]
        pos4, dir4 = _f4(pos4, n), _f4(dir4, n)
        hits = np.empty(n, LITE_HIT_DTYPE)
        cnt = np.empty((n, 3), np.uint32) if counters else None
        self._ck(self.lib.hydra_hip_stage_trace(self.h, n, _ptr(pos4), _ptr(dir4), _ptr(hits), _ptr(cnt) if counters else None), "stage_trace")
        return (hits, cnt) if counters else hits

    def stage_shadow_trace(self, pos4, dir4, tfar):
        n = pos4.shape[0]
        pos4, dir4 = _f4(pos4, n), _f4(dir4, n)
        tfar = np.ascontiguousarray(tfar, dtype=np.float32)
        vis = np.empty(n, np.float32)
        self._ck(self.lib.hydra_hip_stage_shadow_trace(self.h, n, _ptr(pos4), _ptr(dir4), _ptr(tfar), _ptr(vis)), "stage_shadow_trace")
        return vis

    def stage_eval_surface(self, pos4, dir4, hits):
        n = pos4.shape[0]
        pos4, dir4 = _f4(pos4, n), _f4(dir4, n)
        hits = np.ascontiguousarray(hits)
        out = np.empty((n, 24), np.float32)
        self._ck(self.lib.hydra_hip_stage_eval_surface(self.h, n, _ptr(pos4), _ptr(dir4), _ptr(hits), _ptr(out)), "stage_eval_surface")
        return out

    def stage_shade_point(self, surf24, dir4, flags, rnd_light4, rands10):
        """light pick + sample, materialEval, BxDF sampling at given surface points with given random numbers -> float32 [n, 28]"""
        n = len(surf24)
        surf24, dir4 = np.ascontiguousarray(surf24, np.float32), np.ascontiguousarray(dir4, np.float32)
        flags, rnd_light4 = np.ascontiguousarray(flags, np.int32), np.ascontiguousarray(rnd_light4, np.float32)
        rands10 = np.ascontiguousarray(rands10, np.float32)
        out = np.zeros((n, 28), np.float32)
        self._ck(self.lib.hydra_hip_stage_shade_point(self.h, n, _ptr(surf24), _ptr(dir4), _ptr(flags), _ptr(rnd_light4), _ptr(rands10), _ptr(out)),
                 "stage_shade_point")
        return out

    def stage_mmlt_accept(self, old8, new8, gen2, bk_scale):
        """one accept / reject step of n chains through k_mmlt_accept (include/hydra_hip.h) -> (out12 float32 [n, 12], generator states after the draw)"""
        n = len(old8)
        old8, new8 = np.ascontiguousarray(old8, np.float32), np.ascontiguousarray(new8, np.float32)
        gen2 = np.ascontiguousarray(gen2, np.uint32).copy()
        out = np.zeros((n, 12), np.float32)
        self._ck(self.lib.hydra_hip_stage_mmlt_accept(self.h, n, _ptr(old8), _ptr(new8), _ptr(gen2), float(bk_scale), _ptr(out)), "stage_mmlt_accept")
        return out, gen2

    def stage_environment(self, dir4, in8):
        """the miss shader for n rays (include/hydra_hip.h, hydra_hip_stage_environment): in8 = origin xyz, previous pdf, previous specular, flags, pixel x, y (int bits) -> float32 [n, 4]"""
        n = len(in8)
        dir4, in8 = _f4(dir4, n), np.ascontiguousarray(in8, np.float32)
        out = np.zeros((n, 4), np.float32)
        self._ck(self.lib.hydra_hip_stage_environment(self.h, n, _ptr(dir4), _ptr(in8), _ptr(out)), "stage_environment")
        return out

    def proctex_compile(self, text):
        """IHWLayer::RecompileProcTexShaders: build the scene's procedural textures from the program text ('' drops the program)"""
        b = text.encode() if isinstance(text, str) else bytes(text)
        self._ck(self.lib.hydra_hip_proctex_compile(self.h, b if b else None, len(b)), "proctex_compile")

    def stage_proctex(self, pos4, dir4, hits, max_num=16):
        """the compiled program on n hits -> (ids int32 [max_num, n], colours float32 [max_num, n, 4] decoded from the halfs the layer stores)"""
        n = len(hits)
        pos4, dir4 = _f4(pos4, n), _f4(dir4, n)
        hits = np.ascontiguousarray(hits, dtype=LITE_HIT_DTYPE)
        ids = np.zeros((max_num, n), np.int32)
        halfs = np.zeros((max_num, n, 4), np.float16)
        self._ck(self.lib.hydra_hip_stage_proctex(self.h, n, max_num, _ptr(pos4), _ptr(dir4), _ptr(hits), _ptr(ids), _ptr(halfs)), "stage_proctex")
        return ids, halfs.astype(np.float32)

    def stage_set_proctex(self, ids=None, colours=None):
        """the per-point lists the next stage_shade_point / stage_bounce calls of the same n consult: ids [max_num, n], colours [max_num, n, 4] (stored as halfs); None drops them"""
        if ids is None:
            self._ck(self.lib.hydra_hip_stage_set_proctex(self.h, 0, 0, None, None), "stage_set_proctex")
            return
        ids = np.ascontiguousarray(ids, np.int32)
        halfs = np.ascontiguousarray(colours, np.float16)
        assert ids.ndim == 2 and halfs.shape == ids.shape + (4,)
        self._ck(self.lib.hydra_hip_stage_set_proctex(self.h, ids.shape[1], ids.shape[0], _ptr(ids), _ptr(halfs)), "stage_set_proctex")

    def stage_bounce(self, depth, max_depth, pos4, dir4, surf24, in16, rands10):
        """one bounce of n paths with every input handed in (include/hydra_hip.h, hydra_hip_stage_bounce) -> float32 [n, 40]"""
        n = len(surf24)
        pos4, dir4 = _f4(pos4, n), _f4(dir4, n)
        surf24, in16, rands10 = np.ascontiguousarray(surf24, np.float32), np.ascontiguousarray(in16, np.float32), np.ascontiguousarray(rands10, np.float32)
        assert surf24.shape == (n, 24) and in16.shape == (n, 16) and rands10.shape == (n, 10)
        out = np.zeros((n, 40), np.float32)
        self._ck(self.lib.hydra_hip_stage_bounce(self.h, n, depth, max_depth, _ptr(pos4), _ptr(dir4), _ptr(surf24), _ptr(in16), _ptr(rands10), _ptr(out)), "stage_bounce")
        return out

    def stage_path_trace(self, pos4, dir4, rng2):
        n = pos4.shape[0]
        pos4, dir4 = _f4(pos4, n), _f4(dir4, n)
        rng2 = np.ascontiguousarray(rng2, dtype=np.uint32).copy()
        col = np.empty((n, 4), np.float32)
        self._ck(self.lib.hydra_hip_stage_path_trace(self.h, n, _ptr(pos4), _ptr(dir4), _ptr(rng2), _ptr(col)), "stage_path_trace")
        return col, rng2

    # ---- multi-GPU exchange over RCCL (hydra_hip_comm_*)
    def comm_unique_id(self):
        buf = np.zeros(128, np.uint8)
        self._ck(self.lib.hydra_hip_comm_unique_id(self.h, _ptr(buf)), "comm_unique_id")
        return buf

    def comm_init(self, id128, rank, world):
        buf = np.ascontiguousarray(id128, dtype=np.uint8)
        assert buf.size == 128
        self._ck(self.lib.hydra_hip_comm_init(self.h, _ptr(buf), rank, world), "comm_init")

    def comm_gather_frame(self, root=0):
        self._ck(self.lib.hydra_hip_comm_gather_frame(self.h, root), "comm_gather_frame")

    def comm_reduce_frame(self, root=0):
        self._ck(self.lib.hydra_hip_comm_reduce_frame(self.h, root), "comm_reduce_frame")

    def comm_destroy(self):
        self._ck(self.lib.hydra_hip_comm_destroy(self.h), "comm_destroy")

    def stage_pack_unpack(self, w, h):
        out = np.empty((h, w, 4), np.float32)
        self._ck(self.lib.hydra_hip_stage_pack_unpack(self.h, _ptr(out), w, h), "stage_pack_unpack")
        return out

    # ---- bidirectional building blocks (row f3)
    def stage_light_sample_forward(self, light_ids, rands4):
        ids, r = np.ascontiguousarray(light_ids, np.int32), np.ascontiguousarray(rands4, np.float32)
        out = np.zeros((ids.size, 16), np.float32)
        self._ck(self.lib.hydra_hip_stage_light_sample_forward(self.h, ids.size, _ptr(ids), _ptr(r), _ptr(out)), "stage_light_sample_forward")
        return out

    def stage_light_pdf_fwd(self, light_ids, cos_theta):
        ids, ct = np.ascontiguousarray(light_ids, np.int32), np.ascontiguousarray(cos_theta, np.float32)
        out = np.zeros((ids.size, 4), np.float32)
        self._ck(self.lib.hydra_hip_stage_light_pdf_fwd(self.h, ids.size, _ptr(ids), _ptr(ct), _ptr(out)), "stage_light_pdf_fwd")
        return out

    def stage_camera_connect(self, pos4, norm4, disk2):
        p, nn, d = np.ascontiguousarray(pos4, np.float32), np.ascontiguousarray(norm4, np.float32), np.ascontiguousarray(disk2, np.float32)
        out = np.zeros((len(p), 8), np.float32)
        self._ck(self.lib.hydra_hip_stage_camera_connect(self.h, len(p), _ptr(p), _ptr(nn), _ptr(d), _ptr(out)), "stage_camera_connect")
        return out

    def stage_mutate_kelemen(self, values, rands2, p2=64.0, p1=1024.0):
        v, r = np.ascontiguousarray(values, np.float32), np.ascontiguousarray(rands2, np.float32)
        out = np.zeros(v.size, np.float32)
        self._ck(self.lib.hydra_hip_stage_mutate_kelemen(self.h, v.size, _ptr(v), _ptr(r), p2, p1, _ptr(out)), "stage_mutate_kelemen")
        return out

    def stage_mmlt_f(self, depth, xvec):
        """IntegratorMMLT::F for the rows of xvec -> (n, 8): colour, x, y, split, MIS weight, contribFunc"""
        d, x = np.ascontiguousarray(depth, np.int32), np.ascontiguousarray(xvec, np.float32)
        out = np.zeros((d.size, 8), np.float32)
        self._ck(self.lib.hydra_hip_stage_mmlt_f(self.h, d.size, _ptr(d), _ptr(x), x.shape[1], _ptr(out)), "stage_mmlt_f")
        return out

    # ---- IntegratorMMLT (row f3)
    def mmlt_begin(self, chains, seed=777, first_bounce=0, max_depth=0, estimate_passes=0):
        self._ck(self.lib.hydra_hip_mmlt_begin(self.h, chains, seed, first_bounce, max_depth, estimate_passes), "mmlt_begin")
        self._mmlt = (chains, max_depth)

    def mmlt_pass(self, mutations=1):
        self._ck(self.lib.hydra_hip_mmlt_pass(self.h, mutations), "mmlt_pass")

    def mmlt_image(self, width, height):
        """(kScale x indirect image (h, w, 4), info dict)"""
        img, info = np.zeros((height, width, 4), np.float32), np.zeros(8, np.float32)
        self._ck(self.lib.hydra_hip_mmlt_get_image(self.h, _ptr(img), width, height, _ptr(info)), "mmlt_get_image")
        keys = ("avg_brightness", "k_scale", "acceptance", "mutations", "chains", "first_bounce", "max_depth")
        return img, dict(zip(keys, (float(v) for v in info[:7])))

    def mmlt_state(self):
        """chain planes (11, n), d per chain, current x vectors (n, 12 + 10 * max_depth), average brightness per path length"""
        info8 = np.zeros(8, np.float32)
        self._ck(self.lib.hydra_hip_mmlt_get_image(self.h, None, 0, 0, _ptr(info8)), "mmlt_get_image")
        n, max_d = int(info8[4]), int(info8[6])
        ch, depth, x, avg = np.zeros((11, n), np.float32), np.zeros(n, np.int32), np.zeros((n, 12 + 10 * max_d), np.float32), np.zeros(max_d + 1, np.float32)
        self._ck(self.lib.hydra_hip_mmlt_get_state(self.h, _ptr(ch), _ptr(depth), _ptr(x), _ptr(avg)), "mmlt_get_state")
        return ch, depth, x, avg

    def sbdpt_pass(self, passes=1):
        self._ck(self.lib.hydra_hip_sbdpt_pass(self.h, passes), "sbdpt_pass")

    def sbdpt_image(self, width, height):
        img, n = np.zeros((height, width, 4), np.float32), C.c_double(0)
        self._ck(self.lib.hydra_hip_sbdpt_get_image(self.h, _ptr(img), width, height, C.byref(n)), "sbdpt_get_image")
        return img, n.value

    def mmlt_reset_image(self):
        self._ck(self.lib.hydra_hip_mmlt_reset_image(self.h), "mmlt_reset_image")

    def mmlt_end(self):
        self._ck(self.lib.hydra_hip_mmlt_end(self.h), "mmlt_end")

    def eval_gbuffer(self, width, height, inst_remap=None, raw=False):
        """IHWLayer::EvalGBuffer: the two packed float4 layers (+ the unpacked record per pixel when `raw`)"""
        d1, d2 = np.zeros((height, width, 4), np.float32), np.zeros((height, width, 4), np.float32)
        r14 = np.zeros((height, width, 14), np.float32) if raw else None
        rm = np.ascontiguousarray(inst_remap, np.int32) if inst_remap is not None else None
        self._ck(self.lib.hydra_hip_eval_gbuffer(self.h, _ptr(d1), _ptr(d2), width, height, _ptr(rm) if rm is not None else None, 0 if rm is None else rm.size,
                                                 _ptr(r14) if raw else None), "eval_gbuffer")
        return (d1, d2, r14) if raw else (d1, d2)

    def bench_trace(self, pos4, dir4, iters=20, shadow=False):
        n = pos4.shape[0]
        pos4, dir4 = _f4(pos4, n), _f4(dir4, n)
        ms = C.c_float(0)
        self._ck(self.lib.hydra_hip_bench_trace(self.h, n, _ptr(pos4), _ptr(dir4), iters, 1 if shadow else 0, C.byref(ms)), "bench_trace")
        return float(ms.value)


_BUF_KINDS = [("globals", 0, np.int32), ("textures", 1, np.int32), ("textures_aux", 2, np.int32), ("geom", 3, np.float32),
              ("materials", 4, np.float32), ("pdfs", 5, np.float32), ("bvh_nodes", 6, np.float32), ("bvh_tris", 7, np.float32),
              ("inst_matrices", 8, np.float32), ("inst_light_id", 9, np.int32), ("remap_lists", 10, np.int32),
              ("remap_table", 11, np.int32), ("remap_inst", 12, np.int32), ("bvh_alpha", 13, np.uint32), ("bvh_nodes1", 14, np.float32),
              ("bvh_tris1", 15, np.float32), ("bvh_alpha1", 16, np.uint32)]


class HostScene:
    """A HydraAPI scene library committed through RenderDriverLite into an IHWLayer (host-blob or HIP)."""

    def __init__(self, lib_path, width=0, height=0, trace_depth=-1, enable_dof=-1, use_hip=False, device=0, seed=777):
        self.lib = load_host_library()
        err = C.create_string_buffer(1024)
        self.p = self.lib.hydra_host_open_scene(os.fsencode(lib_path), width, height, trace_depth, enable_dof,
                                                1 if use_hip else 0, device, seed, err, 1024)
        if not self.p:
            raise HydraError("open_scene(%s): %s" % (lib_path, err.value.decode()))
        self.width, self.height = self.lib.hydra_host_width(self.p), self.lib.hydra_host_height(self.p)
        self.use_hip = use_hip

    def close(self):
        if getattr(self, "p", None):
            self.lib.hydra_host_close_scene(self.p)
        self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def unsupported(self):
        return self.lib.hydra_host_unsupported(self.p)

    def set_method(self, method):
        """'mmlt': HRT_ENABLE_MMLT in the layer's flags (RenderDriverRTE.cpp:196-202), every draw() pass is then the direct-light pass +
        32 mutations of every Markov chain; anything else: the path tracer.  The accumulated image restarts."""
        if self.lib.hydra_host_set_render_method(self.p, method.encode()) != 0:
            raise HydraError("set_render_method: " + self.lib.hydra_host_last_error(self.p).decode())

    def log(self):
        return self.lib.hydra_host_log(self.p).decode()

    def proctex_program(self):
        """the text the front end handed to IHWLayer::RecompileProcTexShaders ('' = the scene declares no procedural textures)"""
        return self.lib.hydra_host_proctex_program(self.p).decode()

    def buffers(self):
        """numpy COPIES of every buffer the kernels read (same bytes the HIP layer gets)."""
        out = {}
        for name, kind, dt in _BUF_KINDS:
            ptr, n = C.c_void_p(), C.c_size_t()
            if self.lib.hydra_host_get_buffer(self.p, kind, C.byref(ptr), C.byref(n)) != 0:
                raise HydraError("get_buffer(%s)" % name)
            if n.value == 0 or not ptr.value:
                out[name] = np.zeros(0, dt)
            else:
                raw = (C.c_char * n.value).from_address(ptr.value)
                out[name] = np.frombuffer(bytes(raw), dtype=dt).copy()
        out["have_inst"] = self.lib.hydra_host_have_inst(self.p)
        out["trees_num"] = self.lib.hydra_host_trees_num(self.p)
        out["have_inst1"] = self.lib.hydra_host_have_inst_tree(self.p, 1)
        out["width"], out["height"] = self.width, self.height
        return out

    def hip(self):
        h = self.lib.hydra_host_hip_handle(self.p)
        if not h:
            raise HydraError("this scene is not backed by a HipHWLayer")
        return HipCore(_borrowed=h, _owner=self)

    def draw(self, passes=1, spp=1):
        if self.lib.hydra_host_draw(self.p, passes, spp) != 0:
            raise HydraError("draw: %s" % self.lib.hydra_host_last_error(self.p).decode())

    def hdr_image(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        if self.lib.hydra_host_get_hdr(self.p, _ptr(out), self.width, self.height) != 0:
            raise HydraError("get_hdr: %s" % self.lib.hydra_host_last_error(self.p).decode())
        return out

    def spp(self):
        return float(self.lib.hydra_host_get_spp(self.p))

    def shared_image(self, rgba, attach):
        """drive IHWLayer::SetExternalImageAccumulator (attach=True) or one ContribToExternalImageAccumulator call (attach=False)
        with an in-process shared accumulation image over the float32 [h, w, 4] array `rgba`; returns a handle for shared_image_stat / _close"""
        assert rgba.dtype == np.float32 and rgba.shape == (self.height, self.width, 4) and rgba.flags["C_CONTIGUOUS"]
        h = self.lib.hydra_host_shared_image_open(self.p, _ptr(rgba), self.width, self.height, 1 if attach else 0)
        if not h:
            raise HydraError("shared_image: %s" % self.lib.hydra_host_last_error(self.p).decode())
        return h

    def eval_gbuffer(self, depth=3, inst_remap=None, is_empty=1):
        """IHWLayer::EvalGBuffer through the adapter into a shared image of `depth` layers; returns (layers [depth, h, w, 4], gbufferIsEmpty after)"""
        layers = np.zeros((depth, self.height, self.width, 4), np.float32)
        rm = np.ascontiguousarray(inst_remap, np.int32) if inst_remap is not None else None
        st = C.c_int32(is_empty)
        if self.lib.hydra_host_eval_gbuffer(self.p, _ptr(layers), self.width, self.height, depth, _ptr(rm) if rm is not None else None, 0 if rm is None else rm.size, C.byref(st)) != 0:
            raise HydraError("eval_gbuffer: %s" % self.lib.hydra_host_last_error(self.p).decode())
        return layers, int(st.value)

    def shared_image_stat(self, handle):
        spp, rcv = C.c_float(0), C.c_int(0)
        self.lib.hydra_host_shared_image_stat(handle, C.byref(spp), C.byref(rcv))
        return float(spp.value), int(rcv.value)

    def shared_image_close(self, handle):
        self.lib.hydra_host_shared_image_close(self.p, handle)

    def bvh_stats(self):
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self.lib.hydra_host_bvh_stats(self.p, C.byref(a), C.byref(b), C.byref(c))
        return {"inner_quads": a.value, "leaves": b.value, "triangles": c.value}
