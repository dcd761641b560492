"""ctypes wrapper of oracle/liboracle.so -- the CPU checker.  Imported only by tests, smoke() and bench.py's
cpu_baseline leg; never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LITE_HIT_DTYPE = np.dtype([("t", np.float32), ("primId", np.int32), ("instId", np.int32), ("geomId", np.int32)])


class OrcScene(C.Structure):
    _fields_ = [("globals", C.c_void_p), ("matStorage", C.c_void_p), ("texStorage", C.c_void_p), ("geomStorage", C.c_void_p),
                ("pdfStorage", C.c_void_p), ("bvh", C.c_void_p), ("tris", C.c_void_p), ("haveInst", C.c_int32),
                ("instMatrices", C.c_void_p), ("instLightInstId", C.c_void_p), ("instNum", C.c_int32),
                ("remapLists", C.c_void_p), ("remapListsSize", C.c_int32), ("remapTable", C.c_void_p),
                ("remapTableSize", C.c_int32), ("remapInst", C.c_void_p), ("remapInstSize", C.c_int32),
                ("treesNum", C.c_int32), ("bvhN", C.c_void_p * 3), ("trisN", C.c_void_p * 3), ("haveInstN", C.c_int32 * 3), ("alpha", C.c_void_p * 4),
                ("texAuxStorage", C.c_void_p)]


_lib = {}


def load(fast=False):
    """liboracle.so (the checker), or with fast=True liboracle_fast.so: the same source at -O3 with the traversal's visit counters compiled
    out, which only bench.py's cpu_baseline leg times"""
    name = "liboracle_fast.so" if fast else "liboracle.so"
    if name in _lib:
        return _lib[name]
    path = os.path.join(ROOT, "oracle", name)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", ROOT, "oracle/" + name])
    lib = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int
    sp = C.POINTER(OrcScene)
    lib.orc_random_init.argtypes = [C.c_int32, vp]
    lib.orc_next_state.argtypes = [vp]
    lib.orc_next_state.restype = C.c_uint32
    lib.orc_rnd_float4.argtypes = [vp, vp]
    lib.orc_rnd_float1.argtypes = [vp]
    lib.orc_rnd_float1.restype = C.c_float
    lib.orc_make_eye_rays.argtypes = [sp, i32, i32, i32, vp, vp, vp, vp]
    lib.orc_trace.argtypes = [sp, i32, vp, vp, vp, vp, vp]
    lib.orc_shadow_trace.argtypes = [sp, i32, vp, vp, vp, vp]
    lib.orc_shadow_trace_anyhit.argtypes = [sp, i32, vp, vp, vp, vp, vp]
    lib.orc_eval_surface.argtypes = [sp, i32, vp, vp, vp, vp]
    lib.orc_path_trace.argtypes = [sp, i32, vp, vp, vp, vp]
    lib.orc_shade_point.argtypes = [sp, i32, vp, vp, vp, vp, vp, vp]
    lib.orc_stage_bounce.argtypes = [sp, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.orc_stage_set_proctex.argtypes = [i32, i32, vp, vp]
    lib.orc_stage_environment.argtypes = [sp, i32, vp, vp, vp]
    lib.orc_stage_environment.restype = None
    lib.orc_stage_set_proctex.restype = None
    lib.orc_render_pass.argtypes = [sp, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32]
    lib.orc_render_pass.restype = C.c_uint64
    lib.orc_light_sample_forward.argtypes = [sp, i32, vp, vp, vp]
    lib.orc_light_pdf_fwd.argtypes = [sp, i32, vp, vp, vp]
    lib.orc_camera_connect.argtypes = [sp, i32, vp, vp, vp, vp]
    lib.orc_mutate_kelemen.argtypes = [i32, vp, vp, C.c_float, C.c_float, vp]
    lib.orc_mmlt_f.argtypes = [sp, i32, vp, vp, i32, vp]
    lib.orc_sbdpt_pass.argtypes = [sp, i32, vp, i32, i32, vp]
    lib.orc_gbuffer.argtypes = [sp, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.orc_normal_map_from_displacement.argtypes = [i32, i32, vp, C.c_float, i32, C.c_float, vp]
    lib.orc_mmlt_run.argtypes = [sp, i32, vp, vp, i32, i32, vp, vp, vp, i32, vp]
    lib.orc_init_generators.argtypes = [i32, i32, i32, vp]
    lib.orc_collect_rays.argtypes = [sp, i32, i32, i32, i32, i32, vp, vp, vp, C.c_int64]
    lib.orc_collect_rays.restype = C.c_int64
    lib.orc_max_threads.restype = i32
    _lib[name] = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Oracle:
    """Oracle bound to one set of scene buffers (the dict from HostScene.buffers()); keeps the arrays alive."""

    def __init__(self, buffers, fast=False):
        self.lib = load(fast)
        self.b = {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in buffers.items()}
        b = self.b
        s = OrcScene()
        s.globals, s.matStorage, s.texStorage = _p(b["globals"]), _p(b["materials"]), _p(b["textures"])
        s.geomStorage, s.pdfStorage, s.bvh, s.tris = _p(b["geom"]), _p(b["pdfs"]), _p(b["bvh_nodes"]), _p(b["bvh_tris"])
        s.haveInst = int(b["have_inst"])
        s.instMatrices, s.instLightInstId = _p(b["inst_matrices"]), _p(b["inst_light_id"])
        s.instNum = b["inst_matrices"].size // 16
        s.remapLists, s.remapListsSize = _p(b["remap_lists"]), b["remap_lists"].size
        s.remapTable, s.remapTableSize = _p(b["remap_table"]), b["remap_table"].size // 2
        s.remapInst, s.remapInstSize = _p(b["remap_inst"]), b["remap_inst"].size
        # alpha table of tree 0 and, for two-tree scenes, tree 1 with its own (keys absent in single-tree buffer sets)
        s.treesNum = int(b.get("trees_num", 1))
        if "bvh_alpha" in b and b["bvh_alpha"].size:
            s.alpha[0] = b["bvh_alpha"].ctypes.data
        if s.treesNum > 1:
            s.bvhN[0], s.trisN[0], s.haveInstN[0] = b["bvh_nodes1"].ctypes.data, b["bvh_tris1"].ctypes.data, int(b["have_inst1"])
            if b["bvh_alpha1"].size:
                s.alpha[1] = b["bvh_alpha1"].ctypes.data
        if "textures_aux" in b and b["textures_aux"].size:
            s.texAuxStorage = b["textures_aux"].ctypes.data
        self.s = s
        self.w, self.h = b["width"], b["height"]

    # R1
    def random(self, seeds, draws):
        seeds = np.asarray(seeds, np.int32)
        out = np.empty((seeds.size, draws, 4), np.float32)
        st = np.empty((seeds.size, 2), np.uint32)
        tmp = np.empty(4, np.float32)
        for i, sd in enumerate(seeds):
            g = np.zeros(2, np.uint32)
            self.lib.orc_random_init(int(sd), _p(g))
            for d in range(draws):
                self.lib.orc_rnd_float4(_p(g), _p(tmp))
                out[i, d] = tmp
            st[i] = g
        return out, st

    def make_eye_rays(self, xy, offs4):
        xy = np.ascontiguousarray(xy, np.int32)
        offs4 = np.ascontiguousarray(offs4, np.float32)
        n = xy.shape[0]
        pos, dr = np.empty((n, 4), np.float32), np.empty((n, 4), np.float32)
        self.lib.orc_make_eye_rays(C.byref(self.s), n, self.w, self.h, _p(xy), _p(offs4), _p(pos), _p(dr))
        return pos, dr

    def trace(self, pos4, dir4, counters=False):
        pos4, dir4 = np.ascontiguousarray(pos4, np.float32), np.ascontiguousarray(dir4, np.float32)
        n = pos4.shape[0]
        hits = np.empty(n, LITE_HIT_DTYPE)
        cnt = np.empty((n, 3), np.uint32) if counters else None
        leaves = np.empty(n, np.uint32) if counters else None
        self.lib.orc_trace(C.byref(self.s), n, _p(pos4), _p(dir4), _p(hits), _p(cnt), _p(leaves))
        return (hits, cnt, leaves) if counters else hits

    def shadow_trace(self, pos4, dir4, tfar):
        pos4, dir4 = np.ascontiguousarray(pos4, np.float32), np.ascontiguousarray(dir4, np.float32)
        tfar = np.ascontiguousarray(tfar, np.float32)
        n = pos4.shape[0]
        vis = np.empty(n, np.float32)
        self.lib.orc_shadow_trace(C.byref(self.s), n, _p(pos4), _p(dir4), _p(tfar), _p(vis))
        return vis

    def shadow_trace_anyhit(self, pos4, dir4, tfar, counters=False):
        """the early-out shadow walk (ref: ctrace.h:1065-1294); counters = uint32 [n, 4] quads, instance quads, triangles, leaves"""
        pos4, dir4 = np.ascontiguousarray(pos4, np.float32), np.ascontiguousarray(dir4, np.float32)
        tfar = np.ascontiguousarray(tfar, np.float32)
        n = pos4.shape[0]
        vis = np.empty(n, np.float32)
        cnt = np.empty((n, 4), np.uint32) if counters else None
        self.lib.orc_shadow_trace_anyhit(C.byref(self.s), n, _p(pos4), _p(dir4), _p(tfar), _p(vis), _p(cnt))
        return (vis, cnt) if counters else vis

    def eval_surface(self, pos4, dir4, hits):
        pos4, dir4 = np.ascontiguousarray(pos4, np.float32), np.ascontiguousarray(dir4, np.float32)
        hits = np.ascontiguousarray(hits)
        n = pos4.shape[0]
        out = np.empty((n, 24), np.float32)
        self.lib.orc_eval_surface(C.byref(self.s), n, _p(pos4), _p(dir4), _p(hits), _p(out))
        return out

    def shade_point(self, surf24, dir4, flags, rnd_light4, rands10):
        """orc_shade_point: float32 [n, 28] = sample pos xyz, pdf, colour xyz, pick prob, light offset (int bits), isPoint,
        brdf xyz, pdfFwd, btdf xyz, MatSample colour xyz, pdf, direction xyz, flags (int bits), next ray flags (int bits), 2 spare"""
        n = len(surf24)
        surf24, dir4 = np.ascontiguousarray(surf24, np.float32), np.ascontiguousarray(dir4, np.float32)
        flags, rnd_light4 = np.ascontiguousarray(flags, np.int32), np.ascontiguousarray(rnd_light4, np.float32)
        rands10 = np.ascontiguousarray(rands10, np.float32)
        out = np.zeros((n, 28), np.float32)
        self.lib.orc_shade_point(C.byref(self.s), n, _p(surf24), _p(dir4), _p(flags), _p(rnd_light4), _p(rands10), _p(out))
        return out

    # ---- row f3 building blocks
    def light_sample_forward(self, light_ids, rands4):
        ids, r = np.ascontiguousarray(light_ids, np.int32), np.ascontiguousarray(rands4, np.float32)
        out = np.zeros((ids.size, 16), np.float32)
        self.lib.orc_light_sample_forward(C.byref(self.s), ids.size, _p(ids), _p(r), _p(out))
        return out

    def light_pdf_fwd(self, light_ids, cos_theta):
        ids, ct = np.ascontiguousarray(light_ids, np.int32), np.ascontiguousarray(cos_theta, np.float32)
        out = np.zeros((ids.size, 4), np.float32)
        self.lib.orc_light_pdf_fwd(C.byref(self.s), ids.size, _p(ids), _p(ct), _p(out))
        return out

    def camera_connect(self, pos4, norm4, disk2):
        p, nn, d = np.ascontiguousarray(pos4, np.float32), np.ascontiguousarray(norm4, np.float32), np.ascontiguousarray(disk2, np.float32)
        out = np.zeros((len(p), 8), np.float32)
        self.lib.orc_camera_connect(C.byref(self.s), len(p), _p(p), _p(nn), _p(d), _p(out))
        return out

    def mutate_kelemen(self, values, rands2, p2=64.0, p1=1024.0):
        v, r = np.ascontiguousarray(values, np.float32), np.ascontiguousarray(rands2, np.float32)
        out = np.zeros(v.size, np.float32)
        self.lib.orc_mutate_kelemen(v.size, _p(v), _p(r), p2, p1, _p(out))
        return out

    def mmlt_f(self, depth, xvec):
        """IntegratorMMLT::F for n primary-sample vectors (rows of xvec, >= 12 + 10 * depth floats) -> (n, 8): colour, x, y, split, MIS weight, contribFunc"""
        d, x = np.ascontiguousarray(depth, np.int32), np.ascontiguousarray(xvec, np.float32)
        assert x.ndim == 2 and x.shape[0] == d.size and x.shape[1] >= 12 + 10 * int(d.max()) and 1 <= int(d.min()) and int(d.max()) <= 16
        out = np.zeros((d.size, 8), np.float32)
        self.lib.orc_mmlt_f(C.byref(self.s), d.size, _p(d), _p(x), x.shape[1], _p(out))
        return out

    def mmlt_chain_gens(self, n, seed):
        """generator states of n chains as the HIP layer seeds them: RandomGenInit(seed + 2i) for mutations, (seed + 2i + 1) for accept tests"""
        g = np.zeros((n, 4), np.uint32)
        for i in range(n):
            self.lib.orc_random_init(int(seed + 2 * i), _p(g[i, 0:2]))
            self.lib.orc_random_init(int(seed + 2 * i + 1), _p(g[i, 2:4]))
        return g

    def mmlt_fresh(self, gens4, depth, max_depth):
        """InitialSamplePS for every chain: 12 + 10 d draws from its first generator (advanced in place)"""
        x = np.zeros((len(depth), 12 + 10 * max_depth), np.float32)
        for i, d in enumerate(depth):
            st = np.ascontiguousarray(gens4[i, 0:2])
            for j in range(12 + 10 * int(d)):
                x[i, j] = self.lib.orc_rnd_float1(_p(st))
            gens4[i, 0:2] = st
        return x

    def mmlt_run(self, depth, gens4, xrows, mutations):
        """`mutations` steps of the chains from the given states -> (image (h, w, 4), chains (n, 6) = y, colour, pixel, accepted counts); gens4 and xrows advance in place"""
        d = np.ascontiguousarray(depth, np.int32)
        assert gens4.dtype == np.uint32 and gens4.flags.c_contiguous and xrows.dtype == np.float32 and xrows.flags.c_contiguous
        img = np.zeros((self.h, self.w, 4), np.float32)
        ch, acc = np.zeros((d.size, 6), np.float32), np.zeros(d.size, np.int32)
        self.lib.orc_mmlt_run(C.byref(self.s), d.size, _p(gens4), _p(d), mutations, self.w, _p(img), _p(ch), _p(xrows), xrows.shape[1], _p(acc))
        return img, ch, acc

    def sbdpt_pass(self, gens4, max_depth, image=None):
        """one IntegratorSBDPT pass of len(gens4) samples (generators advance in place); returns the splat image (h, w, 4)"""
        assert gens4.dtype == np.uint32 and gens4.flags.c_contiguous
        if image is None:
            image = np.zeros((self.h, self.w, 4), np.float32)
        self.lib.orc_sbdpt_pass(C.byref(self.s), len(gens4), _p(gens4), max_depth, self.w, _p(image))
        return image

    def gbuffer(self, x0=0, y0=0, nx=None, ny=None):
        """IntegratorCommon::gbufferEval for a pixel window (default: the frame): (data1, data2, raw14)"""
        nx, ny = self.w - x0 if nx is None else nx, self.h - y0 if ny is None else ny
        d1, d2, raw = np.zeros((ny, nx, 4), np.float32), np.zeros((ny, nx, 4), np.float32), np.zeros((ny, nx, 14), np.float32)
        self.lib.orc_gbuffer(C.byref(self.s), self.w, self.h, x0, y0, nx, ny, _p(d1), _p(d2), _p(raw))
        return d1, d2, raw

    def stage_mmlt_accept(self, old8, new8, gen2, bk_scale):
        """orc_stage_mmlt_accept -> (out12 float32 [n, 12], generator states after the draw)"""
        n = len(old8)
        old8, new8 = np.ascontiguousarray(old8, np.float32), np.ascontiguousarray(new8, np.float32)
        gen2 = np.ascontiguousarray(gen2, np.uint32).copy()
        out = np.zeros((n, 12), np.float32)
        self.lib.orc_stage_mmlt_accept.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
        self.lib.orc_stage_mmlt_accept.restype = None
        self.lib.orc_stage_mmlt_accept(n, _p(old8), _p(new8), _p(gen2), float(bk_scale), _p(out))
        return out, gen2

    def stage_environment(self, dir4, in8):
        """orc_stage_environment -> float32 [n, 4]"""
        n = len(in8)
        dir4, in8 = np.ascontiguousarray(dir4, np.float32), np.ascontiguousarray(in8, np.float32)
        out = np.zeros((n, 4), np.float32)
        self.lib.orc_stage_environment(C.byref(self.s), n, _p(dir4), _p(in8), _p(out))
        return out

    def stage_set_proctex(self, ids=None, colours=None):
        """orc_stage_set_proctex: the per-point procedural texture lists (ids [max_num, n], colours [max_num, n, 4], stored as halfs) for the following stage_bounce calls of the same n"""
        if ids is None:
            self.lib.orc_stage_set_proctex(0, 0, None, None)
            return
        ids, halfs = np.ascontiguousarray(ids, np.int32), np.ascontiguousarray(colours, np.float16)
        self.lib.orc_stage_set_proctex(ids.shape[1], ids.shape[0], _p(ids), _p(halfs))

    def stage_bounce(self, depth, max_depth, pos4, dir4, surf24, in16, rands10):
        """orc_stage_bounce: one bounce of n paths with every input handed in -> float32 [n, 40] (layouts: include/hydra_hip.h, hydra_hip_stage_bounce)"""
        n = len(surf24)
        pos4, dir4 = np.ascontiguousarray(pos4, np.float32), np.ascontiguousarray(dir4, np.float32)
        surf24, in16, rands10 = np.ascontiguousarray(surf24, np.float32), np.ascontiguousarray(in16, np.float32), np.ascontiguousarray(rands10, np.float32)
        out = np.zeros((n, 40), np.float32)
        self.lib.orc_stage_bounce(C.byref(self.s), n, depth, max_depth, _p(pos4), _p(dir4), _p(surf24), _p(in16), _p(rands10), _p(out))
        return out

    def path_trace(self, pos4, dir4, rng2):
        pos4, dir4 = np.ascontiguousarray(pos4, np.float32), np.ascontiguousarray(dir4, np.float32)
        rng2 = np.ascontiguousarray(rng2, np.uint32).copy()
        n = pos4.shape[0]
        col = np.empty((n, 4), np.float32)
        self.lib.orc_path_trace(C.byref(self.s), n, _p(pos4), _p(dir4), _p(rng2), _p(col))
        return col, rng2

    def init_generators(self, seed):
        g = np.empty((self.h * self.w, 2), np.uint32)
        self.lib.orc_init_generators(self.w, self.h, seed, _p(g))
        return g

    def render(self, spp, seed=777, sum_mode=False, rank=0, world=1, tile=64, threads=0, gens=None, image=None, spp_done=0, streams=1):
        """spp passes of DoPass; returns (image float4 [h,w,4], rays traced, gens).
        streams = K > 1 follows hydra_hip.h "samples_in_flight": sample j draws from generator stream j % K of its pixel,
        stream k of pixel p being RandomGenInit(seed + k * w * h + p); gens is then a list of K state arrays."""
        if gens is None:
            gens = [self.init_generators(seed + k * self.w * self.h) for k in range(streams)]
            if streams == 1:
                gens = gens[0]
        glist = gens if isinstance(gens, list) else [gens]
        if image is None:
            image = np.zeros((self.h, self.w, 4), np.float32)
        rays = 0
        for k in range(spp):
            rays += self.lib.orc_render_pass(C.byref(self.s), self.w, self.h, _p(glist[k % len(glist)]), _p(image), spp_done + k, 1 if sum_mode else 0,
                                             rank, world, tile, threads)
        return image, int(rays), gens

    def collect_rays(self, seed, bounce, shadow=False, cap=None):
        cap = cap or self.w * self.h
        pos, dr, tf = np.empty((cap, 4), np.float32), np.empty((cap, 4), np.float32), np.empty(cap, np.float32)
        n = self.lib.orc_collect_rays(C.byref(self.s), self.w, self.h, seed, bounce, 1 if shadow else 0, _p(pos), _p(dr), _p(tf), cap)
        return pos[:n].copy(), dr[:n].copy(), tf[:n].copy()

    def max_threads(self):
        return self.lib.orc_max_threads()


def normal_map_from_displacement(rgba, bump_amt, inv_height, smooth_lvl):
    """the oracle's CPUSharedData::NormalMapFromDisplacement: uint8 [h, w, 4] -> uint8 [h, w, 4]"""
    lib = load()
    a = np.ascontiguousarray(rgba, np.uint8)
    out = np.zeros_like(a)
    lib.orc_normal_map_from_displacement(a.shape[1], a.shape[0], _p(a), float(bump_amt), int(bool(inv_height)), float(smooth_lvl), _p(out))
    return out

