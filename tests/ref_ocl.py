"""Launch kernels of oracle/_ref/*.hsaco (the reference's own code compiled for gfx950 by oracle/build_ref.sh) through
the HIP module API.  TEST INFRASTRUCTURE: used by tests/golden/make_golden.py and by GPU tests, never by the product."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


class HipError(RuntimeError):
    pass


class RefModule:
    def __init__(self, name="ref_driver.hsaco", device=0):
        path = os.path.join(REF_DIR, name)
        if not os.path.exists(path):
            raise HipError("%s is missing: run oracle/build_ref.sh in the container that has /root/reference" % path)
        self.hip = C.CDLL("libamdhip64.so")
        self._ck(self.hip.hipSetDevice(device), "hipSetDevice")
        self.mod = C.c_void_p()
        self._ck(self.hip.hipModuleLoad(C.byref(self.mod), path.encode()), "hipModuleLoad")
        self.bufs = []

    def _ck(self, rc, what):
        if rc != 0:
            raise HipError("%s failed with hip error %d" % (what, rc))

    def up(self, arr):
        """upload a numpy array, returns device pointer (int)"""
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        self._ck(self.hip.hipMalloc(C.byref(p), max(arr.nbytes, 16)), "hipMalloc")
        if arr.nbytes:
            self._ck(self.hip.hipMemcpy(p, arr.ctypes.data_as(C.c_void_p), arr.nbytes, 1), "hipMemcpy H2D")
        self.bufs.append(p)
        return p.value

    def alloc(self, nbytes):
        p = C.c_void_p()
        self._ck(self.hip.hipMalloc(C.byref(p), max(nbytes, 16)), "hipMalloc")
        self._ck(self.hip.hipMemset(p, 0, max(nbytes, 16)), "hipMemset")
        self.bufs.append(p)
        return p.value

    def down(self, ptr, dtype, shape):
        out = np.empty(shape, dtype)
        self._ck(self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, 2), "hipMemcpy D2H")
        return out

    def launch(self, kernel, n, args, block=64):
        """args: list of ('p', devptr) | ('i', int) | ('f', float).  One work-item per element, 1-D."""
        fn = C.c_void_p()
        self._ck(self.hip.hipModuleGetFunction(C.byref(fn), self.mod, kernel.encode()), "hipModuleGetFunction(%s)" % kernel)
        holders = []
        for kind, v in args:
            holders.append(C.c_void_p(v) if kind == "p" else C.c_float(v) if kind == "f" else C.c_int(v))
        params = (C.c_void_p * len(holders))(*[C.cast(C.byref(h), C.c_void_p) for h in holders])
        grid = (n + block - 1) // block
        self._ck(self.hip.hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, None, params, None), "hipModuleLaunchKernel(%s)" % kernel)
        self._ck(self.hip.hipDeviceSynchronize(), "hipDeviceSynchronize after %s" % kernel)

    def close(self):
        for p in self.bufs:
            self.hip.hipFree(p)
        self.bufs = []
        if self.mod:
            self.hip.hipModuleUnload(self.mod)
            self.mod = None


class RefScene:
    """scene buffers (dict from HostScene.buffers()) resident on the device for the reference kernels"""

    def __init__(self, mod, b):
        self.m, self.b = mod, b
        up = mod.up
        self.globals = up(b["globals"])
        self.mat, self.tex, self.geom = up(b["materials"]), up(b["textures"]), up(b["geom"])
        self.pdf = up(b["pdfs"] if b["pdfs"].size else np.zeros(4, np.float32))
        self.bvh, self.tris = up(b["bvh_nodes"]), up(b["bvh_tris"])
        self.matrices, self.light_id = up(b["inst_matrices"]), up(b["inst_light_id"])
        self.have_inst = int(b["have_inst"])
        self.w, self.h = b["width"], b["height"]
        # alpha table of tree 0, and a second tree with its own table (0 = absent: the kernels test the pointers)
        self.alpha = up(b["bvh_alpha"]) if b.get("bvh_alpha", np.zeros(0)).size else 0
        two = int(b.get("trees_num", 1)) > 1
        self.bvh1, self.tris1 = (up(b["bvh_nodes1"]), up(b["bvh_tris1"])) if two else (0, 0)
        self.alpha1 = up(b["bvh_alpha1"]) if two and b["bvh_alpha1"].size else 0
        self.have_inst1 = int(b.get("have_inst1", 0)) if two else 0
        self.texaux = up(b["textures_aux"]) if b.get("textures_aux", np.zeros(0)).size else self.tex   # the aux arena (normal maps); never read without one

    def random(self, seeds, draws):
        seeds = np.ascontiguousarray(seeds, np.int32)
        n = seeds.size
        out, st = self.m.alloc(n * draws * 16), self.m.alloc(n * 8)
        self.m.launch("ref_random", n, [("p", self.m.up(seeds)), ("i", draws), ("p", out), ("p", st), ("i", n)])
        return self.m.down(out, np.float32, (n, draws, 4)), self.m.down(st, np.uint32, (n, 2))

    def make_eye_rays(self, xy, offs4):
        n = len(xy)
        pos, dr = self.m.alloc(n * 16), self.m.alloc(n * 16)
        self.m.launch("ref_make_eye_rays", n, [("p", self.m.up(np.ascontiguousarray(xy, np.int32))), ("p", self.m.up(np.ascontiguousarray(offs4, np.float32))),
                                               ("p", self.globals), ("i", self.w), ("i", self.h), ("p", pos), ("p", dr), ("i", n)])
        return self.m.down(pos, np.float32, (n, 4)), self.m.down(dr, np.float32, (n, 4))

    def trace(self, pos4, dir4):
        n = len(pos4)
        hits = self.m.alloc(n * 16)
        self.m.launch("ref_trace", n, [("p", self.m.up(np.ascontiguousarray(pos4, np.float32))), ("p", self.m.up(np.ascontiguousarray(dir4, np.float32))),
                                       ("p", self.bvh), ("p", self.tris), ("p", hits), ("i", self.have_inst), ("i", n),
                                       ("p", self.alpha), ("p", self.bvh1), ("p", self.tris1), ("p", self.alpha1), ("i", self.have_inst1), ("p", self.tex), ("p", self.globals)])
        dt = np.dtype([("t", np.float32), ("primId", np.int32), ("instId", np.int32), ("geomId", np.int32)])
        return self.m.down(hits, dt, (n,))

    def shadow_trace(self, trace_mod, pos4, dir4, tfar):
        """the reference's OWN shadow kernels of shaders/trace.cl, unmodified (oracle/_ref/trace.hsaco): BVH4TraversalInstShadowKenrel ->
        BVH4InstTraverseShadow (ctrace.h:1065-1294) for instanced trees, BVH4TraversalShadowKenrel (closest hit below maxDist) otherwise.
        Ray layout of those kernels: origin.w = maxDist, direction.w = target instance id as int bits (-1 = any); flags 0 = active.
        Returns visibility 0/1 (their ushort4 output decompressed)."""
        n = len(pos4)
        org = np.ascontiguousarray(pos4, np.float32).copy()
        org[:, 3] = np.ascontiguousarray(tfar, np.float32)
        dr = np.ascontiguousarray(dir4, np.float32).copy()
        dr[:, 3] = np.int32(-1).view(np.float32)
        m = trace_mod
        flags = m.up(np.zeros(n, np.uint32))
        shadow = m.alloc(n * 8)
        bvh, tris, glob = m.up(self.b["bvh_nodes"]), m.up(self.b["bvh_tris"]), m.up(self.b["globals"])
        kern = "BVH4TraversalInstShadowKenrel" if self.have_inst else "BVH4TraversalShadowKenrel"
        m.launch(kern, n, [("p", flags), ("p", m.up(org)), ("p", m.up(dr)), ("p", shadow), ("p", bvh), ("p", tris), ("p", glob), ("i", 0), ("i", n)], block=256)
        out = m.down(shadow, np.uint16, (n, 4))
        assert ((out[:, 0] == 0) | (out[:, 0] == 65535)).all()
        return (out[:, 0] == 65535).astype(np.float32)

    def eval_surface(self, pos4, dir4, hits):
        n = len(pos4)
        out = self.m.alloc(n * 96)
        self.m.launch("ref_eval_surface", n, [("p", self.m.up(np.ascontiguousarray(pos4, np.float32))), ("p", self.m.up(np.ascontiguousarray(dir4, np.float32))),
                                              ("p", self.m.up(np.ascontiguousarray(hits))), ("p", self.matrices), ("p", self.geom), ("p", self.globals),
                                              ("p", out), ("i", n)])
        return self.m.down(out, np.float32, (n, 24))

    def shade_point(self, surf24, dir4, flags, rnd_light4, rands10):
        n = len(surf24)
        out = self.m.alloc(n * 112)
        self.m.launch("ref_shade_point", n, [("p", self.m.up(np.ascontiguousarray(surf24, np.float32))), ("p", self.m.up(np.ascontiguousarray(dir4, np.float32))),
                                             ("p", self.m.up(np.ascontiguousarray(flags, np.int32))), ("p", self.m.up(np.ascontiguousarray(rnd_light4, np.float32))),
                                             ("p", self.m.up(np.ascontiguousarray(rands10, np.float32))), ("p", self.mat), ("p", self.tex), ("p", self.pdf),
                                             ("p", self.globals), ("p", out), ("i", n), ("p", self.texaux)])
        return self.m.down(out, np.float32, (n, 28))

    # ---- row f3 building blocks
    def light_sample_forward(self, light_ids, rands4):
        n = len(light_ids)
        out = self.m.alloc(n * 64)
        self.m.launch("ref_light_sample_forward", n, [("p", self.m.up(np.ascontiguousarray(light_ids, np.int32))), ("p", self.m.up(np.ascontiguousarray(rands4, np.float32))),
                                                      ("p", self.tex), ("p", self.pdf), ("p", self.globals), ("p", out), ("i", n)])
        return self.m.down(out, np.float32, (n, 16))

    def light_pdf_fwd(self, light_ids, cos_theta):
        n = len(light_ids)
        out = self.m.alloc(n * 16)
        self.m.launch("ref_light_pdf_fwd", n, [("p", self.m.up(np.ascontiguousarray(light_ids, np.int32))), ("p", self.m.up(np.ascontiguousarray(cos_theta, np.float32))),
                                               ("p", self.tex), ("p", self.pdf), ("p", self.globals), ("p", out), ("i", n)])
        return self.m.down(out, np.float32, (n, 4))

    def camera_connect(self, pos4, norm4, disk2):
        n = len(pos4)
        out = self.m.alloc(n * 32)
        self.m.launch("ref_camera_connect", n, [("p", self.m.up(np.ascontiguousarray(pos4, np.float32))), ("p", self.m.up(np.ascontiguousarray(norm4, np.float32))),
                                                ("p", self.m.up(np.ascontiguousarray(disk2, np.float32))), ("p", self.globals), ("p", out), ("i", n)])
        return self.m.down(out, np.float32, (n, 8))

    def environment_extended(self, dir4, in8):
        """environmentColorExtended (cbidir.h:593-629) for n rays that left the scene: in8 = origin xyz, previous pdf, previous specular, flags, pixel x, y (int bits) -> [n, 4]"""
        n = len(in8)
        out = self.m.alloc(n * 16)
        self.m.launch("ref_environment_extended", n, [("p", self.m.up(np.ascontiguousarray(dir4, np.float32))), ("p", self.m.up(np.ascontiguousarray(in8, np.float32))),
                                                      ("p", self.mat), ("p", self.tex), ("p", self.pdf), ("p", self.globals), ("p", out), ("i", n)])
        return self.m.down(out, np.float32, (n, 4))

    def mutate_kelemen(self, values, rands2, p2, p1):
        n = len(values)
        out = self.m.alloc(n * 4)
        self.m.launch("ref_mutate_kelemen", n, [("p", self.m.up(np.ascontiguousarray(values, np.float32))), ("p", self.m.up(np.ascontiguousarray(rands2, np.float32))),
                                                ("f", p2), ("f", p1), ("p", out), ("i", n)])
        return self.m.down(out, np.float32, (n,))

    def mmlt_f(self, depth, xvec):
        d, x = np.ascontiguousarray(depth, np.int32), np.ascontiguousarray(xvec, np.float32)
        n = d.size
        out = self.m.alloc(n * 32)
        self.m.launch("ref_mmlt_f", n, [("p", self.m.up(d)), ("p", self.m.up(x)), ("i", x.shape[1]),
                                        ("p", self.bvh), ("p", self.tris), ("i", self.have_inst), ("p", self.matrices), ("p", self.light_id),
                                        ("p", self.geom), ("p", self.mat), ("p", self.tex), ("p", self.pdf), ("p", self.globals), ("p", out), ("i", n),
                                        ("p", self.alpha), ("p", self.bvh1), ("p", self.tris1), ("p", self.alpha1), ("i", self.have_inst1), ("p", self.texaux)])
        return self.m.down(out, np.float32, (n, 8))

    def gbuffer(self, x0, y0, nx, ny):
        """IntegratorCommon::gbufferEval for the pixel window: (data1, data2, raw14), each (ny, nx, ...)"""
        qmc = plane_hammersley(64)
        n = nx * ny
        o1, o2, raw = self.m.alloc(n * 16), self.m.alloc(n * 16), self.m.alloc(n * 56)
        self.m.launch("ref_gbuffer", n, [("p", self.m.up(qmc)), ("i", x0), ("i", y0), ("i", nx), ("i", ny),
                                         ("p", self.bvh), ("p", self.tris), ("i", self.have_inst), ("p", self.matrices), ("p", self.light_id),
                                         ("p", self.geom), ("p", self.mat), ("p", self.tex), ("p", self.pdf), ("p", self.globals), ("p", o1), ("p", o2), ("p", raw),
                                         ("p", self.alpha), ("p", self.bvh1), ("p", self.tris1), ("p", self.alpha1), ("i", self.have_inst1), ("p", self.texaux)])
        return self.m.down(o1, np.float32, (ny, nx, 4)), self.m.down(o2, np.float32, (ny, nx, 4)), self.m.down(raw, np.float32, (ny, nx, 14))

    def path_trace(self, pos4, dir4, rng2):
        n = len(pos4)
        rng = self.m.up(np.ascontiguousarray(rng2, np.uint32))
        col = self.m.alloc(n * 16)
        self.m.launch("ref_path_trace", n, [("p", self.m.up(np.ascontiguousarray(pos4, np.float32))), ("p", self.m.up(np.ascontiguousarray(dir4, np.float32))),
                                            ("p", rng), ("p", self.bvh), ("p", self.tris), ("i", self.have_inst), ("p", self.matrices), ("p", self.light_id),
                                            ("p", self.geom), ("p", self.mat), ("p", self.tex), ("p", self.pdf), ("p", self.globals), ("p", col), ("i", n),
                                            ("p", self.alpha), ("p", self.bvh1), ("p", self.tris1), ("p", self.alpha1), ("i", self.have_inst1), ("p", self.texaux)])
        return self.m.down(col, np.float32, (n, 4)), self.m.down(rng, np.uint32, (n, 2))


def plane_hammersley(n):
    """PlaneHammersley (hydra_drv/globals_sys.cpp:45-61, host code of the reference: restated, float32 arithmetic): (n, 2)"""
    out = np.zeros((n, 2), np.float32)
    for k in range(n):
        u, p, kk = np.float32(0), np.float32(0.5), k
        while kk:
            if kk & 1:
                u = np.float32(u + p)
            p = np.float32(p * np.float32(0.5))
            kk >>= 1
        out[k, 0] = u
        out[k, 1] = np.float32(np.float32(k + 0.5) / np.float32(n))
    return out


class RefWavefront:
    """The reference's OWN wavefront path-tracing stage kernels (hydra_drv/shaders/trace.cl, material.cl, light.cl: compiled unmodified by
    oracle/build_ref.sh), launched in the order of its host loop (GPUOCLLayer::trace1D_Rev, GPUOCLLayerCore.cpp:9-130: per bounce
    traverse -> ComputeHit -> HitEnvOrLightKernel -> LightSample -> shadow traversal -> Shade -> NextBounce) on n rays, every buffer
    read back after every kernel.  What the kernels draw from their generators is a function of the seed: generator i starts as
    RandomGenInit(seed + i) (InitRandomGen, trace.cl:6-13)."""

    def __init__(self, b, device=0, texproc=None):
        """texproc: name of the scene's procedural-texture program in oracle/_ref (oracle/build_ref.sh texproc: the reference's texproc.cl with the scene's functions
        spliced in as RenderDriverRTE does); ProcTexExec then runs after ComputeHit, as in GPUOCLLayer::runKernel_ComputeHit (GPUOCLKernels.cpp:662-690), and its
        per-ray lists reach HitEnvOrLightKernel, Shade and NextBounce."""
        self.b = b
        self.mods = {k: RefModule(k + ".hsaco", device) for k in ("trace", "material", "light", "ref_driver")}
        self.texproc = RefModule(texproc, device) if texproc else None
        self.have_inst = int(b["have_inst"])

    def close(self):
        for m in self.mods.values():
            m.close()
        if self.texproc:
            self.texproc.close()

    def run(self, pos4, dir4, seed, bounces, xy=None):
        """xy: the pixel of every ray (int [n, 2]) -> in_packXY of HitEnvOrLightKernel (x | y << 16): what a back-plate is projected by"""
        b, n = self.b, len(pos4)
        T, M, L, R = self.mods["trace"], self.mods["material"], self.mods["light"], self.mods["ref_driver"]
        f32, i32, u32 = np.float32, np.int32, np.uint32
        zeros4 = np.zeros((n, 4), f32)

        def scene(m):   # every module is its own HIP module: buffers are uploaded per module (small scenes)
            d = dict(glob=m.up(b["globals"]), mat=m.up(b["materials"]), tex=m.up(b["textures"]), geom=m.up(b["geom"]),
                     pdf=m.up(b["pdfs"] if b["pdfs"].size else np.zeros(4, f32)), bvh=m.up(b["bvh_nodes"]), tris=m.up(b["bvh_tris"]),
                     matrices=m.up(b["inst_matrices"]), light_id=m.up(b["inst_light_id"]),
                     texaux=m.up(b["textures_aux"]) if b.get("textures_aux", np.zeros(0)).size else 0,
                     remap_lists=m.up(b["remap_lists"] if b["remap_lists"].size else np.zeros(4, i32)),
                     remap_table=m.up(b["remap_table"] if b["remap_table"].size else np.zeros(4, i32)),
                     remap_inst=m.up(b["remap_inst"] if b["remap_inst"].size else np.zeros(4, i32)))
            if not d["texaux"]:
                d["texaux"] = d["tex"]
            return d
        sT, sM, sL = scene(T), scene(M), scene(L)
        rpos, rdir = np.ascontiguousarray(pos4, f32).copy(), np.ascontiguousarray(dir4, f32).copy()
        rpos[:, 3] = 0; rdir[:, 3] = 0
        flags = np.zeros(n, u32)
        gens = np.zeros((n, 2), u32)
        g = T.alloc(n * 8)
        T.launch("InitRandomGen", n, [("p", g), ("i", int(seed)), ("i", n)], block=256)
        gens = T.down(g, u32, (n, 2))
        color, thr = zeros4.copy(), np.ones((n, 4), f32)                    # ClearAllInternalTempBuffers (screen.cl:381-406): colour 0, throughput 1
        mis = np.zeros((n, 4), f32); mis[:, 0] = 1.0; mis[:, 1] = 1.0         # makeInitialMisData: pdf 1, cos 1, no material, isSpecular 1
        mis.view(i32)[:, 2] = -1; mis.view(i32)[:, 3] = 1
        fog = zeros4.copy()
        rec = []
        n_remap_table = b["remap_table"].size // 2
        n_inst = b["inst_matrices"].size // 16
        for depth in range(bounces):
            st = dict(rpos=rpos.copy(), rdir=rdir.copy(), flags_in=flags.copy(), gens_in=gens.copy(), color_in=color.copy(), thr_in=thr.copy(), mis_in=mis.copy())
            # ---- closest hit
            d_flags, d_hits = T.up(flags), T.alloc(n * 16)
            d_rpos, d_rdir = T.up(rpos), T.up(rdir)
            T.launch("BVH4TraversalInstKernel" if self.have_inst else "BVH4TraversalKernel", n,
                     [("p", d_rpos), ("p", d_rdir), ("p", sT["bvh"]), ("p", sT["tris"]), ("p", d_flags), ("p", d_hits), ("i", 0), ("i", n)], block=256)
            hits = T.down(d_hits, np.dtype([("t", f32), ("primId", i32), ("instId", i32), ("geomId", i32)]), (n,))
            # ---- ComputeHit
            d_surf = T.alloc(n * 64)
            T.launch("ComputeHit", n, [("p", d_rpos), ("p", d_rdir), ("p", d_hits), ("p", sT["matrices"]), ("p", sT["geom"]), ("p", sT["mat"]),
                                       ("p", sT["remap_lists"]), ("p", sT["remap_table"]), ("p", sT["remap_inst"]), ("p", d_flags), ("p", d_surf),
                                       ("p", sT["glob"]), ("i", n_remap_table), ("i", n_inst), ("i", n)], block=256)
            flags = T.down(d_flags, u32, (n,))
            surf_planes = T.down(d_surf, f32, (4, n, 4))
            r_surf, r_out = R.up(surf_planes), R.alloc(n * 96)
            R.launch("ref_read_surface_hit", n, [("p", r_surf), ("p", r_out), ("i", n)])
            st.update(hits=hits, flags_hit=flags.copy(), surf=R.down(r_out, f32, (n, 24)))
            # ---- ProcTexExec (texproc.cl:94-192): F4_PROCTEX_SIZE = 12 float4 planes per ray -- 16 int planes of ids, then two textures per float4 as halfs
            m_ptl = 0
            if self.texproc is not None:
                P = self.texproc
                if depth == 0:
                    sP = dict(tex=P.up(b["textures"]), mat=P.up(b["materials"]), glob=P.up(b["globals"]), matrices=P.up(b["inst_matrices"]))
                p_out = P.up(np.zeros((12, n, 4), f32))
                P.launch("ProcTexExec", n, [("p", P.up(flags)), ("p", P.up(rdir)), ("p", P.up(surf_planes)), ("p", 0), ("p", 0), ("p", P.up(hits)), ("p", sP["matrices"]),
                                            ("p", p_out), ("p", sP["tex"]), ("p", sP["mat"]), ("p", sP["glob"]), ("i", n)], block=256)
                ptl = P.down(p_out, f32, (12, n, 4))
                st.update(proctex=ptl)
                m_ptl = M.up(ptl)
            # ---- HitEnvOrLightKernel
            m_flags, m_color, m_thr, m_mis, m_emis = M.up(flags), M.up(color), M.up(thr), M.up(mis), M.alloc(n * 16)
            m_rpos, m_rdir, m_surf, m_hits = M.up(rpos), M.up(rdir), M.up(surf_planes), M.up(hits)
            m_xy = M.up((np.asarray(xy, np.int32)[:, 0] | (np.asarray(xy, np.int32)[:, 1] << 16)).astype(np.int32)) if xy is not None else M.alloc(n * 4)
            M.launch("HitEnvOrLightKernel", n, [("p", m_rpos), ("p", m_rdir), ("p", m_flags), ("p", m_xy), ("p", m_surf), ("p", m_ptl),
                                                ("p", m_color), ("p", m_thr), ("p", m_mis), ("p", m_emis), ("p", 0), ("p", m_mis), ("p", 0), ("p", 0), ("p", 0),
                                                ("p", sM["tex"]), ("p", sM["texaux"]), ("p", sM["mat"]), ("p", sM["pdf"]), ("p", sM["glob"]),
                                                ("p", sM["light_id"]), ("p", m_hits), ("f", 1.0), ("i", depth), ("i", 0), ("i", n)], block=256)
            flags, color, thr = M.down(m_flags, u32, (n,)), M.down(m_color, f32, (n, 4)), M.down(m_thr, f32, (n, 4))
            emis = M.down(m_emis, f32, (n, 4))
            st.update(flags_env=flags.copy(), color_env=color.copy(), thr_env=thr.copy(), emission=emis)
            # ---- LightSample
            l_gens, l_lrev, l_srpos, l_srdir = L.up(gens), L.alloc(n * 48), L.alloc(n * 16), L.alloc(n * 16)
            L.launch("LightSample", n, [("p", 0), ("p", 0), ("p", L.up(rpos)), ("p", L.up(rdir)), ("p", L.up(flags)), ("p", L.up(surf_planes)),
                                        ("p", l_gens), ("p", l_lrev), ("p", l_srpos), ("p", l_srdir), ("p", sL["tex"]), ("p", sL["texaux"]), ("p", sL["pdf"]),
                                        ("p", sL["glob"]), ("i", n)], block=256)
            gens_l = L.down(l_gens, u32, (n, 2))
            lrev, srpos, srdir = L.down(l_lrev, f32, (3, n, 4)), L.down(l_srpos, f32, (n, 4)), L.down(l_srdir, f32, (n, 4))
            st.update(gens_light=gens_l.copy(), lrev=lrev, srpos=srpos, srdir=srdir)
            # ---- shadow traversal (the instanced early-out kernel; plain trees: the closest-hit form below maxDist)
            t_shadow = T.alloc(n * 8)
            T.launch("BVH4TraversalInstShadowKenrel" if self.have_inst else "BVH4TraversalShadowKenrel", n,
                     [("p", T.up(flags)), ("p", T.up(srpos)), ("p", T.up(srdir)), ("p", t_shadow), ("p", sT["bvh"]), ("p", sT["tris"]), ("p", sT["glob"]), ("i", 0), ("i", n)], block=256)
            shadow = T.down(t_shadow, np.uint16, (n, 4))
            st.update(shadow=shadow)
            # ---- Shade
            m_shade, m_shadow, m_lrev = M.alloc(n * 16), M.up(shadow), M.up(lrev)
            m_flags = M.up(flags)
            M.launch("Shade", n, [("p", m_rpos), ("p", m_rdir), ("p", m_flags), ("p", m_surf), ("p", m_shadow), ("p", m_lrev), ("p", m_ptl), ("p", 0), ("p", 0),
                                  ("p", m_shade), ("p", 0), ("p", sM["tex"]), ("p", sM["texaux"]), ("p", sM["mat"]), ("p", sM["pdf"]), ("p", sM["glob"]), ("i", n)], block=256)
            shade = M.down(m_shade, f32, (n, 4))
            st.update(shade=shade)
            # ---- NextBounce
            m_gens, m_fog = M.up(gens_l), M.up(fog)
            m_color, m_thr, m_mis = M.up(color), M.up(thr), M.up(mis)
            m_rpos2, m_rdir2 = M.up(rpos), M.up(rdir)
            M.launch("NextBounce", n, [("p", 0), ("p", 0), ("p", m_rpos2), ("p", m_rdir2), ("p", m_flags), ("p", m_gens), ("p", m_surf), ("p", m_ptl),
                                       ("p", m_color), ("p", m_thr), ("p", m_mis), ("p", m_shadow), ("p", m_fog), ("p", m_shade), ("p", m_emis), ("p", 0), ("p", 0),
                                       ("p", sM["tex"]), ("p", sM["texaux"]), ("p", sM["mat"]), ("p", sM["pdf"]), ("p", sM["glob"]), ("i", n)], block=256)
            rpos, rdir = M.down(m_rpos2, f32, (n, 4)), M.down(m_rdir2, f32, (n, 4))
            flags, gens = M.down(m_flags, u32, (n,)), M.down(m_gens, u32, (n, 2))
            color, thr, mis, fog = M.down(m_color, f32, (n, 4)), M.down(m_thr, f32, (n, 4)), M.down(m_mis, f32, (n, 4)), M.down(m_fog, f32, (n, 4))
            st.update(rpos_out=rpos.copy(), rdir_out=rdir.copy(), flags_out=flags.copy(), gens_out=gens.copy(), color_out=color.copy(), thr_out=thr.copy(), mis_out=mis.copy())
            rec.append(st)
            for m in (T, M, L, R):      # the per-bounce buffers; the scene buffers stay
                pass
        return rec


class RefMmltWavefront:
    """The reference's OWN MMLT stage kernels (hydra_drv/shaders/mlt.cl, compiled unmodified by oracle/build_ref.sh) launched in the order of its host
    loop GPUOCLLayer::EvalSBDPT (GPUOCLLayerAdvanced.cpp:949-1024): MMLTMakeEyeRays -> MMLTInitCameraPath -> [traverse -> ComputeHit ->
    MMLTCameraPathBounce] x maxBounce -> MMLTLightSampleForward -> [traverse -> ComputeHit -> MMLTLightPathBounce] x (maxBounce - 1) ->
    MMLTMakeShadowRay -> shadow traversal -> MMLTConnect, for n primary-sample vectors with their (d, s) handed in.

    The OpenCL layer keeps a vector transposed (number k of state i at [k * n + i]) and stores the ten numbers of a bounce as six packed words (four 24-bit
    and six 16-bit integers, crandom.h:262-345, mlt.cl:637-660): `slots` holds those integers; float value = integer x (1 / 16777215) resp. x (1 / 65535), which is what
    the CPU side (IntegratorMMLT, the oracle) is handed as its unpacked vector -- see unpacked()."""

    def __init__(self, b, device=0):
        self.b = b
        self.mods = {k: RefModule(k + ".hsaco", device) for k in ("trace", "mlt")}
        self.have_inst = int(b["have_inst"])

    def close(self):
        for m in self.mods.values():
            m.close()

    @staticmethod
    def pack(slots):
        """[n, nslots, 10] integers (0..3: 24-bit, 4..9: 16-bit) -> [n, nslots, 6] uint32 words (packBounceGroup / packBounceGroup2)"""
        s = np.asarray(slots, np.uint32)
        ix0, ix1, ix2, ix3, iy0, iy1 = (s[..., k] for k in range(6))
        w = np.empty(s.shape[:-1] + (6,), np.uint32)
        w[..., 0] = (ix0 & 0x00FFFFFF) | ((ix1 & 0x00FF0000) << 8)
        w[..., 1] = (ix1 & 0x0000FFFF) | ((iy0 & 0x0000FFFF) << 16)
        w[..., 2] = (ix2 & 0x00FFFFFF) | ((ix3 & 0x00FF0000) << 8)
        w[..., 3] = (ix3 & 0x0000FFFF) | ((iy1 & 0x0000FFFF) << 16)
        w[..., 4] = s[..., 6] | (s[..., 7] << 16)
        w[..., 5] = s[..., 8] | (s[..., 9] << 16)
        return w

    @staticmethod
    def unpacked(head, slots):
        """the vector as the CPU integrator reads it: 12 head numbers + 10 floats per bounce slot (unpackBounceGroup's arithmetic)"""
        s = np.asarray(slots, np.uint32).astype(np.float32)
        f = np.empty(s.shape, np.float32)
        f[..., :4] = s[..., :4] * np.float32(1.0 / 16777215.0)
        f[..., 4:] = s[..., 4:] * np.float32(1.0 / 65535.0)
        return np.concatenate([np.asarray(head, np.float32), f.reshape(len(head), -1)], axis=1)

    def run(self, depth, split, head, slots):
        b, n = self.b, len(depth)
        T, M = self.mods["trace"], self.mods["mlt"]
        f32, i32, u32 = np.float32, np.int32, np.uint32
        max_d = int(np.max(depth))

        def scene(m):
            d = dict(glob=m.up(b["globals"]), mat=m.up(b["materials"]), tex=m.up(b["textures"]), geom=m.up(b["geom"]),
                     pdf=m.up(b["pdfs"] if b["pdfs"].size else np.zeros(4, f32)), bvh=m.up(b["bvh_nodes"]), tris=m.up(b["bvh_tris"]),
                     matrices=m.up(b["inst_matrices"]), light_id=m.up(b["inst_light_id"]),
                     texaux=m.up(b["textures_aux"]) if b.get("textures_aux", np.zeros(0)).size else 0,
                     remap_lists=m.up(b["remap_lists"] if b["remap_lists"].size else np.zeros(4, i32)),
                     remap_table=m.up(b["remap_table"] if b["remap_table"].size else np.zeros(4, i32)),
                     remap_inst=m.up(b["remap_inst"] if b["remap_inst"].size else np.zeros(4, i32)))
            if not d["texaux"]:
                d["texaux"] = d["tex"]
            return d
        sT, sM = scene(T), scene(M)
        n_remap_table = b["remap_table"].size // 2
        n_inst = b["inst_matrices"].size // 16
        width, height = int(b["width"]), int(b["height"])
        lsc = float(width * height)
        # the vector, transposed: 12 head numbers, then 6 words per slot
        words = self.pack(slots)
        nslots = words.shape[1]
        vec = np.empty((12 + 6 * nslots, n), f32)
        vec[:12] = np.asarray(head, f32).T
        vec[12:] = words.reshape(n, nslots * 6).T.view(f32)
        morton = np.array([sum(((i >> k) & 1) << (2 * k) for k in range(8)) for i in range(256)], np.uint16)
        m_vec, m_morton = M.up(vec), M.up(morton)
        m_split = M.up(np.stack([depth, split], 1).astype(i32))
        m_rpos, m_rdir, m_zind = M.alloc(n * 16), M.alloc(n * 16), M.alloc(n * 8)
        m_flags, m_color = M.alloc(n * 4), M.alloc(n * 16)
        m_cvsup, m_lvsup = M.alloc(n * 32), M.alloc(n * 32)
        m_pdf = M.up(np.ones((max_d + 2, n, 2), f32))
        m_gens = M.up(np.zeros((n, 2), u32))
        mis0 = np.zeros((n, 4), f32); mis0[:, 0] = 1.0; mis0[:, 1] = 1.0          # makeInitialMisData (ClearAllInternalTempBuffers, screen.cl:381-406)
        mis0.view(i32)[:, 2] = -1; mis0.view(i32)[:, 3] = 1
        m_mis, m_fog = M.up(mis0), M.up(np.zeros((n, 4), f32))
        M.launch("MMLTMakeEyeRays", n, [("p", m_vec), ("p", m_rpos), ("p", m_rdir), ("p", m_zind), ("p", m_morton), ("p", sM["glob"]), ("i", n)], block=256)
        M.launch("MMLTInitCameraPath", n, [("p", m_flags), ("p", m_color), ("p", m_split), ("p", m_cvsup), ("p", m_pdf), ("i", n)], block=256)
        hit_dt = np.dtype([("t", f32), ("primId", i32), ("instId", i32), ("geomId", i32)])

        def trace_and_hit():
            rpos, rdir, flags = M.down(m_rpos, f32, (n, 4)), M.down(m_rdir, f32, (n, 4)), M.down(m_flags, u32, (n,))
            d_rpos, d_rdir, d_flags, d_hits, d_surf = T.up(rpos), T.up(rdir), T.up(flags), T.alloc(n * 16), T.alloc(n * 64)
            T.launch("BVH4TraversalInstKernel" if self.have_inst else "BVH4TraversalKernel", n,
                     [("p", d_rpos), ("p", d_rdir), ("p", sT["bvh"]), ("p", sT["tris"]), ("p", d_flags), ("p", d_hits), ("i", 0), ("i", n)], block=256)
            T.launch("ComputeHit", n, [("p", d_rpos), ("p", d_rdir), ("p", d_hits), ("p", sT["matrices"]), ("p", sT["geom"]), ("p", sT["mat"]),
                                       ("p", sT["remap_lists"]), ("p", sT["remap_table"]), ("p", sT["remap_inst"]), ("p", d_flags), ("p", d_surf),
                                       ("p", sT["glob"]), ("i", n_remap_table), ("i", n_inst), ("i", n)], block=256)
            hits, surf, flags = T.down(d_hits, hit_dt, (n,)), T.down(d_surf, f32, (4, n, 4)), T.down(d_flags, u32, (n,))
            return hits, surf, flags
        # (1) camera pass
        cv_surf = np.zeros((4, n, 4), f32)
        m_cvhit = M.up(cv_surf)
        for bounce in range(1, max_d + 1):
            hits, surf, flags = trace_and_hit()
            # the layer's ComputeHit writes into cameraVertexHit for the threads still running; a finished thread keeps the vertex it ended on
            act = ((flags >> 16) & (4096 | 128)) == 0
            prev = M.down(m_cvhit, f32, (4, n, 4))
            prev[:, act] = surf[:, act]
            m_cvhit = M.up(prev)
            m_hits, m_fl = M.up(hits), M.up(flags)
            m_flags = m_fl
            M.launch("MMLTCameraPathBounce", n, [("p", m_rpos), ("p", m_rdir), ("p", m_flags), ("p", m_gens), ("p", m_vec), ("p", m_split), ("p", m_hits), ("p", sM["light_id"]),
                                                 ("p", m_cvhit), ("p", 0), ("p", m_color), ("p", m_mis), ("p", m_fog), ("p", m_pdf), ("p", m_cvsup),
                                                 ("p", sM["tex"]), ("p", sM["texaux"]), ("p", sM["mat"]), ("p", sM["pdf"]), ("p", sM["glob"]), ("i", n), ("f", lsc)], block=256)
        cv_sup = M.down(m_cvsup, f32, (2, n, 4))
        # (2) light pass
        M.launch("MMLTLightSampleForward", n, [("p", m_rpos), ("p", m_rdir), ("p", m_flags), ("p", m_gens), ("p", m_vec), ("p", m_color), ("p", m_pdf), ("p", m_lvsup), ("p", m_mis),
                                               ("p", sM["tex"]), ("p", sM["pdf"]), ("p", sM["glob"]), ("i", n)], block=256)
        m_lvhit = M.up(np.zeros((4, n, 4), f32))
        for bounce in range(1, max_d):
            hits, surf, flags = trace_and_hit()
            act = ((flags >> 16) & (4096 | 128)) == 0
            prev = M.down(m_lvhit, f32, (4, n, 4))
            prev[:, act] = surf[:, act]
            m_lvhit = M.up(prev)
            m_flags = M.up(flags)
            M.launch("MMLTLightPathBounce", n, [("p", m_rpos), ("p", m_rdir), ("p", m_flags), ("p", m_gens), ("p", m_vec), ("p", m_split), ("p", m_lvhit), ("p", 0),
                                                ("p", m_color), ("p", m_mis), ("p", m_fog), ("p", m_pdf), ("p", m_lvsup),
                                                ("p", sM["tex"]), ("p", sM["texaux"]), ("p", sM["mat"]), ("p", sM["pdf"]), ("p", sM["glob"]), ("i", n)], block=256)
        lv_sup = M.down(m_lvsup, f32, (2, n, 4))
        # (3) connection
        m_srpos, m_srdir, m_srflags, m_lssam = M.alloc(n * 16), M.alloc(n * 16), M.alloc(n * 4), M.up(np.zeros((3, n, 4), f32))
        M.launch("MMLTMakeShadowRay", n, [("p", m_split), ("p", m_lvhit), ("p", m_lvsup), ("p", m_cvhit), ("p", m_cvsup), ("p", m_srpos), ("p", m_srdir), ("p", m_srflags), ("p", m_lssam),
                                          ("p", m_gens), ("p", m_vec), ("p", sM["mat"]), ("p", sM["pdf"]), ("p", sM["tex"]), ("p", sM["glob"]), ("i", n)], block=256)
        srpos, srdir, srflags = M.down(m_srpos, f32, (n, 4)), M.down(m_srdir, f32, (n, 4)), M.down(m_srflags, u32, (n,))
        t_shadow = T.alloc(n * 8)
        T.launch("BVH4TraversalInstShadowKenrel" if self.have_inst else "BVH4TraversalShadowKenrel", n,
                 [("p", T.up(srflags)), ("p", T.up(srpos)), ("p", T.up(srdir)), ("p", t_shadow), ("p", sT["bvh"]), ("p", sT["tris"]), ("p", sT["glob"]), ("i", 0), ("i", n)], block=256)
        shadow = T.down(t_shadow, np.uint16, (n, 4))
        m_out = M.alloc(n * 16)
        m_scale = M.up(np.ones(max_d + 2, f32))
        M.launch("MMLTConnect", n, [("p", m_split), ("p", m_lvhit), ("p", m_lvsup), ("p", m_cvhit), ("p", m_cvsup), ("p", 0), ("p", M.up(shadow)), ("p", m_lssam), ("p", m_pdf),
                                    ("p", m_out), ("p", m_zind), ("p", sM["tex"]), ("p", sM["texaux"]), ("p", sM["mat"]), ("p", sM["pdf"]), ("p", sM["glob"]), ("p", m_scale), ("p", m_morton),
                                    ("i", n), ("f", lsc), ("i", n)], block=256)
        out = M.down(m_out, f32, (n, 4))
        xy = out[:, 3].view(u32)
        return dict(color=out[:, :3].copy(), x=(xy & 0xFFFF).astype(i32), y=(xy >> 16).astype(i32), cv_sup=cv_sup, lv_sup=lv_sup, shadow=shadow,
                    pdf=M.down(m_pdf, f32, (max_d + 2, n, 2)), srpos=srpos, srdir=srdir)


def ref_mmlt_accept_reject(x_color4, y_color4, depth, scale_table, gens2, max_bounce=2, device=0):
    """The reference's own MMLTAcceptReject (shaders/mlt.cl:205-262, compiled unmodified by oracle/build_ref.sh) on n chains: current and proposed colour (w = the packed
    pixel, passed through), path length d per chain, the per-length scale table, the accept-test generators.  The x / y vectors are 0 / 1 everywhere, so an accepted chain
    shows ones.  -> dict(x_alpha, y_alpha [n, 4], x_color [n, 4] after, gens [n, 2] after, accepted [n] bool)"""
    m = RefModule("mlt.hsaco", device)
    n = len(x_color4)
    planes = 12 + 6 * max_bounce                                     # MMLT_HEAD_TOTAL_SIZE + MMLT_COMPRESSED_F_PERB per bounce
    xv, yv = m.up(np.zeros((planes, n), np.float32)), m.up(np.ones((planes, n), np.float32))
    xc, yc = m.up(np.ascontiguousarray(x_color4, np.float32)), m.up(np.ascontiguousarray(y_color4, np.float32))
    split = np.zeros((n, 2), np.int32)
    split[:, 0] = depth
    g = m.up(np.ascontiguousarray(gens2, np.uint32))
    xa, ya = m.alloc(n * 16), m.alloc(n * 16)
    m.launch("MMLTAcceptReject", n, [("p", xv), ("p", yv), ("p", xc), ("p", yc), ("p", m.up(split)), ("p", m.up(np.ascontiguousarray(scale_table, np.float32))),
                                     ("p", g), ("p", xa), ("p", ya), ("i", max_bounce), ("i", n)], block=256)
    out = dict(x_alpha=m.down(xa, np.float32, (n, 4)), y_alpha=m.down(ya, np.float32, (n, 4)), x_color=m.down(xc, np.float32, (n, 4)),
               gens=m.down(g, np.uint32, (n, 2)), accepted=m.down(xv, np.float32, (planes, n)).min(axis=0) == 1.0)
    assert (m.down(xv, np.float32, (planes, n)).max(axis=0) == out["accepted"]).all()      # a chain takes the whole proposal or nothing
    m.close()
    return out


class RefGBufferKernels:
    """The reference's OWN G-buffer kernels in the order of GPUOCLLayer::EvalGBuffer (GPUOCLLayerOther.cpp:742-765): MakeEyeRaysSPP (screen.cl:35, GBUFFER_SAMPLES = 64 plane
    Hammersley samples per pixel) -> traversal -> ComputeHit -> GetGBufferSample (material.cl:1347, one work-group of 64 per pixel), compiled unmodified.  Returns the two packed
    float4 layers.  (The layer's further steps -- transparent bounces and PutAlphaToGBuffer for the alpha channel -- have no counterpart in CPUExp_GBuffer.cpp and are not run.)"""

    def __init__(self, b, device=0):
        self.b = b
        self.mods = {k: RefModule(k + ".hsaco", device) for k in ("trace", "material", "screen")}
        self.have_inst = int(b["have_inst"])

    def close(self):
        for m in self.mods.values():
            m.close()

    def run(self):
        b = self.b
        T, M, S = self.mods["trace"], self.mods["material"], self.mods["screen"]
        f32, i32, u32 = np.float32, np.int32, np.uint32
        w, h = int(b["width"]), int(b["height"])
        n = w * h * 64
        glob, mat, tex = T.up(b["globals"]), T.up(b["materials"]), T.up(b["textures"])     # one device, one context: pointers are good in every module
        texaux = T.up(b["textures_aux"]) if b.get("textures_aux", np.zeros(0)).size else tex
        geom, bvh, tris, matrices = T.up(b["geom"]), T.up(b["bvh_nodes"]), T.up(b["bvh_tris"]), T.up(b["inst_matrices"])
        remap_lists = T.up(b["remap_lists"] if b["remap_lists"].size else np.zeros(4, i32))
        remap_table = T.up(b["remap_table"] if b["remap_table"].size else np.zeros(4, i32))
        remap_inst = T.up(b["remap_inst"] if b["remap_inst"].size else np.zeros(4, i32))
        rpos, rdir, flags, hits, surf = T.alloc(n * 16), T.alloc(n * 16), T.alloc(n * 4), T.alloc(n * 16), T.alloc(n * 64)
        S.launch("MakeEyeRaysSPP", n, [("p", rpos), ("p", rdir), ("i", w), ("i", h), ("i", 64), ("i", 0), ("p", T.up(plane_hammersley(64))), ("p", glob)], block=256)
        T.launch("BVH4TraversalInstKernel" if self.have_inst else "BVH4TraversalKernel", n,
                 [("p", rpos), ("p", rdir), ("p", bvh), ("p", tris), ("p", flags), ("p", hits), ("i", 0), ("i", n)], block=256)
        T.launch("ComputeHit", n, [("p", rpos), ("p", rdir), ("p", hits), ("p", matrices), ("p", geom), ("p", mat), ("p", remap_lists), ("p", remap_table), ("p", remap_inst),
                                   ("p", flags), ("p", surf), ("p", glob), ("i", b["remap_table"].size // 2), ("i", b["inst_matrices"].size // 16), ("i", n)], block=256)
        g1, g2 = T.alloc(w * h * 16), T.alloc(w * h * 16)
        M.launch("GetGBufferSample", n, [("p", rdir), ("p", hits), ("p", flags), ("p", surf), ("p", 0), ("p", g1), ("p", g2), ("p", mat), ("p", tex), ("p", texaux), ("p", glob), ("i", n)], block=64)
        return T.down(g1, f32, (h, w, 4)), T.down(g2, f32, (h, w, 4))
