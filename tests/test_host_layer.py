"""CPU: the scene front end packs buffers with the reference's layouts, and the BVH4 builder emits a valid tree."""
import os

import numpy as np
import pytest

from conftest import host_scene, make_oracle, random_rays

G_LIGHTS_OFFS, G_LIGHTS_NUM, G_MAT_TABLE, G_GEOM_TABLE, G_TEX_TABLE, G_LSEL_REV_SIZE = 236, 238, 219, 221, 218, 231
G_VARS_I, G_VARS_F, G_FLAGS, G_SKY = 64, 128, 234, 235


def test_globals_blob_layout(t224_small):
    _, b = t224_small
    g = b["globals"]
    gf = g.view(np.float32)
    assert g.size >= 134400 and g[G_MAT_TABLE] == 134400            # header rounded up to 16 words
    assert g[G_LIGHTS_NUM] == 2 and g[G_SKY] == -1
    assert g[G_VARS_I + 9] == 5                                      # HRT_TRACE_DEPTH = trace_depth + 1
    assert g[G_FLAGS] & 32 and g[G_FLAGS] & 1                        # HRT_USE_MIS | HRT_COMPUTE_SHADOWS
    assert g[G_LSEL_REV_SIZE] == 3                                   # prefix sums of 2 lights
    assert abs(gf[G_VARS_F + 14] - np.deg2rad(45.0)) < 1e-6          # HRT_CAM_FOV
    assert (g[192:208] == -1).all()                                  # rmQMC
    lights = gf[g[G_LIGHTS_OFFS]:g[G_LIGHTS_OFFS] + 256].reshape(2, 128)
    assert (lights[:, 0].view(np.int32) == 4).all()                  # PLAIN_LIGHT_TYPE_AREA
    np.testing.assert_allclose(lights[:, 8:11], 160.0)               # colour x multiplier
    np.testing.assert_allclose(lights[:, 14:16], [[0.25, 0.5]] * 2)  # half sizes
    np.testing.assert_allclose(lights[:, 13], 0.5, rtol=1e-5)        # area of a 0.5 x 1 rectangle
    np.testing.assert_allclose(lights[:, 107], 0.5)                  # normalised pick probability
    np.testing.assert_allclose(lights[1, 2:5], [3.75, 3.95, -3.5])   # instance translation
    np.testing.assert_allclose(lights[1, 5:8], [0, -1, 0])


def test_material_arena(t224_small):
    _, b = t224_small
    g, m = b["globals"], b["materials"].reshape(-1, 192)
    table = g[g[G_MAT_TABLE]:g[G_MAT_TABLE] + 12]
    mi = m.view(np.int32)
    assert table[11] == 0 and mi[0, 0] == 7                          # white diffuse dummy first in the arena
    blend = table[1] * 4 // 192                                      # material 1: blend(phong, lambert)
    assert mi[blend, 0] == 9 and mi[blend, 16] == 1 and mi[blend, 17] == 2
    assert mi[blend + 1, 0] == 0 and mi[blend + 2, 0] == 7
    assert mi[blend, 15] == (4 | 8)                                  # REFLECTION_WEIGHT_IS_ONE | EXTRUSION_STRONG
    assert mi[blend, 1] & 2                                          # caustics flag popped up from the phong child
    np.testing.assert_allclose(m[blend, 10:13], [0.367059, 0.345882, 0.0], rtol=1e-6)
    assert mi[blend, 20] == 0 and mi[blend, 19] == -1                # sampler.flags clobbered by FALOFF_SIZE, FALOFF_OFFSET = -1
    assert abs(m[blend, 21] - 2.2) < 1e-6                            # BLEND_TYPE clobbered by sampler.gamma
    em = table[10] * 4 // 192
    assert mi[em, 0] == 10 and mi[em, 9] == 0
    np.testing.assert_allclose(m[em, 4:7], 160.0)
    tex = table[0] * 4 // 192                                        # textured lambert: sampler at float 20
    assert mi[tex, 13] == 2 and mi[tex, 14] == 5 and mi[tex, 22] == 2
    plain = table[6] * 4 // 192
    assert mi[plain, 14] == -2 and mi[plain, 13] == -2               # INVALID_TEXTURE


def test_geometry_arena(t224_small):
    _, b = t224_small
    g, geom = b["globals"], b["geom"]
    offs = g[g[G_GEOM_TABLE]:g[G_GEOM_TABLE] + 6]
    hdr = geom.view(np.int32)[offs[1] * 4: offs[1] * 4 + 16]         # box
    assert hdr[4] == 20 and hdr[7] == 30 and hdr[9] == 10
    assert hdr[0] == 4 and hdr[1] == 4 + 20 and hdr[10] == 4 + 40    # header is 64 B = 4 float4
    idx = geom.view(np.int32)[(offs[1] + hdr[3]) * 4:(offs[1] + hdr[3]) * 4 + 30]
    assert idx.max() == 19
    mats = geom.view(np.int32)[(offs[1] + hdr[8]) * 4:(offs[1] + hdr[8]) * 4 + 10]
    assert list(mats) == [6, 6, 7, 7, 8, 8, 8, 9, 9, 9]
    so = geom[(offs[1] + hdr[13]) * 4:(offs[1] + hdr[13]) * 4 + 10]
    assert (so == 0).all()                                           # flat polygons: no auxiliary shadow offset
    thdr = geom.view(np.int32)[offs[0] * 4: offs[0] * 4 + 16]        # teapot: smooth normals => positive offsets
    tso = geom[(offs[0] + thdr[13]) * 4:(offs[0] + thdr[13]) * 4 + thdr[9]]
    assert (tso > 0).mean() > 0.9 and tso.max() <= 0.00025 * 2.1


def walk_bvh(nodes, tris):
    """python walk of the flattened layout; returns per mesh the primitive ids found, and checks box nesting"""
    n = nodes.reshape(-1, 8)
    ni = n.view(np.int32)
    found = {}
    seen_sub = set()

    def leaf(list_off, mesh_check=None):
        h = tris.view(np.int32)[list_off * 4: list_off * 4 + 4]
        first, cnt = h[0], h[1]
        assert first == list_off + 1 and h[2] == -1 and h[3] == -1
        t = tris[first * 4:(first + cnt * 3) * 4].reshape(cnt, 3, 4)
        return t

    def sub(node_idx, box_lo, box_hi, mesh):
        link = int(ni[node_idx, 3]) & 0xFFFFFFFF
        off = link & 0x7fffffff
        if link & 0x80000000:
            t = leaf(off)
            assert (t[:, :, :3] >= box_lo - 1e-5).all() and (t[:, :, :3] <= box_hi + 1e-5).all()
            ids = t[:, 0, 3].view(np.int32)
            assert (t[:, 1, 3].view(np.int32) == mesh).all() and (t[:, 2, 3].view(np.int32) == -1).all()
            found.setdefault(mesh, []).extend(ids.tolist())
            return
        for k in range(4):
            c = off * 4 + k
            if ni[c, 3] == -1 and ni[c, 7] == -1:
                continue
            lo, hi = n[c, 0:3], n[c, 4:7]
            assert (lo >= box_lo - 1e-5).all() and (hi <= box_hi + 1e-5).all()
            sub(c, lo, hi, mesh)

    insts = []

    def top(node_idx):
        link = int(ni[node_idx, 3]) & 0xFFFFFFFF
        off = link & 0x7fffffff
        if link & 0x80000000:                     # instance leaf
            assert ni[node_idx, 7] == 1
            q = off * 4
            inst_id, mesh = int(ni[q + 3, 0]), int(ni[q + 3, 1])
            insts.append(inst_id)
            if (int(mesh), int(ni[q, 3])) not in seen_sub:
                seen_sub.add((int(mesh), int(ni[q, 3])))
                sub(q, n[q, 0:3], n[q, 4:7], mesh)
            return
        for k in range(4):
            c = off * 4 + k
            if ni[c, 3] == -1 and ni[c, 7] == -1:
                continue
            top(c)

    assert (int(ni[0, 3]) & 0x7fffffff) == 1          # traversal starts at quad 1
    top(0)
    return found, insts


def test_bvh_layout_and_coverage(t224_small):
    sc, b = t224_small
    found, insts = walk_bvh(b["bvh_nodes"], b["bvh_tris"])
    assert sorted(insts) == [0, 1, 2, 3]
    assert sorted(found[1]) == list(range(10))                       # box: all 10 triangles, once
    assert sorted(found[5]) == [0, 1]                                # light quad mesh shared by two instances
    teapot = found[0]
    assert len(teapot) == len(set(teapot)) and 25000 < len(teapot) <= 25600
    st = sc.bvh_stats()
    assert st["triangles"] == len(teapot) + 12


def test_oracle_traversal_matches_brute_force(t42_small):
    """box-only test_42: closest hit over the BVH == brute force over all instanced triangles (float64 reference)"""
    _, b = t42_small
    orc = make_oracle(b)
    pos4, dir4 = random_rays(2000, 7)
    hits = orc.trace(pos4, dir4)
    # world-space triangles from the packed arena + instance matrices
    g, geom = b["globals"], b["geom"]
    offs = g[g[G_GEOM_TABLE]:g[G_GEOM_TABLE] + 9]
    tri_lists = []
    inv = b["inst_matrices"].reshape(-1, 4, 4)                       # [inst][col][row]
    found, insts = walk_bvh(b["bvh_nodes"], b["bvh_tris"])
    n = b["bvh_nodes"].reshape(-1, 8)
    ni = n.view(np.int32)
    # instance -> mesh map from the instance quads
    inst_mesh = {}
    for q in range(ni.shape[0] // 4):
        if ni[q * 4 + 3, 2] == 0 and ni[q * 4 + 3, 3] == 0 and 0 <= ni[q * 4 + 3, 0] < inv.shape[0] and ni[q * 4 + 3, 1] in found:
            pass
    for node in range(ni.shape[0]):
        if ni[node, 7] == 1 and (int(ni[node, 3]) & 0x80000000):
            q = (int(ni[node, 3]) & 0x7fffffff) * 4
            inst_mesh[int(ni[q + 3, 0])] = int(ni[q + 3, 1])
    best_t = np.full(len(pos4), np.inf)
    for inst, mesh in inst_mesh.items():
        hdr = geom.view(np.int32)[offs[mesh] * 4: offs[mesh] * 4 + 16]
        v = geom[(offs[mesh] + hdr[0]) * 4:(offs[mesh] + hdr[0]) * 4 + hdr[4] * 4].reshape(-1, 4)[:, :3].astype(np.float64)
        idx = geom.view(np.int32)[(offs[mesh] + hdr[3]) * 4:(offs[mesh] + hdr[3]) * 4 + hdr[7]].reshape(-1, 3)
        M = np.linalg.inv(inv[inst].T.astype(np.float64))            # object -> world
        vw = (np.c_[v, np.ones(len(v))] @ M.T)[:, :3]
        A, B, Cc = vw[idx[:, 0]], vw[idx[:, 1]], vw[idx[:, 2]]
        o, d = pos4[:, None, :3].astype(np.float64), dir4[:, None, :3].astype(np.float64)
        e1, e2 = (B - A)[None], (Cc - A)[None]
        p = np.cross(d, e2)
        det = (e1 * p).sum(-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            invd = 1.0 / det
            tv = o - A[None]
            vv = (tv * p).sum(-1) * invd
            qv = np.cross(tv, e1)
            uu = (qv * d).sum(-1) * invd
            tt = (e2 * qv).sum(-1) * invd
        ok = (vv > -1e-6) & (uu > -1e-6) & (uu + vv < 1 + 1e-6) & (tt > 0)
        tt = np.where(ok, tt, np.inf)
        best_t = np.minimum(best_t, tt.min(axis=1))
    hit_mask = hits["primId"] != -1
    bf_mask = np.isfinite(best_t)
    assert (hit_mask == bf_mask).mean() > 0.999
    both = hit_mask & bf_mask
    np.testing.assert_allclose(hits["t"][both], best_t[both], rtol=2e-4)


def test_generated_atrium_scene_packs_textures_and_instances(atrium_small):
    sc, b = atrium_small
    assert sc.unsupported() == 0, sc.log()
    g = b["globals"]
    assert b["inst_matrices"].size // 16 == 173 and g[G_LIGHTS_NUM] == 1
    tex_table = g[g[G_TEX_TABLE]:g[G_TEX_TABLE] + 3]
    assert (tex_table >= 0).all()
    hdr = b["textures"][tex_table[1] * 4: tex_table[1] * 4 + 4]
    assert list(hdr) == [256, 256, 4, 4]                             # SWTextureHeader {w, h, channels, bpp}
    found, insts = walk_bvh(b["bvh_nodes"], b["bvh_tris"])
    assert sorted(insts) == list(range(173))
    st = sc.bvh_stats()
    assert st["triangles"] == sum(len(v) for v in found.values())
    orc = make_oracle(b)
    img, rays, _ = orc.render(2, seed=5)
    assert np.isfinite(img).all() and img[..., :3].mean() > 0.01 and rays > 2 * 96 * 54


def test_transparency_layers_pack_like_the_reference_converter():
    """CreateFromHydraMaterialXmlNode, PlainMaterialConverter.cpp:1541-1591: the four transparency shapes of the glass hall"""
    sc, b = host_scene("atrium_glass_small", 96, 54, 8)
    assert sc.unsupported() == 0, sc.log()
    g, m = b["globals"], b["materials"].reshape(-1, 192)
    mi = m.view(np.int32)
    table = g[g[G_MAT_TABLE]:g[G_MAT_TABLE] + 12]
    HAS_T, CAUSTICS, REFL_ONLY = 8, 2, 32768
    # material 3: reflection + transparency -> fresnel blend(mirror, glass), CAN_SAMPLE_REFL_ONLY
    n = table[3] * 4 // 192
    assert mi[n, 0] == 9 and mi[n + mi[n, 16], 0] == 2 and mi[n + mi[n, 17], 0] == 4
    assert mi[n, 15] & 1 and abs(m[n, 18] - 1.5) < 1e-6              # BLEND_MASK_FRESNEL, fresnel IOR
    assert (mi[n, 1] & (HAS_T | REFL_ONLY | CAUSTICS)) == (HAS_T | REFL_ONLY | CAUSTICS)
    gl = n + mi[n, 17]
    assert abs(m[gl, 15] - 1.5) < 1e-6 and abs(m[gl, 21] - 1.0) < 1e-6 and mi[gl, 13] == -2 and mi[gl, 22] == -2   # IOR, gloss, no textures
    assert mi[gl, 1] == (HAS_T | CAUSTICS)
    # material 2: transparency only -> a bare glass node (rough: gloss 0.7, ior 1.33)
    n = table[2] * 4 // 192
    assert mi[n, 0] == 4 and abs(m[n, 21] - 0.7) < 1e-6 and abs(m[n, 15] - 1.33) < 1e-6
    # material 1: reflection + transparency + diffuse -> blend(blend(S, T), D), alpha of the outer mask = transparency colour
    n = table[1] * 4 // 192
    st, d = n + mi[n, 16], n + mi[n, 17]
    assert mi[n, 0] == 9 and mi[st, 0] == 9 and mi[d, 0] == 7
    assert mi[st + mi[st, 16], 0] == 0 and mi[st + mi[st, 17], 0] == 4
    np.testing.assert_allclose(m[n, 10:13], 0.5)
    assert (mi[n, 15] & 1) == 0 and (mi[st, 15] & 1) == 1 and (mi[st, 1] & REFL_ONLY) and (mi[n, 1] & REFL_ONLY) == 0
    # material 6: thin-walled + diffuse -> plain blend(thin glass, lambert) with strong extrusion; the transparency texture
    # drives both the mask and the thin-glass colour sampler (float 20)
    n = table[6] * 4 // 192
    tg = n + mi[n, 16]
    assert mi[n, 0] == 9 and mi[tg, 0] == 3 and mi[n + mi[n, 17], 0] == 7 and (mi[n, 15] & 8)
    assert mi[tg, 13] == 2 and mi[tg, 14] == 5 and abs(m[tg, 16] - 0.85) < 1e-6 and m[tg, 15] == 0.0
    assert mi[n, 13] == 2


def test_ggx_reflection_packs_like_the_reference_converter():
    """GGXMaterial, PlainMaterialConverter.cpp:635-680, chosen by brdf_type="ggx" (:1132-1133)"""
    from conftest import scene_path
    from hydracore_amd import HostScene
    sc = HostScene(scene_path("atrium_ggx_small"), 96, 54, trace_depth=8, enable_dof=0, use_hip=False)   # unpatched buffers
    assert sc.unsupported() == 0, sc.log()
    b = sc.buffers()
    g, m = b["globals"], b["materials"].reshape(-1, 192)
    mi = m.view(np.int32)
    table = g[g[G_MAT_TABLE]:g[G_MAT_TABLE] + 12]
    n = table[9] * 4 // 192                                          # Fresnel blend(GGX, lambert)
    gx = n + mi[n, 16]
    assert mi[n, 0] == 9 and mi[gx, 0] == 15 and mi[n + mi[n, 17], 0] == 7 and (mi[n, 15] & 1)
    assert abs(m[gx, 16] - 0.7) < 1e-6 and abs(m[gx, 19] - 2.5) < 1e-6 and m[gx, 15] == 0.0   # gloss, fresnel IOR, cosPower
    assert mi[gx, 13] == -2 and mi[gx, 17] == -2 and mi[gx, 1] == 2                              # no textures, CAST_CAUSTICS
    np.testing.assert_allclose(m[gx, 10:13], 0.8)
    # the header carries the layer's energy tables (here: the bake the host-blob layer reads, conftest.energy_tables_file), and material 8's
    # GGX lobe asked for the compensation in XML (<multiscatter val="1">: PLAIN_MATERIAL_ENERGY_FIX_OR_MULTISCATTER), material 9's did not
    import conftest
    want = np.load(os.path.join(conftest.ROOT, "tests", "golden", "energy_tables.npz"))
    assert (g.view(np.uint16)[1268 * 2: 1268 * 2 + 4096] == want["ggx"].ravel()).all() and (g.view(np.uint16)[3316 * 2: 3316 * 2 + 64 ** 3] == want["transp"].ravel()).all()
    n8 = table[8] * 4 // 192
    assert (mi[n8 + mi[n8, 16], 1] & (32768 * 256)) and not (mi[gx, 1] & (32768 * 256))
    sc.close()


REF_TESTS = "/root/reference/hydra_app/tests"


@pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="the reference tree is only present in the build container")
def test_every_scene_library_of_the_reference_loads():
    """row f1: the front end against ALL scene libraries the reference ships for its own tests (hydra_app/tests/*: 16 libraries; state XML, .vsgf
    meshes, .image4ub / .image4f textures, read in place -- nothing is copied).  Every one must be packed without an unsupported feature
    (rect / sphere / point / directional / sky lights, lambert / phong / GGX / Beckmann / TRGGX / mirror / glass / blends, height and normal bump ...).
    014_Bump_height is the exception that proves the rule: its float environment texture chunk_00004.image4f is one of the blobs missing from this
    copy of the reference (.MISSING_LARGE_BLOBS), which the front end reports by name instead of rendering a black sky."""
    from hydracore_amd import HostScene, HydraError
    names = sorted(d for d in os.listdir(REF_TESTS) if any(f.startswith("statex") for f in os.listdir(os.path.join(REF_TESTS, d))))
    assert len(names) >= 16, names
    for name in names:
        try:
            sc = HostScene(os.path.join(REF_TESTS, name), 128, 96, trace_depth=4, enable_dof=0, use_hip=False)
        except HydraError as e:
            assert name == "014_Bump_height" and "texture 2 is not loaded" in str(e), (name, str(e))
            continue
        assert sc.unsupported() == 0, (name, sc.log())
        b = sc.buffers()
        assert b["bvh_nodes"].size > 0 and b["globals"][238] >= 1, name            # a tree and at least one light
        # procedural textures: teapot_cylinder and test_aniso declare two whose data/proctex_*.c are not in the tree (no material binds them: logged, not fatal);
        # test_aniso2 has the files -- its program text is assembled, and the run-time compiler stops at the same line an OpenCL compiler does: falloff.c returns a float3
        # expression from a function declared float4 (the reference's own splice fails to build there too, with the image's clang)
        if name in ("teapot_cylinder", "test_aniso"):
            assert "code file 'data/proctex_00001.c' is missing" in sc.log() and sc.proctex_program() == ""
        if name == "test_aniso2":
            from hydracore_amd.capi import proctex_check
            text = sc.proctex_program()
            assert "prtex1_main" in text and "prtex2_main" in text and "texture2D(texX, x_uv, 0)" in text
            with pytest.raises(HydraError, match="float3"):
                proctex_check(text)
            cut = text.index("float3 prtex1_mix"), text.index("float3 prtex2_abs3")       # the hexaplanar texture alone (six samplers, pow, max, texture2D) builds
            eval_cut = text.index("    if(materialHeadHaveTargetProcTex(pHitMaterial,1)"), text.index("    if(materialHeadHaveTargetProcTex(pHitMaterial,2)")
            proctex_check(text[:cut[0]] + text[cut[1]:eval_cut[0]] + text[eval_cut[1]:])
        sc.close()


def test_two_matrices_camera(tmp_path):
    """a camera given as its two matrices (<camera type="two_matrices">, RenderDriverRTE.cpp:1178-1201, CalcCameraMatrices :1301-1324): the matrices
    the ordinary camera of test_224 leads to, written into a copy of its scene library, must give the same globals header -- projection and
    world-view matrices, their inverses and the field of view restored from the projection"""
    import re
    import shutil
    from conftest import scene_path
    from hydracore_amd import HostScene
    src = scene_path("test_224")
    a = HostScene(src, 128, 96, trace_depth=4, enable_dof=0, use_hip=False)
    ga = a.buffers()["globals"].copy()
    mats = ga[0:64].view(np.float32).reshape(4, 4, 4)              # mProj, mWorldView, mProjInverse, mWorldViewInverse: 16 floats each, column-major
    proj, view = mats[0].T, mats[1].T                              # row-major, as the XML wants them
    dst = tmp_path / "lib"
    dst.mkdir()
    os.symlink(os.path.join(src, "data"), dst / "data")
    xml = open(os.path.join(src, "statex_00001.xml")).read()
    cam = '<camera id="0" name="cam" type="two_matrices"><mWorldView>%s</mWorldView><mProj>%s</mProj></camera>' % (
        " ".join(repr(float(x)) for x in view.ravel()), " ".join(repr(float(x)) for x in proj.ravel()))
    xml2, n = re.subn(r"<camera\b.*?</camera>", cam, xml, count=1, flags=re.S)
    assert n == 1
    (dst / "statex_00001.xml").write_text(xml2)
    b = HostScene(str(dst), 128, 96, trace_depth=4, enable_dof=0, use_hip=False)
    assert b.unsupported() == 0, b.log()
    gb = b.buffers()["globals"]
    np.testing.assert_allclose(gb[0:64].view(np.float32), ga[0:64].view(np.float32), rtol=2e-5, atol=2e-6)
    fa, fb = ga[G_VARS_F:G_VARS_F + 64].view(np.float32), gb[G_VARS_F:G_VARS_F + 64].view(np.float32)
    np.testing.assert_allclose(fb[14], fa[14], rtol=1e-5)          # varsF[HRT_CAM_FOV]


def test_cylinder_and_textured_mesh_lights_pack_like_the_reference_converter():
    """CylinderLight + CreateCylinderLightFromXmlNode (PlainLightConverter.cpp:354-443, 867-893), its 2-D table (RenderDriverRTE.cpp:940-941 ->
    UpdatePdfTablesForLight, RenderDriverRTE_PdfTables.cpp:479-570) and the colour texture of a MeshLight (:770-782)"""
    _, b = host_scene("atrium_tubes_small", 96, 54, 5)
    g = b["globals"]
    gf = g.view(np.float32)
    n = g[G_LIGHTS_NUM]
    L = gf[g[G_LIGHTS_OFFS]:g[G_LIGHTS_OFFS] + n * 128].reshape(n, 128)
    Li = L.view(np.int32)
    assert list(Li[:, 0]) == [4, 6, 6, 7]                                            # area, cylinder, cylinder, mesh
    tube, half = L[1], L[2]
    np.testing.assert_allclose(tube[25:29], [0.15, -3.0, 3.0, 2 * np.pi], rtol=1e-6)  # radius, zMin, zMax, phiMax
    np.testing.assert_allclose(tube[13], 6.0 * 0.15 * 2 * np.pi, rtol=1e-6)           # (zMax - zMin) * radius * phiMax under a rigid instance matrix
    np.testing.assert_allclose(half[25:29], [0.3, -1.5, 1.5, np.pi], rtol=1e-6)       # the radius field keeps the light's own value ...
    np.testing.assert_allclose(half[13], 3.0 * 1.3 * (0.3 * 1.3) * np.pi, rtol=1e-5)  # ... the area takes the instance scale twice (:399-408)
    np.testing.assert_allclose(tube[8:11], 30.0)
    m = tube[16:25].reshape(3, 3)
    np.testing.assert_allclose(m @ m.T, np.eye(3), atol=1e-6)                         # the instance's rotation
    np.testing.assert_allclose(np.linalg.norm(half[16:25].reshape(3, 3), axis=0), 1.3, rtol=1e-5)
    # the colour texture: id 1 behind an identity sampler at int4 offset 8 of the record; none on the half tube
    assert Li[1, 29] == 1 and Li[1, 30] == 8 and Li[1, 32 + 2] == 1
    np.testing.assert_allclose(tube[32 + 4:32 + 12], [1, 0, 0, 0, 0, 1, 0, 0])
    assert np.uint32(Li[2, 29]) == 0xFFFFFFFE and np.uint32(Li[2, 30]) == 0xFFFFFFFE
    # the (z, phi) tables: luminance of the 256^2 checker / the 2 x 2 uniform image, as prefix sums behind {w, h, 1, n + 1}
    pdf = b["pdfs"] if "pdfs" in b else None
    tab = g[g[220]:g[220] + g[225]]                                                   # pdf table offsets (float4 units)
    for rec, (w, h) in ((Li[1], (256, 256)), (Li[2], (2, 2))):
        tid = rec[31]
        assert 0 <= tid < len(tab)
        if pdf is not None:
            t = pdf.reshape(-1)[tab[tid] * 4:]
            assert tuple(t[:4].view(np.int32)) == (w, h, 1, w * h + 2)
            acc = t[4:4 + w * h + 1]
            assert acc[0] == 0 and (np.diff(acc) > 0).all() and t[4 + w * h + 1] == 1.0
    # mesh light: colour texture 2 at the record's own sampler slot (MESH_LIGHT_TEX_ID / _TEXMATRIX_ID / _TEX_SAMPLER = 30 / 31 / 32)
    assert Li[3, 30] == 2 and Li[3, 31] == 8 and Li[3, 32 + 2] == 2
    np.testing.assert_allclose(L[:, 107], 0.25)


def test_sky_portal_packs_like_the_reference_converter():
    """AreaDiffuseLight with <sky_portal> (PlainLightConverter.cpp:197-258), SkyPortalMaterial (PlainMaterialConverter.cpp:304-350), the offsets
    RenderDriverRTE::BuildSkyPortalsDependencyDummyInstances writes (RenderDriverRTE.cpp:1653-1684), the pick table that leaves the portal's sky
    out (RenderDriverRTE_PdfTables.cpp:598-602, 631-635) and the header's sun table (IHWLayerDataAssembler.cpp:422-449)"""
    _, b = host_scene("atrium_portal_small", 96, 54, 5)
    g = b["globals"]
    gf = g.view(np.float32)
    n = g[G_LIGHTS_NUM]
    L = gf[g[G_LIGHTS_OFFS]:g[G_LIGHTS_OFFS] + n * 128].reshape(n, 128)
    Li = L.view(np.int32)
    assert list(Li[:, 0]) == [4, 3, 4, 2] and g[G_SKY] == 1                           # roof light, sky, portal, sun
    assert Li[2, 1] == 8 and Li[2, 30] == 1 and Li[2, 29] == -1                      # AREA_LIGHT_SKY_PORTAL, source light id, record offset to the sky
    assert np.uint32(Li[2, 11]) == 0xFFFFFFFE and np.uint32(Li[2, 31]) == 0xFFFFFFFE  # no texture of its own, no blurred texture
    np.testing.assert_allclose(L[2, 13], 16.0 * 8.0)
    np.testing.assert_allclose(L[:, 107], [1 / 3, 0.0, 1 / 3, 1 / 3], rtol=1e-6)      # the sky is never picked: the portal stands in for it
    np.testing.assert_allclose(L[:, 106], [1 / 3, 0.0, 1 / 3, 1 / 3], rtol=1e-6)
    assert g[G_VARS_I + 26] == 1                                                     # HRT_HRT_SCENE_HAVE_PORTALS
    assert g[242] == 8                                                               # sic: MAX_SUN_NUM copies of the one soft sun
    suns = gf[243:243 + 8 * 128].reshape(8, 128)
    assert (suns.view(np.int32) == Li[3]).all()
    # the portal's material: clear thin glass, PLAIN_MATERIAL_HAS_TRANSPARENCY | SKIP_SHADOW | SKIP_SKY_PORTAL, no emission of its own
    m = b["materials"].reshape(-1, 192)
    mi = m.view(np.int32)
    at = g[g[G_MAT_TABLE] + 12] * 4 // 192
    assert mi[at, 0] == 3 and mi[at, 1] == (8 | 256 | 1024)
    np.testing.assert_allclose(m[at, 10:13], 1.0)


def test_ies_lights_pack_like_the_reference_converter():
    """PointLight / AreaDiffuseLight with distribution="ies" (PlainLightConverter.cpp:162-266, 632-697), the lat-long image of the web (IESRender.cpp:29-200) and
    its two tables (AddIesTexTableToStorage, RenderDriverRTE_PdfTables.cpp:385-478), the IES frame carried through the instance matrix (TransformIESMatrix :113-124).
    The LM-63 reader itself is unpinned (the reference's parser and any .ies file are absent from its tree): the check here is against the generator's own function"""
    _, b = host_scene("atrium_ies_small", 96, 54, 5)
    g = b["globals"]
    gf = g.view(np.float32)
    n = g[G_LIGHTS_NUM]
    L = gf[g[G_LIGHTS_OFFS]:g[G_LIGHTS_OFFS] + n * 128].reshape(n, 128)
    Li = L.view(np.int32)
    assert list(Li[:, 0]) == [4, 0, 4, 4] and list(Li[:, 1]) == [0, 16, 16, 16 | 32]     # LIGHT_HAS_IES, + LIGHT_IES_POINT_AREA on the last
    assert Li[1, 127] == Li[3, 127] and Li[1, 126] == Li[3, 126] and Li[2, 127] != Li[1, 127]   # one image + table per file (the cache), ids TEX = 127, PDF = 126
    pdf, tab = b["pdfs"], g[g[220]:g[220] + g[225]]
    # file 1: the whole sphere, 19 x 9 angles -> a 9 x 19 image; pixel (phi index, theta index) holds the candela value, scaled to a maximum of 1
    t = pdf[tab[Li[1, 127]] * 4:]
    assert tuple(t[:4].view(np.int32)) == (9, 19, 1, 4)
    img = t[4:4 + 9 * 19].reshape(19, 9)
    fn = lambda th, ph: 800.0 * max(np.cos(th), 0.0) ** 2 * (1.0 + 0.6 * np.cos(ph - 1.05)) + 150.0 * max(-np.cos(th), 0.0) + 20.0
    want = np.array([[float("%.4f" % fn(np.radians(10.0 * iy), np.radians(45.0 * ix))) for ix in range(9)] for iy in range(19)])
    # the reference's own placement rule rounds theta / 180 * h and phi / 360 * w with steps of span / count: columns and rows land where the loop of IESRender.cpp:105-124 puts them
    assert np.isclose(img.max(), 1.0) and (img > 0).all()
    np.testing.assert_allclose(img[0, 0], want[0, 0] / want.max(), rtol=1e-5)
    s = pdf[tab[Li[1, 126]] * 4:]
    assert tuple(s[:4].view(np.int32)) == (9, 19, 1, 4)
    acc = s[4:4 + 9 * 19 + 1]
    assert acc[0] == 0 and (np.diff(acc) > 0).all()                                 # blurred web + 0.05 x mean: no cell without probability
    # file 2: a quadrant of the lower hemisphere (13 x 5 angles) -> 20 x 26, mirrored into the four quadrants, upper hemisphere dark
    t2 = pdf[tab[Li[2, 127]] * 4:]
    assert tuple(t2[:4].view(np.int32)) == (20, 26, 1, 4)
    img2 = t2[4:4 + 20 * 26].reshape(26, 20)
    assert (img2[14:] == 0).all() and (img2[:13].max(axis=0) > 0).sum() >= 16
    # the IES frame: rotation, its inverse in the second slot
    for k in (1, 2, 3):
        m, inv = L[k, 117:126].reshape(3, 3), L[k, 108:117].reshape(3, 3)
        np.testing.assert_allclose(m @ inv, np.eye(3) * (m @ inv)[0, 0], atol=1e-5)
        np.testing.assert_allclose(m @ inv, np.eye(3), atol=1e-5)


def test_hydra_blend_materials_compose_materials_of_the_library(built):
    """hydra_blend (CreateBlendDefferedProxyFromXmlNode + EndMaterialUpdate, PlainMaterialConverter.cpp:1457-1500, 1787-1842): a blend-mask node over the converted trees of
    node_top / node_bottom, resolved after all other materials and in id order -- children with higher ids than the blend, and a blend of a blend, both work; mask texture
    and sampler matrix, Fresnel flag with the blend's own IOR, the extrusion read from the material node, white colour factor"""
    from conftest import host_scene
    sc, b = host_scene("atrium_blend_small", 96, 54, 5)
    assert sc.unsupported() == 0, sc.log()
    g, mats = b["globals"], b["materials"].reshape(-1)
    mi = mats.view(np.int32)
    table = g[g[219]:g[219] + g[224]]

    def node(mid, k=0):
        o = int(table[mid]) * 4 + 192 * k
        return mats[o:o + 192], mi[o:o + 192]
    f1, i1 = node(1)
    f12, i12 = node(12)
    assert i1[0] == 9 and (i1[16], i1[17]) == (1, 1 + 3)                     # [blend][material 12: blend + 2 leaves][material 13]
    assert (node(1, 1)[1] == i12).all() and (node(1, 4)[1] == node(13)[1]).all()               # compared as words: invalid ids are NaN patterns
    HM_COLOR, HM_TEXID, HM_TEXMATRIXID = 10, 13, 14                          # include/hydra_layouts.h
    sampler = i1[HM_TEXMATRIXID] * 4                                         # the embedded sampler: texture 2, matrix scale 3
    assert i1[HM_TEXID] == 2 and i1[sampler + 2] == 2 and f1[sampler + 4] == 3.0 and f1[sampler + 9] == 3.0
    assert (i1[15] & 1) == 0 and tuple(f1[HM_COLOR:HM_COLOR + 3]) == (1.0, 1.0, 1.0)   # BLEND_MASK_FLAGS: a plain mask; white colour factor
    f3, i3 = node(3)
    assert i3[0] == 9 and (i3[15] & 1) == 1 and (i3[15] & 16) == 16 and np.isclose(f3[18], 1.8)   # Fresnel blend, luminance extrusion, its IOR (BLEND_MASK_FRESNEL_IOR)
    assert (node(3, 1)[1] == node(14)[1]).all() and (node(3, 2)[1] == node(15)[1]).all()
    f5, i5 = node(5)
    assert i5[0] == 9 and (i5[16], i5[17]) == (1, 6)                          # blend over [material 1's five nodes][material 13]
    for k in range(5):
        assert (node(5, 1 + k)[1] == node(1, k)[1]).all()
    assert (node(5, 6)[1] == node(13)[1]).all()
