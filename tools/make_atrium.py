#!/usr/bin/env python3
"""Procedural "atrium250k" scene (BASELINE.json configs[2]: Sponza-class, ~250 k triangles, deep BVH).

Writes a HydraAPI scene library (statex_00001.xml + data/chunk_*.vsgf / .image4ub) -- the same on-disk format as the
reference's hydra_app/tests/* fixtures -- so it goes through the same front end as the reference scenes.
Everything is generated from one seed (20250213) with numpy; nothing is downloaded.

    python tools/make_atrium.py /tmp/atrium250k            # full size (~249 k triangles incl. instances)
    python tools/make_atrium.py /tmp/atrium_small --scale 0.25
"""
import argparse
import os
import struct

import numpy as np

SEED = 20250213


def grid_indices(nu, nv):
    """triangles of a (nu+1) x (nv+1) vertex grid, row-major in v"""
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel()
    b = a + (nv + 1)
    tri = np.stack([a, b, a + 1, a + 1, b, b + 1], 1).reshape(-1, 3)
    return tri.astype(np.int32)


def finish_mesh(pos, uv, tri, mat):
    """per-vertex smooth normals and tangents from the triangles"""
    pos = pos.astype(np.float64)
    fn = np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]])
    nrm = np.zeros_like(pos)
    for k in range(3):
        np.add.at(nrm, tri[:, k], fn)
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = np.where(ln > 1e-20, nrm / np.maximum(ln, 1e-20), np.array([[0.0, 1.0, 0.0]]))
    ref = np.where(np.abs(nrm[:, 1:2]) < 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    tan = np.cross(ref, nrm)
    tan /= np.maximum(np.linalg.norm(tan, axis=1, keepdims=True), 1e-20)
    n = len(pos)
    p4 = np.ones((n, 4), np.float32); p4[:, :3] = pos
    n4 = np.zeros((n, 4), np.float32); n4[:, :3] = nrm
    t4 = np.ones((n, 4), np.float32); t4[:, :3] = tan
    return dict(pos=p4, norm=n4, tan=t4, uv=uv.astype(np.float32), idx=tri.astype(np.int32).ravel(), mat=np.asarray(mat, np.int32))


def column(sides, rings, height=8.0, radius=0.45, flutes=16):
    th = np.linspace(0, 2 * np.pi, sides + 1)
    y = np.linspace(0, height, rings + 1)
    T, Y = np.meshgrid(th, y, indexing="ij")
    entasis = 1.0 - 0.12 * (Y / height) ** 2
    R = radius * (1.0 + 0.05 * np.cos(flutes * T)) * entasis
    pos = np.stack([R * np.cos(T), Y, R * np.sin(T)], -1).reshape(-1, 3)
    uv = np.stack([T / (2 * np.pi) * 4.0, Y / height * 4.0], -1).reshape(-1, 2)
    tri = grid_indices(sides, rings)[:, ::-1].copy()      # outward facing
    ring_of_tri = (np.arange(len(tri)) // 2) % rings
    mat = np.where((ring_of_tri < 2) | (ring_of_tri >= rings - 2), 1, 0)   # base / capital band in a second material
    return finish_mesh(pos, uv, tri, mat)


def arch(seg_u, seg_v, span=3.2, tube=0.22):
    u = np.linspace(0, np.pi, seg_u + 1)
    v = np.linspace(0, 2 * np.pi, seg_v + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    R = span * 0.5
    cx, cy = R * np.cos(U), R * np.sin(U)
    pos = np.stack([cx + tube * np.cos(V) * np.cos(U), cy + tube * np.cos(V) * np.sin(U), tube * np.sin(V)], -1).reshape(-1, 3)
    uv = np.stack([U / np.pi * 3.0, V / (2 * np.pi)], -1).reshape(-1, 2)
    tri = grid_indices(seg_u, seg_v)
    return finish_mesh(pos, uv, tri, np.full(len(tri), 2))


def pot(seg_t, seg_p):
    t = np.linspace(0, 1, seg_p + 1)
    prof_r = 0.18 + 0.22 * np.sin(np.pi * np.clip(t * 1.15, 0, 1)) ** 1.5
    prof_y = 0.7 * t
    th = np.linspace(0, 2 * np.pi, seg_t + 1)
    T, P = np.meshgrid(th, np.arange(seg_p + 1), indexing="ij")
    pos = np.stack([prof_r[P] * np.cos(T), prof_y[P], prof_r[P] * np.sin(T)], -1).reshape(-1, 3)
    uv = np.stack([T / (2 * np.pi) * 2, t[P]], -1).reshape(-1, 2)
    tri = grid_indices(seg_t, seg_p)[:, ::-1].copy()
    return finish_mesh(pos, uv, tri, np.full(len(tri), 3))


def floor(nx, nz, lx=40.0, lz=20.0, rng=None):
    x = np.linspace(-lx / 2, lx / 2, nx + 1)
    z = np.linspace(-lz / 2, lz / 2, nz + 1)
    X, Z = np.meshgrid(x, z, indexing="ij")
    Y = 0.03 * np.sin(1.7 * X) * np.cos(2.3 * Z) + 0.01 * np.sin(9.0 * X + 4.0 * Z)
    pos = np.stack([X, Y, Z], -1).reshape(-1, 3)
    uv = np.stack([X / 2.0, Z / 2.0], -1).reshape(-1, 2)
    tri = grid_indices(nx, nz)[:, ::-1].copy()            # +y up
    cell = ((np.arange(len(tri)) // 2) // nz // max(nx // 20, 1) + (np.arange(len(tri)) // 2) % nz // max(nz // 10, 1)) % 2
    return finish_mesh(pos, uv, tri, np.where(cell == 0, 4, 5))


def curtain(nu, nv, width=4.0, height=6.0):
    u = np.linspace(0, 1, nu + 1)
    v = np.linspace(0, 1, nv + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    pos = np.stack([width * (U - 0.5), height * V, 0.25 * np.sin(14.0 * U) * (0.3 + 0.7 * (1 - V))], -1).reshape(-1, 3)
    uv = np.stack([U * 2, V * 2], -1).reshape(-1, 2)
    return finish_mesh(pos, uv, grid_indices(nu, nv), np.full(2 * nu * nv, 6))


def room(lx=40.0, ly=10.0, lz=20.0, open_roof=False):
    hx, hz = lx / 2, lz / 2
    quads = [  # inward-facing walls and ceiling (the floor is its own mesh); open_roof leaves the ceiling out (sky variant)
        ([-hx, 0, -hz], [-hx, 0, hz], [-hx, ly, hz], [-hx, ly, -hz]),
        ([hx, 0, hz], [hx, 0, -hz], [hx, ly, -hz], [hx, ly, hz]),
        ([-hx, 0, -hz], [-hx, ly, -hz], [hx, ly, -hz], [hx, 0, -hz]),
        ([hx, 0, hz], [hx, ly, hz], [-hx, ly, hz], [-hx, 0, hz]),
        ([-hx, ly, -hz], [-hx, ly, hz], [hx, ly, hz], [hx, ly, -hz]),
    ]
    if open_roof:
        quads = quads[:4]
    pos, uv, tri = [], [], []
    for q in quads:
        b = len(pos)
        pos += q
        uv += [[0, 0], [4, 0], [4, 4], [0, 4]]
        tri += [[b, b + 1, b + 2], [b, b + 2, b + 3]]
    m = finish_mesh(np.array(pos, float), np.array(uv, float), np.array(tri, np.int32), [7, 7, 7, 7, 8, 8, 8, 8, 9, 9][:len(tri)])
    return m


def light_quad(hl, hw):
    pos = np.array([[-hl, 0, -hw], [-hl, 0, hw], [hl, 0, hw], [hl, 0, -hw]], float)
    m = finish_mesh(pos, np.array([[0, 0], [0, 1], [1, 1], [1, 0]], float), np.array([[0, 1, 2], [2, 3, 0]], np.int32), [10, 10])
    m["norm"][:, :3] = [0, -1, 0]
    return m


def globe(radius, seg_t=24, seg_p=12, mat=12):
    """uv sphere around the origin with outward normals: the visible surface of a sphere light"""
    th = np.linspace(0, 2 * np.pi, seg_t + 1)
    ph = np.linspace(0, np.pi, seg_p + 1)
    T, P = np.meshgrid(th, ph, indexing="ij")
    pos = radius * np.stack([np.sin(P) * np.cos(T), np.cos(P), np.sin(P) * np.sin(T)], -1).reshape(-1, 3)
    uv = np.stack([T / (2 * np.pi), P / np.pi], -1).reshape(-1, 2)
    tri = grid_indices(seg_t, seg_p)
    area = np.linalg.norm(np.cross(pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]]), axis=1)
    tri = tri[area > 1e-12]                                  # the collapsed triangles at the poles
    m = finish_mesh(pos, uv, tri, np.full(len(tri), mat))
    m["norm"][:, :3] = pos / radius
    if np.dot(np.cross(pos[tri[0, 1]] - pos[tri[0, 0]], pos[tri[0, 2]] - pos[tri[0, 0]]), pos[tri[0]].mean(axis=0)) < 0:
        m["idx"] = tri[:, ::-1].astype(np.int32).ravel()     # counter-clockwise seen from outside
    return m


def tube(radius, height, angle_deg, seg_p=32, seg_z=8, mat=12):
    """open cylinder around the local z axis, the visible surface of a cylinder light: the point (z, phi) of CylinderLightSamplePos (clight.h:785-798)
    carries the texture coordinate ((z - zMin) / height, phi / phiMax), so that a hit looks up the same texel and table cell a sample of that point does"""
    ph = np.linspace(0, np.radians(angle_deg), seg_p + 1)
    zz = np.linspace(-0.5 * height, 0.5 * height, seg_z + 1)
    P, Z = np.meshgrid(ph, zz, indexing="ij")
    pos = np.stack([radius * np.cos(P), radius * np.sin(P), Z], -1).reshape(-1, 3)
    uv = np.stack([(Z + 0.5 * height) / height, P / np.radians(angle_deg)], -1).reshape(-1, 2)
    tri = grid_indices(seg_p, seg_z)
    m = finish_mesh(pos, uv, tri, np.full(len(tri), mat))
    nrm = pos.copy(); nrm[:, 2] = 0; nrm /= radius
    m["norm"][:, :3] = nrm
    c = pos[tri[0]].mean(axis=0); c[2] = 0
    if np.dot(np.cross(pos[tri[0, 1]] - pos[tri[0, 0]], pos[tri[0, 2]] - pos[tri[0, 0]]), c) < 0:
        m["idx"] = tri[:, ::-1].astype(np.int32).ravel()     # counter-clockwise seen from outside
    return m


def plant(cards=6, height=1.6, width=0.9):
    """crossed vertical cards around the y axis, uv = the whole mask on every card, material 11"""
    pos, uv, tri = [], [], []
    for k in range(cards):
        a = np.pi * k / cards
        dx, dz = 0.5 * width * np.cos(a), 0.5 * width * np.sin(a)
        b = len(pos)
        pos += [[-dx, 0, -dz], [dx, 0, dz], [dx, height, dz], [-dx, height, -dz]]
        uv += [[0.02, 0.02], [0.98, 0.02], [0.98, 0.98], [0.02, 0.98]]
        tri += [[b, b + 1, b + 2], [b, b + 2, b + 3]]
    return finish_mesh(np.array(pos, float), np.array(uv, float), np.array(tri, np.int32), np.full(len(tri), 11))


def leaf_mask(n=128):
    """RGBA8: opaque green blobs (alpha 255) on a transparent ground (alpha 0); the material reads alpha as the opacity (input_alpha="alpha")"""
    rng = np.random.default_rng(SEED + 7)
    y, x = np.meshgrid((np.arange(n) + 0.5) / n, (np.arange(n) + 0.5) / n, indexing="ij")
    a = np.zeros((n, n))
    for _ in range(40):
        cx, cy, r = rng.uniform(0.1, 0.9), rng.uniform(0.05, 0.95), rng.uniform(0.04, 0.11)
        a = np.maximum(a, ((x - cx) ** 2 / r ** 2 + (y - cy) ** 2 / (1.6 * r) ** 2 < 1.0).astype(float))
    a = np.maximum(a, (np.abs(x - 0.5) < 0.03).astype(float))      # the stem
    img = np.zeros((n, n, 4), np.uint8)
    img[..., 0] = 40; img[..., 1] = 150; img[..., 2] = 60
    img[..., 3] = (a * 255).astype(np.uint8)
    return img


def tile_normal_map(n=256, periods=2, amp=0.04):
    """tangent-space normal map of smooth round bumps: height = amp sin(2 pi k u) sin(2 pi k v), normal = (-dh/du, -dh/dv, 1) normalised.
    Low frequency on purpose: a normal map multiplies a path's sensitivity to its inputs by |dn/duv| per bounce, and the parity fixtures
    compare implementations whose floats differ in the last bit."""
    u = (np.arange(n) + 0.5) / n
    k = 2.0 * np.pi * periods
    dhdu = amp * k * np.cos(k * u)[None, :] * np.sin(k * u)[:, None]
    dhdv = amp * k * np.sin(k * u)[None, :] * np.cos(k * u)[:, None]
    nrm = np.stack([-dhdu, -dhdv, np.ones_like(dhdu)], axis=-1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    img = np.empty((n, n, 4), np.uint8)
    img[..., :2] = np.clip((nrm[..., :2] * 0.5 + 0.5) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    img[..., 2] = np.clip(nrm[..., 2] * 255.0 + 0.5, 0, 255).astype(np.uint8)    # the fetch reads z as it is, x and y as 2c - 1 (cmaterial.h:2216)
    img[..., 3] = 255
    return img


def tile_height_map(n=256, periods=3):
    """grey height map of smooth round bumps with a flat rim: what <displacement type="height_bump"> hands to IHWLayer::NormalMapFromDisplacement"""
    u = (np.arange(n) + 0.5) / n
    k = 2.0 * np.pi * periods
    hgt = 0.5 + 0.45 * np.sin(k * u)[None, :] * np.sin(k * u)[:, None]
    img = np.empty((n, n, 4), np.uint8)
    img[..., :3] = np.clip(hgt * 255.0 + 0.5, 0, 255).astype(np.uint8)[..., None]
    img[..., 3] = 255
    return img


def write_vsgf(path, m):
    vn, tn = len(m["pos"]), len(m["idx"]) // 3
    blobs = [m["pos"].tobytes(), m["norm"].tobytes(), m["tan"].tobytes(), m["uv"].tobytes(), m["idx"].tobytes(), m["mat"].tobytes()]
    total = 24 + sum(len(b) for b in blobs)
    with open(path, "wb") as f:
        f.write(struct.pack("<QIIII", total, vn, tn * 3, 0, 1))
        for b in blobs:
            f.write(b)
    offs, o = [], 24
    for b in blobs:
        offs.append((o, len(b)))
        o += len(b)
    return vn, tn, total, offs


def checker(n, c0, c1, cells=8):
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    m = ((i * cells // n) + (j * cells // n)) % 2
    img = np.where(m[..., None] == 0, np.array(c0, np.uint8), np.array(c1, np.uint8)).astype(np.uint8)
    noise = np.random.default_rng(SEED + n).integers(0, 24, (n, n, 1), dtype=np.uint8)
    img = np.clip(img.astype(np.int32) - noise, 0, 255).astype(np.uint8)
    return np.concatenate([img, np.full((n, n, 1), 255, np.uint8)], -1)


def mat4(scale=1.0, yaw=0.0, t=(0, 0, 0), rot_x=0.0):
    cy, sy = np.cos(yaw), np.sin(yaw)
    cx, sx = np.cos(rot_x), np.sin(rot_x)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    m = np.eye(4)
    m[:3, :3] = ry @ rx * scale
    m[:3, 3] = t
    return " ".join("%.7g" % v for v in m.ravel()) + " "


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--scale", type=float, default=1.0, help="tessellation scale (1.0 ~ 249 k triangles)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sky", action="store_true", help="open roof + constant sky light (1,1,1) x 0.5 next to the roof light (SURVEY 8d, S2)")
    ap.add_argument("--delta-lights", action="store_true", help="open roof, no sky: adds a point, a spot and a directional (soft sun) light to the roof light")
    ap.add_argument("--sky-hdr", action="store_true", help="like --sky-tex, but the environment map is a float texture (.image4f, sun at 25x the sky) sampled through a sampler matrix "
                    "that turns it by a quarter of the horizon: the table of LuminanceFromFloat4Image and the inverse-matrix path of SkyLightSampleRev")
    ap.add_argument("--sky-tex", action="store_true", help="like --sky, but the sky is a 512x256 lat-long texture (horizon gradient + sun) x 0.8")
    ap.add_argument("--perez", action="store_true", help="like --sky, but the sky uses the Perez all-weather model (turbidity 2.5) with a directional sun (light id 3) "
                    "that also lights the hall through the open roof")
    ap.add_argument("--tube-lights", action="store_true", help="closed hall; adds a cylinder light textured with the 256^2 checker, an untextured half cylinder and a mesh light "
                    "textured with the 128^2 checker (clight.h:753-830, 957-1062), each with its visible surface")
    ap.add_argument("--ies", action="store_true", help="closed hall; adds a point light and two rectangular area lights (one evaluated from its centre, point_area) whose "
                    "distribution is a photometric web: two generated IESNA LM-63 files, one covering the whole sphere without symmetry, one a quadrant of the lower hemisphere "
                    "(clight.h:405-426, 465-495; IESRender.cpp)")
    ap.add_argument("--portal", action="store_true", help="like --sky-tex plus a soft sun; a 16 x 8 sky portal (area light with <sky_portal>, material sky_portal_mtl) lies in the "
                    "open roof and stands in for the sky light, which is then never sampled (clight.h:590-629, 1670-1695)")
    ap.add_argument("--glass", action="store_true", help="closed hall; pots = clear glass + Fresnel mirror, arches = rough (GGX) glass, column bands = "
                    "reflection + glass + diffuse, curtains = textured glossy thin glass over diffuse")
    ap.add_argument("--cutouts", action="store_true", help="adds 60 instanced plants made of crossed cards whose material has an <opacity> leaf mask (alpha-tested traversal, "
                    "BVH4InstTraverseAlpha) and hangs a perforated screen (same mask) in front of the camera")
    ap.add_argument("--two-trees", action="store_true", help="with --cutouts: the render settings ask the front end to put the instances of alpha-tested meshes into a second BVH "
                    "tree (<split_alpha_tree>), the way Embree hands the reference several trees")
    ap.add_argument("--translucent", action="store_true", help="material 6 (curtains) becomes diffuse + translucency (a blend of a lambert and a translucent node), "
                    "material 7 translucency alone (diffuse transmission, cmaterial.h:1852-1909); the reflectivity lobes of materials 1 and 8 become "
                    "Blinn/Torrance-Sparrow (cmaterial.h:1020-1168)")
    ap.add_argument("--normal-maps", action="store_true", help="materials 0, 1, 4, 5, 8 and 9 (floor, walls, columns: lambert, textured lambert and lambert + glossy blends) get "
                    "<displacement type='normal_bump'> with a generated 256x256 normal map (smooth round bumps), y inverted on two of them")
    ap.add_argument("--aniso", action="store_true", help="the reflectivity lobe of material 1 becomes Beckmann (anisotropy 0.7, rotated by 0.15 turns, glossiness "
                    "from texture 2), that of material 8 TRGGX (anisotropy 0.5, flipped axes), material 9 a Fresnel blend of an isotropic Beckmann lobe "
                    "over diffuse (cmaterial.h:1558-1846, cmatpbrt.h:105-540)")
    ap.add_argument("--height-bump", action="store_true", help="materials 0, 4 (floor) and 9 (a wall) get <displacement type='height_bump'> over a generated 256x256 "
                    "height map (amount 0.8; the wall's copy smoothed, smooth_lvl 0.3): the layer bakes the normal maps (IHWLayer::NormalMapFromDisplacement)")
    ap.add_argument("--proctex", nargs="?", const="full", default=None, choices=("full", "flat", "plain"), help="closed hall; declares four procedural textures (<texture type='proc'> + data/proctex_*.c in the dialect HydraAPI "
                    "generates: a 3-D checker over the local position, a view falloff over hr_viewVectorHack, a tri-planar blend of the two stored checkers through "
                    "texture2D, a rippled normal through StoreNormal) and binds them: floor = tri-planar diffuse, columns = checker diffuse + falloff reflection "
                    "(two textures in one material), pots = falloff diffuse, a wall = the procedural normal map.  'flat': the checker and the falloff with both colours "
                    "equal (exactly representable in half precision) on materials 0, 1, 3, 4 over white; 'plain': no procedural textures, those colours as the materials' own -- "
                    "the two must render the same image bit for bit")
    ap.add_argument("--back", nargs="?", const="camera", default=None, choices=("camera", "spherical"), help="like --sky-tex, and the sky light carries a <back> node: the 256^2 "
                    "checker (texture 1) x (0.9, 1.0, 0.8) is what the camera sees where a ray leaves the hall -- projected by pixel, or as a sphere map -- while the light of the "
                    "sky stays the environment texture (the OpenCL layer's environmentColorExtended, cbidir.h:593-629)")
    ap.add_argument("--blend", action="store_true", help="closed hall; materials 1, 3 and 5 become hydra_blend materials -- two materials of the library under a mask: "
                    "1 = its former self (now id 12) over a red lambert (13) through the 128^2 checker, 3 = a glossy lobe (14) over a lambert (15) by Fresnel (IOR 1.8, "
                    "luminance extrusion), 5 = blend 1 over material 13 under a constant 0.5 grey texture-less mask: a blend of a blend, and children with higher ids than the blend")
    ap.add_argument("--catcher", action="store_true", help="materials 0 and 4 (the floor) become shadow catchers (<material type='shadow_catcher'>): with --back the camera sees "
                    "the back-plate through them, darkened where the roof light and the sky are occluded (the OpenCL layer's shadow matte)")
    ap.add_argument("--ggx", action="store_true", help="every reflectivity layer is a GGX lobe instead of Phong; material 9 (a wall) becomes Fresnel GGX over diffuse")
    args = ap.parse_args()
    args.sky_tex = args.sky_tex or args.sky_hdr or args.portal or (args.back is not None)
    args.sky = args.sky or args.sky_tex or args.perez
    refl = "ggx" if args.ggx else "torranse_sparrow" if args.translucent else "phong"   # "torranse_sparrow" (sic) = Blinn in a Torrance-Sparrow model
    s = np.sqrt(args.scale)
    rng = np.random.default_rng(SEED)
    out = args.out
    os.makedirs(os.path.join(out, "data"), exist_ok=True)

    def r(n, lo=2):
        return max(lo, int(round(n * s)))
    meshes = [("column", column(r(64, 8), r(32, 4))), ("arch", arch(r(32, 4), r(32, 4))), ("pot", pot(r(20, 4), r(10, 2))),
              ("floor", floor(r(192, 4), r(96, 2))), ("curtain", curtain(r(40, 2), r(64, 2))), ("room", room(open_roof=args.sky or args.delta_lights)), ("light", light_quad(2.0, 0.5))]
    if args.cutouts:
        meshes.append(("plant", plant()))
    globe_mesh = lamp_mesh = None
    if args.delta_lights:      # ... and a sphere light with its visible globe (clight.h:1287-1332)
        globe_mesh = len(meshes)
        meshes.append(("globe", globe(0.35)))
        lamp_mesh = len(meshes)                 # ... and a mesh light: a coarse, squashed globe (its own triangles are what is sampled, clight.h:966-1062)
        lamp = globe(0.3, 8, 4, mat=13)
        lamp["pos"][:, 1] *= 0.6
        meshes.append(("lamp", lamp))

    tube_mesh = half_tube_mesh = tlamp_mesh = portal_mesh = None
    if args.tube_lights:
        tube_mesh = len(meshes)
        meshes.append(("tube", tube(0.15, 6.0, 360.0)))
        half_tube_mesh = len(meshes)
        meshes.append(("half_tube", tube(0.3, 3.0, 180.0, 16, 4, mat=13)))
        tlamp_mesh = len(meshes)
        tl = globe(0.3, 8, 4, mat=14)
        tl["pos"][:, 1] *= 0.6
        meshes.append(("tlamp", tl))
    ies_quad1 = ies_quad2 = None
    if args.ies:
        ies_quad1 = len(meshes)
        q1 = light_quad(0.6, 0.3)
        q1["mat"][:] = 12
        meshes.append(("ies_panel", q1))
        ies_quad2 = len(meshes)
        q2 = light_quad(0.4, 0.4)
        q2["mat"][:] = 13
        meshes.append(("ies_panel2", q2))

        def write_ies(path, vert, horz, fn):
            with open(path, "w") as f:
                f.write("IESNA:LM-63-1995\n[TEST] generated by tools/make_atrium.py\n[MANUFAC] none\nTILT=NONE\n")
                f.write("1 1000 1 %d %d 1 2 0 0 0\n1 1 50\n" % (len(vert), len(horz)))
                f.write(" ".join("%g" % v for v in vert) + "\n" + " ".join("%g" % v for v in horz) + "\n")
                for hz in horz:
                    f.write(" ".join("%.4f" % fn(np.radians(v), np.radians(hz)) for v in vert) + "\n")
        # the whole sphere, no lateral symmetry: a downward lobe that leans towards phi = 60 degrees, a weaker upward one
        write_ies(os.path.join(out, "data", "web_00001.ies"), np.arange(0, 181, 10), np.arange(0, 361, 45),
                  lambda t, p: 800.0 * max(np.cos(t), 0.0) ** 2 * (1.0 + 0.6 * np.cos(p - 1.05)) + 150.0 * max(-np.cos(t), 0.0) + 20.0)
        # a quadrant (0..90 degrees both ways) of the lower hemisphere: the file leaves the mirroring to the reader
        write_ies(os.path.join(out, "data", "web_00002.ies"), np.arange(0, 91, 7.5), np.arange(0, 91, 22.5),
                  lambda t, p: 600.0 * np.cos(t) ** 4 * (1.0 + 0.5 * np.cos(2 * p)) + 10.0)
    if args.portal:
        portal_mesh = len(meshes)
        pm = light_quad(8.0, 4.0)
        pm["mat"][:] = 12
        meshes.append(("portal", pm))

    # textures: id 0 = white dummy (as in the reference fixtures), 1..2 = checkers
    texs = [(2, np.full((2, 2, 4), 255, np.uint8)), (256, checker(256, (200, 170, 120), (120, 90, 60))), (128, checker(128, (90, 110, 160), (210, 210, 220), 4))]
    if args.sky_tex:   # id 3: environment map, y = 0 is straight down (texCoord2DToSphereMap: theta = v * pi measured from -y)
        hh, ww = 256, 512
        v, u = np.meshgrid((np.arange(hh) + 0.5) / hh, (np.arange(ww) + 0.5) / ww, indexing="ij")
        up = np.clip((v - 0.5) * 2.0, 0.0, 1.0)                       # 0 at the horizon, 1 at the zenith
        env = np.zeros((hh, ww, 4), np.float64)
        env[..., 0] = 0.35 + 0.25 * (1 - up); env[..., 1] = 0.45 + 0.25 * (1 - up); env[..., 2] = 0.75 - 0.15 * (1 - up)
        env[v < 0.5] = [0.12, 0.11, 0.10, 0]                          # ground half
        sun = np.exp(-(((u - 0.62) * 2.0) ** 2 + (v - 0.80) ** 2) / (2 * 0.02 ** 2))
        if args.sky_hdr:
            env[..., :3] = env[..., :3] + sun[..., None] * np.array([25.0, 23.0, 18.0])
            env[..., 3] = 1.0
            texs.append((None, env.astype(np.float32)))
        else:
            env[..., :3] = np.clip(env[..., :3] + sun[..., None] * np.array([1.0, 0.95, 0.8]), 0, 1)
            env[..., 3] = 1.0
            texs.append((None, (env * 255.0 + 0.5).astype(np.uint8)))
    mask_tex = None
    if args.cutouts:
        mask_tex = len(texs)
        texs.append((None, leaf_mask()))
    nmap_tex = None
    if args.normal_maps:
        nmap_tex = len(texs)
        texs.append((None, tile_normal_map()))
    hmap_tex = None
    if args.height_bump:
        hmap_tex = len(texs)
        texs.append((None, tile_height_map()))
    xml = ['<?xml version="1.0"?>', '<textures_lib total_chunks="%d">' % (len(texs) + len(meshes))]
    chunk = 0
    for tid, (n, img) in enumerate(texs):
        name = "data/chunk_%05d.%s" % (chunk, "image4f" if img.dtype == np.float32 else "image4ub")
        th, tw = img.shape[0], img.shape[1]
        with open(os.path.join(out, name), "wb") as f:
            f.write(struct.pack("<II", tw, th))
            f.write(img.tobytes())
        xml.append('  <texture id="%d" name="tex%d" loc="%s" offset="8" bytesize="%d" width="%d" height="%d" dl="0" />' % (tid, tid, name, tw * th * img.dtype.itemsize * 4, tw, th))
        chunk += 1
    proc_ids = {}
    if args.proctex in ("full", "flat"):   # procedural textures: functions in the scene library, one generated call each (the layout of HydraAPI's HRTextureNodeProc export)
        procs = [
            ("checker3d", """float prtex%(n)d_cell(float x, float s)
{
  return floor(x*s);
}

float4 prtex%(n)d_main(const SurfaceInfo* sHit, float3 colorA, float3 colorB, float cells, _PROCTEXTAILTAG_)
{
  const float3 p = readAttr_LocalPos(sHit);
  const float k = prtex%(n)d_cell(p.x, cells) + prtex%(n)d_cell(p.y, cells) + prtex%(n)d_cell(p.z, cells);
  const float odd = fabs(fmod(k, 2.0f));
  const float3 c = (odd > 0.5f) ? colorA : colorB;
  return make_float4(c.x, c.y, c.z, 0.0f);
}
""", [("float3", "colorA", 3), ("float3", "colorB", 3), ("float", "cells", 1)]),
            ("falloff", """float3 prtex%(n)d_mix(float3 x, float3 y, float a)
{
  return x*(1.0f - a) + y*a;
}

float4 prtex%(n)d_main(const SurfaceInfo* sHit, float3 color1, float3 color2, _PROCTEXTAILTAG_)
{
  const float3 norm   = readAttr_ShadeNorm(sHit);
  const float3 rayDir = hr_viewVectorHack;
  const float cosAlpha = fabs(dot(norm, rayDir));
  return to_float4(prtex%(n)d_mix(color1, color2, cosAlpha), 0.0f);
}
""", [("float3", "color1", 3), ("float3", "color2", 3)]),
            ("triplanar", """float3 prtex%(n)d_weights(float3 n, float sharp)
{
  float3 w = make_float3(pow(fabs(n.x), sharp), pow(fabs(n.y), sharp), pow(fabs(n.z), sharp));
  w = max(w, 0.00001f);
  const float b = w.x + w.y + w.z;
  return w / b;
}

float4 prtex%(n)d_main(const SurfaceInfo* sHit, sampler2D texSide, sampler2D texTop, float sharp, float mapScale, _PROCTEXTAILTAG_)
{
  const float3 norm = readAttr_ShadeNorm(sHit);
  const float3 pos  = readAttr_WorldPos(sHit);
  const float3 w    = prtex%(n)d_weights(norm, sharp);
  const float2 x_uv = make_float2(pos.z / mapScale, pos.y / mapScale);
  const float2 y_uv = make_float2(pos.x / mapScale, pos.z / mapScale);
  const float2 z_uv = make_float2(pos.x / mapScale, pos.y / mapScale);
  const float4 cx = texture2D(texSide, x_uv, 0);
  const float4 cy = texture2D(texTop,  y_uv, TEX_CLAMP_U);
  const float4 cz = texture2D(texSide, z_uv, TEX_POINT_SAM);
  return cx * w.x + cy * w.y + cz * w.z;
}
""", [("sampler2D", "texSide", 1), ("sampler2D", "texTop", 1), ("float", "sharp", 1), ("float", "mapScale", 1)]),
            ("ripples", """float4 prtex%(n)d_main(const SurfaceInfo* sHit, float freq, float amp, _PROCTEXTAILTAG_)
{
  const float2 tc = readAttr_TexCoord0(sHit);
  const float dx = amp * cos(freq * tc.x) ;
  const float dy = amp * sin(freq * tc.y + 0.5f * tc.x);
  return StoreNormal(make_float3(-dx, -dy, 1.0f), NORMAL_IN_TANGENT_SPACE);
}
""", [("float", "freq", 1), ("float", "amp", 1)]),
        ]
        for k, (pname, code, pargs) in enumerate(procs):
            tid = len(texs) + k
            proc_ids[pname] = tid
            fname = "data/proctex_%05d.c" % tid
            with open(os.path.join(out, fname), "w") as f:
                f.write(code % {"n": tid})
            call, arg_xml, wo = [], [], 0
            for ai, (atype, aname, words) in enumerate(pargs):
                arg_xml.append('        <arg id="%d" type="%s" name="%s" size="1" wsize="%d" woffset="%d" />' % (ai, atype, aname, words, wo))
                if atype == "float3":
                    call.append("make_float3(stack[%d], stack[%d], stack[%d])" % (wo, wo + 1, wo + 2))
                elif atype == "sampler2D":
                    call.append("as_int(stack[%d])" % wo)
                else:
                    call.append("stack[%d]" % wo)
                wo += words
            xml.append('  <texture id="%d" name="%s" type="proc">' % (tid, pname))
            xml.append('    <code file="%s.c" main="main" loc="%s">' % (pname, fname))
            xml.append('      <generated>')
            xml.extend(arg_xml)
            xml.append('        <return type="float4" />')
            xml.append('        <call>prtex%d_main(sHit, %s, _PROCTEXTAILTAG_)</call>' % (tid, ", ".join(call)))
            xml.append('      </generated>')
            xml.append('    </code>')
            xml.append('  </texture>')
    xml.append("</textures_lib>")

    cols = rng.uniform(0.2, 0.8, (10, 3))
    xml.append("<materials_lib>")
    for mid in range(10):
        c = "%.4f %.4f %.4f" % tuple(cols[mid])
        if args.glass and mid in (1, 2, 3, 6):
            body = {
                3: '<reflectivity brdf_type="phong"><color val="0.9 0.9 0.9" /><glossiness val="1" /><fresnel val="1" /><fresnel_ior val="1.5" /></reflectivity>'
                   '<transparency><color val="0.9 0.97 0.92" /><glossiness val="1" /><thin_walled val="0" /><fog_color val="1 1 1" /><fog_multiplier val="0" /><ior val="1.5" /></transparency>',
                2: '<transparency><color val="0.85 0.9 0.95" /><glossiness val="0.7" /><thin_walled val="0" /><ior val="1.33" /></transparency>',
                1: '<diffuse brdf_type="lambert"><color val="%s" /></diffuse>'
                   '<reflectivity brdf_type="%s"><color val="0.3 0.3 0.3" /><glossiness val="0.8" /><fresnel val="1" /><fresnel_ior val="1.6" /></reflectivity>'
                   '<transparency><color val="0.5 0.5 0.5" /><glossiness val="1" /><thin_walled val="0" /><ior val="1.6" /></transparency>' % (c, refl),
                6: '<diffuse brdf_type="lambert"><color val="%s" /></diffuse>'
                   '<transparency><color val="0.6 0.6 0.6"><texture id="2" type="texref" /></color><glossiness val="0.85" /><thin_walled val="1" /><ior val="1.5" /></transparency>' % c,
            }[mid]
            xml.append('  <material id="%d" name="m%d" type="hydra_material">%s</material>' % (mid, mid, body))
        elif mid in (1, 8):      # lambert + phong (or GGX) blend
            gloss = 0.5 if mid == 1 else 0.85
            # with --ggx material 8's lobe asks for the multi-scattering energy compensation (PlainMaterialConverter.cpp:1073-1077, 1145-1146): the
            # layer's energy tables are read; material 1's lobe stays without it
            ms = '<multiscatter val="1" />' if (args.ggx and mid == 8) else ''
            xml.append('  <material id="%d" name="m%d" type="hydra_material"><diffuse brdf_type="lambert"><color val="%s" /></diffuse>'
                       '<reflectivity brdf_type="%s"><color val="0.35 0.33 0.3" /><glossiness val="%.2f" />%s</reflectivity></material>' % (mid, mid, c, refl, gloss, ms))
        elif args.ggx and mid == 9:
            xml.append('  <material id="%d" name="m%d" type="hydra_material"><diffuse brdf_type="lambert"><color val="%s" /></diffuse>'
                       '<reflectivity brdf_type="ggx"><color val="0.8 0.8 0.8" /><glossiness val="0.7" /><fresnel val="1" /><fresnel_ior val="2.5" /></reflectivity></material>' % (mid, mid, c))
        elif mid in (0, 4, 6):  # textured lambert
            tex = {0: 1, 4: 1, 6: 2}[mid]
            xml.append('  <material id="%d" name="m%d" type="hydra_material"><diffuse brdf_type="lambert"><color val="%s" />'
                       '<texture id="%d" type="texref" /></diffuse></material>' % (mid, mid, c, tex))
        elif args.delta_lights and mid in (2, 7):   # rough plaster: Oren-Nayar
            xml.append('  <material id="%d" name="m%d" type="hydra_material"><diffuse brdf_type="orennayar"><color val="%s" /><roughness val="%.2f" /></diffuse></material>' % (mid, mid, c, 0.4 if mid == 2 else 0.9))
        else:
            xml.append('  <material id="%d" name="m%d" type="hydra_material"><diffuse brdf_type="lambert"><color val="%s" /></diffuse></material>' % (mid, mid, c))
    if args.proctex:
        def texref_proc(name, vals):   # a bound procedural texture: <arg> values in the order of the declaration
            a = "".join('<arg id="%d" type="%s" size="1" val="%s" />' % (i, t, v) for i, (t, v) in enumerate(vals))
            return '<texture id="%d" type="texref_proc">%s</texture>' % (proc_ids[name], a)
        checker_args = [("float3", "0.85 0.8 0.7"), ("float3", "0.25 0.2 0.3"), ("float", "2.5")]
        falloff_args = [("float3", "0.9 0.2 0.1"), ("float3", "0.1 0.3 0.9")]
        new_mats = {} if args.proctex != "full" else {
            0: '<diffuse brdf_type="lambert"><color val="0.9 0.9 0.9" />%s</diffuse>' % texref_proc("triplanar", [("sampler2D", "1"), ("sampler2D", "2"), ("float", "4"), ("float", "3.0")]),
            4: '<diffuse brdf_type="lambert"><color val="0.9 0.9 0.9" />%s</diffuse>' % texref_proc("triplanar", [("sampler2D", "2"), ("sampler2D", "1"), ("float", "2"), ("float", "1.5")]),
            1: '<diffuse brdf_type="lambert"><color val="0.8 0.8 0.8" />%s</diffuse><reflectivity brdf_type="%s"><color val="0.5 0.5 0.5">%s</color><glossiness val="0.6" /></reflectivity>'
               % (texref_proc("checker3d", checker_args), refl, texref_proc("falloff", falloff_args)),
            3: '<diffuse brdf_type="lambert"><color val="1 1 1" />%s</diffuse>' % texref_proc("falloff", [("float3", "0.9 0.9 0.2"), ("float3", "0.2 0.7 0.3")]),
            9: '<diffuse brdf_type="lambert"><color val="%.4f %.4f %.4f" /></diffuse><displacement type="normal_bump"><normal_map><invert x="0" y="0" swap_xy="0" />%s</normal_map></displacement>'
               % (tuple(cols[9]) + (texref_proc("ripples", [("float", "40.0"), ("float", "0.35")]),)),
        }
        flat_a, flat_b = "0.5 0.25 0.75", "0.75 0.5 0.25"
        if args.proctex == "flat":
            fa = [("float3", flat_a), ("float3", flat_a), ("float", "2.5")]
            fb = [("float3", flat_b), ("float3", flat_b)]
            new_mats = {
                0: '<diffuse brdf_type="lambert"><color val="1 1 1" />%s</diffuse>' % texref_proc("checker3d", fa),
                4: '<diffuse brdf_type="lambert"><color val="1 1 1" />%s</diffuse>' % texref_proc("falloff", fb),
                # the reflectivity colour also sets the blend weight on the host (PlainMaterialConverter.cpp:1541-1591): it stays as in 'plain', under a white falloff
                1: '<diffuse brdf_type="lambert"><color val="1 1 1" />%s</diffuse><reflectivity brdf_type="%s"><color val="%s">%s</color><glossiness val="0.6" /></reflectivity>'
                   % (texref_proc("checker3d", fa), refl, flat_b, texref_proc("falloff", [("float3", "1 1 1"), ("float3", "1 1 1")])),
                3: '<diffuse brdf_type="lambert"><color val="1 1 1" />%s</diffuse>' % texref_proc("falloff", fb),
            }
        elif args.proctex == "plain":
            new_mats = {
                0: '<diffuse brdf_type="lambert"><color val="%s" /></diffuse>' % flat_a,
                4: '<diffuse brdf_type="lambert"><color val="%s" /></diffuse>' % flat_b,
                1: '<diffuse brdf_type="lambert"><color val="%s" /></diffuse><reflectivity brdf_type="%s"><color val="%s" /><glossiness val="0.6" /></reflectivity>' % (flat_a, refl, flat_b),
                3: '<diffuse brdf_type="lambert"><color val="%s" /></diffuse>' % flat_b,
            }
        for i, line in enumerate(xml):
            for mid, body in new_mats.items():
                if line.startswith('  <material id="%d" ' % mid):
                    xml[i] = '  <material id="%d" name="m%d" type="hydra_material">%s</material>' % (mid, mid, body)
    if args.blend:
        for i, line in enumerate(xml):
            if line.startswith('  <material id="1" '):
                moved = line.replace('<material id="1" name="m1"', '<material id="12" name="m12"')
                xml[i] = ('  <material id="1" name="m1" type="hydra_blend" node_top="12" node_bottom="13"><blend type="mask_blend"><mask val="1.0">'
                          '<texture id="2" type="texref" matrix="3 0 0 0 0 3 0 0 0 0 1 0 0 0 0 1" /></mask></blend></material>')
            if line.startswith('  <material id="3" '):
                xml[i] = ('  <material id="3" name="m3" type="hydra_blend" node_top="14" node_bottom="15"><blend type="fresnel_blend"><fresnel_ior val="1.8" /></blend>'
                          '<extrusion val="luminance" /></material>')
            if line.startswith('  <material id="5" '):
                xml[i] = '  <material id="5" name="m5" type="hydra_blend" node_top="1" node_bottom="13"><blend type="mask_blend"><mask val="1.0" /></blend></material>'
        blend_children = [moved,
                          '  <material id="13" name="m13" type="hydra_material"><diffuse brdf_type="lambert"><color val="0.7 0.15 0.1" /></diffuse></material>',
                          '  <material id="14" name="m14" type="hydra_material"><reflectivity brdf_type="%s"><color val="0.9 0.9 0.9" /><glossiness val="0.9" /></reflectivity></material>' % refl,
                          '  <material id="15" name="m15" type="hydra_material"><diffuse brdf_type="lambert"><color val="0.2 0.4 0.7" /></diffuse></material>']
    if args.catcher:
        for i, line in enumerate(xml):
            for mid in (0, 4):
                if line.startswith('  <material id="%d" ' % mid):
                    xml[i] = '  <material id="%d" name="m%d" type="shadow_catcher"></material>' % (mid, mid)
    if args.height_bump:
        for i, line in enumerate(xml):
            for mid in (0, 4, 9):
                if line.startswith('  <material id="%d" ' % mid):
                    bump = ('<displacement type="height_bump"><height_map amount="0.8" smooth_lvl="%s"><texture id="%d" type="texref" '
                            'matrix="2 0 0 0 0 2 0 0 0 0 1 0 0 0 0 1" /></height_map></displacement>' % ("0.3" if mid == 9 else "0.0", hmap_tex))
                    xml[i] = line.replace("</material>", bump + "</material>")
    if args.aniso:   # ... and the pots (material 3) become shadow catchers: black pass-through surfaces on the CPU integrator's path
        for i, line in enumerate(xml):
            if line.startswith('  <material id="3" '):
                xml[i] = '  <material id="3" name="m3" type="shadow_catcher"></material>'
        for i, line in enumerate(xml):
            if line.startswith('  <material id="1" '):
                xml[i] = ('  <material id="1" name="m1" type="hydra_material"><diffuse brdf_type="lambert"><color val="%.4f %.4f %.4f" /></diffuse>'
                          '<reflectivity brdf_type="beckmann"><color val="0.35 0.33 0.3" /><glossiness val="0.75"><texture id="2" type="texref" /></glossiness>'
                          '<anisotropy val="0.7" rot="0.15" flip_axis="0" /></reflectivity></material>' % tuple(cols[1]))
            if line.startswith('  <material id="8" '):
                xml[i] = ('  <material id="8" name="m8" type="hydra_material"><diffuse brdf_type="lambert"><color val="%.4f %.4f %.4f" /></diffuse>'
                          '<reflectivity brdf_type="trggx"><color val="0.4 0.4 0.4" /><glossiness val="0.8" />'
                          '<anisotropy val="0.5" rot="0.0" flip_axis="1" /></reflectivity></material>' % tuple(cols[8]))
            if line.startswith('  <material id="9" '):
                xml[i] = ('  <material id="9" name="m9" type="hydra_material"><diffuse brdf_type="lambert"><color val="%.4f %.4f %.4f" /></diffuse>'
                          '<reflectivity brdf_type="beckmann"><color val="0.8 0.8 0.8" /><glossiness val="0.6" /><fresnel val="1" /><fresnel_ior val="2.0" /></reflectivity></material>' % tuple(cols[9]))
    if args.translucent:
        for i, line in enumerate(xml):
            if line.startswith('  <material id="6" '):
                xml[i] = line.replace("</material>", '<translucency><color val="0.6 0.7 0.5" /></translucency></material>')
            if line.startswith('  <material id="7" '):
                xml[i] = '  <material id="7" name="m7" type="hydra_material"><diffuse brdf_type="lambert"><color val="0 0 0" /></diffuse><translucency><color val="0.8 0.8 0.7" /></translucency></material>'
    if args.normal_maps:
        for i, line in enumerate(xml):
            for mid in (0, 1, 4, 5, 8, 9):
                if line.startswith('  <material id="%d" ' % mid):
                    bump = ('<displacement type="normal_bump"><normal_map><invert x="0" y="%d" swap_xy="0" /><texture id="%d" type="texref" '
                            'matrix="1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1" /></normal_map></displacement>' % (1 if mid in (4, 9) else 0, nmap_tex))
                    xml[i] = line.replace("</material>", bump + "</material>")
    xml.append('  <material id="10" name="light_mat" type="hydra_material" light_id="0" visible="1"><emission><color val="60 56 50" /></emission></material>')
    if args.blend:
        xml.extend(blend_children)
    if args.cutouts:   # leaves: textured lambert, the mask's alpha channel is the opacity
        xml.append('  <material id="11" name="leaves" type="hydra_material"><diffuse brdf_type="lambert"><color val="0.9 0.9 0.9" /><texture id="%d" type="texref" /></diffuse>'
                   '<opacity smooth="0"><skip_shadow val="0" /><texture id="%d" type="texref" input_alpha="alpha" input_gamma="1" /></opacity></material>' % (mask_tex, mask_tex))
    if args.delta_lights:
        xml.append('  <material id="12" name="globe_mat" type="hydra_material" light_id="4" visible="1"><emission><color val="25 22 18" /></emission></material>')
        xml.append('  <material id="13" name="lamp_mat" type="hydra_material" light_id="5" visible="1"><emission><color val="18 24 30" /></emission></material>')
    if args.tube_lights:
        xml.append('  <material id="12" name="tube_mat" type="hydra_material" light_id="1" visible="1"><emission><color val="30 30 30" /></emission></material>')
        xml.append('  <material id="13" name="half_tube_mat" type="hydra_material" light_id="2" visible="1"><emission><color val="20 14 8" /></emission></material>')
        xml.append('  <material id="14" name="tlamp_mat" type="hydra_material" light_id="3" visible="1"><emission><color val="18 24 30" /></emission></material>')
    if args.ies:
        xml.append('  <material id="12" name="ies_panel_mat" type="hydra_material" light_id="2" visible="1"><emission><color val="20 20 18" /></emission></material>')
        xml.append('  <material id="13" name="ies_panel2_mat" type="hydra_material" light_id="3" visible="1"><emission><color val="14 16 20" /></emission></material>')
    if args.portal:
        xml.append('  <material id="12" name="portal_mat" type="sky_portal_mtl" light_id="2" visible="1"><emission><color val="1 1 1" /><multiplier val="1.0" /></emission></material>')
    xml.append("</materials_lib>")
    xml.append('<lights_lib>\n  <light id="0" name="roof_light" type="area" shape="rect" distribution="diffuse" visible="1" mat_id="10" mesh_id="6">'
               '<size half_length="2.0" half_width="0.5" /><intensity><color val="1 0.933 0.833" /><multiplier val="60.0" /></intensity></light>'
               + ('\n  <light id="1" name="sky" type="sky" shape="point" distribution="uniform" visible="1"><intensity><color val="1 1 1">'
                  '<texture id="3" type="texref" matrix="1 0 0 -0.25 0 1 0 0 0 0 1 0 0 0 0 1" input_gamma="1" /></color><multiplier val="0.8" /></intensity></light>' if args.sky_hdr else
                  '\n  <light id="1" name="sky" type="sky" shape="point" distribution="uniform" visible="1"><intensity><color val="1 1 1">'
                  '<texture id="3" type="texref" input_gamma="2.2" /></color><multiplier val="0.8" /></intensity></light>' if args.sky_tex else
                  '\n  <light id="1" name="sky" type="sky" shape="point" distribution="uniform" visible="1"><intensity><color val="1 1 1" />'
                  '<multiplier val="1.5" /></intensity><perez turbidity="2.5" sun_id="3" /></light>'
                  '\n  <light id="3" name="sun" type="directional" shape="point" distribution="directional" visible="1"><size inner_radius="30" outer_radius="40" />'
                  '<shadow_softness val="1.0" /><intensity><color val="1 0.92 0.8" /><multiplier val="3.0" /></intensity></light>' if args.perez else
                  '\n  <light id="1" name="sky" type="sky" shape="point" distribution="uniform" visible="1"><intensity><color val="1 1 1" />'
                  '<multiplier val="0.5" /></intensity></light>' if args.sky else
                  '\n  <light id="1" name="tube" type="area" shape="cylinder" distribution="uniform" visible="1" mat_id="12" mesh_id="%d"><size radius="0.15" height="6" angle="360" />'
                  '<intensity><color val="1 1 1"><texture id="1" type="texref" /></color><multiplier val="30.0" /></intensity></light>'
                  '\n  <light id="2" name="half_tube" type="area" shape="cylinder" distribution="uniform" visible="1" mat_id="13" mesh_id="%d"><size radius="0.3" height="3" angle="180" />'
                  '<intensity><color val="1 0.7 0.4" /><multiplier val="20.0" /></intensity></light>'
                  '\n  <light id="3" name="tlamp" type="area" shape="mesh" distribution="uniform" visible="1" mat_id="14" mesh_id="%d">'
                  '<intensity><color val="0.6 0.8 1"><texture id="2" type="texref" /></color><multiplier val="30.0" /></intensity></light>' % (tube_mesh, half_tube_mesh, tlamp_mesh) if args.tube_lights else
                  '\n  <light id="1" name="web_bulb" type="point" shape="point" distribution="ies" visible="1"><ies data="web_00001.ies" loc="data/web_00001.ies" matrix="0.8 0 0.6 0 0 1 0 0 -0.6 0 0.8 0 0 0 0 1" />'
                  '<intensity><color val="1 0.9 0.7" /><multiplier val="120.0" /></intensity></light>'
                  '\n  <light id="2" name="web_panel" type="area" shape="rect" distribution="ies" visible="1" mat_id="12" mesh_id="%d"><size half_length="0.6" half_width="0.3" />'
                  '<ies data="web_00002.ies" loc="data/web_00002.ies" point_area="0" /><intensity><color val="1 1 0.9" /><multiplier val="60.0" /></intensity></light>'
                  '\n  <light id="3" name="web_panel2" type="area" shape="rect" distribution="ies" visible="1" mat_id="13" mesh_id="%d"><size half_length="0.4" half_width="0.4" />'
                  '<ies data="web_00001.ies" loc="data/web_00001.ies" point_area="1" matrix="1 0 0 0 0 0.8 -0.6 0 0 0.6 0.8 0 0 0 0 1" /><intensity><color val="0.7 0.8 1" /><multiplier val="80.0" /></intensity></light>' % (ies_quad1, ies_quad2) if args.ies else
                  '\n  <light id="1" name="bulb" type="point" shape="point" distribution="uniform" visible="1"><intensity><color val="1 0.8 0.6" /><multiplier val="40.0" /></intensity></light>'
                  '\n  <light id="2" name="spot" type="point" shape="point" distribution="spot" visible="1"><falloff_angle val="70" /><falloff_angle2 val="40" />'
                  '<intensity><color val="0.7 0.8 1" /><multiplier val="90.0" /></intensity></light>'
                  '\n  <light id="3" name="sun" type="directional" shape="point" distribution="directional" visible="1"><size inner_radius="30" outer_radius="40" />'
                  '<shadow_softness val="2.0" /><intensity><color val="1 0.95 0.85" /><multiplier val="2.5" /></intensity></light>'
                  '\n  <light id="4" name="globe" type="area" shape="sphere" distribution="uniform" visible="1" mat_id="12" mesh_id="%d"><size radius="0.35" />'
                  '<intensity><color val="1 0.88 0.72" /><multiplier val="25.0" /></intensity></light>'
                  '\n  <light id="5" name="lamp" type="area" shape="mesh" distribution="uniform" visible="1" mat_id="13" mesh_id="%d">'
                  '<intensity><color val="0.6 0.8 1" /><multiplier val="30.0" /></intensity></light>' % (globe_mesh or 0, lamp_mesh or 0) if args.delta_lights else '')
               + ('\n  <light id="2" name="portal" type="area" shape="rect" distribution="diffuse" visible="1" mat_id="12" mesh_id="%d"><size half_length="8.0" half_width="4.0" />'
                  '<intensity><color val="1 1 1" /><multiplier val="1.0" /></intensity><sky_portal val="1" source_id="1" /></light>'
                  '\n  <light id="3" name="sun" type="directional" shape="point" distribution="directional" visible="1"><size inner_radius="30" outer_radius="40" />'
                  '<shadow_softness val="2.0" /><intensity><color val="1 0.95 0.85" /><multiplier val="2.5" /></intensity></light>' % portal_mesh if args.portal else '')
               + '\n</lights_lib>')
    if args.back:   # the sky light's <back>: what the camera sees instead of the environment (RenderDriverRTE.cpp:946-967)
        lights = xml[-1]
        at = lights.index('name="sky"')
        end = lights.index("</light>", at)
        xml[-1] = lights[:end] + '<back mode="%s" multcolor="0.9 1.0 0.8"><texture id="1" type="texref" input_gamma="2.2" /></back>' % args.back + lights[end:]
    xml.append('<cam_lib>\n  <camera id="0" name="cam" type="uvn"><fov>60</fov><nearClipPlane>0.01</nearClipPlane><farClipPlane>200.0</farClipPlane>'
               '<up>0 1 0</up><position>-17 2.2 0.6</position><look_at>10 2.6 -0.4</look_at></camera>\n</cam_lib>')

    xml.append('<geometry_lib total_chunks="%d">' % (len(texs) + len(meshes)))
    tri_counts = {}
    for gid, (name, m) in enumerate(meshes):
        loc = "data/chunk_%05d.vsgf" % chunk
        vn, tn, total, offs = write_vsgf(os.path.join(out, loc), m)
        tri_counts[gid] = tn
        p = m["pos"][:, :3]
        bbox = " ".join("%.6g" % v for v in (p[:, 0].min(), p[:, 0].max(), p[:, 1].min(), p[:, 1].max(), p[:, 2].min(), p[:, 2].max()))
        xml.append('  <mesh id="%d" name="%s" type="vsgf" bytesize="%d" loc="%s" offset="0" vertNum="%d" triNum="%d" dl="0" path="" bbox="%s">' % (gid, name, total, loc, vn, tn, bbox))
        for tag, typ, (o, sz), app in zip(("positions", "normals", "tangents", "texcoords", "indices", "matindices"),
                                          ("array4f", "array4f", "array4f", "array2f", "array1i", "array1i"), offs,
                                          ("vertex", "vertex", "vertex", "vertex", "tlist", "primitive")):
            xml.append('    <%s type="%s" bytesize="%d" offset="%d" apply="%s" />' % (tag, typ, sz, o, app))
        xml.append("  </mesh>")
        chunk += 1
    xml.append("</geometry_lib>")
    xml.append('<render_lib>\n  <render_settings type="HydraModern" id="0"><width>%d</width><height>%d</height><method_primary>pathtracing</method_primary>'
               '<trace_depth>8</trace_depth><diff_trace_depth>8</diff_trace_depth><maxRaysPerPixel>1024</maxRaysPerPixel>%s</render_settings>\n</render_lib>'
               % (args.width, args.height, "<split_alpha_tree>1</split_alpha_tree>" if args.two_trees else ""))

    inst, total_tris = [], 0

    def add(mesh_id, matrix, extra=""):
        nonlocal total_tris
        inst.append('    <instance id="%d" mesh_id="%d" rmap_id="-1" scn_id="0" scn_sid="0" matrix="%s"%s />' % (len(inst), mesh_id, matrix, extra))
        total_tris += tri_counts[mesh_id]
    xs = np.linspace(-16.5, 16.5, 12)
    for row_z in (-4.0, 4.0):
        for x in xs:
            add(0, mat4(t=(x, 0, row_z)))
        for x0, x1 in zip(xs[:-1], xs[1:]):
            add(1, mat4(t=((x0 + x1) / 2, 7.6, row_z), scale=(x1 - x0) / 3.2 * 0.98))
    for _ in range(120):
        add(2, mat4(scale=rng.uniform(0.8, 1.2), yaw=rng.uniform(0, 2 * np.pi), t=(rng.uniform(-18.5, 18.5), 0.02, rng.uniform(-9, 9))))
    add(3, mat4())
    for k, x in enumerate((-12.0, -4.0, 4.0, 12.0)):
        add(4, mat4(t=(x, 1.8, -9.2 if k % 2 == 0 else 9.2), yaw=0.0 if k % 2 == 0 else np.pi))
    add(5, mat4())
    if args.cutouts:
        for _ in range(60):
            add(7, mat4(scale=rng.uniform(0.8, 1.6), yaw=rng.uniform(0, 2 * np.pi), t=(rng.uniform(-18.0, 18.0), 0.0, rng.uniform(-8.5, 8.5))))
        add(7, mat4(scale=2.2, yaw=np.pi / 2, t=(-9.0, 0.1, 2.5)))       # a big one in the camera's view
    light_m = mat4(t=(2.0, 9.6, 0.0))
    add(6, light_m, ' light_id="0" linst_id="0"')
    globe_m = mat4(scale=1.4, t=(-6.0, 3.2, 2.0))
    lamp_m = mat4(scale=1.8, yaw=0.7, t=(5.0, 2.4, -2.5))
    if args.delta_lights:
        add(globe_mesh, globe_m, ' light_id="4" linst_id="4"')
        add(lamp_mesh, lamp_m, ' light_id="5" linst_id="5"')
    tube_m = mat4(yaw=0.6, t=(-6.0, 6.5, 1.0), rot_x=0.2)
    half_tube_m = mat4(scale=1.3, yaw=-0.4, t=(7.0, 1.2, -3.0), rot_x=1.2)
    portal_m = mat4(t=(0.0, 10.0, 0.0))
    if args.tube_lights:
        add(tube_mesh, tube_m, ' light_id="1" linst_id="1"')
        add(half_tube_mesh, half_tube_m, ' light_id="2" linst_id="2"')
        add(tlamp_mesh, lamp_m, ' light_id="3" linst_id="3"')
    ies_panel_m = mat4(yaw=0.5, t=(-7.0, 6.0, -2.0), rot_x=0.25)
    ies_panel2_m = mat4(scale=1.2, yaw=-0.8, t=(8.0, 5.0, 2.5), rot_x=-0.3)
    if args.ies:
        add(ies_quad1, ies_panel_m, ' light_id="2" linst_id="2"')
        add(ies_quad2, ies_panel2_m, ' light_id="3" linst_id="3"')
    if args.portal:
        add(portal_mesh, portal_m, ' light_id="2" linst_id="2"')
    xml.append('<scenes>\n  <scene id="0" name="atrium250k" discard="1" bbox="-20 20 0 10 -10 10">')
    xml.append('    <instance_light id="0" light_id="0" matrix="%s" lgroup_id="-1" />' % light_m)
    if args.sky:
        xml.append('    <instance_light id="1" light_id="1" matrix="%s" lgroup_id="-1" />' % mat4())
    if args.perez:
        xml.append('    <instance_light id="2" light_id="3" matrix="%s" lgroup_id="-1" />' % mat4(t=(0.0, 30.0, 0.0), rot_x=-0.45))
    if args.delta_lights:
        xml.append('    <instance_light id="1" light_id="1" matrix="%s" lgroup_id="-1" />' % mat4(t=(-10.0, 5.0, 1.0)))
        xml.append('    <instance_light id="2" light_id="2" matrix="%s" lgroup_id="-1" />' % mat4(t=(6.0, 7.5, -2.0), rot_x=0.3))
        xml.append('    <instance_light id="3" light_id="3" matrix="%s" lgroup_id="-1" />' % mat4(t=(0.0, 30.0, 0.0), rot_x=-0.35))
        xml.append('    <instance_light id="4" light_id="4" matrix="%s" lgroup_id="-1" />' % globe_m)
        xml.append('    <instance_light id="5" light_id="5" matrix="%s" lgroup_id="-1" />' % lamp_m)
    if args.tube_lights:
        xml.append('    <instance_light id="1" light_id="1" matrix="%s" lgroup_id="-1" />' % tube_m)
        xml.append('    <instance_light id="2" light_id="2" matrix="%s" lgroup_id="-1" />' % half_tube_m)
        xml.append('    <instance_light id="3" light_id="3" matrix="%s" lgroup_id="-1" />' % lamp_m)
    if args.ies:
        xml.append('    <instance_light id="1" light_id="1" matrix="%s" lgroup_id="-1" />' % mat4(yaw=0.9, t=(-1.0, 5.5, 1.5), rot_x=0.4))
        xml.append('    <instance_light id="2" light_id="2" matrix="%s" lgroup_id="-1" />' % ies_panel_m)
        xml.append('    <instance_light id="3" light_id="3" matrix="%s" lgroup_id="-1" />' % ies_panel2_m)
    if args.portal:
        xml.append('    <instance_light id="2" light_id="2" matrix="%s" lgroup_id="-1" />' % portal_m)
        xml.append('    <instance_light id="3" light_id="3" matrix="%s" lgroup_id="-1" />' % mat4(t=(0.0, 30.0, 0.0), rot_x=-0.35))
    xml += inst
    xml.append("  </scene>\n</scenes>")
    with open(os.path.join(out, "statex_00001.xml"), "w") as f:
        f.write("\n".join(xml) + "\n")
    print("atrium: %d instances, %d triangles (unique + instanced), %d unique" % (len(inst), total_tris, sum(tri_counts.values())))


if __name__ == "__main__":
    main()
