// hk_trace.h -- BVH4 (two-level, instanced) traversal + Moeller-Trumbore for one ray per lane.
//
// Behaviour contract (rows a/T1, a/T2): closest hit exactly as BVH4InstTraverse / BVH4Traverse
// (hydra_drv/ctrace.h:841-1062, :669-838) over the reference's flattened layout: children tested with
// RayBoxIntersectionLite2 (:32-53), ordered near->far by the same 5 compare-swaps (:906-960) so that equal-t ties
// resolve to the same triangle, <=3 pushes per quad, 80-entry stack (+2, see HkStackT) with silent drop, instance leaves re-express the
// ray in object space with the direction left un-normalised; leaf test :124-182 (u,v > -1e-6, u+v < 1+1e-6,
// t_min < t < best).  The shadow form answers "any triangle with t_min < t < t_far" which is what
// IntegratorCommon::shadowTrace computes from a closest hit (CPUExp_Integrators_Common.cpp:156-180).
//
// MI355X mapping: one ray per lane, the 4 child boxes of a quad are one 128-byte line; the traversal stack lives in
// LDS, transposed ([entry][lane]) so a wave's push/pop touches 64 consecutive banks; entries beyond HK_LDS_DEPTH
// spill to scratch, keeping the reference's 80-entry semantics without paying for them in LDS occupancy.
#pragma once
#include "hk_common.h"
#ifndef HK_HOST_EMU
typedef float hk_v4f_t __attribute__((ext_vector_type(4)));
#endif

#define HK_STACK_SIZE 80
#ifndef HK_LDS_DEPTH
#define HK_LDS_DEPTH 24
#endif
#ifndef HK_TRACE_BLOCK
#define HK_TRACE_BLOCK 128
#endif
// LDS copy of the quads rays visit most (persistent traversal kernels only): HK_TOP_QUADS quads, HK_TOP_STRIDE float4 apart
// (one float4 of padding so that lanes reading the same piece of different quads spread over the banks).  In the node copy
// those kernels walk, an inner link to cached quad `slot` reads HK_TOP_FLAG | slot (bit 31 clear: not a leaf).
#ifndef HK_TOP_QUADS
#define HK_TOP_QUADS 21
#endif
#ifndef HK_TOP_STRIDE
#define HK_TOP_STRIDE 9
#endif
#define HK_TOP_FLAG 0x40000000
// LDS copy of the triangles of the leaves rays visit most (same kernels): a pool of HK_TOP_TRIS triangles (3 float4 each).  In the
// node copy those kernels walk, the link to a cached leaf reads LEAF | count << 27 | HK_LEAF_LDS_FLAG | first triangle of the pool.
#ifndef HK_TOP_TRIS
#define HK_TOP_TRIS 16
#endif
#define HK_LEAF_LDS_FLAG 0x04000000

// the software texture fetch of hk_shading.h (declared here for the alpha test of the leaf loop)
HK_DEV_CALL f3 sample2DExtCall(int samplerOffset, f2 texCoord, const float* blob, const int* texTable, const int4* texStorage, const float* srgbLut);

struct TravCounters { uint32_t quads, insts, tris, leaves, oob; };   // oob: fetches a range-checked buffer load would have answered with zeros (must stay 0)

HK_DEV f3 SafeInverse(f3 d) {   // hydra_drv/cglobals.h:726-735
  const float ooeps = 1.0e-36f;
  f3 r;
  r.x = 1.0f / (fabsf(d.x) > ooeps ? d.x : copysignf(ooeps, d.x));
  r.y = 1.0f / (fabsf(d.y) > ooeps ? d.y : copysignf(ooeps, d.y));
  r.z = 1.0f / (fabsf(d.z) > ooeps ? d.z : copysignf(ooeps, d.z));
  return r;
}

HK_DEV float2 RayBox(f3 o, f3 inv, float4 lo4, float4 hi4) {
  const float lo = inv.x * (lo4.x - o.x), hi = inv.x * (hi4.x - o.x);
  const float lo1 = inv.y * (lo4.y - o.y), hi1 = inv.y * (hi4.y - o.y);
  const float lo2 = inv.z * (lo4.z - o.z), hi2 = inv.z * (hi4.z - o.z);
  float tmin = fminf(lo, hi), tmax = fmaxf(lo, hi);
  tmin = fmaxf(tmin, fminf(lo1, hi1)); tmax = fminf(tmax, fmaxf(lo1, hi1));
  tmin = fmaxf(tmin, fminf(lo2, hi2)); tmax = fminf(tmax, fmaxf(lo2, hi2));
  return make_float2(tmin, tmax);
}

// LDS pointers are typed with address space 3 so that pushes/pops compile to ds_write_b32/ds_read_b32 with an
// immediate stride (a generic int* makes the compiler emit flat_* accesses plus 64-bit address arithmetic).
#ifdef HK_HOST_EMU
typedef int hk_lds_int;
typedef float4 hk_lds_f4;
#else
typedef __attribute__((address_space(3))) int hk_lds_int;
typedef __attribute__((address_space(3))) hk_v4f_t hk_lds_f4;
#endif

// Capacity: the reference tests `top < 80` ONCE per quad and then pushes up to three links (ctrace.h:964-985), so with
// top == 79 it writes entries 79, 80 and 81 of an 80-entry array.  Entries 80 and 81 exist here (HK_STACK_SLACK), so a
// tree deeper than the stack behaves like the reference does when its two stray writes land on harmless memory -- same
// pushes, same pops, same visit order -- instead of corrupting the neighbouring scratch words.
#define HK_STACK_SLACK 2
template <int LDS_DEPTH>
struct HkStackT {
  hk_lds_int* lds;   // this lane's column in the block's LDS stack, stride = HK_TRACE_BLOCK entries
  int spill[HK_STACK_SIZE + HK_STACK_SLACK - LDS_DEPTH];
  HK_DEV_MEMBER void init(int* sharedBase, int lane) { lds = (hk_lds_int*)sharedBase + lane; }
  HK_DEV_MEMBER void put(int top, int v) {
    if (top < LDS_DEPTH) lds[top * HK_TRACE_BLOCK] = v; else spill[top - LDS_DEPTH] = v;
  }
  HK_DEV_MEMBER int get(int top) const {
    if (top < 0) return 0;   // the reference reads an unused slot here; the value is never acted on
    return (top < LDS_DEPTH) ? lds[top * HK_TRACE_BLOCK] : spill[top - LDS_DEPTH];
  }
};
typedef HkStackT<HK_LDS_DEPTH> HkStack;
// The persistent shadow kernel needs fewer registers than the closest-hit one (no hit ids to carry): with a shorter LDS part
// of the stack its blocks are small enough for 12 of them (6 waves per SIMD) to be resident per CU instead of 10.
#ifndef HK_LDS_DEPTH_SHADOW
#define HK_LDS_DEPTH_SHADOW 19
#endif
#ifndef HK_TRACE_MIN_WAVES_SHADOW
#define HK_TRACE_MIN_WAVES_SHADOW 6
#endif

// How the traversal reads the node and triangle arrays.  Measured on MI355X (tools/micro/ta_bench.hip,
// profiles/r01/ta_microbench_divergent_loads.log): when every lane reads its own 128-byte line, a 16-byte global_load
// (64-bit address per lane) costs the CU's address path ~2.0 clk per lane even for the 2nd..8th piece of the same line,
// while raw buffer loads (one 32-bit offset per lane + immediate piece offset) cost ~1.0 clk per lane.  The traversal
// kernels are bound by exactly that path, so on the device both arrays are raw buffers.
#ifdef HK_HOST_EMU
struct BvhView {
  const float4* nodes; const float4* tris; bool leafEnc; const hk_lds_f4* top; const hk_lds_f4* topTri;
  const uint2* alpha; const int* texTable; const int4* texStorage; const float* srgbLut;
  HK_DEV_MEMBER float4 topPiece(int link, int piece) const { return top[(link & 0xff) * HK_TOP_STRIDE + piece]; }
  HK_DEV_MEMBER float4 topTriPiece(int index) const { return topTri[index]; }
  HK_DEV_MEMBER float4 node(int quad, int piece) const { return nodes[size_t(quad) * 8 + piece]; }
  HK_DEV_MEMBER float4 tri(int index) const { return tris[index]; }
  HK_DEV_MEMBER bool nodeInRange(int) const { return true; }   // the host build runs under AddressSanitizer instead
  HK_DEV_MEMBER bool triInRange(int, int) const { return true; }
};
HK_DEV BvhView make_bvh_view(const float4* nodes, unsigned, const float4* tris, unsigned, bool leafEnc = false) { BvhView v; v.nodes = nodes; v.tris = tris; v.leafEnc = leafEnc; v.top = nullptr; v.topTri = nullptr; v.alpha = nullptr; v.texTable = nullptr; v.texStorage = nullptr; v.srgbLut = nullptr; return v; }
#else
typedef float hk_v4f __attribute__((ext_vector_type(4)));
struct BvhView {
  __amdgpu_buffer_rsrc_t nodes, tris;
  uint32_t nodeBytes, triBytes;
  bool leafEnc;   // triangle-leaf links of the device copy carry the triangle count (see HK_LEAF_COUNT_SHIFT)
  const hk_lds_f4* top;   // LDS copy of the hottest quads (trav_run<.., TOPCACHE = true> only)
  const hk_lds_f4* topTri;   // LDS pool of the hottest leaves' triangles
  // alpha-tested trees (IntersectLeaf<.., ALPHA = true>): the tree's alpha table and what its texture fetch needs
  const uint2* alpha; const int* texTable; const int4* texStorage; const float* srgbLut;
  HK_DEV_MEMBER float4 topPiece(int link, int piece) const {
    const hk_v4f_t v = top[(link & 0xff) * HK_TOP_STRIDE + piece];
    return make_float4(v.x, v.y, v.z, v.w);
  }
  HK_DEV_MEMBER float4 topTriPiece(int index) const {
    const hk_v4f_t v = topTri[index];
    return make_float4(v.x, v.y, v.z, v.w);
  }
  HK_DEV_MEMBER float4 node(int quad, int piece) const {
    const hk_v4f v = __builtin_bit_cast(hk_v4f, __builtin_amdgcn_raw_buffer_load_b128(nodes, uint32_t(quad) * 128u + uint32_t(piece) * 16u, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  HK_DEV_MEMBER float4 tri(int index) const {
    const hk_v4f v = __builtin_bit_cast(hk_v4f, __builtin_amdgcn_raw_buffer_load_b128(tris, uint32_t(index) * 16u, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  // raw buffer loads answer an out-of-range offset with zeros instead of faulting; the counting kernel variants use these
  // two predicates to prove that no fetch ever relies on that (TravCounters::oob, tests/test_parity_gpu.py)
  HK_DEV_MEMBER bool nodeInRange(int quad) const { return quad >= 0 && (unsigned long long)(uint32_t)quad * 128ull + 128ull <= nodeBytes; }
  HK_DEV_MEMBER bool triInRange(int first, int end) const { return first >= 0 && end >= first && (unsigned long long)(uint32_t)end * 16ull <= triBytes; }
};
// both sizes come from kernel arguments (wave-uniform), < 4 GiB each (checked by upload_bvh)
HK_DEV BvhView make_bvh_view(const float4* nodes, unsigned nodeBytes, const float4* tris, unsigned triBytes, bool leafEnc = false) {
  BvhView v;
  v.leafEnc = leafEnc;
  v.top = nullptr; v.topTri = nullptr;
  v.alpha = nullptr; v.texTable = nullptr; v.texStorage = nullptr; v.srgbLut = nullptr;
  v.nodeBytes = nodeBytes; v.triBytes = triBytes;
  v.nodes = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(nodes), 0, nodeBytes, 0x00020000);
  v.tris = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(tris), 0, triBytes, 0x00020000);
  return v;
}
#endif

// Device copy only (hydra_hip_upload_bvh): a link to a triangle leaf whose list starts right behind its header, holds 1..15
// triangles and sits below float4 index 2^27 carries the count in bits 27..30, so the leaf test can issue its triangle loads
// without first waiting for the header (one dependent memory round trip less per leaf visit).  Count 0 = read the header.
#define HK_LEAF_COUNT_SHIFT 27
#define HK_LEAF_OFFSET_MASK 0x07ffffff
// Device copy of the triangle lists only: bits 24..27 of a triangle's geomId word carry the shading class of its material
// (hydra_hip.hip, k_tag_triangle_classes), so a closest hit arrives at the bounce kernel already labelled and the kernel can
// group the paths of a workgroup by class before shading them.  HK_GEOM_ID() gives the reference's geomId back; the miss
// value 0xC0000000 (Make_Lite_Hit, cglobals.h:1256-1266) has no class bits and stays as it is.
#define HK_CLASS_SHIFT 24
#define HK_CLASS_BITS 0x0f000000
#define HK_GEOM_ID(g) ((g) & ~HK_CLASS_BITS)
#define HK_GEOM_CLASS(g) (((g) >> HK_CLASS_SHIFT) & 15)

// Alpha test of a candidate hit, IntersectAllPrimitivesInLeafAlpha (ctrace.h:330-412): the tree's alpha table holds one uint2 per
// float4 of the triangle list -- {sampler position in the table | flags, texture coordinate of the vertex packed 2 x 16 bit}
// (RenderDriverRTE_AlphaTestTable.cpp:65-224) -- and the opacity samplers behind them; the hit counts when the texel's largest
// channel exceeds 0.5.  decompressTexCoord16: ctrace.h:316-325; sample2DLite: cfetch.h:738-760.
HK_DEV f2 decompressTexCoord16(uint32_t packed) {
  const float fx = (1.0f / 65535.0f) * float(packed & 0x0000FFFFu), fy = (1.0f / 65535.0f) * float((packed & 0xFFFF0000u) >> 16);
  return mk2(2.0f * fx - 1.0f, 2.0f * fy - 1.0f);
}
HK_DEV bool alphaTestPasses(const BvhView& bv, int triAddress, float u, float v) {
  const uint2 a0 = bv.alpha[triAddress], a1 = bv.alpha[triAddress + 1], a2 = bv.alpha[triAddress + 2];
  const f2 A = decompressTexCoord16(a0.y), B = decompressTexCoord16(a1.y), C = decompressTexCoord16(a2.y);
  const float w = 1.0f - u - v;
  const f2 tc = mk2((w * A.x + v * B.x) + u * C.x, (w * A.y + v * B.y) + u * C.y);
  if (a0.x == 0xFFFFFFFFu || a0.x == HYDRA_INVALID_TEXTURE || int(a0.x) <= 0) return true;   // no opacity texture: white, selector 1
  const f3 c = sample2DExtCall(0, tc, reinterpret_cast<const float*>(bv.alpha + a0.x), bv.texTable, bv.texStorage, bv.srgbLut);
  return fmaxf(c.x, fmaxf(c.y, c.z)) > 0.5f;
}

template <bool ANYHIT, bool COUNT, bool TOPTRIS = false, bool ALPHA = false>
HK_DEV HydraLiteHit IntersectLeaf(f3 ray_pos, f3 ray_dir, int leaf_offset, float t_min, HydraLiteHit res,
                                  const BvhView& bv, int instId, bool useInstId, TravCounters& cnt) {
  int first, count;
  const int enc = bv.leafEnc ? ((leaf_offset >> HK_LEAF_COUNT_SHIFT) & 15) : 0;
  const bool inLds = TOPTRIS && enc != 0 && (leaf_offset & HK_LEAF_LDS_FLAG) != 0;   // one of the hottest leaves: its triangles sit in LDS
  if (inLds) { first = (leaf_offset & 0xff) * 3; count = enc; }
  else if (enc != 0) { first = (leaf_offset & HK_LEAF_OFFSET_MASK) + 1; count = enc; }
  else {
    const float4 hdr = bv.tri(bv.leafEnc ? (leaf_offset & HK_LEAF_OFFSET_MASK) : leaf_offset);
    first = as_int(hdr.x); count = as_int(hdr.y);
  }
  const int end = first + count * 3;
  if (COUNT) { cnt.tris += uint32_t(count); cnt.leaves++; if (!inLds && (!bv.triInRange(first, end) || !bv.triInRange(leaf_offset & HK_LEAF_OFFSET_MASK, (leaf_offset & HK_LEAF_OFFSET_MASK) + 1))) cnt.oob++; }
  for (int a = first; a < end; a += 3) {
    float4 d1, d2, d3;
    if (inLds) { d1 = bv.topTriPiece(a); d2 = bv.topTriPiece(a + 1); d3 = bv.topTriPiece(a + 2); }
    else { d1 = bv.tri(a); d2 = bv.tri(a + 1); d3 = bv.tri(a + 2); }
    const f3 A = xyz(d1), B = xyz(d2), C = xyz(d3);
    const f3 edge1 = B - A, edge2 = C - A;
    const f3 pvec = cross(ray_dir, edge2);
    const f3 tvec = ray_pos - A;
    const f3 qvec = cross(tvec, edge1);
    const float invDet = 1.0f / dot(edge1, pvec);
    const float v = dot(tvec, pvec) * invDet;
    const float u = dot(qvec, ray_dir) * invDet;
    const float t = dot(edge2, qvec) * invDet;
    if (v > -1e-6f && u > -1e-6f && (u + v < 1.0f + 1e-6f) && t > t_min && t < res.t) {
      if (ALPHA && !alphaTestPasses(bv, a, u, v)) continue;
      res.t = t;
      res.primId = as_int(d1.w);
      res.geomId = as_int(d2.w);
      res.instId = useInstId ? instId : as_int(d3.w);
      if (ANYHIT) return res;
    }
  }
  return res;
}

// Resumable traversal state: everything BVH4InstTraverse keeps in locals.  trav_run() can be left early (when too few
// lanes of the wave are still walking) and re-entered after the idle lanes fetched new rays; the sequence of node
// visits, pushes and triangle tests of each ray is the same as in the uninterrupted loop.
struct TravState {
  f3 pos, dir, inv, opos, odir, oinv;   // oinv = SafeInverse(odir), kept instead of recomputed when the ray leaves an instance
  HydraLiteHit hit;
  int top, left, instDeep, instTop, instId;
  bool searching;
  // speculative walk (trav_run_vote, HK_SPEC_LEAF): a triangle leaf put aside while the lane goes on with inner nodes, and whether the lane has since popped its way out of
  // the instance that leaf belongs to (the world ray is restored only after the leaf has been tested)
  int pend;
  bool pendExit;
};
HK_DEV void trav_init(TravState& t, f3 pos, f3 dir, const HydraLiteHit& hit, int rootLink = 1) {
  t.pos = pos; t.dir = dir; t.inv = SafeInverse(dir);
  t.opos = mk3(0, 0, 0); t.odir = mk3(0, 0, 0); t.oinv = mk3(0, 0, 0);
  t.hit = hit;
  t.top = 0; t.left = rootLink; t.instDeep = 0; t.instTop = 0; t.instId = -1;
  t.searching = true;
  t.pend = -1; t.pendExit = false;
}

// ---- the three kinds of step a ray alternates between.  Each is what BVH4InstTraverse does in one turn of its inner loop
// (a quad), at a triangle leaf, or at an instance leaf; a ray's sequence of steps -- hence every box test, push, pop and triangle
// test, and their order -- is the same whatever schedules them (trav_run: the reference's loop nest; trav_run_vote: by wave vote).

// one quad: fetch its 4 child boxes, test, order near -> far, push, descend or pop (ctrace.h:866-1006)
template <bool COUNT, bool TOPCACHE, class STACK, bool SORTED = true>
HK_DEV void trav_quad_step(TravState& t, const BvhView& bv, const bool haveInst, const float t_rayMin, STACK& stack, TravCounters& cnt, const bool deferExit = false) {
  float4 n0a, n0b, n1a, n1b, n2a, n2b, n3a, n3b;
  if (TOPCACHE && (t.left & HK_TOP_FLAG)) {   // one of the hottest quads: 8 LDS reads instead of 8 trips through the texture addresser
    n0a = bv.topPiece(t.left, 0); n0b = bv.topPiece(t.left, 1); n1a = bv.topPiece(t.left, 2); n1b = bv.topPiece(t.left, 3);
    n2a = bv.topPiece(t.left, 4); n2b = bv.topPiece(t.left, 5); n3a = bv.topPiece(t.left, 6); n3b = bv.topPiece(t.left, 7);
  } else {
    n0a = bv.node(t.left, 0); n0b = bv.node(t.left, 1); n1a = bv.node(t.left, 2); n1b = bv.node(t.left, 3);
    n2a = bv.node(t.left, 4); n2b = bv.node(t.left, 5); n3a = bv.node(t.left, 6); n3b = bv.node(t.left, 7);
  }
  if (COUNT) { cnt.quads++; if (!(TOPCACHE && (t.left & HK_TOP_FLAG)) && !bv.nodeInRange(t.left)) cnt.oob++; }
  int c0 = as_int(n0a.w), c1 = as_int(n1a.w), c2 = as_int(n2a.w), c3 = as_int(n3a.w);
#ifdef HK_HOST_EMU
  const bool v0 = !((uint32_t(c0) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n0b.w)) == HYDRA_BVH_INVALID));
  const bool v1 = !((uint32_t(c1) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n1b.w)) == HYDRA_BVH_INVALID));
  const bool v2 = !((uint32_t(c2) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n2b.w)) == HYDRA_BVH_INVALID));
  const bool v3 = !((uint32_t(c3) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n3b.w)) == HYDRA_BVH_INVALID));
#else
  // the device copy of the node array has the boxes of invalid children (both link words 0xFFFFFFFF, ctrace.h:889-892)
  // overwritten with NaN by k_prepare_bvh at upload: every comparison below is then false for them, which is what the
  // reference's IsValidNode term achieves -- 12 VALU instructions less per quad in a loop that is VALU-issue bound
  const bool v0 = true, v1 = true, v2 = true, v3 = true;
#endif
#ifdef HK_EXP_SETPRIO   /* timing experiment: the wave that has its quad runs its box tests ahead of the others */
  __builtin_amdgcn_s_setprio(HK_EXP_SETPRIO);
#endif
  const float2 t0 = RayBox(t.pos, t.inv, n0a, n0b), t1 = RayBox(t.pos, t.inv, n1a, n1b);
  const float2 t2 = RayBox(t.pos, t.inv, n2a, n2b), t3 = RayBox(t.pos, t.inv, n3a, n3b);
  float k0 = ((t0.x <= t0.y) && (t0.y >= t_rayMin) && (t0.x <= t.hit.t) && v0) ? t0.x : HK_MAXFLOAT;
  float k1 = ((t1.x <= t1.y) && (t1.y >= t_rayMin) && (t1.x <= t.hit.t) && v1) ? t1.x : HK_MAXFLOAT;
  float k2 = ((t2.x <= t2.y) && (t2.y >= t_rayMin) && (t2.x <= t.hit.t) && v2) ? t2.x : HK_MAXFLOAT;
  float k3 = ((t3.x <= t3.y) && (t3.y >= t_rayMin) && (t3.x <= t.hit.t) && v3) ? t3.x : HK_MAXFLOAT;
  if (!SORTED) {   // any-hit rays only (the answer -- is anything in the way -- does not depend on the order): the children that were hit in stored order, no sorting network
#ifdef HK_EXP_ANYHIT_NEAREST   /* experiment: the nearest child first (three exchanges instead of the network's five), the others in stored order */
#define HK_CSWAP0(ka, kb, ca, cb) { const bool sw = (kb < ka); const float tk = sw ? kb : ka; kb = sw ? ka : kb; ka = tk; const int tc = sw ? cb : ca; cb = sw ? ca : cb; ca = tc; }
    HK_CSWAP0(k0, k1, c0, c1) HK_CSWAP0(k0, k2, c0, c2) HK_CSWAP0(k0, k3, c0, c3)
#undef HK_CSWAP0
#endif
    const bool h0 = k0 < HK_MAXFLOAT, h1 = k1 < HK_MAXFLOAT, h2 = k2 < HK_MAXFLOAT, h3 = k3 < HK_MAXFLOAT;
    const bool stackHaveSpace = (t.top < HK_STACK_SIZE);
    if (h3 && (h0 || h1 || h2) && stackHaveSpace) { stack.put(t.top, c3); t.top++; }
    if (h2 && (h0 || h1) && stackHaveSpace) { stack.put(t.top, c2); t.top++; }
    if (h1 && h0 && stackHaveSpace) { stack.put(t.top, c1); t.top++; }
    if (h0 || h1 || h2 || h3) t.left = h0 ? c0 : (h1 ? c1 : (h2 ? c2 : c3));
    else if (t.top >= 0) { t.top--; t.left = stack.get(t.top); }
    t.searching = !(t.left & int(HYDRA_BVH_LEAF)) && (t.top >= 0);
    t.left = t.left & 0x7fffffff;
    if (haveInst && t.top < t.instTop && t.instDeep == 1) { t.pos = t.opos; t.dir = t.odir; t.inv = t.oinv; t.instDeep = 0; }
    return;
  }
#define HK_CSWAP(ka, kb, ca, cb) { const bool sw = (kb < ka); const float tk = sw ? kb : ka; kb = sw ? ka : kb; ka = tk; const int tc = sw ? cb : ca; cb = sw ? ca : cb; ca = tc; }
  HK_CSWAP(k0, k1, c0, c1) HK_CSWAP(k2, k3, c2, c3) HK_CSWAP(k0, k2, c0, c2) HK_CSWAP(k1, k3, c1, c3) HK_CSWAP(k1, k2, c1, c2)
#ifdef HK_EXP_EXTRA_SORT   /* timing experiment: the network again on sorted keys changes nothing but costs its ~25 VALU instructions */
  for (int rep = 0; rep < HK_EXP_EXTRA_SORT; rep++) {
    asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3));
    HK_CSWAP(k0, k1, c0, c1) HK_CSWAP(k2, k3, c2, c3) HK_CSWAP(k0, k2, c0, c2) HK_CSWAP(k1, k3, c1, c3) HK_CSWAP(k1, k2, c1, c2)
  }
#endif
#undef HK_CSWAP
  // keys are sorted and misses carry MAXFLOAT, so the children hit are a prefix: nothing is pushed unless k1 is a hit
  // (same pushes, in the same order, as the three separate tests of ctrace.h:962-975)
  if (k1 < HK_MAXFLOAT) {
    const bool stackHaveSpace = (t.top < HK_STACK_SIZE);
    if (k3 < HK_MAXFLOAT && stackHaveSpace) { stack.put(t.top, c3); t.top++; }
    if (k2 < HK_MAXFLOAT && stackHaveSpace) { stack.put(t.top, c2); t.top++; }
    if (stackHaveSpace) { stack.put(t.top, c1); t.top++; }
  }
  if (k0 < HK_MAXFLOAT) t.left = c0;
  else if (t.top >= 0) { t.top--; t.left = stack.get(t.top); }
  t.searching = !(t.left & int(HYDRA_BVH_LEAF)) && (t.top >= 0);
  t.left = t.left & 0x7fffffff;
#ifdef HK_EXP_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  if (haveInst && t.top < t.instTop && t.instDeep == 1) {
    if (deferExit) t.pendExit = true;   // a leaf of this instance is still waiting for its triangle tests: it needs the object-space ray
    else { t.pos = t.opos; t.dir = t.odir; t.inv = t.oinv; t.instDeep = 0; }   // = SafeInverse(t.odir), same bits (ctrace.h:1000-1006)
  }
}
// what follows every leaf of either kind (ctrace.h:1043-1056)
HK_DEV void trav_after_leaf(TravState& t, const bool haveInst) {
  t.searching = !(t.left & int(HYDRA_BVH_LEAF));
  t.left = t.left & 0x7fffffff;
  if (haveInst && t.top < t.instTop && t.instDeep == 1) {
    t.pos = t.opos; t.dir = t.odir; t.inv = t.oinv; t.instDeep = 0;
  }
}
// a triangle leaf (of a plain tree, or inside an instance): test its triangles, pop.  Returns true when an any-hit ray is finished by it.
template <bool ANYHIT, bool COUNT, class STACK, bool TOPTRIS, bool ALPHA>
HK_DEV bool trav_tri_step(TravState& t, const BvhView& bv, const bool haveInst, const float t_rayMin, STACK& stack, TravCounters& cnt) {
  if (!haveInst) {
    if (t.top >= 0) {
      t.hit = IntersectLeaf<ANYHIT, COUNT, TOPTRIS, false>(t.pos, t.dir, t.left, t_rayMin, t.hit, bv, 0, false, cnt);   // BVH4Traverse has no alpha form (Common.cpp:146)
      if (ANYHIT && t.hit.primId != -1) { t.top = -1; return true; }
    }
  } else {
    t.hit = IntersectLeaf<ANYHIT, COUNT, TOPTRIS, ALPHA>(t.pos, t.dir, t.left, t_rayMin, t.hit, bv, t.instId, true, cnt);
    if (ANYHIT && t.hit.primId != -1) { t.top = -1; return true; }
  }
  t.top--;
  t.left = stack.get(t.top);
  trav_after_leaf(t, haveInst);
  return false;
}
// an instance leaf: the ray goes into the object's space and on to the root of its tree (ctrace.h:1023-1041)
template <bool COUNT>
HK_DEV void trav_inst_step(TravState& t, const BvhView& bv, TravCounters& cnt) {
  t.instDeep = 1;
  t.opos = t.pos; t.odir = t.dir; t.oinv = t.inv;
  const int nextOffset = as_int(bv.node(t.left, 0).w);
  m44 matrix;
  matrix.c[0] = bv.node(t.left, 2); matrix.c[1] = bv.node(t.left, 3); matrix.c[2] = bv.node(t.left, 4); matrix.c[3] = bv.node(t.left, 5);
  t.instId = as_int(bv.node(t.left, 6).x);
  if (COUNT) { cnt.insts++; if (!bv.nodeInRange(t.left)) cnt.oob++; }
  t.pos = mul4x3(matrix, t.pos);
  t.dir = mul3x3(matrix, t.dir);   // stays un-normalised so t keeps world units
  t.inv = SafeInverse(t.dir);
  t.instTop = t.top;
  t.left = nextOffset;
  trav_after_leaf(t, true);
}

// The reference's loop nest, one ray per lane: quads until a leaf, the leaf, again.  Returns true when the ray is finished; false when
// it was suspended because fewer than minActive lanes were still traversing (minActive <= 0: never suspend).
// (A/B reference for the vote: the text of the loop nest as it was measured in rounds 1-2, not yet expressed through the step functions)
template <bool ANYHIT, bool COUNT, bool TOPCACHE = false, class STACK = HkStack, bool TOPTRIS = false, bool ALPHA = false>
HK_DEV bool trav_run(TravState& t, const BvhView& bv, const bool haveInst,
                     const float t_rayMin, STACK& stack, TravCounters& cnt, const int minActive) {
  while (t.top >= 0) {
    while (t.searching) {
      float4 n0a, n0b, n1a, n1b, n2a, n2b, n3a, n3b;
      if (TOPCACHE && (t.left & HK_TOP_FLAG)) {   // one of the hottest quads: 8 LDS reads instead of 8 trips through the texture addresser
        n0a = bv.topPiece(t.left, 0); n0b = bv.topPiece(t.left, 1); n1a = bv.topPiece(t.left, 2); n1b = bv.topPiece(t.left, 3);
        n2a = bv.topPiece(t.left, 4); n2b = bv.topPiece(t.left, 5); n3a = bv.topPiece(t.left, 6); n3b = bv.topPiece(t.left, 7);
      } else {
        n0a = bv.node(t.left, 0); n0b = bv.node(t.left, 1); n1a = bv.node(t.left, 2); n1b = bv.node(t.left, 3);
        n2a = bv.node(t.left, 4); n2b = bv.node(t.left, 5); n3a = bv.node(t.left, 6); n3b = bv.node(t.left, 7);
      }
      if (COUNT) { cnt.quads++; if (!(TOPCACHE && (t.left & HK_TOP_FLAG)) && !bv.nodeInRange(t.left)) cnt.oob++; }
      int c0 = as_int(n0a.w), c1 = as_int(n1a.w), c2 = as_int(n2a.w), c3 = as_int(n3a.w);
#ifdef HK_HOST_EMU
      const bool v0 = !((uint32_t(c0) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n0b.w)) == HYDRA_BVH_INVALID));
      const bool v1 = !((uint32_t(c1) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n1b.w)) == HYDRA_BVH_INVALID));
      const bool v2 = !((uint32_t(c2) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n2b.w)) == HYDRA_BVH_INVALID));
      const bool v3 = !((uint32_t(c3) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n3b.w)) == HYDRA_BVH_INVALID));
#else
      // the device copy of the node array has the boxes of invalid children (both link words 0xFFFFFFFF, ctrace.h:889-892)
      // overwritten with NaN by k_prepare_bvh at upload: every comparison below is then false for them, which is what the
      // reference's IsValidNode term achieves -- 12 VALU instructions less per quad in a loop that is VALU-issue bound
      const bool v0 = true, v1 = true, v2 = true, v3 = true;
#endif
#ifdef HK_EXP_SETPRIO   /* timing experiment: the wave that has its quad runs its box tests ahead of the others */
      __builtin_amdgcn_s_setprio(HK_EXP_SETPRIO);
#endif
      const float2 t0 = RayBox(t.pos, t.inv, n0a, n0b), t1 = RayBox(t.pos, t.inv, n1a, n1b);
      const float2 t2 = RayBox(t.pos, t.inv, n2a, n2b), t3 = RayBox(t.pos, t.inv, n3a, n3b);
      float k0 = ((t0.x <= t0.y) && (t0.y >= t_rayMin) && (t0.x <= t.hit.t) && v0) ? t0.x : HK_MAXFLOAT;
      float k1 = ((t1.x <= t1.y) && (t1.y >= t_rayMin) && (t1.x <= t.hit.t) && v1) ? t1.x : HK_MAXFLOAT;
      float k2 = ((t2.x <= t2.y) && (t2.y >= t_rayMin) && (t2.x <= t.hit.t) && v2) ? t2.x : HK_MAXFLOAT;
      float k3 = ((t3.x <= t3.y) && (t3.y >= t_rayMin) && (t3.x <= t.hit.t) && v3) ? t3.x : HK_MAXFLOAT;
#define HK_CSWAP(ka, kb, ca, cb) { const bool sw = (kb < ka); const float tk = sw ? kb : ka; kb = sw ? ka : kb; ka = tk; const int tc = sw ? cb : ca; cb = sw ? ca : cb; ca = tc; }
      HK_CSWAP(k0, k1, c0, c1) HK_CSWAP(k2, k3, c2, c3) HK_CSWAP(k0, k2, c0, c2) HK_CSWAP(k1, k3, c1, c3) HK_CSWAP(k1, k2, c1, c2)
#ifdef HK_EXP_EXTRA_SORT   /* timing experiment: the network again on sorted keys changes nothing but costs its ~25 VALU instructions */
      for (int rep = 0; rep < HK_EXP_EXTRA_SORT; rep++) {
        asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3));
        HK_CSWAP(k0, k1, c0, c1) HK_CSWAP(k2, k3, c2, c3) HK_CSWAP(k0, k2, c0, c2) HK_CSWAP(k1, k3, c1, c3) HK_CSWAP(k1, k2, c1, c2)
      }
#endif
#undef HK_CSWAP
      // keys are sorted and misses carry MAXFLOAT, so the children hit are a prefix: nothing is pushed unless k1 is a hit
      // (same pushes, in the same order, as the three separate tests of ctrace.h:962-975)
      if (k1 < HK_MAXFLOAT) {
        const bool stackHaveSpace = (t.top < HK_STACK_SIZE);
        if (k3 < HK_MAXFLOAT && stackHaveSpace) { stack.put(t.top, c3); t.top++; }
        if (k2 < HK_MAXFLOAT && stackHaveSpace) { stack.put(t.top, c2); t.top++; }
        if (stackHaveSpace) { stack.put(t.top, c1); t.top++; }
      }
      if (k0 < HK_MAXFLOAT) t.left = c0;
      else if (t.top >= 0) { t.top--; t.left = stack.get(t.top); }
      t.searching = !(t.left & int(HYDRA_BVH_LEAF)) && (t.top >= 0);
      t.left = t.left & 0x7fffffff;
#ifdef HK_EXP_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      if (haveInst && t.top < t.instTop && t.instDeep == 1) {
        t.pos = t.opos; t.dir = t.odir; t.inv = t.oinv; t.instDeep = 0;   // = SafeInverse(t.odir), same bits (ctrace.h:1000-1006)
      }
    }
    if (!haveInst) {
      if (t.top >= 0) {
        t.hit = IntersectLeaf<ANYHIT, COUNT, TOPTRIS, false>(t.pos, t.dir, t.left, t_rayMin, t.hit, bv, 0, false, cnt);   // BVH4Traverse has no alpha form (Common.cpp:146)
        if (ANYHIT && t.hit.primId != -1) { t.top = -1; return true; }
      }
      t.top--;
      t.left = stack.get(t.top);
    } else if (t.top >= 0 && t.instDeep == 1) {
      t.hit = IntersectLeaf<ANYHIT, COUNT, TOPTRIS, ALPHA>(t.pos, t.dir, t.left, t_rayMin, t.hit, bv, t.instId, true, cnt);
      if (ANYHIT && t.hit.primId != -1) { t.top = -1; return true; }
      t.top--;
      t.left = stack.get(t.top);
    } else if (t.top >= 0 && t.instDeep == 0) {
      t.instDeep = 1;
      t.opos = t.pos; t.odir = t.dir; t.oinv = t.inv;
      const int nextOffset = as_int(bv.node(t.left, 0).w);
      m44 matrix;
      matrix.c[0] = bv.node(t.left, 2); matrix.c[1] = bv.node(t.left, 3); matrix.c[2] = bv.node(t.left, 4); matrix.c[3] = bv.node(t.left, 5);
      t.instId = as_int(bv.node(t.left, 6).x);
      if (COUNT) { cnt.insts++; if (!bv.nodeInRange(t.left)) cnt.oob++; }
      t.pos = mul4x3(matrix, t.pos);
      t.dir = mul3x3(matrix, t.dir);   // stays un-normalised so t keeps world units
      t.inv = SafeInverse(t.dir);
      t.instTop = t.top;
      t.left = nextOffset;
    }
    t.searching = !(t.left & int(HYDRA_BVH_LEAF));
    t.left = t.left & 0x7fffffff;
    if (haveInst && t.top < t.instTop && t.instDeep == 1) {
      t.pos = t.opos; t.dir = t.odir; t.inv = t.oinv; t.instDeep = 0;
    }
    if (minActive > 0 && t.top >= 0 && HK_WAVE_ACTIVE_LANES() < minActive) return false;   // let the wave refill
  }
  return true;
}

// Scheduling by wave vote.  In the loop nest above a wave repeats the quad step until its LAST lane has reached a leaf, then runs the
// leaf code for whichever lanes hold a triangle leaf and again for those at an instance leaf: with incoherent rays a quarter of the
// lanes of a VALU instruction do work (measured: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = 0.23 for bounce >= 1 on test_224,
// profiles/r03).  Here every turn of ONE loop runs the step that the most lanes are waiting for -- quad, triangle leaf or instance
// entry -- and the others wait that turn out: no lane idles through a phase that only a few stragglers need, and each phase's
// instructions run with at least a third (typically half or more) of the live lanes.  A lane's own sequence of steps is untouched, so hits, `t` bits and
// visit counters are those of trav_run (tests: the persistent kernels against the static ones, both against the oracle).
// Wave-uniform: every lane of the wave (or of the active set) calls it, `busy` says whether the lane holds a ray.  Leaves when no lane
// is traversing any more, or when fewer than minActive are (minActive <= 0: only when none is) so that the wave can refill.  Refilling a
// third of the lanes at once beats refilling each lane as it finishes (profiles/r03/vote_minactive_*.log): rays that start together want
// the same steps at the same time.
// A finished ray has t.top < 0.  wq / wt / wi: votes count wq x quad lanes against wt x triangle lanes against wi x instance lanes.
#ifdef HK_HOST_EMU
#define HK_BALLOT(x) ((x) ? 1ull : 0ull)
#define HK_POPC(m) __builtin_popcountll(m)
#else
#define HK_BALLOT(x) __ballot(x)
#define HK_POPC(m) __popcll(m)
#endif
// UNORD (any-hit, non-counting kernels only; option shadow_unordered): a quad's children are taken in stored order instead of near to far; closest-hit rays and the
// counting kernels always walk near to far, as BVH4InstTraverse / BVH4InstTraverseShadow do
#ifndef HK_SPEC_LEAF
#define HK_SPEC_LEAF 0
#endif
template <bool ANYHIT, bool COUNT, bool TOPCACHE = false, class STACK = HkStack, bool TOPTRIS = false, bool ALPHA = false, bool UNORD = false>
HK_DEV void trav_run_vote(TravState& t, const bool busy, const BvhView& bv, const bool haveInst,
                          const float t_rayMin, STACK& stack, TravCounters& cnt, const int minActive, const int wq, const int wt, const int wi) {
  static_assert(!UNORD || (ANYHIT && !COUNT), "only an any-hit query is independent of the order");
  // SPEC (experiment, closest-hit production kernels): a lane that stands at a triangle leaf while the wave does inner nodes puts the leaf aside and walks on; the leaf is
  // tested at the wave's next triangle turn, before any leaf found later.  The leaves a ray tests, and their order, are a superset of the reference's in the same order
  // (what is walked in between was not yet pruned by the waiting leaf's hit: nothing in it can be nearer or equal), so hits and their bits do not change.
  constexpr bool SPEC = (HK_SPEC_LEAF != 0) && !ANYHIT && !COUNT && !TOPTRIS && !ALPHA;
  if (SPEC) {
    while (true) {
      const bool walking = busy && t.top >= 0 && !t.pendExit;
      const bool alive = busy && (t.top >= 0 || t.pend >= 0);
      const bool atInst = haveInst && t.instDeep == 0;
      const bool wantQuad = walking && t.searching, wantInst = walking && !t.searching && atInst && t.pend < 0;
      const bool atLeaf = walking && !t.searching && !atInst;
      const bool wantTri = busy && (atLeaf || t.pend >= 0);
      const int nq = HK_POPC(HK_BALLOT(wantQuad)), nt = HK_POPC(HK_BALLOT(wantTri)), ni = haveInst ? HK_POPC(HK_BALLOT(wantInst)) : 0;
      const int nAlive = HK_POPC(HK_BALLOT(alive));
      if (nAlive == 0 || nAlive < minActive) return;
      const int vq = nq * wq, vt = nt * wt, vi = ni * wi;
      if (vq >= vt && vq >= vi) {
        if (wantQuad) trav_quad_step<COUNT, TOPCACHE, STACK, true>(t, bv, haveInst, t_rayMin, stack, cnt, t.pend >= 0);
        else if (atLeaf && t.pend < 0 && (HK_SPEC_LEAF != 2 || t.hit.primId != -1)) {      // put the leaf aside, take the next node off the stack (what trav_tri_step does after its tests); variant 2: only rays that already have a hit to prune with
          t.pend = t.left;
          t.top--;
          t.left = stack.get(t.top);
          t.searching = !(t.left & int(HYDRA_BVH_LEAF));
          t.left = t.left & 0x7fffffff;
          if (haveInst && t.top < t.instTop && t.instDeep == 1) t.pendExit = true;
        }
      } else if (vt >= vi) {
        if (t.pend >= 0 && busy) {
          t.hit = haveInst ? IntersectLeaf<false, COUNT, false, false>(t.pos, t.dir, t.pend, t_rayMin, t.hit, bv, t.instId, true, cnt)
                           : IntersectLeaf<false, COUNT, false, false>(t.pos, t.dir, t.pend, t_rayMin, t.hit, bv, 0, false, cnt);
          t.pend = -1;
          if (t.pendExit) { t.pos = t.opos; t.dir = t.odir; t.inv = t.oinv; t.instDeep = 0; t.pendExit = false; }
        } else if (atLeaf) (void)trav_tri_step<false, COUNT, STACK, false, false>(t, bv, haveInst, t_rayMin, stack, cnt);
      } else { if (wantInst) trav_inst_step<COUNT>(t, bv, cnt); }
    }
  }
  while (true) {
    const bool alive = busy && t.top >= 0;
    const bool atInst = haveInst && t.instDeep == 0;
    const bool wantQuad = alive && t.searching, wantInst = alive && !t.searching && atInst, wantTri = alive && !t.searching && !atInst;
    const int nq = HK_POPC(HK_BALLOT(wantQuad)), nt = HK_POPC(HK_BALLOT(wantTri)), ni = haveInst ? HK_POPC(HK_BALLOT(wantInst)) : 0;
    const int nAlive = nq + nt + ni;
    if (nAlive == 0 || nAlive < minActive) return;
    const int vq = nq * wq, vt = nt * wt, vi = ni * wi;
    if (vq >= vt && vq >= vi) { if (wantQuad) trav_quad_step<COUNT, TOPCACHE, STACK, !UNORD>(t, bv, haveInst, t_rayMin, stack, cnt); }
    else if (vt >= vi) { if (wantTri) (void)trav_tri_step<ANYHIT, COUNT, STACK, TOPTRIS, ALPHA>(t, bv, haveInst, t_rayMin, stack, cnt); }
    else { if (wantInst) trav_inst_step<COUNT>(t, bv, cnt); }
  }
}

template <bool ANYHIT, bool COUNT, bool ALPHA = false>
HK_DEV HydraLiteHit hk_traverse(const BvhView& bv, const bool haveInst,
                                f3 ray_pos, f3 ray_dir, const float t_rayMin, HydraLiteHit hit, HkStack& stack, TravCounters& cnt) {
  TravState t;
  trav_init(t, ray_pos, ray_dir, hit);
  (void)trav_run<ANYHIT, COUNT, false, HkStack, false, ALPHA>(t, bv, haveInst, t_rayMin, stack, cnt, 0);
  return t.hit;
}

HK_DEV HydraLiteHit hk_miss_hit() {   // Make_Lite_Hit(MAXFLOAT, -1), hydra_drv/cglobals.h:1256-1266
  HydraLiteHit h;
  h.t = HK_MAXFLOAT; h.primId = -1; h.instId = -1; h.geomId = int(0xC0000000u);
  return h;
}
HK_DEV bool HitSome(const HydraLiteHit& h) { return (h.primId != -1) && isfinite(h.t); }
