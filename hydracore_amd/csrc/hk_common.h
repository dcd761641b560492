// hk_common.h -- device-side vector helpers and scene view for the gfx950 kernels.
// Arithmetic is plain IEEE float32 in the order written; the library is built with -ffp-contract=off so that
// integer/byte-level results (hit ids, flags) match the CPU path bit for bit and floats stay within a few ulp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <string.h>
#include "../../include/hydra_layouts.h"

// HK_HOST_EMU: the same device functions compiled for the host (tests/emu/, AddressSanitizer/UBSan runs on CPU only;
// GPU sanitizers are not available on the pool).  Never defined in the product build.
#ifdef HK_HOST_EMU
#include <math.h>
#define HK_DEV static inline
#define HK_DEV_CALL static
#define HK_DEV_MEMBER inline
#define HK_WAVE_ACTIVE_LANES() 64
#else
#define HK_DEV __device__ __forceinline__
#ifdef HK_INLINE_TEXTURE_FETCH
#define HK_DEV_CALL __device__ __forceinline__
#else
#define HK_DEV_CALL __device__ __noinline__   /* a real call: the body appears once in a kernel instead of once per call site */
#endif
#define HK_DEV_MEMBER __device__ __forceinline__
#define HK_WAVE_ACTIVE_LANES() __popcll(__ballot(1))
#endif

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct m44 { float4 c[4]; };   // four columns (reference: make_float4x4, hydra_drv/cglobals.h:792-800)

HK_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
HK_DEV f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
HK_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
HK_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
HK_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
HK_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
HK_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HK_DEV f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HK_DEV float length(f3 a) { return sqrtf(dot(a, a)); }
HK_DEV f3 normalize(f3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
HK_DEV float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
HK_DEV f3 clamp3(f3 v, float a, float b) { return mk3(clampf(v.x, a, b), clampf(v.y, a, b), clampf(v.z, a, b)); }
HK_DEV f3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }
HK_DEV float4 mk4(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
#ifdef HK_HOST_EMU
HK_DEV int   as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
HK_DEV float as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
#else
HK_DEV int   as_int(float f) { return __float_as_int(f); }
HK_DEV float as_float(int i) { return __int_as_float(i); }
#endif
#ifdef HK_HOST_EMU
HK_DEV void hk_atomic_add(float* p, float v) { *p += v; }   // the emulation runs one lane at a time
#else
HK_DEV void hk_atomic_add(float* p, float v) { atomicAdd(p, v); }
#endif
HK_DEV bool  finite3(f3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }

// reference: mul4x3 / mul3x3, hydra_drv/cglobals.h:288-304; mul4x4x4 :828-836
HK_DEV f3 mul4x3(const m44& m, f3 v) {
  return mk3(v.x * m.c[0].x + v.y * m.c[1].x + v.z * m.c[2].x + m.c[3].x,
             v.x * m.c[0].y + v.y * m.c[1].y + v.z * m.c[2].y + m.c[3].y,
             v.x * m.c[0].z + v.y * m.c[1].z + v.z * m.c[2].z + m.c[3].z);
}
HK_DEV f3 mul3x3(const m44& m, f3 v) {
  return mk3(v.x * m.c[0].x + v.y * m.c[1].x + v.z * m.c[2].x,
             v.x * m.c[0].y + v.y * m.c[1].y + v.z * m.c[2].y,
             v.x * m.c[0].z + v.y * m.c[1].z + v.z * m.c[2].z);
}
HK_DEV float4 mul4x4x4(const m44& m, float4 v) {
  float4 r;
  r.x = v.x * m.c[0].x + v.y * m.c[1].x + v.z * m.c[2].x + v.w * m.c[3].x;
  r.y = v.x * m.c[0].y + v.y * m.c[1].y + v.z * m.c[2].y + v.w * m.c[3].y;
  r.z = v.x * m.c[0].z + v.y * m.c[1].z + v.z * m.c[2].z + v.w * m.c[3].z;
  r.w = v.x * m.c[0].w + v.y * m.c[1].w + v.z * m.c[2].w + v.w * m.c[3].w;
  return r;
}
HK_DEV m44 load_m44(const float4* p) { m44 m; m.c[0] = p[0]; m.c[1] = p[1]; m.c[2] = p[2]; m.c[3] = p[3]; return m; }
HK_DEV m44 transpose44(const m44& a) {
  m44 r;
  r.c[0] = make_float4(a.c[0].x, a.c[1].x, a.c[2].x, a.c[3].x);
  r.c[1] = make_float4(a.c[0].y, a.c[1].y, a.c[2].y, a.c[3].y);
  r.c[2] = make_float4(a.c[0].z, a.c[1].z, a.c[2].z, a.c[3].z);
  r.c[3] = make_float4(a.c[0].w, a.c[1].w, a.c[2].w, a.c[3].w);
  return r;
}
// object->world matrix back from the stored world->object one (kernel_EvalSurface step 4, PT_Loop.cpp:57):
// instance matrices are affine, so invert the 3x3 block by cofactors and rotate the translation.
HK_DEV m44 inverse_affine(const m44& m) {
  const float a00 = m.c[0].x, a10 = m.c[0].y, a20 = m.c[0].z;
  const float a01 = m.c[1].x, a11 = m.c[1].y, a21 = m.c[1].z;
  const float a02 = m.c[2].x, a12 = m.c[2].y, a22 = m.c[2].z;
  const float c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const float det = a00 * c00 + a01 * c01 + a02 * c02;
  const float id = 1.0f / det;
  m44 r;
  r.c[0] = make_float4(c00 * id, c01 * id, c02 * id, 0.0f);
  r.c[1] = make_float4((a02 * a21 - a01 * a22) * id, (a00 * a22 - a02 * a20) * id, (a01 * a20 - a00 * a21) * id, 0.0f);
  r.c[2] = make_float4((a01 * a12 - a02 * a11) * id, (a02 * a10 - a00 * a12) * id, (a00 * a11 - a01 * a10) * id, 0.0f);
  const float tx = m.c[3].x, ty = m.c[3].y, tz = m.c[3].z;
  r.c[3] = make_float4(-(r.c[0].x * tx + r.c[1].x * ty + r.c[2].x * tz),
                       -(r.c[0].y * tx + r.c[1].y * ty + r.c[2].y * tz),
                       -(r.c[0].z * tx + r.c[1].z * ty + r.c[2].z * tz), 1.0f);
  return r;
}

// ---- device view of the scene: raw pointers into the uploaded blobs (layouts: include/hydra_layouts.h)
struct SceneDev {
  const int*    globals;       // [EngineGlobals | tables | lights]
  const float4* matStorage;
  const int4*   texStorage;
  const float4* geomStorage;
  // private re-layout of the geometry arena, built once per scene on the device (k_geom_fill): one 128-byte, line-aligned
  // record per triangle = A.pos|u, B.pos|u, C.pos|u, A.norm|v, B.norm|v, C.norm|v, (matId, shadow offset, -, -), pad;
  // tangents in a second array (3 float4 per triangle).  triBase[geomId] = first record of that mesh.  Same bits as the arena.
  const float4* triRec; const float4* triTan; const int* triBase;
  const float4* pdfStorage;
  const float4* bvh;           // tree 0: 2 float4 per node, 8 per quad
  unsigned      bvhBytes;      // size of the node array (< 4 GiB: the traversal kernels address it as a raw buffer)
  const float4* bvhTop;        // second node copy whose links to the hottest quads are tagged (hk_trace.h, HK_TOP_FLAG); nullptr = none
  const int*    topQuads;      // quad index of each cached slot, slot 0 = the root
  int           topCount;      // 0..HK_TOP_QUADS
  const int*    topTriF4;      // float4 index (in tris) of every float4 of the triangle pool kept in LDS (3 per triangle)
  int           topTriCount;   // triangles in that pool, 0..HK_TOP_TRIS
  int           leafEnc;       // 1: triangle-leaf links of the device node copy carry the triangle count (hk_trace.h, HK_LEAF_COUNT_SHIFT)
  const uint2*  alpha;         // the tree's alpha table (ctrace.h:380-391), nullptr = no alpha-tested triangles
  const float4* tris;          // tree 0 triangle lists
  unsigned      trisBytes;
  int           haveInst;
  const float4* instMatrices;  // 4 float4 per instance (world -> object)
  const int*    instLightInstId;
  int           instNum;
  const int*    remapLists; int remapListsSize;
  const int*    remapTable; int remapTableSize;
  const int*    remapInst;  int remapInstSize;
  // sRGBToLinear(byte / 255) for the 256 byte values, filled once per context ON THE DEVICE by the same device function the
  // per-tap decode would call (k_fill_srgb_lut), so a look-up has the bits of the computation it replaces; nullptr = compute
  const float*  srgbLut;
  // The small, hot tables every shaded path walks through -- material arena, material-id and texture-id tables, lights -- by
  // the pointers the shading code uses.  The host points them at the uploaded blobs; k_bounce re-points them at its block's
  // LDS copy when the scene's tables fit (hydra_hip.hip, SceneStage): a dependent table read then costs an LDS access instead
  // of a trip to L2, which is what the kernel spends its time waiting for (profiles/r02: 59 % of wave cycles on memory waits).
  const float*  matBase;       // material arena, float-addressed (node = HM_NODE_FLOATS floats)
  const int*    matTable;      // material id -> arena offset in float4 units (cfetch.h:192-197)
  const float*  lightsBase;    // PlainLight[lightsNum], HL_FLOATS floats each (clight.h:1739-1749)
  const int*    texTable;      // texture id -> texture arena offset in int4 units (cfetch.h:141-145)
  const int4*   texAuxStorage; // the second texture arena (normal maps, RenderDriverRTE_AuxTextures.cpp:195-216)
  const int*    texAuxTable;   // aux texture id -> offset in it (textureAuxHeaderOffset, cfetch.h:147-151)
  // the scalar part of EngineGlobals (matrices, variables, table offsets and sizes: words 0..HK_HDR_WORDS-1 of `globals`) and the
  // reverse light-selection table (clight.h:1774-1793), by their own pointers for the same reason: every shaded path reads them
  const int*    hdr;
  const float*  lselRev;
  // Procedural textures (hk_proctex_rt.h): what the run-time compiled kernel k_proctex left for every path of the bounce -- the ids of
  // the procedural textures of the hit material (plane k of ptlIds: slot + k * ptlStride, the list ends at HYDRA_INVALID_TEXTURE or after
  // ptlMax planes) and their colours as four halfs each (ptlVals, same addressing), the precision the reference's layer hands them
  // on with (WriteProcTextureList, cglobals.h:2327-2359).  ptlSlot is PER LANE: the path slot whose list the texture fetches of this lane
  // consult (sample2DExt), -1 = none.  Only kernels instantiated with HK_FEAT_PROCTEX set it; everywhere else it is the constant -1 and
  // the look-up folds away.
  const int*    ptlIds;
  const uint2*  ptlVals;
  int           ptlStride, ptlMax;
  int           ptlSlot;
};
#ifndef HK_HOST_EMU
// Segmented path queues.  One global "next free slot" word saturates at ~88 returning atomics per microsecond on MI355X
// (MI355X_MICROARCH.md, dequeue row); with one atomic per wave that alone cost k_hit ~1 ms per sample at 1080p.  The
// path arrays are therefore split into `nseg` segments of `cap` slots, every thread block works on exactly one segment
// (block b -> segment b % nseg) and appends survivors to the SAME segment of the next queue through that segment's own
// counter (counters sit HK_CSTRIDE words = 128 B apart).  A segment can never grow, so cap = its initial share is a hard
// bound and memory use does not change; results are independent of the segmentation because accumulation is keyed by pixel.
#define HK_CSTRIDE 32
#define HK_MAX_SEG 64
#define HK_CROW (HK_MAX_SEG * HK_CSTRIDE)   // words per counter row: one row per bounce, [segment] inside
struct SegQ {
  const uint32_t* counts;   // counts[seg * HK_CSTRIDE]; nullptr => countImm items in one segment
  int countImm, nseg, cap;
};
struct SegIter { int seg, base, count, first, step; };
HK_DEV SegIter segq_iter(const SegQ& q) {
  SegIter it;
  const int bps = int(gridDim.x) / q.nseg;            // blocks per segment (grid is a multiple of nseg)
  it.seg = int(blockIdx.x) % q.nseg;
  const int bis = int(blockIdx.x) / q.nseg;
  it.count = (bis < bps) ? (q.counts ? int(q.counts[it.seg * HK_CSTRIDE]) : q.countImm) : 0;
  it.base = it.seg * q.cap;
  it.first = bis * int(blockDim.x) + int(threadIdx.x);
  it.step = (bps > 0 ? bps : 1) * int(blockDim.x);
  return it;
}
#endif
#define HK_HDR_WORDS 240   /* HG_MPROJ .. HG_DUMMY1, include/hydra_layouts.h */

HK_DEV const float* g_varsF(const SceneDev& s) { return reinterpret_cast<const float*>(s.hdr + HG_VARS_F); }
HK_DEV const int*   g_varsI(const SceneDev& s) { return s.hdr + HG_VARS_I; }

#define HK_GEPSILON  5e-6f
#define HK_DEPSILON  1e-20f
#define HK_DEPSILON2 1e-30f
#define HK_PI 3.14159265358979323846f   /* the pinned build of the reference compiles with -cl-single-precision-constant */
#define HK_INV_PI    0.31830988618379067154f
#define HK_INV_TWOPI 0.15915494309189533577f
#define HK_TWOPI     6.28318530717958647692f
#define HK_MAXFLOAT  FLT_MAX   /* ctrace.h:665-667: MAXFLOAT is glibc's FLT_MAX on the CPU path */
