// hk_proctex_rt.h -- the run-time half of procedural textures.  NOT compiled into the library as code: the build embeds this file
// (with hydra_layouts.h, hk_common.h, hk_trace.h and hk_shading.h in front of it) as text, and hydra_hip_proctex_compile hands that
// text plus the scene's own functions to hiprtc.  What the reference does with OpenCL at scene load
// (RenderDriverRTE::BeginTexturesUpdate / UpdateImageProc / EndTexturesUpdate, RenderDriverRTE_ProcTex.cpp:446-629, splice the scene's
// data/proctex_*.c into shaders/texproc.cl and rebuild the program; GPUOCLLayer::RecompileProcTexShaders, GPUOCLLayer.cpp:788-810) happens
// here with HIP: the two spliced regions of that text -- the user functions and the generated calls -- are placed into the frame below.
//
// The frame carries two marker lines the host side replaces (hydra_proctex.hip): one where the user functions go, one where the generated calls go.
//
// The user functions are written in the C dialect HydraAPI generates for OpenCL: float2/3/4 with component access and arithmetic,
// make_float3, texture2D(sampler2D, float2, flags), readAttr_*(sHit), the tail arguments (in_texStorage1, in_globals, hr_viewVectorHack).
// They are compiled inside namespace hk_user, where those names mean clang's native vector types and the helpers below, so that nothing of
// it collides with HIP's own float3 or with this library's f3.

namespace hk_user {

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int   int2   __attribute__((ext_vector_type(2)));
typedef int   int3   __attribute__((ext_vector_type(3)));
typedef int   int4   __attribute__((ext_vector_type(4)));
typedef unsigned int uint;
typedef unsigned int uint2 __attribute__((ext_vector_type(2)));
typedef unsigned int uint3 __attribute__((ext_vector_type(3)));
typedef unsigned int uint4 __attribute__((ext_vector_type(4)));
typedef int sampler2D;
typedef ::SceneDev EngineGlobals;   // what `in_globals` points at: the tail argument is only ever handed on to texture2D

#define __global
#define __private
#define __constant const
#define restrict __restrict__
#ifdef HK_HOST_EMU   /* the same frame built for the host (tests/proctex_host.py: the scene's functions feeding the CPU oracle); never in the product */
#define HKU static inline
#define __float_as_int(x) ::as_int(x)
#define __int_as_float(x) ::as_float(x)
#define __float_as_uint(x) ((unsigned)::as_int(x))
#else
#define HKU __device__ __forceinline__
#endif

HKU float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
HKU float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
HKU float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
HKU int2 make_int2(int x, int y) { int2 r = {x, y}; return r; }
HKU int3 make_int3(int x, int y, int z) { int3 r = {x, y, z}; return r; }
HKU int4 make_int4(int x, int y, int z, int w) { int4 r = {x, y, z, w}; return r; }
HKU float3 to_float3(float4 v) { return make_float3(v.x, v.y, v.z); }                        // cglobals.h:158
HKU float4 to_float4(float3 v, float w) { return make_float4(v.x, v.y, v.z, w); }            // cglobals.h:159
HKU int   as_int(float x) { return __float_as_int(x); }
HKU float as_float(int x) { return __int_as_float(x); }
HKU uint  as_uint(float x) { return __float_as_uint(x); }

// the OpenCL built-ins HydraAPI's procedural textures use, over the device math library HIP and the OpenCL compiler share (ocml)
#define HKU_UNARY(name, expr) \
  HKU float  name(float x)  { return expr; } \
  HKU float2 name(float2 v) { return make_float2(name(v.x), name(v.y)); } \
  HKU float3 name(float3 v) { return make_float3(name(v.x), name(v.y), name(v.z)); } \
  HKU float4 name(float4 v) { return make_float4(name(v.x), name(v.y), name(v.z), name(v.w)); }
HKU_UNARY(fabs, ::fabsf(x))   HKU_UNARY(floor, ::floorf(x)) HKU_UNARY(ceil, ::ceilf(x))   HKU_UNARY(sqrt, ::sqrtf(x))  HKU_UNARY(rsqrt, 1.0f / ::sqrtf(x))
HKU_UNARY(sin, ::sinf(x))     HKU_UNARY(cos, ::cosf(x))     HKU_UNARY(tan, ::tanf(x))     HKU_UNARY(asin, ::asinf(x))  HKU_UNARY(acos, ::acosf(x))
HKU_UNARY(atan, ::atanf(x))   HKU_UNARY(exp, ::expf(x))     HKU_UNARY(exp2, ::exp2f(x))   HKU_UNARY(log, ::logf(x))    HKU_UNARY(log2, ::log2f(x))
HKU_UNARY(trunc, ::truncf(x)) HKU_UNARY(round, ::roundf(x)) HKU_UNARY(sign, (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f))
HKU_UNARY(fract, ::fminf(x - ::floorf(x), 0x1.fffffep-1f))
#define HKU_BINARY(name, expr) \
  HKU float  name(float x, float y)   { return expr; } \
  HKU float2 name(float2 a, float2 b) { return make_float2(name(a.x, b.x), name(a.y, b.y)); } \
  HKU float3 name(float3 a, float3 b) { return make_float3(name(a.x, b.x), name(a.y, b.y), name(a.z, b.z)); } \
  HKU float4 name(float4 a, float4 b) { return make_float4(name(a.x, b.x), name(a.y, b.y), name(a.z, b.z), name(a.w, b.w)); } \
  HKU float2 name(float2 a, float b)  { return make_float2(name(a.x, b), name(a.y, b)); } \
  HKU float3 name(float3 a, float b)  { return make_float3(name(a.x, b), name(a.y, b), name(a.z, b)); } \
  HKU float4 name(float4 a, float b)  { return make_float4(name(a.x, b), name(a.y, b), name(a.z, b), name(a.w, b)); }
HKU_BINARY(fmin, ::fminf(x, y)) HKU_BINARY(fmax, ::fmaxf(x, y)) HKU_BINARY(min, ::fminf(x, y)) HKU_BINARY(max, ::fmaxf(x, y))
HKU_BINARY(pow, ::powf(x, y))   HKU_BINARY(fmod, ::fmodf(x, y)) HKU_BINARY(atan2, ::atan2f(x, y)) HKU_BINARY(step, (y < x) ? 0.0f : 1.0f)
HKU int min(int a, int b) { return a < b ? a : b; }
HKU int max(int a, int b) { return a > b ? a : b; }
HKU int abs(int a) { return a < 0 ? -a : a; }
HKU int clamp(int x, int a, int b) { return min(max(x, a), b); }
HKU float  clamp(float x, float a, float b)   { return ::fminf(::fmaxf(x, a), b); }
HKU float2 clamp(float2 v, float a, float b)  { return make_float2(clamp(v.x, a, b), clamp(v.y, a, b)); }
HKU float3 clamp(float3 v, float a, float b)  { return make_float3(clamp(v.x, a, b), clamp(v.y, a, b), clamp(v.z, a, b)); }
HKU float4 clamp(float4 v, float a, float b)  { return make_float4(clamp(v.x, a, b), clamp(v.y, a, b), clamp(v.z, a, b), clamp(v.w, a, b)); }
HKU float  mix(float x, float y, float a)     { return x + (y - x) * a; }
HKU float2 mix(float2 x, float2 y, float a)   { return x + (y - x) * a; }
HKU float3 mix(float3 x, float3 y, float a)   { return x + (y - x) * a; }
HKU float4 mix(float4 x, float4 y, float a)   { return x + (y - x) * a; }
HKU float3 mix(float3 x, float3 y, float3 a)  { return x + (y - x) * a; }
HKU float4 mix(float4 x, float4 y, float4 a)  { return x + (y - x) * a; }
HKU float  smoothstep(float e0, float e1, float x) { const float t = clamp((x - e0) / (e1 - e0), 0.0f, 1.0f); return t * t * (3.0f - 2.0f * t); }
HKU float  dot(float2 a, float2 b) { return a.x * b.x + a.y * b.y; }
HKU float  dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HKU float  dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
HKU float3 cross(float3 a, float3 b) { return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HKU float  length(float2 a) { return ::sqrtf(dot(a, a)); }
HKU float  length(float3 a) { return ::sqrtf(dot(a, a)); }
HKU float  length(float4 a) { return ::sqrtf(dot(a, a)); }
HKU float  distance(float2 a, float2 b) { return length(a - b); }
HKU float  distance(float3 a, float3 b) { return length(a - b); }
HKU float2 normalize(float2 a) { return a * (1.0f / length(a)); }
HKU float3 normalize(float3 a) { return a * (1.0f / length(a)); }
HKU float4 normalize(float4 a) { return a * (1.0f / length(a)); }
HKU float3 reflect(float3 i, float3 n) { return i - n * (2.0f * dot(n, i)); }

// what the reference's texproc.cl declares around the user functions (shaders/texproc.cl:5-63)
#define NORMAL_IN_TANGENT_SPACE 16.0f
#define NORMAL_IN_WORLD_SPACE   32.0f
#define TEX_POINT_SAM      1    /* the sampler flags a user function may pass to texture2D (cglobals.h:18-24) */
#define TEX_ALPHASRC_W     2
#define TEX_CLAMP_U        4
#define TEX_CLAMP_V        8
#define TEX_DATA_HDR       64
HKU float4 StoreNormal(float3 res, float a_mode) {
  const float3 res2 = normalize(res);
  const float zeroPointFive = 0.49995f;
  return make_float4(clamp(0.5f * res2.x + zeroPointFive, 0.0f, 1.0f), clamp(0.5f * res2.y + zeroPointFive, 0.0f, 1.0f), clamp(1.0f * res2.z + 0.0f, 0.0f, 1.0f), 0.0f);
}
// texture2D(name, uv, flags): the stored texture `name` at uv, bilinear, sRGB-decoded unless the flags say HDR data (InternalFetch, texproc.cl:20-31)
HKU float4 InternalFetch(int a_texId, const float2 texCoord, const int a_flags, const float4* in_texStorage1, const EngineGlobals* in_globals) {
  if (a_texId < 0 || a_texId >= in_globals->hdr[HG_TEX_TABLE_SIZE]) return make_float4(1, 1, 1, 1);
  const int offset = in_globals->texTable[a_texId];
  if (offset < 0) return make_float4(1, 1, 1, 1);   // an id without a stored texture (another procedural texture, a missing chunk): the reference reads whatever lies before the arena here
  const ::float4 c = ::read_imagef_sw4(reinterpret_cast<const ::int4*>(in_texStorage1) + offset, ::mk2(texCoord.x, texCoord.y), a_flags, (a_flags & HTEX_DATA_HDR) == 0, in_globals->srgbLut);
  return make_float4(c.x, c.y, c.z, c.w);
}
#define texture2D(texName, texCoord, flags) InternalFetch((texName), (texCoord), (flags), in_texStorage1, in_globals)

typedef struct SurfaceInfoT { float3 wp, lp, n, tg, bn; float2 tc0; float ao, ao2; } SurfaceInfo;
#define readAttr_WorldPos(sHit)  (sHit->wp)
#define readAttr_LocalPos(sHit)  (sHit->lp)
#define readAttr_ShadeNorm(sHit) (sHit->n)
#define readAttr_Tangent(sHit)   (sHit->tg)
#define readAttr_Bitangent(sHit) (sHit->bn)
#define readAttr_TexCoord0(sHit) (sHit->tc0)
#define readAttr_AO(sHit)        (sHit->ao)
#define readAttr_AO1(sHit)       (sHit->ao2)

// the names the generated calls use (RenderDriverRTE::EndTexturesUpdate, RenderDriverRTE_ProcTex.cpp:528-549)
#define MAXPROCTEX 16
typedef struct ProcTextureListT { int currMaxProcTex; int id_f4[MAXPROCTEX]; float3 fdata4[MAXPROCTEX]; } ProcTextureList;   // cglobals.h:2312-2318
typedef float PlainMaterial;   // `pHitMaterial` is the material head as floats
HKU bool materialHeadHaveTargetProcTex(const PlainMaterial* a_pMat, int a_texId) {   // cglobals.h:2755-2773
  bool have = false;
  for (int k = 0; k < MAXPROCTEX; k++) have = have || (__float_as_int(a_pMat[HM_PROC_TEX_IDS + k]) == a_texId);
  return have;
}
HKU int findArgDataOffsetInTable(int a_texId, const int* a_table) {   // texproc.cl:74-90
  const int totalTexNum = a_table[HM_NODE_FLOATS - 1];
  int offset = 0;
  for (int i = 0; i < totalTexNum; i++)
    if (a_table[i * 2 + 0] == a_texId) { offset = a_table[i * 2 + 1]; break; }
  return offset;
}

//#HK_PROCTEX_USER_CODE

// every procedural texture of one hit material (the body of ProcTexExec, texproc.cl:127-190, after its flags test)
HKU void evalAll(const PlainMaterial* pHitMaterial, const SurfaceInfo* sHit, const float3 hr_viewVectorHack, const EngineGlobals* in_globals, ProcTextureList& ptl) {
  const float4* in_texStorage1 = reinterpret_cast<const float4*>(in_globals->texStorage);
  const int* head = reinterpret_cast<const int*>(pHitMaterial);
  const int* table = head + head[HM_PROC_TEX_TABLE];
  const int* argdata = table + HM_NODE_FLOATS;
  const float* fdata = reinterpret_cast<const float*>(argdata);
  (void)in_texStorage1; (void)fdata;
  for (int k = 0; k < 5; k++) ptl.fdata4[k] = make_float3(0, 0, 1);
//#HK_PROCTEX_EVAL_CODE
}

}   // namespace hk_user

#ifndef HK_HOST_EMU
// K_proctex: one thread per path of the bounce, between the closest-hit traversal and k_bounce (GPUOCLLayer::runKernel_ComputeHit runs ProcTexExec in the
// same place, GPUOCLKernels.cpp:662-690).  Writes the path's list: ids[k * stride + slot] (HYDRA_INVALID_TEXTURE ends it), vals likewise as four halfs.
__device__ __forceinline__ unsigned hk_float_to_half_bits(float f) { return unsigned(__builtin_bit_cast(unsigned short, _Float16(f))); }   // round to nearest even, as vstore_half does
__device__ __forceinline__ void proctex_one_path(const SceneDev& s, const int i, const float4* __restrict__ pos4, const float4* __restrict__ dir4, const float4* __restrict__ hits,
                                                 int* __restrict__ ids, uint2* __restrict__ vals, const int stride, const int maxNum) {
  const float4 h4 = hits[i];
  HydraLiteHit hit; hit.t = h4.x; hit.primId = as_int(h4.y); hit.instId = as_int(h4.z); hit.geomId = as_int(h4.w);
  int n = 0;
  hk_user::ProcTextureList ptl;
  ptl.currMaxProcTex = 0;
  if (HitSome(hit)) {
    // the material first (triangle record -> remap lists -> head): most paths of most scenes end here, before the surface is worth evaluating
    const TriData td = fetchTri(s, hit);
    const float* head = materialAt(s, remapMaterialId(td.matId, hit.instId, s));
    if ((as_int(head[HM_FLAGS]) & HMF_HAVE_PROC_TEXTURES) != 0) {
      const f3 ray_pos = xyz(pos4[i]), ray_dir = xyz(dir4[i]);
      const m44 worldToObject = load_m44(s.instMatrices + size_t(hit.instId) * 4);
      const SurfaceHit surf = evalSurfaceWith(s, ray_pos, ray_dir, hit, td, worldToObject);
      hk_user::SurfaceInfo si;
      si.wp = hk_user::make_float3(surf.pos.x, surf.pos.y, surf.pos.z);
      const f3 lp = mul4x3(worldToObject, surf.pos);
      si.lp = hk_user::make_float3(lp.x, lp.y, lp.z);
      si.n = hk_user::make_float3(surf.normal.x, surf.normal.y, surf.normal.z);
      si.tg = hk_user::make_float3(surf.tangent.x, surf.tangent.y, surf.tangent.z);
      si.bn = hk_user::make_float3(surf.biTangent.x, surf.biTangent.y, surf.biTangent.z);
      si.tc0 = hk_user::make_float2(surf.texCoord.x, surf.texCoord.y);
      si.ao = 1.0f; si.ao2 = 1.0f;   // ambient-occlusion inputs are not computed by this layer (the front end refuses <ao> nodes)
      hk_user::evalAll(head, &si, hk_user::make_float3(ray_dir.x, ray_dir.y, ray_dir.z), &s, ptl);
      n = ptl.currMaxProcTex < maxNum ? ptl.currMaxProcTex : maxNum;
    }
  }
  for (int k = 0; k < n; k++) {
    ids[size_t(k) * stride + i] = ptl.id_f4[k];
    const hk_user::float3 v = ptl.fdata4[k];
    vals[size_t(k) * stride + i] = make_uint2(hk_float_to_half_bits(v.x) | (hk_float_to_half_bits(v.y) << 16), hk_float_to_half_bits(v.z));
  }
  if (n < maxNum) ids[size_t(n) * stride + i] = int(HYDRA_INVALID_TEXTURE);
}
extern "C" __global__ void __launch_bounds__(256) k_proctex(SceneDev s, SegQ q, const float4* __restrict__ pos4, const float4* __restrict__ dir4, const float4* __restrict__ hits,
                                                           int* __restrict__ ids, uint2* __restrict__ vals, int stride, int maxNum) {
  const SegIter it = segq_iter(q);
  s.ptlSlot = -1;
  for (int idx = it.first; idx < it.count; idx += it.step) proctex_one_path(s, it.base + idx, pos4, dir4, hits, ids, vals, stride, maxNum);
}
// the same for n points handed in (hydra_hip_stage_proctex): slot = point index
extern "C" __global__ void __launch_bounds__(256) k_proctex_points(SceneDev s, int nPoints, const float4* __restrict__ pos4, const float4* __restrict__ dir4, const float4* __restrict__ hits,
                                                                  int* __restrict__ ids, uint2* __restrict__ vals, int stride, int maxNum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  s.ptlSlot = -1;
  if (i < nPoints) proctex_one_path(s, i, pos4, dir4, hits, ids, vals, stride, maxNum);
}
#endif   // HK_HOST_EMU
