/* hydra_oracle.c -- see hydra_oracle.h.  TEST INFRASTRUCTURE ONLY (CPU checker / CPU baseline).
 *
 * All "ref:" comments cite files under /root/reference/hydra_drv/ unless another directory is named.
 * Arithmetic is IEEE float32, evaluated in the order the reference writes it; build with -ffp-contract=off
 * (the reference's x86 build has no FMA contraction either).  vector helpers the reference takes from HydraAPI's
 * LiteMath (absent here) use the OpenCL-branch definitions in cglobals.h:288-304 / the textbook formulas.
 */
#include "hydra_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ constants */
/* ref: cglobals.h:16, 61-66, 405-434, 438-538 */
#define INVALID_TEXTURE 0xFFFFFFFEu
#define M_PI_F      3.14159265358979323846f
#define INV_PI      0.31830988618379067154f
#define INV_TWOPI   0.15915494309189533577f
#define M_TWOPI_F   6.28318530717958647692f
#define GEPSILON    5e-6f
#define DEPSILON    1e-20f
#define DEPSILON2   1e-30f
#define MAXFLOAT_T  FLT_MAX      /* ref: ctrace.h:665-667 -- glibc <math.h> defines MAXFLOAT as FLT_MAX */

enum { TEX_POINT_SAM = 1, TEX_ALPHASRC_W = 2, TEX_CLAMP_U = 4, TEX_CLAMP_V = 8, TEX_COORD_CAM_PROJ = 32 };
enum { HRT_STUPID_PT_MODE = 65536 * 8, HRT_ENABLE_PT_CAUSTICS = 65536 * 2048 };
enum { HRT_ENABLE_DOF = 0, HRT_TRACE_DEPTH = 9, HRT_DIFFUSE_TRACE_DEPTH = 13 };
enum { HRT_DOF_LENS_RADIUS = 0, HRT_DOF_FOCAL_PLANE_DIST = 1, HRT_TILT_ROT_X = 2, HRT_TILT_ROT_Y = 4, HRT_CAM_FOV = 14 };
/* EngineGlobals word offsets, ref: cfetch.h:21-81 */
enum { G_MPROJ_INV = 32, G_MWORLDVIEW_INV = 48, G_VARS_I = 64, G_VARS_F = 128,
       G_TEX_TABLE = 218, G_MAT_TABLE = 219, G_PDF_TABLE = 220, G_GEOM_TABLE = 221, G_FLOAT_ARRAYS = 228,
       G_LSEL_REV_OFFS = 230, G_LSEL_REV_SIZE = 231, G_FLAGS = 234, G_SKY_LIGHT_ID = 235, G_LIGHTS_OFFS = 236, G_LIGHTS_NUM = 238 };
/* materials, ref: cglobals.h:2604-2722, cmaterial.h:200-210, 374-382, 887-903, 1965-2006 */
enum { MAT_FLOATS = 192, MAT_TYPE = 0, MAT_FLAGS = 1, EMISSIVE_COLOR = 4, EMISSIVE_TEXMATRIXID = 8,
       NORMAL_TEX = 83, MAT_COLOR = 10, MAT_TEXMATRIXID = 14,
       PHONG_GLOSINESS = 16, PHONG_GLOSS_TEXID = 17, PHONG_GLOSS_TEXMATRIXID = 18,
       BLEND_FLAGS_OFFSET = 15, BLEND_MAT1 = 16, BLEND_MAT2 = 17, BLEND_FRESNEL_IOR = 18, BLEND_FALOFF_OFFSET = 19,
       BLEND_FALOFF_SIZE = 20, BLEND_TYPE = 21, BLEND_SIGMOID_EXP = 22, BLEND_FLAGS2 = 23 };
enum { MT_PHONG = 0, MT_BLINN = 1, MT_MIRROR = 2, MT_THIN_GLASS = 3, MT_GLASS = 4, MT_TRANSLUCENT = 5, MT_SHADOW_MATTE = 6, MT_LAMBERT = 7, MT_OREN_NAYAR = 8, MT_BLEND_MASK = 9, MT_EMISSIVE = 10, MT_BECKMANN = 13, MT_TRGGX = 14, MT_GGX = 15 };
enum { ORENNAYAR_A = 16, ORENNAYAR_B = 17 };
enum { THINGLASS_GLOSINESS = 16, THINGLASS_GLOSINESS_TEXMATRIXID = 18,                  /* cmaterial.h:472-491 */
       GLASS_IOR = 15, GLASS_GLOSINESS = 21, GLASS_GLOSINESS_TEXMATRIXID = 23,         /* cmaterial.h:566-590 */
       GGX_GLOSINESS = 16, GGX_GLOSINESS_TEXID = 17, GGX_GLOSINESS_TEXMATRIXID = 18 };  /* cmaterial.h:1165-1185 */
enum { G_ESS_GGX_TABLE = 1268, G_ESS_TRANSP_TABLE = 3316 };   /* EngineGlobals::m_essGgx2017Table / m_essTranspTable in int32 words, cfetch.h:21-81 */   /* cmaterial.h:264-276; colour and sampler offsets equal lambert's */
enum { MF_CAST_CAUSTICS = 2, MF_FORBID_EMISSIVE_GI = 512, MF_SKIP_SKY_PORTAL = 1024, MF_CAN_SAMPLE_REFL_ONLY = 32768,
       MF_FLIP_TANGENT = 32768 * 128, MF_ENERGY_FIX = 32768 * 256 };
enum { BMF_FRESNEL = 1, BMF_FALOFF = 2, BMF_REFL_WEIGHT_IS_ONE = 4, BMF_EXTRUSION_LUMINANCE = 16 };
enum { BLEND_SIGMOID = 4, BLEND_INVERT_FALOFF = 1 };
enum { MIX_TREE_MAX_DEEP = 7, FLOATS_PER_SAMPLE = 3, FLOATS_PER_MLAYER = 7 };
/* lights, ref: clight.h:14-62, 493-521; cglobals.h:2236-2252 */
enum { LIGHT_FLOATS = 128, PL_TYPE = 0, PL_FLAGS = 1, PL_POS = 2, PL_NORM = 5, PL_COLOR = 8, PL_COLOR_TEX = 11,
       PL_SURFACE_AREA = 13, AL_SIZE_X = 14, AL_SIZE_Y = 15, AL_MATRIX = 16, AL_IS_DISK = 25, AL_SPOT_DISTR = 26,
       AL_SPOT_COS1 = 27, AL_SPOT_COS2 = 28, PL_PICK_PROB_REV = 107,
       PL_COLOR_TEX_MATRIX = 12, SKY_DOME_PDF_TABLE0 = 30, SKY_DOME_SAMPLER0 = 32, SKY_DOME_MATRIX0 = 36, SKY_DOME_INV_MATRIX0 = 56 };   /* clight.h:131-165 */
enum { LT_POINT_OMNI = 0, LT_POINT_SPOT = 1, LT_DIRECT = 2, LT_SKY_DOME = 3, LT_AREA = 4, LT_SPHERE = 5, LT_CYLINDER = 6, LT_MESH = 7 };
enum { CYLINDER_LIGHT_MATRIX_E00 = 16, CYLINDER_LIGHT_RADIUS = 25, CYLINDER_LIGHT_ZMIN = 26, CYLINDER_LIGHT_ZMAX = 27, CYLINDER_LIGHT_PHIMAX = 28,
       CYLINDER_TEXMATRIX_ID = 30, CYLINDER_PDF_TABLE_ID = 31 };   /* clight.h:96-114 */
enum { AREA_LIGHT_SKY_OFFSET = 29, MESH_LIGHT_TEXMATRIX_ID = 31, G_SUN_NUMBER = 242, G_SUNS = 243 };   /* clight.h:57, 174; cfetch.h:74-75 in int32 words */
enum { MESH_LIGHT_MESH_OFFSET_ID = 14, MESH_LIGHT_TABLE_OFFSET_ID = 15, MESH_LIGHT_TRI_NUM = 16, MESH_LIGHT_MATRIX_E00 = 20 };   /* clight.h:169-176 */
enum { SPHERE_LIGHT_RADIUS = 14 };   /* clight.h:33 */
enum { POINT_LIGHT_SPOT_COS1 = 14, POINT_LIGHT_SPOT_COS2 = 15, DIRECT_LIGHT_RADIUS1 = 14, DIRECT_LIGHT_RADIUS2 = 15,
       DIRECT_LIGHT_SSOFTNESS = 16, DIRECT_LIGHT_ALPHA_TAN = 17, DIRECT_LIGHT_ALPHA_COS = 18 };   /* clight.h:118-127 */
enum { HRT_BSPHERE_RADIUS = 21 };
enum { LF_SKY_PORTAL = 8, LF_HAS_IES = 16, LF_IES_POINT_AREA = 32 };
enum { IES_INV_MATRIX_E00 = 108, IES_LIGHT_MATRIX_E00 = 117, IES_SPHERE_PDF_ID = 126, IES_SPHERE_TEX_ID = 127 };   /* clight.h:40-62 */
/* ray flags, ref: cglobals.h:1330-1376 */
enum { RAY_EVENT_S = 1, RAY_EVENT_D = 2, RAY_EVENT_G = 4, RAY_EVENT_T = 8, RAY_EVENT_TNINGLASS = 64 };
enum { RAY_GRAMMAR_DIRECT_LIGHT = 64, RAY_IS_DEAD = 4096 };

/* ------------------------------------------------------------------------------------------------ small vectors */
typedef struct { float x, y; } f2;
typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;
typedef struct { f4 c[4]; } m44;      /* columns, ref: cglobals.h:792-800 make_float4x4 */

static inline f3 v3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 add3(f3 a, f3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 scale3(f3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 cross3(f3 a, f3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float length3(f3 a) { return sqrtf(dot3(a, a)); }
static inline f3 normalize3(f3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }
static inline float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
static inline f3 clamp3(f3 v, float a, float b) { return v3(clampf(v.x, a, b), clampf(v.y, a, b), clampf(v.z, a, b)); }
static inline int32_t as_int(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline float as_float(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline int finite3(f3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }

static inline m44 load_m44(const float* p) { m44 m; memcpy(&m, p, 64); return m; }
/* ref: cglobals.h:288-304 (mul4x3, mul3x3), :828-849 (mul4x4x4, mul) */
static inline f3 mul4x3(m44 m, f3 v) {
  return v3(v.x * m.c[0].x + v.y * m.c[1].x + v.z * m.c[2].x + m.c[3].x,
            v.x * m.c[0].y + v.y * m.c[1].y + v.z * m.c[2].y + m.c[3].y,
            v.x * m.c[0].z + v.y * m.c[1].z + v.z * m.c[2].z + m.c[3].z);
}
static inline f3 mul3x3(m44 m, f3 v) {
  return v3(v.x * m.c[0].x + v.y * m.c[1].x + v.z * m.c[2].x,
            v.x * m.c[0].y + v.y * m.c[1].y + v.z * m.c[2].y,
            v.x * m.c[0].z + v.y * m.c[1].z + v.z * m.c[2].z);
}
static inline f4 mul4x4x4(m44 m, f4 v) {
  f4 r;
  r.x = v.x * m.c[0].x + v.y * m.c[1].x + v.z * m.c[2].x + v.w * m.c[3].x;
  r.y = v.x * m.c[0].y + v.y * m.c[1].y + v.z * m.c[2].y + v.w * m.c[3].y;
  r.z = v.x * m.c[0].z + v.y * m.c[1].z + v.z * m.c[2].z + v.w * m.c[3].z;
  r.w = v.x * m.c[0].w + v.y * m.c[1].w + v.z * m.c[2].w + v.w * m.c[3].w;
  return r;
}
static inline m44 transpose44(m44 a) {  /* ref: cglobals.h:1038-1047 */
  m44 r;
  r.c[0].x = a.c[0].x; r.c[0].y = a.c[1].x; r.c[0].z = a.c[2].x; r.c[0].w = a.c[3].x;
  r.c[1].x = a.c[0].y; r.c[1].y = a.c[1].y; r.c[1].z = a.c[2].y; r.c[1].w = a.c[3].y;
  r.c[2].x = a.c[0].z; r.c[2].y = a.c[1].z; r.c[2].z = a.c[2].z; r.c[2].w = a.c[3].z;
  r.c[3].x = a.c[0].w; r.c[3].y = a.c[1].w; r.c[3].z = a.c[2].w; r.c[3].w = a.c[3].w;
  return r;
}
/* inverse4x4: LiteMath (absent).  Instance matrices are affine, so: inverse of the upper 3x3 by cofactors and the
 * translation -R^-1 t, all in float32.  The HIP kernels use the same formula. */
static inline m44 inverse_affine(m44 m) {
  const float a00 = m.c[0].x, a10 = m.c[0].y, a20 = m.c[0].z;
  const float a01 = m.c[1].x, a11 = m.c[1].y, a21 = m.c[1].z;
  const float a02 = m.c[2].x, a12 = m.c[2].y, a22 = m.c[2].z;
  const float c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const float det = a00 * c00 + a01 * c01 + a02 * c02;
  const float id = 1.0f / det;
  m44 r;
  r.c[0].x = c00 * id;                          r.c[0].y = c01 * id;                          r.c[0].z = c02 * id;                          r.c[0].w = 0.0f;
  r.c[1].x = (a02 * a21 - a01 * a22) * id;      r.c[1].y = (a00 * a22 - a02 * a20) * id;      r.c[1].z = (a01 * a20 - a00 * a21) * id;      r.c[1].w = 0.0f;
  r.c[2].x = (a01 * a12 - a02 * a11) * id;      r.c[2].y = (a02 * a10 - a00 * a12) * id;      r.c[2].z = (a00 * a11 - a01 * a10) * id;      r.c[2].w = 0.0f;
  const f3 t = v3(m.c[3].x, m.c[3].y, m.c[3].z);
  r.c[3].x = -(r.c[0].x * t.x + r.c[1].x * t.y + r.c[2].x * t.z);
  r.c[3].y = -(r.c[0].y * t.x + r.c[1].y * t.y + r.c[2].y * t.z);
  r.c[3].z = -(r.c[0].z * t.x + r.c[1].z * t.y + r.c[2].z * t.z);
  r.c[3].w = 1.0f;
  return r;
}

/* ------------------------------------------------------------------------------------------------ R1: random numbers */
/* ref: crandom.h:20-26 NextState */
uint32_t orc_next_state(uint32_t st[2]) {
  const uint32_t x = st[0] * 17u + st[1] * 13123u;
  st[0] = (x << 13) ^ x;
  st[1] ^= (x << 7);
  return x;
}
/* ref: crandom.h:28-43 RandomGenInit (int arithmetic wraps; evaluated here in uint32) */
void orc_random_init(int32_t a_seed, uint32_t st[2]) {
  const uint32_t s = (uint32_t)a_seed;
  st[0] = (s * (s * s * 15731u + 74323u) + 871483u);
  st[1] = (s * (s * s * 13734u + 37828u) + 234234u);
  for (int i = 0; i < (a_seed % 7); i++) orc_next_state(st);
}
/* ref: crandom.h:51-63 rndFloat4_Pseudo */
void orc_rnd_float4(uint32_t st[2], float out[4]) {
  const uint32_t x = orc_next_state(st);
  const uint32_t x1 = (x * (x * x * 15731u + 74323u) + 871483u);
  const uint32_t y1 = (x * (x * x * 13734u + 37828u) + 234234u);
  const uint32_t z1 = (x * (x * x * 11687u + 26461u) + 137589u);
  const uint32_t w1 = (x * (x * x * 15707u + 789221u) + 1376312589u);
  const float scale = (1.0f / 4294967296.0f);
  out[0] = (float)(x1) * scale; out[1] = (float)(y1) * scale; out[2] = (float)(z1) * scale; out[3] = (float)(w1) * scale;
}
/* ref: crandom.h:77-83 rndFloat1_Pseudo */
float orc_rnd_float1(uint32_t st[2]) {
  const uint32_t x = orc_next_state(st);
  const uint32_t tmp = (x * (x * x * 15731u + 74323u) + 871483u);
  return ((float)(tmp)) * (1.0f / 4294967296.0f);
}

/* ------------------------------------------------------------------------------------------------ shared helpers */
static inline const float* g_varsF(const OrcScene* s) { return (const float*)(s->globals + G_VARS_F); }
static inline const int32_t* g_varsI(const OrcScene* s) { return s->globals + G_VARS_I; }

/* ref: cglobals.h:726-735 SafeInverse */
static inline f3 SafeInverse(f3 d) {
  const float ooeps = 1.0e-36f;
  f3 r;
  r.x = 1.0f / (fabsf(d.x) > ooeps ? d.x : copysignf(ooeps, d.x));
  r.y = 1.0f / (fabsf(d.y) > ooeps ? d.y : copysignf(ooeps, d.y));
  r.z = 1.0f / (fabsf(d.z) > ooeps ? d.z : copysignf(ooeps, d.z));
  return r;
}
/* ref: cglobals.h:737-745 */
static inline float epsilonOfPos(f3 p) { return fmaxf(fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))), 2.0f * GEPSILON) * GEPSILON; }
static inline float misHeuristicPower1(float p) { return isfinite(p) ? fabsf(p) : 0.0f; }
static inline float misWeightHeuristic(float a, float b) {
  const float w = misHeuristicPower1(a) / fmaxf(misHeuristicPower1(a) + misHeuristicPower1(b), DEPSILON2);
  return isfinite(w) ? w : 0.0f;
}
/* ref: cglobals.h:764-784 */
static inline f3 OffsRayPos(f3 hitPos, f3 n, f3 sampleDir) {
  const float sgn = dot3(sampleDir, n) < 0.0f ? -1.0f : 1.0f;
  const float eps = epsilonOfPos(hitPos);
  return add3(hitPos, scale3(n, sgn * eps));
}
static inline f3 OffsShadowRayPos(f3 hitPos, f3 n, f3 sampleDir, float aux) {
  const float sgn = dot3(sampleDir, n) < 0.0f ? -1.0f : 1.0f;
  const float eps = epsilonOfPos(hitPos);
  return add3(hitPos, scale3(n, sgn * (eps + aux)));
}
/* ref: cglobals.h:686-691 reflect */
static inline f3 reflect3(f3 dir, f3 n) { return normalize3(add3(scale3(scale3(n, dot3(dir, n)), -2.0f), dir)); }

/* ref: cglobals.h:1502-1518 CoordinateSystem */
static inline void CoordinateSystem(f3 v1, f3* v2, f3* v3o) {
  float invLen;
  if (fabsf(v1.x) > fabsf(v1.y)) {
    invLen = 1.0f / sqrtf(v1.x * v1.x + v1.z * v1.z);
    *v2 = v3((-1.0f) * v1.z * invLen, 0.0f, v1.x * invLen);
  } else {
    invLen = 1.0f / sqrtf(v1.y * v1.y + v1.z * v1.z);
    *v2 = v3(0.0f, v1.z * invLen, (-1.0f) * v1.y * invLen);
  }
  *v3o = cross3(v1, *v2);
}
/* ref: cglobals.h:1521-1559 MapSampleToCosineDistribution */
static f3 MapSampleToCosineDistribution(float r1, float r2, f3 direction, f3 hit_norm, float power) {
  if (power >= 1e6f) return direction;
  const float sin_phi = sinf(2.0f * r1 * 3.141592654f);
  const float cos_phi = cosf(2.0f * r1 * 3.141592654f);
  const float cos_theta = powf(1.0f - r2, 1.0f / (power + 1.0f));
  const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
  const f3 dev = v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
  f3 ny = direction, nx, nz;
  CoordinateSystem(ny, &nx, &nz);
  { f3 t = ny; ny = nz; nz = t; }
  f3 res = add3(add3(scale3(nx, dev.x), scale3(ny, dev.y)), scale3(nz, dev.z));
  const float invSign = dot3(direction, hit_norm) > 0.0f ? 1.0f : -1.0f;
  if (invSign * dot3(res, hit_norm) < 0.0f)
    res = sub3(add3(scale3(scale3(nx, -1.0f), dev.x), scale3(ny, dev.y)), scale3(nz, dev.z));
  return res;
}
/* ref: cglobals.h:1563-1601 MapSampleToModifiedCosineDistribution */
static f3 MapSampleToModifiedCosineDistribution(float r1, float r2, f3 direction, f3 hit_norm, float power, int* under) {
  if (power >= 1e6f) return direction;   /* note: *under keeps the caller's initial 'false' */
  const float sin_phi = sinf(2.0f * r1 * 3.141592654f);
  const float cos_phi = cosf(2.0f * r1 * 3.141592654f);
  const float sin_theta = sqrtf(1.0f - powf(r2, 2.0f / (power + 1.0f)));
  f3 dev;
  dev.x = sin_theta * cos_phi;
  dev.y = sin_theta * sin_phi;
  dev.z = sqrtf(1.0f - dev.x * dev.x - dev.y * dev.y);
  f3 ny = direction, nx, nz;
  CoordinateSystem(ny, &nx, &nz);
  { f3 t = ny; ny = nz; nz = t; }
  f3 res = add3(add3(scale3(nx, dev.x), scale3(ny, dev.y)), scale3(nz, dev.z));
  *under = 0;
  const float invSign = dot3(direction, hit_norm) >= 0.0f ? 1.0f : -1.0f;
  if (invSign * dot3(res, hit_norm) < 0.0f) {
    res = add3(sub3(scale3(scale3(nx, -1.0f), dev.x), scale3(ny, dev.y)), scale3(nz, dev.z));
    *under = 1;
  }
  return res;
}
/* ref: cglobals.h:1609-1652 MapSamplesToDisc */
static f2 MapSamplesToDisc(f2 xy) {
  const float x = xy.x, y = xy.y;
  float r = 0, phi = 0;
  if (x > y && x > -y) { r = x; phi = 0.25f * 3.141592654f * (y / x); }
  if (x < y && x > -y) { r = y; phi = 0.25f * 3.141592654f * (2.0f - x / y); }
  if (x < y && x < -y) { r = -x; phi = 0.25f * 3.141592654f * (4.0f + y / x); }
  if (x > y && x < -y) { r = -y; phi = 0.25f * 3.141592654f * (6 - x / y); }
  const float sin_phi = sinf(phi), cos_phi = cosf(phi);
  f2 res = {r * sin_phi, r * cos_phi};
  return res;
}
/* ref: cglobals.h:3024-3038 */
static inline float sRGBToLinear(float s) {
  if (s <= 0.0404482362771082f) return s * 0.077399381f;
  return powf((s + 0.055f) * 0.947867299f, 2.4f);
}

/* ------------------------------------------------------------------------------------------------ P1: camera */
/* ref: cglobals.h:1069-1087 EyeRayDirNormalized, matrix4x4f_mult_ray3 */
static f3 EyeRayDirNormalized(float x, float y, m44 projInv) {
  f4 pos = {2.0f * x - 1.0f, 2.0f * y - 1.0f, 0.0f, 1.0f};
  pos = mul4x4x4(projInv, pos);
  pos.x /= pos.w; pos.y /= pos.w; pos.z /= pos.w;
  return normalize3(v3(pos.x, pos.y, pos.z));
}
/* ref: cfetch.h:832-863 tiltCorrection.  Rotation by make_matrix_rotationX/Y cglobals.h:802-826 */
static f3 tiltCorrection(f3 ray_pos, f3 ray_dir, const OrcScene* s) {
  const float tiltX = g_varsF(s)[HRT_TILT_ROT_X], tiltY = g_varsF(s)[HRT_TILT_ROT_Y];
  if ((fabsf(tiltX) > 0.0f || fabsf(tiltY) > 0.0f) && fabsf(ray_dir.z) > 0.0f) {
    const float t = (-1.0f - ray_pos.z) / ray_dir.z;
    f3 p = add3(ray_pos, scale3(ray_dir, t));
    p.z += 1.0f;
    if (fabsf(tiltY) > 0.0f) {
      const float sn = sinf(-tiltY), cs = cosf(-tiltY);
      p = v3(p.x * cs + p.z * sn, p.y, p.x * (-sn) + p.z * cs);
    }
    if (fabsf(tiltX) > 0.0f) {
      const float sn = sinf(-tiltX), cs = cosf(-tiltX);
      p = v3(p.x, p.y * cs + p.z * (-sn), p.y * sn + p.z * cs);
    }
    p.z -= 1.0f;
    ray_dir = normalize3(sub3(p, ray_pos));
  }
  return ray_dir;
}
/* ref: cfetch.h:877-930 MakeRandEyeRay */
static void MakeRandEyeRay(int x, int y, int w, int h, const float offs[4], const OrcScene* s, f3* pRayPos, f3* pRayDir) {
  const m44 projInv = load_m44((const float*)(s->globals + G_MPROJ_INV));
  const m44 wvInv = load_m44((const float*)(s->globals + G_MWORLDVIEW_INV));
  const float sx = (float)x + 0.5f, sy = (float)y + 0.5f;
  f3 ray_pos = v3(0.0f, 0.0f, 0.0f);
  f3 ray_dir = EyeRayDirNormalized(sx / (float)w, sy / (float)h, projInv);
  {
    const float sinFov = sinf(0.5f * g_varsF(s)[HRT_CAM_FOV]);
    const float pxSizeX = sinFov * (1.0f / (float)w);
    const float pxSizeY = sinFov * (1.0f / (float)h);
    ray_dir.x += pxSizeX * offs[0];
    ray_dir.y += pxSizeY * offs[1];
    ray_dir.z = -sqrtf(1.0f - (ray_dir.x * ray_dir.x + ray_dir.y * ray_dir.y));
  }
  ray_dir = tiltCorrection(ray_pos, ray_dir, s);
  if (g_varsI(s)[HRT_ENABLE_DOF] == 1) {
    const float tFocus = g_varsF(s)[HRT_DOF_FOCAL_PLANE_DIST] / (-ray_dir.z);
    const f3 focusPosition = add3(ray_pos, scale3(ray_dir, tFocus));
    f2 in = {1.0f * offs[2], 1.0f * offs[3]};
    const f2 d = MapSamplesToDisc(in);
    const float R = g_varsF(s)[HRT_DOF_LENS_RADIUS];
    ray_pos.x += R * d.x;
    ray_pos.y += R * d.y;
    ray_dir = normalize3(sub3(focusPosition, ray_pos));
  }
  {
    const f3 pos = mul4x3(wvInv, ray_pos);
    const f3 pos2 = mul4x3(wvInv, add3(ray_pos, scale3(ray_dir, 100.0f)));
    *pRayPos = pos;
    *pRayDir = normalize3(sub3(pos2, pos));
  }
}

void orc_make_eye_rays(const OrcScene* s, int n, int w, int h, const int32_t* xy, const float* offs4, float* pos4, float* dir4) {
  for (int i = 0; i < n; i++) {
    f3 p, d;
    MakeRandEyeRay(xy[2 * i], xy[2 * i + 1], w, h, offs4 + 4 * i, s, &p, &d);
    pos4[4 * i] = p.x; pos4[4 * i + 1] = p.y; pos4[4 * i + 2] = p.z; pos4[4 * i + 3] = 0.0f;
    dir4[4 * i] = d.x; dir4[4 * i + 1] = d.y; dir4[4 * i + 2] = d.z; dir4[4 * i + 3] = 0.0f;
  }
}

/* ------------------------------------------------------------------------------------------------ T1/T2: traversal */
#define STACK_SIZE 80   /* ref: ctrace.h:576 */
typedef struct { uint32_t quads, insts, tris, leaves; } TravStat;

/* ref: ctrace.h:32-53 RayBoxIntersectionLite2 */
static inline f2 RayBox(f3 o, f3 inv, const float* node) {
  const float lo = inv.x * (node[0] - o.x), hi = inv.x * (node[4] - o.x);
  const float lo1 = inv.y * (node[1] - o.y), hi1 = inv.y * (node[5] - o.y);
  const float lo2 = inv.z * (node[2] - o.z), hi2 = inv.z * (node[6] - o.z);
  float tmin = fminf(lo, hi), tmax = fmaxf(lo, hi);
  tmin = fmaxf(tmin, fminf(lo1, hi1)); tmax = fminf(tmax, fmaxf(lo1, hi1));
  tmin = fmaxf(tmin, fminf(lo2, hi2)); tmax = fminf(tmax, fmaxf(lo2, hi2));
  f2 r = {tmin, tmax};
  return r;
}
static f3 sample2DExt(int samplerOffset, f2 texCoord, const float* blob, const OrcScene* s);
/* ref: ctrace.h:316-325 decompressTexCoord16 */
static inline f2 decompressTexCoord16(uint32_t packed) {
  const float fx = (1.0f / 65535.0f) * (float)(packed & 0x0000FFFFu), fy = (1.0f / 65535.0f) * (float)((packed & 0xFFFF0000u) >> 16);
  f2 r = {2.0f * fx - 1.0f, 2.0f * fy - 1.0f};
  return r;
}
/* ref: ctrace.h:330-412 IntersectAllPrimitivesInLeafAlpha, the part behind a geometric hit: texture coordinate from the packed
 * per-vertex ones, opacity texel through sample2DLite (cfetch.h:738-760; the sampler sits in the alpha table itself), accept when the
 * largest channel exceeds 0.5 */
static inline int alphaTestPasses(const OrcScene* s, const uint32_t* alpha, int triAddress, float u, float v) {
  const uint32_t* a0 = alpha + 2 * (size_t)triAddress, *a1 = a0 + 2, *a2 = a0 + 4;
  const f2 A = decompressTexCoord16(a0[1]), B = decompressTexCoord16(a1[1]), C = decompressTexCoord16(a2[1]);
  const float w = 1.0f - u - v;
  const f2 tc = {(w * A.x + v * B.x) + u * C.x, (w * A.y + v * B.y) + u * C.y};
  if (a0[0] == 0xFFFFFFFFu || a0[0] == 0xFFFFFFFEu || (int32_t)a0[0] <= 0) return 1;
  const f3 c = sample2DExt(0, tc, (const float*)(alpha + 2 * (size_t)a0[0]), s);
  return fmaxf(c.x, fmaxf(c.y, c.z)) > 0.5f;
}
/* ref: ctrace.h:63-182 IntersectAllPrimitivesInLeaf(1): Moeller-Trumbore, first-found wins ties (t < best); alpha != NULL: :330-412 */
static inline OrcHit IntersectLeaf(f3 ray_pos, f3 ray_dir, int leaf_offset, float t_min, OrcHit res, const float* tris, int instIdOverride, int useOverride, TravStat* st,
                                   const OrcScene* s, const uint32_t* alpha) {
  const int first = as_int(tris[leaf_offset * 4 + 0]), count = as_int(tris[leaf_offset * 4 + 1]);
  const int end = first + count * 3;
  if (st) { st->leaves++; st->tris += (uint32_t)count; }
  for (int a = first; a < end; a += 3) {
    const float* d1 = tris + a * 4; const float* d2 = d1 + 4; const float* d3 = d1 + 8;
    const f3 A = v3(d1[0], d1[1], d1[2]), B = v3(d2[0], d2[1], d2[2]), C = v3(d3[0], d3[1], d3[2]);
    const f3 edge1 = sub3(B, A), edge2 = sub3(C, A);
    const f3 pvec = cross3(ray_dir, edge2);
    const f3 tvec = sub3(ray_pos, A);
    const f3 qvec = cross3(tvec, edge1);
    const float invDet = 1.0f / dot3(edge1, pvec);
    const float v = dot3(tvec, pvec) * invDet;
    const float u = dot3(qvec, ray_dir) * invDet;
    const float t = dot3(edge2, qvec) * invDet;
    if (v > -1e-6f && u > -1e-6f && (u + v < 1.0f + 1e-6f) && t > t_min && t < res.t) {
      if (alpha != NULL && !alphaTestPasses(s, alpha, a, u, v)) continue;
      res.t = t; res.primId = as_int(d1[3]); res.geomId = as_int(d2[3]);
      res.instId = useOverride ? instIdOverride : as_int(d3[3]);
    }
  }
  return res;
}

/* ref: ctrace.h:841-1062 BVH4InstTraverse (haveInst) and :669-838 BVH4Traverse (!haveInst): one state machine, the
 * instancing steps are skipped for plain trees exactly as the reference's second function omits them. */
/* anyHit: the shadow form BVH4InstTraverseShadow (ref: ctrace.h:1065-1294): the caller seeds hit.t with the ray's far end and
 * the walk stops behind the first leaf that produced a hit in (t_rayMin, far) (ref: :1243-1251, `top = 0`). */
static OrcHit BVH4Traverse_(f3 ray_pos, f3 ray_dir, float t_rayMin, OrcHit hit, const float* bvh, const float* tris, int haveInst, int anyHit, TravStat* st,
                            const OrcScene* s, const uint32_t* alpha) {   /* alpha: BVH4InstTraverseAlpha, ref: ctrace.h:1297-1520 (instanced trees only) */
#ifdef ORC_NO_STATS   /* liboracle_fast.so, bench.py's cpu_baseline leg only: the visit counters are compiled out of the walk */
  st = NULL;
#endif
  f3 invDir = SafeInverse(ray_dir);
  /* the reference declares stackData[80] with stack = stackData + 2 and tests `top < 80` once before up to three pushes
   * (ctrace.h:846-847,964-985): on a tree deeper than the stack it writes stack[78..81], i.e. up to four ints past its
   * array.  Here those four words exist, which is the reference's behaviour whenever its stray writes land on harmless
   * memory; the HIP kernels size their stack the same way (hk_trace.h, HK_STACK_SLACK). */
  int stackData[STACK_SIZE + 4];
  int* stack = stackData + 2;
  stackData[0] = stackData[1] = 0;   /* the reference reads stack[-1] uninitialised after the last pop */
  int top = 0, leftNodeOffset = 1, searchingForLeaf = 1;
  int instDeep = 0, instTop = 0, instId = -1;
  f3 old_pos = v3(0, 0, 0), old_dir = v3(0, 0, 0);

  while (top >= 0) {
    while (searchingForLeaf) {
      int child[4]; float key[4];
      if (st) st->quads++;
      for (int k = 0; k < 4; k++) {
        const float* node = bvh + (size_t)(4 * leftNodeOffset + k) * 8;
        const uint32_t loal = (uint32_t)as_int(node[3]), esc = (uint32_t)as_int(node[7]);
        const int valid = !(loal == 0xffffffffu && esc == 0xffffffffu);
        const f2 tm = RayBox(ray_pos, invDir, node);
        const int hitChild = (tm.x <= tm.y) && (tm.y >= t_rayMin) && (tm.x <= hit.t) && valid;
        child[k] = (int)loal;
        key[k] = hitChild ? tm.x : MAXFLOAT_T;
      }
      /* 5-comparator sorting network, ref: ctrace.h:906-960 */
#define CSWAP(i, j) if (key[j] < key[i]) { float tk = key[i]; key[i] = key[j]; key[j] = tk; int tc = child[i]; child[i] = child[j]; child[j] = tc; }
      CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
#undef CSWAP
      const int stackHaveSpace = (top < STACK_SIZE);
      if (key[3] < MAXFLOAT_T && stackHaveSpace) { stack[top] = child[3]; top++; }
      if (key[2] < MAXFLOAT_T && stackHaveSpace) { stack[top] = child[2]; top++; }
      if (key[1] < MAXFLOAT_T && stackHaveSpace) { stack[top] = child[1]; top++; }
      if (key[0] < MAXFLOAT_T) leftNodeOffset = child[0];
      else if (top >= 0) { top--; leftNodeOffset = stack[top]; }
      searchingForLeaf = !(leftNodeOffset & 0x80000000) && (top >= 0);
      leftNodeOffset = leftNodeOffset & 0x7fffffff;
      if (haveInst && top < instTop && instDeep == 1) {
        ray_pos = old_pos; ray_dir = old_dir; invDir = SafeInverse(ray_dir); instDeep = 0;
      }
    }
    if (!haveInst) {
      if (top >= 0) hit = IntersectLeaf(ray_pos, ray_dir, leftNodeOffset, t_rayMin, hit, tris, 0, 0, st, s, NULL);
      if (anyHit && hit.primId != -1) break;
      top--;
      leftNodeOffset = stack[top];
    } else if (top >= 0 && instDeep == 1) {
      hit = IntersectLeaf(ray_pos, ray_dir, leftNodeOffset, t_rayMin, hit, tris, instId, 1, st, s, alpha);
      if (anyHit && hit.primId != -1) break;
      top--;
      leftNodeOffset = stack[top];
    } else if (top >= 0 && instDeep == 0) {
      instDeep = 1;
      old_pos = ray_pos; old_dir = ray_dir;
      const float* q = bvh + (size_t)leftNodeOffset * 32;
      const int nextOffset = as_int(q[3]);
      const m44 matrix = load_m44(q + 8);
      instId = as_int(q[24]);
      if (st) st->insts++;
      ray_pos = mul4x3(matrix, ray_pos);
      ray_dir = mul3x3(matrix, ray_dir);   /* not normalised: t stays in world units */
      invDir = SafeInverse(ray_dir);
      instTop = top;
      leftNodeOffset = nextOffset;
    }
    searchingForLeaf = !(leftNodeOffset & 0x80000000);
    leftNodeOffset = leftNodeOffset & 0x7fffffff;
    if (haveInst && top < instTop && instDeep == 1) {
      ray_pos = old_pos; ray_dir = old_dir; invDir = SafeInverse(ray_dir); instDeep = 0;
    }
  }
  return hit;
}

/* ref: CPUExp_Integrators_Common.cpp:122-154 IntegratorCommon::rayTrace (tree 0; Make_Lite_Hit cglobals.h:1256-1266) */
static OrcHit rayTrace(const OrcScene* s, f3 pos, f3 dir, TravStat* st) {
  OrcHit h; h.t = MAXFLOAT_T; h.primId = -1; h.instId = -1; h.geomId = (int32_t)(((uint32_t)(-1) << 30) & 0xC0000000u);
  h = BVH4Traverse_(pos, dir, 0.0f, h, s->bvh, s->tris, s->haveInst, 0, st, s, s->haveInst ? s->alpha[0] : NULL);
  for (int i = 1; i < s->treesNum && i < 4; i++)   /* one running Lite_Hit over all trees */
    if (s->bvhN[i - 1] != NULL) h = BVH4Traverse_(pos, dir, 0.0f, h, s->bvhN[i - 1], s->trisN[i - 1], s->haveInstN[i - 1], 0, st, s, s->haveInstN[i - 1] ? s->alpha[i] : NULL);
  return h;
}
static inline int HitSome(OrcHit h) { return (h.primId != -1) && isfinite(h.t); }
/* ref: Common.cpp:156-180 IntegratorCommon::shadowTrace: full closest hit, then 0 < t < t_far */
static float shadowTrace(const OrcScene* s, f3 pos, f3 dir, float t_far) {
  OrcHit h; h.t = MAXFLOAT_T; h.primId = -1; h.instId = -1; h.geomId = (int32_t)0xC0000000u;
  h = BVH4Traverse_(pos, dir, 0.0f, h, s->bvh, s->tris, s->haveInst, 0, NULL, s, NULL);   /* tree 0 only, plain BVH4InstTraverse: no alpha test for shadow rays on the CPU path */
  return (HitSome(h) && h.t > 0.0f && h.t < t_far) ? 0.0f : 1.0f;
}

void orc_trace(const OrcScene* s, int n, const float* pos4, const float* dir4, OrcHit* hits, uint32_t* c3, uint32_t* leaves1) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int i = 0; i < n; i++) {
    TravStat st = {0, 0, 0, 0};
    hits[i] = rayTrace(s, v3(pos4[4 * i], pos4[4 * i + 1], pos4[4 * i + 2]), v3(dir4[4 * i], dir4[4 * i + 1], dir4[4 * i + 2]), &st);
    if (c3) { c3[3 * i] = st.quads; c3[3 * i + 1] = st.insts; c3[3 * i + 2] = st.tris; }
    if (leaves1) leaves1[i] = st.leaves;
  }
}
/* ref: ctrace.h:1065-1294 BVH4InstTraverseShadow as launched by BVH4TraversalInstShadowKenrel (shaders/trace.cl:309-353):
 * hit seeded with Make_Lite_Hit(maxDist, -1), visibility 0 when some triangle lies in (0, maxDist).  c4 (optional) = quads,
 * instance quads, triangles, leaves visited by the early-out walk. */
void orc_shadow_trace_anyhit(const OrcScene* s, int n, const float* pos4, const float* dir4, const float* tfar, float* vis, uint32_t* c4) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int i = 0; i < n; i++) {
    TravStat st = {0, 0, 0, 0};
    OrcHit h; h.t = tfar[i]; h.primId = -1; h.instId = -1; h.geomId = (int32_t)0xC0000000u;
    h = BVH4Traverse_(v3(pos4[4 * i], pos4[4 * i + 1], pos4[4 * i + 2]), v3(dir4[4 * i], dir4[4 * i + 1], dir4[4 * i + 2]), 0.0f, h, s->bvh, s->tris, s->haveInst, 1, &st, s, NULL);
    vis[i] = (h.primId != -1) ? 0.0f : 1.0f;
    if (c4) { c4[4 * i] = st.quads; c4[4 * i + 1] = st.insts; c4[4 * i + 2] = st.tris; c4[4 * i + 3] = st.leaves; }
  }
}
void orc_shadow_trace(const OrcScene* s, int n, const float* pos4, const float* dir4, const float* tfar, float* vis) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int i = 0; i < n; i++)
    vis[i] = shadowTrace(s, v3(pos4[4 * i], pos4[4 * i + 1], pos4[4 * i + 2]), v3(dir4[4 * i], dir4[4 * i + 1], dir4[4 * i + 2]), tfar[i]);
}

/* ------------------------------------------------------------------------------------------------ H1: surface */
typedef struct {
  f3 pos, normal, flatNormal, tangent, biTangent;
  f2 texCoord;
  int matId; float t, sRayOff; int hfi;
} SurfaceHit;

/* ref: cglobals.h:2931-2983 remapMaterialId */
static int remapMaterialId(int mId, int instId, const OrcScene* s) {
  if (mId < 0 || instId < 0 || instId >= s->remapInstSize || s->remapInst == NULL || s->remapLists == NULL || s->remapTable == NULL) return mId;
  const int listId = s->remapInst[instId];
  if (listId < 0 || listId >= s->remapTableSize) return mId;
  const int offs = s->remapTable[2 * listId], size = s->remapTable[2 * listId + 1];
  int low = 0, high = size - 1;
  while (low <= high) {
    const int mid = low + ((high - low) / 2);
    if (s->remapLists[offs + mid * 2] >= mId) high = mid - 1; else low = mid + 1;
  }
  if (high + 1 < size) {
    const int from = s->remapLists[offs + (high + 1) * 2], to = s->remapLists[offs + (high + 1) * 2 + 1];
    return (from == mId) ? to : mId;
  }
  return mId;
}

/* ref: ctrace.h:1988-2109 triBaricentrics + surfaceEvalLS; mesh accessors cfetch.h:1038-1119 */
static SurfaceHit surfaceEvalLS(f3 a_rpos, f3 a_rdir, OrcHit hit, const float* mesh) {
  const int32_t* hdr = (const int32_t*)mesh;
  const float* vertPos = mesh + (size_t)hdr[0] * 4;
  const float* vertNorm = mesh + (size_t)hdr[1] * 4;
  const float* vertTang = mesh + (size_t)hdr[10] * 4;
  const int32_t* vertIndices = (const int32_t*)(mesh + (size_t)hdr[3] * 4);
  const int32_t* matIndices = (const int32_t*)(mesh + (size_t)hdr[8] * 4);
  const float* shadowRayOff = mesh + (size_t)hdr[13] * 4;

  SurfaceHit sh;
  sh.matId = matIndices[hit.primId];
  const int o = hit.primId * 3;
  const int iA = vertIndices[o], iB = vertIndices[o + 1], iC = vertIndices[o + 2];
  const float* A1 = vertPos + iA * 4; const float* B1 = vertPos + iB * 4; const float* C1 = vertPos + iC * 4;
  const float* A2 = vertNorm + iA * 4; const float* B2 = vertNorm + iB * 4; const float* C2 = vertNorm + iC * 4;
  const f3 A_pos = v3(A1[0], A1[1], A1[2]), B_pos = v3(B1[0], B1[1], B1[2]), C_pos = v3(C1[0], C1[1], C1[2]);
  const f3 A_norm = v3(A2[0], A2[1], A2[2]), B_norm = v3(B2[0], B2[1], B2[2]), C_norm = v3(C2[0], C2[1], C2[2]);
  const f2 A_tex = {A1[3], A2[3]}, B_tex = {B1[3], B2[3]}, C_tex = {C1[3], C2[3]};

  f2 uv;
  {
    const f3 edge1 = sub3(B_pos, A_pos), edge2 = sub3(C_pos, A_pos);
    const f3 pvec = cross3(a_rdir, edge2);
    const float det = dot3(edge1, pvec);
    const float inv_det = 1.0f / det;
    const f3 tvec = sub3(a_rpos, A_pos);
    const float v = dot3(tvec, pvec) * inv_det;
    const f3 qvec = cross3(tvec, edge1);
    const float u = dot3(a_rdir, qvec) * inv_det;
    uv.x = u; uv.y = v;
  }
  const float w0 = (1.0f - uv.x - uv.y);
  sh.pos = add3(add3(scale3(A_pos, w0), scale3(B_pos, uv.y)), scale3(C_pos, uv.x));
  sh.texCoord.x = w0 * A_tex.x + uv.y * B_tex.x + uv.x * C_tex.x;
  sh.texCoord.y = w0 * A_tex.y + uv.y * B_tex.y + uv.x * C_tex.y;
  sh.normal = add3(add3(scale3(A_norm, w0), scale3(B_norm, uv.y)), scale3(C_norm, uv.x));
  sh.t = hit.t;
  sh.sRayOff = shadowRayOff[hit.primId];

  const float* At = vertTang + iA * 4; const float* Bt = vertTang + iB * 4; const float* Ct = vertTang + iC * 4;
  sh.flatNormal = normalize3(cross3(sub3(A_pos, B_pos), sub3(A_pos, C_pos)));
  if (dot3(a_rdir, sh.flatNormal) > 0.025f) sh.flatNormal = scale3(sh.flatNormal, -1.0f);
  const float maxEdge = fmaxf(fmaxf(length3(sub3(A_pos, B_pos)), length3(sub3(A_pos, C_pos))), length3(sub3(B_pos, C_pos)));

  if (sh.sRayOff > 1e-5f * maxEdge) {
    if (dot3(a_rdir, sh.normal) > 0.120f) { sh.normal = scale3(sh.normal, -1.0f); sh.hfi = 1; }
    else if (dot3(a_rdir, sh.normal) > 0.0f) { sh.normal = sh.flatNormal; sh.hfi = 0; }
    else sh.hfi = 0;
  } else {
    if (dot3(a_rdir, sh.normal) > 0.0f) { sh.normal = scale3(sh.normal, -1.0f); sh.hfi = 1; }
    else sh.hfi = 0;
  }
  const float handed = (At[3] < 0.0f || Bt[3] < 0.0f || Ct[3] < 0.0f) ? -1.0f : 1.0f;
  sh.tangent = normalize3(add3(add3(scale3(v3(At[0], At[1], At[2]), w0), scale3(v3(Bt[0], Bt[1], Bt[2]), uv.y)), scale3(v3(Ct[0], Ct[1], Ct[2]), uv.x)));
  sh.biTangent = normalize3(handed > 0.0f ? cross3(sh.normal, sh.tangent) : cross3(sh.tangent, sh.normal));
  const int badTangent = !finite3(sh.biTangent);
  if (fabsf(fabsf(dot3(sh.normal, sh.tangent)) - 1.0f) < 1e-4f || badTangent) CoordinateSystem(sh.normal, &sh.tangent, &sh.biTangent);
  return sh;
}

/* ref: CPUExp_Integrators_PT_Loop.cpp:35-84 kernel_EvalSurface */
static SurfaceHit evalSurface(const OrcScene* s, f3 ray_pos, f3 ray_dir, OrcHit hit) {
  const m44 instInv = load_m44(s->instMatrices + (size_t)hit.instId * 16);
  const f3 posLS = mul4x3(instInv, ray_pos), dirLS = mul3x3(instInv, ray_dir);
  const int meshOffset = s->globals[s->globals[G_GEOM_TABLE] + hit.geomId];   /* ref: cfetch.h:135-139 */
  const float* mesh = s->geomStorage + (size_t)meshOffset * 4;
  const SurfaceHit ls = surfaceEvalLS(posLS, dirLS, hit, mesh);
  const m44 inst = inverse_affine(instInv);
  SurfaceHit ws = ls;
  const float multInv = 1.0f / sqrtf(3.0f);
  const f3 shadowStart = mul3x3(inst, v3(multInv * ws.sRayOff, multInv * ws.sRayOff, multInv * ws.sRayOff));
  const m44 nm = transpose44(instInv);
  ws.pos = mul4x3(inst, ls.pos);
  ws.normal = normalize3(mul3x3(nm, ls.normal));
  ws.flatNormal = normalize3(mul3x3(nm, ls.flatNormal));
  ws.tangent = normalize3(mul3x3(nm, ls.tangent));
  ws.biTangent = normalize3(mul3x3(nm, ls.biTangent));
  ws.t = length3(sub3(ws.pos, ray_pos));
  ws.sRayOff = length3(shadowStart);
  ws.matId = remapMaterialId(ws.matId, hit.instId, s);
  return ws;
}

void orc_eval_surface(const OrcScene* s, int n, const float* pos4, const float* dir4, const OrcHit* hits, float* o) {
  for (int i = 0; i < n; i++) {
    float* r = o + 24 * (size_t)i;
    memset(r, 0, 96);
    if (!HitSome(hits[i])) { r[17] = as_float(-1); continue; }
    const SurfaceHit sh = evalSurface(s, v3(pos4[4 * i], pos4[4 * i + 1], pos4[4 * i + 2]), v3(dir4[4 * i], dir4[4 * i + 1], dir4[4 * i + 2]), hits[i]);
    r[0] = sh.pos.x; r[1] = sh.pos.y; r[2] = sh.pos.z; r[3] = sh.normal.x; r[4] = sh.normal.y; r[5] = sh.normal.z;
    r[6] = sh.flatNormal.x; r[7] = sh.flatNormal.y; r[8] = sh.flatNormal.z; r[9] = sh.tangent.x; r[10] = sh.tangent.y; r[11] = sh.tangent.z;
    r[12] = sh.biTangent.x; r[13] = sh.biTangent.y; r[14] = sh.biTangent.z; r[15] = sh.texCoord.x; r[16] = sh.texCoord.y;
    r[17] = as_float(sh.matId); r[18] = sh.t; r[19] = sh.sRayOff; r[20] = sh.hfi ? 1.0f : 0.0f;
  }
}

/* ------------------------------------------------------------------------------------------------ textures */
/* ref: cfetch.h:312-362 bilinearOffsets */
static void bilinearOffsets(float ffx, float ffy, int flags, int w, int h, int out[4]) {
  const int sx = (ffx > 0.0f) ? 1 : -1, sy = (ffy > 0.0f) ? 1 : -1;
  const int px = (int)(ffx), py = (int)(ffy);
  int px_w0, px_w1, py_w0, py_w1;
  if (flags & TEX_CLAMP_U) {
    px_w0 = (px >= w) ? w - 1 : px; px_w1 = (px + 1 >= w) ? w - 1 : px + 1;
    px_w0 = (px_w0 < 0) ? 0 : px_w0; px_w1 = (px_w1 < 0) ? 0 : px_w1;
  } else {
    px_w0 = px % w; px_w1 = (px + sx) % w;
    px_w0 = (px_w0 < 0) ? px_w0 + w : px_w0; px_w1 = (px_w1 < 0) ? px_w1 + w : px_w1;
  }
  if (flags & TEX_CLAMP_V) {
    py_w0 = (py >= h) ? h - 1 : py; py_w1 = (py + 1 >= h) ? h - 1 : py + 1;
    py_w0 = (py_w0 < 0) ? 0 : py_w0; py_w1 = (py_w1 < 0) ? 0 : py_w1;
  } else {
    py_w0 = py % h; py_w1 = (py + sy) % h;
    py_w0 = (py_w0 < 0) ? py_w0 + h : py_w0; py_w1 = (py_w1 < 0) ? py_w1 + h : py_w1;
  }
  out[0] = py_w0 * w + px_w0; out[1] = py_w0 * w + px_w1; out[2] = py_w1 * w + px_w0; out[3] = py_w1 * w + px_w1;
}
static inline f4 read_uchar4(const uint8_t* data, int offset, int srgb) {   /* ref: cfetch.h:298-303 + sRGBToLinear4f */
  const float mult = 0.003921568f;
  const uint8_t* c = data + (size_t)offset * 4;
  f4 r = {mult * (float)c[0], mult * (float)c[1], mult * (float)c[2], mult * (float)c[3]};
  if (srgb) { r.x = sRGBToLinear(r.x); r.y = sRGBToLinear(r.y); r.z = sRGBToLinear(r.z); r.w = sRGBToLinear(r.w); }
  return r;
}
/* ref: cfetch.h:461-584 read_imagef_sw4 (4-channel textures; depth==1 single-channel variant :364-459 is outside the subset) */
static f4 read_imagef_sw4(const int32_t* tex, f2 tc, int flags, int srgb) {
  const int w = tex[0], h = tex[1], bpp = tex[3];
  const float fw = (float)w, fh = (float)h;
  float ffx = tc.x * fw - 0.5f, ffy = tc.y * fh - 0.5f;
  if ((flags & TEX_CLAMP_U) != 0 && ffx < 0) ffx = 0.0f;
  if ((flags & TEX_CLAMP_V) != 0 && ffy < 0) ffy = 0.0f;
  const uint8_t* bytes = (const uint8_t*)(tex + 4);
  const float* fdata = (const float*)(tex + 4);
  f4 res = {0, 0, 0, 0};
  if (flags & TEX_POINT_SAM) {
    int px = (int)(ffx + 0.5f), py = (int)(ffy + 0.5f);
    if (flags & TEX_CLAMP_U) { px = (px >= w) ? w - 1 : px; px = (px < 0) ? 0 : px; } else { px = px % w; px = (px < 0) ? px + w : px; }
    if (flags & TEX_CLAMP_V) { py = (py >= h) ? h - 1 : py; py = (py < 0) ? 0 : py; } else { py = py % h; py = (py < 0) ? py + h : py; }
    const int offset = py * w + px;
    if (bpp == 4) res = read_uchar4(bytes, offset, srgb);
    else if (bpp == 16) { const float* p = fdata + (size_t)offset * 4; res.x = p[0]; res.y = p[1]; res.z = p[2]; res.w = p[3]; }
    return res;
  }
  const int px = (int)(ffx), py = (int)(ffy);
  const float fx = fabsf(ffx - (float)px), fy = fabsf(ffy - (float)py);
  const float fx1 = 1.0f - fx, fy1 = 1.0f - fy;
  const float w1 = fx1 * fy1, w2 = fx * fy1, w3 = fx1 * fy, w4 = fx * fy;
  int offs[4];
  bilinearOffsets(ffx, ffy, flags, w, h, offs);
  f4 f[4];
  for (int k = 0; k < 4; k++) {
    if (bpp == 4) f[k] = read_uchar4(bytes, offs[k], srgb);
    else { const float* p = fdata + (size_t)offs[k] * 4; f[k].x = p[0]; f[k].y = p[1]; f[k].z = p[2]; f[k].w = p[3]; }
  }
  res.x = f[0].x * w1 + f[1].x * w2 + f[2].x * w3 + f[3].x * w4;
  res.y = f[0].y * w1 + f[1].y * w2 + f[2].y * w3 + f[3].y * w4;
  res.z = f[0].z * w1 + f[1].z * w2 + f[2].z * w3 + f[3].z * w4;
  res.w = f[0].w * w1 + f[1].w * w2 + f[2].w * w3 + f[3].w * w4;
  return res;
}
/* ---- procedural textures.  The CPU integrator evaluates none (IntegratorCommon hands materialEval a dummy list, CPUExp_Integrators_Common.cpp:91-93); the OpenCL
 * layer runs the scene's own functions in ProcTexExec (shaders/texproc.cl) and hands every ray a ProcTextureList (cglobals.h:2312-2318) whose colours went through
 * half precision (WriteProcTextureList / ReadProcTextureList, :2327-2396).  The oracle restates the CONSUMER: readProcTex (:2402-2441) inside sample2DExt /
 * sample2DAuxExt, over the list of the point being shaded -- handed in by the test (orc_stage_set_proctex: the lists the reference's ProcTexExec wrote) -- kept in a
 * thread-local pointer because the reference threads it through every material function as an argument. */
static _Thread_local float g_matteShadow = 0.0f;   /* the shadow value a shadow catcher is sampled with (see MT_SHADOW_MATTE) */
static int orc_have_back_plate(const OrcScene* s);
typedef struct { int n; int ids[16]; float vals[16][3]; } OrcPtl;
static _Thread_local const OrcPtl* g_ptl = NULL;
static inline f4 readProcTex(int texId) {
  f4 r = {1.0f, 1.0f, 1.0f, -1.0f};
  if (g_ptl == NULL) return r;
  for (int k = 0; k < g_ptl->n && k < 16; k++)
    if (g_ptl->ids[k] == texId) { r.x = g_ptl->vals[k][0]; r.y = g_ptl->vals[k][1]; r.z = g_ptl->vals[k][2]; r.w = 0.0f; break; }   /* ids are unique in a list: first = only match */
  return r;
}
static float half_bits_to_float(unsigned h) {
  const unsigned sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
  uint32_t bits;
  if (e == 0) { const float v = (float)m * 5.9604644775390625e-08f; return sign ? -v : v; }
  if (e == 31) bits = sign | 0x7f800000u | (m << 13); else bits = sign | ((e + 112u) << 23) | (m << 13);
  float f; memcpy(&f, &bits, 4); return f;
}
static int g_stagePtlN = 0, g_stagePtlMax = 0;
static int* g_stagePtlIds = NULL;
static uint16_t* g_stagePtlHalfs = NULL;
/* the lists the following orc_stage_bounce / orc_stage_shade_point calls of the same n consult: ids[max_num][n], halfs[max_num][n][4]; n = 0 drops them */
void orc_stage_set_proctex(int n, int max_num, const int* ids, const uint16_t* halfs4) {
  free(g_stagePtlIds); free(g_stagePtlHalfs); g_stagePtlIds = NULL; g_stagePtlHalfs = NULL; g_stagePtlN = 0; g_stagePtlMax = 0;
  if (n <= 0 || max_num <= 0 || !ids || !halfs4) return;
  g_stagePtlIds = (int*)malloc(sizeof(int) * (size_t)n * (size_t)max_num);
  g_stagePtlHalfs = (uint16_t*)malloc(sizeof(uint16_t) * 4 * (size_t)n * (size_t)max_num);
  memcpy(g_stagePtlIds, ids, sizeof(int) * (size_t)n * (size_t)max_num);
  memcpy(g_stagePtlHalfs, halfs4, sizeof(uint16_t) * 4 * (size_t)n * (size_t)max_num);
  g_stagePtlN = n; g_stagePtlMax = max_num > 16 ? 16 : max_num;
}
/* ReadProcTextureList for point i of a stage call over n points: NULL when no lists were handed in for this n */
static const OrcPtl* stage_ptl(int n, int i, OrcPtl* tmp) {
  if (g_stagePtlN != n || g_stagePtlIds == NULL) return NULL;
  tmp->n = 0;
  for (int k = 0; k < g_stagePtlMax; k++) {
    const int id = g_stagePtlIds[(size_t)k * n + i];
    if ((uint32_t)id == INVALID_TEXTURE) break;
    const uint16_t* h = g_stagePtlHalfs + ((size_t)k * n + i) * 4;
    tmp->ids[k] = id; tmp->vals[k][0] = half_bits_to_float(h[0]); tmp->vals[k][1] = half_bits_to_float(h[1]); tmp->vals[k][2] = half_bits_to_float(h[2]);
    tmp->n = k + 1;
  }
  return tmp;
}

/* The scene's own texture functions are user code, not reference code: for whole-frame comparisons the test builds them for the host (tests/proctex_host.py) and
 * hands the oracle the entry; PathTrace then does what ProcTexExec does for a hit whose material head carries PLAIN_MATERIAL_HAVE_PROC_TEXTURES (texproc.cl:127-190):
 * world and local position, frame, texture coordinate, AO inputs = 1, the ray direction as `hr_viewVectorHack`; colours go through half precision as in
 * WriteProcTextureList / ReadProcTextureList. */
typedef void (*orc_proctex_fn)(void* user, const float* surf19, const float* head, const float* view3, int* count, int* ids16, float* vals48);
static orc_proctex_fn g_proctexFn = NULL;
static void* g_proctexUser = NULL;
void orc_set_proctex_eval(void* fn, void* user) { g_proctexFn = (orc_proctex_fn)fn; g_proctexUser = user; }
static unsigned float_to_half_bits(float f) {   /* round to nearest even (vstore_half) */
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const int32_t e = (int32_t)((x >> 23) & 0xffu) - 127 + 15;
  uint32_t m = x & 0x7fffffu;
  if (((x >> 23) & 0xffu) == 0xffu) return sign | 0x7c00u | (m ? 0x200u : 0u);
  if (e >= 31) return sign | 0x7c00u;
  if (e <= 0) {
    if (e < -10) return sign;
    m |= 0x800000u;
    const int shift = 14 - e;
    uint32_t h = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;
    return sign | h;
  }
  uint32_t h = ((uint32_t)e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
  return sign | h;
}
enum { MF_HAVE_PROC_TEXTURES = 65536 };   /* cglobals.h:2647 */

/* ref: cfetch.h:677-709 sample2DExt.
 * blob = the material / light the sampler is embedded in, addressed in int4 units. */
static f3 sample2DExt(int samplerOffset, f2 texCoord, const float* blob, const OrcScene* s) {
  if ((uint32_t)samplerOffset == INVALID_TEXTURE || samplerOffset < 0) return v3(1, 1, 1);
  const float* sm = blob + (size_t)samplerOffset * 4;
  const int flags = as_int(sm[0]); const float gamma = sm[1]; const int texId = as_int(sm[2]);
  if (texId <= 0) return v3(1, 1, 1);
  f2 tct;   /* mul2x4, cfetch.h:642-648 */
  tct.x = sm[4] * texCoord.x + sm[5] * texCoord.y + sm[7];
  tct.y = sm[8] * texCoord.x + sm[9] * texCoord.y + sm[11];
  const int offset = s->globals[s->globals[G_TEX_TABLE] + texId];
  f4 c;
  if (offset >= 0) c = read_imagef_sw4(s->texStorage + (size_t)offset * 4, tct, flags, (gamma != 1.0f));
  else { c.x = c.y = c.z = c.w = 1.0f; }
  const f4 c1 = readProcTex(texId);
  if (!(fabsf(c1.w + 1.0f) < 1e-5f)) c = c1;
  if (flags & TEX_ALPHASRC_W) { c.x = c.w; c.y = c.w; c.z = c.w; }
  return v3(c.x, c.y, c.z);
}
/* ---- normal maps: sample2DAuxExt cfetch.h:795-820 (no procedural textures), materialNormalMapFetch cmaterial.h:2208-2233,
 * BumpMapping :2235-2243 with make_float3x3 / inverse / mul3x3x3 of cglobals.h:849-918 ---- */
enum { NORMAL_TEX_OFFSET = 83, NORMAL_TEX_MATRIX = 84, G_TEXAUX_TABLE = 222, MF_INVERT_NMAP_X = 16, MF_INVERT_NMAP_Y = 32, MF_INVERT_SWAP_NMAP_XY = 64 };   /* cglobals.h:2631-2633, 2681-2682 */
static inline int hasNormalMap(const float* m) { return (uint32_t)as_int(m[NORMAL_TEX_OFFSET]) != INVALID_TEXTURE; }
static f3 sample2DAuxExt(int auxTexId, int samplerOffset, f2 texCoord, const float* blob, const OrcScene* s) {
  if ((uint32_t)samplerOffset == INVALID_TEXTURE) return v3(1, 1, 1);
  const float* sm = blob + (size_t)samplerOffset * 4;
  const int flags = as_int(sm[0]); const float gamma = sm[1]; const int texId = as_int(sm[2]);
  f2 tct;
  tct.x = sm[4] * texCoord.x + sm[5] * texCoord.y + sm[7];
  tct.y = sm[8] * texCoord.x + sm[9] * texCoord.y + sm[11];
  if (texId == 0) return v3(1, 1, 1);
  f4 c = readProcTex(texId);   /* a procedural normal map keeps its texture id in the aux slot (PlainMaterialConverter.cpp:1396-1399): the aux arena holds nothing for it and is not read */
  if (fabsf(c.w + 1.0f) < 1e-5f) {
    const int offset = s->globals[s->globals[G_TEXAUX_TABLE] + auxTexId];
    c = read_imagef_sw4(s->texAuxStorage + (size_t)offset * 4, tct, flags, (gamma != 1.0f));
  }
  if (flags & TEX_ALPHASRC_W) { c.x = c.w; c.y = c.w; c.z = c.w; }
  return v3(c.x, c.y, c.z);
}
static f3 materialNormalMapFetch(const float* m, f2 tc, const OrcScene* s) {
  const int flags = as_int(m[1]);
  const f3 t = sample2DAuxExt(as_int(m[NORMAL_TEX_OFFSET]), as_int(m[NORMAL_TEX_MATRIX]), tc, m, s);
  f3 normalTS = v3(2.0f * t.x - 1.0f, 2.0f * t.y - 1.0f, t.z);
  if (flags & MF_INVERT_NMAP_Y) normalTS.y *= (-1.0f);
  if (flags & MF_INVERT_NMAP_X) normalTS.x *= (-1.0f);
  if (flags & MF_INVERT_SWAP_NMAP_XY) { const float tmp = normalTS.x; normalTS.x = normalTS.y; normalTS.y = tmp; }
  return normalize3(normalTS);
}
static f3 BumpMapping(f3 tangent, f3 bitangent, f3 normal, f2 tc, const float* m, const OrcScene* s) {
  const f3 nts = materialNormalMapFetch(m, tc, s);
  const f3 r0 = tangent, r1 = bitangent, r2 = normal;
  const float det = r0.x * (r1.y * r2.z - r1.z * r2.y) - r0.y * (r1.x * r2.z - r1.z * r2.x) + r0.z * (r1.x * r2.y - r1.y * r2.x);
  f3 b0 = v3((r1.y * r2.z - r1.z * r2.y), -(r0.y * r2.z - r0.z * r2.y), (r0.y * r1.z - r0.z * r1.y));
  f3 b1 = v3(-(r1.x * r2.z - r1.z * r2.x), (r0.x * r2.z - r0.z * r2.x), -(r0.x * r1.z - r0.z * r1.x));
  f3 b2 = v3((r1.x * r2.y - r1.y * r2.x), -(r0.x * r2.y - r0.y * r2.x), (r0.x * r1.y - r0.y * r1.x));
  const float sc = 1.0f / det;
  b0 = scale3(b0, sc); b1 = scale3(b1, sc); b2 = scale3(b2, sc);
  return normalize3(v3(b0.x * nts.x + b0.y * nts.y + b0.z * nts.z, b1.x * nts.x + b1.y * nts.y + b1.z * nts.z, b2.x * nts.x + b2.y * nts.y + b2.z * nts.z));
}
/* sampler flags for the TEX_COORD_CAM_PROJ test in lambert (cmaterial.h:224-226, 240-242).  With no texture the
 * reference reads the sampler at int4 index -2 (out of the node); that word is zero for every material after the
 * first arena entry, so the flags are taken as 0 here. */
static inline int samplerFlagsOrZero(int samplerOffset, const float* blob) {
  if ((uint32_t)samplerOffset == INVALID_TEXTURE || samplerOffset < 0) return 0;
  return as_int(blob[(size_t)samplerOffset * 4]);
}

/* ------------------------------------------------------------------------------------------------ materials */
static inline const float* materialAt(const OrcScene* s, int matId) {   /* ref: cfetch.h:192-213 */
  if (matId == -1) return NULL;
  const int matOffset = s->globals[s->globals[G_MAT_TABLE] + matId];
  return s->matStorage + (size_t)matOffset * 4;
}
static inline int matType(const float* m) { return as_int(m[MAT_TYPE]); }
static inline int matFlags(const float* m) { return as_int(m[MAT_FLAGS]); }
static inline f3 matColor(const float* m) { return v3(m[MAT_COLOR], m[MAT_COLOR + 1], m[MAT_COLOR + 2]); }

typedef struct { f3 color; f3 direction; float pdf; int flags; } MatSample;
typedef struct { f3 brdf; float pdfFwd; f3 btdf; float pdfRev; int diffuse; } BxDFResult;
typedef struct { f3 l, v, n, fn, tg, bn; f2 tc; } ShadeContext;

/* ref: cmaterial.h:435-466 glosscoeff + cosPowerFromGlosiness */
static const float glosscoeff[10][4] = {
    {8.88178419700125e-14f, -1.77635683940025e-14f, 5.0f, 1.0f},
    {357.142857142857f, -35.7142857142857f, 5.0f, 1.5f},
    {-2142.85714285714f, 428.571428571429f, 8.57142857142857f, 2.0f},
    {428.571428571431f, -42.8571428571432f, 30.0f, 5.0f},
    {2095.23809523810f, -152.380952380952f, 34.2857142857143f, 8.0f},
    {-4761.90476190476f, 1809.52380952381f, 66.6666666666667f, 12.0f},
    {9914.71215351811f, 1151.38592750533f, 285.714285714286f, 32.0f},
    {45037.7068059246f, 9161.90096119855f, 813.432835820895f, 82.0f},
    {167903.678757035f, 183240.189801913f, 3996.94423223835f, 300.0f},
    {-20281790.7444668f, 6301358.14889336f, 45682.0925553320f, 2700.0f}};
static float cosPowerFromGlosiness(float glosiness) {
  const float cMax = 1000000.0f;
  const float x = glosiness;
  const int k = (fabsf(x - 1.0f) < 1e-5f) ? 10 : (int)(x * 10.0f);
  const float x1 = (x - (float)(k)*0.1f);
  if (k == 10 || x >= 0.99f) return cMax;
  return glosscoeff[k][3] + glosscoeff[k][2] * x1 + glosscoeff[k][1] * x1 * x1 + glosscoeff[k][0] * x1 * x1 * x1;
}

/* ---- lambert, ref: cmaterial.h:219-263 */
static inline float lambertEvalPDF(f3 l, f3 n) { return fabsf(dot3(l, n)) * INV_PI; }
static f3 lambertEvalBxDF(const float* m, f2 tc, const OrcScene* s) {
  const int so = as_int(m[MAT_TEXMATRIXID]);
  (void)samplerFlagsOrZero(so, m);   /* TEX_COORD_CAM_PROJ: camera-projected coords are outside the subset */
  const f3 tex = sample2DExt(so, tc, m, s);
  return scale3(clamp3(mul3(tex, matColor(m)), 0.0f, 1.0f), INV_PI);
}
static void LambertSampleAndEvalBRDF(const float* m, float r1, float r2, f3 n, f2 tc, const OrcScene* s, MatSample* out) {
  const f3 tex = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  const f3 color = clamp3(mul3(tex, matColor(m)), 0.0f, 1.0f);
  const f3 newDir = MapSampleToCosineDistribution(r1, r2, n, n, 1.0f);
  const float cosTheta = dot3(newDir, n);
  out->direction = newDir;
  out->pdf = cosTheta * INV_PI;
  out->color = scale3(color, INV_PI);
  if (cosTheta <= DEPSILON) out->color = v3(0, 0, 0);
  out->flags = RAY_EVENT_D;
}
static float phongGlosiness(const float* m, f2 tc, const OrcScene* s);
/* ---- Blinn distribution in a Torrance-Sparrow model, ref: cmaterial.h:1020-1168 (offsets = phong's), cmatpbrt.h:33-103 */
static float TorranceSparrowG1(f3 wo, f3 wi, f3 wh) {
  const float NdotWh = fabsf(wh.z), NdotWo = fabsf(wo.z), NdotWi = fabsf(wi.z);
  const float WOdotWh = fmaxf(fabsf(dot3(wo, wh)), DEPSILON);
  return fminf(1.f, fminf((2.f * NdotWh * NdotWo / WOdotWh), (2.f * NdotWh * NdotWi / WOdotWh)));
}
static float TorranceSparrowGF1(f3 wo, f3 wi) {
  const float cosThetaO = fabsf(wo.z), cosThetaI = fabsf(wi.z);
  if (cosThetaI == 0.0f || cosThetaO == 0.0f) return 0.0f;
  f3 wh = add3(wi, wo);
  if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return 0.0f;
  wh = normalize3(wh);
  const float F = 1.0f;
  return fminf(TorranceSparrowG1(wo, wi, wh) * F / fmaxf(4.0f * cosThetaI * cosThetaO, DEPSILON), 250.0f);
}
static float TorranceSparrowGF2(f3 wo, f3 wi, f3 n) {
  const float cosThetaO = fabsf(dot3(wo, n)), cosThetaI = fabsf(dot3(wi, n));
  if (cosThetaI == 0.f || cosThetaO == 0.0f) return 0.0f;
  f3 wh = add3(wi, wo);
  if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return 0.0f;
  wh = normalize3(wh);
  const float NdotWh = fabsf(dot3(wh, n)), WOdotWh = fmaxf(fabsf(dot3(wo, wh)), DEPSILON);
  const float G = fminf(1.f, fminf((2.f * NdotWh * cosThetaO / WOdotWh), (2.f * NdotWh * cosThetaI / WOdotWh)));
  const float F = 1.0f;
  return fminf(G * F / fmaxf(4.0f * cosThetaI * cosThetaO, DEPSILON), 10.0f);
}
static float blinnEvalPDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const OrcScene* s) {
  if (dot3(n, v) < 1e-6f || dot3(n, l) < 1e-6f) return 1.0f;
  const float exponent = cosPowerFromGlosiness(phongGlosiness(m, tc, s));
  const f3 wh = normalize3(add3(l, v));
  const float costheta = fabsf(dot3(wh, n));
  return ((exponent + 1.0f) * powf(costheta, exponent)) / (M_TWOPI_F * 4.0f * dot3(l, wh));
}
static f3 blinnEvalBxDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const OrcScene* s) {
  if (dot3(n, v) < 1e-6f || dot3(n, l) < 1e-6f) return v3(0, 0, 0);
  const f3 color = clamp3(mul3(matColor(m), sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s)), 0.0f, 1.0f);
  const float exponent = cosPowerFromGlosiness(phongGlosiness(m, tc, s));
  const f3 wh = normalize3(add3(l, v));
  const float D = (exponent + 2.0f) * INV_TWOPI * powf(fabsf(dot3(wh, n)), exponent);
  return scale3(scale3(color, D), TorranceSparrowGF2(l, v, n));
}
static void BlinnSampleAndEvalBRDF(const float* m, float r1, float r2, f3 ray_dir, f3 n, f2 tc, const OrcScene* s, MatSample* out) {
  const f3 color = clamp3(mul3(matColor(m), sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s)), 0.0f, 1.0f);
  const float gloss = phongGlosiness(m, tc, s);
  f3 nx, ny;
  const f3 nz = n;
  CoordinateSystem(nz, &nx, &ny);
  const f3 wo = v3(-dot3(ray_dir, nx), -dot3(ray_dir, ny), -dot3(ray_dir, nz));
  const float exponent = cosPowerFromGlosiness(gloss);
  const float costheta = powf(r1, 1.0f / (exponent + 1.0f));
  const float sintheta = sqrtf(fmaxf(0.0f, 1.0f - costheta * costheta));
  const float phi = r2 * M_TWOPI_F;
  const f3 wh = v3(sintheta * cosf(phi), sintheta * sinf(phi), costheta);
  const f3 wi = sub3(scale3(wh, 2.0f * dot3(wo, wh)), wo);
  const f3 newDir = normalize3(add3(add3(scale3(nx, wi.x), scale3(ny, wi.y)), scale3(nz, wi.z)));
  const f3 v = scale3(ray_dir, -1.0f);
  if (dot3(n, v) < 1e-6f || dot3(n, newDir) < 1e-6f) { out->color = v3(0, 0, 0); out->pdf = 1.0f; }
  else {
    const float D = ((exponent + 2.0f) * INV_TWOPI * powf(costheta, exponent));
    out->color = scale3(scale3(color, D), TorranceSparrowGF1(wo, wi));
    out->pdf = ((exponent + 1.0f) * powf(costheta, exponent)) / fmaxf(M_TWOPI_F * 4.0f * dot3(wo, wh), DEPSILON);
  }
  out->direction = newDir;
  out->flags = (gloss >= 0.99f) ? RAY_EVENT_S : RAY_EVENT_G;
}
/* ---- anisotropic microfacet lobes: PBRT v3's Beckmann and Trowbridge-Reitz distributions (cmatpbrt.h:105-540) behind the wrappers of
 * cmaterial.h:1558-1846; one node layout for both classes (BECKMANN_* offsets, :1531-1556) */
enum { BECKMANN_ANISOTROPY = 19, BECKMANN_ANISO_ROT = 68, BECKMANN_ANISO_TEXMATRIXID = 70, BECKMANN_ROT_TEXMATRIXID = 72 };
static inline float Cos2ThetaPBRT(f3 w) { return w.z * w.z; }
static inline float AbsCosThetaPBRT(f3 w) { return fabsf(w.z); }
static inline float Sin2ThetaPBRT(f3 w) { return fmaxf(0.0f, 1.0f - Cos2ThetaPBRT(w)); }
static inline float SinThetaPBRT(f3 w) { return sqrtf(Sin2ThetaPBRT(w)); }
static inline float TanThetaPBRT(f3 w) { return (fabsf(w.z) < 1e-6f) ? 0.0f : SinThetaPBRT(w) / w.z; }
static inline float Tan2ThetaPBRT(f3 w) { return Sin2ThetaPBRT(w) / fmaxf(Cos2ThetaPBRT(w), 1e-6f); }
static inline float CosPhiPBRT(f3 w) { const float st = SinThetaPBRT(w); return (st == 0.0f) ? 1.0f : clampf(w.x / st, -1.0f, 1.0f); }
static inline float SinPhiPBRT(f3 w) { const float st = SinThetaPBRT(w); return (st == 0.0f) ? 0.0f : clampf(w.y / st, -1.0f, 1.0f); }
static inline float Cos2PhiPBRT(f3 w) { return CosPhiPBRT(w) * CosPhiPBRT(w); }
static inline float Sin2PhiPBRT(f3 w) { return SinPhiPBRT(w) * SinPhiPBRT(w); }
static float ErfPBRT(float x) {
  const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
  int sign = 1;
  if (x < 0.0f) sign = -1;
  x = fabsf(x);
  const float t = 1.0f / (1.0f + p * x);
  const float y = 1.0f - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * expf(-x * x);
  return sign * y;
}
static float ErfInvPBRT(float x) {
  float w, p;
  x = clampf(x, -0.99999f, 0.99999f);
  w = -logf((1.0f - x) * (1.0f + x));
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f; p = 3.43273939e-07f + p * w; p = -3.5233877e-06f + p * w; p = -4.39150654e-06f + p * w; p = 0.00021858087f + p * w;
    p = -0.00125372503f + p * w; p = -0.00417768164f + p * w; p = 0.246640727f + p * w; p = 1.50140941f + p * w;
  } else {
    w = sqrtf(w) - 3.0f;
    p = -0.000200214257f; p = 0.000100950558f + p * w; p = 0.00134934322f + p * w; p = -0.00367342844f + p * w; p = 0.00573950773f + p * w;
    p = -0.0076224613f + p * w; p = 0.00943887047f + p * w; p = 1.00167406f + p * w; p = 2.83297682f + p * w;
  }
  return p * x;
}
static float BeckmannDistributionD(f3 wh, float ax, float ay) {
  const float tan2Theta = Tan2ThetaPBRT(wh), cos4Theta = Cos2ThetaPBRT(wh) * Cos2ThetaPBRT(wh);
  return expf((-1.0f) * tan2Theta * (Cos2PhiPBRT(wh) / fmaxf(ax * ax, 1e-6f) + Sin2PhiPBRT(wh) / fmaxf(ay * ay, 1e-6f))) / fmaxf(M_PI_F * ax * ay * cos4Theta, 1e-6f);
}
static float BeckmannDistributionLambda(f3 w, float ax, float ay) {
  const float absTanTheta = fabsf(TanThetaPBRT(w));
  if (!isfinite(absTanTheta) || absTanTheta == 0.0f) return 0.0f;
  const float alpha = sqrtf(fmaxf(Cos2PhiPBRT(w) * ax * ax + Sin2PhiPBRT(w) * ay * ay, 1e-6f));
  const float a = 1.0f / fmaxf(alpha * absTanTheta, 1e-6f);
  if (a >= 1.6f) return 0.0f;
  return (1.0f - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
}
static void BeckmannSample11(float cosThetaI, float U1, float U2, float* slope_x, float* slope_y) {
  if (cosThetaI > 0.9999f) {
    const float r = sqrtf(logf(1.0f - U1) * (-1.0f));
    const float sinPhi = sinf(M_TWOPI_F * U2), cosPhi = cosf(M_TWOPI_F * U2);
    *slope_x = r * cosPhi; *slope_y = r * sinPhi;
    return;
  }
  const float sinThetaI = sqrtf(fmaxf(0.0f, 1.0f - cosThetaI * cosThetaI));
  const float tanThetaI = sinThetaI / fmaxf(cosThetaI, 1e-6f);
  const float cotThetaI = 1.0f / fmaxf(tanThetaI, 1e-6f);
  float a = -1.0f;
  float c = ErfPBRT(cotThetaI);
  const float sample_x = fmaxf(U1, 1e-6f);
  const float thetaI = acosf(cosThetaI);
  const float fit = 1.0f + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
  float b = c - (1.0f + c) * powf(1.0f - sample_x, fit);
  const float SQRT_PI_INV = 1.0f / sqrtf(M_PI_F);
  const float normalization = 1.0f / fmaxf(1.0f + c + SQRT_PI_INV * tanThetaI * expf((-1.0f) * cotThetaI * cotThetaI), 1e-6f);
  int it = 0;
  while (++it < 10) {
    if (!(b >= a && b <= c)) b = 0.5f * (a + c);
    const float invErf = ErfInvPBRT(b);
    const float value = normalization * (1.0f + b + SQRT_PI_INV * tanThetaI * expf((-1.0f) * invErf * invErf)) - sample_x;
    const float derivative = normalization * (1.0f - invErf * tanThetaI);
    if (fabsf(value) < 1e-5f) break;
    if (value > 0.0f) c = b; else a = b;
    b -= value / fmaxf(derivative, 1e-6f);
  }
  *slope_x = ErfInvPBRT(b);
  *slope_y = ErfInvPBRT(2.0f * fmaxf(U2, 1e-6f) - 1.0f);
}
static void TrowbridgeReitzSample11(float cosTheta, float U1, float U2, float* slope_x, float* slope_y) {
  if (cosTheta > 0.9999f) {
    const float r = sqrtf(U1 / fmaxf(1.0f - U1, 1e-6f));
    const float phi = M_TWOPI_F * U2;
    *slope_x = r * cosf(phi); *slope_y = r * sinf(phi);
    return;
  }
  const float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
  const float tanTheta = sinTheta / cosTheta;
  const float a = 1.0f / tanTheta;
  const float G1 = 2.0f / (1.0f + sqrtf(1.0f + 1.0f / (a * a)));
  const float A = 2.0f * U1 / G1 - 1.0f;
  float tmp = 1.0f / (A * A - 1.0f);
  if (tmp > 1e10f) tmp = 1e10f;
  const float B = tanTheta;
  const float D = sqrtf(fmaxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.0f));
  const float slope_x_1 = B * tmp - D, slope_x_2 = B * tmp + D;
  *slope_x = (A < 0.0f || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
  float S;
  if (U2 > 0.5f) { S = 1.0f; U2 = 2.0f * (U2 - 0.5f); } else { S = -1.0f; U2 = 2.0f * (0.5f - U2); }
  const float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) / (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
  *slope_y = S * z * sqrtf(1.0f + *slope_x * *slope_x);
}
static f3 microfacetSampleWH(int TR, f3 wo, float u1, float u2, float ax, float ay) {
  const int flip = (wo.z < 0.0f);
  const f3 wi = flip ? scale3(wo, -1.0f) : wo;
  const f3 wiStretched = normalize3(v3(ax * wi.x, ay * wi.y, wi.z));
  float slope_x, slope_y;
  if (TR) TrowbridgeReitzSample11(wiStretched.z, u1, u2, &slope_x, &slope_y); else BeckmannSample11(wiStretched.z, u1, u2, &slope_x, &slope_y);
  const float tmp = CosPhiPBRT(wiStretched) * slope_x - SinPhiPBRT(wiStretched) * slope_y;
  slope_y = SinPhiPBRT(wiStretched) * slope_x + CosPhiPBRT(wiStretched) * slope_y;
  slope_x = tmp;
  slope_x = ax * slope_x;
  slope_y = ay * slope_y;
  f3 wh = normalize3(v3(slope_x * (-1.0f), slope_y * (-1.0f), 1.0f));
  if (flip) wh = scale3(wh, -1.0f);
  return wh;
}
static float TrowbridgeReitzDistributionD(f3 wh, float ax, float ay) {
  const float tan2Theta = Tan2ThetaPBRT(wh);
  if (!isfinite(tan2Theta)) return 0.0f;
  const float cos4Theta = Cos2ThetaPBRT(wh) * Cos2ThetaPBRT(wh);
  const float e = (Cos2PhiPBRT(wh) / (ax * ax) + Sin2PhiPBRT(wh) / (ay * ay)) * tan2Theta;
  return 1.0f / (M_PI_F * ax * ay * cos4Theta * (1.0f + e) * (1.0f + e));
}
static float TrowbridgeReitzDistributionLambda(f3 w, float ax, float ay) {
  const float absTanTheta = fabsf(TanThetaPBRT(w));
  if (!isfinite(absTanTheta)) return 0.0f;
  const float alpha = sqrtf(Cos2PhiPBRT(w) * ax * ax + Sin2PhiPBRT(w) * ay * ay);
  const float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
  return (-1.0f + sqrtf(1.0f + alpha2Tan2Theta)) / 2.0f;
}
static float microfacetD(int TR, f3 wh, float ax, float ay) { return TR ? TrowbridgeReitzDistributionD(wh, ax, ay) : BeckmannDistributionD(wh, ax, ay); }
static float microfacetLambda(int TR, f3 w, float ax, float ay) { return TR ? TrowbridgeReitzDistributionLambda(w, ax, ay) : BeckmannDistributionLambda(w, ax, ay); }
static float microfacetPdf(int TR, f3 wo, f3 wh, float ax, float ay) {
  return microfacetD(TR, wh, ax, ay) * (1.0f / (1.0f + microfacetLambda(TR, wo, ax, ay))) / fmaxf(4.0f * AbsCosThetaPBRT(wo), 1e-6f);
}
static float microfacetBRDF_PBRT(int TR, f3 wo, f3 wi, float ax, float ay) {
  const float cosThetaO = AbsCosThetaPBRT(wo), cosThetaI = AbsCosThetaPBRT(wi);
  f3 wh = add3(wi, wo);
  if (cosThetaI <= 1e-6f || cosThetaO <= 1e-6f) return 0.0f;
  if (fabsf(wh.x) <= 1e-6f && fabsf(wh.y) <= 1e-6f && fabsf(wh.z) <= 1e-6f) return 0.0f;
  wh = normalize3(wh);
  const float G = 1.0f / (1.0f + microfacetLambda(TR, wo, ax, ay) + microfacetLambda(TR, wi, ax, ay));
  return microfacetD(TR, wh, ax, ay) * G / fmaxf(4.0f * cosThetaI * cosThetaO, 1e-6f);   /* F = 1 in both forms */
}
static float BeckmannRoughnessToAlpha(float roughness) {
  const float x = logf(fmaxf(roughness, 1.0e-4f));
  return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
static f2 beckmannAlphaXY(const float* m, f2 tc, const OrcScene* s) {
  const float roughness = 0.5f - 0.5f * phongGlosiness(m, tc, s);   /* beckmannGlosiness: the phong offsets */
  const f3 ac = sample2DExt(as_int(m[BECKMANN_ANISO_TEXMATRIXID]), tc, m, s);
  const float anisoMult = 1.0f - clampf(m[BECKMANN_ANISOTROPY] * fmaxf(ac.x, fmaxf(ac.y, ac.z)), 0.0f, 1.0f);
  f2 r; r.x = BeckmannRoughnessToAlpha(roughness * roughness); r.y = BeckmannRoughnessToAlpha(roughness * roughness * anisoMult * anisoMult);
  return r;
}
static void BeckmanTangentSpace(const float* m, f2 alpha, f3 nz, f3 a_tan, f3 a_bitan, f2 tc, const OrcScene* s, f3* pNx, f3* pNy) {
  if (fabsf(alpha.x - alpha.y) > 1e-5f) {
    *pNx = a_bitan; *pNy = a_tan;
    const f3 rc = sample2DExt(as_int(m[BECKMANN_ROT_TEXMATRIXID]), tc, m, s);
    const float rotVal = clampf(m[BECKMANN_ANISO_ROT] * fmaxf(rc.x, fmaxf(rc.y, rc.z)), 0.0f, 1.0f);
    const float rotAngle = rotVal * M_TWOPI_F;
    const float cos_t = cosf(rotAngle), sin_t = sinf(rotAngle);
    const f3 v = nz;   /* RotateAroundVector4x4 (cglobals.h:1122-1150) through mul3x3 (:297-304) */
    const f3 r0 = v3((1.0f - cos_t) * v.x * v.x + cos_t, (1.0f - cos_t) * v.x * v.y - sin_t * v.z, (1.0f - cos_t) * v.x * v.z + sin_t * v.y);
    const f3 r1 = v3((1.0f - cos_t) * v.y * v.x + sin_t * v.z, (1.0f - cos_t) * v.y * v.y + cos_t, (1.0f - cos_t) * v.y * v.z - sin_t * v.x);
    const f3 r2 = v3((1.0f - cos_t) * v.x * v.z - sin_t * v.y, (1.0f - cos_t) * v.z * v.y + sin_t * v.x, (1.0f - cos_t) * v.z * v.z + cos_t);
    const f3 px = *pNx, py = *pNy;
    *pNx = v3(px.x * r0.x + px.y * r0.y + px.z * r0.z, px.x * r1.x + px.y * r1.y + px.z * r1.z, px.x * r2.x + px.y * r2.y + px.z * r2.z);
    *pNy = v3(py.x * r0.x + py.y * r0.y + py.z * r0.z, py.x * r1.x + py.y * r1.y + py.z * r1.z, py.x * r2.x + py.y * r2.y + py.z * r2.z);
  } else
    CoordinateSystem(nz, pNx, pNy);
  if ((matFlags(m) & MF_FLIP_TANGENT) != 0) { const f3 t = *pNx; *pNx = *pNy; *pNy = t; }
}
static float anisoEvalPDF(int TR, const float* m, f3 l, f3 v, f3 n, f3 a_tan, f3 a_bitan, f2 tc, const OrcScene* s) {
  if (dot3(n, v) < 1e-6f || dot3(n, l) < 1e-6f) return 1.0f;
  const f2 alpha = beckmannAlphaXY(m, tc, s);
  f3 nx, ny;
  BeckmanTangentSpace(m, alpha, n, a_tan, a_bitan, tc, s, &nx, &ny);
  const f3 wo = v3(-dot3(v, nx), -dot3(v, ny), -dot3(v, n));
  const f3 wh = normalize3(add3(l, v));
  return microfacetPdf(TR, wo, wh, alpha.x, alpha.y);
}
static f3 anisoEvalBxDF(int TR, const float* m, f3 l, f3 v, f3 n, f3 a_tan, f3 a_bitan, f2 tc, const OrcScene* s) {
  if (dot3(n, v) < 1e-6f || dot3(n, l) < 1e-6f) return v3(0, 0, 0);
  const f3 color = clamp3(mul3(sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s), matColor(m)), 0.0f, 1.0f);
  const f2 alpha = beckmannAlphaXY(m, tc, s);
  f3 nx, ny;
  BeckmanTangentSpace(m, alpha, n, a_tan, a_bitan, tc, s, &nx, &ny);
  const f3 wo = v3(-dot3(v, nx), -dot3(v, ny), -dot3(v, n)), wi = v3(-dot3(l, nx), -dot3(l, ny), -dot3(l, n));
  return scale3(color, microfacetBRDF_PBRT(TR, wo, wi, alpha.x, alpha.y));
}
static void AnisoSampleAndEvalBRDF(int TR, const float* m, float r1, float r2, f3 ray_dir, f3 a_normal, f2 tc, f3 a_tan, f3 a_bitan, const OrcScene* s, MatSample* out) {
  const f3 color = clamp3(mul3(sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s), matColor(m)), 0.0f, 1.0f);
  const f2 alpha = beckmannAlphaXY(m, tc, s);
  const float gloss = phongGlosiness(m, tc, s);
  f3 nx, ny;
  const f3 nz = a_normal;
  BeckmanTangentSpace(m, alpha, nz, a_tan, a_bitan, tc, s, &nx, &ny);
  const f3 wo = v3(-dot3(ray_dir, nx), -dot3(ray_dir, ny), -dot3(ray_dir, nz));
  const f3 wh = microfacetSampleWH(TR, wo, r1, r2, alpha.x, alpha.y);
  const f3 wi = sub3(scale3(wh, 2.0f * dot3(wo, wh)), wo);
  const f3 newDir = normalize3(add3(add3(scale3(nx, wi.x), scale3(ny, wi.y)), scale3(nz, wi.z)));
  const f3 v = scale3(ray_dir, -1.0f), l = newDir;
  if (dot3(a_normal, v) < 1e-6f || dot3(a_normal, l) < 1e-6f) { out->color = v3(0, 0, 0); out->pdf = 1.0f; }
  else { out->color = scale3(color, microfacetBRDF_PBRT(TR, wo, wi, alpha.x, alpha.y)); out->pdf = microfacetPdf(TR, wo, wh, alpha.x, alpha.y); }
  out->direction = newDir;
  out->flags = (gloss >= 0.99f) ? RAY_EVENT_S : RAY_EVENT_G;
}
/* ---- translucent (diffuse transmission), ref: cmaterial.h:1852-1909; colour and sampler at the lambert offsets */
static float translucentEvalPDF(f3 l, f3 v, f3 n) {
  const float sign1 = dot3(l, n) > 0 ? 1.0f : -1.0f, sign2 = dot3(v, n) > 0 ? 1.0f : -1.0f;
  const float coeff = (sign1 * sign2 < 0.0f) ? 1.0f : 0.0f;
  return fabsf(dot3(l, n)) * INV_PI * coeff;
}
static f3 translucentEvalBxDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const OrcScene* s) {
  const f3 tex = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  const float sign1 = dot3(l, n) > 0 ? 1.0f : -1.0f, sign2 = dot3(v, n) > 0 ? 1.0f : -1.0f;
  const float coeff = (sign1 * sign2 < 0.0f) ? 1.0f : 0.0f;
  return scale3(scale3(clamp3(mul3(tex, matColor(m)), 0.0f, 1.0f), coeff), INV_PI);
}
static void TranslucentSampleAndEvalBRDF(const float* m, float r1, float r2, f3 n, f2 tc, const OrcScene* s, MatSample* out) {
  const f3 tex = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  const f3 kd = clamp3(mul3(tex, matColor(m)), 0.0f, 1.0f);
  const f3 nn = scale3(n, -1.0f);
  const f3 newDir = MapSampleToCosineDistribution(r1, r2, nn, nn, 1.0f);
  const float cosTheta = dot3(newDir, nn);
  out->direction = newDir;
  out->pdf = cosTheta * INV_PI;
  out->color = scale3(kd, INV_PI);
  if (cosTheta <= 1e-6f) out->color = v3(0, 0, 0);
  out->flags = (RAY_EVENT_D | RAY_EVENT_T);
}
/* ---- oren-nayar, ref: cmaterial.h:288-371; CosPhiPBRT1 / SinPhiPBRT1 cmatpbrt.h:17-31 */
static float orennayarFunc(f3 l, f3 v, f3 n, float A, float B) {
  const float cosTheta_wi = dot3(l, n), cosTheta_wo = dot3(v, n);
  const float sinTheta_wi = sqrtf(fmaxf(0.0f, 1.0f - cosTheta_wi * cosTheta_wi));
  const float sinTheta_wo = sqrtf(fmaxf(0.0f, 1.0f - cosTheta_wo * cosTheta_wo));
  f3 nx, ny;
  CoordinateSystem(n, &nx, &ny);
  const f3 wo = v3(-dot3(v, nx), -dot3(v, ny), -dot3(v, n));
  const f3 wi = v3(-dot3(l, nx), -dot3(l, ny), -dot3(l, n));
  float maxcos = 0.f;
  if (sinTheta_wi > 1e-4f && sinTheta_wo > 1e-4f) {
    const float sinphii = clampf(wi.y / sinTheta_wi, -1.f, 1.f), cosphii = clampf(wi.x / sinTheta_wi, -1.f, 1.f);
    const float sinphio = clampf(wo.y / sinTheta_wo, -1.f, 1.f), cosphio = clampf(wo.x / sinTheta_wo, -1.f, 1.f);
    const float dcos = cosphii * cosphio + sinphii * sinphio;
    maxcos = fmaxf(0.f, dcos);
  }
  float sinalpha, tanbeta;
  if (fabsf(cosTheta_wi) > fabsf(cosTheta_wo)) { sinalpha = sinTheta_wo; tanbeta = sinTheta_wi / fmaxf(fabsf(cosTheta_wi), DEPSILON); }
  else { sinalpha = sinTheta_wi; tanbeta = sinTheta_wo / fmaxf(fabsf(cosTheta_wo), DEPSILON); }
  return (A + B * maxcos * sinalpha * tanbeta);
}
static f3 orennayarEvalBxDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const OrcScene* s) {
  return scale3(lambertEvalBxDF(m, tc, s), orennayarFunc(l, v, n, m[ORENNAYAR_A], m[ORENNAYAR_B]));
}
static void OrennayarSampleAndEvalBRDF(const float* m, float r1, float r2, f3 ray_dir, f3 n, f2 tc, const OrcScene* s, MatSample* out) {
  const f3 tex = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  const f3 color = clamp3(mul3(tex, matColor(m)), 0.0f, 1.0f);
  const f3 newDir = MapSampleToCosineDistribution(r1, r2, n, n, 1.0f);
  const float cosTheta = dot3(newDir, n);
  out->direction = newDir;
  out->pdf = cosTheta * INV_PI;
  out->color = scale3(scale3(color, INV_PI), orennayarFunc(newDir, scale3(ray_dir, -1.0f), n, m[ORENNAYAR_A], m[ORENNAYAR_B]));
  if (cosTheta <= DEPSILON) out->color = v3(0, 0, 0);
  out->flags = RAY_EVENT_D;
}
/* ---- phong, ref: cmaterial.h:915-1033 */
static float phongGlosiness(const float* m, f2 tc, const OrcScene* s) {
  if ((uint32_t)as_int(m[PHONG_GLOSS_TEXID]) != INVALID_TEXTURE) {
    const f3 g = sample2DExt(as_int(m[PHONG_GLOSS_TEXMATRIXID]), tc, m, s);
    return clampf(m[PHONG_GLOSINESS] * fmaxf(g.x, fmaxf(g.y, g.z)), 0.0f, 0.99f);
  }
  return m[PHONG_GLOSINESS];
}
static float phongEvalPDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const OrcScene* s) {
  const float dotNV = dot3(n, v), dotNL = dot3(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) return 1.0f;
  const float cosPower = cosPowerFromGlosiness(phongGlosiness(m, tc, s));
  const f3 r = reflect3(scale3(v, -1.0f), n);
  const float cosTheta = clampf(fabsf(dot3(l, r)), 0.0f, 1.0f);
  return powf(cosTheta, cosPower) * (cosPower + 1.0f) * INV_TWOPI;
}
static inline float PhongEnergyFix(float dotRL, f3 l, f3 n) { return dotRL / fmaxf(dot3(n, l), 1e-6f); }
static f3 phongEvalBxDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const OrcScene* s) {
  const float dotNV = dot3(n, v), dotNL = dot3(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) return v3(0, 0, 0);
  const f3 tex = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  const f3 color = clamp3(mul3(matColor(m), tex), 0.0f, 1.0f);
  const float cosPower = cosPowerFromGlosiness(phongGlosiness(m, tc, s));
  const f3 r = reflect3(scale3(v, -1.0f), n);
  const float cosAlpha = clampf(dot3(l, r), 0.0f, 1.0f);
  const float fix = (matFlags(m) & MF_ENERGY_FIX) ? PhongEnergyFix(cosAlpha, l, n) : 1.0f;
  return scale3(scale3(scale3(scale3(color, (cosPower + 2.0f)), INV_TWOPI), powf(cosAlpha, cosPower)), fix);
}
static void PhongSampleAndEvalBRDF(const float* m, float r1, float r2, f3 ray_dir, f3 n, f2 tc, const OrcScene* s, MatSample* out) {
  const f3 tex = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  const f3 color = clamp3(mul3(matColor(m), tex), 0.0f, 1.0f);
  const float gloss = phongGlosiness(m, tc, s);
  const float cosPower = cosPowerFromGlosiness(gloss);
  int under = 0;
  const f3 r = reflect3(ray_dir, n);
  const f3 newDir = MapSampleToModifiedCosineDistribution(r1, r2, r, n, cosPower, &under);
  const f3 v = scale3(ray_dir, -1.0f), l = newDir;
  const float dotNV = dot3(n, v), dotNL = dot3(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f || under) { out->color = v3(0, 0, 0); out->pdf = 1.0f; }
  else {
    const float cosAlpha = clampf(dot3(newDir, r), 0.0f, 1.0f);
    const float eqTemp = powf(cosAlpha, cosPower) * INV_TWOPI;
    const float fix = (matFlags(m) & MF_ENERGY_FIX) ? PhongEnergyFix(cosAlpha, newDir, n) : 1.0f;
    out->pdf = eqTemp * (cosPower + 1.0f);
    out->color = scale3(scale3(color, eqTemp * (cosPower + 2.0f)), fix);
  }
  out->direction = newDir;
  out->flags = (gloss >= 0.99f) ? RAY_EVENT_S : RAY_EVENT_G;
}
/* ---- mirror, ref: cmaterial.h:395-430 */
static void MirrorSampleAndEvalBRDF(const float* m, f3 ray_dir, f3 n, f2 tc, const OrcScene* s, MatSample* out) {
  const f3 tex = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  f3 newDir = reflect3(ray_dir, n);
  if (dot3(ray_dir, n) > 0.0f) newDir = ray_dir;
  const float cosOut = dot3(newDir, n);
  out->direction = newDir;
  out->pdf = 1.0f;
  out->color = scale3(mul3(matColor(m), tex), (1.0f / fmaxf(cosOut, 1e-6f)));
  if (cosOut <= 1e-6f) out->color = v3(0, 0, 0);
  out->flags = RAY_EVENT_S;
}

/* ---- multi-scattering tables, ref: cmaterial.h:61-196 (BilinearFrom2dTable, BilinearFrom3dTable, GetMultiscatteringFrom2dTable / 3dTable) */
static float BilinearFrom2dTable(const uint16_t* a_inData, float a_newPosX, float a_newPosY, int a_width, int a_height) {
  a_newPosX = clampf(a_newPosX, 0.0f, (float)a_width - 1.0001f);
  a_newPosY = clampf(a_newPosY, 0.0f, (float)a_height - 1.0001f);
  const int floorY = (int)floorf(a_newPosY), floorX = (int)floorf(a_newPosX);
  const int dxy1 = floorY * a_width + floorX, dxy2 = dxy1 + 1, dxy3 = (floorY + 1) * a_width + floorX, dxy4 = dxy3 + 1;
  const float dx = a_newPosX - (float)floorX, dy = a_newPosY - (float)floorY;
  const float mult1 = (1.0f - dx) * (1.0f - dy), mult2 = dx * (1.0f - dy), mult3 = dy * (1.0f - dx), mult4 = dx * dy;
  if (floorY >= 0 && floorX >= 0 && floorY <= a_height - 2 && floorX <= a_width - 2)
    return (float)a_inData[dxy1] * mult1 + (float)a_inData[dxy2] * mult2 + (float)a_inData[dxy3] * mult3 + (float)a_inData[dxy4] * mult4;
  return 1.0f;
}
static float BilinearFrom3dTable(const uint16_t* a_inData, float x, float y, float z, int a_width, int a_height, int a_depth, int a_size2dTable) {
  x = clampf(x, 0.0f, (float)a_width - 1.0001f);
  y = clampf(y, 0.0f, (float)a_height - 1.0001f);
  z = clampf(z, 0.0f, (float)a_depth - 1.0001f);
  const int floorX = (int)floorf(x), floorY = (int)floorf(y), floorZ = (int)floorf(z);
  const int zOffset = floorZ * a_size2dTable, zOffset2 = (floorZ + 1) * a_size2dTable;
  const int dxy1 = zOffset + floorY * a_width + floorX, dxy3 = zOffset + (floorY + 1) * a_width + floorX, dxy2 = dxy1 + 1, dxy4 = dxy3 + 1;
  const int dxy5 = zOffset2 + floorY * a_width + floorX, dxy7 = zOffset2 + (floorY + 1) * a_width + floorX, dxy6 = dxy5 + 1, dxy8 = dxy7 + 1;
  const float dx = x - (float)floorX, dy = y - (float)floorY, dz = z - (float)floorZ;
  const float mult1 = (1.0f - dx) * (1.0f - dy), mult2 = dx * (1.0f - dy), mult3 = dy * (1.0f - dx), mult4 = dx * dy;
  if (floorY >= 0 && floorX >= 0 && floorY <= a_height - 2 && floorX <= a_width - 2) {
    const float plane1 = (float)a_inData[dxy1] * mult1 + (float)a_inData[dxy2] * mult2 + (float)a_inData[dxy3] * mult3 + (float)a_inData[dxy4] * mult4;
    const float plane2 = (float)a_inData[dxy5] * mult1 + (float)a_inData[dxy6] * mult2 + (float)a_inData[dxy7] * mult3 + (float)a_inData[dxy8] * mult4;
    return plane1 + dz * (plane2 - plane1);
  }
  return 1.0f;
}
static f3 multiscatter(float Ess, f3 color) {   /* 1.0f + color * (1.0f - Ess) / fmax(Ess, 1e-6f), component-wise */
  const float k = 1.0f - Ess, d = fmaxf(Ess, 1e-6f);
  return v3(1.0f + (color.x * k) / d, 1.0f + (color.y * k) / d, 1.0f + (color.z * k) / d);
}
static f3 GetMultiscatteringFrom2dTable(const OrcScene* s, float roughness, float dotNV, f3 color) {
  const uint16_t* msTable = (const uint16_t*)(s->globals + G_ESS_GGX_TABLE);
  const float Ess = BilinearFrom2dTable(msTable, dotNV * 64.0f, roughness * 64.0f, 64, 64) * (1.0f / 65535.0f);
  return multiscatter(Ess, color);
}
static f3 GetMultiscatteringFrom3dTable(const OrcScene* s, float a_roughness, float a_dotNV, float a_ior, f3 a_color) {
  if (a_ior >= 0.4166f && a_ior <= 2.4f) {
    const uint16_t* msTable = (const uint16_t*)(s->globals + G_ESS_TRANSP_TABLE);
    const float iorNormal = (a_ior - 0.4166f) / (2.4f - 0.4166f);
    const float Ess = BilinearFrom3dTable(msTable, a_dotNV * 64.0f, a_roughness * 64.0f, iorNormal * 64.0f, 64, 64, 64, 64 * 64) * (1.0f / 65536.0f);
    return multiscatter(Ess, a_color);
  }
  return v3(1.0f, 1.0f, 1.0f);
}
/* ---- thin glass, ref: cmaterial.h:496-556.  thinglassEvalBxDF / EvalPDF return 0 (:510-520). */
static float transparencyGloss(const float* m, int multOffs, int texMatrixOffs, f2 tc, const OrcScene* s) {   /* :496-505, :610-618 */
  const f3 glossColor = sample2DExt(as_int(m[texMatrixOffs]), tc, m, s);
  return clampf(m[multOffs] * fmaxf(glossColor.x, fmaxf(glossColor.y, glossColor.z)), 0.0f, 1.0f);
}
static void ThinglassSampleAndEvalBRDF(const float* m, float r1, float r2, f3 ray_dir, f3 n, f2 tc, const OrcScene* s, MatSample* out) {
  const f3 texColor = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  const float cosPower = cosPowerFromGlosiness(transparencyGloss(m, THINGLASS_GLOSINESS, THINGLASS_GLOSINESS_TEXMATRIXID, tc, s));
  float pdf = 1.0f, fVal = 1.0f;
  if (cosPower < 1e6f) {
    int underSurface = 0;
    const f3 oldDir = ray_dir;
    ray_dir = MapSampleToModifiedCosineDistribution(r1, r2, ray_dir, scale3(n, -1.0f), cosPower, &underSurface);
    const float cosTheta = clampf(dot3(oldDir, ray_dir), 0.0f, M_PI_F * 0.499995f);
    fVal = (cosPower + 2.0f) * INV_TWOPI * powf(cosTheta, cosPower);
    if (underSurface) fVal = 0.0f;
    pdf = powf(cosTheta, cosPower) * (cosPower + 1.0f) * (0.5f * INV_PI);
  }
  const float cosThetaOut = dot3(ray_dir, n);
  const float cosMult = 1.0f / fmaxf(fabsf(cosThetaOut), 1e-6f);
  out->direction = ray_dir;
  out->pdf = pdf;
  out->color = scale3(mul3(scale3(matColor(m), fVal), texColor), cosMult);
  if (cosThetaOut >= -1e-6f) out->color = v3(0, 0, 0);
  out->flags = (RAY_EVENT_S | RAY_EVENT_T | RAY_EVENT_TNINGLASS);
}
/* ---- glass, ref: cmaterial.h:684-714 myRefractGgx, :775-882 GlassGGXSampleAndEvalBRDF (the one :2298-2301 dispatches to),
 *      :1214-1252 SmithGGXMasking / GgxVndf / SmithGGXMaskingShadowing.  glassEvalBxDF / EvalPDF return 0 (:622-630).
 */
typedef struct { f3 ray_dir; int success; float eta; } RefractResult;
static RefractResult myRefractGgx(f3 ray_dir, f3 a_normal, float a_matIOR, float a_outsideIOR) {
  RefractResult res;
  res.eta = a_outsideIOR / a_matIOR;
  float cosTheta = dot3(a_normal, ray_dir) * (-1.0f);
  if (cosTheta < 0.0f) { cosTheta = cosTheta * (-1.0f); a_normal = scale3(a_normal, -1.0f); res.eta = 1.0f / res.eta; }
  const float dotVN = cosTheta * (-1.0f);
  const float k = 1.0f - res.eta * res.eta * (1.0f - cosTheta * cosTheta);
  if (k > 0.0f) {
    res.ray_dir = normalize3(add3(scale3(ray_dir, res.eta), scale3(a_normal, res.eta * cosTheta - sqrtf(k))));
    res.success = 1;
  } else {
    res.ray_dir = normalize3(add3(scale3(a_normal, dotVN * (-2.0f)), ray_dir));
    res.success = 0;
    res.eta = 1.0f;
  }
  return res;
}
static float SmithGGXMasking(float dotNV, float roughSqr) {
  const float denomC = sqrtf(roughSqr + (1.0f - roughSqr) * dotNV * dotNV) + dotNV;
  return 2.0f * dotNV / fmaxf(denomC, 1e-6f);
}
static float SmithGGXMaskingShadowing(float dotNL, float dotNV, float roughSqr) {
  const float denomA = dotNV * sqrtf(roughSqr + (1.0f - roughSqr) * dotNL * dotNL);
  const float denomB = dotNL * sqrtf(roughSqr + (1.0f - roughSqr) * dotNV * dotNV);
  return 2.0f * dotNL * dotNV / fmaxf(denomA + denomB, 1e-6f);
}
static f3 GgxVndf(f3 wo, float roughness, float u1, float u2) {
  const f3 v = normalize3(v3(wo.x * roughness, wo.y * roughness, wo.z));
  const f3 XAxis = v3(1.0f, 0.0f, 0.0f), ZAxis = v3(0.0f, 0.0f, 1.0f);
  const f3 t1 = (v.z < 0.999f) ? normalize3(cross3(v, ZAxis)) : XAxis;
  const f3 t2 = cross3(t1, v);
  const float a = 1.0f / (1.0f + v.z);
  const float r = sqrtf(u1);
  const float phi = (u2 < a) ? (u2 / a) * M_PI_F : M_PI_F + (u2 - a) / (1.0f - a) * M_PI_F;
  const float p1 = r * cosf(phi);
  const float p2 = r * sinf(phi) * ((u2 < a) ? 1.0f : v.z);
  const f3 n = add3(add3(scale3(t1, p1), scale3(t2, p2)), scale3(v, sqrtf(fmaxf(0.0f, 1.0f - p1 * p1 - p2 * p2))));
  return normalize3(v3(roughness * n.x, roughness * n.y, fmaxf(0.0f, n.z)));
}
static void GlassGGXSampleAndEvalBRDF(const float* m, const float* rands, f3 ray_dir, f3 a_normal, f2 tc, int a_hitFromInside, int a_isFwdDir,
                                      const OrcScene* s, MatSample* out) {
  const f3 texColor = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  const f3 color = clamp3(mul3(matColor(m), texColor), 0.0f, 1.0f);
  const float gloss = transparencyGloss(m, GLASS_GLOSINESS, GLASS_GLOSINESS_TEXMATRIXID, tc, s);
  const float roughness = clampf(1.0f - gloss, 0.0f, 1.0f);
  const float roughSqr = roughness * roughness;
  const float IOR = m[GLASS_IOR];
  const f3 normal2 = a_hitFromInside ? scale3(a_normal, -1.0f) : a_normal;
  int spec = 1;
  float Pss = 1.0f;
  f3 Pms = v3(1.0f, 1.0f, 1.0f);
  out->pdf = 1.0f;
  RefractResult refrData = myRefractGgx(ray_dir, normal2, IOR, 1.0f);
  if (gloss < 0.999f) {
    spec = 0;
    float eta = 1.0f / IOR;
    const float cosTheta = dot3(normal2, ray_dir) * (-1.0f);
    if (cosTheta < 0.0f) eta = 1.0f / eta;
    f3 nx, ny;
    const f3 nz = a_normal;
    CoordinateSystem(nz, &nx, &ny);
    const f3 wo = v3(-dot3(ray_dir, nx), -dot3(ray_dir, ny), -dot3(ray_dir, nz));
    const f3 wh = GgxVndf(wo, roughSqr, rands[0], rands[1]);
    const float dotWoWh = dot3(wo, wh);
    f3 newDir;
    const float radicand = 1.0f + eta * eta * (dotWoWh * dotWoWh - 1.0f);
    if (radicand > 0.0f) {
      newDir = sub3(scale3(wh, eta * dotWoWh - sqrtf(radicand)), scale3(wo, eta));
      refrData.success = 1;
      refrData.eta = eta;
    } else {
      newDir = sub3(scale3(wh, 2.0f * dotWoWh), wo);
      refrData.success = 0;
      refrData.eta = 1.0f;
    }
    refrData.ray_dir = normalize3(add3(add3(scale3(nx, newDir.x), scale3(ny, newDir.y)), scale3(nz, newDir.z)));
    const f3 v = scale3(ray_dir, -1.0f), l = refrData.ray_dir;
    const float dotNV = fabsf(dot3(a_normal, v)), dotNL = fabsf(dot3(a_normal, l));
    const float G1 = SmithGGXMasking(dotNV, roughSqr);
    const float G2 = SmithGGXMaskingShadowing(dotNL, dotNV, roughSqr);
    Pss = G2 / fmaxf(G1, 1e-6f);
    if (matFlags(m) & MF_ENERGY_FIX) Pms = GetMultiscatteringFrom3dTable(s, roughness, dotNV, 1.0f / eta, color);
  }
  const float cosThetaOut = dot3(refrData.ray_dir, a_normal);
  const float cosMult = 1.0f / fmaxf(fabsf(cosThetaOut), 1e-6f);
  out->direction = refrData.ray_dir;
  const float adjointBtdfMult = a_isFwdDir ? 1.0f : (refrData.eta * refrData.eta);
  if (refrData.success) out->color = scale3(mul3(scale3(scale3(color, adjointBtdfMult), Pss), Pms), cosMult);
  else out->color = scale3(mul3(scale3(v3(1.0f, 1.0f, 1.0f), Pss), Pms), cosMult);
  out->flags = spec ? (RAY_EVENT_S | RAY_EVENT_T) : (RAY_EVENT_G | RAY_EVENT_T);
  if (refrData.success && cosThetaOut >= -1e-6f) out->color = v3(0, 0, 0);
  else if (!refrData.success && cosThetaOut < 1e-6f) out->color = v3(0, 0, 0);
}

/* ---- GGX reflection, ref: cmaterial.h:1197-1210 ggxGlosiness, :1285-1291 GGX_Distribution, :1317-1344 ggx2EvalPDF,
 *      :1346-1381 ggxEvalBxDF, :1454-1520 GGXSample2AndEvalBRDF (the forms :2288-2291 and :2491-2497 dispatch to) */
static float ggxGlosiness(const float* m, f2 tc, const OrcScene* s) {
  if ((uint32_t)as_int(m[GGX_GLOSINESS_TEXID]) != INVALID_TEXTURE) {
    const f3 glossColor = sample2DExt(as_int(m[GGX_GLOSINESS_TEXMATRIXID]), tc, m, s);
    return clampf(m[GGX_GLOSINESS] * fmaxf(glossColor.x, fmaxf(glossColor.y, glossColor.z)), 0.0f, 0.99f);
  }
  return m[GGX_GLOSINESS];
}
static float GGX_Distribution(float cosThetaNH, float alpha) {
  const float alpha2 = alpha * alpha;
  const float NH_sqr = clampf(cosThetaNH * cosThetaNH, 0.0f, 1.0f);
  const float den = NH_sqr * alpha2 + (1.0f - NH_sqr);
  return alpha2 / fmaxf(M_PI_F * den * den, 1e-6f);
}
static float ggx2EvalPDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const OrcScene* s) {
  const float dotNV = dot3(n, v), dotNL = dot3(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) return 1.0f;
  const float gloss = ggxGlosiness(m, tc, s);
  const float roughness = 1.0f - gloss;
  const float roughSqr = roughness * roughness;
  const f3 h = normalize3(add3(v, l));
  const float dotNH = dot3(n, h), dotHV = dot3(h, v);
  const float G1 = SmithGGXMasking(dotNV, roughSqr);
  const float D = GGX_Distribution(dotNH, roughSqr);
  const float Dv = D * G1 * dotHV / fmaxf(dotNV, 1e-6f);
  const float jacob = 1.0f / fmaxf(4.0f * dotHV, 1e-6f);
  return Dv * jacob;
}
static f3 ggxEvalBxDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const OrcScene* s) {
  const float dotNV = dot3(n, v), dotNL = dot3(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) return v3(0.0f, 0.0f, 0.0f);
  const f3 texColor = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  const f3 color = clamp3(mul3(matColor(m), texColor), 0.0f, 1.0f);
  const float gloss = ggxGlosiness(m, tc, s);
  const float roughness = 1.0f - gloss;
  const float roughSqr = roughness * roughness;
  const f3 h = normalize3(add3(v, l));
  const float dotNH = dot3(n, h);
  const float D = GGX_Distribution(dotNH, roughSqr);
  const float G = SmithGGXMaskingShadowing(dotNL, dotNV, roughSqr);
  const float Pss = D * G / fmaxf(4.0f * dotNV * dotNL, 1e-6f);
  f3 Pms = v3(1, 1, 1);
  if (matFlags(m) & MF_ENERGY_FIX) Pms = GetMultiscatteringFrom2dTable(s, roughness, dotNV, color);
  return mul3(scale3(color, Pss), Pms);
}
static void GGXSample2AndEvalBRDF(const float* m, float a_r1, float a_r2, f3 ray_dir, f3 a_normal, f2 tc, const OrcScene* s, MatSample* out) {
  const f3 texColor = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  const f3 color = clamp3(mul3(matColor(m), texColor), 0.0f, 1.0f);
  const float gloss = ggxGlosiness(m, tc, s);
  const float roughness = 1.0f - gloss;
  const float roughSqr = roughness * roughness;
  f3 nx, ny;
  const f3 nz = a_normal;
  CoordinateSystem(nz, &nx, &ny);
  float Pss = 1.0f;
  f3 Pms = v3(1.0f, 1.0f, 1.0f);
  const f3 wo = v3(-dot3(ray_dir, nx), -dot3(ray_dir, ny), -dot3(ray_dir, nz));
  const f3 wh = GgxVndf(wo, roughSqr, a_r1, a_r2);
  const f3 wi = sub3(scale3(wh, 2.0f * dot3(wo, wh)), wo);
  const f3 newDir = normalize3(add3(add3(scale3(nx, wi.x), scale3(ny, wi.y)), scale3(nz, wi.z)));
  const f3 v = scale3(ray_dir, -1.0f), l = newDir;
  const float dotNV = dot3(a_normal, v), dotNL = dot3(a_normal, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) {
    Pss = 0.0f;
    out->pdf = 1.0f;
  } else {
    const f3 h = normalize3(add3(v, l));
    const float dotNH = dot3(a_normal, h), dotHV = dot3(h, v);
    const float D = GGX_Distribution(dotNH, roughSqr);
    const float G1 = SmithGGXMasking(dotNV, roughSqr);
    const float G2 = SmithGGXMaskingShadowing(dotNL, dotNV, roughSqr);
    Pss = D * G2 / fmaxf(4.0f * dotNV, 1e-6f);
    const float Dv = D * G1 * dotHV / fmaxf(dotNV, 1e-6f);
    const float jacob = 1.0f / fmaxf(4.0f * dotHV, 1e-6f);
    out->pdf = Dv * jacob;
    if (matFlags(m) & MF_ENERGY_FIX) Pms = GetMultiscatteringFrom2dTable(s, roughness, dotNV, color);
  }
  out->direction = newDir;
  const f3 c = mul3(scale3(color, Pss), Pms);
  const float d = fmaxf(dotNL, 1e-6f);
  out->color = v3(c.x / d, c.y / d, c.z / d);
  out->flags = (gloss >= 0.99f) ? RAY_EVENT_S : RAY_EVENT_G;
}

/* ---- blend, ref: cglobals.h:1880-1926 fresnel helpers, cmaterial.h:2008-2137 */
static float fresnelDielectric(float cosTheta1, float cosTheta2, float etaExt, float etaInt) {   /* ref: cglobals.h:1868-1877 */
  const float Rs = (etaExt * cosTheta1 - etaInt * cosTheta2) / (etaExt * cosTheta1 + etaInt * cosTheta2);
  const float Rp = (etaInt * cosTheta1 - etaExt * cosTheta2) / (etaInt * cosTheta1 + etaExt * cosTheta2);
  return (Rs * Rs + Rp * Rp) / 2.0f;
}
static float fresnelReflectionCoeff(float cosTheta1, float etaExt, float etaInt) {               /* ref: cglobals.h:1879-1921 */
  if (cosTheta1 < 0.0f) { const float t = etaInt; etaInt = etaExt; etaExt = t; }
  const float sinTheta2 = etaExt / etaInt * sqrtf(fmaxf(0.0f, 1.0f - cosTheta1 * cosTheta1));
  if (sinTheta2 > 1.0f) return 1.0f;
  const float cosTheta2 = sqrtf(fmaxf(0.0f, 1.0f - sinTheta2 * sinTheta2));
  return fresnelDielectric(fabsf(cosTheta1), cosTheta2, etaInt, etaExt);
}
static float hermiteSplineEvalT(float t, const float* points, const float* tangents, int numPoints) {   /* cmaterial.h:2008-2040 */
  int ps = (int)(t * (float)(numPoints - 1));
  if (ps == numPoints - 1) ps--;
  const int pe = ps + 1;
  const float tStart = (float)(ps) / (float)(numPoints - 1), tEnd = (float)(pe) / (float)(numPoints - 1);
  const float sx = fabsf(t - tStart) / (tEnd - tStart);
  const float s2 = sx * sx, s3 = s2 * sx;
  const float h1 = 2.0f * s3 - 3.0f * s2 + 1.0f, h2 = -2.0f * s3 + 3.0f * s2, h3 = s3 - 2.0f * s2 + sx, h4 = s3 - s2;
  return 1.0f - clampf(h1 * points[2 * ps + 1] + h2 * points[2 * pe + 1] + h3 * tangents[2 * ps + 1] + h4 * tangents[2 * pe + 1], 0.0f, 1.0f);
}
static float blendMaskAlpha2(const float* m, f3 v, f3 n, f2 tc, const OrcScene* s) {
  const f3 tex = sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s);
  const f3 lum1 = clamp3(mul3(tex, matColor(m)), 0.0f, 1.0f);
  const int bflags = as_int(m[BLEND_FLAGS_OFFSET]);
  float lum;
  if (bflags & BMF_EXTRUSION_LUMINANCE) lum = dot3(v3(0.2126f, 0.7152f, 0.0722f), lum1);
  else lum = fmaxf(lum1.x, fmaxf(lum1.y, lum1.z));
  const float normAngle = fabsf(dot3(v, n));
  float faloff = 0.0f;
  if (bflags & BMF_FALOFF) {
    const int start = as_int(m[BLEND_FALOFF_OFFSET]), size = as_int(m[BLEND_FALOFF_SIZE]);
    const float* points = (const float*)(s->globals + s->globals[G_FLOAT_ARRAYS]) + start;
    const float* tangents = points + size / 2;
    const float param = (as_int(m[BLEND_FLAGS2]) & BLEND_INVERT_FALOFF) ? normAngle : 1.0f - normAngle;
    faloff = hermiteSplineEvalT(param, points, tangents, size / 4);
  }
  if (as_int(m[BLEND_TYPE]) == BLEND_SIGMOID) {   /* maxSigmoid, cmaterial.h:2042-2046 */
    const float x2 = -5.0f + 10.0f * lum;
    lum = 1.04f / (1.0f + expf(-m[BLEND_SIGMOID_EXP] * x2)) - 0.02f;
  }
  if (bflags & BMF_FALOFF) return clampf(faloff, 0.0f, 1.0f);
  if (bflags & BMF_FRESNEL) return clampf(lum * fresnelReflectionCoeff(fabsf(normAngle), 1.0f, m[BLEND_FRESNEL_IOR]), 0.0f, 1.0f);
  return clampf(lum, 0.0f, 1.0f);
}
typedef struct { float w; int localOffs; } BRDFSelector;
static BRDFSelector blendSelectBRDF(const float* m, float r3, f3 rayDir, f3 n, f2 tc, int reflOnly, const OrcScene* s) {   /* :2091-2137 */
  float alpha = blendMaskAlpha2(m, rayDir, n, tc, s);
  BRDFSelector m1, m2;
  m1.localOffs = as_int(m[BLEND_MAT1]); m2.localOffs = as_int(m[BLEND_MAT2]);
  const float* comp1 = m + (size_t)m1.localOffs * MAT_FLOATS;
  m1.w = 1.0f; m2.w = 1.0f;
  const int bflags = as_int(m[BLEND_FLAGS_OFFSET]);
  if ((bflags & BMF_REFL_WEIGHT_IS_ONE) && matType(comp1) != MT_BLEND_MASK) { m1.w = alpha; m2.w = 1.0f; }
  if ((bflags & BMF_FRESNEL) != 0 && reflOnly) { m1.w = alpha; alpha = 1.0f; }
  return (r3 <= alpha) ? m1 : m2;
}
/* ref: cmaterial.h:2180-2207 materialRandomWalkBRDF */
static BRDFSelector materialRandomWalkBRDF(const float* m, const float* rands, f3 rayDir, f3 n, f2 tc, const OrcScene* s, int reflOnly) {
  BRDFSelector res = {1.0f, 0}, sel = {1.0f, 0};
  const float* node = m;
  int i = 0;
  while (matType(node) == MT_BLEND_MASK && i < FLOATS_PER_MLAYER) {
    const float rnd = rands[FLOATS_PER_SAMPLE + i];
    sel = blendSelectBRDF(node, rnd, rayDir, n, tc, (reflOnly && (i == 0)), s);
    res.w = res.w * sel.w;
    res.localOffs = res.localOffs + sel.localOffs;
    node = node + (size_t)sel.localOffs * MAT_FLOATS;
    i++;
  }
  return res;
}
/* ref: cmaterial.h:2245-2335 MaterialLeafSampleAndEvalBRDF */
static void MaterialLeafSampleAndEvalBRDF(const float* m, const SurfaceHit* sh, f3 ray_dir, const float* rands, int a_isFwdDir, const OrcScene* s, MatSample* out) {
  f3 n = sh->normal;
  const int nmap = hasNormalMap(m);
  if (nmap) {
    const int isGlass = (matType(m) == MT_GLASS);
    const f3 flatNorm = (sh->hfi && !isGlass) ? scale3(sh->flatNormal, -1.0f) : sh->flatNormal;
    n = BumpMapping(sh->tangent, sh->biTangent, flatNorm, sh->texCoord, m, s);
  }
  out->color = v3(0, 0, 0); out->direction = v3(0, 1, 0); out->pdf = 1.0f; out->flags = 0;
  switch (matType(m)) {
    case MT_PHONG: PhongSampleAndEvalBRDF(m, rands[0], rands[1], ray_dir, n, sh->texCoord, s, out); break;
    case MT_MIRROR: MirrorSampleAndEvalBRDF(m, ray_dir, n, sh->texCoord, s, out); break;
    case MT_LAMBERT: LambertSampleAndEvalBRDF(m, rands[0], rands[1], n, sh->texCoord, s, out); break;
    case MT_OREN_NAYAR: OrennayarSampleAndEvalBRDF(m, rands[0], rands[1], ray_dir, n, sh->texCoord, s, out); break;
    case MT_GGX: GGXSample2AndEvalBRDF(m, rands[0], rands[1], ray_dir, n, sh->texCoord, s, out); break;
    case MT_THIN_GLASS: ThinglassSampleAndEvalBRDF(m, rands[0], rands[1], ray_dir, n, sh->texCoord, s, out); break;
    case MT_TRANSLUCENT: TranslucentSampleAndEvalBRDF(m, rands[0], rands[1], n, sh->texCoord, s, out); break;
    case MT_BLINN: BlinnSampleAndEvalBRDF(m, rands[0], rands[1], ray_dir, n, sh->texCoord, s, out); break;
    case MT_SHADOW_MATTE: {   /* ShadowmatteSampleAndEvalBRDF, cmaterial.h:1929-1942; a_shadow = (0,0,0) from kernel_NextBounce, PT_Loop.cpp:240 -- or, with a back-plate in the
      header, the traced shadow of this bounce as the OpenCL layer's NextBounce hands it in (material.cl:812, 897): g_matteShadow, set by stage_next */
      const float sv = orc_have_back_plate(s) ? g_matteShadow : 0.0f;
      out->direction = ray_dir; out->pdf = 1.0f; out->color = scale3(v3(sv, sv, sv), 1.0f / fmaxf(fabsf(dot3(ray_dir, n)), 1e-5f)); out->flags = RAY_EVENT_S | RAY_EVENT_T;
      break;
    }
    case MT_BECKMANN: AnisoSampleAndEvalBRDF(0, m, rands[0], rands[1], ray_dir, n, sh->texCoord, sh->tangent, sh->biTangent, s, out); break;
    case MT_TRGGX: AnisoSampleAndEvalBRDF(1, m, rands[0], rands[1], ray_dir, n, sh->texCoord, sh->tangent, sh->biTangent, s, out); break;
    case MT_GLASS: GlassGGXSampleAndEvalBRDF(m, rands, ray_dir, n, sh->texCoord, sh->hfi, a_isFwdDir, s, out); break;   /* CPUExp_Integrators_PT_Loop.cpp:240 passes false, the light paths of MMLT true */
    default: break;
  }
  if (nmap) {
    const float cosThetaOut1 = fabsf(dot3(out->direction, sh->normal)), cosThetaOut2 = fabsf(dot3(out->direction, n));
    out->color = scale3(out->color, (cosThetaOut2 / fmaxf(cosThetaOut1, DEPSILON2)));
  }
  if (out->pdf <= 0.0f) out->color = v3(0, 0, 0);
}
/* ref: cglobals.h:1366-1376 isEyeRay */
static inline int isEyeRay(uint32_t flags) {
  const uint32_t other = (flags & 0xFFFF0000u) >> 16;
  const int nonSpec = (other & RAY_EVENT_D) || (other & RAY_EVENT_G);
  return (((flags & 0x0000FF00u) >> 8) == 0) || !nonSpec;
}
/* ref: cmaterial.h:2345-2371 MaterialSampleAndEvalBxDF */
static void MaterialSampleAndEvalBxDFEx(const float* m, const float* rands, const SurfaceHit* sh, f3 rayDir, uint32_t rayFlags, int a_isFwdDir, const OrcScene* s, MatSample* out) {
  const uint32_t other = (rayFlags & 0xFFFF0000u) >> 16;
  const int canReflOnly = (matFlags(m) & MF_CAN_SAMPLE_REFL_ONLY) != 0;
  const int reflOnly = ((other & RAY_GRAMMAR_DIRECT_LIGHT) != 0) && canReflOnly;
  const BRDFSelector mix = materialRandomWalkBRDF(m, rands, rayDir, sh->normal, sh->texCoord, s, reflOnly);
  const float* leaf = m + (size_t)mix.localOffs * MAT_FLOATS;
  MaterialLeafSampleAndEvalBRDF(leaf, sh, rayDir, rands, a_isFwdDir, s, out);
  out->color = scale3(out->color, 1.0f / fmaxf(mix.w, 0.015625f));
  if ((matFlags(leaf) & MF_SKIP_SKY_PORTAL) && isEyeRay(rayFlags)) { out->color = v3(1, 1, 1); out->pdf = 1.0f; }
}
static void MaterialSampleAndEvalBxDF(const float* m, const float* rands, const SurfaceHit* sh, f3 rayDir, uint32_t rayFlags, const OrcScene* s, MatSample* out) {
  MaterialSampleAndEvalBxDFEx(m, rands, sh, rayDir, rayFlags, 0, s, out);
}
/* ref: cmaterial.h:2425-2551 materialLeafEval (EVAL_FLAG_DEFAULT: no forward-direction fix, no normal map) */
/* ref: cmaterial.h:2398-2417 adjointBsdfShadeNormalFix */
static float adjointBsdfShadeNormalFix(f3 toLightWo, f3 toCamWi, f3 shadeNorm, f3 geomNorm, float maxVal) {
  if (dot3(shadeNorm, geomNorm) < 0) geomNorm = scale3(geomNorm, -1.0f);
  if (1.0f - fabsf(dot3(shadeNorm, geomNorm)) <= 1e-6f) return 1.0f;
  else if (dot3(toCamWi, geomNorm) * dot3(toCamWi, shadeNorm) <= 0 || dot3(toLightWo, geomNorm) * dot3(toLightWo, shadeNorm) <= 0) return 1.0f;
  const float k1 = dot3(toLightWo, shadeNorm), k2 = dot3(toCamWi, geomNorm), k3 = dot3(toLightWo, geomNorm), k4 = dot3(toCamWi, shadeNorm);
  const float res = (k1 * k2) / fmaxf(k3 * k4, DEPSILON2);
  return fminf(fmaxf(res, 0.1f), maxVal);
}
static BxDFResult materialLeafEval(const float* m, const ShadeContext* sc0, int a_fwdDir, const OrcScene* s) {
  BxDFResult r;
  r.brdf = v3(0, 0, 0); r.btdf = v3(0, 0, 0); r.pdfFwd = 0.0f; r.pdfRev = 0.0f; r.diffuse = 0;
  float cosMult = 1.0f, cosMult2 = 1.0f;
  ShadeContext scn = *sc0;
  if (hasNormalMap(m)) {   /* :2431-2459 */
    const f3 nb = BumpMapping(sc0->tg, sc0->bn, sc0->fn, sc0->tc, m, s);
    const f3 lDir = a_fwdDir ? sc0->v : sc0->l;
    const float clampVal = a_fwdDir ? 0.15f : 1e-6f;
    const float cosThetaOut1 = fmaxf(dot3(lDir, sc0->n), 0.0f), cosThetaOut2 = fmaxf(dot3(lDir, nb), 0.0f);
    cosMult = (cosThetaOut2 / fmaxf(cosThetaOut1, clampVal));
    if (cosThetaOut1 <= 0.0f) cosMult = 0.0f;
    const float cosThetaOut3 = fmaxf(-dot3(lDir, sc0->n), 0.0f), cosThetaOut4 = fmaxf(-dot3(lDir, nb), 0.0f);
    cosMult2 = (cosThetaOut4 / fmaxf(cosThetaOut3, clampVal));
    if (cosThetaOut3 <= 0.0f) cosMult2 = 0.0f;
    if (a_fwdDir && dot3(sc0->l, sc0->fn) <= 0.0f) { cosMult = 0.0f; cosMult2 = 0.0f; }
    scn.n = nb;
  }
  const ShadeContext* sc = &scn;
  switch (matType(m)) {
    case MT_PHONG:
      r.brdf = scale3(phongEvalBxDF(m, sc->l, sc->v, sc->n, sc->tc, s), cosMult);
      r.pdfFwd = phongEvalPDF(m, sc->l, sc->v, sc->n, sc->tc, s);
      r.pdfRev = phongEvalPDF(m, sc->v, sc->l, sc->n, sc->tc, s);
      break;
    case MT_GGX:
      r.brdf = scale3(ggxEvalBxDF(m, sc->l, sc->v, sc->n, sc->tc, s), cosMult);
      r.pdfFwd = ggx2EvalPDF(m, sc->l, sc->v, sc->n, sc->tc, s);
      r.pdfRev = ggx2EvalPDF(m, sc->v, sc->l, sc->n, sc->tc, s);
      break;
    case MT_MIRROR: break;   /* mirrorEvalBxDF / PDF return 0, cmaterial.h:395-403 */
    case MT_THIN_GLASS: case MT_GLASS: break;   /* thinglass / glass EvalBxDF and EvalPDF return 0, cmaterial.h:510-520, 622-630 */
    case MT_LAMBERT:
      r.brdf = scale3(lambertEvalBxDF(m, sc->tc, s), cosMult);
      r.pdfFwd = lambertEvalPDF(sc->l, sc->n);
      r.pdfRev = lambertEvalPDF(sc->v, sc->n);
      r.diffuse = 1;
      break;
    case MT_BLINN:
      r.brdf = scale3(blinnEvalBxDF(m, sc->l, sc->v, sc->n, sc->tc, s), cosMult);
      r.pdfFwd = blinnEvalPDF(m, sc->l, sc->v, sc->n, sc->tc, s);
      r.pdfRev = blinnEvalPDF(m, sc->v, sc->l, sc->n, sc->tc, s);
      break;
    case MT_BECKMANN:
    case MT_TRGGX: {
      const int TR = (matType(m) == MT_TRGGX);
      r.brdf = scale3(anisoEvalBxDF(TR, m, sc->l, sc->v, sc->n, sc->tg, sc->bn, sc->tc, s), cosMult);
      r.pdfFwd = anisoEvalPDF(TR, m, sc->l, sc->v, sc->n, sc->tg, sc->bn, sc->tc, s);
      r.pdfRev = anisoEvalPDF(TR, m, sc->v, sc->l, sc->n, sc->tg, sc->bn, sc->tc, s);
      break;
    }
    case MT_TRANSLUCENT:
      r.btdf = scale3(translucentEvalBxDF(m, sc->l, sc->v, sc->n, sc->tc, s), cosMult2);
      r.pdfFwd = translucentEvalPDF(sc->l, sc->v, sc->n);
      r.pdfRev = translucentEvalPDF(sc->v, sc->l, sc->n);
      r.diffuse = 1;
      break;
    case MT_OREN_NAYAR:
      r.brdf = scale3(orennayarEvalBxDF(m, sc->l, sc->v, sc->n, sc->tc, s), cosMult);
      r.pdfFwd = lambertEvalPDF(sc->l, sc->n);
      r.pdfRev = lambertEvalPDF(sc->v, sc->n);
      r.diffuse = 1;
      break;
    default: break;
  }
  if (a_fwdDir) r.brdf = scale3(r.brdf, adjointBsdfShadeNormalFix(sc0->v, sc0->l, sc0->n, sc0->fn, r.diffuse ? 20.0f : 2.0f));   /* cmaterial.h:2540-2548: the shading normal, not the bumped one */
  return r;
}
/* ref: cmaterial.h:2554-2628 materialEval: explicit-stack walk over the blend tree */
static BxDFResult materialEvalEx(const float* a_m, const ShadeContext* sc, int a_fwdDir, const OrcScene* s) {
  BxDFResult val;
  val.brdf = v3(0, 0, 0); val.btdf = v3(0, 0, 0); val.pdfFwd = 0.0f; val.pdfRev = 0.0f; val.diffuse = 1;
  float stackW[MIX_TREE_MAX_DEEP]; int stackO[MIX_TREE_MAX_DEEP];
  int top = 0, currOffset = 0;
  float currW = 1.0f;
  do {
    if (top > 0) { top--; currOffset = stackO[top]; currW = stackW[top]; }
    const float* m = a_m + (size_t)currOffset * MAT_FLOATS;
    if (matType(m) == MT_BLEND_MASK) {
      const float alpha = blendMaskAlpha2(m, sc->v, sc->n, sc->tc, s);
      const int o1 = as_int(m[BLEND_MAT1]), o2 = as_int(m[BLEND_MAT2]);
      float w1 = alpha, w2 = 1.0f - alpha;
      const float* comp1 = m + (size_t)o1 * MAT_FLOATS;
      if ((as_int(m[BLEND_FLAGS_OFFSET]) & BMF_REFL_WEIGHT_IS_ONE) && matType(comp1) != MT_BLEND_MASK) w1 = 1.0f;
      if (top < MIX_TREE_MAX_DEEP) { stackW[top] = currW * w1; stackO[top] = currOffset + o1; top++; }
      if (top < MIX_TREE_MAX_DEEP) { stackW[top] = currW * w2; stackO[top] = currOffset + o2; top++; }
    } else {
      const BxDFResult b = materialLeafEval(m, sc, a_fwdDir, s);
      val.brdf = add3(val.brdf, scale3(b.brdf, currW));
      val.btdf = add3(val.btdf, scale3(b.btdf, currW));
      val.pdfFwd += currW * b.pdfFwd;
      val.pdfRev += currW * b.pdfRev;
      val.diffuse = val.diffuse && b.diffuse;
    }
  } while (top > 0);
  return val;
}
static BxDFResult materialEval(const float* a_m, const ShadeContext* sc, const OrcScene* s) { return materialEvalEx(a_m, sc, 0, s); }   /* EVAL_FLAG_DEFAULT */
/* ref: cmaterial.h:2918-2978 materialEvalEmission; leaf: :20-26 */
static f3 materialEvalEmission(const float* a_m, f3 v, f3 n, f2 tc, const OrcScene* s) {
  f3 val = v3(0, 0, 0);
  float stackW[MIX_TREE_MAX_DEEP]; int stackO[MIX_TREE_MAX_DEEP];
  int top = 0, currOffset = 0;
  float currW = 1.0f;
  do {
    if (top > 0) { top--; currOffset = stackO[top]; currW = stackW[top]; }
    const float* m = a_m + (size_t)currOffset * MAT_FLOATS;
    if (matType(m) == MT_BLEND_MASK) {
      const float alpha = blendMaskAlpha2(m, v, n, tc, s);
      const int o1 = as_int(m[BLEND_MAT1]), o2 = as_int(m[BLEND_MAT2]);
      if (top < MIX_TREE_MAX_DEEP) { stackW[top] = currW * alpha; stackO[top] = currOffset + o1; top++; }
      if (top < MIX_TREE_MAX_DEEP) { stackW[top] = currW * (1.0f - alpha); stackO[top] = currOffset + o2; top++; }
    }
    {
      const f3 tex = sample2DExt(as_int(m[EMISSIVE_TEXMATRIXID]), tc, m, s);
      const f3 e = mul3(v3(m[EMISSIVE_COLOR], m[EMISSIVE_COLOR + 1], m[EMISSIVE_COLOR + 2]), tex);
      val = add3(val, scale3(e, currW));
    }
  } while (top > 0);
  return val;
}
/* ref: cmaterial.h:3262-3293 flagsNextBounceLite */
static uint32_t flagsNextBounceLite(uint32_t flags, const MatSample* ms, const OrcScene* s) {
  const int thisDiffuse = (ms->flags & RAY_EVENT_D) != 0;
  const uint32_t bounce = (flags & 0x0000FF00u) >> 8, diff = (flags & 0x000000FFu);
  uint32_t other = (flags & 0xFFFF0000u) >> 16;
  flags = (flags & 0xFFFF00FFu) | ((bounce + 1) << 8);
  if (thisDiffuse) flags = (flags & 0xFFFFFF00u) | (diff + 1);
  const uint32_t bounce2 = bounce + 1, diff2 = flags & 0xFFu;
  if ((bounce2 >= (uint32_t)g_varsI(s)[HRT_TRACE_DEPTH]) || (diff2 >= (uint32_t)g_varsI(s)[HRT_DIFFUSE_TRACE_DEPTH] + 1)) other |= RAY_IS_DEAD;
  if (ms->flags & RAY_EVENT_G) other |= RAY_EVENT_G;
  if ((ms->flags & RAY_EVENT_S) || (ms->flags & RAY_EVENT_T)) other |= RAY_EVENT_S;
  if (ms->flags & RAY_EVENT_D) other |= RAY_EVENT_D;
  if (ms->flags & RAY_EVENT_T) other |= RAY_EVENT_T;
  return (flags & 0x0000FFFFu) | (other << 16);
}

/* ------------------------------------------------------------------------------------------------ lights */
static inline const float* lightAt(const OrcScene* s, int id) {   /* ref: clight.h:1739-1749 */
  if (id < 0) return NULL;
  return (const float*)(s->globals + s->globals[G_LIGHTS_OFFS]) + (size_t)id * LIGHT_FLOATS;
}
static inline f3 lightPos(const float* L) { return v3(L[PL_POS], L[PL_POS + 1], L[PL_POS + 2]); }
static inline f3 lightNorm(const float* L) { return v3(L[PL_NORM], L[PL_NORM + 1], L[PL_NORM + 2]); }
static inline f3 lightColor(const float* L) { return v3(L[PL_COLOR], L[PL_COLOR + 1], L[PL_COLOR + 2]); }

/* ref: clight.h:524-530 areaDiffuseLightEvalPDF (+ PdfAtoW cglobals.h:1754-1757) */
static float areaDiffuseLightEvalPDF(const float* L, f3 rayDir, float hitDist) {
  const f3 ln = lightNorm(L);
  const float pdfA = 1.0f / fmaxf(L[PL_SURFACE_AREA], DEPSILON);
  const float d = dot3(rayDir, scale3(ln, -1.0f));
  const float cosVal = (as_int(L[PL_FLAGS]) & LF_HAS_IES) ? fabsf(d) : fmaxf(d, 0.0f);
  return (pdfA * hitDist * hitDist) / fmaxf(cosVal, DEPSILON2);
}
/* ref: clight.h:542-611 areaDiffuseLightGetIntensity (no texture, no IES, no sky portal in the subset; spot kept) */
static f3 areaDiffuseLightGetIntensity(const float* L, f3 rayDir, int eyeRay) {
  f3 color = lightColor(L);
  if (as_int(L[AL_SPOT_DISTR]) != 0) {
    const float cos1 = L[AL_SPOT_COS1], cos2 = L[AL_SPOT_COS2];
    const float cos_theta = fmaxf(dot3(scale3(rayDir, -1.0f), lightNorm(L)), 0.0f);
    const float tVal = (cos_theta - cos2) / (cos1 - cos2);           /* mylocalsmoothstep, clight.h:7-12 */
    const float tt = fminf(fmaxf(tVal, 0.0f), 1.0f);
    const float atten = tt * tt * (3.0f - 2.0f * tt);
    if (!eyeRay) color = scale3(color, clampf(atten, 0.0f, 1.0f));
    else color = scale3(color, 1.0f / fmaxf(color.x, fmaxf(color.y, color.z)));
  }
  return color;
}
/* ref: clight.h:614-629 areaLightSkyPortalCustomColor and the tail of areaDiffuseLightGetIntensity :590-607 (defined after the sky) */
static f3 portalSkyColor(const OrcScene* s, const float* L, f3 rayDir);
static f3 areaLightIntensity(const OrcScene* s, const float* L, f3 rayDir, int eyeRay);
static float lightDistributionMask(const OrcScene* s, const float* L, f3 rayDir);
/* ref: clight.h:1180-1229 AreaLightSampleRev */
typedef struct { f3 pos, color; float pdf, maxDist, cosAtLight; int isPoint; } ShadowSample;
static void AreaLightSampleRev(const OrcScene* s, const float* L, f3 rands, f3 illum, ShadowSample* out) {
  const float offsetX = rands.x * 2.0f - 1.0f, offsetY = rands.y * 2.0f - 1.0f;
  f3 sp = v3(offsetX * L[AL_SIZE_X], 0.0f, offsetY * L[AL_SIZE_Y]);
  if (as_int(L[AL_IS_DISK]) != 0) {
    f2 in = {offsetX, offsetY};
    const f2 xz = MapSamplesToDisc(in);
    sp = v3(xz.x * L[AL_SIZE_X], 0, xz.y * L[AL_SIZE_X]);
  }
  const float* M = L + AL_MATRIX;   /* matrix3x3f_mult_float3, cglobals.h:1091-1098 */
  sp = v3(M[0] * sp.x + M[1] * sp.y + M[2] * sp.z, M[3] * sp.x + M[4] * sp.y + M[5] * sp.z, M[6] * sp.x + M[7] * sp.y + M[8] * sp.z);
  sp = add3(sp, lightPos(L));
  const f3 rayDir = normalize3(sub3(sp, illum));
  const float hitDist = length3(sub3(sp, illum));
  f3 customRayDir = rayDir;
  if (as_int(L[PL_FLAGS]) & LF_IES_POINT_AREA) customRayDir = normalize3(sub3(lightPos(L), illum));
  const f3 color = (as_int(L[PL_FLAGS]) & LF_SKY_PORTAL) ? mul3(lightColor(L), portalSkyColor(s, L, rayDir)) : areaLightIntensity(s, L, customRayDir, 0);
  const f3 ln = lightNorm(L);
  out->isPoint = 0;
  out->pos = add3(sp, scale3(ln, epsilonOfPos(sp)));
  out->color = color;
  out->pdf = areaDiffuseLightEvalPDF(L, rayDir, hitDist);
  out->maxDist = hitDist;
  out->cosAtLight = -dot3(rayDir, ln);
}
/* ref: cglobals.h:2808-2859 SelectIndexPropToOpt */
static int SelectIndexPropToOpt(float a_r, const float* a_accum, int N, float* pPDF) {
  int leftBound = 0, rightBound = N - 2, counter = 0, currPos = -1;
  const int maxStep = 50;
  const float x = a_r * a_accum[N - 1];
  while (rightBound - leftBound > 1 && counter < maxStep) {
    const int currSize = rightBound + leftBound;
    const int currPos1 = (currSize % 2 == 0) ? (currSize + 1) / 2 : (currSize + 0) / 2;
    const float a = a_accum[currPos1 + 0], b = a_accum[currPos1 + 1];
    if (a < x && x <= b) { currPos = currPos1; break; }
    else if (x <= a) rightBound = currPos1;
    else if (x > b) leftBound = currPos1;
    counter++;
  }
  if (currPos < 0) {
    const float a1 = a_accum[leftBound + 0], b1 = a_accum[leftBound + 1], a2 = a_accum[rightBound + 0], b2 = a_accum[rightBound + 1];
    if (a1 < x && x <= b1) currPos = leftBound;
    if (a2 < x && x <= b2) currPos = rightBound;
  }
  if (x == 0.0f) currPos = 0;
  else if (currPos < 0) currPos = (rightBound + leftBound + 1) / 2;
  *pPDF = (a_accum[currPos + 1] - a_accum[currPos]) / a_accum[N - 1];
  return currPos;
}
/* ref: clight.h:1774-1793 SelectRandomLightRev */
static int SelectRandomLightRev(float r, const OrcScene* s, float* pickProb) {
  const int tableSize = s->globals[G_LSEL_REV_SIZE];
  if (tableSize == 0) { *pickProb = 1.0f; return -1; }
  if (tableSize <= 2) { *pickProb = 1.0f; return 0; }
  return SelectIndexPropToOpt(r, (const float*)(s->globals + s->globals[G_LSEL_REV_OFFS]), tableSize, pickProb);
}

/* ---- sky dome light (constant colour or lat-long texture; the Perez model next to environmentColor) ---- */
#define ORC_PI 3.14159265358979323846f   /* float: the pinned build of the reference uses -cl-single-precision-constant */
/* ref: cfetch.h:259-281 sphereMapToPhiTheta + sphereMapTo2DTexCoord */
static f2 sphereMapTo2DTexCoord(f3 ray_dir, float* pSinTheta) {
  const float x = ray_dir.z, y = ray_dir.x, z = -ray_dir.y;
  const float theta = acosf(z);
  float phi = atan2f(y, x);
  if (phi < 0.0f) phi += 2.0f * ORC_PI;
  f2 r;
  r.x = clampf(phi * 0.5f * INV_PI, 0.0f, 1.0f);
  r.y = clampf(theta * INV_PI, 0.0f, 1.0f);
  *pSinTheta = sqrtf(1.0f - ray_dir.y * ray_dir.y);
  return r;
}
/* ref: cfetch.h:283-296 texCoord2DToSphereMap */
static f3 texCoord2DToSphereMap(f2 tc, float* pSinTheta) {
  const float phi = tc.x * 2.f * ORC_PI, theta = tc.y * ORC_PI;
  const float sinTheta = sinf(theta);
  const float x = sinTheta * cosf(phi), y = sinTheta * sinf(phi), z = cosf(theta);
  *pSinTheta = sinTheta;
  return v3(y, -z, x);
}
/* ref: cfetch.h:153-163 pdfTableHeader (table offsets are in float4 units) */
static const float* pdfTableHeader(const OrcScene* s, int tableId) {
  const int offset = s->globals[s->globals[G_PDF_TABLE] + tableId];
  return s->pdfStorage + (size_t)offset * 4;
}
/* ref: clight.h:308-337 evalMap2DPdf (including its second test of texCoordT.x) */
static float evalMap2DPdf(f2 tc, const float* intervals, const int sizeX, const int sizeY) {
  const float fw = (float)sizeX, fh = (float)sizeY;
  if (tc.x < 0.0f || tc.x > 1.0f) tc.x -= (float)((int)(tc.x));
  if (tc.y < 0.0f || tc.x > 1.0f) tc.y -= (float)((int)(tc.y));
  int pixelX = (int)(fw * tc.x - 0.5f), pixelY = (int)(fh * tc.y - 0.5f);
  if (pixelX >= sizeX) pixelX = sizeX - 1;
  if (pixelY >= sizeY) pixelY = sizeY - 1;
  if (pixelX < 0) pixelX += sizeX;
  if (pixelY < 0) pixelY += sizeY;
  const int pixelOffset = pixelY * sizeX + pixelX, maxSize = sizeX * sizeY;
  const int offset0 = (pixelOffset + 0 < maxSize + 0) ? pixelOffset + 0 : maxSize - 1;
  const int offset1 = (pixelOffset + 1 < maxSize + 1) ? pixelOffset + 1 : maxSize;
  return (intervals[offset1] - intervals[offset0]) * (fw * fh) / intervals[sizeX * sizeY];
}
/* ref: clight.h:339-364 skyLightEvalPDF */
static float skyLightEvalPDF(const OrcScene* s, const float* L, f3 rayDir) {
  const float* hdr = pdfTableHeader(s, as_int(L[SKY_DOME_PDF_TABLE0]));
  const int sizeX = as_int(hdr[0]), sizeY = as_int(hdr[1]);
  float sintheta = 0.0f;
  const f2 tc = sphereMapTo2DTexCoord(rayDir, &sintheta);
  if (sintheta == 0.f) return 0.f;
  const float* r0 = L + SKY_DOME_MATRIX0;   /* mul2x4, cfetch.h:642-648 */
  f2 tcT;
  tcT.x = r0[0] * tc.x + r0[1] * tc.y + r0[3];
  tcT.y = r0[4] * tc.x + r0[5] * tc.y + r0[7];
  const float mapPdf = evalMap2DPdf(tcT, hdr + 4, sizeX, sizeY);
  return (mapPdf * 1.0f) / (2.f * ORC_PI * ORC_PI * fmaxf(fabsf(sintheta), DEPSILON));
}
/* ---- IES distributions.  ref: cfetch.h:364-462 read_imagef_sw1 (the float branch: IES images are {w, h, 1, 4} + w * h floats, RenderDriverRTE_PdfTables.cpp:425-441) */
static float read_imagef_sw1(const int32_t* tex, f2 tc, int flags) {
  const int w = tex[0], h = tex[1];
  float ffx = tc.x * (float)w - 0.5f, ffy = tc.y * (float)h - 0.5f;
  if ((flags & TEX_CLAMP_U) != 0 && ffx < 0) ffx = 0.0f;
  if ((flags & TEX_CLAMP_V) != 0 && ffy < 0) ffy = 0.0f;
  const float* fdata = (const float*)(tex + 4);
  const int px = (int)(ffx), py = (int)(ffy);
  const float fx = fabsf(ffx - (float)px), fy = fabsf(ffy - (float)py);
  const float fx1 = 1.0f - fx, fy1 = 1.0f - fy;
  const float w1 = fx1 * fy1, w2 = fx * fy1, w3 = fx1 * fy, w4 = fx * fy;
  int offs[4];
  bilinearOffsets(ffx, ffy, flags, w, h, offs);
  return ((fdata[offs[0]] * w1 + fdata[offs[1]] * w2) + fdata[offs[2]] * w3) + fdata[offs[3]] * w4;
}
static f3 lightMatrixMul3(const float* M, f3 v) { return v3(M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z); }
/* ref: clight.h:465-484 lightDistributionMask for a light with LIGHT_HAS_IES (1 otherwise) */
static float lightDistributionMask(const OrcScene* s, const float* L, f3 rayDir) {
  rayDir = normalize3(lightMatrixMul3(L + IES_LIGHT_MATRIX_E00, rayDir));
  float sintheta = 0.0f;
  const f2 tc = sphereMapTo2DTexCoord(scale3(rayDir, -1.0f), &sintheta);
  return read_imagef_sw1((const int32_t*)pdfTableHeader(s, as_int(L[IES_SPHERE_TEX_ID])), tc, TEX_CLAMP_U | TEX_CLAMP_V);
}
/* areaDiffuseLightGetIntensity with its IES branch, ref: clight.h:563-577 */
static f3 areaLightIntensity(const OrcScene* s, const float* L, f3 rayDir, int eyeRay) {
  if (as_int(L[PL_FLAGS]) & LF_HAS_IES) {
    f3 color = lightColor(L);
    const float atten = lightDistributionMask(s, L, rayDir);
    if (!eyeRay) color = scale3(color, atten);
    else color = scale3(color, 1.0f / fmaxf(color.x, fmaxf(color.y, color.z)));
    return color;
  }
  return areaDiffuseLightGetIntensity(L, rayDir, eyeRay);
}
/* ref: clight.h:411-426 LightSampleIESSphere */
static f3 texCoord2DToSphereMap(f2 tc, float* pSinTheta);
static void LightSampleIESSphere(const OrcScene* s, const float* L, f3 rands, f3* outDir, float* outPdfW) {
  const float* hdr = pdfTableHeader(s, as_int(L[IES_SPHERE_PDF_ID]));
  const int sizeX = as_int(hdr[0]), sizeY = as_int(hdr[1]);
  const float fw = (float)sizeX, fh = (float)sizeY;
  float pdf = 1.0f;
  int pixelOffset = SelectIndexPropToOpt(rands.z, hdr + 4, sizeX * sizeY + 1, &pdf);
  if (pixelOffset >= sizeX * sizeY) pixelOffset = sizeX * sizeY - 1;
  const int yPos = pixelOffset / sizeX, xPos = pixelOffset - yPos * sizeX;
  f2 tc;
  tc.x = (1.0f / fw) * (((float)(xPos) + 0.5f) + (rands.x * 2.0f - 1.0f) * 0.5f);
  tc.y = (1.0f / fh) * (((float)(yPos) + 0.5f) + (rands.y * 2.0f - 1.0f) * 0.5f);
  const float mapPdf = pdf * (fw * fh);
  float sinTheta = 0.0f;
  const f3 lsDir = texCoord2DToSphereMap(tc, &sinTheta);
  *outDir = normalize3(lightMatrixMul3(L + IES_INV_MATRIX_E00, lsDir));
  *outPdfW = INV_PI * INV_PI * 0.5f * (mapPdf / fmaxf(fabsf(sinTheta), DEPSILON2));
}
/* the IES branch of lightPdfFwd, ref: clight.h:1152-1164 */
static float lightPdfFwdIES(const OrcScene* s, const float* L, f3 ray_dir) {
  const f3 rayDir = lightMatrixMul3(L + IES_LIGHT_MATRIX_E00, ray_dir);
  const float* hdr = pdfTableHeader(s, as_int(L[IES_SPHERE_PDF_ID]));
  float sintheta = 0.0f;
  const f2 tc = sphereMapTo2DTexCoord(scale3(rayDir, -1.0f), &sintheta);
  const float mapPdf = evalMap2DPdf(tc, hdr + 4, as_int(hdr[0]), as_int(hdr[1]));
  return mapPdf / (2.f * ORC_PI * ORC_PI * fmaxf(sintheta, DEPSILON2));
}
/* ref: clight.h:427-462 SkyLightSampleRev, :378-403 sampleMap2D */
static void SkyLightSampleRev(const OrcScene* s, const float* L, f3 rands, f3 illum, ShadowSample* out) {
  const float* hdr = pdfTableHeader(s, as_int(L[SKY_DOME_PDF_TABLE0]));
  const int sizeX = as_int(hdr[0]), sizeY = as_int(hdr[1]);
  const float fw = (float)sizeX, fh = (float)sizeY;
  float pdf = 1.0f;
  int pixelOffset = SelectIndexPropToOpt(rands.z, hdr + 4, sizeX * sizeY + 1, &pdf);
  if (pixelOffset >= sizeX * sizeY) pixelOffset = sizeX * sizeY - 1;
  const int yPos = pixelOffset / sizeX, xPos = pixelOffset - yPos * sizeX;
  const float texX = (1.0f / fw) * (((float)(xPos) + 0.5f) + (rands.x * 2.0f - 1.0f) * 0.5f);
  const float texY = (1.0f / fh) * (((float)(yPos) + 0.5f) + (rands.y * 2.0f - 1.0f) * 0.5f);
  const float mapPdf = pdf * (fw * fh);
  const float* m = L + SKY_DOME_INV_MATRIX0;   /* mul(float4x4, float3) with four float4 columns, cglobals.h:839-846; z = 0 */
  f2 tcT;
  tcT.x = texX * m[0] + texY * m[4] + 0.0f * m[8] + m[12];
  tcT.y = texX * m[1] + texY * m[5] + 0.0f * m[9] + m[13];
  float sintheta = 0.0f;
  const f3 sampleDir = texCoord2DToSphereMap(tcT, &sintheta);
  const f3 samplePos = add3(illum, scale3(sampleDir, g_varsF(s)[HRT_BSPHERE_RADIUS]));
  const f3 txClr = sample2DExt(as_int(L[PL_COLOR_TEX_MATRIX]), tcT, L + SKY_DOME_SAMPLER0, s);   /* sample2D, cfetch.h:650-675 */
  out->isPoint = 0;
  out->pos = samplePos;
  out->color = mul3(v3(L[PL_COLOR], L[PL_COLOR + 1], L[PL_COLOR + 2]), txClr);
  out->pdf = (mapPdf * 1.0f) / (2.f * ORC_PI * ORC_PI * fmaxf(fabsf(sintheta), DEPSILON));
  out->maxDist = length3(sub3(illum, samplePos));
  out->cosAtLight = 1.0f;
}
/* ---- delta lights ---- */
/* ref: clight.h:7-12 mylocalsmoothstep */
static float mylocalsmoothstep(float edge0, float edge1, float x) {
  const float tVal = (x - edge0) / (edge1 - edge0);
  const float t = fminf(fmaxf(tVal, 0.0f), 1.0f);
  return t * t * (3.0f - 2.0f * t);
}
/* ref: cglobals.h:1754-1757 PdfAtoW */
static float PdfAtoW_full(float aPdfA, float aDist, float aCosThere) { return (aPdfA * aDist * aDist) / fmaxf(aCosThere, DEPSILON2); }
/* ref: clight.h:1394-1407 PointLightSampleRev; lightDistributionMask (:465-484) is (1,1,1) without IES */
static void PointLightSampleRev(const OrcScene* s, const float* L, f3 illum, ShadowSample* out) {
  const f3 samplePos = v3(L[PL_POS], L[PL_POS + 1], L[PL_POS + 2]);
  const float hitDist = length3(sub3(samplePos, illum));
  out->isPoint = 1;
  out->pos = samplePos;
  const float mask = (as_int(L[PL_FLAGS]) & LF_HAS_IES) ? lightDistributionMask(s, L, normalize3(sub3(samplePos, illum))) : 1.0f;
  out->color = mul3(v3(mask, mask, mask), v3(L[PL_COLOR], L[PL_COLOR + 1], L[PL_COLOR + 2]));
  out->pdf = PdfAtoW_full(1.0f, hitDist, 1.0f);
  out->maxDist = hitDist;
  out->cosAtLight = 1.0f;
}
/* ref: clight.h:1416-1450 pointSpotLightAttenuation + SpotLightSampleRev */
static void SpotLightSampleRev(const float* L, f3 illum, ShadowSample* out) {
  const f3 samplePos = v3(L[PL_POS], L[PL_POS + 1], L[PL_POS + 2]), norm = v3(L[PL_NORM], L[PL_NORM + 1], L[PL_NORM + 2]);
  const float hitDist = length3(sub3(samplePos, illum));
  const f3 rayDir = normalize3(sub3(samplePos, illum));
  const float cos_theta = fmaxf(dot3(scale3(rayDir, -1.0f), norm), 0.0f);
  const float atten = mylocalsmoothstep(L[POINT_LIGHT_SPOT_COS2], L[POINT_LIGHT_SPOT_COS1], cos_theta);
  out->isPoint = 1;
  out->pos = samplePos;
  out->color = scale3(v3(L[PL_COLOR], L[PL_COLOR + 1], L[PL_COLOR + 2]), atten);
  out->pdf = PdfAtoW_full(1.0f, hitDist, 1.0f);
  out->maxDist = hitDist;
  out->cosAtLight = fmaxf(-dot3(rayDir, norm), 0.0f);
}
/* ref: cglobals.h:1655-1681 MapSamplesToCone */
static f3 MapSamplesToCone(float cosCutoff, f2 sample, f3 direction) {
  const float cosTheta = (1.0f - sample.x) + sample.x * cosCutoff;
  const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
  const float sinPhi = sinf(2.0f * ORC_PI * sample.y), cosPhi = cosf(2.0f * ORC_PI * sample.y);
  const f3 deviation = v3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
  f3 nx, nz;
  CoordinateSystem(direction, &nx, &nz);
  const f3 ny = nz, nz2 = direction;   /* the reference swaps ny and nz */
  return add3(add3(scale3(nx, deviation.x), scale3(ny, deviation.y)), scale3(nz2, deviation.z));
}
/* ref: clight.h:1478-1506 DirectLightSampleRev, :892-912 directLightAttenuation */
static void DirectLightSampleRev(const float* L, f3 rands, f3 illum, ShadowSample* out) {
  const f3 lpos = v3(L[PL_POS], L[PL_POS + 1], L[PL_POS + 2]);
  const f3 n0 = v3(L[PL_NORM], L[PL_NORM + 1], L[PL_NORM + 2]);
  f3 norm = n0;
  const float pdfW = 1.0f;
  if (L[DIRECT_LIGHT_SSOFTNESS] > 1e-5f) { f2 sm; sm.x = rands.x; sm.y = rands.y; norm = MapSamplesToCone(L[DIRECT_LIGHT_ALPHA_COS], sm, norm); }
  const f3 AC = sub3(illum, lpos);
  const float CBLen = dot3(normalize3(AC), norm) * length3(AC);
  float atten = 0.0f;
  const float cos_alpha = dot3(normalize3(sub3(illum, lpos)), n0);
  if (cos_alpha > 0.0f) {
    const float sinAlpha = sqrtf(1.0f - cos_alpha * cos_alpha);
    const float d = length3(sub3(illum, lpos)) * sinAlpha;
    const float r1 = L[DIRECT_LIGHT_RADIUS1], r2 = L[DIRECT_LIGHT_RADIUS2];
    atten = mylocalsmoothstep(fmaxf(r2, r1), fminf(r2, r1), d);
  }
  out->isPoint = 1;
  out->pos = sub3(illum, scale3(norm, CBLen));
  out->color = scale3(scale3(v3(L[PL_COLOR], L[PL_COLOR + 1], L[PL_COLOR + 2]), atten), pdfW);
  out->pdf = pdfW;
  out->maxDist = CBLen;
  out->cosAtLight = 1.0f;
}
/* ref: clight.h:1561-1610 LightSampleRev, the types the layer accepts */
/* ref: clight.h:1287-1332 sphere lights */
static float sphereLightEvalPDF(const float* L, f3 illum, f3 lpos, f3 lnorm) {
  const float lradius = L[SPHERE_LIGHT_RADIUS];
  const f3 lcenter = lightPos(L);
  const f3 dc = sub3(lcenter, illum);
  if (dot3(dc, dc) - lradius * lradius <= 0.0f) return 1.0f;
  const float pdfA = 1.0f / L[PL_SURFACE_AREA];
  const float dist = length3(sub3(lpos, illum));
  const f3 dirToV = normalize3(sub3(lpos, illum));
  return PdfAtoW_full(pdfA, dist, fabsf(dot3(dirToV, lnorm)));
}
static f3 sphereLightUnitSample(float r1, float r2) {
  const float theta = 2.0f * M_PI_F * r1;
  const float phi = acosf(1.0f - 2.0f * r2);
  return v3(sinf(phi) * cosf(theta), sinf(phi) * sinf(theta), cosf(phi));
}
static void SphereLightSampleRev(const float* L, f3 rands, f3 illum, ShadowSample* out) {
  const f3 lcenter = lightPos(L);
  const f3 samplePos = add3(lcenter, scale3(sphereLightUnitSample(rands.x, rands.y), L[SPHERE_LIGHT_RADIUS]));
  const f3 lnorm = normalize3(sub3(samplePos, lcenter));
  const f3 dirToV = normalize3(sub3(samplePos, illum));
  out->isPoint = 0;
  out->pos = samplePos;
  out->color = lightColor(L);
  out->pdf = sphereLightEvalPDF(L, illum, samplePos, lnorm);
  out->maxDist = length3(sub3(samplePos, illum));
  out->cosAtLight = fabsf(dot3(lnorm, dirToV));
}
/* ref: clight.h:957-1062, 1513-1546 mesh lights */
static f3 meshLightGetIntensity(const OrcScene* s, const float* L, f2 tc) { return mul3(sample2DExt(as_int(L[MESH_LIGHT_TEXMATRIX_ID]), tc, L, s), lightColor(L)); }
static void MeshLightSamplePos(const OrcScene* s, const float* L, f3 rands, f3* pPos, f3* pNorm, f2* pTexCoord, float* pdfA) {
  const int meshId = as_int(L[MESH_LIGHT_MESH_OFFSET_ID]), pdftId = as_int(L[MESH_LIGHT_TABLE_OFFSET_ID]), triNum = as_int(L[MESH_LIGHT_TRI_NUM]);
  const float* mesh = pdfTableHeader(s, meshId);
  const float* table = pdfTableHeader(s, pdftId);
  const int32_t* hdr = (const int32_t*)mesh;
  const float* vpos = mesh + (size_t)hdr[0] * 4;
  const float* vnorm = mesh + (size_t)hdr[1] * 4;
  const int32_t* indices = (const int32_t*)(mesh + (size_t)hdr[3] * 4);
  float pickProb = 1.0f;
  const int triangleId = SelectIndexPropToOpt(rands.z, table, triNum + 1, &pickProb);
  const int iA = indices[triangleId * 3 + 0], iB = indices[triangleId * 3 + 1], iC = indices[triangleId * 3 + 2];
  const f3 A = v3(vpos[iA * 4], vpos[iA * 4 + 1], vpos[iA * 4 + 2]), B = v3(vpos[iB * 4], vpos[iB * 4 + 1], vpos[iB * 4 + 2]), C = v3(vpos[iC * 4], vpos[iC * 4 + 1], vpos[iC * 4 + 2]);
  const f3 nA = v3(vnorm[iA * 4], vnorm[iA * 4 + 1], vnorm[iA * 4 + 2]), nB = v3(vnorm[iB * 4], vnorm[iB * 4 + 1], vnorm[iB * 4 + 2]), nC = v3(vnorm[iC * 4], vnorm[iC * 4 + 1], vnorm[iC * 4 + 2]);
  float u = rands.x, v = rands.y;
  if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
  const float w = 1.0f - u - v;
  *pPos = add3(add3(scale3(A, u), scale3(B, v)), scale3(C, w));
  *pNorm = add3(add3(scale3(nA, u), scale3(nB, v)), scale3(nC, w));
  pTexCoord->x = (vpos[iA * 4 + 3] * u + vpos[iB * 4 + 3] * v) + vpos[iC * 4 + 3] * w;   /* tA = (pos.w, norm.w), :1003-1005 */
  pTexCoord->y = (vnorm[iA * 4 + 3] * u + vnorm[iB * 4 + 3] * v) + vnorm[iC * 4 + 3] * w;
  *pdfA = 1.0f / L[PL_SURFACE_AREA];
}
static f3 meshLightMatrixMul(const float* M, f3 v) { return v3(M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z); }
static void MeshLightSampleRev(const OrcScene* s, const float* L, f3 rands, f3 illum, ShadowSample* out) {
  f3 samplePos, sampleNorm; f2 tc; float pdfA;
  MeshLightSamplePos(s, L, rands, &samplePos, &sampleNorm, &tc, &pdfA);
  samplePos = meshLightMatrixMul(L + MESH_LIGHT_MATRIX_E00, samplePos);
  sampleNorm = normalize3(meshLightMatrixMul(L + MESH_LIGHT_MATRIX_E00, sampleNorm));
  samplePos = add3(samplePos, lightPos(L));
  const f3 rayDir = normalize3(sub3(samplePos, illum));
  const float hitDist = length3(sub3(samplePos, illum));
  const float cosVal = fmaxf(-dot3(rayDir, sampleNorm), 0.0f);
  out->isPoint = 0;
  out->pos = add3(samplePos, scale3(sampleNorm, epsilonOfPos(samplePos)));
  out->color = meshLightGetIntensity(s, L, tc);
  out->pdf = PdfAtoW_full(pdfA, hitDist, cosVal);
  out->maxDist = hitDist;
  out->cosAtLight = cosVal;
}
static float meshLightEvalPDF(const float* L, f3 rayDir, f3 lnorm, float hitDist) {
  const float pdfA = 1.0f / fmaxf(L[PL_SURFACE_AREA], DEPSILON);
  return PdfAtoW_full(pdfA, hitDist, fmaxf(dot3(rayDir, scale3(lnorm, -1.0f)), 0.0f));
}
/* ref: clight.h:753-830, 1338-1385 cylinder lights; sampleMap2D :378-403 */
static f3 cylinderLightGetIntensity(const OrcScene* s, const float* L, f2 tc) { return mul3(sample2DExt(as_int(L[CYLINDER_TEXMATRIX_ID]), tc, L, s), lightColor(L)); }
static void CylinderLightSamplePos(const OrcScene* s, const float* L, f3 rands, f3* pPos, f3* pNormal, f2* pTexCoord, float* pPdfA) {
  f2 tc = {rands.x, rands.y};
  float mapPdf = 1.0f;
  const int texId = as_int(L[CYLINDER_PDF_TABLE_ID]);
  if (texId > 0) {
    const float* hdr = pdfTableHeader(s, texId);
    const int sizeX = as_int(hdr[0]), sizeY = as_int(hdr[1]);
    const float fw = (float)sizeX, fh = (float)sizeY;
    float pdf = 1.0f;
    int pixelOffset = SelectIndexPropToOpt(rands.z, hdr + 4, sizeX * sizeY + 1, &pdf);
    if (pixelOffset >= sizeX * sizeY) pixelOffset = sizeX * sizeY - 1;
    const int yPos = pixelOffset / sizeX, xPos = pixelOffset - yPos * sizeX;
    tc.x = (1.0f / fw) * (((float)(xPos) + 0.5f) + (rands.x * 2.0f - 1.0f) * 0.5f);
    tc.y = (1.0f / fh) * (((float)(yPos) + 0.5f) + (rands.y * 2.0f - 1.0f) * 0.5f);
    mapPdf = pdf * (fw * fh);
  }
  *pPdfA = mapPdf / L[PL_SURFACE_AREA];
  const float zMin = L[CYLINDER_LIGHT_ZMIN], zMax = L[CYLINDER_LIGHT_ZMAX], radius = L[CYLINDER_LIGHT_RADIUS], phiMax = L[CYLINDER_LIGHT_PHIMAX];
  const float z = zMin + tc.x * (zMax - zMin);
  const float phi = tc.y * phiMax;
  const float sinPhi = sinf(phi), cosPhi = cosf(phi);   /* sincos2f, cglobals.h:338-341 */
  f3 pObj = v3(radius * cosPhi, radius * sinPhi, z);
  f3 n = normalize3(v3(pObj.x, pObj.y, 0.0f));
  const float hitRad = sqrtf(pObj.x * pObj.x + pObj.y * pObj.y);
  pObj.x *= radius / hitRad;
  pObj.y *= radius / hitRad;
  n = normalize3(meshLightMatrixMul(L + CYLINDER_LIGHT_MATRIX_E00, n));
  const f3 center = lightPos(L);
  *pPos = add3(add3(center, meshLightMatrixMul(L + CYLINDER_LIGHT_MATRIX_E00, pObj)), scale3(n, epsilonOfPos(center)));
  *pNormal = n;
  *pTexCoord = tc;
}
static float cylinderLightEvalPDF(const OrcScene* s, const float* L, f3 illum, f3 lpos, f3 lnorm, f2 texCoord) {
  float mapPdf = 1.0f;
  const int texId = as_int(L[CYLINDER_PDF_TABLE_ID]);
  if (texId) {
    const float* hdr = pdfTableHeader(s, texId);
    mapPdf = evalMap2DPdf(texCoord, hdr + 4, as_int(hdr[0]), as_int(hdr[1]));
  }
  const float hitDist = length3(sub3(lpos, illum));
  const f3 rayDir = normalize3(sub3(lpos, illum));
  const float pdfA = mapPdf / fmaxf(L[PL_SURFACE_AREA], DEPSILON);
  return PdfAtoW_full(pdfA, hitDist, fmaxf(dot3(rayDir, scale3(lnorm, -1.0f)), 0.0f));
}
static void CylinderLightSampleRev(const OrcScene* s, const float* L, f3 rands, f3 illum, ShadowSample* out) {
  f3 samplePos, n; f2 tc; float pdfA;
  CylinderLightSamplePos(s, L, rands, &samplePos, &n, &tc, &pdfA);
  const float hitDist = length3(sub3(samplePos, illum));
  const f3 rayDir = normalize3(sub3(samplePos, illum));
  const float cosVal = fmaxf(dot3(rayDir, scale3(n, -1.0f)), 0.0f);
  out->isPoint = 0;
  out->pos = samplePos;
  out->color = cylinderLightGetIntensity(s, L, tc);
  out->pdf = PdfAtoW_full(pdfA, hitDist, cosVal);
  out->maxDist = hitDist;
  out->cosAtLight = cosVal;
}
/* ref: clight.h:1613-1633 lightEvalPDF for the lights that have a surface in this subset */
static float lightEvalPDF(const OrcScene* s, const float* L, f3 illum, f3 rayDir, f3 lpos, f3 lnorm, f2 texCoord) {
  if (as_int(L[PL_TYPE]) == LT_SPHERE) return sphereLightEvalPDF(L, illum, lpos, lnorm);
  if (as_int(L[PL_TYPE]) == LT_CYLINDER) return cylinderLightEvalPDF(s, L, illum, lpos, lnorm, texCoord);
  if (as_int(L[PL_TYPE]) == LT_MESH) return meshLightEvalPDF(L, rayDir, lnorm, length3(sub3(illum, lpos)));
  return areaDiffuseLightEvalPDF(L, rayDir, length3(sub3(illum, lpos)));
}
static void LightSampleRev(const OrcScene* s, const float* L, f3 rands, f3 illum, ShadowSample* out) {
  switch (as_int(L[PL_TYPE])) {
    case LT_SPHERE: SphereLightSampleRev(L, rands, illum, out); break;
    case LT_MESH: MeshLightSampleRev(s, L, rands, illum, out); break;
    case LT_SKY_DOME: SkyLightSampleRev(s, L, rands, illum, out); break;
    case LT_DIRECT: DirectLightSampleRev(L, rands, illum, out); break;
    case LT_POINT_SPOT: SpotLightSampleRev(L, illum, out); break;
    case LT_POINT_OMNI: PointLightSampleRev(s, L, illum, out); break;
    case LT_CYLINDER: CylinderLightSampleRev(s, L, rands, illum, out); break;
    default: AreaLightSampleRev(s, L, rands, illum, out); break;
  }
}

/* ------------------------------------------------------------------------------------------------ f3: bidirectional building blocks */
enum { G_MPROJ = 0, G_MWORLDVIEW = 16, G_CAM_FORWARD = 208, G_CAM_UP = 211, G_IMAGE_PLANE_DIST = 217,
       HRT_FOV_X = 23, HRT_FOV_Y = 24, HRT_WIDTH_F = 25, HRT_HEIGHT_F = 26, PL_PICK_PROB_FWD = 106 };   /* cfetch.h:21-81, cglobals.h varsF names, clight.h:164 */
typedef struct { f3 pos, dir, norm, color; float pdfA, pdfW, cosTheta; int isPoint; } LightSampleFwd;   /* clight.h:632-643 */
/* ref: cglobals.h:1160-1168 */
static f3 UniformSampleSphere(float u1, float u2) {
  const float z = 1.0f - 2.0f * u1;
  const float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
  const float phi = 2.0f * ORC_PI * u2;
  return v3(r * cosf(phi), r * sinf(phi), z);
}
/* ref: clight.h:720-751 SphereLightSampleForward */
static f3 sphereLightUnitSample(float r1, float r2);
static void SphereLightSampleForward(const float* L, const float r[4], LightSampleFwd* out) {
  const f3 lcenter = lightPos(L);
  const f3 samplePos = add3(lcenter, scale3(sphereLightUnitSample(r[0], r[1]), L[SPHERE_LIGHT_RADIUS]));
  const f3 lnorm = normalize3(sub3(samplePos, lcenter));
  const f3 sampleDir = MapSampleToCosineDistribution(r[2], r[3], lnorm, lnorm, 1.0f);
  const float cosTheta = fmaxf(dot3(sampleDir, lnorm), 0.0f);
  out->isPoint = 0;
  out->pos = add3(samplePos, scale3(lnorm, epsilonOfPos(samplePos)));
  out->dir = sampleDir;
  out->color = scale3(lightColor(L), cosTheta);
  out->pdfA = 1.0f / L[PL_SURFACE_AREA];
  out->pdfW = cosTheta * INV_PI;
  out->cosTheta = cosTheta;
  out->norm = lnorm;
}
/* ref: clight.h:1023-1062 MeshLightSampleForward; r2x = rands2.x, the triangle choice */
static void MeshLightSamplePos(const OrcScene* s, const float* L, f3 rands, f3* pPos, f3* pNorm, f2* pTexCoord, float* pdfA);
static f3 meshLightGetIntensity(const OrcScene* s, const float* L, f2 tc);
static f3 meshLightMatrixMul(const float* M, f3 v);
static void MeshLightSampleForward(const OrcScene* s, const float* L, const float r[4], float r2x, LightSampleFwd* out) {
  f3 samplePos, sampleNorm; f2 tc; float pdfA;
  MeshLightSamplePos(s, L, v3(r[0], r[1], r2x), &samplePos, &sampleNorm, &tc, &pdfA);
  samplePos = meshLightMatrixMul(L + MESH_LIGHT_MATRIX_E00, samplePos);
  sampleNorm = normalize3(meshLightMatrixMul(L + MESH_LIGHT_MATRIX_E00, sampleNorm));
  samplePos = add3(samplePos, lightPos(L));
  const f3 sampleDir = MapSampleToCosineDistribution(r[2], r[3], sampleNorm, sampleNorm, 1.0f);
  const float cosTheta = fmaxf(dot3(sampleDir, sampleNorm), 0.0f);
  out->isPoint = 0;
  out->pos = add3(samplePos, scale3(sampleNorm, epsilonOfPos(samplePos)));
  out->dir = sampleDir;
  out->color = scale3(meshLightGetIntensity(s, L, tc), cosTheta);
  out->pdfA = 1.0f / L[PL_SURFACE_AREA];
  out->pdfW = cosTheta * INV_PI;
  out->cosTheta = cosTheta;
  out->norm = sampleNorm;
}
/* ref: clight.h:813-835 CylinderLightSampleForward */
static void CylinderLightSamplePos(const OrcScene* s, const float* L, f3 rands, f3* pPos, f3* pNormal, f2* pTexCoord, float* pPdfA);
static f3 cylinderLightGetIntensity(const OrcScene* s, const float* L, f2 tc);
static void CylinderLightSampleForward(const OrcScene* s, const float* L, const float r[4], float r2x, LightSampleFwd* out) {
  f3 samplePos, ln; f2 tc; float pdfA;
  CylinderLightSamplePos(s, L, v3(r[0], r[1], r2x), &samplePos, &ln, &tc, &pdfA);
  const f3 sampleDir = MapSampleToCosineDistribution(r[2], r[3], ln, ln, 1.0f);
  const float cosTheta = fmaxf(dot3(sampleDir, ln), 0.0f);
  out->isPoint = 0;
  out->pos = add3(samplePos, scale3(ln, epsilonOfPos(samplePos)));
  out->dir = sampleDir;
  out->color = scale3(cylinderLightGetIntensity(s, L, tc), cosTheta);
  out->pdfA = pdfA;
  out->pdfW = cosTheta * INV_PI;
  out->cosTheta = cosTheta;
  out->norm = ln;
}
/* ref: clight.h:654-719 AreaLightSampleForward; r2x = rands2.x, the third number of the IES table sample */
static void AreaLightSampleForward(const OrcScene* s, const float* L, const float r[4], float r2x, LightSampleFwd* out) {
  const float offsetX = r[0] * 2.0f - 1.0f, offsetY = r[1] * 2.0f - 1.0f;
  f3 sp = v3(offsetX * L[AL_SIZE_X], 0.0f, offsetY * L[AL_SIZE_Y]);
  if (as_int(L[AL_IS_DISK]) != 0) {
    f2 in = {offsetX, offsetY};
    const f2 xz = MapSamplesToDisc(in);
    sp = v3(xz.x * L[AL_SIZE_X], 0.0f, xz.y * L[AL_SIZE_X]);
  }
  const float* M = L + AL_MATRIX;
  sp = v3(M[0] * sp.x + M[1] * sp.y + M[2] * sp.z, M[3] * sp.x + M[4] * sp.y + M[5] * sp.z, M[6] * sp.x + M[7] * sp.y + M[8] * sp.z);
  sp = add3(sp, lightPos(L));
  if (as_int(L[PL_FLAGS]) & LF_IES_POINT_AREA) sp = lightPos(L);
  f3 ln = lightNorm(L);
  f3 sampleDir = MapSampleToCosineDistribution(r[2], r[3], ln, ln, 1.0f);
  float cosTheta = fmaxf(dot3(sampleDir, ln), 0.0f);
  float pdfW = cosTheta * INV_PI;
  if (as_int(L[PL_FLAGS]) & LF_HAS_IES) {
    LightSampleIESSphere(s, L, v3(r[2], r[3], r2x), &sampleDir, &pdfW);
    ln = dot3(ln, sampleDir) > 0.0f ? ln : scale3(ln, -1.0f);
  } else if (as_int(L[AL_SPOT_DISTR]) != 0) {
    const float cos2 = L[AL_SPOT_COS2];
    f2 sm = {r[2], r[3]};
    sampleDir = MapSamplesToCone(cos2, sm, ln);
    pdfW = 1.0f / (2.0f * ORC_PI * (1.0f - cos2));
  }
  cosTheta = fmaxf(dot3(sampleDir, ln), 0.0f);
  const f3 color = (as_int(L[PL_FLAGS]) & LF_SKY_PORTAL) ? mul3(lightColor(L), portalSkyColor(s, L, scale3(sampleDir, -1.0f))) : areaLightIntensity(s, L, scale3(sampleDir, -1.0f), 0);
  out->isPoint = 0;
  out->pos = add3(sp, scale3(ln, epsilonOfPos(sp)));
  out->dir = sampleDir;
  out->color = scale3(color, cosTheta);
  out->pdfA = 1.0f / L[PL_SURFACE_AREA];
  out->pdfW = pdfW;
  out->cosTheta = cosTheta;
  out->norm = ln;
}
/* ref: clight.h:838-862 PointLightSampleForward (pointLightGetIntensity :486-495 is the base colour without IES) */
static void PointLightSampleForward(const OrcScene* s, const float* L, const float r[4], LightSampleFwd* out) {
  float pdfW = INV_PI * 0.25f;
  f3 sampleDir = UniformSampleSphere(r[0], r[1]);
  const int ies = (as_int(L[PL_FLAGS]) & LF_HAS_IES) != 0;
  if (ies) LightSampleIESSphere(s, L, v3(r[0], r[1], r[2]), &sampleDir, &pdfW);
  const f3 samplePos = lightPos(L);
  const float mask = ies ? lightDistributionMask(s, L, scale3(sampleDir, -1.0f)) : 1.0f;
  out->isPoint = 1;
  out->pos = add3(samplePos, scale3(sampleDir, epsilonOfPos(samplePos)));
  out->dir = sampleDir;
  out->color = scale3(mul3(v3(mask, mask, mask), lightColor(L)), 1.0f / L[PL_SURFACE_AREA]);
  out->pdfA = 1.0f / L[PL_SURFACE_AREA];
  out->pdfW = pdfW;
  out->cosTheta = 1.0f;
  out->norm = sampleDir;
}
/* ref: clight.h:865-890 PointSpotSampleForward */
static void PointSpotSampleForward(const float* L, const float r[4], LightSampleFwd* out) {
  const f3 ln = lightNorm(L), samplePos = lightPos(L);
  const float cos1 = L[POINT_LIGHT_SPOT_COS1], cos2 = L[POINT_LIGHT_SPOT_COS2];
  f2 sm = {r[0], r[1]};
  const f3 sampleDir = MapSamplesToCone(cos2, sm, ln);
  const float cosThetaOut = fmaxf(dot3(sampleDir, ln), 0.0f);
  const float k1 = mylocalsmoothstep(cos2, cos1, cosThetaOut);
  out->isPoint = 1;
  out->pos = add3(samplePos, scale3(sampleDir, epsilonOfPos(samplePos)));
  out->dir = sampleDir;
  out->color = scale3(scale3(lightColor(L), k1), 1.0f / L[PL_SURFACE_AREA]);
  out->pdfA = 1.0f / L[PL_SURFACE_AREA];
  out->pdfW = 1.0f / (2.0f * ORC_PI * (1.0f - cos2));
  out->cosTheta = cosThetaOut;
  out->norm = sampleDir;
}
/* ref: clight.h:915-958 DirectLightSampleForward */
static void DirectLightSampleForward(const float* L, const float r[4], LightSampleFwd* out) {
  const f3 ln = lightNorm(L), lcenter = lightPos(L);
  const float radius1 = L[DIRECT_LIGHT_RADIUS1], radius2 = L[DIRECT_LIGHT_RADIUS2];
  f2 in = {2.0f * (r[0] - 0.5f), 2.0f * (r[1] - 0.5f)};
  const f2 d0 = MapSamplesToDisc(in);
  const f2 diskSam = {radius2 * d0.x, radius2 * d0.y};
  const float d = sqrtf(diskSam.x * diskSam.x + diskSam.y * diskSam.y);
  const float atten = mylocalsmoothstep(fmaxf(radius2, radius1), fminf(radius2, radius1), d);
  f3 nx, nz;
  CoordinateSystem(ln, &nx, &nz);
  const f3 samplePos = add3(add3(lcenter, scale3(nx, diskSam.x)), scale3(nz, diskSam.y));
  f3 sampleDir = ln;
  const float pdfW = 1.0f;
  if (L[DIRECT_LIGHT_SSOFTNESS] > 1e-5f) { f2 sm = {r[2], r[3]}; sampleDir = MapSamplesToCone(L[DIRECT_LIGHT_ALPHA_COS], sm, ln); }
  out->isPoint = 1;
  out->pos = add3(samplePos, scale3(sampleDir, epsilonOfPos(samplePos)));
  out->dir = sampleDir;
  out->color = scale3(scale3(lightColor(L), atten), pdfW);
  out->pdfA = 1.0f / L[PL_SURFACE_AREA];
  out->pdfW = pdfW;
  out->cosTheta = 1.0f;
  out->norm = sampleDir;
}
/* ref: clight.h:1064-1110 LightSampleForward over the light types the layer accepts -> out16 = pos, dir, norm, colour, pdfA, pdfW, cosTheta, isPoint */
void orc_light_sample_forward(const OrcScene* s, int n, const int32_t* lightIds, const float* rands4, float* out16) {
  for (int i = 0; i < n; i++) {
    const float* L = lightAt(s, lightIds[i]);
    LightSampleFwd sam;
    switch (as_int(L[PL_TYPE])) {
      case LT_MESH: MeshLightSampleForward(s, L, rands4 + 4 * i, 0.0f, &sam); break;   /* rands2 = (0, 0), as the reference-side fixture hands in */
      case LT_SPHERE: SphereLightSampleForward(L, rands4 + 4 * i, &sam); break;
      case LT_DIRECT: DirectLightSampleForward(L, rands4 + 4 * i, &sam); break;
      case LT_POINT_SPOT: PointSpotSampleForward(L, rands4 + 4 * i, &sam); break;
      case LT_POINT_OMNI: PointLightSampleForward(s, L, rands4 + 4 * i, &sam); break;
      case LT_CYLINDER: CylinderLightSampleForward(s, L, rands4 + 4 * i, 0.0f, &sam); break;
      default: AreaLightSampleForward(s, L, rands4 + 4 * i, 0.0f, &sam); break;
    }
    float* o = out16 + 16 * (size_t)i;
    o[0] = sam.pos.x; o[1] = sam.pos.y; o[2] = sam.pos.z; o[3] = sam.dir.x; o[4] = sam.dir.y; o[5] = sam.dir.z;
    o[6] = sam.norm.x; o[7] = sam.norm.y; o[8] = sam.norm.z; o[9] = sam.color.x; o[10] = sam.color.y; o[11] = sam.color.z;
    o[12] = sam.pdfA; o[13] = sam.pdfW; o[14] = sam.cosTheta; o[15] = sam.isPoint ? 1.0f : 0.0f;
  }
}
/* ref: clight.h:1117-1175 lightPdfFwd (no IES) -> out4 = pdfA, pdfW, pickProb, 0 */
void orc_light_pdf_fwd(const OrcScene* s, int n, const int32_t* lightIds, const float* cosTheta, float* out4) {
  for (int i = 0; i < n; i++) {
    const float* L = lightAt(s, lightIds[i]);
    const float ct = cosTheta[i];
    float pdfA = 1.0f / L[PL_SURFACE_AREA], pdfW = fmaxf(ct * INV_PI, 0.0f);
    const int ltype = as_int(L[PL_TYPE]);
    if (ltype == LT_POINT_OMNI) pdfW = INV_PI * 0.25f;
    else if (ltype == LT_POINT_SPOT) {
      const float cos2 = L[POINT_LIGHT_SPOT_COS2];
      pdfW = 1.0f / (2.0f * ORC_PI * (1.0f - cos2));
      if (ct < cos2) pdfW = 0.0f;
    } else if (ltype == LT_DIRECT) {
      const float radius2 = L[DIRECT_LIGHT_RADIUS2];
      pdfA = 1.0f / (ORC_PI * radius2 * radius2);
      pdfW = 0.0f;
    }
    if (as_int(L[PL_FLAGS]) & LF_HAS_IES) pdfW = lightPdfFwdIES(s, L, v3(0.0f, 0.0f, 1.0f));   /* the direction the reference-side fixture hands in */
    else if (ltype == LT_AREA && as_int(L[AL_SPOT_DISTR]) != 0) {
      const float cos2 = L[AL_SPOT_COS2];
      pdfW = 1.0f / (2.0f * ORC_PI * (1.0f - cos2));
      if (ct < cos2) pdfW = 0.0f;
    }
    out4[4 * i + 0] = pdfA; out4[4 * i + 1] = pdfW; out4[4 * i + 2] = L[PL_PICK_PROB_FWD]; out4[4 * i + 3] = 0.0f;
  }
}
/* ref: cbidir.h:78-115 CameraImageToSurfaceFactor, :117-131 worldPosToScreenSpace -> out8 = factor, camDir xyz, zDepth, screen xy, 0 */
void orc_camera_connect(const OrcScene* s, int n, const float* pos4, const float* norm4, const float* disk2, float* out8) {
  const float* gf = (const float*)s->globals;
  const m44 wvInv = load_m44(gf + G_MWORLDVIEW_INV), wv = load_m44(gf + G_MWORLDVIEW), proj = load_m44(gf + G_MPROJ);
  const f3 camForward = v3(gf[G_CAM_FORWARD], gf[G_CAM_FORWARD + 1], gf[G_CAM_FORWARD + 2]);
  const f3 camUp = v3(gf[G_CAM_UP], gf[G_CAM_UP + 1], gf[G_CAM_UP + 2]);
  const f3 camLeft = normalize3(cross3(camForward, camUp));
  const float imagePlaneDist = gf[G_IMAGE_PLANE_DIST], lensR = g_varsF(s)[HRT_DOF_LENS_RADIUS];
  const float fw = g_varsF(s)[HRT_WIDTH_F], fh = g_varsF(s)[HRT_HEIGHT_F];
  for (int i = 0; i < n; i++) {
    const f3 hitPos = v3(pos4[4 * i], pos4[4 * i + 1], pos4[4 * i + 2]), hitNorm = v3(norm4[4 * i], norm4[4 * i + 1], norm4[4 * i + 2]);
    const float dx = disk2[2 * i], dy = disk2[2 * i + 1];
    const f3 camPos = add3(add3(mul4x3(wvInv, v3(0, 0, 0)), scale3(scale3(camUp, dy), lensR)), scale3(scale3(camLeft, dx), lensR));
    const float zDepth = length3(sub3(camPos, hitPos));
    const f3 camDir = scale3(sub3(camPos, hitPos), 1.0f / zDepth);
    const float cosToCamera = fabsf(dot3(hitNorm, camDir));
    const float cosAtCamera = dot3(camForward, scale3(camDir, -1.0f));
    const float relation = fw / fh;
    const float fov = relation * fmaxf(g_varsF(s)[HRT_FOV_X], g_varsF(s)[HRT_FOV_Y]);
    float factor = 0.0f;
    if (!(cosAtCamera <= cosf(fov))) {
      const float imagePointToCameraDist = imagePlaneDist / cosAtCamera;
      const float imageToSolidAngleFactor = (imagePointToCameraDist * imagePointToCameraDist) / cosAtCamera;
      const float imageToSurfaceFactor = imageToSolidAngleFactor * cosToCamera / (zDepth * zDepth);
      factor = isfinite(imageToSurfaceFactor) ? imageToSurfaceFactor / (relation * relation) : 0.0f;
    }
    const f4 pw = {hitPos.x, hitPos.y, hitPos.z, 1.0f};
    const f4 ndc = mul4x4x4(proj, mul4x4x4(wv, pw));
    const float inv = 1.0f / fmaxf(ndc.w, DEPSILON);
    float* o = out8 + 8 * (size_t)i;
    o[0] = factor; o[1] = camDir.x; o[2] = camDir.y; o[3] = camDir.z; o[4] = zDepth;
    o[5] = (ndc.x * inv * 0.5f + 0.5f) * fw; o[6] = (ndc.y * inv * 0.5f + 0.5f) * fh; o[7] = 0.0f;
  }
}
/* ref: crandom.h:189-210 MutateKelemen */
void orc_mutate_kelemen(int n, const float* values, const float* rands2, float p2, float p1, float* out) {
  const float s1 = 1.0f / p1, s2 = 1.0f / p2;
  const float power = -logf(s2 / s1);
  for (int i = 0; i < n; i++) {
    float x = values[i];
    const float dv = fmaxf(s2 * (expf(power * sqrtf(rands2[2 * i])) - expf(power)), 0.0f);
    if (rands2[2 * i + 1] < 0.5f) { x += dv; if (x > 1.0f) x -= 1.0f; }
    else { x -= dv; if (x < 0.0f) x += 1.0f; }
    out[i] = x;
  }
}

/* ---- Perez all-weather sky, ref: clight.h:178-282 (perezZenith, perezFunc, perezSky, convertColor, skyLightPerezColor) ---- */
enum { SKY_DOME_SUN_DIR = 23, SKY_DOME_TURBIDITY = 26, SKY_SUN_COLOR = 27, SKY_LIGHT_USE_PEREZ = 4 };   /* clight.h:143-151, cglobals.h:2249 */
static f3 perezZenith(float t, float thetaSun) {
  const float pi = 3.1415926f;
  const float t2 = t * t;
  const float chi = (4.0f / 9.0f - t / 120.0f) * (pi - 2.0f * thetaSun);
  const float th[4] = {1.0f, thetaSun, thetaSun * thetaSun, thetaSun * thetaSun * thetaSun};
  static const float cx1[4] = {0.0f, 0.00209f, -0.00375f, 0.00165f}, cx2[4] = {0.00394f, -0.03202f, 0.06377f, -0.02903f}, cx3[4] = {0.25886f, 0.06052f, -0.21196f, 0.11693f};
  static const float cy1[4] = {0.0f, 0.00317f, -0.00610f, 0.00275f}, cy2[4] = {0.00516f, -0.04153f, 0.08970f, -0.04214f}, cy3[4] = {0.26688f, 0.06670f, -0.26756f, 0.15346f};
#define ORC_DOT4(c) ((c)[0] * th[0] + (c)[1] * th[1] + (c)[2] * th[2] + (c)[3] * th[3])
  const float Y = (4.0453f * t - 4.9710f) * tanf(chi) - 0.2155f * t + 2.4192f;
  const float x = t2 * ORC_DOT4(cx1) + t * ORC_DOT4(cx2) + ORC_DOT4(cx3);
  const float y = t2 * ORC_DOT4(cy1) + t * ORC_DOT4(cy2) + ORC_DOT4(cy3);
#undef ORC_DOT4
  return v3(Y, x, y);
}
static f3 perezFunc(float t, float cosTheta, float cosGamma) {
  const float gamma = acosf(cosGamma), cosGammaSq = cosGamma * cosGamma;
  const float aY = 0.17872f * t - 1.46303f, bY = -0.35540f * t + 0.42749f, cY = -0.02266f * t + 5.32505f, dY = 0.12064f * t - 2.57705f, eY = -0.06696f * t + 0.37027f;
  const float ax = -0.01925f * t - 0.25922f, bx = -0.06651f * t + 0.00081f, cx = -0.00041f * t + 0.21247f, dx = -0.06409f * t - 0.89887f, ex = -0.00325f * t + 0.04517f;
  const float ay = -0.01669f * t - 0.26078f, by = -0.09495f * t + 0.00921f, cy = -0.00792f * t + 0.21023f, dy = -0.04405f * t - 1.65369f, ey = -0.01092f * t + 0.05291f;
  return v3((1.0f + aY * expf(bY / cosTheta)) * (1.0f + cY * expf(dY * gamma) + eY * cosGammaSq),
            (1.0f + ax * expf(bx / cosTheta)) * (1.0f + cx * expf(dx * gamma) + ex * cosGammaSq),
            (1.0f + ay * expf(by / cosTheta)) * (1.0f + cy * expf(dy * gamma) + ey * cosGammaSq));
}
static f3 perezSky(float turbidity, float cosTheta, float cosGamma, float cosThetaSun) {
  const f3 a = mul3(perezZenith(turbidity, acosf(cosThetaSun)), perezFunc(turbidity, cosTheta, cosGamma)), b = perezFunc(turbidity, 1.0f, cosThetaSun);
  return v3(a.x / b.x, a.y / b.y, a.z / b.z);
}
static f3 perezConvertColor(f3 c) {
  c.x = 1.0f - expf(-c.x / 20.0f);
  const float ratio = c.x / fmaxf(c.z, 1e-10f);
  f3 XYZ;
  XYZ.x = c.y * ratio;
  XYZ.y = c.x;
  XYZ.z = ratio - XYZ.x - XYZ.y;
  return clamp3(v3(dot3(v3(3.240479f, -1.53715f, -0.49853f), XYZ), dot3(v3(-0.969256f, 1.875991f, 0.041556f), XYZ), dot3(v3(0.055684f, -0.204043f, 1.057311f), XYZ)), 0.0f, 1.0f);
}
static f3 skyLightPerezColor(const float* L, f3 ray_dir) {
  const f3 sunDir = v3(L[SKY_DOME_SUN_DIR], L[SKY_DOME_SUN_DIR + 1], L[SKY_DOME_SUN_DIR + 2]);
  const float turbidity = L[SKY_DOME_TURBIDITY];
  const f3 colorYxy = perezSky(turbidity, fmaxf(ray_dir.y, 0.0f) + 0.05f, fmaxf(dot3(sunDir, scale3(ray_dir, -1.0f)), 0.0f), fmaxf(-sunDir.y, 0.0f));
  f3 rgb = perezConvertColor(colorYxy);
  rgb.x = powf(rgb.x, 2.2f); rgb.y = powf(rgb.y, 2.2f); rgb.z = powf(rgb.z, 2.2f);
  const float tSunAngle = fmaxf(-sunDir.y, 0.0f);
  const float threshold = 0.9985f + tSunAngle * (0.9995f - 0.9985f);
  const float tSun = dot3(sunDir, scale3(ray_dir, -1.0f));
  if (tSun >= threshold) {
    const f3 sunColor = scale3(v3(L[SKY_SUN_COLOR], L[SKY_SUN_COLOR + 1], L[SKY_SUN_COLOR + 2]), 2.0f + 2.0f * tSunAngle);
    float tSun2 = (tSun - threshold) / (1.0f - threshold);
    tSun2 = tSun2 * tSun2;
    rgb = add3(scale3(sunColor, tSun2), scale3(rgb, 1.0f - tSun2));
  }
  return rgb;
}
/* ref: cbidir.h:492-533 environmentColor; misPrev.prevMaterialOffset is -1 on this path (PT_Loop.cpp:247-249) */
static f3 environmentColor(const OrcScene* s, f3 rayDir, float prevPdf, int prevSpecular, uint32_t flags) {
  const int skyId = s->globals[G_SKY_LIGHT_ID];
  if (skyId == -1) return v3(0, 0, 0);
  const float* L = (const float*)(s->globals + s->globals[G_LIGHTS_OFFS]) + (size_t)skyId * LIGHT_FLOATS;
  float sintheta = 0.0f;
  const f2 tc = sphereMapTo2DTexCoord(rayDir, &sintheta);   /* skyLightGetIntensityTexturedENV, clight.h:285-306 */
  f3 envColor = mul3(v3(L[PL_COLOR], L[PL_COLOR + 1], L[PL_COLOR + 2]), sample2DExt(as_int(L[PL_COLOR_TEX_MATRIX]), tc, L + SKY_DOME_SAMPLER0, s));
  if (as_int(L[PL_FLAGS]) & SKY_LIGHT_USE_PEREZ) envColor = mul3(v3(L[PL_COLOR], L[PL_COLOR + 1], L[PL_COLOR + 2]), skyLightPerezColor(L, rayDir));
  const uint32_t rayBounceNum = (flags >> 8) & 0xFFu;
  if (rayBounceNum > 0 && !((uint32_t)s->globals[G_FLAGS] & HRT_STUPID_PT_MODE) && !prevSpecular) {
    const float lgtPdf = L[PL_PICK_PROB_REV] * skyLightEvalPDF(s, L, rayDir);
    envColor = scale3(envColor, misWeightHeuristic(prevPdf, lgtPdf));
  }
  return envColor;
}

/* ------------------------------------------------------------------------------------------------ path tracer */
typedef struct { float matSamplePdf; int isSpecular; } MisData;   /* ref: cglobals.h:1382-1400 (fields the PT path reads) */

/* the sky a portal names sits AREA_LIGHT_SKY_OFFSET records away (RenderDriverRTE.cpp:1670-1682); a Perez sky counts half (clight.h:598-601, 622-625) */
static f3 portalSkyColor(const OrcScene* s, const float* L, f3 rayDir) {
  const float* sky = L + (ptrdiff_t)as_int(L[AREA_LIGHT_SKY_OFFSET]) * LIGHT_FLOATS;
  if (as_int(sky[PL_FLAGS]) & SKY_LIGHT_USE_PEREZ) return scale3(skyLightPerezColor(sky, rayDir), 0.5f);
  float sintheta = 0.0f;
  const f2 tc = sphereMapTo2DTexCoord(rayDir, &sintheta);   /* skyLightGetIntensityTexturedENV, clight.h:285-306 */
  return mul3(lightColor(sky), sample2DExt(as_int(sky[PL_COLOR_TEX_MATRIX]), tc, sky + SKY_DOME_SAMPLER0, s));
}
/* ref: clight.h:1636-1657 hitDirectLight, :892-912 directLightAttenuation, :1462-1476 directLightEvalPDF */
static int hitDirectLight(const OrcScene* s, f3 ray_dir) {
  for (int sunId = 0; sunId < s->globals[G_SUN_NUMBER]; sunId++) {
    const float* sun = (const float*)(s->globals + G_SUNS) + (size_t)sunId * LIGHT_FLOATS;
    if (-dot3(ray_dir, lightNorm(sun)) > sun[DIRECT_LIGHT_ALPHA_COS]) return sunId;
  }
  return -1;
}
static float directLightAttenuation(const float* L, f3 illum) {
  const f3 lpos = lightPos(L);
  const float cos_alpha = dot3(normalize3(sub3(illum, lpos)), lightNorm(L));
  if (cos_alpha > 0.0f) {
    const float sinAlpha = sqrtf(1.0f - cos_alpha * cos_alpha);
    const float d = length3(sub3(illum, lpos)) * sinAlpha;
    const float r1 = L[DIRECT_LIGHT_RADIUS1], r2 = L[DIRECT_LIGHT_RADIUS2];
    return mylocalsmoothstep(fmaxf(r2, r1), fminf(r2, r1), d);
  }
  return 0.0f;
}
static float directLightEvalPDF(const float* L, f3 ray_dir) {
  if (L[DIRECT_LIGHT_SSOFTNESS] > 1e-5f) {
    const float tanAlpha = L[DIRECT_LIGHT_ALPHA_TAN], cosTheta = -dot3(ray_dir, lightNorm(L));
    return ORC_PI * (tanAlpha * tanAlpha) * (cosTheta * cosTheta * cosTheta);
  }
  return 1.0f;
}
/* ref: clight.h:1661-1706 lightGetIntensity */
/* ---- the back-plate.  ref: cbidir.h:543-573 backColorOfSecondEnv, :593-629 environmentColorExtended -- the miss shader of the OpenCL layer (HitEnvOrLightKernel,
 * shaders/material.cl:354).  The CPU integrator calls plain environmentColor (PT_Loop.cpp:28) and knows no back-plate; PathTrace takes the extended form when, and only
 * when, the header names a back texture (HRT_SHADOW_MATTE_BACK), like the HIP layer does.  The pixel of the path comes from the caller (thread-local). */
enum { HRT_SHADOW_MATTE_BACK = 35, HRT_SHADOW_MATTE_BACK_MODE = 41, HRT_SHADOW_MATTE_BACK_COLOR_X = 42, HRT_BACK_TEXINPUT_GAMMA = 36, HRT_3WAY_MIS_WEIGHTS = 1024 };   /* cglobals.h:416, 475-484, 537 */
static _Thread_local int g_screenX = 0, g_screenY = 0;
static inline int haveBackPlate(const OrcScene* s) { return (uint32_t)g_varsI(s)[HRT_SHADOW_MATTE_BACK] != INVALID_TEXTURE; }
static int orc_have_back_plate(const OrcScene* s) { return haveBackPlate(s); }
static f3 backColorOfSecondEnv(const OrcScene* s, f3 ray_dir, float screenX, float screenY) {
  const float* vf = g_varsF(s);
  const int offset = s->globals[s->globals[G_TEX_TABLE] + g_varsI(s)[HRT_SHADOW_MATTE_BACK]];
  const f3 mult = v3(vf[HRT_SHADOW_MATTE_BACK_COLOR_X], vf[HRT_SHADOW_MATTE_BACK_COLOR_X + 1], vf[HRT_SHADOW_MATTE_BACK_COLOR_X + 2]);
  f2 tc = {screenX / vf[HRT_WIDTH_F], screenY / vf[HRT_HEIGHT_F]};
  if (g_varsI(s)[HRT_SHADOW_MATTE_BACK_MODE] == 1) { float sintheta = 0.0f; tc = sphereMapTo2DTexCoord(ray_dir, &sintheta); }
  const f4 c = read_imagef_sw4(s->texStorage + (size_t)offset * 4, tc, TEX_CLAMP_U | TEX_CLAMP_V, 1);
  f3 env = mul3(mult, v3(c.x, c.y, c.z));
  if (vf[HRT_BACK_TEXINPUT_GAMMA] != 1.0f) env = v3(sRGBToLinear(env.x), sRGBToLinear(env.y), sRGBToLinear(env.z));
  return env;
}
static f3 environmentColor(const OrcScene* s, f3 rayDir, float prevPdf, int prevSpecular, uint32_t flags);
static f3 environmentColorExtended(const OrcScene* s, f3 ray_pos, f3 ray_dir, float prevPdf, int prevSpecular, uint32_t flags, int screenX, int screenY) {
  const int hitId = hitDirectLight(s, ray_dir);
  if (hitId >= 0) {
    const float* sun = (const float*)(s->globals + G_SUNS) + (size_t)hitId * LIGHT_FLOATS;
    f3 envColor = scale3(lightColor(sun), directLightAttenuation(sun, ray_pos));
    const float pdfW = directLightEvalPDF(sun, ray_dir);
    const uint32_t gflags = (uint32_t)s->globals[G_FLAGS];
    if (((flags >> 8) & 0xFFu) > 0 && !(gflags & HRT_STUPID_PT_MODE) && !prevSpecular) envColor = v3(0, 0, 0);
    else if ((prevSpecular && (gflags & HRT_ENABLE_PT_CAUSTICS)) || (gflags & HRT_STUPID_PT_MODE)) envColor = scale3(envColor, 1.0f / pdfW);
    if (gflags & HRT_3WAY_MIS_WEIGHTS) envColor = v3(0, 0, 0);
    return envColor;
  }
  f3 envColor = environmentColor(s, ray_dir, prevPdf, prevSpecular, flags);
  const uint32_t rayBounce = (flags >> 8) & 0xFFu, other = flags >> 16;
  const int transparent = (other & 8u) != 0 && (other & 2u) == 0 && (other & 4u) == 0;   /* RAY_EVENT_T without _D and _G */
  if (rayBounce == 0 || transparent) envColor = backColorOfSecondEnv(s, ray_dir, (float)screenX + 0.5f, (float)screenY + 0.5f);
  return envColor;
}
/* the miss shader on rays handed in (layout: include/hydra_hip.h, hydra_hip_stage_environment) */
void orc_stage_environment(const OrcScene* s, int n, const float* dir4, const float* in8, float* out4) {
  for (int i = 0; i < n; i++) {
    const float* in = in8 + 8 * (size_t)i;
    const f3 d = v3(dir4[4 * i], dir4[4 * i + 1], dir4[4 * i + 2]);
    const f3 c = haveBackPlate(s) ? environmentColorExtended(s, v3(in[0], in[1], in[2]), d, in[3], in[4] != 0.0f, (uint32_t)as_int(in[5]), as_int(in[6]), as_int(in[7]))
                                  : environmentColor(s, d, in[3], in[4] != 0.0f, (uint32_t)as_int(in[5]));
    out4[4 * i] = c.x; out4[4 * i + 1] = c.y; out4[4 * i + 2] = c.z; out4[4 * i + 3] = 0.0f;
  }
}
static f3 lightGetIntensity(const OrcScene* s, const float* L, f3 ray_pos, f3 ray_dir, f2 texCoord, uint32_t flags, int wasSpecular) {
  const int eyeRay = ((flags & 0xFFu) == 0);
  const int type = as_int(L[PL_TYPE]);
  if ((as_int(L[PL_FLAGS]) & LF_SKY_PORTAL) && (flags & 0xFFu) > 0) {
    const int hitId = hitDirectLight(s, ray_dir);
    if (hitId >= 0) {
      const float* sun = (const float*)(s->globals + G_SUNS) + (size_t)hitId * LIGHT_FLOATS;
      f3 sunColor = scale3(lightColor(sun), directLightAttenuation(sun, ray_pos));
      const float pdfW = directLightEvalPDF(sun, ray_dir);
      const uint32_t gflags = (uint32_t)s->globals[G_FLAGS];
      if (((flags >> 8) & 0xFFu) > 0 && !(gflags & HRT_STUPID_PT_MODE) && !wasSpecular) sunColor = v3(0, 0, 0);
      else if ((wasSpecular && (gflags & HRT_ENABLE_PT_CAUSTICS)) || (gflags & HRT_STUPID_PT_MODE)) sunColor = scale3(sunColor, 1.0f / pdfW);
      return sunColor;
    }
    return mul3(lightColor(L), portalSkyColor(s, L, ray_dir));
  }
  if (type == LT_AREA) {
    f3 customDir = ray_dir;
    if (as_int(L[PL_FLAGS]) & LF_IES_POINT_AREA) customDir = normalize3(sub3(lightPos(L), ray_pos));
    f3 color = areaLightIntensity(s, L, customDir, eyeRay);
    if (as_int(L[PL_FLAGS]) & LF_SKY_PORTAL) color = mul3(color, portalSkyColor(s, L, ray_dir));
    return color;
  }
  if (type == LT_CYLINDER) return cylinderLightGetIntensity(s, L, texCoord);
  if (type == LT_MESH) return meshLightGetIntensity(s, L, texCoord);
  return lightColor(L);
}
/* ref: cbidir.h:653-678 emissionEval + CPUExp_Integrators_Common.cpp:516-527 */
static f3 emissionEval(const OrcScene* s, f3 ray_pos, f3 ray_dir, const SurfaceHit* sh, uint32_t flags, int wasSpecular, const float* pLight, const float* mat) {
  const f3 normal = sh->hfi ? scale3(sh->normal, -1.0f) : sh->normal;
  int hasIES = 0;
  const int lightsNum = s->globals[G_LIGHTS_NUM];
  if (lightsNum > 0 && pLight != NULL) hasIES = (as_int(pLight[PL_FLAGS]) & LF_HAS_IES) != 0;
  if (dot3(ray_dir, normal) >= 0.0f && !hasIES) return v3(0, 0, 0);
  f3 out = materialEvalEmission(mat, ray_dir, normal, sh->texCoord, s);
  if ((matFlags(mat) & MF_FORBID_EMISSIVE_GI) && (flags & 0xFFu) > 0) out = v3(0, 0, 0);
  if (lightsNum > 0 && pLight != NULL) out = lightGetIntensity(s, pLight, ray_pos, ray_dir, sh->texCoord, flags, wasSpecular);
  return out;
}

typedef struct { uint64_t rays; } PathStat;

/* optional ray recorder used by orc_collect_rays */
typedef struct { int bounce, shadow; f3 pos, dir; float tfar; int have; } RayProbe;

/* ---- the stages of IntegratorMISPTLoop2 (ref: CPUExp_Integrators_PT_Loop.cpp:9-262), one function each; PathTrace below strings them
 * together with the generator, orc_stage_bounce runs them on handed-in inputs (fixtures made by the reference's own stage kernels) */

/* kernel_EvalEmission :86-139.  Returns 1 when the path ends on an emitter (currColor = its radiance after MIS), else 0. */
static int stage_emission(const OrcScene* s, f3 ray_pos, f3 ray_dir, const SurfaceHit* surf, const float* mat, int hitInstId, uint32_t flags, MisData misPrev, f3* currColor) {
  const int lightOffset0 = (s->globals[G_LIGHTS_NUM] != 0) ? s->instLightInstId[hitInstId] : -1;
  const float* pLightHit = lightAt(s, lightOffset0);
  const f3 emission = emissionEval(s, ray_pos, ray_dir, surf, flags, misPrev.isSpecular == 1, pLightHit, mat);
  if (!(dot3(emission, emission) > 1e-3f)) return 0;
  if (pLightHit != NULL) {
    const float lgtPdf = pLightHit[PL_PICK_PROB_REV] * lightEvalPDF(s, pLightHit, ray_pos, ray_dir, surf->pos, surf->normal, surf->texCoord);
    float misWeight = misWeightHeuristic(misPrev.matSamplePdf, lgtPdf);
    if (misPrev.isSpecular) misWeight = 1.0f;
    *currColor = scale3(emission, misWeight);
  } else
    *currColor = emission;
  return 1;
}
/* kernel_LightSelect :141-151 + kernel_LightSample :154-168 + the far end of kernel_ShadowTrace's ray :170-179; rl = rndLight's four numbers,
 * pickRand = the one that picks the light (the CPU path passes rl[2]) */
static void stage_light(const OrcScene* s, const SurfaceHit* surf, const float rl[4], float pickRand, float* lightPickProb, int* lightOffset,
                        ShadowSample* explicitSam, f3* shadowRayPos, f3* shadowRayDir, float* tfar) {
  *lightPickProb = 1.0f;
  *lightOffset = SelectRandomLightRev(pickRand, s, lightPickProb);
  *shadowRayPos = v3(0, 0, 0); *shadowRayDir = v3(0, 0, 0); *tfar = -1.0f;
  memset(explicitSam, 0, sizeof(*explicitSam));
  if (*lightOffset >= 0) {
    const float* pl = lightAt(s, *lightOffset);   /* LightSampleRev, clight.h:1561-1610 */
    LightSampleRev(s, pl, v3(rl[0], rl[1], rl[2]), surf->pos, explicitSam);
    *shadowRayDir = normalize3(sub3(explicitSam->pos, surf->pos));
    *shadowRayPos = OffsShadowRayPos(surf->pos, surf->normal, *shadowRayDir, surf->sRayOff);
    *tfar = length3(sub3(*shadowRayPos, explicitSam->pos)) * 0.995f;
  }
}
/* kernel_Shade :181-216 */
static f3 stage_shade(const OrcScene* s, const float* mat, const SurfaceHit* surf, f3 ray_dir, f3 shadowRayDir, const ShadowSample* explicitSam,
                      float lightPickProb, int lightOffset, float shadow) {
  if (lightOffset < 0) return v3(0, 0, 0);
  ShadeContext sc;
  sc.l = shadowRayDir; sc.v = scale3(ray_dir, -1.0f); sc.n = surf->normal; sc.fn = surf->flatNormal;
  sc.tg = surf->tangent; sc.bn = surf->biTangent; sc.tc = surf->texCoord;
  const BxDFResult ev = materialEval(mat, &sc, s);
  const float cos1 = fmaxf(+dot3(shadowRayDir, surf->normal), 0.0f), cos2 = fmaxf(-dot3(shadowRayDir, surf->normal), 0.0f);
  const f3 bxdfVal = add3(scale3(ev.brdf, cos1), scale3(ev.btdf, cos2));
  const float lgtPdf = explicitSam->pdf * lightPickProb;
  float misWeight = misWeightHeuristic(lgtPdf, ev.pdfFwd);
  if (explicitSam->isPoint) misWeight = 1.0f;
  const f3 lc = scale3(explicitSam->color, (1.0f / fmaxf(explicitSam->pdf, DEPSILON2)));
  return scale3(scale3(mul3(scale3(lc, (1.0f / lightPickProb)), bxdfVal), misWeight), shadow);
}
/* kernel_NextBounce :218-256 with RndMatAll's numbers handed in */
static void stage_next(const OrcScene* s, const float* mat, const SurfaceHit* surf, const float* allRands, f3 explicitColor,
                       f3* ray_pos, f3* ray_dir, uint32_t* flags, MisData* misPrev, f3* accumColor, f3* thoroughput, float shadow) {
  MatSample ms;
  g_matteShadow = shadow;
  MaterialSampleAndEvalBxDF(mat, allRands, surf, *ray_dir, *flags, s, &ms);
  g_matteShadow = 0.0f;
  const f3 bxdfVal = scale3(ms.color, (1.0f / fmaxf(ms.pdf, 1e-20f)));
  const float cosTheta = fabsf(dot3(ms.direction, surf->normal));
  *ray_dir = ms.direction;
  *ray_pos = OffsRayPos(surf->pos, surf->normal, ms.direction);
  misPrev->isSpecular = ((ms.flags & RAY_EVENT_S) != 0 || (ms.flags & RAY_EVENT_T) != 0);
  misPrev->matSamplePdf = ms.pdf;
  *flags = flagsNextBounceLite(*flags, &ms, s);
  *accumColor = add3(*accumColor, mul3(*thoroughput, explicitColor));
  *thoroughput = mul3(*thoroughput, scale3(bxdfVal, cosTheta));
}

/* ref: CPUExp_Integrators_PT_Loop.cpp:264-321 IntegratorMISPTLoop2::PathTrace with its kernel_* stages :9-262 */
static f3 PathTrace(const OrcScene* s, f3 ray_pos, f3 ray_dir, uint32_t gen[2], PathStat* st, RayProbe* probe) {
  f3 accumColor = v3(0, 0, 0), thoroughput = v3(1, 1, 1), currColor = v3(0, 0, 0);
  MisData misPrev = {1.0f, 1};   /* makeInitialMisData */
  uint32_t flags = 0;
  const int maxDepth = g_varsI(s)[HRT_TRACE_DEPTH];   /* SetMaxDepth(varsI[HRT_TRACE_DEPTH]), CPUExpLayer.cpp:130 */

  for (int depth = 0; depth < maxDepth; depth++) {
    /* kernel_RayTrace */
    if (probe && !probe->shadow && probe->bounce == depth) { probe->pos = ray_pos; probe->dir = ray_dir; probe->have = 1; }
    const OrcHit hit = rayTrace(s, ray_pos, ray_dir, NULL);
    st->rays++;
    /* kernel_HitEnvironment :23-33 */
    if (!HitSome(hit)) {
      currColor = haveBackPlate(s) ? environmentColorExtended(s, ray_pos, ray_dir, misPrev.matSamplePdf, misPrev.isSpecular, flags, g_screenX, g_screenY)
                                   : environmentColor(s, ray_dir, misPrev.matSamplePdf, misPrev.isSpecular, flags);
      break;
    }
    /* kernel_EvalSurface */
    const SurfaceHit surf = evalSurface(s, ray_pos, ray_dir, hit);
    const float* mat = materialAt(s, surf.matId);
    OrcPtl ptlOfHit;
    g_ptl = NULL;
    if (g_proctexFn != NULL && (as_int(mat[1]) & MF_HAVE_PROC_TEXTURES)) {   /* ProcTexExec, shaders/texproc.cl:127-190 */
      const m44 worldToObject = load_m44(s->instMatrices + (size_t)hit.instId * 16);
      const f3 lp = mul4x3(worldToObject, surf.pos);
      const float surf19[19] = {surf.pos.x, surf.pos.y, surf.pos.z, lp.x, lp.y, lp.z, surf.normal.x, surf.normal.y, surf.normal.z, surf.tangent.x, surf.tangent.y, surf.tangent.z,
                                surf.biTangent.x, surf.biTangent.y, surf.biTangent.z, surf.texCoord.x, surf.texCoord.y, 1.0f, 1.0f};
      const float view3[3] = {ray_dir.x, ray_dir.y, ray_dir.z};
      int count = 0;
      float vals48[48];
      g_proctexFn(g_proctexUser, surf19, mat, view3, &count, ptlOfHit.ids, vals48);
      ptlOfHit.n = count < 16 ? count : 16;
      for (int k = 0; k < ptlOfHit.n; k++) for (int c = 0; c < 3; c++) ptlOfHit.vals[k][c] = half_bits_to_float(float_to_half_bits(vals48[3 * k + c]));
      g_ptl = &ptlOfHit;
    }
    if (stage_emission(s, ray_pos, ray_dir, &surf, mat, hit.instId, flags, misPrev, &currColor)) break;
    else if (depth >= maxDepth - 1) { currColor = v3(0, 0, 0); break; }
    float rl[4];
    orc_rnd_float4(gen, rl);   /* rndLight, crandom.h:404-418 (pseudo-random branch) */
    float lightPickProb, tfar;
    int lightOffset;
    f3 shadowRayPos, shadowRayDir;
    ShadowSample explicitSam;
    stage_light(s, &surf, rl, rl[2], &lightPickProb, &lightOffset, &explicitSam, &shadowRayPos, &shadowRayDir, &tfar);
    /* kernel_ShadowTrace :170-179 */
    float shadow = 0.0f;
    if (lightOffset >= 0) {
      if (probe && probe->shadow && probe->bounce == depth) { probe->pos = shadowRayPos; probe->dir = shadowRayDir; probe->tfar = tfar; probe->have = 1; }
      shadow = shadowTrace(s, shadowRayPos, shadowRayDir, tfar);
      st->rays++;
    }
    const f3 explicitColor = stage_shade(s, mat, &surf, ray_dir, shadowRayDir, &explicitSam, lightPickProb, lightOffset, shadow);
    /* RndMatAll crandom.h:478-494 */
    float allRands[FLOATS_PER_SAMPLE + FLOATS_PER_MLAYER];
    {
      float r4[4];
      orc_rnd_float4(gen, r4);
      allRands[0] = r4[0]; allRands[1] = r4[1]; allRands[2] = r4[2];
      for (int k = 0; k < FLOATS_PER_MLAYER; k++) allRands[FLOATS_PER_SAMPLE + k] = orc_rnd_float1(gen);
    }
    stage_next(s, mat, &surf, allRands, explicitColor, &ray_pos, &ray_dir, &flags, &misPrev, &accumColor, &thoroughput, shadow);
  }
  g_ptl = NULL;
  accumColor = add3(accumColor, mul3(thoroughput, currColor));   /* kernel_AddLastBouceContrib */
  return accumColor;
}

/* One bounce of n paths with every input handed in: the stage functions above in PathTrace's order.  Layouts as hydra_hip_stage_bounce
 * (include/hydra_hip.h): surf24 as written by orc_eval_surface; in16 = throughput xyz, previous BSDF pdf, radiance xyz, previous bounce specular,
 * rndLight's four numbers, the number that picks the light, shadow visibility, Lite_Hit.instId (int), ray flags (int); out40 there. */
void orc_stage_bounce(const OrcScene* s, int n, int depth, int maxDepth, const float* pos4, const float* dir4, const float* surf24, const float* in16,
                      const float* rands10, float* out40) {
  for (int i = 0; i < n; i++) {
    const float* r = surf24 + 24 * (size_t)i;
    const float* in = in16 + 16 * (size_t)i;
    float* o = out40 + 40 * (size_t)i;
    memset(o, 0, 40 * sizeof(float));
    OrcPtl ptlTmp;
    g_ptl = stage_ptl(n, i, &ptlTmp);
    f3 ray_pos = v3(pos4[4 * i], pos4[4 * i + 1], pos4[4 * i + 2]), ray_dir = v3(dir4[4 * i], dir4[4 * i + 1], dir4[4 * i + 2]);
    f3 thoroughput = v3(in[0], in[1], in[2]), accumColor = v3(in[4], in[5], in[6]);
    MisData misPrev = {in[3], in[7] != 0.0f};
    uint32_t flags = (uint32_t)as_int(in[15]);
    SurfaceHit surf;
    memset(&surf, 0, sizeof(surf));
    surf.pos = v3(r[0], r[1], r[2]); surf.normal = v3(r[3], r[4], r[5]); surf.flatNormal = v3(r[6], r[7], r[8]);
    surf.tangent = v3(r[9], r[10], r[11]); surf.biTangent = v3(r[12], r[13], r[14]);
    surf.texCoord.x = r[15]; surf.texCoord.y = r[16];
    surf.matId = as_int(r[17]); surf.t = r[18]; surf.sRayOff = r[19]; surf.hfi = (r[20] != 0.0f);
    f3 currColor = v3(0, 0, 0);
    if (surf.matId < 0) {   /* in[14] of a ray that left the scene: its pixel, x | y << 16 (the reference's in_packXY), read by the back-plate */
      currColor = haveBackPlate(s) ? environmentColorExtended(s, ray_pos, ray_dir, misPrev.matSamplePdf, misPrev.isSpecular, flags, as_int(in[14]) & 0xFFFF, (as_int(in[14]) >> 16) & 0xFFFF)
                                   : environmentColor(s, ray_dir, misPrev.matSamplePdf, misPrev.isSpecular, flags);
      const f3 fin = add3(accumColor, mul3(thoroughput, currColor));
      o[0] = currColor.x; o[1] = currColor.y; o[2] = currColor.z; o[3] = as_float(1); o[34] = fin.x; o[35] = fin.y; o[36] = fin.z;
      continue;
    }
    const float* mat = materialAt(s, surf.matId);
    if (stage_emission(s, ray_pos, ray_dir, &surf, mat, as_int(in[14]), flags, misPrev, &currColor)) {
      const f3 fin = add3(accumColor, mul3(thoroughput, currColor));
      o[0] = currColor.x; o[1] = currColor.y; o[2] = currColor.z; o[3] = as_float(2); o[34] = fin.x; o[35] = fin.y; o[36] = fin.z;
      continue;
    }
    if (depth >= maxDepth - 1) { o[3] = as_float(4); o[34] = accumColor.x; o[35] = accumColor.y; o[36] = accumColor.z; continue; }
    float lightPickProb, tfar;
    int lightOffset;
    f3 shadowRayPos, shadowRayDir;
    ShadowSample sam;
    stage_light(s, &surf, in + 8, in[12], &lightPickProb, &lightOffset, &sam, &shadowRayPos, &shadowRayDir, &tfar);
    o[4] = sam.pos.x; o[5] = sam.pos.y; o[6] = sam.pos.z; o[7] = sam.pdf; o[8] = sam.color.x; o[9] = sam.color.y; o[10] = sam.color.z;
    o[11] = sam.isPoint ? 1.0f : 0.0f; o[12] = lightPickProb; o[13] = as_float(lightOffset);
    o[14] = shadowRayPos.x; o[15] = shadowRayPos.y; o[16] = shadowRayPos.z; o[17] = tfar; o[18] = shadowRayDir.x; o[19] = shadowRayDir.y; o[20] = shadowRayDir.z;
    const f3 explicitColor = stage_shade(s, mat, &surf, ray_dir, shadowRayDir, &sam, lightPickProb, lightOffset, in[13]);
    o[21] = explicitColor.x; o[22] = explicitColor.y; o[23] = explicitColor.z;
    stage_next(s, mat, &surf, rands10 + 10 * (size_t)i, explicitColor, &ray_pos, &ray_dir, &flags, &misPrev, &accumColor, &thoroughput, in[13]);
    o[24] = ray_pos.x; o[25] = ray_pos.y; o[26] = ray_pos.z; o[27] = ray_dir.x; o[28] = ray_dir.y; o[29] = ray_dir.z; o[30] = as_float((int)flags);
    o[31] = thoroughput.x; o[32] = thoroughput.y; o[33] = thoroughput.z; o[34] = accumColor.x; o[35] = accumColor.y; o[36] = accumColor.z;
    o[37] = misPrev.matSamplePdf; o[38] = misPrev.isSpecular ? 1.0f : 0.0f;
  }
  g_ptl = NULL;
}

/* One shading point: kernel_LightSelect + kernel_LightSample + materialEval towards the sample + the BxDF sampling of
   kernel_NextBounce, with the random numbers handed in (fixtures 5 and 6 of SURVEY.md 8c).  surf24 as written by
   orc_eval_surface; out28: [0..2] sample pos, [3] sample pdf, [4..6] sample colour, [7] pick prob, [8] light offset (int;
   -2 = no surface), [9] isPoint, [10..12] brdf, [13] pdfFwd, [14..16] btdf, [17..19] MatSample.color, [20] pdf,
   [21..23] direction, [24] MatSample.flags (int), [25] flagsNextBounceLite (int). */
void orc_shade_point(const OrcScene* s, int n, const float* surf24, const float* dir4, const int32_t* flagsIn, const float* rndLight4,
                     const float* rands10, float* out28) {
  for (int i = 0; i < n; i++) {
    const float* r = surf24 + 24 * (size_t)i;
    float* o = out28 + 28 * (size_t)i;
    memset(o, 0, 28 * sizeof(float));
    SurfaceHit surf;
    memset(&surf, 0, sizeof(surf));
    surf.pos = v3(r[0], r[1], r[2]); surf.normal = v3(r[3], r[4], r[5]); surf.flatNormal = v3(r[6], r[7], r[8]);
    surf.tangent = v3(r[9], r[10], r[11]); surf.biTangent = v3(r[12], r[13], r[14]);
    surf.texCoord.x = r[15]; surf.texCoord.y = r[16];
    surf.matId = as_int(r[17]); surf.t = r[18]; surf.sRayOff = r[19]; surf.hfi = (r[20] != 0.0f);
    if (surf.matId < 0) { o[8] = as_float(-2); continue; }
    const f3 ray_dir = v3(dir4[4 * i], dir4[4 * i + 1], dir4[4 * i + 2]);
    const uint32_t flags = (uint32_t)flagsIn[i];
    const float* mat = materialAt(s, surf.matId);
    const float* rl = rndLight4 + 4 * (size_t)i;
    float lightPickProb = 1.0f;
    const int lightOffset = SelectRandomLightRev(rl[2], s, &lightPickProb);
    o[7] = lightPickProb; o[8] = as_float(lightOffset);
    if (lightOffset >= 0) {
      ShadowSample sam;
      memset(&sam, 0, sizeof(sam));
      const float* pl = lightAt(s, lightOffset);
      LightSampleRev(s, pl, v3(rl[0], rl[1], rl[2]), surf.pos, &sam);
      const f3 shadowRayDir = normalize3(sub3(sam.pos, surf.pos));
      o[0] = sam.pos.x; o[1] = sam.pos.y; o[2] = sam.pos.z; o[3] = sam.pdf;
      o[4] = sam.color.x; o[5] = sam.color.y; o[6] = sam.color.z; o[9] = sam.isPoint ? 1.0f : 0.0f;
      ShadeContext sc;
      sc.l = shadowRayDir; sc.v = scale3(ray_dir, -1.0f); sc.n = surf.normal; sc.fn = surf.flatNormal;
      sc.tg = surf.tangent; sc.bn = surf.biTangent; sc.tc = surf.texCoord;
      const BxDFResult ev = materialEval(mat, &sc, s);
      o[10] = ev.brdf.x; o[11] = ev.brdf.y; o[12] = ev.brdf.z; o[13] = ev.pdfFwd;
      o[14] = ev.btdf.x; o[15] = ev.btdf.y; o[16] = ev.btdf.z;
    }
    MatSample ms;
    MaterialSampleAndEvalBxDF(mat, rands10 + 10 * (size_t)i, &surf, ray_dir, flags, s, &ms);
    o[17] = ms.color.x; o[18] = ms.color.y; o[19] = ms.color.z; o[20] = ms.pdf;
    o[21] = ms.direction.x; o[22] = ms.direction.y; o[23] = ms.direction.z;
    o[24] = as_float(ms.flags); o[25] = as_float((int)flagsNextBounceLite(flags, &ms, s));
  }
}

void orc_path_trace(const OrcScene* s, int n, const float* pos4, const float* dir4, uint32_t* rng2, float* color4) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int i = 0; i < n; i++) {
    PathStat st = {0};
    { const int wdt = (int)g_varsF(s)[HRT_WIDTH_F]; g_screenX = wdt > 0 ? i % wdt : 0; g_screenY = wdt > 0 ? i / wdt : 0; }   /* path i plays pixel i (back-plate scenes) */
    const f3 c = PathTrace(s, v3(pos4[4 * i], pos4[4 * i + 1], pos4[4 * i + 2]), v3(dir4[4 * i], dir4[4 * i + 1], dir4[4 * i + 2]), rng2 + 2 * (size_t)i, &st, NULL);
    color4[4 * i] = c.x; color4[4 * i + 1] = c.y; color4[4 * i + 2] = c.z; color4[4 * i + 3] = (float)st.rays;
  }
}

/* ------------------------------------------------------------------------------------------------ f3: IntegratorMMLT::F */
/* The contribution function of multiplexed MLT over the simplified bidirectional sampler, restated from
 * CPUExp_Integrators_MMLT.cpp:146-315 (F), :637-754 (LightPath, TraceLightPath), :756-929 (CameraPath), :931-1047 (ConnectEye,
 * ConnectShadow, ConnectEndPoints) and cbidir.h:190-477 (ConnectEyeP, ConnectShadowP, ConnectEndPointsP).  Every random number
 * comes from the primary-sample vector (gen.rptr != 0 in crandom.h:340-520), so F is a pure function of (xVec, d).  The two
 * recursions are unrolled into loops; CameraPath's colour is multiplied on the way back up exactly as the recursion does.
 * Not restated: m_mask, the debug ray recorder, and the reuse of the previous sub-path for a one-sided mutation (:160-161, :178,
 * :209: both vertices are recomputed whenever they are used, so the result does not depend on it). */
enum { MMLT_HEAD_TOTAL_SIZE = 12, MMLT_FLOATS_PER_BOUNCE = 10, MMLT_DIM_LGT_X = 4, MMLT_DIM_LGT_N = 10, MMLT_DIM_SPLIT = 11, MMLT_MAX_DEPTH = 16,
       G_LSEL_FWD_OFFS = 232, G_LSEL_FWD_SIZE = 233, HRT_MMLT_FIRST_BOUNCE = 34, MF_HAVE_BTDF = 8192 };   /* cglobals.h:2644 */
typedef struct { float pdfFwd, pdfRev; } PdfVertex;
typedef struct { SurfaceHit hit; f3 ray_dir, accColor; float lastGTerm; int valid, wasSpecOnly; } PathVertex;
typedef struct { float matSamplePdf, cosThetaPrev; int isSpecular; } MisDataB;
static inline f3 div3s(f3 a, float b) { return v3(a.x / b, a.y / b, a.z / b); }   /* vector / scalar as the OpenCL branch (the pinned build) computes it */
static inline int mapRndFloatToInt(float a_val, int a, int b) {   /* crandom.h:507-518 */
  const float fa = (float)(a + 0), fb = (float)(b + 1);
  const int res = (int)(fa + a_val * (fb - fa));
  return (res > b) ? b : res;
}
static inline int isPureSpecularSam(const MatSample* ms) { return (ms->flags & RAY_EVENT_S) != 0 || (ms->flags & RAY_EVENT_T) != 0; }   /* cglobals.h:1342 */
static inline int flagsHaveOnlySpecular(uint32_t flags) {   /* cmaterial.h:3295-3301 */
  const uint32_t other = (flags & 0xFFFF0000u) >> 16;
  return ((other & RAY_EVENT_G) == 0) && ((other & RAY_EVENT_D) == 0);
}
/* ref: cfetch.h:933-968 MakeEyeRayFromF4Rnd */
static void MakeEyeRayFromF4Rnd(const float lensOffs[4], const OrcScene* s, f3* pRayPos, f3* pRayDir, float* pX, float* pY) {
  const float fwidth = g_varsF(s)[HRT_WIDTH_F], fheight = g_varsF(s)[HRT_HEIGHT_F];
  const float x = fwidth * lensOffs[0], y = fheight * lensOffs[1];
  const m44 projInv = load_m44((const float*)(s->globals + G_MPROJ_INV));
  const m44 wvInv = load_m44((const float*)(s->globals + G_MWORLDVIEW_INV));
  f3 ray_pos = v3(0.0f, 0.0f, 0.0f);
  f3 ray_dir = EyeRayDirNormalized(x / fwidth, y / fheight, projInv);
  ray_dir = tiltCorrection(ray_pos, ray_dir, s);
  if (g_varsI(s)[HRT_ENABLE_DOF] == 1) {
    const float tFocus = g_varsF(s)[HRT_DOF_FOCAL_PLANE_DIST] / (-ray_dir.z);
    const f3 focusPosition = add3(ray_pos, scale3(ray_dir, tFocus));
    f2 in = {lensOffs[2] - 0.5f, lensOffs[3] - 0.5f};
    const f2 d = MapSamplesToDisc(in);
    const float k = g_varsF(s)[HRT_DOF_LENS_RADIUS] * 2.0f;
    ray_pos.x += k * d.x;
    ray_pos.y += k * d.y;
    ray_dir = normalize3(sub3(focusPosition, ray_pos));
  }
  {
    const f3 pos = mul4x3(wvInv, ray_pos);
    const f3 pos2 = mul4x3(wvInv, add3(ray_pos, scale3(ray_dir, 100.0f)));
    *pRayPos = pos;
    *pRayDir = normalize3(sub3(pos2, pos));
  }
  *pX = lensOffs[0] * fwidth;
  *pY = lensOffs[1] * fheight;
}
/* ref: clight.h:1808-1822 SelectRandomLightFwd */
static int SelectRandomLightFwd(float r, const OrcScene* s, float* pickProb) {
  const int tableSize = s->globals[G_LSEL_FWD_SIZE];
  *pickProb = 1.0f;
  if (tableSize <= 2) return 0;
  return SelectIndexPropToOpt(r, (const float*)(s->globals + s->globals[G_LSEL_FWD_OFFS]), tableSize, pickProb);
}
static void LightSampleForwardAny(const OrcScene* s, const float* L, const float r[4], float r2x, LightSampleFwd* sam) {
  switch (as_int(L[PL_TYPE])) {
    case LT_MESH: MeshLightSampleForward(s, L, r, r2x, sam); break;
    case LT_SPHERE: SphereLightSampleForward(L, r, sam); break;
    case LT_DIRECT: DirectLightSampleForward(L, r, sam); break;
    case LT_POINT_SPOT: PointSpotSampleForward(L, r, sam); break;
    case LT_POINT_OMNI: PointLightSampleForward(s, L, r, sam); break;
    case LT_CYLINDER: CylinderLightSampleForward(s, L, r, r2x, sam); break;
    default: AreaLightSampleForward(s, L, r, r2x, sam); break;
  }
}
static void lightPdfFwdOne(const OrcScene* s, const float* L, f3 ray_dir, float ct, float* pdfA, float* pdfW) {
  float out4[4];
  const int ltype = as_int(L[PL_TYPE]);
  out4[0] = 1.0f / L[PL_SURFACE_AREA]; out4[1] = fmaxf(ct * INV_PI, 0.0f);
  if (ltype == LT_POINT_OMNI) out4[1] = INV_PI * 0.25f;
  else if (ltype == LT_POINT_SPOT) { const float cos2 = L[POINT_LIGHT_SPOT_COS2]; out4[1] = 1.0f / (2.0f * ORC_PI * (1.0f - cos2)); if (ct < cos2) out4[1] = 0.0f; }
  else if (ltype == LT_DIRECT) { const float r2 = L[DIRECT_LIGHT_RADIUS2]; out4[0] = 1.0f / (ORC_PI * r2 * r2); out4[1] = 0.0f; }
  if (as_int(L[PL_FLAGS]) & LF_HAS_IES) out4[1] = lightPdfFwdIES(s, L, ray_dir);
  else if (ltype == LT_AREA && as_int(L[AL_SPOT_DISTR]) != 0) { const float cos2 = L[AL_SPOT_COS2]; out4[1] = 1.0f / (2.0f * ORC_PI * (1.0f - cos2)); if (ct < cos2) out4[1] = 0.0f; }
  *pdfA = out4[0]; *pdfW = out4[1];
}
static ShadeContext makeShadeContext(const SurfaceHit* h, f3 l, f3 v) {
  ShadeContext sc;
  memset(&sc, 0, sizeof(sc));
  sc.l = l; sc.v = v; sc.n = h->normal; sc.fn = h->flatNormal; sc.tg = h->tangent; sc.bn = h->biTangent; sc.tc = h->texCoord;
  return sc;
}
static void InitPathVertex(PathVertex* v) { memset(v, 0, sizeof(*v)); v->lastGTerm = 1.0f; v->accColor = v3(1, 1, 1); }
/* camera connection factor for one point (orc_camera_connect above holds the same arithmetic for arrays) */
static float cameraImageToSurfaceFactor1(const OrcScene* s, f3 hitPos, f3 hitNorm, f3* camDir, float* zDepth, float* scrX, float* scrY) {
  float p4[4] = {hitPos.x, hitPos.y, hitPos.z, 0}, n4[4] = {hitNorm.x, hitNorm.y, hitNorm.z, 0}, d2[2] = {0, 0}, o[8];
  orc_camera_connect(s, 1, p4, n4, d2, o);
  *camDir = v3(o[1], o[2], o[3]); *zDepth = o[4]; *scrX = o[5]; *scrY = o[6];
  return o[0];
}
/* out8 = colour xyz (MIS-weighted), x, y, split s, MIS weight, contribFunc(colour) */
static void mmltF(const OrcScene* s, const float* xVec, int d, float* out8) {
  PdfVertex pdfArray[MMLT_MAX_DEPTH + 2];
  memset(pdfArray, 0, sizeof(pdfArray));
  const int width = (int)g_varsF(s)[HRT_WIDTH_F], height = (int)g_varsF(s)[HRT_HEIGHT_F];
  const float mLightSubPathCount = (float)(width * height);
  const int splitDLByGrammar = g_varsI(s)[HRT_MMLT_FIRST_BOUNCE] > 3;   /* Common.cpp:28 */
  const int sp = mapRndFloatToInt(xVec[MMLT_DIM_SPLIT], 0, d), t = d - sp;
  const int lightTraceDepth = sp - 1, camTraceDepth = t;
  const float* lensOffs = xVec;
  int x = (int)(lensOffs[0] * (float)width + 0.5f), y = (int)(lensOffs[1] * (float)height + 0.5f);

  /* (1) camera sub-path, CameraPath :756-929 */
  PathVertex cv;
  InitPathVertex(&cv);
  if (camTraceDepth > 0) {
    const float* rptr = xVec + MMLT_HEAD_TOTAL_SIZE + MMLT_FLOATS_PER_BOUNCE * sp;
    float fx, fy;
    f3 ray_pos, ray_dir;
    MakeEyeRayFromF4Rnd(lensOffs, s, &ray_pos, &ray_dir, &fx, &fy);
    x = (int)(fx + 0.5f); y = (int)(fy + 0.5f);
    if (x >= width) x = width - 1;
    if (y >= height) y = height - 1;
    const int haveToHitLight = (lightTraceDepth == -1);
    MisDataB misPrev = {1.0f, 1.0f, 1};
    uint32_t flags = 0;
    f3 factors[MMLT_MAX_DEPTH + 2];
    int nFactors = 0, zeroFrom = -1;
    cv.valid = 0; cv.accColor = v3(0, 0, 0);
    for (int currDepth = 1; currDepth <= camTraceDepth; currDepth++) {
      const int prevVertexId = d - currDepth + 1;
      const OrcHit hit = rayTrace(s, ray_pos, ray_dir, NULL);
      if (!HitSome(hit)) break;
      const SurfaceHit surf = evalSurface(s, ray_pos, ray_dir, hit);
      const float cosHere = fabsf(dot3(ray_dir, surf.normal)), cosPrev = fabsf(misPrev.cosThetaPrev);
      float GTerm = 1.0f;
      if (currDepth == 1) {
        f3 cd; float zd, sx, sy;
        const float imageToSurfaceFactor = cameraImageToSurfaceFactor1(s, surf.pos, surf.normal, &cd, &zd, &sx, &sy);
        pdfArray[d].pdfRev = imageToSurfaceFactor / mLightSubPathCount;
        pdfArray[d].pdfFwd = 1.0f;
      } else {
        const float dist = length3(sub3(ray_pos, surf.pos));
        GTerm = cosHere * cosPrev / fmaxf(dist * dist, DEPSILON2);
      }
      const float* mat = materialAt(s, surf.matId);
      const float* pLight = lightAt(s, s->instLightInstId[hit.instId]);
      const f3 emission = emissionEval(s, ray_pos, ray_dir, &surf, flags, misPrev.isSpecular == 1, pLight, mat);
      if (dot3(emission, emission) > 1e-6f) {
        if (currDepth == camTraceDepth && haveToHitLight) {
          float pdfA, pdfW;
          lightPdfFwdOne(s, pLight, ray_dir, cosHere, &pdfA, &pdfW);
          const float pdfLightWP = pdfW / fmaxf(cosHere, DEPSILON);
          const float pdfMatRevWP = misPrev.matSamplePdf / fmaxf(cosPrev, DEPSILON);
          pdfArray[0].pdfFwd = pdfA / (float)s->globals[G_LIGHTS_NUM];
          pdfArray[0].pdfRev = 1.0f;
          pdfArray[1].pdfFwd = pdfLightWP * GTerm;
          pdfArray[1].pdfRev = misPrev.isSpecular ? -1.0f * GTerm : pdfMatRevWP * GTerm;
          cv.hit = surf; cv.ray_dir = ray_dir; cv.accColor = emission; cv.valid = 1;
        }
        break;
      } else if (currDepth == camTraceDepth && !haveToHitLight) {
        cv.hit = surf; cv.ray_dir = ray_dir; cv.valid = 1; cv.accColor = v3(1, 1, 1);
        cv.wasSpecOnly = splitDLByGrammar ? flagsHaveOnlySpecular(flags) : 0;
        if (camTraceDepth != 1) {
          const float lastPdfWP = misPrev.matSamplePdf / fmaxf(cosPrev, DEPSILON);
          cv.lastGTerm = GTerm;
          pdfArray[prevVertexId].pdfRev = misPrev.isSpecular ? -1.0f * GTerm : GTerm * lastPdfWP;
        } else cv.lastGTerm = 1.0f;
        break;
      }
      /* (3) sample the next direction; random numbers of this bounce: rptr + rndMatOffsetMMLT(currDepth - 1) */
      MatSample matSam;
      MaterialSampleAndEvalBxDFEx(mat, rptr + MMLT_FLOATS_PER_BOUNCE * (currDepth - 1), &surf, ray_dir, (uint32_t)(currDepth - 1) << 8, 0, s, &matSam);
      const float cosNext = fabsf(dot3(matSam.direction, surf.normal));
      if (currDepth == 1) {
        if (isPureSpecularSam(&matSam)) pdfArray[d].pdfFwd = 0.0f;
      } else {
        if (!isPureSpecularSam(&matSam)) {
          const ShadeContext sc = makeShadeContext(&surf, scale3(ray_dir, -1.0f), matSam.direction);
          const float pdfFwdW = materialEval(mat, &sc, s).pdfFwd;
          pdfArray[prevVertexId].pdfFwd = (pdfFwdW / fmaxf(cosHere, DEPSILON)) * GTerm;
        } else pdfArray[prevVertexId].pdfFwd = -1.0f * GTerm;
        const float pdfCamPrevWP = misPrev.matSamplePdf / fmaxf(cosPrev, DEPSILON);
        pdfArray[prevVertexId].pdfRev = misPrev.isSpecular ? -1.0f * GTerm : pdfCamPrevWP * GTerm;
      }
      const int stopDL = splitDLByGrammar ? flagsHaveOnlySpecular(flags) : 0;
      factors[nFactors] = div3s(scale3(matSam.color, cosNext), fmaxf(matSam.pdf, DEPSILON2));
      if (stopDL && haveToHitLight && currDepth + 1 == camTraceDepth) zeroFrom = nFactors;
      nFactors++;
      ray_pos = OffsRayPos(surf.pos, surf.normal, matSam.direction);
      ray_dir = matSam.direction;
      misPrev.isSpecular = isPureSpecularSam(&matSam);
      misPrev.matSamplePdf = matSam.pdf;
      misPrev.cosThetaPrev = dot3(ray_dir, surf.normal);
      flags = flagsNextBounceLite(flags, &matSam, s);
    }
    /* the recursion multiplies on the way back: deepest factor first; an invalid vertex carries colour 0 through */
    if (!cv.valid) cv.accColor = v3(0, 0, 0);
    for (int k = nFactors - 1; k >= 0; k--) {
      cv.accColor = mul3(cv.accColor, factors[k]);
      if (k == zeroFrom) cv.accColor = v3(0, 0, 0);
    }
  }

  /* (2) light sub-path, LightPath :637-669 + TraceLightPath :671-754 */
  PathVertex lv;
  InitPathVertex(&lv);
  if (lightTraceDepth > 0) {
    float lightPickProb = 1.0f;
    const int lightId = SelectRandomLightFwd(xVec[MMLT_DIM_LGT_N], s, &lightPickProb);
    const float* pLight = lightAt(s, lightId);
    LightSampleFwd sample;
    LightSampleForwardAny(s, pLight, xVec + MMLT_DIM_LGT_X, xVec[MMLT_DIM_LGT_X + 4], &sample);   /* group1 = dims 4..7, group2.x = dim 8 (RndLightMMLT) */
    pdfArray[0].pdfFwd = sample.pdfA * lightPickProb;
    pdfArray[0].pdfRev = 1.0f;
    f3 color = div3s(scale3(sample.color, 1.0f / lightPickProb), sample.pdfA * sample.pdfW);
    const float* rptr = xVec + MMLT_HEAD_TOTAL_SIZE;
    f3 ray_pos = sample.pos, ray_dir = sample.dir;
    float prevLightCos = sample.cosTheta, prevPdf = sample.pdfW;
    int wasSpecular = 0;
    for (int currDepth = 1; currDepth <= lightTraceDepth; currDepth++) {
      const OrcHit hit = rayTrace(s, ray_pos, ray_dir, NULL);
      if (!HitSome(hit)) break;
      const SurfaceHit surf = evalSurface(s, ray_pos, ray_dir, hit);
      const float cosCurr = fabsf(-dot3(ray_dir, surf.normal));
      const float dist = length3(sub3(surf.pos, ray_pos));
      const float GTermPrev = (prevLightCos * cosCurr / fmaxf(dist * dist, DEPSILON2));
      const float prevPdfWP = prevPdf / fmaxf(prevLightCos, DEPSILON);
      pdfArray[currDepth].pdfFwd = !wasSpecular ? prevPdfWP * GTermPrev : -1.0f * GTermPrev;
      const float* mat = materialAt(s, surf.matId);
      MatSample matSam;
      MaterialSampleAndEvalBxDFEx(mat, rptr + MMLT_FLOATS_PER_BOUNCE * (currDepth - 1), &surf, ray_dir, (uint32_t)(currDepth - 1) << 8, 1, s, &matSam);
      const f3 nextRay_dir = matSam.direction;
      const f3 nextRay_pos = OffsRayPos(surf.pos, surf.normal, matSam.direction);
      const float cosNext = fabsf(+dot3(nextRay_dir, surf.normal));
      if (currDepth == lightTraceDepth) {
        lv.hit = surf; lv.ray_dir = ray_dir; lv.accColor = color; lv.valid = 1; lv.lastGTerm = GTermPrev;
        break;
      }
      if (!isPureSpecularSam(&matSam)) {
        const ShadeContext sc = makeShadeContext(&surf, scale3(ray_dir, -1.0f), scale3(nextRay_dir, -1.0f));
        const float pdfW = materialEval(mat, &sc, s).pdfFwd;
        pdfArray[currDepth].pdfRev = (pdfW / fmaxf(cosCurr, DEPSILON)) * GTermPrev;
      } else pdfArray[currDepth].pdfRev = -1.0f * GTermPrev;
      color = mul3(color, scale3(scale3(matSam.color, cosNext), (1.0f / fmaxf(matSam.pdf, DEPSILON2))));
      ray_pos = nextRay_pos; ray_dir = nextRay_dir;
      prevLightCos = cosNext; prevPdf = matSam.pdf; wasSpecular = isPureSpecularSam(&matSam);
    }
  }

  /* (3) connect */
  f3 sampleColor = v3(0, 0, 0);
  if (lightTraceDepth == -1) sampleColor = cv.accColor;
  else if (camTraceDepth == 0) {   /* ConnectEye :931-958 + ConnectEyeP cbidir.h:190-253 */
    if (lv.valid) {
      f3 camDir; float zDepth, scrX, scrY;
      const float imageToSurfaceFactor = cameraImageToSurfaceFactor1(s, lv.hit.pos, lv.hit.normal, &camDir, &zDepth, &scrX, &scrY);
      const float* mat = materialAt(s, lv.hit.matId);
      float signOfNormal = 1.0f;
      if ((matFlags(mat) & MF_HAVE_BTDF) != 0 && dot3(camDir, lv.hit.normal) < -0.01f) signOfNormal = -1.0f;
      const OrcHit hit = rayTrace(s, add3(lv.hit.pos, scale3(lv.hit.normal, epsilonOfPos(lv.hit.pos) * signOfNormal)), camDir, NULL);
      if (imageToSurfaceFactor <= 0.0f || (HitSome(hit) && hit.t <= zDepth)) { x = -1; y = -1; }
      else {
        const float surfaceToImageFactor = 1.f / imageToSurfaceFactor;
        const ShadeContext sc = makeShadeContext(&lv.hit, camDir, scale3(lv.ray_dir, -1.0f));
        const BxDFResult colorAndPdf = materialEvalEx(mat, &sc, 1, s);
        const f3 colorConnect = add3(colorAndPdf.brdf, colorAndPdf.btdf);
        const float pdfRevW = colorAndPdf.pdfRev;
        const float cosCurr = fabsf(dot3(lv.ray_dir, lv.hit.normal));
        const float pdfRevWP = pdfRevW / fmaxf(cosCurr, DEPSILON2);
        const float cameraPdfA = imageToSurfaceFactor / mLightSubPathCount;
        pdfArray[lightTraceDepth].pdfRev = (pdfRevW == 0.0f) ? -1.0f * lv.lastGTerm : pdfRevWP * lv.lastGTerm;
        pdfArray[lightTraceDepth + 1].pdfFwd = 1.0f;
        pdfArray[lightTraceDepth + 1].pdfRev = cameraPdfA;
        const f3 sc3 = mul3(lv.accColor, div3s(colorConnect, mLightSubPathCount * surfaceToImageFactor));
        if (dot3(sc3, sc3) > 1e-12f) { x = (int)scrX; y = (int)scrY; sampleColor = sc3; }
      }
    }
  } else if (lightTraceDepth == 0) {   /* ConnectShadow :962-1009 + ConnectShadowP cbidir.h:285-345 */
    if (cv.valid && !cv.wasSpecOnly) {
      f3 explicitColor = v3(0, 0, 0);
      float lightPickProb = 1.0f;
      const int lightOffset = SelectRandomLightRev(xVec[MMLT_DIM_LGT_N], s, &lightPickProb);
      if (lightOffset >= 0) {
        const float* pLight = lightAt(s, lightOffset);
        ShadowSample explicitSam;
        LightSampleRev(s, pLight, v3(xVec[MMLT_DIM_LGT_X], xVec[MMLT_DIM_LGT_X + 1], xVec[MMLT_DIM_LGT_X + 2]), cv.hit.pos, &explicitSam);
        const f3 shadowRayDir = normalize3(sub3(explicitSam.pos, cv.hit.pos));
        const f3 shadowRayPos = OffsRayPos(cv.hit.pos, cv.hit.normal, shadowRayDir);
        const float shadow = shadowTrace(s, shadowRayPos, shadowRayDir, explicitSam.maxDist * 0.9995f);
        if (shadow * shadow * 3.0f > 1e-12f) {
          const float* mat = materialAt(s, cv.hit.matId);
          const ShadeContext sc = makeShadeContext(&cv.hit, shadowRayDir, scale3(cv.ray_dir, -1.0f));
          const BxDFResult evalData = materialEval(mat, &sc, s);
          const float pdfFwdAt1W = evalData.pdfRev;
          const float cosThetaOut1 = fmaxf(+dot3(shadowRayDir, cv.hit.normal), DEPSILON), cosThetaOut2 = fmaxf(-dot3(shadowRayDir, cv.hit.normal), DEPSILON);
          const int inverseCos = ((matFlags(mat) & MF_HAVE_BTDF) != 0 && dot3(shadowRayDir, cv.hit.normal) < -0.01f);
          const float cosThetaOut = inverseCos ? cosThetaOut2 : cosThetaOut1;
          const float cosAtLight = fmaxf(explicitSam.cosAtLight, DEPSILON);
          const float cosThetaPrev = fmaxf(-dot3(cv.ray_dir, cv.hit.normal), DEPSILON);
          const f3 brdfVal = add3(scale3(evalData.brdf, cosThetaOut1), scale3(evalData.btdf, cosThetaOut2));
          const float pdfRevWP = evalData.pdfFwd / fmaxf(cosThetaOut, DEPSILON);
          const float shadowDist = length3(sub3(cv.hit.pos, explicitSam.pos));
          const float GTerm = cosThetaOut * cosAtLight / fmaxf(shadowDist * shadowDist, DEPSILON2);
          float pdfA, pdfW;
          lightPdfFwdOne(s, pLight, shadowRayDir, cosAtLight, &pdfA, &pdfW);
          pdfArray[0].pdfFwd = pdfA * lightPickProb;
          pdfArray[0].pdfRev = 1.0f;
          pdfArray[1].pdfFwd = (pdfW / cosAtLight) * GTerm;
          pdfArray[1].pdfRev = (evalData.pdfFwd == 0) ? -1.0f * GTerm : pdfRevWP * GTerm;
          if (t > 1) pdfArray[2].pdfFwd = (pdfFwdAt1W == 0.0f) ? -1.0f * cv.lastGTerm : (pdfFwdAt1W / cosThetaPrev) * cv.lastGTerm;
          float envMisMult = 1.0f;
          if (as_int(pLight[PL_TYPE]) == LT_SKY_DOME) envMisMult = misWeightHeuristic(explicitSam.pdf * lightPickProb, evalData.pdfFwd);
          const float explicitPdfW = fmaxf(explicitSam.pdf, DEPSILON2);
          explicitColor = scale3(div3s(mul3(scale3(scale3(explicitSam.color, envMisMult), 1.0f / lightPickProb), brdfVal), explicitPdfW), shadow);
        }
      }
      sampleColor = mul3(cv.accColor, explicitColor);
    }
  } else {   /* ConnectEndPoints :1011-1047 + ConnectEndPointsP cbidir.h:366-477 */
    if (cv.valid) {
      f3 explicitColor = v3(0, 0, 0);
      if (lv.valid) {
        const f3 diff = sub3(cv.hit.pos, lv.hit.pos);
        const float dist2 = fmaxf(dot3(diff, diff), DEPSILON2);
        const float dist = sqrtf(dist2);
        const f3 lToC = div3s(diff, dist);
        const float GTerm0 = (+dot3(lv.hit.normal, lToC)) * (-dot3(cv.hit.normal, lToC)) / dist2;
        if (!(GTerm0 < 0.0f)) {
          const f3 shadowRayPos = OffsRayPos(lv.hit.pos, lv.hit.normal, lToC);
          const float shadow = shadowTrace(s, shadowRayPos, lToC, dist * 0.9995f);
          if (!(shadow * shadow * 3.0f < 1e-12f)) {
            PdfVertex* vSplitBefore = &pdfArray[sp - 1]; PdfVertex* vSplit = &pdfArray[sp]; PdfVertex* vSplitAfter = &pdfArray[sp + 1];
            const float* matL = materialAt(s, lv.hit.matId);
            const ShadeContext scL = makeShadeContext(&lv.hit, lToC, scale3(lv.ray_dir, -1.0f));
            const BxDFResult evL = materialEvalEx(matL, &scL, 1, s);
            const f3 lightBRDF = add3(evL.brdf, evL.btdf);
            const float lightVPdfFwdW = evL.pdfFwd, lightVPdfRevW = evL.pdfRev;
            float signOfNormalL = 1.0f, signOfNormalC = 1.0f;
            if ((matFlags(matL) & MF_HAVE_BTDF) != 0 && dot3(lToC, lv.hit.normal) < -0.01f) signOfNormalL = -1.0f;
            const float* matC = materialAt(s, cv.hit.matId);
            const ShadeContext scC = makeShadeContext(&cv.hit, scale3(lToC, -1.0f), scale3(cv.ray_dir, -1.0f));
            const BxDFResult evC = materialEval(matC, &scC, s);
            const f3 camBRDF = add3(evC.brdf, evC.btdf);
            const float camVPdfRevW = evC.pdfFwd, camVPdfFwdW = evC.pdfRev;
            if ((matFlags(matC) & MF_HAVE_BTDF) != 0 && dot3(scale3(lToC, -1.0f), cv.hit.normal) < -0.01f) signOfNormalC = -1.0f;
            const float cosAtLightVertex = +signOfNormalL * dot3(lv.hit.normal, lToC), cosAtCameraVertex = -signOfNormalC * dot3(cv.hit.normal, lToC);
            const float cosAtLightVertexPrev = -dot3(lv.hit.normal, lv.ray_dir), cosAtCameraVertexPrev = -dot3(cv.hit.normal, cv.ray_dir);
            const float GTerm = cosAtLightVertex * cosAtCameraVertex / dist2;
            if (!(GTerm < 0.0f)) {
              const float lightPdfFwdWP = lightVPdfFwdW / fmaxf(cosAtLightVertex, DEPSILON2);
              const float cameraPdfRevWP = camVPdfRevW / fmaxf(cosAtCameraVertex, DEPSILON2);
              vSplit->pdfFwd = (lightPdfFwdWP == 0.0f) ? -1.0f * GTerm : lightPdfFwdWP * GTerm;
              vSplit->pdfRev = (cameraPdfRevWP == 0.0f) ? -1.0f * GTerm : cameraPdfRevWP * GTerm;
              vSplitBefore->pdfRev = (lightVPdfRevW == 0.0f) ? -1.0f * lv.lastGTerm : lv.lastGTerm * (lightVPdfRevW / fmaxf(cosAtLightVertexPrev, DEPSILON));
              if (d > 3) vSplitAfter->pdfFwd = (camVPdfFwdW == 0.0f) ? -1.0f * cv.lastGTerm : cv.lastGTerm * (camVPdfFwdW / fmaxf(cosAtCameraVertexPrev, DEPSILON));
              const int fwdCanNotBeEvaluated = (lightPdfFwdWP < DEPSILON2) || (d > 3 && camVPdfFwdW < DEPSILON2);
              const int revCanNotBeEvaluated = (cameraPdfRevWP < DEPSILON2) || (lightVPdfRevW < DEPSILON2);
              if (!(fwdCanNotBeEvaluated && revCanNotBeEvaluated)) explicitColor = scale3(scale3(mul3(lightBRDF, camBRDF), GTerm), shadow);
            }
          }
        }
      }
      sampleColor = mul3(mul3(cv.accColor, explicitColor), lv.accColor);
    }
  }

  /* (4) MIS weight :252-285 */
  float misWeight = 1.0f;
  if (dot3(sampleColor, sampleColor) > 1e-12f) {
    float pdfThisWay = 1.0f, pdfSumm = 0.0f;
    for (int split = 0; split <= d; split++) {
      const int s1 = split, t1 = d - split;
      const int specularMet = (split > 0) && (split < d) && (pdfArray[split].pdfRev < 0.0f || pdfArray[split].pdfFwd < 0.0f);
      float pdfOtherWay = specularMet ? 0.0f : 1.0f;
      if (split == d) pdfOtherWay = misHeuristicPower1(pdfArray[d].pdfFwd);
      for (int i = 0; i < s1; i++) pdfOtherWay *= misHeuristicPower1(pdfArray[i].pdfFwd);
      for (int i = s1 + 1; i <= d; i++) pdfOtherWay *= misHeuristicPower1(pdfArray[i].pdfRev);
      if (s1 == sp && t1 == t) pdfThisWay = pdfOtherWay;
      pdfSumm += pdfOtherWay;
    }
    misWeight = pdfThisWay / fmaxf(pdfSumm, DEPSILON2);
  }
  sampleColor = scale3(sampleColor, misWeight);
  if (!(x >= 0 && x < width && y >= 0 && y < height)) { x = 0; y = 0; sampleColor = v3(0, 0, 0); }
  out8[0] = sampleColor.x; out8[1] = sampleColor.y; out8[2] = sampleColor.z; out8[3] = (float)x; out8[4] = (float)y; out8[5] = (float)sp;
  out8[6] = misWeight; out8[7] = fmaxf(0.33334f * (sampleColor.x + sampleColor.y + sampleColor.z), 0.0f);   /* contribFunc, cglobals.h:1929-1932 */
}
/* xvec: n rows of `stride` floats (stride >= 12 + 10 * depth[i]), depth[i] in 1..MMLT_MAX_DEPTH */
void orc_mmlt_f(const OrcScene* s, int n, const int32_t* depth, const float* xvec, int stride, float* out8) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int i = 0; i < n; i++) mmltF(s, xvec + (size_t)i * stride, depth[i], out8 + 8 * (size_t)i);
}

/* The Markov chains: InitialSamplePS :51-57, MutatePrimarySpace :93-144 (MutateLightPart :59-73, MutateCameraPart :75-90), the accept /
 * contribute loop of DoPassIndirectMLT :379-447, for n independent chains (the HIP layer runs one chain per GPU thread; the reference one
 * per OpenMP thread).  gens4[i] = the two generator states of chain i (mutations, accept tests), in and out; xrows[i] = its current x vector
 * (stride floats), in and out; depth[i] = its d.  The clock()-driven stirring (:97-103, :362-368) is left out.  Outputs: image4 (w*h*4,
 * contributions added), chains6 = y, colour, pixel x, y per chain, accepted = count per chain. */
static float mutateKelemen1(float x, float r0, float r1, float p2, float p1) { float v = x, rr[2] = {r0, r1}, o; orc_mutate_kelemen(1, &v, rr, p2, p1, &o); return o; }
/* the acceptance probability and the two expected-value contributions of one mutation (CPUExp_Integrators_MMLT.cpp:400-428; MMLTAcceptReject, shaders/mlt.cl:205-262) */
static void mmltAcceptContrib(float yOld, const float* yOldColor, float yNew, const float* yNewColor, float bkScale, float* a_out, float* cX, float* cY) {
  const float a = (yOld == 0.0f) ? 1.0f : fminf(1.0f, yNew / yOld);
  const float kx = (1.0f / fmaxf(yOld, 1e-6f)), ky = (1.0f / fmaxf(yNew, 1e-6f));
  for (int q = 0; q < 3; q++) { cX[q] = yOldColor[q] * bkScale * kx * (1.0f - a); cY[q] = yNewColor[q] * bkScale * ky * a; }
  *a_out = a;
}
/* that step for n chains handed in (layout: include/hydra_hip.h, hydra_hip_stage_mmlt_accept) */
void orc_stage_mmlt_accept(int n, const float* old8, const float* new8, uint32_t* gen2, float bkScale, float* out12) {
  for (int i = 0; i < n; i++) {
    const float* o = old8 + 8 * (size_t)i; const float* w = new8 + 8 * (size_t)i;
    float* r = out12 + 12 * (size_t)i;
    float a, cX[3], cY[3];
    mmltAcceptContrib(o[7], o, w[7], w, bkScale, &a, cX, cY);
    const float p = orc_rnd_float1(gen2 + 2 * (size_t)i);
    memset(r, 0, 12 * sizeof(float));
    if (cX[0] * cX[0] + cX[1] * cX[1] + cX[2] * cX[2] > 1e-12f) { r[0] = cX[0]; r[1] = cX[1]; r[2] = cX[2]; r[3] = 1.0f - a; }
    if (cY[0] * cY[0] + cY[1] * cY[1] + cY[2] * cY[2] > 1e-12f) { r[4] = cY[0]; r[5] = cY[1]; r[6] = cY[2]; r[7] = a; }
    r[8] = (p <= a) ? 1.0f : 0.0f;
  }
}
void orc_mmlt_run(const OrcScene* s, int n, uint32_t* gens4, const int32_t* depth, int mutations, int w, float* image4, float* chains6, float* xrows, int stride, int32_t* accepted) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < n; i++) {
    uint32_t* gen = gens4 + 4 * (size_t)i;
    uint32_t* gen2 = gen + 2;
    const int d = depth[i], size = MMLT_HEAD_TOTAL_SIZE + MMLT_FLOATS_PER_BOUNCE * d;
    float xCur[MMLT_HEAD_TOTAL_SIZE + MMLT_FLOATS_PER_BOUNCE * MMLT_MAX_DEPTH], xNew[MMLT_HEAD_TOTAL_SIZE + MMLT_FLOATS_PER_BOUNCE * MMLT_MAX_DEPTH], o[8];
    memcpy(xCur, xrows + (size_t)i * stride, sizeof(float) * (size_t)size);
    mmltF(s, xCur, d, o);
    float y = o[7], yColor[3] = {o[0], o[1], o[2]};
    int xScr = (int)o[3], yScr = (int)o[4], acc = 0;
    for (int k = 0; k < mutations; k++) {
      const float plarge = 0.33f, plight = 0.20f, pmultiChain = 0.16f;
      const float selector = orc_rnd_float1(gen);
      if (selector < plarge) { for (int j = 0; j < size; j++) xNew[j] = orc_rnd_float1(gen); }
      else {
        memcpy(xNew, xCur, sizeof(float) * (size_t)size);
        const int currSplit = mapRndFloatToInt(xCur[MMLT_DIM_SPLIT], 0, d);
        const int camBegin = MMLT_HEAD_TOTAL_SIZE + MMLT_FLOATS_PER_BOUNCE * currSplit;
        const int lightPart = (plarge < selector && selector <= plarge + plight + pmultiChain);
        const int cameraPart = !(plarge < selector && selector <= plarge + plight);
        if (lightPart) {
          for (int j = 4; j < 10; j++) { const float r0 = orc_rnd_float1(gen), r1 = orc_rnd_float1(gen); xNew[j] = mutateKelemen1(xNew[j], r0, r1, 64.0f, 1024.0f); }
          for (int j = MMLT_HEAD_TOTAL_SIZE; j < camBegin; j++) { const float r0 = orc_rnd_float1(gen), r1 = orc_rnd_float1(gen); xNew[j] = mutateKelemen1(xNew[j], r0, r1, 64.0f, 1024.0f); }
        }
        if (cameraPart) {
          for (int j = 0; j < 4; j++) { const float r0 = orc_rnd_float1(gen), r1 = orc_rnd_float1(gen); xNew[j] = mutateKelemen1(xNew[j], r0, r1, j < 2 ? 128.0f : 64.0f, 1024.0f); }
          for (int j = camBegin; j < size; j++) { const float r0 = orc_rnd_float1(gen), r1 = orc_rnd_float1(gen); xNew[j] = mutateKelemen1(xNew[j], r0, r1, 64.0f, 1024.0f); }
        }
      }
      mmltF(s, xNew, d, o);
      const float yNew = o[7], yOld = y;
      const float yOldColor[3] = {yColor[0], yColor[1], yColor[2]};
      const int xScrOld = xScr, yScrOld = yScr, xScrNew = (int)o[3], yScrNew = (int)o[4];
      float a, cX[3], cY[3];
      mmltAcceptContrib(yOld, yOldColor, yNew, o, 1.0f, &a, cX, cY);
      const float p = orc_rnd_float1(gen2);
      if (p <= a) { memcpy(xCur, xNew, sizeof(float) * (size_t)size); y = yNew; yColor[0] = o[0]; yColor[1] = o[1]; yColor[2] = o[2]; xScr = xScrNew; yScr = yScrNew; acc++; }
      if (cX[0] * cX[0] + cX[1] * cX[1] + cX[2] * cX[2] > 1e-12f) {
        float* px = image4 + 4 * (size_t)(yScrOld * w + xScrOld);
        for (int q = 0; q < 3; q++) {
#pragma omp atomic
          px[q] += cX[q];
        }
#pragma omp atomic
        px[3] += (1.0f - a);
      }
      if (cY[0] * cY[0] + cY[1] * cY[1] + cY[2] * cY[2] > 1e-12f) {
        float* px = image4 + 4 * (size_t)(yScrNew * w + xScrNew);
        for (int q = 0; q < 3; q++) {
#pragma omp atomic
          px[q] += cY[q];
        }
#pragma omp atomic
        px[3] += a;
      }
    }
    if (chains6) { float* c6 = chains6 + 6 * (size_t)i; c6[0] = y; c6[1] = yColor[0]; c6[2] = yColor[1]; c6[3] = yColor[2]; c6[4] = (float)xScr; c6[5] = (float)yScr; }
    memcpy(xrows + (size_t)i * stride, xCur, sizeof(float) * (size_t)size);
    if (accepted) accepted[i] = acc;
  }
}

/* ref: CPUExp_Integrators_SBDPT.cpp:11-216 IntegratorSBDPT::DoPass expressed through F (its sub-path, connection and MIS code is IntegratorMMLT's):
 * sample i draws d = rndInt(2, maxDepth + 1) (:21, crandom.h:594-611), then a fresh primary-sample vector, from gens4[i][0..1]; the split is
 * x[MMLT_DIM_SPLIT] and the pixel comes from the lens dimensions (the reference draws both with rndInt, :22, :40-41); splat = F (d + 1)(maxDepth - 1) */
void orc_sbdpt_pass(const OrcScene* s, int n, uint32_t* gens4, int maxDepth, int w, float* image4) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int i = 0; i < n; i++) {
    uint32_t* gen = gens4 + 4 * (size_t)i;
    const float r = orc_rnd_float1(gen);
    int d = (int)(2.0f + r * ((float)(maxDepth + 1) - 2.0f));
    if (d > maxDepth) d = maxDepth;
    float x[MMLT_HEAD_TOTAL_SIZE + MMLT_FLOATS_PER_BOUNCE * MMLT_MAX_DEPTH], o[8];
    for (int j = 0; j < MMLT_HEAD_TOTAL_SIZE + MMLT_FLOATS_PER_BOUNCE * d; j++) x[j] = orc_rnd_float1(gen);
    mmltF(s, x, d, o);
    const float k = (float)((d + 1) * (maxDepth - 1));
    const float c[3] = {o[0] * k, o[1] * k, o[2] * k};
    if (c[0] * c[0] + c[1] * c[1] + c[2] * c[2] > 1e-20f) {
      float* px = image4 + 4 * (size_t)((int)o[4] * w + (int)o[3]);
      for (int q = 0; q < 3; q++) {
#pragma omp atomic
        px[q] += c[q];
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------------ P0: passes */
/* generator of pixel i = RandomGenInit(seed + i): the per-slot seeding of the reference's wavefront layer
 * (shaders/trace.cl:6-13 InitRandomGen) with slot = pixel, instead of the CPU layer's per-OpenMP-thread generators
 * (CPUExp_Integrators_Common.cpp:43-44), which make the reference image thread-count dependent. */
/* ------------------------------------------------------------------------------------------------ G-buffer (IHWLayer::EvalGBuffer) */
/* IntegratorCommon::gbufferEval / gbufferSample, CPUExp_GBuffer.cpp:15-113, with GBufferAll, initGBufferAll, packGBuffer1/2, projectedPixelSize,
 * surfaceSimilarity, gbuffDiff (cglobals.h:2057-2205), encodeNormal (:1401-1411), RealColorToUint32 (:711-724), PlaneHammersley
 * (globals_sys.cpp:45-61) and materialEvalDiffuse / materialLeafEvalDiffuse (cmaterial.h:2830-2916). */
#define GBUFFER_SAMPLES 64
typedef struct { float depth; f3 norm; float rgba[4]; int matId; float coverage; f2 texCoord; int objId, instId; } GBufferAll;
static void initGBufferAll(GBufferAll* g) {
  g->depth = 1e+6f; g->norm = v3(0, 0, 0); g->rgba[0] = g->rgba[1] = g->rgba[2] = 0.0f; g->rgba[3] = 1.0f; g->matId = -1; g->coverage = 0.0f;
  g->texCoord.x = g->texCoord.y = 0.0f; g->objId = -1; g->instId = -1;
}
static void PlaneHammersley(float* result, int n) {
  for (int k = 0; k < n; k++) {
    float u = 0;
    int kk = k;
    for (float p = 0.5f; kk; p *= 0.5f, kk >>= 1)
      if (kk & 1) u += p;
    const float v = (k + 0.5f) / n;
    result[2 * k + 0] = u;
    result[2 * k + 1] = v;
  }
}
static f3 materialLeafEvalDiffuse(const float* m, f2 tc, const OrcScene* s) {
  if (matType(m) == MT_LAMBERT || matType(m) == MT_OREN_NAYAR)   /* colour at 10..12, sampler ids at 13/14 in both (cmaterial.h:200-218, 264-284) */
    return mul3(sample2DExt(as_int(m[MAT_TEXMATRIXID]), tc, m, s), matColor(m));
  return v3(0, 0, 0);
}
static f3 materialEvalDiffuse(const float* a_m, f3 l, f3 n, f2 tc, const OrcScene* s) {
  f3 val = v3(0, 0, 0);
  float stackW[MIX_TREE_MAX_DEEP]; int stackO[MIX_TREE_MAX_DEEP];
  int top = 0, currOffset = 0;
  float currW = 1.0f;
  do {
    if (top > 0) { top--; currOffset = stackO[top]; currW = stackW[top]; }
    const float* m = a_m + (size_t)currOffset * MAT_FLOATS;
    if (matType(m) == MT_BLEND_MASK) {
      const float alpha = blendMaskAlpha2(m, l, n, tc, s);
      const int o1 = as_int(m[BLEND_MAT1]), o2 = as_int(m[BLEND_MAT2]);
      float w1 = alpha;
      const float w2 = 1.0f - alpha;
      if ((as_int(m[BLEND_FLAGS_OFFSET]) & BMF_REFL_WEIGHT_IS_ONE) && matType(m + (size_t)o1 * MAT_FLOATS) != MT_BLEND_MASK) w1 = 1.0f;
      if (top < MIX_TREE_MAX_DEEP) { stackW[top] = currW * w1; stackO[top] = currOffset + o1; top++; }
      if (top < MIX_TREE_MAX_DEEP) { stackW[top] = currW * w2; stackO[top] = currOffset + o2; top++; }
    } else
      val = add3(val, scale3(materialLeafEvalDiffuse(m, tc, s), currW));
  } while (top > 0);
  return val;
}
static GBufferAll gbufferSample(const OrcScene* s, f3 ray_pos, f3 ray_dir) {
  GBufferAll r;
  initGBufferAll(&r);
  const OrcHit hit = rayTrace(s, ray_pos, ray_dir, NULL);
  if (!HitSome(hit)) { r.rgba[0] = r.rgba[1] = r.rgba[2] = 0.0f; r.rgba[3] = 1.0f; return r; }
  const SurfaceHit surf = evalSurface(s, ray_pos, ray_dir, hit);
  const f3 c = materialEvalDiffuse(materialAt(s, surf.matId), ray_dir, surf.normal, surf.texCoord, s);   /* evalDiffuseColor, Common.cpp:244-251 */
  r.depth = hit.t; r.norm = surf.normal;
  r.rgba[0] = c.x; r.rgba[1] = c.y; r.rgba[2] = c.z; r.rgba[3] = 0.0f;
  r.matId = surf.matId; r.coverage = 1.0f;
  r.texCoord = surf.texCoord; r.objId = hit.geomId; r.instId = hit.instId;
  return r;
}
static float projectedPixelSize(float dist, float FOV, float w, float h) {
  const float ppx = (FOV / w) * dist, ppy = (FOV / h) * dist;
  return (dist > 0.0f) ? 2.0f * fmaxf(ppx, ppy) : 1000.0f;
}
static float surfaceSimilarity(f3 n1, float d1, f3 n2, float d2, const float MADXDIFF) {
  const float MANXDIFF = 0.15f;
  const float dist = length3(sub3(n1, n2));
  if (dist >= MANXDIFF) return 0.0f;
  if (fabsf(d1 - d2) >= MADXDIFF) return 0.0f;
  const float normalSimilar = sqrtf(1.0f - (dist / MANXDIFF));
  const float depthSimilar = sqrtf(1.0f - fabsf(d1 - d2) / MADXDIFF);
  return normalSimilar * depthSimilar;
}
static float gbuffDiff(const GBufferAll* s1, const GBufferAll* s2, const float a_fov, float w, float h) {
  const float ppSize = projectedPixelSize(s1->depth, a_fov, w, h);
  const float surfaceSimilar = surfaceSimilarity(s1->norm, s1->depth, s2->norm, s2->depth, ppSize * 2.0f);
  const float surfaceDiff = 1.0f - surfaceSimilar;
  const float objDiff = (s1->instId == s2->instId && s1->objId == s2->objId) ? 0.0f : 1.0f;
  const float matDiff = (s1->matId == s2->matId) ? 0.0f : 1.0f;
  const float alphaDiff = fabsf(s1->rgba[3] - s2->rgba[3]);
  return surfaceDiff + objDiff + matDiff + alphaDiff;
}
static GBufferAll gbufferEval(const OrcScene* s, int x, int y, int m_width, int m_height) {
  const float fov = (M_PI_F / 180.f) * 90.0f;   /* DEG_TO_RAD * 90 */
  GBufferAll samples[GBUFFER_SAMPLES];
  float qmc[2 * GBUFFER_SAMPLES];
  PlaneHammersley(qmc, GBUFFER_SAMPLES);
  const float sizeInvX = 1.0f / (float)(m_width), sizeInvY = 1.0f / (float)(m_width);   /* sic: both by the width */
  for (int i = 0; i < GBUFFER_SAMPLES; i++) {
    float lensOffs[4] = {qmc[2 * i], qmc[2 * i + 1], 0, 0};
    lensOffs[0] = sizeInvX * (lensOffs[0] + (float)x);
    lensOffs[1] = sizeInvY * (lensOffs[1] + (float)y);
    float fx, fy;
    f3 ray_pos, ray_dir;
    MakeEyeRayFromF4Rnd(lensOffs, s, &ray_pos, &ray_dir, &fx, &fy);
    samples[i] = gbufferSample(s, ray_pos, ray_dir);
  }
  float minDiff = 100000000.0f;
  int minDiffId = 0;
  for (int i = 0; i < GBUFFER_SAMPLES; i++) {
    float diff = 0.0f, coverage = 0.0f;
    for (int j = 0; j < GBUFFER_SAMPLES; j++) {
      const float thisDiff = gbuffDiff(&samples[i], &samples[j], fov, (float)m_width, (float)m_height);
      diff += thisDiff;
      if (thisDiff < 1.0f) coverage += 1.0f;
    }
    coverage *= (1.0f / (float)GBUFFER_SAMPLES);
    samples[i].coverage = coverage;
    if (diff < minDiff) { minDiff = diff; minDiffId = i; }
  }
  return samples[minDiffId];
}
static uint32_t encodeNormal(f3 n) {
  const int x = (int)(n.x * 32767.0f), y = (int)(n.y * 32767.0f);
  const uint32_t sign = (n.z >= 0) ? 0 : 1;
  const uint32_t sx = ((uint32_t)(x & 0xfffe) | sign), sy = ((uint32_t)(y & 0xffff) << 16);
  return sx | sy;
}
static uint32_t RealColorToUint32(const float c[4]) {   /* (unsigned char) of a float: the integer conversion's low byte, as the x86 build of the reference does */
  const uint32_t r = (uint32_t)(int)(c[0] * 255.0f) & 255u, g = (uint32_t)(int)(c[1] * 255.0f) & 255u;
  const uint32_t b = (uint32_t)(int)(c[2] * 255.0f) & 255u, a = (uint32_t)(int)(c[3] * 255.0f) & 255u;
  return r | (g << 8) | (b << 16) | (a << 24);
}
/* pixels [x0, x0+nx) x [y0, y0+ny) of a width x height frame; data1 / data2 = packGBuffer1 / packGBuffer2 per pixel of the window (row-major),
 * raw14 (may be null) = depth, normal, rgba, matId, coverage, texCoord, objId, instId */
void orc_gbuffer(const OrcScene* s, int width, int height, int x0, int y0, int nx, int ny, float* data1, float* data2, float* raw14) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int p = 0; p < nx * ny; p++) {
    const GBufferAll g = gbufferEval(s, x0 + p % nx, y0 + p / nx, width, height);
    const float clampedCoverage = fminf(fmaxf(g.coverage * 255.0f, 0.0f), 255.0f);
    const int compressedCoverage = (int)((uint32_t)((int)clampedCoverage) << 24);
    const int packedMIdAncCov = (g.matId & 0x00FFFFFF) | (compressedCoverage & (int)0xFF000000u);
    float* d1 = data1 + 4 * (size_t)p, *d2 = data2 + 4 * (size_t)p;
    d1[0] = g.depth; d1[1] = as_float((int)encodeNormal(g.norm)); d1[2] = as_float(packedMIdAncCov); d1[3] = as_float((int)RealColorToUint32(g.rgba));
    d2[0] = g.texCoord.x; d2[1] = g.texCoord.y; d2[2] = as_float(g.objId); d2[3] = as_float(g.instId);
    if (raw14) {
      float* o = raw14 + 14 * (size_t)p;
      o[0] = g.depth; o[1] = g.norm.x; o[2] = g.norm.y; o[3] = g.norm.z; o[4] = g.rgba[0]; o[5] = g.rgba[1]; o[6] = g.rgba[2]; o[7] = g.rgba[3];
      o[8] = as_float(g.matId); o[9] = g.coverage; o[10] = g.texCoord.x; o[11] = g.texCoord.y; o[12] = as_float(g.objId); o[13] = as_float(g.instId);
    }
  }
}

/* ------------------------------------------------------------------------------------------------ IHWLayer::NormalMapFromDisplacement */
/* CPUSharedData::NormalMapFromDisplacement + BilateralFilter, CPUBilateralFilter2D.cpp:15-246.  parity unpinned: the function lives in host C++ that
 * needs HydraAPI (absent), so no fixture from the reference itself exists; float3 normalize / float4 lerp / dot3f come from HydraAPI's LiteMath and are
 * taken as u / length(u), u + t (v - u) and x^2 + y^2 + z^2. */
static inline int clampi(int x, int a, int b) { return x < a ? a : (x > b ? b : x); }
static void BilateralFilter(const float* in4, float* out4, int w, int h, int a_windowRadius, float a_smoothLvl) {
  const float g_NoiseLevel = 1.0f / (a_smoothLvl * a_smoothLvl), g_GaussianSigma = 1.0f / 50.0f, g_WeightThreshold = 0.03f, g_LerpCoefficeint = 0.80f, g_CounterThreshold = 0.05f;
  const float windowArea = (2.0f * (float)a_windowRadius + 1.0f) * (2.0f * (float)a_windowRadius + 1.0f);
#pragma omp parallel for
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const int minX = clampi(x - a_windowRadius, 0, w - 1), maxX = clampi(x + a_windowRadius, 0, w - 1);
      const int minY = clampi(y - a_windowRadius, 0, h - 1), maxY = clampi(y + a_windowRadius, 0, h - 1);
      const float* c0 = in4 + 4 * ((size_t)y * w + x);
      int counterPass = 0;
      float fSum = 0.0f, result[4] = {0, 0, 0, 0};
      for (int y1 = minY; y1 <= maxY; y1++)
        for (int x1 = minX; x1 <= maxX; x1++) {
          const float* c1 = in4 + 4 * ((size_t)y1 * w + x1);
          const float dx = c1[0] - c0[0], dy = c1[1] - c0[1], dz = c1[2] - c0[2];
          const int i = x1 - x, j = y1 - y;
          const float w1 = dx * dx + dy * dy + dz * dz;
          const float w2 = expf(-(w1 * g_NoiseLevel + (float)(i * i + j * j) * g_GaussianSigma));
          if (w2 > g_WeightThreshold) counterPass++;
          fSum += w2;
          for (int k = 0; k < 4; k++) result[k] += c1[k] * w2;
        }
      for (int k = 0; k < 4; k++) result[k] = result[k] * (1.0f / fSum);
      const float lerpQ = ((float)counterPass > (g_CounterThreshold * windowArea)) ? 1.0f - g_LerpCoefficeint : g_LerpCoefficeint;
      for (int k = 0; k < 4; k++) out4[4 * ((size_t)y * w + x) + k] = result[k] + lerpQ * (c0[k] - result[k]);
    }
}
void orc_normal_map_from_displacement(int w, int h, const uint8_t* a_data, float bumpAmt, int invHeight, float smoothLvl, uint8_t* out) {
  float* heightDataf = (float*)malloc(sizeof(float) * (size_t)w * h);
  float* normals = (float*)malloc(sizeof(float) * 4 * (size_t)w * h);
  for (size_t i = 0; i < (size_t)w * h; i++)
    heightDataf[i] = 255.0f - fmaxf((float)a_data[4 * i], fmaxf((float)a_data[4 * i + 1], (float)a_data[4 * i + 2]));
  const float kScale = 2000.0f / fminf((float)w, (float)h);
#pragma omp parallel for
  for (int y = 0; y < h; y++) {
    const int offsetY = y * w;
    int offsetYPlusOne = (y + 1) * w, offsetYMinusOne = (y - 1) * w;
    if (y + 1 >= h) offsetYPlusOne = 0;
    if (y - 1 <= 0) offsetYMinusOne = (h - 1) * w;
    for (int x = 0; x < w; x++) {
      int offsetXPlusOne = x + 1, offsetXMinusOne = x - 1;
      if (x + 1 >= w) offsetXPlusOne = 0;
      if (x - 1 <= 0) offsetXMinusOne = w - 1;
      float diff[8];
      const float c = heightDataf[offsetY + x];
      diff[0] = c - heightDataf[offsetYMinusOne + offsetXMinusOne];
      diff[1] = c - heightDataf[offsetYMinusOne + x];
      diff[2] = c - heightDataf[offsetYMinusOne + offsetXPlusOne];
      diff[3] = c - heightDataf[offsetY + offsetXMinusOne];
      diff[4] = c - heightDataf[offsetY + offsetXPlusOne];
      diff[5] = c - heightDataf[offsetYPlusOne + offsetXMinusOne];
      diff[6] = c - heightDataf[offsetYPlusOne + x];
      diff[7] = c - heightDataf[offsetYPlusOne + offsetXPlusOne];
      if (!invHeight)
        for (int i = 0; i < 8; i++) diff[i] *= -1.0f;
      for (int i = 0; i < 8; i++) diff[i] *= (bumpAmt * bumpAmt);
      const float scale = kScale;
      const float vx[8] = {-diff[0], 0.f, diff[2], -diff[3], diff[4], -diff[5], 0.f, diff[7]};
      const float vy[8] = {-diff[0], -diff[1], -diff[2], 0.f, 0.f, diff[5], diff[6], diff[7]};
      f3 res = v3(0, 0, 0);
      for (int i = 0; i < 8; i++) { res.x += vx[i]; res.y += vy[i]; res.z += scale; }
      res = scale3(res, 1.0f / 8.0f);
      res.x *= -1.0f;
      float len = sqrtf(res.x * res.x + res.y * res.y + res.z * res.z);
      res.x /= len; res.y /= len; res.z /= len;
      if (res.z < 0.65f) {
        res.z = 0.65f;
        len = sqrtf(res.x * res.x + res.y * res.y + res.z * res.z);
        res.x /= len; res.y /= len; res.z /= len;
      }
      float* o = normals + 4 * ((size_t)offsetY + x);
      o[0] = res.x; o[1] = res.y; o[2] = res.z; o[3] = (255.0f - c) / 255.0f;
    }
  }
  if (smoothLvl >= 1.0f) {
    const int radius = 5;
    if (smoothLvl > 10.0f) smoothLvl = 10.0f;
    float* image2 = (float*)malloc(sizeof(float) * 4 * (size_t)w * h);
    BilateralFilter(normals, image2, w, h, radius, smoothLvl * 0.1f);
    free(normals);
    normals = image2;
  }
  for (size_t i = 0; i < (size_t)w * h; i++) {
    float cr[4] = {0.5f * normals[4 * i] + 0.5f, 0.5f * normals[4 * i + 1] + 0.5f, normals[4 * i + 2], normals[4 * i + 3]};
    for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)fminf(fmaxf(cr[k] * 255.0f, 0.0f), 255.0f);
  }
  free(heightDataf);
  free(normals);
}

void orc_init_generators(int w, int h, int seed, uint32_t* gens) {
  for (int i = 0; i < w * h; i++) orc_random_init(seed + i, gens + 2 * (size_t)i);
}
/* Tile partition of the HIP layer (hydra_hip.h, set_tile_partition): tiles in Morton order of (tx, ty), the i-th tile of that
 * order belongs to rank i % world.  No counterpart in the reference (replicas with different seeds). */
static uint32_t morton_spread(uint32_t v) {
  v &= 0xffffu; v = (v | (v << 8)) & 0x00ff00ffu; v = (v | (v << 4)) & 0x0f0f0f0fu; v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u;
  return v;
}
typedef struct { uint32_t code; int tile; } MortonTile;
static int morton_cmp(const void* a, const void* b) {
  const uint32_t x = ((const MortonTile*)a)->code, y = ((const MortonTile*)b)->code;
  return (x > y) - (x < y);
}
/* owner[ty * tilesX + tx]; caller frees */
static int* tile_owner_table(int w, int h, int world, int tile, int* tilesXOut) {
  const int tilesX = (w + tile - 1) / tile, tilesY = (h + tile - 1) / tile, n = tilesX * tilesY;
  MortonTile* t = (MortonTile*)malloc((size_t)n * sizeof(MortonTile));
  int* owner = (int*)malloc((size_t)n * sizeof(int));
  for (int ty = 0; ty < tilesY; ty++)
    for (int tx = 0; tx < tilesX; tx++) { t[ty * tilesX + tx].code = morton_spread((uint32_t)tx) | (morton_spread((uint32_t)ty) << 1); t[ty * tilesX + tx].tile = ty * tilesX + tx; }
  qsort(t, (size_t)n, sizeof(MortonTile), morton_cmp);
  for (int i = 0; i < n; i++) owner[t[i].tile] = i % (world > 0 ? world : 1);
  free(t);
  *tilesXOut = tilesX;
  return owner;
}
/* ref: CPUExp_Integrators_Common.cpp:278-316 IntegratorCommon::DoPass + :347-359 makeEyeRay */
uint64_t orc_render_pass(const OrcScene* s, int w, int h, uint32_t* gens, float* image4, int spp_done, int sum_mode,
                         int rank, int world, int tile, int threads) {
  const float alpha = 1.0f / (float)(spp_done + 1);
  uint64_t totalRays = 0;
  int tilesX = 1;
  int* owner = (world > 1) ? tile_owner_table(w, h, world, tile, &tilesX) : NULL;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for collapse(2) schedule(dynamic, 64) reduction(+ : totalRays)
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      if (owner && owner[(y / tile) * tilesX + (x / tile)] != rank) continue;
      uint32_t* gen = gens + 2 * ((size_t)y * w + x);
      float r4[4], offs[4];
      orc_rnd_float4(gen, r4);   /* rndUniform(gen, -1, 1), crandom.h:617-620 */
      for (int k = 0; k < 4; k++) offs[k] = -1.0f + (1.0f - (-1.0f)) * r4[k];
      f3 ray_pos, ray_dir;
      MakeRandEyeRay(x, y, w, h, offs, s, &ray_pos, &ray_dir);
      PathStat st = {0};
      g_screenX = x; g_screenY = y;
      const f3 color = PathTrace(s, ray_pos, ray_dir, gen, &st, NULL);
      totalRays += st.rays;
      float* px = image4 + 4 * ((size_t)y * w + x);
      if (sum_mode) { px[0] += color.x; px[1] += color.y; px[2] += color.z; px[3] += 0.0f; }
      else {   /* the debug red pixel of Common.cpp:295-301 is deliberately not reproduced */
        px[0] = px[0] * (1.0f - alpha) + color.x * alpha;
        px[1] = px[1] * (1.0f - alpha) + color.y * alpha;
        px[2] = px[2] * (1.0f - alpha) + color.z * alpha;
        px[3] = px[3] * (1.0f - alpha) + 0.0f * alpha;
      }
    }
  }
  free(owner);
  return totalRays;
}

int64_t orc_collect_rays(const OrcScene* s, int w, int h, int seed, int bounce, int shadow, float* pos4, float* dir4, float* tfar, int64_t cap) {
  /* every pixel records into its own slot in parallel, then the recorded slots are packed in pixel order */
  const int64_t npix = (int64_t)w * h;
  float* tp = (float*)malloc((size_t)npix * 4 * sizeof(float));
  float* td = (float*)malloc((size_t)npix * 4 * sizeof(float));
  float* tt = (float*)malloc((size_t)npix * sizeof(float));
  unsigned char* have = (unsigned char*)calloc((size_t)npix, 1);
  if (!tp || !td || !tt || !have) { free(tp); free(td); free(tt); free(have); return -1; }
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t p = 0; p < npix; p++) {
    const int x = (int)(p % w), y = (int)(p / w);
    uint32_t gen[2];
    orc_random_init(seed + (int)p, gen);
    float r4[4], offs[4];
    orc_rnd_float4(gen, r4);
    for (int k = 0; k < 4; k++) offs[k] = -1.0f + 2.0f * r4[k];
    f3 ray_pos, ray_dir;
    MakeRandEyeRay(x, y, w, h, offs, s, &ray_pos, &ray_dir);
    PathStat st = {0};
    RayProbe pr;
    memset(&pr, 0, sizeof(pr));
    pr.bounce = bounce; pr.shadow = shadow;
    (void)PathTrace(s, ray_pos, ray_dir, gen, &st, &pr);
    if (pr.have) {
      have[p] = 1;
      tp[4 * p] = pr.pos.x; tp[4 * p + 1] = pr.pos.y; tp[4 * p + 2] = pr.pos.z; tp[4 * p + 3] = 0.0f;
      td[4 * p] = pr.dir.x; td[4 * p + 1] = pr.dir.y; td[4 * p + 2] = pr.dir.z; td[4 * p + 3] = 0.0f;
      tt[p] = pr.tfar;
    }
  }
  int64_t count = 0;
  for (int64_t p = 0; p < npix && count < cap; p++) {
    if (!have[p]) continue;
    memcpy(pos4 + 4 * count, tp + 4 * p, 16);
    memcpy(dir4 + 4 * count, td + 4 * p, 16);
    if (tfar) tfar[count] = tt[p];
    count++;
  }
  free(tp); free(td); free(tt); free(have);
  return count;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
