/* hydra_layouts.h -- byte-level data contract between RenderDriverRTE-style callers and the
 * MI355X HIP layer (SURVEY.md row a/D1).
 *
 * Nothing here is executable; it names the offsets the reference kernels read so the
 * globals blob, the storage arenas, the BVH arrays and the instance tables can be handed over
 * unchanged.  Every block cites the reference file:line that defines the same layout.
 *
 * Plain C (also included from HIP device code and from the host C++ layer).
 */
#ifndef HYDRA_LAYOUTS_H
#define HYDRA_LAYOUTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- EngineGlobals header (reference hydra_drv/cfetch.h:21-81), offsets in 32-bit words ---- */
enum {
  HG_MPROJ             = 0,    /* float[16] column-major                       */
  HG_MWORLDVIEW        = 16,
  HG_MPROJ_INV         = 32,
  HG_MWORLDVIEW_INV    = 48,
  HG_VARS_I            = 64,   /* int[64]                                      */
  HG_VARS_F            = 128,  /* float[64]                                    */
  HG_RM_QMC            = 192,  /* int[16], all -1 on the pseudo-random path    */
  HG_CAM_FORWARD       = 208,  /* float[3]                                     */
  HG_CAM_UP            = 211,
  HG_CAM_LOOKAT        = 214,
  HG_IMAGE_PLANE_DIST  = 217,
  HG_TEX_TABLE_OFFS    = 218,
  HG_MAT_TABLE_OFFS    = 219,
  HG_PDF_TABLE_OFFS    = 220,
  HG_GEOM_TABLE_OFFS   = 221,
  HG_TEXAUX_TABLE_OFFS = 222,
  HG_TEX_TABLE_SIZE    = 223,
  HG_MAT_TABLE_SIZE    = 224,
  HG_PDF_TABLE_SIZE    = 225,
  HG_GEOM_TABLE_SIZE   = 226,
  HG_TEXAUX_TABLE_SIZE = 227,
  HG_FLOAT_ARRAYS_OFFS = 228,
  HG_FLOAT_ARRAYS_SIZE = 229,
  HG_LSEL_REV_OFFS     = 230,
  HG_LSEL_REV_SIZE     = 231,
  HG_LSEL_FWD_OFFS     = 232,
  HG_LSEL_FWD_SIZE     = 233,
  HG_FLAGS             = 234,
  HG_SKY_LIGHT_ID      = 235,
  HG_LIGHTS_OFFS       = 236,
  HG_LIGHTS_SIZE       = 237,
  HG_LIGHTS_NUM        = 238,
  HG_DUMMY1            = 239,
  HG_SUN_NUMBER        = 242,
  HG_SUNS              = 243,  /* 8 x 128 floats                               */
  HG_TABLES_READY      = 1267,
  HG_ESS_GGX_TABLE     = 1268, /* u16[64*64]   = 2048 words                    */
  HG_ESS_TRANSP_TABLE  = 3316, /* u16[64^3]    = 131072 words                  */
  HG_HEADER_WORDS      = 134388,
  HG_HEADER_WORDS_PADDED = 134400  /* rounded up to 16 words, IHWLayerDataAssembler.cpp:149-153 */
};

/* int / float render variables the path reads (reference hydra_drv/cglobals.h:438-538) */
enum {
  HV_I_ENABLE_DOF          = 0,
  HV_I_TRACE_DEPTH         = 9,
  HV_I_DIFFUSE_TRACE_DEPTH = 13,
  HV_I_MMLT_FIRST_BOUNCE   = 34,
  HV_I_SHADOW_MATTE_BACK      = 35,   /* texture id of the back-plate the camera sees where a ray leaves the scene (HYDRA_INVALID_TEXTURE: none), cglobals.h:475 */
  HV_I_SHADOW_MATTE_BACK_MODE = 41    /* 0 = camera projected, 1 = spherical (BACK_MODE, cglobals.h:436) */
};
enum {
  HV_F_DOF_LENS_RADIUS      = 0,
  HV_F_DOF_FOCAL_PLANE_DIST = 1,
  HV_F_TILT_ROT_X           = 2,
  HV_F_TILT_ROT_Y           = 4,
  HV_F_IMAGE_GAMMA          = 6,
  HV_F_TEXINPUT_GAMMA       = 7,
  HV_F_CAM_FOV              = 14,
  HV_F_BSPHERE_CENTER_X     = 18,
  HV_F_BSPHERE_RADIUS       = 21,
  HV_F_FOV_X                = 23,
  HV_F_FOV_Y                = 24,
  HV_F_WIDTH_F              = 25,
  HV_F_HEIGHT_F             = 26,
  HV_F_BACK_TEXINPUT_GAMMA  = 36,   /* cglobals.h:537 */
  HV_F_SHADOW_MATTE_BACK_COLOR_X = 42   /* .. Z = 44: the back-plate's colour multiplier, read from varsF by index (cbidir.h:549-551) */
};
/* g_flags bits (cglobals.h:405-434) */
enum {
  HF_COMPUTE_SHADOWS    = 1,
  HF_USE_MIS            = 32,
  HF_ENABLE_MMLT        = 16384,        /* HRT_ENABLE_MMLT, cglobals.h:419 */
  HF_STUPID_PT_MODE     = 65536 * 8,
  HF_3WAY_MIS_WEIGHTS   = 1024,           /* HRT_3WAY_MIS_WEIGHTS, cglobals.h:416 */
  HF_ENABLE_PT_CAUSTICS = 65536 * 2048
};

/* ---- PlainMesh header inside the geometry arena (cfetch.h:1038-1059), 16 words ---- */
typedef struct HydraPlainMesh {
  int32_t vPosOffset, vNormOffset, vTexCoordOffset, vIndicesOffset;
  int32_t vPosNum, vNormNum, vTexCoordNum, tIndicesNum;
  int32_t mIndicesOffset, mIndicesNum, vTangentOffset, vTangentNum;
  uint32_t totalBytesNum;
  int32_t polyShadowOffset, dummy2, pad;
} HydraPlainMesh;

/* ---- BVH node, 32 B; quad q = nodes 4q..4q+3 (cglobals.h:1280-1321) ---- */
typedef struct HydraBVHNode {
  float    boxMin[3];
  uint32_t leftOffsetAndLeaf;   /* bit31 = leaf, low 31 bits = child quad / triangle list / instance quad */
  float    boxMax[3];
  uint32_t escapeIndex;         /* 1 marks an instance leaf; 0xFFFFFFFF on inner nodes                    */
} HydraBVHNode;
#define HYDRA_BVH_INVALID 0xFFFFFFFFu
#define HYDRA_BVH_LEAF    0x80000000u

/* ---- closest-hit record, 16 B (cglobals.h:1248-1254) ---- */
typedef struct HydraLiteHit {
  float   t;
  int32_t primId, instId, geomId;
} HydraLiteHit;

/* ---- materials: 192 floats per node (cglobals.h:2657-2722, cmaterial.h) ---- */
enum {
  HM_NODE_FLOATS = 192,
  HM_TYPE = 0, HM_FLAGS = 1,
  HM_EMISSIVE_COLOR = 4, HM_EMISSIVE_TEXID = 7, HM_EMISSIVE_TEXMATRIXID = 8, HM_EMISSIVE_LIGHTID = 9,
  HM_OPACITY_TEX = 81, HM_OPACITY_TEX_MATRIX = 82, HM_NORMAL_TEX = 83, HM_NORMAL_TEX_MATRIX = 84,
  HM_EMISSIVE_SAMPLER = 88, HM_NORMAL_SAMPLER = 100, HM_OPACITY_SAMPLER = 112,
  HM_PROC_TEX_TABLE = 129, HM_AO_TYPE = 130, HM_AO_TEX_ID = 144, HM_AO_TEXMATRIX_ID = 145, HM_AO_LENGTH = 146,
  HM_AO_TYPE2 = 147, HM_AO_TEX_ID2 = 160, HM_AO_TEXMATRIX_ID2 = 161, HM_AO_LENGTH2 = 162,
  HM_PROC_TEX_IDS = 163,       /* 16 ids terminated by HYDRA_INVALID_TEXTURE */
  /* leaf BxDF fields (lambert cmaterial.h:200-210, phong :887-903, mirror :374-382) */
  HM_COLOR = 10, HM_TEXID = 13, HM_TEXMATRIXID = 14,
  HM_LAMBERT_SAMPLER = 20,
  HM_ORENNAYAR_ROUGHNESS = 15, HM_ORENNAYAR_A = 16, HM_ORENNAYAR_B = 17, HM_ORENNAYAR_SAMPLER = 20,   /* cmaterial.h:264-276 */
  HM_PHONG_COSPOWER = 15, HM_PHONG_GLOSINESS = 16, HM_PHONG_GLOSS_TEXID = 17, HM_PHONG_GLOSS_TEXMATRIXID = 18,
  HM_PHONG_SAMPLER0 = 20, HM_PHONG_SAMPLER1 = 32,
  HM_BECKMANN_ANISOTROPY = 19, HM_BECKMANN_SAMPLER2 = 44, HM_BECKMANN_SAMPLER3 = 56, HM_BECKMANN_ANISO_ROT = 68,   /* Beckmann and TRGGX nodes: cmaterial.h:1531-1556, the rest = phong's offsets */
  HM_BECKMANN_ANISO_TEXID = 69, HM_BECKMANN_ANISO_TEXMATRIXID = 70, HM_BECKMANN_ROT_TEXID = 71, HM_BECKMANN_ROT_TEXMATRIXID = 72,
  HM_BLINN_ANISOTROPY = 19,   /* BLINN_ANISOTROPY_OFFSET, cmaterial.h:1034; Blinn shares every phong offset */
  HM_MIRROR_SAMPLER = 16,
  HM_GGX_COSPOWER = 15, HM_GGX_GLOSINESS = 16, HM_GGX_GLOSS_TEXID = 17, HM_GGX_GLOSS_TEXMATRIXID = 18, HM_GGX_FRESNEL_IOR = 19,   /* cmaterial.h:1165-1185 */
  HM_GGX_SAMPLER0 = 20, HM_GGX_SAMPLER1 = 32,
  HM_THINGLASS_COS_POWER = 15, HM_THINGLASS_GLOSINESS = 16, HM_THINGLASS_GLOSS_TEXID = 17, HM_THINGLASS_GLOSS_TEXMATRIXID = 18,   /* cmaterial.h:472-491 */
  HM_THINGLASS_SAMPLER0 = 20, HM_THINGLASS_SAMPLER1 = 32,
  HM_GLASS_IOR = 15, HM_GLASS_FOG_COLOR = 16, HM_GLASS_FOG_MULT = 19, HM_GLASS_COS_POWER = 20, HM_GLASS_GLOSINESS = 21,          /* cmaterial.h:566-590 */
  HM_GLASS_GLOSS_TEXID = 22, HM_GLASS_GLOSS_TEXMATRIXID = 23, HM_GLASS_SAMPLER0 = 24, HM_GLASS_SAMPLER1 = 36,
  /* blend node (cmaterial.h:1965-2006) */
  HM_BLEND_FLAGS = 15, HM_BLEND_MAT1 = 16, HM_BLEND_MAT2 = 17, HM_BLEND_FRESNEL_IOR = 18,
  HM_BLEND_FALOFF_OFFSET = 19, HM_BLEND_FALOFF_SIZE = 20, HM_BLEND_TYPE = 21, HM_BLEND_SIGMOID_EXP = 22,
  HM_BLEND_FLAGS2 = 23, HM_BLEND_SAMPLER = 20
};
enum { /* PLAIN_MAT_TYPES cglobals.h:2604-2621 */
  HMT_PHONG = 0, HMT_BLINN = 1, HMT_MIRROR = 2, HMT_THIN_GLASS = 3, HMT_GLASS = 4, HMT_TRANSLUCENT = 5,
  HMT_SHADOW_MATTE = 6, HMT_LAMBERT = 7, HMT_OREN_NAYAR = 8, HMT_BLEND_MASK = 9, HMT_EMISSIVE = 10,
  HMT_BECKMANN = 13, HMT_TRGGX = 14, HMT_GGX = 15
};
enum { /* PLAIN_MAT_FLAGS cglobals.h:2624-2655 */
  HMF_CAST_CAUSTICS = 2, HMF_HAS_DIFFUSE = 4, HMF_HAS_TRANSPARENCY = 8, HMF_SKIP_SHADOW = 256, HMF_FORBID_EMISSIVE_GI = 512, HMF_INVIS_LIGHT = 16384,
  HMF_INVERT_NMAP_X = 16, HMF_INVERT_NMAP_Y = 32, HMF_INVERT_SWAP_NMAP_XY = 64, HMF_INVERT_HEIGHT = 128,   /* cglobals.h:2631-2634 */
  HMF_SKIP_SKY_PORTAL = 1024, HMF_HAVE_BTDF = 8192, HMF_CAN_SAMPLE_REFL_ONLY = 32768,
  HMF_HAVE_PROC_TEXTURES = 65536,   /* PLAIN_MATERIAL_HAVE_PROC_TEXTURES, cglobals.h:2647: the head lists procedural texture ids (HM_PROC_TEX_IDS) and is followed by their argument table (HM_PROC_TEX_TABLE) */
  HMF_FLIP_TANGENT = 32768 * 128,   /* cglobals.h:2653 */
  HMF_ENERGY_FIX = 32768 * 256
};
enum { /* BLEND_MASK_FLAGS cmaterial.h:1975-1979 */
  HBF_FRESNEL = 1, HBF_FALOFF = 2, HBF_REFLECTION_WEIGHT_IS_ONE = 4, HBF_EXTRUSION_STRONG = 8,
  HBF_EXTRUSION_LUMINANCE = 16
};
#define HYDRA_INVALID_TEXTURE 0xFFFFFFFEu

/* texture sampler embedded in a material / light blob, 12 words (cfetch.h:108-131) */
enum { HS_FLAGS = 0, HS_GAMMA = 1, HS_TEXID = 2, HS_DUMMY = 3, HS_ROW0 = 4, HS_ROW1 = 8 };
enum { HTEX_POINT_SAM = 1, HTEX_ALPHASRC_W = 2, HTEX_CLAMP_U = 4, HTEX_CLAMP_V = 8,
       HTEX_COORD_SECOND = 16, HTEX_COORD_CAM_PROJ = 32, HTEX_DATA_HDR = 64 /* cglobals.h:18-24 */ };

/* ---- lights: 128 floats each (clight.h:14-62, 493-521) ---- */
enum {
  HL_FLOATS = 128,
  HL_TYPE = 0, HL_FLAGS = 1, HL_POS = 2, HL_NORM = 5, HL_COLOR = 8, HL_COLOR_TEX = 11, HL_COLOR_TEX_MATRIX = 12,
  HL_SURFACE_AREA = 13, HL_SPHERE_RADIUS = 14 /* SPHERE_LIGHT_RADIUS, clight.h:33 */, HL_MESH_MESH_ID = 14, HL_MESH_TABLE_ID = 15, HL_MESH_TRI_NUM = 16, HL_MESH_MATRIX = 20, HL_MESH_TEX_ID = 30, HL_MESH_TEXMATRIX_ID = 31, HL_MESH_TEX_SAMPLER = 32 /* MESH_LIGHT_*, clight.h:169-176 */, HL_AREA_SIZE_X = 14, HL_AREA_SIZE_Y = 15, HL_AREA_MATRIX = 16, HL_AREA_IS_DISK = 25,
  HL_CYL_MATRIX = 16, HL_CYL_RADIUS = 25, HL_CYL_ZMIN = 26, HL_CYL_ZMAX = 27, HL_CYL_PHIMAX = 28, HL_CYL_TEX_ID = 29, HL_CYL_TEXMATRIX_ID = 30, HL_CYL_PDF_TABLE_ID = 31, HL_CYL_TEX_SAMPLER = 32 /* CYLINDER_*, clight.h:96-114 */,
  HL_AREA_SPOT_DISTR = 26, HL_AREA_SPOT_COS1 = 27, HL_AREA_SPOT_COS2 = 28, HL_AREA_SKY_OFFSET = 29,
  HL_AREA_SKY_SOURCE = 30, HL_AREA_SKYPORTAL_BTEX = 31, HL_AREA_SKYPORTAL_BTEX_MATRIX = 32,
  HL_AREA_SAMPLER0 = 40, HL_AREA_SAMPLER1 = 52,
  /* point/spot (clight.h:118-119) and directional lights (:123-127) */
  HL_POINT_SPOT_COS1 = 14, HL_POINT_SPOT_COS2 = 15,
  HL_DIRECT_RADIUS1 = 14, HL_DIRECT_RADIUS2 = 15, HL_DIRECT_SSOFTNESS = 16, HL_DIRECT_ALPHA_TAN = 17, HL_DIRECT_ALPHA_COS = 18,
  /* sky dome (clight.h:131-165): pdf table ids, sampler (float4 + 2 matrix rows), inverse sampler matrix (float4x4) */
  HL_SKY_COLOR_AUX = 17, HL_SKY_COLOR_TEX_AUX = 20, HL_SKY_COLOR_TEX_MATRIX_AUX = 21, HL_SKY_AUX_TEX_MATRIX_INV = 22,
  HL_SKY_SUN_DIR = 23, HL_SKY_TURBIDITY = 26, HL_SKY_SUN_COLOR = 27, HL_SKY_PDF_TABLE0 = 30, HL_SKY_PDF_TABLE1 = 31,
  HL_SKY_SAMPLER0 = 32, HL_SKY_MATRIX0 = 36, HL_SKY_SAMPLER1 = 44, HL_SKY_MATRIX1 = 48, HL_SKY_INV_MATRIX0 = 56,
  HL_SKY_INV_MATRIX1 = 72, HL_SKY_SUN_DIR_ID = 88,
  HL_PROB_MULT = 104, HL_GROUP_ID = 105, HL_PICK_PROB_FWD = 106, HL_PICK_PROB_REV = 107,
  HL_IES_INV_MATRIX = 108, HL_IES_LIGHT_MATRIX = 117, HL_IES_SPHERE_PDF_ID = 126, HL_IES_SPHERE_TEX_ID = 127
};
enum { HLT_POINT_OMNI = 0, HLT_POINT_SPOT = 1, HLT_DIRECT = 2, HLT_SKY_DOME = 3, HLT_AREA = 4,
       HLT_SPHERE = 5, HLT_CYLINDER = 6, HLT_MESH = 7 };
enum { HLF_DISABLE_SAMPLING = 1, HLF_SKY_USE_PEREZ = 4, HLF_SKY_PORTAL = 8, HLF_HAS_IES = 16, HLF_IES_POINT_AREA = 32,
       HLF_DO_NOT_SAMPLE_ME = 64 };   /* cglobals.h:2245-2254 */

/* ---- ray flags word (cglobals.h:1330-1376): diffuse bounces | bounces<<8 | events<<16 ---- */
enum { HRE_S = 1, HRE_D = 2, HRE_G = 4, HRE_T = 8, HRE_THINGLASS = 64 };   /* cglobals.h:1331-1340 */
enum { HRF_OUT_OF_SCENE = 128, HRF_IS_DEAD = 4096 };

/* ---- per-kernel timing record returned by hydra_hip_get_stats (MRaysStat, cglobals.h:1764-1787) ---- */
typedef struct HydraRaysStat {
  float raysPerSec;
  float traversalTimeMs, samLightTimeMs, shadowTimeMs, shadeTimeMs, bounceTimeMs, evalHitMs, nextBounceMs;
  float raygenTimeMs, accumTimeMs, passTimeMs;
  int32_t traceTimePerCent;
  uint64_t extensionRays, shadowRays, samples;
  uint64_t traceLaunches, shadowLaunches;   /* kernel launches behind traversalTimeMs / shadowTimeMs */
} HydraRaysStat;

#ifdef __cplusplus
}
#endif
#endif
