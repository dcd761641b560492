// hk_bidir.h -- device functions of the bidirectional integrators (row f3: MMLT / SBDPT, hydra_drv/CPUExp_Integrators_MMLT.cpp):
// forward light sampling with its pdfs, the camera connection factor and screen projection, the Kelemen primary-space mutation.
// First milestone: the building blocks, each pinned one call at a time (hydra_hip_stage_light_sample_forward, ..._light_pdf_fwd,
// ..._camera_connect, ..._mutate_kelemen); the path builders (CameraPath / LightPath / ConnectEye / ConnectShadow /
// ConnectEndPoints, MMLT.cpp:637-1047) are not built yet.
#pragma once
#include "hk_shading.h"

struct LightSampleFwd { f3 pos, dir, norm, color; float pdfA, pdfW, cosTheta; bool isPoint; };   // clight.h:632-643
struct LightPdfFwd { float pdfA, pdfW, pickProb; };                                               // clight.h:646-651

HK_DEV f3 UniformSampleSphere(float u1, float u2) {   // cglobals.h:1160-1168
  const float z = 1.0f - 2.0f * u1;
  const float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
  const float phi = 2.0f * HK_PI * u2;
  return mk3(r * cosf(phi), r * sinf(phi), z);
}
HK_DEV f3 lightMatrixMul(const float* M, f3 v) {   // matrix3x3f_mult_float3, row-major 3x3
  return mk3(M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z);
}
// LightSampleIESSphere, clight.h:411-426: a direction drawn from the photometric web's 2-D table, turned into world space by the inverse IES matrix
HK_DEV void LightSampleIESSphere(const SceneDev& s, const float* L, f3 rands, f3& outDir, float& outPdfW) {
  const float* hdr = pdfTableHeader(s, as_int(L[HL_IES_SPHERE_PDF_ID]));
  const Map2DSample sample = sampleMap2D(rands, hdr + 4, as_int(hdr[0]), as_int(hdr[1]));
  float sinTheta = 0.0f;
  const f3 lsDir = texCoord2DToSphereMap(sample.texCoord, sinTheta);
  outDir = normalize(lightMatrixMul3(L + HL_IES_INV_MATRIX, lsDir));
  outPdfW = HK_INV_PI * HK_INV_PI * 0.5f * (sample.mapPdf / fmaxf(fabsf(sinTheta), HK_DEPSILON2));
}
// clight.h:654-719
HK_DEV void AreaLightSampleForward(const SceneDev& s, const float* L, float4 rands, float rands2x, LightSampleFwd& out) {
  const float offsetX = rands.x * 2.0f - 1.0f, offsetY = rands.y * 2.0f - 1.0f;
  f3 samplePos = mk3(offsetX * L[HL_AREA_SIZE_X], 0.0f, offsetY * L[HL_AREA_SIZE_Y]);
  if (as_int(L[HL_AREA_IS_DISK]) != 0) {
    const f2 d = MapSamplesToDisc(mk2(offsetX, offsetY));
    samplePos = mk3(d.x * L[HL_AREA_SIZE_X], 0.0f, d.y * L[HL_AREA_SIZE_X]);
  }
  samplePos = lightMatrixMul(L + HL_AREA_MATRIX, samplePos) + lightPos(L);
  if (as_int(L[HL_FLAGS]) & HLF_IES_POINT_AREA) samplePos = lightPos(L);
  f3 lnorm = lightNorm(L);
  f3 sampleDir = MapSampleToCosineDistribution(rands.z, rands.w, lnorm, lnorm, 1.0f);
  float cosTheta = fmaxf(dot(sampleDir, lnorm), 0.0f);
  float pdfW = cosTheta * HK_INV_PI;
  if (as_int(L[HL_FLAGS]) & HLF_HAS_IES) {
    LightSampleIESSphere(s, L, mk3(rands.z, rands.w, rands2x), sampleDir, pdfW);
    lnorm = dot(lnorm, sampleDir) > 0.0f ? lnorm : lnorm * (-1.0f);
  } else if (as_int(L[HL_AREA_SPOT_DISTR]) != 0) {
    const float cos2 = L[HL_AREA_SPOT_COS2];
    sampleDir = MapSamplesToCone(cos2, mk2(rands.z, rands.w), lnorm);
    pdfW = 1.0f / (2.0f * HK_PI * (1.0f - cos2));
  }
  cosTheta = fmaxf(dot(sampleDir, lnorm), 0.0f);
  const f3 color = (as_int(L[HL_FLAGS]) & HLF_SKY_PORTAL) ? areaLightSkyPortalCustomColor<HK_FEAT_ALL>(s, L, sampleDir * (-1.0f)) : areaLightIntensity<HK_FEAT_ALL>(s, L, sampleDir * (-1.0f), false);
  out.isPoint = false;
  out.pos = samplePos + lnorm * epsilonOfPos(samplePos);
  out.dir = sampleDir;
  out.color = color * cosTheta;
  out.pdfA = 1.0f / L[HL_SURFACE_AREA];
  out.pdfW = pdfW;
  out.cosTheta = cosTheta;
  out.norm = lnorm;
}
HK_DEV void PointLightSampleForward(const SceneDev& s, const float* L, float4 rands, LightSampleFwd& out) {   // clight.h:838-862
  float pdfW = HK_INV_PI * 0.25f;
  f3 sampleDir = UniformSampleSphere(rands.x, rands.y);
  const bool ies = (as_int(L[HL_FLAGS]) & HLF_HAS_IES) != 0;
  if (ies) LightSampleIESSphere(s, L, mk3(rands.x, rands.y, rands.z), sampleDir, pdfW);
  const f3 samplePos = lightPos(L);
  const float mask = ies ? lightDistributionMask(s, L, sampleDir * (-1.0f)) : 1.0f;
  out.isPoint = true;
  out.pos = samplePos + sampleDir * epsilonOfPos(samplePos);
  out.dir = sampleDir;
  out.color = (mk3(mask, mask, mask) * lightColor(L)) * (1.0f / L[HL_SURFACE_AREA]);   // pointLightGetIntensity: the base colour without IES
  out.pdfA = 1.0f / L[HL_SURFACE_AREA];
  out.pdfW = pdfW;
  out.cosTheta = 1.0f;
  out.norm = sampleDir;
}
HK_DEV void PointSpotSampleForward(const float* L, float4 rands, LightSampleFwd& out) {   // clight.h:865-890
  const f3 lnorm = lightNorm(L), samplePos = lightPos(L);
  const float cos1 = L[HL_POINT_SPOT_COS1], cos2 = L[HL_POINT_SPOT_COS2];
  const f3 sampleDir = MapSamplesToCone(cos2, mk2(rands.x, rands.y), lnorm);
  const float cosThetaOut = fmaxf(dot(sampleDir, lnorm), 0.0f);
  const float k1 = mylocalsmoothstep(cos2, cos1, cosThetaOut);
  out.isPoint = true;
  out.pos = samplePos + sampleDir * epsilonOfPos(samplePos);
  out.dir = sampleDir;
  out.color = (lightColor(L) * k1) * (1.0f / L[HL_SURFACE_AREA]);
  out.pdfA = 1.0f / L[HL_SURFACE_AREA];
  out.pdfW = 1.0f / (2.0f * HK_PI * (1.0f - cos2));
  out.cosTheta = cosThetaOut;
  out.norm = sampleDir;
}
HK_DEV void DirectLightSampleForward(const float* L, float4 rands, LightSampleFwd& out) {   // clight.h:915-958
  const f3 lnorm = lightNorm(L), lcenter = lightPos(L);
  const float radius1 = L[HL_DIRECT_RADIUS1], radius2 = L[HL_DIRECT_RADIUS2];
  const f2 d0 = MapSamplesToDisc(mk2(2.0f * (rands.x - 0.5f), 2.0f * (rands.y - 0.5f)));
  const f2 diskSam = mk2(radius2 * d0.x, radius2 * d0.y);
  const float d = sqrtf(diskSam.x * diskSam.x + diskSam.y * diskSam.y);
  const float atten = mylocalsmoothstep(fmaxf(radius2, radius1), fminf(radius2, radius1), d);
  f3 nx, nz;
  CoordinateSystem(lnorm, nx, nz);
  const f3 samplePos = (lcenter + nx * diskSam.x) + nz * diskSam.y;
  f3 sampleDir = lnorm;
  const float pdfW = 1.0f;
  if (L[HL_DIRECT_SSOFTNESS] > 1e-5f) sampleDir = MapSamplesToCone(L[HL_DIRECT_ALPHA_COS], mk2(rands.z, rands.w), lnorm);
  out.isPoint = true;
  out.pos = samplePos + sampleDir * epsilonOfPos(samplePos);
  out.dir = sampleDir;
  out.color = (lightColor(L) * atten) * pdfW;
  out.pdfA = 1.0f / L[HL_SURFACE_AREA];
  out.pdfW = pdfW;
  out.cosTheta = 1.0f;
  out.norm = sampleDir;
}
// LightSampleForward, clight.h:1064-1110: the light types this layer accepts; the sky dome takes the default branch there too
HK_DEV void SphereLightSampleForward(const float* L, float4 rands, LightSampleFwd& out) {   // clight.h:720-751
  const f3 lcenter = lightPos(L);
  const f3 samplePos = lcenter + (sphereLightUnitSample(rands.x, rands.y) * L[HL_SPHERE_RADIUS]);
  const f3 lnorm = normalize(samplePos - lcenter);
  const f3 sampleDir = MapSampleToCosineDistribution(rands.z, rands.w, lnorm, lnorm, 1.0f);
  const float cosTheta = fmaxf(dot(sampleDir, lnorm), 0.0f);
  out.isPoint = false;
  out.pos = samplePos + lnorm * epsilonOfPos(samplePos);
  out.dir = sampleDir;
  out.color = lightColor(L) * cosTheta;
  out.pdfA = 1.0f / L[HL_SURFACE_AREA];
  out.pdfW = cosTheta * HK_INV_PI;
  out.cosTheta = cosTheta;
  out.norm = lnorm;
}
HK_DEV void MeshLightSampleForward(const SceneDev& s, const float* L, float4 rands, float rands2x, LightSampleFwd& out) {   // clight.h:1023-1062
  f3 samplePos, sampleNorm; f2 tc; float pdfA;
  MeshLightSamplePos(s, L, mk3(rands.x, rands.y, rands2x), samplePos, sampleNorm, tc, pdfA);
  samplePos = meshLightMatrixMul(L + HL_MESH_MATRIX, samplePos);
  sampleNorm = normalize(meshLightMatrixMul(L + HL_MESH_MATRIX, sampleNorm));
  samplePos = samplePos + lightPos(L);
  const f3 sampleDir = MapSampleToCosineDistribution(rands.z, rands.w, sampleNorm, sampleNorm, 1.0f);
  const float cosTheta = fmaxf(dot(sampleDir, sampleNorm), 0.0f);
  out.isPoint = false;
  out.pos = samplePos + sampleNorm * epsilonOfPos(samplePos);
  out.dir = sampleDir;
  out.color = meshLightGetIntensity(s, L, tc) * cosTheta;
  out.pdfA = 1.0f / L[HL_SURFACE_AREA];
  out.pdfW = cosTheta * HK_INV_PI;
  out.cosTheta = cosTheta;
  out.norm = sampleNorm;
}
HK_DEV void CylinderLightSampleForward(const SceneDev& s, const float* L, float4 rands, float rands2x, LightSampleFwd& out) {   // clight.h:813-835
  f3 samplePos, lnorm; f2 tc; float pdfA;
  CylinderLightSamplePos(s, L, mk3(rands.x, rands.y, rands2x), samplePos, lnorm, tc, pdfA);
  const f3 sampleDir = MapSampleToCosineDistribution(rands.z, rands.w, lnorm, lnorm, 1.0f);
  const float cosTheta = fmaxf(dot(sampleDir, lnorm), 0.0f);
  out.isPoint = false;
  out.pos = samplePos + lnorm * epsilonOfPos(samplePos);
  out.dir = sampleDir;
  out.color = cylinderLightGetIntensity(s, L, tc) * cosTheta;
  out.pdfA = pdfA;
  out.pdfW = cosTheta * HK_INV_PI;
  out.cosTheta = cosTheta;
  out.norm = lnorm;
}
// rands2x: the first of the two extra light dimensions (LightGroup2::group2.x, primary-sample dimension 8), which only mesh and cylinder lights read
HK_DEV void LightSampleForward(const SceneDev& s, const float* L, float4 rands, float rands2x, LightSampleFwd& out) {
  switch (as_int(L[HL_TYPE])) {
    case HLT_MESH: MeshLightSampleForward(s, L, rands, rands2x, out); break;
    case HLT_CYLINDER: CylinderLightSampleForward(s, L, rands, rands2x, out); break;
    case HLT_SPHERE: SphereLightSampleForward(L, rands, out); break;
    case HLT_DIRECT: DirectLightSampleForward(L, rands, out); break;
    case HLT_POINT_SPOT: PointSpotSampleForward(L, rands, out); break;
    case HLT_POINT_OMNI: PointLightSampleForward(s, L, rands, out); break;
    default: AreaLightSampleForward(s, L, rands, rands2x, out); break;
  }
}
// lightPdfFwd, clight.h:1117-1175
HK_DEV LightPdfFwd lightPdfFwd(const SceneDev& s, const float* L, f3 ray_dir, float cosTheta) {
  LightPdfFwd res;
  res.pdfA = 1.0f / L[HL_SURFACE_AREA];
  res.pdfW = fmaxf(cosTheta * HK_INV_PI, 0.0f);
  res.pickProb = L[HL_PICK_PROB_FWD];
  const int ltype = as_int(L[HL_TYPE]);
  if (ltype == HLT_POINT_OMNI) res.pdfW = HK_INV_PI * 0.25f;
  else if (ltype == HLT_POINT_SPOT) {
    const float cos2 = L[HL_POINT_SPOT_COS2];
    res.pdfW = 1.0f / (2.0f * HK_PI * (1.0f - cos2));
    if (cosTheta < cos2) res.pdfW = 0.0f;
  } else if (ltype == HLT_DIRECT) {
    const float radius2 = L[HL_DIRECT_RADIUS2];
    res.pdfA = 1.0f / (HK_PI * radius2 * radius2);
    res.pdfW = 0.0f;
  }
  if (as_int(L[HL_FLAGS]) & HLF_HAS_IES) {
    const f3 rayDir = lightMatrixMul3(L + HL_IES_LIGHT_MATRIX, ray_dir);
    const float* hdr = pdfTableHeader(s, as_int(L[HL_IES_SPHERE_PDF_ID]));
    float sintheta = 0.0f;
    const f2 tc = sphereMapTo2DTexCoord(rayDir * (-1.0f), sintheta);
    const float mapPdf = evalMap2DPdf(tc, hdr + 4, as_int(hdr[0]), as_int(hdr[1]));
    res.pdfW = mapPdf / (2.f * HK_PI * HK_PI * fmaxf(sintheta, HK_DEPSILON2));
  } else if (ltype == HLT_AREA && as_int(L[HL_AREA_SPOT_DISTR]) != 0) {
    const float cos2 = L[HL_AREA_SPOT_COS2];
    res.pdfW = 1.0f / (2.0f * HK_PI * (1.0f - cos2));
    if (cosTheta < cos2) res.pdfW = 0.0f;
  }
  return res;
}
// CameraImageToSurfaceFactor, cbidir.h:78-115
HK_DEV float CameraImageToSurfaceFactor(const SceneDev& s, f3 hitPos, f3 hitNorm, f2 diskOffs, f3& camDirOut, float& zDepthOut) {
  const float* gf = reinterpret_cast<const float*>(s.globals);
  const m44 wvInv = load_m44(reinterpret_cast<const float4*>(s.globals + HG_MWORLDVIEW_INV));
  const f3 camForward = mk3(gf[HG_CAM_FORWARD], gf[HG_CAM_FORWARD + 1], gf[HG_CAM_FORWARD + 2]);
  const f3 camUp = mk3(gf[HG_CAM_UP], gf[HG_CAM_UP + 1], gf[HG_CAM_UP + 2]);
  const f3 camLeft = normalize(cross(camForward, camUp));
  const float imagePlaneDist = gf[HG_IMAGE_PLANE_DIST];
  const float lensR = g_varsF(s)[HV_F_DOF_LENS_RADIUS];
  const f3 camPos = (mul4x3(wvInv, mk3(0, 0, 0)) + (camUp * diskOffs.y) * lensR) + (camLeft * diskOffs.x) * lensR;
  const float zDepth = length(camPos - hitPos);
  const f3 camDir = (camPos - hitPos) * (1.0f / zDepth);
  camDirOut = camDir;
  zDepthOut = zDepth;
  const float cosToCamera = fabsf(dot(hitNorm, camDir));
  const float cosAtCamera = dot(camForward, camDir * (-1.0f));
  const float relation = g_varsF(s)[HV_F_WIDTH_F] / g_varsF(s)[HV_F_HEIGHT_F];
  const float fov = relation * fmaxf(g_varsF(s)[HV_F_FOV_X], g_varsF(s)[HV_F_FOV_Y]);
  if (cosAtCamera <= cosf(fov)) return 0.0f;
  const float imagePointToCameraDist = imagePlaneDist / cosAtCamera;
  const float imageToSolidAngleFactor = (imagePointToCameraDist * imagePointToCameraDist) / cosAtCamera;
  const float imageToSurfaceFactor = imageToSolidAngleFactor * cosToCamera / (zDepth * zDepth);
  return isfinite(imageToSurfaceFactor) ? imageToSurfaceFactor / (relation * relation) : 0.0f;
}
// worldPosToScreenSpace, cbidir.h:123-131
HK_DEV f2 worldPosToScreenSpace(const SceneDev& s, f3 wpos) {
  const m44 wv = load_m44(reinterpret_cast<const float4*>(s.globals + HG_MWORLDVIEW)), proj = load_m44(reinterpret_cast<const float4*>(s.globals + HG_MPROJ));
  const float4 cam = mul4x4x4(wv, mk4(wpos, 1.0f));
  const float4 ndc = mul4x4x4(proj, cam);
  const float inv = 1.0f / fmaxf(ndc.w, HK_DEPSILON);
  return mk2((ndc.x * inv * 0.5f + 0.5f) * g_varsF(s)[HV_F_WIDTH_F], (ndc.y * inv * 0.5f + 0.5f) * g_varsF(s)[HV_F_HEIGHT_F]);
}
// MutateKelemen, crandom.h:189-210
HK_DEV float MutateKelemen(float valueX, f2 rands, float p2, float p1) {
  const float s1 = 1.0f / p1, s2 = 1.0f / p2;
  const float power = -logf(s2 / s1);
  const float dv = fmaxf(s2 * (expf(power * sqrtf(rands.x)) - expf(power)), 0.0f);
  if (rands.y < 0.5f) { valueX += dv; if (valueX > 1.0f) valueX -= 1.0f; }
  else { valueX -= dv; if (valueX < 0.0f) valueX += 1.0f; }
  return valueX;
}

// ================================================================================================ IntegratorMMLT::F, wavefront form
// The contribution function of multiplexed MLT (hydra_drv/CPUExp_Integrators_MMLT.cpp:146-315) for many chains at once, cut where the
// reference calls rayTrace / shadowTrace so that the traversal kernels of the path tracer do that work:
//   mmltBegin        F :150-199 up to the first camera ray, LightPath :637-669 up to the first light ray
//   (the callers keep the rays: hydra_hip.hip in compacted per-level queues, the host emulation in plain arrays)
//   mmltCameraStep   one level of CameraPath :756-929 (the recursion's return-trip products are kept per level and applied in mmltConnectEnd)
//   mmltLightStep    one level of TraceLightPath :671-754
//   mmltConnectBegin the rays of ConnectEye :931-958 (closest hit towards the camera), ConnectShadow :962-1009 and ConnectEndPoints :1011-1047 (shadow rays)
//   mmltConnectEnd   ConnectEyeP / ConnectShadowP / ConnectEndPointsP (cbidir.h:190-477), the MIS weight F :252-285 and the screen test :289-302
// Every random number comes from the chain's primary-sample vector (gen.rptr != 0, crandom.h:340-520).  Chain state lives in planes of n floats
// (plane p of chain i = st[p * n + i]) so that a wave reads and writes whole cache lines; x vectors are stored the same way (x[j * n + i]).
#define HK_MMLT_MAX_DEPTH 16
#define HK_MMLT_HEAD 12            // MMLT_HEAD_TOTAL_SIZE, cglobals.h:102
#define HK_MMLT_PER_BOUNCE 10      // MMLT_FLOATS_PER_BOUNCE, cglobals.h:126-128
enum {
  MP_S = 0, MP_BITS = 1, MP_X = 2, MP_Y = 3, MP_FLAGS = 4, MP_MIS_PDF = 5, MP_MIS_COS = 6,
  MP_CV_GTERM = 7, MP_CV_DIR = 8, MP_CV_ACC = 11, MP_CV_HIT = 14,            // hit record: pos 3, normal 3, flat normal 3, uv 2, matId, t, sRayOff, hfi, tangent 3, bitangent 3 = 21 planes
  MP_L_COLOR = 35, MP_L_COS = 38, MP_L_PDF = 39,
  MP_LV_GTERM = 40, MP_LV_DIR = 41, MP_LV_ACC = 44, MP_LV_HIT = 47,
  MP_NFAC = 68, MP_ZERO_FROM = 69, MP_LBITS = 70, MP_PDF = 71                // then 2 * (maxD + 1) pdf planes and 3 * maxD factor planes
};
// the camera and the light sub-path of a chain advance in different threads of one launch: each side owns its flag word (MP_BITS / MP_LBITS),
// its rays and its pdf entries (camera: s+1..d, light: 0..s-1), so the two never write the same word
enum { MB_CAM_ACTIVE = 1, MB_CV_VALID = 4, MB_CV_SPEC_ONLY = 16, MB_MIS_SPECULAR = 32 };   // MP_BITS
enum { MB_LIGHT_ACTIVE = 2, MB_LV_VALID = 8, MB_LIGHT_SPECULAR = 64 };                      // MP_LBITS
#define mmltPlanes(maxD) (MP_PDF + 2 * ((maxD) + 1) + 3 * (maxD))
#define mmltStride(maxD) (HK_MMLT_HEAD + HK_MMLT_PER_BOUNCE * (maxD))   // randArraySizeOfDepthMMLT, crandom.h:630-633

struct MmltView {
  int n, maxD;
  float* st;                        // mmltPlanes(maxD) planes of n floats
  const float* x;                   // mmltStride(maxD) planes of n floats: the primary-sample vectors
  const int* depth;                 // d per chain, 1..maxD
  float4* eyePos; float4* eyeDir; const HydraLiteHit* eyeHit;   // n: connection towards the camera (closest hit)
  float4* shPos; float4* shDir; const float* shVis;             // n: shadow connections, t_far in shPos.w
  float* out8;                      // n x 8: colour, x, y, split, MIS weight, contribFunc
};
HK_DEV float& mst(const MmltView& v, int plane, int i) { return v.st[size_t(plane) * v.n + i]; }
HK_DEV int& msti(const MmltView& v, int plane, int i) { return reinterpret_cast<int*>(v.st)[size_t(plane) * v.n + i]; }
HK_DEV float mx(const MmltView& v, int j, int i) { return v.x[size_t(j) * v.n + i]; }
HK_DEV f3 mst3(const MmltView& v, int plane, int i) { return mk3(mst(v, plane, i), mst(v, plane + 1, i), mst(v, plane + 2, i)); }
HK_DEV void mstSet3(const MmltView& v, int plane, int i, f3 a) { mst(v, plane, i) = a.x; mst(v, plane + 1, i) = a.y; mst(v, plane + 2, i) = a.z; }
HK_DEV float& mpdfFwd(const MmltView& v, int k, int i) { return mst(v, MP_PDF + 2 * k, i); }
HK_DEV float& mpdfRev(const MmltView& v, int k, int i) { return mst(v, MP_PDF + 2 * k + 1, i); }
HK_DEV void mstoreHit(const MmltView& v, int plane, int i, const SurfaceHit& h) {
  mstSet3(v, plane, i, h.pos); mstSet3(v, plane + 3, i, h.normal); mstSet3(v, plane + 6, i, h.flatNormal);
  mst(v, plane + 9, i) = h.texCoord.x; mst(v, plane + 10, i) = h.texCoord.y;
  msti(v, plane + 11, i) = h.matId; mst(v, plane + 12, i) = h.t; mst(v, plane + 13, i) = h.sRayOff; msti(v, plane + 14, i) = h.hfi ? 1 : 0;
  mstSet3(v, plane + 15, i, h.tangent); mstSet3(v, plane + 18, i, h.biTangent);
}
HK_DEV SurfaceHit mloadHit(const MmltView& v, int plane, int i) {
  SurfaceHit h;
  h.pos = mst3(v, plane, i); h.normal = mst3(v, plane + 3, i); h.flatNormal = mst3(v, plane + 6, i);
  h.tangent = mst3(v, plane + 15, i); h.biTangent = mst3(v, plane + 18, i);
  h.texCoord = mk2(mst(v, plane + 9, i), mst(v, plane + 10, i));
  h.matId = msti(v, plane + 11, i); h.t = mst(v, plane + 12, i); h.sRayOff = mst(v, plane + 13, i); h.hfi = msti(v, plane + 14, i) != 0;
  return h;
}
HK_DEV void mmltDeadRay(float4& pos, float4& dir) {   // misses the root box of any scene; the traversal kernels take every ray of the connection arrays; sub-path rays are compacted instead
  pos = make_float4(1e18f, 1e18f, 1e18f, 0.0f);
  dir = make_float4(0.57735026f, 0.57735026f, 0.57735026f, 0.0f);
}
HK_DEV int mapRndFloatToInt(float a_val, int a, int b) {   // crandom.h:507-518
  const float fa = float(a + 0), fb = float(b + 1);
  const int res = int(fa + a_val * (fb - fa));
  return (res > b) ? b : res;
}
HK_DEV bool isPureSpecular(const MatSample& ms) { return (ms.flags & HRE_S) != 0 || (ms.flags & HRE_T) != 0; }   // cglobals.h:1342
HK_DEV bool flagsHaveOnlySpecular(uint32_t flags) { const uint32_t other = flags >> 16; return ((other & HRE_G) == 0) && ((other & HRE_D) == 0); }   // cmaterial.h:3295-3301
HK_DEV f3 div3s(f3 a, float b) { return mk3(a.x / b, a.y / b, a.z / b); }
HK_DEV void MakeEyeRayFromF4Rnd(float4 lensOffs, const SceneDev& s, f3& outPos, f3& outDir, float& pX, float& pY) {   // cfetch.h:933-968
  const float fwidth = g_varsF(s)[HV_F_WIDTH_F], fheight = g_varsF(s)[HV_F_HEIGHT_F];
  const float x = fwidth * lensOffs.x, y = fheight * lensOffs.y;
  const m44 projInv = load_m44(reinterpret_cast<const float4*>(s.globals + HG_MPROJ_INV));
  const m44 wvInv = load_m44(reinterpret_cast<const float4*>(s.globals + HG_MWORLDVIEW_INV));
  f3 ray_pos = mk3(0.0f, 0.0f, 0.0f);
  f3 ray_dir = EyeRayDirNormalized(x / fwidth, y / fheight, projInv);
  ray_dir = tiltCorrection(ray_pos, ray_dir, s);
  if (g_varsI(s)[HV_I_ENABLE_DOF] == 1) {
    const float tFocus = g_varsF(s)[HV_F_DOF_FOCAL_PLANE_DIST] / (-ray_dir.z);
    const f3 focusPosition = ray_pos + ray_dir * tFocus;
    const f2 d = MapSamplesToDisc(mk2(lensOffs.z - 0.5f, lensOffs.w - 0.5f));
    const float k = g_varsF(s)[HV_F_DOF_LENS_RADIUS] * 2.0f;
    ray_pos.x += k * d.x;
    ray_pos.y += k * d.y;
    ray_dir = normalize(focusPosition - ray_pos);
  }
  const f3 pos = mul4x3(wvInv, ray_pos);
  const f3 pos2 = mul4x3(wvInv, ray_pos + ray_dir * 100.0f);
  outPos = pos;
  outDir = normalize(pos2 - pos);
  pX = lensOffs.x * fwidth;
  pY = lensOffs.y * fheight;
}
HK_DEV int SelectRandomLightFwd(float r, const SceneDev& s, float& pickProb) {   // clight.h:1808-1822
  const int tableSize = s.hdr[HG_LSEL_FWD_SIZE];
  pickProb = 1.0f;
  if (tableSize <= 2) return 0;
  return SelectIndexPropToOpt(r, reinterpret_cast<const float*>(s.globals + s.hdr[HG_LSEL_FWD_OFFS]), tableSize, pickProb);
}
HK_DEV ShadeContext mmltShadeContext(const SurfaceHit& h, f3 l, f3 v) {
  ShadeContext sc;
  sc.l = l; sc.v = v; sc.n = h.normal; sc.tc = h.texCoord; sc.fn = h.flatNormal; sc.tg = h.tangent; sc.bn = h.biTangent;
  return sc;
}
HK_DEV void mmltRands(const MmltView& v, int i, int base, float* rands) {   // RndMatAll with rptr set, crandom.h:478-494
  for (int k = 0; k < HK_MMLT_PER_BOUNCE; k++) rands[k] = mx(v, base + k, i);
}

// returns the first camera ray and the first light ray of chain i (camActive / lightActive say which exist)
HK_DEV void mmltBegin(const SceneDev& s, const MmltView& v, int i, float4& cpos, float4& cdir, bool& camActive, float4& lpos, float4& ldir, bool& lightActive) {
  const int d = v.depth[i];
  for (int k = 0; k <= d; k++) { mpdfFwd(v, k, i) = 0.0f; mpdfRev(v, k, i) = 0.0f; }
  const int width = int(g_varsF(s)[HV_F_WIDTH_F]), height = int(g_varsF(s)[HV_F_HEIGHT_F]);
  const int sp = mapRndFloatToInt(mx(v, 11, i), 0, d), t = d - sp;   // rndSplitMMLT, MMLT_DIM_SPLIT
  const int lightTraceDepth = sp - 1, camTraceDepth = t;
  const float4 lensOffs = make_float4(mx(v, 0, i), mx(v, 1, i), mx(v, 2, i), mx(v, 3, i));   // rndLens
  int x = int(lensOffs.x * float(width) + 0.5f), y = int(lensOffs.y * float(height) + 0.5f);
  int bits = 0, lbits = 0;
  mmltDeadRay(cpos, cdir); mmltDeadRay(lpos, ldir);
  // InitPathVertex, cbidir.h:26-33
  mst(v, MP_CV_GTERM, i) = 1.0f; mstSet3(v, MP_CV_ACC, i, mk3(1, 1, 1));
  mst(v, MP_LV_GTERM, i) = 1.0f; mstSet3(v, MP_LV_ACC, i, mk3(1, 1, 1));
  msti(v, MP_NFAC, i) = 0; msti(v, MP_ZERO_FROM, i) = -1;
  if (camTraceDepth > 0) {
    f3 rp, rd; float fx, fy;
    MakeEyeRayFromF4Rnd(lensOffs, s, rp, rd, fx, fy);
    x = int(fx + 0.5f); y = int(fy + 0.5f);
    if (x >= width) x = width - 1;
    if (y >= height) y = height - 1;
    cpos = make_float4(rp.x, rp.y, rp.z, 0.0f); cdir = make_float4(rd.x, rd.y, rd.z, 0.0f);
    bits |= MB_CAM_ACTIVE | MB_MIS_SPECULAR;                 // makeInitialMisData: pdf 1, cos 1, specular
    mst(v, MP_MIS_PDF, i) = 1.0f; mst(v, MP_MIS_COS, i) = 1.0f; msti(v, MP_FLAGS, i) = 0;
    mstSet3(v, MP_CV_ACC, i, mk3(0, 0, 0));                  // an unfinished camera path is an invalid vertex with colour 0 (:761-777)
  }
  if (lightTraceDepth > 0) {
    float pick = 1.0f;
    const int lightId = SelectRandomLightFwd(mx(v, 10, i), s, pick);   // RndLightMMLT: group2.z = MMLT_DIM_LGT_N
    LightSampleFwd sam;
    LightSampleForward(s, lightAt(s, lightId), make_float4(mx(v, 4, i), mx(v, 5, i), mx(v, 6, i), mx(v, 7, i)), mx(v, 8, i), sam);
    mpdfFwd(v, 0, i) = sam.pdfA * pick;
    mpdfRev(v, 0, i) = 1.0f;
    mstSet3(v, MP_L_COLOR, i, div3s(sam.color * (1.0f / pick), sam.pdfA * sam.pdfW));
    mst(v, MP_L_COS, i) = sam.cosTheta; mst(v, MP_L_PDF, i) = sam.pdfW;
    lpos = make_float4(sam.pos.x, sam.pos.y, sam.pos.z, 0.0f); ldir = make_float4(sam.dir.x, sam.dir.y, sam.dir.z, 0.0f);
    lbits |= MB_LIGHT_ACTIVE;
  }
  msti(v, MP_S, i) = sp; msti(v, MP_BITS, i) = bits; msti(v, MP_LBITS, i) = lbits; msti(v, MP_X, i) = x; msti(v, MP_Y, i) = y;
  camActive = (bits & MB_CAM_ACTIVE) != 0; lightActive = (lbits & MB_LIGHT_ACTIVE) != 0;
}

// one level of CameraPath (:756-929) for the hit of the ray in rayPos[i]; currDepth counts from 1
// (ray, hit) in, next ray out; returns whether the sub-path goes on
template <int F = HK_FEAT_ALL>
HK_DEV bool mmltCameraStep(const SceneDev& s, const MmltView& v, int i, int currDepth, float4 rayPos4, float4 rayDir4, const HydraLiteHit& hit, float4& npos, float4& ndir) {
  int bits = msti(v, MP_BITS, i);
  mmltDeadRay(npos, ndir);
  if (!(bits & MB_CAM_ACTIVE)) return false;
  const int d = v.depth[i], sp = msti(v, MP_S, i), camTraceDepth = d - sp;
  const bool haveToHitLight = (sp == 0);
  const int prevVertexId = d - currDepth + 1;
  const f3 ray_pos = xyz(rayPos4), ray_dir = xyz(rayDir4);
  bits &= ~MB_CAM_ACTIVE;                                  // every exit but the last one ends the sub-path
  if (HitSome(hit)) {
    const SurfaceHit surf = evalSurface(s, ray_pos, ray_dir, hit);
    const float cosHere = fabsf(dot(ray_dir, surf.normal)), cosPrev = fabsf(mst(v, MP_MIS_COS, i));
    const float misPdf = mst(v, MP_MIS_PDF, i);
    const bool misSpec = (bits & MB_MIS_SPECULAR) != 0;
    const uint32_t flags = uint32_t(msti(v, MP_FLAGS, i));
    float GTerm = 1.0f;
    if (currDepth == 1) {
      f3 cd; float zd;
      const float imageToSurfaceFactor = CameraImageToSurfaceFactor(s, surf.pos, surf.normal, mk2(0, 0), cd, zd);
      mpdfRev(v, d, i) = imageToSurfaceFactor / (g_varsF(s)[HV_F_WIDTH_F] * g_varsF(s)[HV_F_HEIGHT_F]);   // mLightSubPathCount = w * h (:371-372)
      mpdfFwd(v, d, i) = 1.0f;
    } else {
      const float dist = length(ray_pos - surf.pos);
      GTerm = cosHere * cosPrev / fmaxf(dist * dist, HK_DEPSILON2);
    }
    const float* mat = materialAt(s, surf.matId);
    const int lightOffset = (s.hdr[HG_LIGHTS_NUM] != 0) ? s.instLightInstId[hit.instId] : -1;
    const float* pLight = lightAt(s, lightOffset);
    const f3 emission = emissionEval<F>(s, ray_pos, ray_dir, surf, flags, misSpec, pLight, mat);
    const bool splitDL = g_varsI(s)[HV_I_MMLT_FIRST_BOUNCE] > 3;   // m_splitDLByGrammar, Common.cpp:28
    if (dot(emission, emission) > 1e-6f) {
      if (currDepth == camTraceDepth && haveToHitLight && pLight != nullptr) {
        const LightPdfFwd lp = lightPdfFwd(s, pLight, ray_dir, cosHere);
        const float pdfLightWP = lp.pdfW / fmaxf(cosHere, HK_DEPSILON);
        const float pdfMatRevWP = misPdf / fmaxf(cosPrev, HK_DEPSILON);
        mpdfFwd(v, 0, i) = lp.pdfA / float(s.hdr[HG_LIGHTS_NUM]);
        mpdfRev(v, 0, i) = 1.0f;
        mpdfFwd(v, 1, i) = pdfLightWP * GTerm;
        mpdfRev(v, 1, i) = misSpec ? -1.0f * GTerm : pdfMatRevWP * GTerm;
        mstoreHit(v, MP_CV_HIT, i, surf); mstSet3(v, MP_CV_DIR, i, ray_dir); mstSet3(v, MP_CV_ACC, i, emission);
        bits |= MB_CV_VALID;
      }
    } else if (currDepth == camTraceDepth && !haveToHitLight) {
      mstoreHit(v, MP_CV_HIT, i, surf); mstSet3(v, MP_CV_DIR, i, ray_dir); mstSet3(v, MP_CV_ACC, i, mk3(1, 1, 1));
      bits |= MB_CV_VALID;
      if (splitDL && flagsHaveOnlySpecular(flags)) bits |= MB_CV_SPEC_ONLY;
      if (camTraceDepth != 1) {
        const float lastPdfWP = misPdf / fmaxf(cosPrev, HK_DEPSILON);
        mst(v, MP_CV_GTERM, i) = GTerm;
        mpdfRev(v, prevVertexId, i) = misSpec ? -1.0f * GTerm : GTerm * lastPdfWP;
      } else mst(v, MP_CV_GTERM, i) = 1.0f;
    } else if (currDepth < camTraceDepth) {
      float rands[HK_MMLT_PER_BOUNCE];
      mmltRands(v, i, HK_MMLT_HEAD + HK_MMLT_PER_BOUNCE * sp + HK_MMLT_PER_BOUNCE * (currDepth - 1), rands);   // camOffsetInRandArrayMMLT(s) + rndMatOffsetMMLT(bounce)
      MatSample ms;
      MaterialSampleAndEvalBxDF<F>(mat, rands, surf, ray_dir, uint32_t(currDepth - 1) << 8, s, ms, false);
      const float cosNext = fabsf(dot(ms.direction, surf.normal));
      if (currDepth == 1) {
        if (isPureSpecular(ms)) mpdfFwd(v, d, i) = 0.0f;
      } else {
        if (!isPureSpecular(ms)) {
          const float pdfFwdW = materialEval<F>(mat, mmltShadeContext(surf, ray_dir * (-1.0f), ms.direction), s).pdfFwd;
          mpdfFwd(v, prevVertexId, i) = (pdfFwdW / fmaxf(cosHere, HK_DEPSILON)) * GTerm;
        } else mpdfFwd(v, prevVertexId, i) = -1.0f * GTerm;
        const float pdfCamPrevWP = misPdf / fmaxf(cosPrev, HK_DEPSILON);
        mpdfRev(v, prevVertexId, i) = misSpec ? -1.0f * GTerm : pdfCamPrevWP * GTerm;
      }
      const int nf = msti(v, MP_NFAC, i);
      mstSet3(v, MP_PDF + 2 * (v.maxD + 1) + 3 * nf, i, div3s(ms.color * cosNext, fmaxf(ms.pdf, HK_DEPSILON2)));
      if (splitDL && flagsHaveOnlySpecular(flags) && haveToHitLight && currDepth + 1 == camTraceDepth) msti(v, MP_ZERO_FROM, i) = nf;   // stopDL, :917-926
      msti(v, MP_NFAC, i) = nf + 1;
      const f3 np = OffsRayPos(surf.pos, surf.normal, ms.direction);
      npos = make_float4(np.x, np.y, np.z, 0.0f); ndir = make_float4(ms.direction.x, ms.direction.y, ms.direction.z, 0.0f);
      bits = (bits & ~MB_MIS_SPECULAR) | (isPureSpecular(ms) ? MB_MIS_SPECULAR : 0) | MB_CAM_ACTIVE;
      mst(v, MP_MIS_PDF, i) = ms.pdf; mst(v, MP_MIS_COS, i) = dot(ms.direction, surf.normal);
      msti(v, MP_FLAGS, i) = int(flagsNextBounceLite(flags, ms, s));
    }
  }
  msti(v, MP_BITS, i) = bits;
  return (bits & MB_CAM_ACTIVE) != 0;
}

// one level of TraceLightPath (:671-754) for the hit of the ray in rayPos[n + i]
template <int F = HK_FEAT_ALL>
HK_DEV bool mmltLightStep(const SceneDev& s, const MmltView& v, int i, int currDepth, float4 rayPos4, float4 rayDir4, const HydraLiteHit& hit, float4& npos, float4& ndir) {
  int bits = msti(v, MP_LBITS, i);
  mmltDeadRay(npos, ndir);
  if (!(bits & MB_LIGHT_ACTIVE)) return false;
  const int sp = msti(v, MP_S, i), lightTraceDepth = sp - 1;
  const f3 ray_pos = xyz(rayPos4), ray_dir = xyz(rayDir4);
  bits &= ~MB_LIGHT_ACTIVE;
  if (HitSome(hit)) {
    const SurfaceHit surf = evalSurface(s, ray_pos, ray_dir, hit);
    const float prevLightCos = mst(v, MP_L_COS, i), prevPdf = mst(v, MP_L_PDF, i);
    const float cosCurr = fabsf(-dot(ray_dir, surf.normal));
    const float dist = length(surf.pos - ray_pos);
    const float GTermPrev = (prevLightCos * cosCurr / fmaxf(dist * dist, HK_DEPSILON2));
    const float prevPdfWP = prevPdf / fmaxf(prevLightCos, HK_DEPSILON);
    mpdfFwd(v, currDepth, i) = !(bits & MB_LIGHT_SPECULAR) ? prevPdfWP * GTermPrev : -1.0f * GTermPrev;
    const float* mat = materialAt(s, surf.matId);
    float rands[HK_MMLT_PER_BOUNCE];
    mmltRands(v, i, HK_MMLT_HEAD + HK_MMLT_PER_BOUNCE * (currDepth - 1), rands);
    MatSample ms;
    MaterialSampleAndEvalBxDF<F>(mat, rands, surf, ray_dir, uint32_t(currDepth - 1) << 8, s, ms, true);
    const float cosNext = fabsf(+dot(ms.direction, surf.normal));
    if (currDepth == lightTraceDepth) {
      mstoreHit(v, MP_LV_HIT, i, surf); mstSet3(v, MP_LV_DIR, i, ray_dir); mstSet3(v, MP_LV_ACC, i, mst3(v, MP_L_COLOR, i));
      mst(v, MP_LV_GTERM, i) = GTermPrev;
      bits |= MB_LV_VALID;
    } else {
      if (!isPureSpecular(ms)) {
        const float pdfW = materialEval<F>(mat, mmltShadeContext(surf, ray_dir * (-1.0f), ms.direction * (-1.0f)), s).pdfFwd;
        mpdfRev(v, currDepth, i) = (pdfW / fmaxf(cosCurr, HK_DEPSILON)) * GTermPrev;
      } else mpdfRev(v, currDepth, i) = -1.0f * GTermPrev;
      mstSet3(v, MP_L_COLOR, i, mst3(v, MP_L_COLOR, i) * ((ms.color * cosNext) * (1.0f / fmaxf(ms.pdf, HK_DEPSILON2))));
      const f3 np = OffsRayPos(surf.pos, surf.normal, ms.direction);
      npos = make_float4(np.x, np.y, np.z, 0.0f); ndir = make_float4(ms.direction.x, ms.direction.y, ms.direction.z, 0.0f);
      mst(v, MP_L_COS, i) = cosNext; mst(v, MP_L_PDF, i) = ms.pdf;
      bits = (bits & ~MB_LIGHT_SPECULAR) | (isPureSpecular(ms) ? MB_LIGHT_SPECULAR : 0) | MB_LIGHT_ACTIVE;
    }
  }
  msti(v, MP_LBITS, i) = bits;
  return (bits & MB_LIGHT_ACTIVE) != 0;
}

// what mmltConnectBegin and mmltConnectEnd both need of a shadow connection to a sampled light (ConnectShadow :962-990)
struct MmltLightConn { int lightOffset; float pick; ShadowSample sam; f3 dir, pos; };
template <int F = HK_FEAT_ALL>
HK_DEV MmltLightConn mmltLightConnection(const SceneDev& s, const MmltView& v, int i, const SurfaceHit& cvHit) {
  MmltLightConn c;
  c.pick = 1.0f;
  c.lightOffset = SelectRandomLightRev(mx(v, 10, i), s, c.pick);
  if (c.lightOffset >= 0) {
    LightSampleRev<F>(s, lightAt(s, c.lightOffset), mk3(mx(v, 4, i), mx(v, 5, i), mx(v, 6, i)), cvHit.pos, c.sam);
    c.dir = normalize(c.sam.pos - cvHit.pos);
    c.pos = OffsRayPos(cvHit.pos, cvHit.normal, c.dir);
  }
  return c;
}
HK_DEV f3 mmltCameraColor(const MmltView& v, int i) {   // the products CameraPath applies while its recursion returns, deepest level first
  f3 acc = mst3(v, MP_CV_ACC, i);
  const int nf = msti(v, MP_NFAC, i), zeroFrom = msti(v, MP_ZERO_FROM, i);
  if (!(msti(v, MP_BITS, i) & MB_CV_VALID)) acc = mk3(0, 0, 0);
  for (int k = nf - 1; k >= 0; k--) {
    acc = acc * mst3(v, MP_PDF + 2 * (v.maxD + 1) + 3 * k, i);
    if (k == zeroFrom) acc = mk3(0, 0, 0);
  }
  return acc;
}

template <int F = HK_FEAT_ALL>
HK_DEV void mmltConnectBegin(const SceneDev& s, const MmltView& v, int i) {
  const int d = v.depth[i], sp = msti(v, MP_S, i), bits = msti(v, MP_BITS, i) | msti(v, MP_LBITS, i);
  const int lightTraceDepth = sp - 1, camTraceDepth = d - sp;
  float4 epos, edir, spos, sdir;
  mmltDeadRay(epos, edir); mmltDeadRay(spos, sdir);
  if (lightTraceDepth == -1) {
  } else if (camTraceDepth == 0) {
    if (bits & MB_LV_VALID) {
      const SurfaceHit lv = mloadHit(v, MP_LV_HIT, i);
      f3 camDir; float zDepth;
      CameraImageToSurfaceFactor(s, lv.pos, lv.normal, mk2(0, 0), camDir, zDepth);
      const float* mat = materialAt(s, lv.matId);
      float signOfNormal = 1.0f;
      if ((matFlags(mat) & HMF_HAVE_BTDF) != 0 && dot(camDir, lv.normal) < -0.01f) signOfNormal = -1.0f;
      const f3 p = lv.pos + lv.normal * (epsilonOfPos(lv.pos) * signOfNormal);
      epos = make_float4(p.x, p.y, p.z, 0.0f); edir = make_float4(camDir.x, camDir.y, camDir.z, 0.0f);
    }
  } else if (lightTraceDepth == 0) {
    if ((bits & MB_CV_VALID) && !(bits & MB_CV_SPEC_ONLY)) {
      const SurfaceHit cv = mloadHit(v, MP_CV_HIT, i);
      const MmltLightConn c = mmltLightConnection<F>(s, v, i, cv);
      if (c.lightOffset >= 0) { spos = make_float4(c.pos.x, c.pos.y, c.pos.z, c.sam.maxDist * 0.9995f); sdir = make_float4(c.dir.x, c.dir.y, c.dir.z, 0.0f); }
    }
  } else if ((bits & MB_CV_VALID) && (bits & MB_LV_VALID)) {
    const SurfaceHit cv = mloadHit(v, MP_CV_HIT, i), lv = mloadHit(v, MP_LV_HIT, i);
    const f3 diff = cv.pos - lv.pos;
    const float dist2 = fmaxf(dot(diff, diff), HK_DEPSILON2);
    const float dist = sqrtf(dist2);
    const f3 lToC = div3s(diff, dist);
    const float GTerm = (+dot(lv.normal, lToC)) * (-dot(cv.normal, lToC)) / dist2;
    if (!(GTerm < 0.0f)) {
      const f3 p = OffsRayPos(lv.pos, lv.normal, lToC);
      spos = make_float4(p.x, p.y, p.z, dist * 0.9995f); sdir = make_float4(lToC.x, lToC.y, lToC.z, 0.0f);
    }
  }
  v.eyePos[i] = epos; v.eyeDir[i] = edir; v.shPos[i] = spos; v.shDir[i] = sdir;
}

template <int F = HK_FEAT_ALL>
HK_DEV void mmltConnectEnd(const SceneDev& s, const MmltView& v, int i) {
  const int d = v.depth[i], sp = msti(v, MP_S, i), bits = msti(v, MP_BITS, i) | msti(v, MP_LBITS, i);
  const int t = d - sp, lightTraceDepth = sp - 1, camTraceDepth = t;
  const int width = int(g_varsF(s)[HV_F_WIDTH_F]), height = int(g_varsF(s)[HV_F_HEIGHT_F]);
  const float mLightSubPathCount = g_varsF(s)[HV_F_WIDTH_F] * g_varsF(s)[HV_F_HEIGHT_F];
  int x = msti(v, MP_X, i), y = msti(v, MP_Y, i);
  f3 sampleColor = mk3(0, 0, 0);
  const bool cvValid = (bits & MB_CV_VALID) != 0, lvValid = (bits & MB_LV_VALID) != 0;
  const f3 cvAcc = (camTraceDepth > 0) ? mmltCameraColor(v, i) : mk3(1, 1, 1);
  const f3 lvAcc = mst3(v, MP_LV_ACC, i);
  if (lightTraceDepth == -1) sampleColor = cvAcc;
  else if (camTraceDepth == 0) {   // ConnectEye + ConnectEyeP
    if (lvValid) {
      const SurfaceHit lv = mloadHit(v, MP_LV_HIT, i);
      const f3 lvDir = mst3(v, MP_LV_DIR, i);
      f3 camDir; float zDepth;
      const float imageToSurfaceFactor = CameraImageToSurfaceFactor(s, lv.pos, lv.normal, mk2(0, 0), camDir, zDepth);
      const HydraLiteHit hit = v.eyeHit[i];
      if (imageToSurfaceFactor <= 0.0f || (HitSome(hit) && hit.t <= zDepth)) { x = -1; y = -1; }
      else {
        const float surfaceToImageFactor = 1.f / imageToSurfaceFactor;
        const float* mat = materialAt(s, lv.matId);
        const BxDFResult ev = materialEval<F>(mat, mmltShadeContext(lv, camDir, lvDir * (-1.0f)), s, true);
        const f3 colorConnect = ev.brdf + ev.btdf;
        const float pdfRevW = ev.pdfRev;
        const float cosCurr = fabsf(dot(lvDir, lv.normal));
        const float pdfRevWP = pdfRevW / fmaxf(cosCurr, HK_DEPSILON2);
        const float lastG = mst(v, MP_LV_GTERM, i);
        mpdfRev(v, lightTraceDepth, i) = (pdfRevW == 0.0f) ? -1.0f * lastG : pdfRevWP * lastG;
        mpdfFwd(v, lightTraceDepth + 1, i) = 1.0f;
        mpdfRev(v, lightTraceDepth + 1, i) = imageToSurfaceFactor / mLightSubPathCount;
        const f3 sc3 = lvAcc * div3s(colorConnect, mLightSubPathCount * surfaceToImageFactor);
        if (dot(sc3, sc3) > 1e-12f) {
          const f2 scr = worldPosToScreenSpace(s, lv.pos);
          x = int(scr.x); y = int(scr.y);
          sampleColor = sc3;
        }
      }
    }
  } else if (lightTraceDepth == 0) {   // ConnectShadow + ConnectShadowP
    if (cvValid && !(bits & MB_CV_SPEC_ONLY)) {
      f3 explicitColor = mk3(0, 0, 0);
      const SurfaceHit cv = mloadHit(v, MP_CV_HIT, i);
      const f3 cvDir = mst3(v, MP_CV_DIR, i);
      const MmltLightConn c = mmltLightConnection<F>(s, v, i, cv);
      const float shadow = v.shVis[i];
      if (c.lightOffset >= 0 && shadow * shadow * 3.0f > 1e-12f) {
        const float* pLight = lightAt(s, c.lightOffset);
        const float* mat = materialAt(s, cv.matId);
        const BxDFResult ev = materialEval<F>(mat, mmltShadeContext(cv, c.dir, cvDir * (-1.0f)), s);
        const float pdfFwdAt1W = ev.pdfRev;
        const float cosThetaOut1 = fmaxf(+dot(c.dir, cv.normal), HK_DEPSILON), cosThetaOut2 = fmaxf(-dot(c.dir, cv.normal), HK_DEPSILON);
        const bool inverseCos = ((matFlags(mat) & HMF_HAVE_BTDF) != 0 && dot(c.dir, cv.normal) < -0.01f);
        const float cosThetaOut = inverseCos ? cosThetaOut2 : cosThetaOut1;
        const float cosAtLight = fmaxf(c.sam.cosAtLight, HK_DEPSILON);
        const float cosThetaPrev = fmaxf(-dot(cvDir, cv.normal), HK_DEPSILON);
        const f3 brdfVal = (ev.brdf * cosThetaOut1) + (ev.btdf * cosThetaOut2);
        const float pdfRevWP = ev.pdfFwd / fmaxf(cosThetaOut, HK_DEPSILON);
        const float shadowDist = length(cv.pos - c.sam.pos);
        const float GTerm = cosThetaOut * cosAtLight / fmaxf(shadowDist * shadowDist, HK_DEPSILON2);
        const LightPdfFwd lp = lightPdfFwd(s, pLight, c.dir, cosAtLight);
        mpdfFwd(v, 0, i) = lp.pdfA * c.pick;
        mpdfRev(v, 0, i) = 1.0f;
        mpdfFwd(v, 1, i) = (lp.pdfW / cosAtLight) * GTerm;
        mpdfRev(v, 1, i) = (ev.pdfFwd == 0) ? -1.0f * GTerm : pdfRevWP * GTerm;
        const float lastG = mst(v, MP_CV_GTERM, i);
        if (t > 1) mpdfFwd(v, 2, i) = (pdfFwdAt1W == 0.0f) ? -1.0f * lastG : (pdfFwdAt1W / cosThetaPrev) * lastG;
        float envMisMult = 1.0f;
        if (as_int(pLight[HL_TYPE]) == HLT_SKY_DOME) envMisMult = misWeightHeuristic(c.sam.pdf * c.pick, ev.pdfFwd);
        const float explicitPdfW = fmaxf(c.sam.pdf, HK_DEPSILON2);
        explicitColor = div3s(((c.sam.color * envMisMult) * (1.0f / c.pick)) * brdfVal, explicitPdfW) * shadow;
      }
      sampleColor = cvAcc * explicitColor;
    }
  } else if (cvValid) {   // ConnectEndPoints + ConnectEndPointsP
    f3 explicitColor = mk3(0, 0, 0);
    const float shadow = v.shVis[i];
    if (lvValid) {
      const SurfaceHit cv = mloadHit(v, MP_CV_HIT, i), lv = mloadHit(v, MP_LV_HIT, i);
      const f3 cvDir = mst3(v, MP_CV_DIR, i), lvDir = mst3(v, MP_LV_DIR, i);
      const f3 diff = cv.pos - lv.pos;
      const float dist2 = fmaxf(dot(diff, diff), HK_DEPSILON2);
      const float dist = sqrtf(dist2);
      const f3 lToC = div3s(diff, dist);
      const float GTerm0 = (+dot(lv.normal, lToC)) * (-dot(cv.normal, lToC)) / dist2;
      if (!(GTerm0 < 0.0f) && !(shadow * shadow * 3.0f < 1e-12f)) {
        const float* matL = materialAt(s, lv.matId);
        const BxDFResult evL = materialEval<F>(matL, mmltShadeContext(lv, lToC, lvDir * (-1.0f)), s, true);
        const f3 lightBRDF = evL.brdf + evL.btdf;
        float signOfNormalL = 1.0f, signOfNormalC = 1.0f;
        if ((matFlags(matL) & HMF_HAVE_BTDF) != 0 && dot(lToC, lv.normal) < -0.01f) signOfNormalL = -1.0f;
        const float* matC = materialAt(s, cv.matId);
        const BxDFResult evC = materialEval<F>(matC, mmltShadeContext(cv, lToC * (-1.0f), cvDir * (-1.0f)), s);
        const f3 camBRDF = evC.brdf + evC.btdf;
        const float camVPdfRevW = evC.pdfFwd, camVPdfFwdW = evC.pdfRev;
        if ((matFlags(matC) & HMF_HAVE_BTDF) != 0 && dot(lToC * (-1.0f), cv.normal) < -0.01f) signOfNormalC = -1.0f;
        const float cosAtLightVertex = +signOfNormalL * dot(lv.normal, lToC), cosAtCameraVertex = -signOfNormalC * dot(cv.normal, lToC);
        const float cosAtLightVertexPrev = -dot(lv.normal, lvDir), cosAtCameraVertexPrev = -dot(cv.normal, cvDir);
        const float GTerm = cosAtLightVertex * cosAtCameraVertex / dist2;
        if (!(GTerm < 0.0f)) {
          const float lightPdfFwdWP = evL.pdfFwd / fmaxf(cosAtLightVertex, HK_DEPSILON2);
          const float cameraPdfRevWP = camVPdfRevW / fmaxf(cosAtCameraVertex, HK_DEPSILON2);
          mpdfFwd(v, sp, i) = (lightPdfFwdWP == 0.0f) ? -1.0f * GTerm : lightPdfFwdWP * GTerm;
          mpdfRev(v, sp, i) = (cameraPdfRevWP == 0.0f) ? -1.0f * GTerm : cameraPdfRevWP * GTerm;
          const float lastGL = mst(v, MP_LV_GTERM, i), lastGC = mst(v, MP_CV_GTERM, i);
          mpdfRev(v, sp - 1, i) = (evL.pdfRev == 0.0f) ? -1.0f * lastGL : lastGL * (evL.pdfRev / fmaxf(cosAtLightVertexPrev, HK_DEPSILON));
          if (d > 3) mpdfFwd(v, sp + 1, i) = (camVPdfFwdW == 0.0f) ? -1.0f * lastGC : lastGC * (camVPdfFwdW / fmaxf(cosAtCameraVertexPrev, HK_DEPSILON));
          const bool fwdCanNotBeEvaluated = (lightPdfFwdWP < HK_DEPSILON2) || (d > 3 && camVPdfFwdW < HK_DEPSILON2);
          const bool revCanNotBeEvaluated = (cameraPdfRevWP < HK_DEPSILON2) || (evL.pdfRev < HK_DEPSILON2);
          if (!(fwdCanNotBeEvaluated && revCanNotBeEvaluated)) explicitColor = ((lightBRDF * camBRDF) * GTerm) * shadow;
        }
      }
    }
    sampleColor = (cvAcc * explicitColor) * lvAcc;
  }
  // (4) MIS weight, :252-285
  float misWeight = 1.0f;
  if (dot(sampleColor, sampleColor) > 1e-12f) {
    float pdfThisWay = 1.0f, pdfSumm = 0.0f;
    for (int split = 0; split <= d; split++) {
      const bool specularMet = (split > 0) && (split < d) && (mpdfRev(v, split, i) < 0.0f || mpdfFwd(v, split, i) < 0.0f);
      float pdfOtherWay = specularMet ? 0.0f : 1.0f;
      if (split == d) pdfOtherWay = misHeuristicPower1(mpdfFwd(v, d, i));
      for (int k = 0; k < split; k++) pdfOtherWay *= misHeuristicPower1(mpdfFwd(v, k, i));
      for (int k = split + 1; k <= d; k++) pdfOtherWay *= misHeuristicPower1(mpdfRev(v, k, i));
      if (split == sp) pdfThisWay = pdfOtherWay;
      pdfSumm += pdfOtherWay;
    }
    misWeight = pdfThisWay / fmaxf(pdfSumm, HK_DEPSILON2);
  }
  sampleColor = sampleColor * misWeight;
  if (!(x >= 0 && x < width && y >= 0 && y < height)) { x = 0; y = 0; sampleColor = mk3(0, 0, 0); }
  float* o = v.out8 + size_t(i) * 8;
  o[0] = sampleColor.x; o[1] = sampleColor.y; o[2] = sampleColor.z; o[3] = float(x); o[4] = float(y); o[5] = float(sp);
  o[6] = misWeight; o[7] = fmaxf(0.33334f * (sampleColor.x + sampleColor.y + sampleColor.z), 0.0f);   // contribFunc, cglobals.h:1929-1932
}

// ================================================================================================ the Markov chains of IntegratorMMLT
// DoPassIndirectMLT (CPUExp_Integrators_MMLT.cpp:358-461) with one chain per thread instead of one per OpenMP thread: InitialSamplePS :51-57,
// MutatePrimarySpace :93-144 (MutateLightPart :59-73, MutateCameraPart :75-90) and the accept / contribute step :376-447.  The reference
// stirs its generators with clock() (:97-103, :362-368); chains here are functions of their seed alone.
enum { CH_Y = 0, CH_COLOR = 1, CH_XS = 4, CH_YS = 5, CH_GEN = 6, CH_GEN2 = 8, CH_ACCEPTED = 10, CH_PLANES = 11 };
#define HK_MUTATE_COEFF_SCREEN 128.0f   // crandom.h:219-220
#define HK_MUTATE_COEFF_BSDF 64.0f
struct MmltChains {
  int n, maxD;
  float* ch;          // CH_PLANES planes of n
  const int* depth;   // d per chain
  float* xCur;        // mmltStride(maxD) planes of n: the current state of every chain
  float* xNew;        // ... and the proposal
};
HK_DEV float& mch(const MmltChains& c, int plane, int i) { return c.ch[size_t(plane) * c.n + i]; }
HK_DEV RandomGen mchGen(const MmltChains& c, int plane, int i) { RandomGen g; g.x = uint32_t(as_int(mch(c, plane, i))); g.y = uint32_t(as_int(mch(c, plane + 1, i))); return g; }
HK_DEV void mchSetGen(const MmltChains& c, int plane, int i, RandomGen g) { mch(c, plane, i) = as_float(int(g.x)); mch(c, plane + 1, i) = as_float(int(g.y)); }
// chain i draws from RandomGenInit(seed + 2i) for samples and mutations (PerThread().gen) and RandomGenInit(seed + 2i + 1) for the accept test (gen2)
HK_DEV void mmltInitChain(const MmltChains& c, int i, int seed) {
  mchSetGen(c, CH_GEN, i, RandomGenInit(seed + 2 * i));
  mchSetGen(c, CH_GEN2, i, RandomGenInit(seed + 2 * i + 1));
  mch(c, CH_Y, i) = 0.0f; mch(c, CH_COLOR, i) = 0.0f; mch(c, CH_COLOR + 1, i) = 0.0f; mch(c, CH_COLOR + 2, i) = 0.0f;
  mch(c, CH_XS, i) = 0.0f; mch(c, CH_YS, i) = 0.0f; mch(c, CH_ACCEPTED, i) = 0.0f;
}
HK_DEV void mmltFreshSample(const MmltChains& c, int i, float* x) {   // InitialSamplePS: every dimension from the chain's generator, in order
  RandomGen g = mchGen(c, CH_GEN, i);
  const int size = mmltStride(c.depth[i]);
  for (int j = 0; j < size; j++) x[size_t(j) * c.n + i] = rndFloat1_Pseudo(g);
  mchSetGen(c, CH_GEN, i, g);
}
HK_DEV f2 rndFloat2_Pseudo(RandomGen& g) { f2 r; r.x = rndFloat1_Pseudo(g); r.y = rndFloat1_Pseudo(g); return r; }   // crandom.h:158-164
HK_DEV void mmltMutate(const MmltChains& c, int i) {   // xCur -> xNew
  RandomGen g = mchGen(c, CH_GEN, i);
  const int d = c.depth[i], size = mmltStride(d), n = c.n;
  const float plarge = 0.33f, plight = 0.20f, pmultiChain = 0.16f;
  const float selector = rndFloat1_Pseudo(g);
  if (selector < plarge) {
    for (int j = 0; j < size; j++) c.xNew[size_t(j) * n + i] = rndFloat1_Pseudo(g);
  } else {
    // every dimension is read from xCur and written to xNew once; the generator is drawn from in the reference's order: light head 4..9, light
    // bounces, lens 0..3, camera bounces (MutatePrimarySpace, CPUExp_Integrators_MMLT.cpp:540-600).  Dimensions 10 and 11 (light choice, split) only move in a large step
    const int currSplit = mapRndFloatToInt(c.xCur[size_t(11) * n + i], 0, d);
    const int camBegin = mmltStride(currSplit);   // camOffsetInRandArrayMMLT
    const bool lightPart = (plarge < selector && selector <= plarge + plight + pmultiChain);
    const bool cameraPart = !(plarge < selector && selector <= plarge + plight);
    for (int j = 4; j < 10; j++) {
      float v = c.xCur[size_t(j) * n + i];
      if (lightPart) v = MutateKelemen(v, rndFloat2_Pseudo(g), HK_MUTATE_COEFF_BSDF, 1024.0f);
      c.xNew[size_t(j) * n + i] = v;
    }
    for (int j = HK_MMLT_HEAD; j < camBegin; j++) {
      float v = c.xCur[size_t(j) * n + i];
      if (lightPart) v = MutateKelemen(v, rndFloat2_Pseudo(g), HK_MUTATE_COEFF_BSDF, 1024.0f);
      c.xNew[size_t(j) * n + i] = v;
    }
    for (int j = 0; j < 4; j++) {
      float v = c.xCur[size_t(j) * n + i];
      if (cameraPart) v = MutateKelemen(v, rndFloat2_Pseudo(g), j < 2 ? HK_MUTATE_COEFF_SCREEN * 1.0f : HK_MUTATE_COEFF_BSDF, 1024.0f);
      c.xNew[size_t(j) * n + i] = v;
    }
    for (int j = camBegin; j < size; j++) {
      float v = c.xCur[size_t(j) * n + i];
      if (cameraPart) v = MutateKelemen(v, rndFloat2_Pseudo(g), HK_MUTATE_COEFF_BSDF, 1024.0f);
      c.xNew[size_t(j) * n + i] = v;
    }
    c.xNew[size_t(10) * n + i] = c.xCur[size_t(10) * n + i];
    c.xNew[size_t(11) * n + i] = c.xCur[size_t(11) * n + i];
  }
  mchSetGen(c, CH_GEN, i, g);
}
// the chain takes the state xNew whose F is out8 (first call after a fresh sample: no test, no contribution)
HK_DEV void mmltSeedChain(const MmltChains& c, int i, const float* out8) {
  const float* o = out8 + size_t(i) * 8;
  mch(c, CH_Y, i) = o[7]; mch(c, CH_COLOR, i) = o[0]; mch(c, CH_COLOR + 1, i) = o[1]; mch(c, CH_COLOR + 2, i) = o[2];
  mch(c, CH_XS, i) = o[3]; mch(c, CH_YS, i) = o[4];
}
// accept / reject and the two expected-value contributions (:388-447); image4 = float4 per pixel, row-major, w wide
HK_DEV void mmltAcceptReject(const MmltChains& c, int i, const float* out8, float bkScale, float* image4, int w) {
  const float* o = out8 + size_t(i) * 8;
  const f3 yNewColor = mk3(o[0], o[1], o[2]);
  const float yNew = o[7];
  const int xScrNew = int(o[3]), yScrNew = int(o[4]);
  const float yOld = mch(c, CH_Y, i);
  const f3 yOldColor = mk3(mch(c, CH_COLOR, i), mch(c, CH_COLOR + 1, i), mch(c, CH_COLOR + 2, i));
  const int xScrOld = int(mch(c, CH_XS, i)), yScrOld = int(mch(c, CH_YS, i));
  const float a = (yOld == 0.0f) ? 1.0f : fminf(1.0f, yNew / yOld);
  RandomGen g2 = mchGen(c, CH_GEN2, i);
  const float p = rndFloat1_Pseudo(g2);
  mchSetGen(c, CH_GEN2, i, g2);
  if (p <= a) {
    const int size = mmltStride(c.depth[i]);
    for (int j = 0; j < size; j++) c.xCur[size_t(j) * c.n + i] = c.xNew[size_t(j) * c.n + i];
    mmltSeedChain(c, i, out8);
    mch(c, CH_ACCEPTED, i) += 1.0f;
  }
  const f3 contribAtX = (yOldColor * bkScale) * (1.0f / fmaxf(yOld, 1e-6f)) * (1.0f - a);
  const f3 contribAtY = (yNewColor * bkScale) * (1.0f / fmaxf(yNew, 1e-6f)) * a;
  if (dot(contribAtX, contribAtX) > 1e-12f) {
    float* px = image4 + 4 * size_t(yScrOld * w + xScrOld);
    hk_atomic_add(px + 0, contribAtX.x); hk_atomic_add(px + 1, contribAtX.y); hk_atomic_add(px + 2, contribAtX.z); hk_atomic_add(px + 3, 1.0f - a);
  }
  if (dot(contribAtY, contribAtY) > 1e-12f) {
    float* px = image4 + 4 * size_t(yScrNew * w + xScrNew);
    hk_atomic_add(px + 0, contribAtY.x); hk_atomic_add(px + 1, contribAtY.y); hk_atomic_add(px + 2, contribAtY.z); hk_atomic_add(px + 3, a);
  }
}

// ================================================================================================ IntegratorSBDPT::DoPass through F
// The stochastic-connection bidirectional pass (hydra_drv/CPUExp_Integrators_SBDPT.cpp:11-216): every sample picks a path length d uniformly in
// 2..maxDepth (:21) and a split uniformly in 0..d (:22), builds the two sub-paths, connects, weights by MIS over the splits and by the selector's
// 1/pdf = (d + 1)(maxDepth - 1) (:24), and splats.  Its sub-path, connection and MIS code is IntegratorMMLT's (the reference keeps two copies),
// so a sample here is F of a fresh primary-sample vector: the split comes from x[MMLT_DIM_SPLIT] and the pixel from the lens dimensions instead of
// rndInt draws (:22, :40-41) -- the same estimator with its random numbers laid out as MMLT's.
HK_DEV int rndIntFromFloat(float r, int a, int b) {   // crandom.h:594-606: integers a .. b-1
  const int res = int(float(a) + r * (float(b) - float(a)));
  return (res > b - 1) ? b - 1 : res;
}
HK_DEV void sbdptPickDepth(const MmltChains& c, int i, int* depth, int maxDepth) {
  RandomGen g = mchGen(c, CH_GEN, i);
  depth[i] = rndIntFromFloat(rndFloat1_Pseudo(g), 2, maxDepth + 1);
  mchSetGen(c, CH_GEN, i, g);
}
HK_DEV void sbdptSplat(int i, const int* depth, int maxDepth, const float* out8, float* image4, int w) {
  const float* o = out8 + size_t(i) * 8;
  const float selectorInvPdf = float((depth[i] + 1) * (maxDepth - 1));
  const f3 c = mk3(o[0], o[1], o[2]) * selectorInvPdf;
  if (dot(c, c) > 1e-20f) {
    float* px = image4 + 4 * size_t(int(o[4]) * w + int(o[3]));
    hk_atomic_add(px + 0, c.x); hk_atomic_add(px + 1, c.y); hk_atomic_add(px + 2, c.z);
  }
}
