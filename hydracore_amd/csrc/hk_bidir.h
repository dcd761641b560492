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
// clight.h:654-719 (no IES, no sky portal: both are refused at upload)
HK_DEV void AreaLightSampleForward(const float* L, float4 rands, LightSampleFwd& out) {
  const float offsetX = rands.x * 2.0f - 1.0f, offsetY = rands.y * 2.0f - 1.0f;
  f3 samplePos = mk3(offsetX * L[HL_AREA_SIZE_X], 0.0f, offsetY * L[HL_AREA_SIZE_Y]);
  if (as_int(L[HL_AREA_IS_DISK]) != 0) {
    const f2 d = MapSamplesToDisc(mk2(offsetX, offsetY));
    samplePos = mk3(d.x * L[HL_AREA_SIZE_X], 0.0f, d.y * L[HL_AREA_SIZE_X]);
  }
  samplePos = lightMatrixMul(L + HL_AREA_MATRIX, samplePos) + lightPos(L);
  const f3 lnorm = lightNorm(L);
  f3 sampleDir = MapSampleToCosineDistribution(rands.z, rands.w, lnorm, lnorm, 1.0f);
  float cosTheta = fmaxf(dot(sampleDir, lnorm), 0.0f);
  float pdfW = cosTheta * HK_INV_PI;
  if (as_int(L[HL_AREA_SPOT_DISTR]) != 0) {
    const float cos2 = L[HL_AREA_SPOT_COS2];
    sampleDir = MapSamplesToCone(cos2, mk2(rands.z, rands.w), lnorm);
    pdfW = 1.0f / (2.0f * HK_PI * (1.0f - cos2));
  }
  cosTheta = fmaxf(dot(sampleDir, lnorm), 0.0f);
  const f3 color = areaDiffuseLightGetIntensity(L, sampleDir * (-1.0f), false);
  out.isPoint = false;
  out.pos = samplePos + lnorm * epsilonOfPos(samplePos);
  out.dir = sampleDir;
  out.color = color * cosTheta;
  out.pdfA = 1.0f / L[HL_SURFACE_AREA];
  out.pdfW = pdfW;
  out.cosTheta = cosTheta;
  out.norm = lnorm;
}
HK_DEV void PointLightSampleForward(const float* L, float4 rands, LightSampleFwd& out) {   // clight.h:838-862, no IES
  const f3 sampleDir = UniformSampleSphere(rands.x, rands.y);
  const f3 samplePos = lightPos(L);
  out.isPoint = true;
  out.pos = samplePos + sampleDir * epsilonOfPos(samplePos);
  out.dir = sampleDir;
  out.color = lightColor(L) * (1.0f / L[HL_SURFACE_AREA]);   // pointLightGetIntensity without IES = the base colour
  out.pdfA = 1.0f / L[HL_SURFACE_AREA];
  out.pdfW = HK_INV_PI * 0.25f;
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
HK_DEV void LightSampleForward(const float* L, float4 rands, LightSampleFwd& out) {
  switch (as_int(L[HL_TYPE])) {
    case HLT_DIRECT: DirectLightSampleForward(L, rands, out); break;
    case HLT_POINT_SPOT: PointSpotSampleForward(L, rands, out); break;
    case HLT_POINT_OMNI: PointLightSampleForward(L, rands, out); break;
    default: AreaLightSampleForward(L, rands, out); break;
  }
}
// lightPdfFwd, clight.h:1117-1175 (no IES)
HK_DEV LightPdfFwd lightPdfFwd(const float* L, float cosTheta) {
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
  if (ltype == HLT_AREA && as_int(L[HL_AREA_SPOT_DISTR]) != 0) {
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
