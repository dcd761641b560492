// hk_gbuffer.h -- the G-buffer of IHWLayer::EvalGBuffer: per pixel 64 Hammersley-placed primary rays, one surface sample each, the sample
// most similar to all others wins and carries the share of samples like it as coverage.
//
// Behaviour contract: IntegratorCommon::gbufferEval / gbufferSample (hydra_drv/CPUExp_GBuffer.cpp:15-113), which restates on the CPU what
// GPUOCLLayer::EvalGBuffer (GPUOCLLayerOther.cpp:694-870) runs as kernels, with the CPU form's alpha (0 on every hit, its "#TODO: eval alpha").
// Types and helpers: GBuffer1/GBuffer2/GBufferAll, initGBufferAll, packGBuffer1/2, projectedPixelSize, surfaceSimilarity, gbuffDiff
// (cglobals.h:2057-2205), encodeNormal (:1401-1411), RealColorToUint32 (:711-724), PlaneHammersley (globals_sys.cpp:45-61),
// materialEvalDiffuse / materialLeafEvalDiffuse (cmaterial.h:2830-2916).
#pragma once
#include "hk_bidir.h"

#define HK_GBUFFER_SAMPLES 64   // GBUFFER_SAMPLES, cglobals.h:2095 -- one wavefront per pixel on the device

struct GBufferSample {          // GBufferAll, cglobals.h:2057-2080
  float depth; f3 norm; float4 rgba; int matId; float coverage;
  f2 texCoord; int objId, instId;
};
HK_DEV void initGBufferSample(GBufferSample& g) {   // initGBufferAll, cglobals.h:2082-2093
  g.depth = 1e+6f; g.norm = mk3(0, 0, 0); g.rgba = make_float4(0, 0, 0, 1); g.matId = -1; g.coverage = 0.0f;
  g.texCoord = mk2(0, 0); g.objId = -1; g.instId = -1;
}
HK_DEV f2 planeHammersley(int k, int n) {   // globals_sys.cpp:45-61: sums of powers of two and one division, exact in float
  float u = 0.0f;
  int kk = k;
  for (float p = 0.5f; kk; p *= 0.5f, kk >>= 1)
    if (kk & 1) u += p;
  return mk2(u, (float(k) + 0.5f) / float(n));
}
// the ray of sample k of pixel (x, y), CPUExp_GBuffer.cpp:31-45 -- both scale factors are 1 / width there, and here
HK_DEV void gbufferEyeRay(const SceneDev& s, int x, int y, int k, int width, f3& ray_pos, f3& ray_dir) {
  const f2 qmc = planeHammersley(k, HK_GBUFFER_SAMPLES);
  const float sizeInvX = 1.0f / float(width), sizeInvY = 1.0f / float(width);
  const float4 lensOffs = make_float4(sizeInvX * (qmc.x + float(x)), sizeInvY * (qmc.y + float(y)), 0.0f, 0.0f);
  float fx, fy;
  MakeEyeRayFromF4Rnd(lensOffs, s, ray_pos, ray_dir, fx, fy);
}
HK_DEV f3 materialLeafEvalDiffuse(const float* m, f2 tc, const SceneDev& s) {   // cmaterial.h:2830-2848
  const int type = matType(m);
  if (type == HMT_LAMBERT || type == HMT_OREN_NAYAR)   // both keep colour and sampler at the same offsets (cmaterial.h:200-218, 264-284)
    return sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s) * mk3(m[HM_COLOR], m[HM_COLOR + 1], m[HM_COLOR + 2]);
  return mk3(0, 0, 0);
}
HK_DEV f3 materialEvalDiffuse(const float* a_m, f3 l, f3 n, f2 tc, const SceneDev& s) {   // cmaterial.h:2853-2916
  f3 val = mk3(0, 0, 0);
  float stackW[7]; int stackO[7];   // MIX_TREE_MAX_DEEP
  int top = 0, currOffset = 0;
  float currW = 1.0f;
  do {
    if (top > 0) { top--; currOffset = stackO[top]; currW = stackW[top]; }
    const float* m = a_m + size_t(currOffset) * HM_NODE_FLOATS;
    if (matType(m) == HMT_BLEND_MASK) {
      const float alpha = blendMaskAlpha2(m, l, n, tc, s);
      const int o1 = as_int(m[HM_BLEND_MAT1]), o2 = as_int(m[HM_BLEND_MAT2]);
      float w1 = alpha;
      const float w2 = 1.0f - alpha;
      if ((as_int(m[HM_BLEND_FLAGS]) & HBF_REFLECTION_WEIGHT_IS_ONE) && matType(m + size_t(o1) * HM_NODE_FLOATS) != HMT_BLEND_MASK) w1 = 1.0f;
      if (top < 7) { stackW[top] = currW * w1; stackO[top] = currOffset + o1; top++; }
      if (top < 7) { stackW[top] = currW * w2; stackO[top] = currOffset + o2; top++; }
    } else
      val = val + (materialLeafEvalDiffuse(m, tc, s) * currW);
  } while (top > 0);
  return val;
}
// gbufferSample, CPUExp_GBuffer.cpp:84-113; `hit` with the reference's geomId (no class label)
HK_DEV GBufferSample gbufferSampleOf(const SceneDev& s, f3 ray_pos, f3 ray_dir, const HydraLiteHit& hit) {
  GBufferSample g;
  initGBufferSample(g);
  if (!HitSome(hit)) { g.rgba = make_float4(0, 0, 0, 1); return g; }
  const SurfaceHit surf = evalSurface(s, ray_pos, ray_dir, hit);
  g.depth = hit.t;
  g.norm = surf.normal;
  g.rgba = mk4(materialEvalDiffuse(materialAt(s, surf.matId), ray_dir, surf.normal, surf.texCoord, s), 0.0f);
  g.matId = surf.matId;
  g.coverage = 1.0f;
  g.texCoord = surf.texCoord;
  g.objId = hit.geomId;
  g.instId = hit.instId;
  return g;
}
HK_DEV float projectedPixelSize(float dist, float FOV, float w, float h) {   // cglobals.h:2156-2165
  const float ppx = (FOV / w) * dist, ppy = (FOV / h) * dist;
  return (dist > 0.0f) ? 2.0f * fmaxf(ppx, ppy) : 1000.0f;
}
HK_DEV float surfaceSimilarity(f3 n1, float d1, f3 n2, float d2, const float MADXDIFF) {   // cglobals.h:2167-2191
  const float MANXDIFF = 0.15f;
  const float dist = length(n1 - n2);
  if (dist >= MANXDIFF) return 0.0f;
  if (fabsf(d1 - d2) >= MADXDIFF) return 0.0f;
  const float normalSimilar = sqrtf(1.0f - (dist / MANXDIFF));
  const float depthSimilar = sqrtf(1.0f - fabsf(d1 - d2) / MADXDIFF);
  return normalSimilar * depthSimilar;
}
// what gbuffDiff reads of a sample (cglobals.h:2193-2205)
struct GBufferKey { f3 norm; float depth; int instId, objId, matId; float alpha; };
HK_DEV GBufferKey gbufferKey(const GBufferSample& g) { GBufferKey k; k.norm = g.norm; k.depth = g.depth; k.instId = g.instId; k.objId = g.objId; k.matId = g.matId; k.alpha = g.rgba.w; return k; }
HK_DEV float gbuffDiff(const GBufferKey& s1, const GBufferKey& s2, const float a_fov, float w, float h) {
  const float ppSize = projectedPixelSize(s1.depth, a_fov, w, h);
  const float surfaceSimilar = surfaceSimilarity(s1.norm, s1.depth, s2.norm, s2.depth, ppSize * 2.0f);
  const float surfaceDiff = 1.0f - surfaceSimilar;
  const float objDiff = (s1.instId == s2.instId && s1.objId == s2.objId) ? 0.0f : 1.0f;
  const float matDiff = (s1.matId == s2.matId) ? 0.0f : 1.0f;
  const float alphaDiff = fabsf(s1.alpha - s2.alpha);
  return surfaceDiff + objDiff + matDiff + alphaDiff;
}
#define HK_GBUFFER_FOV ((HK_PI / 180.f) * 90.0f)   /* DEG_TO_RAD * 90, cglobals.h:65, CPUExp_GBuffer.cpp:17 */
// steps (3)-(4) of gbufferEval for one pixel, serial form (host emulation and the reading aid for the wave form in hydra_hip.hip):
// the winner is the FIRST sample with the smallest summed difference; it leaves with its own coverage
HK_DEV GBufferSample gbufferResolve(GBufferSample* samples, float width, float height) {
  float minDiff = 100000000.0f;
  int minDiffId = 0;
  for (int i = 0; i < HK_GBUFFER_SAMPLES; i++) {
    float diff = 0.0f, coverage = 0.0f;
    const GBufferKey ki = gbufferKey(samples[i]);
    for (int j = 0; j < HK_GBUFFER_SAMPLES; j++) {
      const float thisDiff = gbuffDiff(ki, gbufferKey(samples[j]), HK_GBUFFER_FOV, width, height);
      diff += thisDiff;
      if (thisDiff < 1.0f) coverage += 1.0f;
    }
    coverage *= (1.0f / float(HK_GBUFFER_SAMPLES));
    samples[i].coverage = coverage;
    if (diff < minDiff) { minDiff = diff; minDiffId = i; }
  }
  return samples[minDiffId];
}
HK_DEV uint32_t encodeNormal(f3 n) {   // cglobals.h:1401-1411
  const int x = int(n.x * 32767.0f), y = int(n.y * 32767.0f);
  const uint32_t sign = (n.z >= 0) ? 0u : 1u;
  const uint32_t sx = (uint32_t(x & 0xfffe) | sign), sy = (uint32_t(y & 0xffff) << 16);
  return sx | sy;
}
// RealColorToUint32, cglobals.h:711-724.  `(unsigned char)r` on a float outside 0..255 is undefined in C; the reference's CPU build
// converts to a 32-bit integer and keeps the low byte, which is what is written here (diffuse colours are 0..1 in every fixture)
HK_DEV uint32_t RealColorToUint32(float4 c) {
  const uint32_t r = uint32_t(int(c.x * 255.0f)) & 255u, g = uint32_t(int(c.y * 255.0f)) & 255u;
  const uint32_t b = uint32_t(int(c.z * 255.0f)) & 255u, a = uint32_t(int(c.w * 255.0f)) & 255u;
  return r | (g << 8) | (b << 16) | (a << 24);
}
HK_DEV float4 packGBuffer1(const GBufferSample& g) {   // cglobals.h:2098-2114
  const float clampedCoverage = fminf(fmaxf(g.coverage * 255.0f, 0.0f), 255.0f);
  const int compressedCoverage = int(uint32_t(int(clampedCoverage)) << 24);
  const int packedMIdAncCov = (g.matId & 0x00FFFFFF) | (compressedCoverage & int(0xFF000000u));
  return make_float4(g.depth, as_float(int(encodeNormal(g.norm))), as_float(packedMIdAncCov), as_float(int(RealColorToUint32(g.rgba))));
}
HK_DEV float4 packGBuffer2(const GBufferSample& g) {   // cglobals.h:2137-2145
  return make_float4(g.texCoord.x, g.texCoord.y, as_float(g.objId), as_float(g.instId));
}
