/* ref_driver.cl -- kernel entry points (ours) around the REFERENCE's own inline functions.
 *
 * TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes the reference headers where they lie
 * (/root/reference/hydra_drv/c*.h, their OpenCL branch, -D OCL_COMPILER) and calls their functions, the same way the
 * reference's shaders/*.cl do.  oracle/build_ref.sh compiles it with the image's clang for gfx950 into
 * oracle/_ref/ref_driver.hsaco (git-ignored; travels to the GPU box as a built binary, the sources do not).
 * tests/golden/make_golden.py launches these kernels through the HIP module API and stores the outputs as golden
 * vectors that pin the CPU oracle.
 *
 * The per-path loop below mirrors IntegratorMISPTLoop2::PathTrace (hydra_drv/CPUExp_Integrators_PT_Loop.cpp:264-321,
 * C++ class code that cannot be compiled as OpenCL) stage by stage; every stage body is a call into reference code.
 */
#include "cglobals.h"
#include "cfetch.h"
#include "crandom.h"
#include "ctrace.h"
#include "cmaterial.h"
#include "clight.h"
#include "cbidir.h"

__kernel void ref_random(__global const int* seeds, int draws, __global float4* out4, __global uint2* state2, int n)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  RandomGen gen = RandomGenInit(seeds[i]);
  for (int d = 0; d < draws; d++)
    out4[i * draws + d] = rndFloat4_Pseudo(&gen);
  state2[i] = gen.state;
}

__kernel void ref_make_eye_rays(__global const int2* xy, __global const float4* offs, __global const EngineGlobals* a_globals,
                                int w, int h, __global float4* pos4, __global float4* dir4, int n)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  float3 p, d;
  MakeRandEyeRay(xy[i].x, xy[i].y, w, h, offs[i], a_globals, &p, &d);
  pos4[i] = to_float4(p, 0.0f);
  dir4[i] = to_float4(d, 0.0f);
}

/* IntegratorCommon::rayTrace, CPUExp_Integrators_Common.cpp:122-154: one running Lite_Hit over the scene's trees (two here), the alpha
 * form on an instanced tree that has an alpha table.  Pointers of an absent tree / table are 0. */
typedef struct RefTreesT
{
  __global const float4* bvh0; __global const float4* tris0; __global const uint2* alpha0; int haveInst0;
  __global const float4* bvh1; __global const float4* tris1; __global const uint2* alpha1; int haveInst1;
  __global const int4* texStorage; __global const EngineGlobals* globals;
} RefTrees;

static inline Lite_Hit ref_traceOneTree(float3 pos, float3 dir, Lite_Hit hit, __global const float4* bvh, __global const float4* tris, __global const uint2* alpha, int haveInst,
                                        __global const int4* texStorage, __global const EngineGlobals* globals)
{
  if (haveInst)
  {
    if (alpha != 0)
      hit = BVH4InstTraverseAlpha(pos, dir, 0.0f, hit, bvh, tris, alpha, texStorage, globals);
    else
      hit = BVH4InstTraverse(pos, dir, 0.0f, hit, bvh, tris);
  }
  else
    hit = BVH4Traverse(pos, dir, 0.0f, hit, bvh, tris);
  return hit;
}

static inline Lite_Hit ref_rayTrace(float3 pos, float3 dir, const RefTrees t)
{
  Lite_Hit hit = Make_Lite_Hit(MAXFLOAT, -1);
  hit = ref_traceOneTree(pos, dir, hit, t.bvh0, t.tris0, t.alpha0, t.haveInst0, t.texStorage, t.globals);
  if (t.bvh1 != 0)
    hit = ref_traceOneTree(pos, dir, hit, t.bvh1, t.tris1, t.alpha1, t.haveInst1, t.texStorage, t.globals);
  return hit;
}

/* IntegratorCommon::shadowTrace, Common.cpp:156-180: tree 0 only, the plain closest-hit walk (no alpha test) */
static inline Lite_Hit ref_shadowClosest(float3 pos, float3 dir, const RefTrees t)
{
  Lite_Hit hit = Make_Lite_Hit(MAXFLOAT, -1);
  if (t.haveInst0)
    hit = BVH4InstTraverse(pos, dir, 0.0f, hit, t.bvh0, t.tris0);
  else
    hit = BVH4Traverse(pos, dir, 0.0f, hit, t.bvh0, t.tris0);
  return hit;
}

static inline RefTrees ref_makeTrees(__global const float4* bvh, __global const float4* tris, __global const uint2* alpha, int haveInst,
                                     __global const float4* bvh1, __global const float4* tris1, __global const uint2* alpha1, int haveInst1,
                                     __global const int4* texStorage, __global const EngineGlobals* globals)
{
  RefTrees t;
  t.bvh0 = bvh; t.tris0 = tris; t.alpha0 = alpha; t.haveInst0 = haveInst;
  t.bvh1 = bvh1; t.tris1 = tris1; t.alpha1 = alpha1; t.haveInst1 = haveInst1;
  t.texStorage = texStorage; t.globals = globals;
  return t;
}

__kernel void ref_trace(__global const float4* pos4, __global const float4* dir4, __global const float4* bvh, __global const float4* tris,
                        __global Lite_Hit* hits, int haveInst, int n,
                        __global const uint2* alpha, __global const float4* bvh1, __global const float4* tris1, __global const uint2* alpha1, int haveInst1,
                        __global const int4* in_texStorage, __global const EngineGlobals* a_globals)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  const RefTrees t = ref_makeTrees(bvh, tris, alpha, haveInst, bvh1, tris1, alpha1, haveInst1, in_texStorage, a_globals);
  hits[i] = ref_rayTrace(to_float3(pos4[i]), to_float3(dir4[i]), t);
}

/* kernel_EvalSurface, CPUExp_Integrators_PT_Loop.cpp:35-84 */
static inline SurfaceHit ref_evalSurface(float3 ray_pos, float3 ray_dir, Lite_Hit hit, __global const float4* in_matrices,
                                         __global const float4* in_geomStorage, __global const EngineGlobals* a_globals)
{
  const float4x4 instanceMatrixInv = fetchMatrix(hit, in_matrices);
  const float3 rayPosLS = mul4x3(instanceMatrixInv, ray_pos);
  const float3 rayDirLS = mul3x3(instanceMatrixInv, ray_dir);
  __global const PlainMesh* mesh = fetchMeshHeader(hit, in_geomStorage, a_globals);
  const SurfaceHit surfHit = surfaceEvalLS(rayPosLS, rayDirLS, hit, mesh);
  const float4x4 instanceMatrix = inverse4x4(instanceMatrixInv);
  SurfaceHit surfHitWS = surfHit;
  const float multInv = 1.0f / sqrt(3.0f);
  const float3 shadowStartPos = mul3x3(instanceMatrix, make_float3(multInv*surfHitWS.sRayOff, multInv*surfHitWS.sRayOff, multInv*surfHitWS.sRayOff));
  const float4x4 normalMatrix = transpose(instanceMatrixInv);
  surfHitWS.pos        = mul4x3(instanceMatrix, surfHit.pos);
  surfHitWS.normal     = normalize(mul3x3(normalMatrix, surfHit.normal));
  surfHitWS.flatNormal = normalize(mul3x3(normalMatrix, surfHit.flatNormal));
  surfHitWS.tangent    = normalize(mul3x3(normalMatrix, surfHit.tangent));
  surfHitWS.biTangent  = normalize(mul3x3(normalMatrix, surfHit.biTangent));
  surfHitWS.t          = length(surfHitWS.pos - ray_pos);
  surfHitWS.sRayOff    = length(shadowStartPos);
  surfHitWS.texCoordCamProj = make_float2(0.0f, 0.0f);
  return surfHitWS;   /* no remap lists in the fixtures: remapMaterialId returns the id unchanged for a null table */
}

__kernel void ref_eval_surface(__global const float4* pos4, __global const float4* dir4, __global const Lite_Hit* hits,
                               __global const float4* in_matrices, __global const float4* in_geomStorage,
                               __global const EngineGlobals* a_globals, __global float* out24, int n)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  __global float* r = out24 + i * 24;
  for (int k = 0; k < 24; k++) r[k] = 0.0f;
  const Lite_Hit hit = hits[i];
  if (HitNone(hit)) { r[17] = as_float(-1); return; }
  const SurfaceHit sh = ref_evalSurface(to_float3(pos4[i]), to_float3(dir4[i]), hit, in_matrices, in_geomStorage, a_globals);
  r[0] = sh.pos.x; r[1] = sh.pos.y; r[2] = sh.pos.z; r[3] = sh.normal.x; r[4] = sh.normal.y; r[5] = sh.normal.z;
  r[6] = sh.flatNormal.x; r[7] = sh.flatNormal.y; r[8] = sh.flatNormal.z; r[9] = sh.tangent.x; r[10] = sh.tangent.y; r[11] = sh.tangent.z;
  r[12] = sh.biTangent.x; r[13] = sh.biTangent.y; r[14] = sh.biTangent.z; r[15] = sh.texCoord.x; r[16] = sh.texCoord.y;
  r[17] = as_float(sh.matId); r[18] = sh.t; r[19] = sh.sRayOff; r[20] = sh.hfi ? 1.0f : 0.0f;
}

/* IntegratorMISPTLoop2::PathTrace: the stage order of PT_Loop.cpp:264-321 with the reference's functions as bodies */
__kernel void ref_path_trace(__global const float4* pos4, __global const float4* dir4, __global uint2* rng2,
                             __global const float4* bvh, __global const float4* tris, int haveInst,
                             __global const float4* in_matrices, __global const int* instLightInstId,
                             __global const float4* in_geomStorage, __global const float4* in_mtlStorage,
                             __global const int4* in_texStorage, __global const float4* in_pdfStorage,
                             __global const EngineGlobals* a_globals, __global float4* color4, int n,
                             __global const uint2* alpha, __global const float4* bvh1, __global const float4* tris1, __global const uint2* alpha1, int haveInst1,
                             __global const int4* in_texStorageAux)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  const RefTrees trees = ref_makeTrees(bvh, tris, alpha, haveInst, bvh1, tris1, alpha1, haveInst1, in_texStorage, a_globals);

  float3 ray_pos = to_float3(pos4[i]);
  float3 ray_dir = to_float3(dir4[i]);
  RandomGen gen;
  gen.state = rng2[i];

  float3 accumColor        = make_float3(0, 0, 0);
  float3 accumuThoroughput = make_float3(1, 1, 1);
  float3 currColor         = make_float3(0, 0, 0);
  MisData misPrev          = makeInitialMisData();
  uint flags               = 0;
  float rays               = 0.0f;
  const int maxDepth       = a_globals->varsI[HRT_TRACE_DEPTH];

  for (int depth = 0; depth < maxDepth; depth++)
  {
    /* kernel_RayTrace */
    const Lite_Hit hit = ref_rayTrace(ray_pos, ray_dir, trees);
    rays += 1.0f;

    /* kernel_HitEnvironment */
    if (HitNone(hit))
    {
      currColor = environmentColor(ray_dir, misPrev, flags, a_globals, in_mtlStorage, in_pdfStorage, in_texStorage);
      break;
    }

    /* kernel_EvalSurface */
    const SurfaceHit surfElem = ref_evalSurface(ray_pos, ray_dir, hit, in_matrices, in_geomStorage, a_globals);
    __global const PlainMaterial* pHitMaterial = materialAt(a_globals, in_mtlStorage, surfElem.matId);

    /* kernel_EvalEmission */
    {
      const int lightOffset0 = instLightInstId[hit.instId];
      __global const PlainLight* pLightHit = lightAt(a_globals, lightOffset0);
      ProcTextureList ptl;
      InitProcTextureList(&ptl);
      const float3 emission = emissionEval(ray_pos, ray_dir, &surfElem, flags, (misPrev.isSpecular == 1), pLightHit, pHitMaterial,
                                           in_texStorage, in_pdfStorage, a_globals, &ptl);
      if (dot(emission, emission) > 1e-3f)
      {
        __global const PlainLight* pLight = 0;
        if (a_globals->lightsNum != 0 && lightOffset0 >= 0)
          pLight = lightAt(a_globals, lightOffset0);
        if (pLight != 0)
        {
          const float lgtPdf  = lightPdfSelectRev(pLight)*lightEvalPDF(pLight, ray_pos, ray_dir, surfElem.pos, surfElem.normal, surfElem.texCoord, in_pdfStorage, a_globals);
          const float bsdfPdf = misPrev.matSamplePdf;
          float misWeight     = misWeightHeuristic(bsdfPdf, lgtPdf);
          if (misPrev.isSpecular)
            misWeight = 1.0f;
          currColor = emission*misWeight;
        }
        else
          currColor = emission;
        break;
      }
      else if (depth >= maxDepth - 1)
      {
        currColor = make_float3(0, 0, 0);
        break;
      }
    }

    /* kernel_LightSelect */
    const float4 rndLightData = rndLight(&gen, depth, a_globals->rmQMC, 0, 0);
    float lightPickProb = 1.0f;
    const int lightOffset = SelectRandomLightRev(rndLightData.z, surfElem.pos, a_globals, &lightPickProb);

    /* kernel_LightSample */
    float3 shadowRayPos = make_float3(0, 0, 0), shadowRayDir = make_float3(0, 0, 0);
    ShadowSample explicitSam;
    explicitSam.pos = make_float3(0, 0, 0); explicitSam.color = make_float3(0, 0, 0); explicitSam.pdf = 0.0f;
    explicitSam.maxDist = 0.0f; explicitSam.cosAtLight = 0.0f; explicitSam.isPoint = false;
    if (lightOffset >= 0)
    {
      __global const PlainLight* pLight = lightAt(a_globals, lightOffset);
      LightSampleRev(pLight, to_float3(rndLightData), surfElem.pos, a_globals, in_pdfStorage, in_texStorage, &explicitSam);
      shadowRayDir = normalize(explicitSam.pos - surfElem.pos);
      shadowRayPos = OffsShadowRayPos(surfElem.pos, surfElem.normal, shadowRayDir, surfElem.sRayOff);
    }

    /* kernel_ShadowTrace: IntegratorCommon::shadowTrace, Common.cpp:156-180 */
    float3 shadow = make_float3(0, 0, 0);
    if (lightOffset >= 0)
    {
      const float t_far = length(shadowRayPos - explicitSam.pos)*0.995f;
      const Lite_Hit sh = ref_shadowClosest(shadowRayPos, shadowRayDir, trees);
      rays += 1.0f;
      shadow = (HitSome(sh) && sh.t > 0.0f && sh.t < t_far) ? make_float3(0.0f, 0.0f, 0.0f) : make_float3(1.0f, 1.0f, 1.0f);
    }

    /* kernel_Shade */
    float3 explicitColor = make_float3(0, 0, 0);
    if (lightOffset >= 0)
    {
      ShadeContext sc;
      sc.wp = surfElem.pos;
      sc.l  = shadowRayDir;
      sc.v  = (-1.0f)*ray_dir;
      sc.n  = surfElem.normal;
      sc.fn = surfElem.flatNormal;
      sc.tg = surfElem.tangent;
      sc.bn = surfElem.biTangent;
      sc.tc = surfElem.texCoord;
      sc.tccp = surfElem.texCoordCamProj;
      sc.hfi = surfElem.hfi;
      ProcTextureList ptlCopy;
      InitProcTextureList(&ptlCopy);
      GetProcTexturesIdListFromMaterialHead(pHitMaterial, &ptlCopy);
      const BxDFResult evalData = materialEval(pHitMaterial, &sc, (EVAL_FLAG_DEFAULT), a_globals, in_texStorage, in_texStorageAux, &ptlCopy);

      const float cosThetaOut1 = fmax(+dot(shadowRayDir, surfElem.normal), 0.0f);
      const float cosThetaOut2 = fmax(-dot(shadowRayDir, surfElem.normal), 0.0f);
      const float3 bxdfVal     = (evalData.brdf*cosThetaOut1 + evalData.btdf*cosThetaOut2);
      const float lgtPdf       = explicitSam.pdf*lightPickProb;
      float misWeight = misWeightHeuristic(lgtPdf, evalData.pdfFwd);
      if (explicitSam.isPoint)
        misWeight = 1.0f;
      explicitColor = (1.0f / lightPickProb)*(explicitSam.color * (1.0f / fmax(explicitSam.pdf, DEPSILON2)))*bxdfVal*misWeight*shadow;
    }

    /* kernel_NextBounce */
    {
      const int rayBounceNum = unpackBounceNum(flags);
      float allRands[MMLT_FLOATS_PER_BOUNCE];
      RndMatAll(&gen, 0, rayBounceNum, a_globals->rmQMC, 0, 0, allRands);

      ProcTextureList ptlDummy;
      InitProcTextureList(&ptlDummy);
      MatSample matSam; int matOffset;
      MaterialSampleAndEvalBxDF(pHitMaterial, allRands, &surfElem, ray_dir, make_float3(0, 0, 0), flags, false,
                                a_globals, in_texStorage, in_texStorageAux, &ptlDummy, &matSam, &matOffset);

      const float3 bxdfVal = matSam.color * (1.0f / fmax(matSam.pdf, 1e-20f));
      const float cosTheta = fabs(dot(matSam.direction, surfElem.normal));

      ray_dir              = matSam.direction;
      ray_pos              = OffsRayPos(surfElem.pos, surfElem.normal, matSam.direction);
      misPrev              = makeInitialMisData();
      misPrev.isSpecular   = isPureSpecular(matSam);
      misPrev.matSamplePdf = matSam.pdf;
      flags                = flagsNextBounceLite(flags, matSam, a_globals);

      accumColor        += accumuThoroughput*explicitColor;
      accumuThoroughput *= cosTheta*bxdfVal;
    }
  }

  accumColor += accumuThoroughput*currColor;   /* kernel_AddLastBouceContrib */
  color4[i] = to_float4(accumColor, rays);
  rng2[i]   = gen.state;
}

/* One shading point through the reference's light and material functions (fixtures 5 and 6 of SURVEY.md 8c):
   kernel_LightSelect + kernel_LightSample + materialEval towards the sample + kernel_NextBounce's BxDF sampling, with the
   random numbers handed in instead of drawn.  out = 28 floats per point, layout documented in tests/oracle_lib.py. */
__kernel void ref_shade_point(__global const float* surf24, __global const float4* dir4, __global const int* flagsIn,
                              __global const float4* rndLight4, __global const float* rands10,
                              __global const float4* in_mtlStorage, __global const int4* in_texStorage, __global const float4* in_pdfStorage,
                              __global const EngineGlobals* a_globals, __global float* out28, int n, __global const int4* in_texStorageAux)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  __global const float* r = surf24 + i * 24;
  __global float* o = out28 + i * 28;
  for (int k = 0; k < 28; k++) o[k] = 0.0f;
  SurfaceHit surfElem;
  surfElem.pos = make_float3(r[0], r[1], r[2]); surfElem.normal = make_float3(r[3], r[4], r[5]);
  surfElem.flatNormal = make_float3(r[6], r[7], r[8]); surfElem.tangent = make_float3(r[9], r[10], r[11]);
  surfElem.biTangent = make_float3(r[12], r[13], r[14]); surfElem.texCoord = make_float2(r[15], r[16]);
  surfElem.texCoordCamProj = make_float2(0, 0);
  surfElem.matId = as_int(r[17]); surfElem.t = r[18]; surfElem.sRayOff = r[19]; surfElem.hfi = (r[20] != 0.0f);
  if (surfElem.matId < 0) { o[8] = as_float(-2); return; }
  const float3 ray_dir = to_float3(dir4[i]);
  const uint flags = (uint)flagsIn[i];
  __global const PlainMaterial* pHitMaterial = materialAt(a_globals, in_mtlStorage, surfElem.matId);

  const float4 rndLightData = rndLight4[i];
  float lightPickProb = 1.0f;
  const int lightOffset = SelectRandomLightRev(rndLightData.z, surfElem.pos, a_globals, &lightPickProb);
  ShadowSample explicitSam;
  explicitSam.pos = make_float3(0, 0, 0); explicitSam.color = make_float3(0, 0, 0); explicitSam.pdf = 0.0f;
  explicitSam.maxDist = 0.0f; explicitSam.cosAtLight = 0.0f; explicitSam.isPoint = false;
  o[7] = lightPickProb; o[8] = as_float(lightOffset);
  if (lightOffset >= 0)
  {
    __global const PlainLight* pLight = lightAt(a_globals, lightOffset);
    LightSampleRev(pLight, to_float3(rndLightData), surfElem.pos, a_globals, in_pdfStorage, in_texStorage, &explicitSam);
    const float3 shadowRayDir = normalize(explicitSam.pos - surfElem.pos);
    o[0] = explicitSam.pos.x; o[1] = explicitSam.pos.y; o[2] = explicitSam.pos.z; o[3] = explicitSam.pdf;
    o[4] = explicitSam.color.x; o[5] = explicitSam.color.y; o[6] = explicitSam.color.z; o[9] = explicitSam.isPoint ? 1.0f : 0.0f;
    ShadeContext sc;
    sc.wp = surfElem.pos; sc.l = shadowRayDir; sc.v = (-1.0f)*ray_dir; sc.n = surfElem.normal; sc.fn = surfElem.flatNormal;
    sc.tg = surfElem.tangent; sc.bn = surfElem.biTangent; sc.tc = surfElem.texCoord; sc.tccp = surfElem.texCoordCamProj; sc.hfi = surfElem.hfi;
    ProcTextureList ptlCopy;
    InitProcTextureList(&ptlCopy);
    GetProcTexturesIdListFromMaterialHead(pHitMaterial, &ptlCopy);
    const BxDFResult evalData = materialEval(pHitMaterial, &sc, (EVAL_FLAG_DEFAULT), a_globals, in_texStorage, in_texStorageAux, &ptlCopy);
    o[10] = evalData.brdf.x; o[11] = evalData.brdf.y; o[12] = evalData.brdf.z; o[13] = evalData.pdfFwd;
    o[14] = evalData.btdf.x; o[15] = evalData.btdf.y; o[16] = evalData.btdf.z;
  }
  float allRands[MMLT_FLOATS_PER_BOUNCE];
  for (int k = 0; k < MMLT_FLOATS_PER_BOUNCE; k++) allRands[k] = 0.0f;
  for (int k = 0; k < 10; k++) allRands[k] = rands10[i * 10 + k];
  ProcTextureList ptlDummy;
  InitProcTextureList(&ptlDummy);
  MatSample matSam; int matOffset;
  MaterialSampleAndEvalBxDF(pHitMaterial, allRands, &surfElem, ray_dir, make_float3(0, 0, 0), flags, false,
                            a_globals, in_texStorage, in_texStorageAux, &ptlDummy, &matSam, &matOffset);
  o[17] = matSam.color.x; o[18] = matSam.color.y; o[19] = matSam.color.z; o[20] = matSam.pdf;
  o[21] = matSam.direction.x; o[22] = matSam.direction.y; o[23] = matSam.direction.z;
  o[24] = as_float(matSam.flags); o[25] = as_float((int)flagsNextBounceLite(flags, matSam, a_globals));
}

/* ---- row f3 building blocks: the reference's own LightSampleForward / lightPdfFwd / CameraImageToSurfaceFactor /
   worldPosToScreenSpace / MutateKelemen, one call per item with the random numbers handed in. */
__kernel void ref_light_sample_forward(__global const int* lightIds, __global const float4* rands4, __global const int4* in_texStorage,
                                       __global const float4* in_pdfStorage, __global const EngineGlobals* a_globals, __global float* out16, int n)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  __global const PlainLight* pLight = lightAt(a_globals, lightIds[i]);
  LightSampleFwd sam;
  LightSampleForward(pLight, rands4[i], make_float2(0.0f, 0.0f), a_globals, in_texStorage, in_pdfStorage, &sam);
  __global float* o = out16 + i * 16;
  o[0] = sam.pos.x; o[1] = sam.pos.y; o[2] = sam.pos.z; o[3] = sam.dir.x; o[4] = sam.dir.y; o[5] = sam.dir.z;
  o[6] = sam.norm.x; o[7] = sam.norm.y; o[8] = sam.norm.z; o[9] = sam.color.x; o[10] = sam.color.y; o[11] = sam.color.z;
  o[12] = sam.pdfA; o[13] = sam.pdfW; o[14] = sam.cosTheta; o[15] = sam.isPoint ? 1.0f : 0.0f;
}

__kernel void ref_light_pdf_fwd(__global const int* lightIds, __global const float* cosTheta, __global const int4* in_texStorage,
                                __global const float4* in_pdfStorage, __global const EngineGlobals* a_globals, __global float4* out4, int n)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  __global const PlainLight* pLight = lightAt(a_globals, lightIds[i]);
  const LightPdfFwd p = lightPdfFwd(pLight, make_float3(0.0f, 0.0f, 1.0f), cosTheta[i], a_globals, in_texStorage, in_pdfStorage);
  out4[i] = make_float4(p.pdfA, p.pdfW, p.pickProb, 0.0f);
}

__kernel void ref_camera_connect(__global const float4* pos4, __global const float4* norm4, __global const float2* disk2,
                                 __global const EngineGlobals* a_globals, __global float* out8, int n)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  float3 camDir; float zDepth;
  const float f = CameraImageToSurfaceFactor(to_float3(pos4[i]), to_float3(norm4[i]), a_globals, disk2[i], &camDir, &zDepth);
  const float2 scr = worldPosToScreenSpace(to_float3(pos4[i]), a_globals);
  __global float* o = out8 + i * 8;
  o[0] = f; o[1] = camDir.x; o[2] = camDir.y; o[3] = camDir.z; o[4] = zDepth; o[5] = scr.x; o[6] = scr.y; o[7] = 0.0f;
}

/* The miss shader of the reference's wavefront layer as HitEnvOrLightKernel calls it (shaders/material.cl:354): environmentColorExtended, cbidir.h:593-629 -- suns of the
   header's table, the environment, the back-plate for camera rays and rays that only crossed transparent surfaces.  in8 per ray: origin xyz, previous BSDF pdf, previous
   bounce specular (0/1), ray flags, pixel x, pixel y (the last three as int bits); prevMaterialOffset = -1 as the path tracer leaves it (CPUExp_Integrators_PT_Loop.cpp:247-249). */
__kernel void ref_environment_extended(__global const float4* dir4, __global const float* in8, __global const float4* in_mtlStorage, __global const int4* in_texStorage,
                                       __global const float4* in_pdfStorage, __global const EngineGlobals* a_globals, __global float4* out4, int n)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  __global const float* in = in8 + i * 8;
  MisData misPrev;
  misPrev.matSamplePdf = in[3]; misPrev.cosThetaPrev = 1.0f; misPrev.prevMaterialOffset = -1; misPrev.isSpecular = (in[4] != 0.0f) ? 1 : 0;
  const float3 c = environmentColorExtended(make_float3(in[0], in[1], in[2]), to_float3(dir4[i]), misPrev, (uint)as_int(in[5]), as_int(in[6]), as_int(in[7]),
                                            a_globals, in_mtlStorage, in_pdfStorage, in_texStorage);
  out4[i] = make_float4(c.x, c.y, c.z, 0.0f);
}

__kernel void ref_mutate_kelemen(__global const float* values, __global const float2* rands2, float p2, float p1, __global float* out, int n)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  out[i] = MutateKelemen(values[i], rands2[i], p2, p1);
}

/* IntegratorMMLT::F, CPUExp_Integrators_MMLT.cpp:146-315: the control flow of F, LightPath/TraceLightPath (:637-754), CameraPath (:756-929),
 * ConnectEye / ConnectShadow / ConnectEndPoints (:931-1047) around the reference's own inline functions (rndSplitMMLT, rndLens, RndLightMMLT,
 * RndMatAll, MakeEyeRayFromF4Rnd, LightSampleForward, lightPdfFwd, MaterialSampleAndEvalBxDF, materialEval, emissionEval,
 * CameraImageToSurfaceFactor, ConnectEyeP, ConnectShadowP, ConnectEndPointsP).  OpenCL has no recursion: the two recursions are loops and
 * CameraPath's return-trip products are applied afterwards, deepest level first.  out8 = colour, x, y, split, MIS weight, contribFunc. */
#define REF_MMLT_MAX_DEPTH 16
__kernel void ref_mmlt_f(__global const int* depth, __global const float* xvec, int stride,
                         __global const float4* bvh, __global const float4* tris, int haveInst,
                         __global const float4* in_matrices, __global const int* instLightInstId,
                         __global const float4* in_geomStorage, __global const float4* in_mtlStorage,
                         __global const int4* in_texStorage, __global const float4* in_pdfStorage,
                         __global const EngineGlobals* a_globals, __global float* out8, int n,
                         __global const uint2* alpha, __global const float4* bvh1, __global const float4* tris1, __global const uint2* alpha1, int haveInst1,
                         __global const int4* in_texStorageAux)
{
  const int tid = get_global_id(0);
  if (tid >= n) return;
  const RefTrees trees = ref_makeTrees(bvh, tris, alpha, haveInst, bvh1, tris1, alpha1, haveInst1, in_texStorage, a_globals);
  __global const float* rptr0 = xvec + (size_t)tid * stride;
  const int d = depth[tid];
  RandomGen gen;
  gen.state = (uint2)(1, 2);   /* never drawn from: every rnd* call below has its rptr set */
  ProcTextureList ptlDummy;
  InitProcTextureList(&ptlDummy);

  PdfVertex pdfArray[REF_MMLT_MAX_DEPTH + 2];
  for (int k = 0; k < REF_MMLT_MAX_DEPTH + 2; k++) { pdfArray[k].pdfFwd = 0.0f; pdfArray[k].pdfRev = 0.0f; }

  const int m_width  = (int)(a_globals->varsF[HRT_WIDTH_F]);
  const int m_height = (int)(a_globals->varsF[HRT_HEIGHT_F]);
  const float mLightSubPathCount = (float)(m_width*m_height);
  const bool m_splitDLByGrammar  = (a_globals->varsI[HRT_MMLT_FIRST_BOUNCE] > 3);

  const int s = rndSplitMMLT(&gen, rptr0, d);
  const int t = d - s;
  const int lightTraceDepth = s - 1;
  const int camTraceDepth   = t;
  const float4 lensOffs = rndLens(&gen, rptr0, make_float2(1.0f, 1.0f), 0, 0, 0);
  int x = (int)(lensOffs.x*(float)(m_width)  + 0.5f);
  int y = (int)(lensOffs.y*(float)(m_height) + 0.5f);

  /* (1) camera sub-path */
  PathVertex cv;
  InitPathVertex(&cv);
  if (camTraceDepth > 0)
  {
    __global const float* rptr = rptr0 + camOffsetInRandArrayMMLT(s);
    float fx, fy;
    float3 ray_pos, ray_dir;
    MakeEyeRayFromF4Rnd(lensOffs, a_globals, &ray_pos, &ray_dir, &fx, &fy);
    x = (int)(fx + 0.5f);
    y = (int)(fy + 0.5f);
    if (x >= m_width)  x = m_width - 1;
    if (y >= m_height) y = m_height - 1;
    const bool haveToHitLight = (lightTraceDepth == -1);
    MisData misPrev = makeInitialMisData();
    uint flags = 0;
    float3 factors[REF_MMLT_MAX_DEPTH + 2];
    int nFactors = 0, zeroFrom = -1;
    cv.valid = false; cv.accColor = make_float3(0, 0, 0);
    for (int a_currDepth = 1; a_currDepth <= camTraceDepth; a_currDepth++)
    {
      const int prevVertexId = d - a_currDepth + 1;
      const Lite_Hit hit = ref_rayTrace(ray_pos, ray_dir, trees);
      if (HitNone(hit)) break;
      const SurfaceHit surfElem = ref_evalSurface(ray_pos, ray_dir, hit, in_matrices, in_geomStorage, a_globals);
      const float cosHere = fabs(dot(ray_dir, surfElem.normal));
      const float cosPrev = fabs(misPrev.cosThetaPrev);
      float GTerm = 1.0f;
      if (a_currDepth == 1)
      {
        float3 camDirDummy; float zDepthDummy;
        const float imageToSurfaceFactor = CameraImageToSurfaceFactor(surfElem.pos, surfElem.normal, a_globals, make_float2(0,0), &camDirDummy, &zDepthDummy);
        const float cameraPdfA = imageToSurfaceFactor / mLightSubPathCount;
        pdfArray[d].pdfRev = cameraPdfA;
        pdfArray[d].pdfFwd = 1.0f;
      }
      else
      {
        const float dist = length(ray_pos - surfElem.pos);
        GTerm = cosHere*cosPrev / fmax(dist*dist, DEPSILON2);
      }
      __global const PlainMaterial* pHitMaterial = materialAt(a_globals, in_mtlStorage, surfElem.matId);
      const int lightOffset0 = instLightInstId[hit.instId];
      __global const PlainLight* pLightHit = lightAt(a_globals, lightOffset0);
      ProcTextureList ptl;
      InitProcTextureList(&ptl);
      const float3 emission = emissionEval(ray_pos, ray_dir, &surfElem, flags, (misPrev.isSpecular == 1), pLightHit, pHitMaterial,
                                           in_texStorage, in_pdfStorage, a_globals, &ptl);
      if (dot(emission, emission) > 1e-6f)
      {
        if (a_currDepth == camTraceDepth && haveToHitLight)
        {
          const LightPdfFwd lPdfFwd = lightPdfFwd(pLightHit, ray_dir, cosHere, a_globals, in_texStorage, in_pdfStorage);
          const float pdfLightWP    = lPdfFwd.pdfW / fmax(cosHere, DEPSILON);
          const float pdfMatRevWP   = misPrev.matSamplePdf / fmax(cosPrev, DEPSILON);
          pdfArray[0].pdfFwd = lPdfFwd.pdfA / (float)(a_globals->lightsNum);
          pdfArray[0].pdfRev = 1.0f;
          pdfArray[1].pdfFwd = pdfLightWP*GTerm;
          pdfArray[1].pdfRev = misPrev.isSpecular ? -1.0f*GTerm : pdfMatRevWP*GTerm;
          cv.hit = surfElem; cv.ray_dir = ray_dir; cv.accColor = emission; cv.valid = true;
        }
        break;
      }
      else if (a_currDepth == camTraceDepth && !haveToHitLight)
      {
        cv.hit = surfElem; cv.ray_dir = ray_dir; cv.valid = true; cv.accColor = make_float3(1, 1, 1);
        cv.wasSpecOnly = m_splitDLByGrammar ? flagsHaveOnlySpecular(flags) : false;
        if (camTraceDepth != 1)
        {
          const float lastPdfWP = misPrev.matSamplePdf / fmax(cosPrev, DEPSILON);
          cv.lastGTerm = GTerm;
          pdfArray[prevVertexId].pdfRev = misPrev.isSpecular ? -1.0f*GTerm : GTerm*lastPdfWP;
        }
        else
          cv.lastGTerm = 1.0f;
        break;
      }

      float allRands[MMLT_FLOATS_PER_BOUNCE];
      RndMatAll(&gen, rptr + rndMatOffsetMMLT(a_currDepth - 1), a_currDepth - 1, a_globals->rmQMC, 0, 0, allRands);
      MatSample matSam; int matOffset;
      MaterialSampleAndEvalBxDF(pHitMaterial, allRands, &surfElem, ray_dir, make_float3(0, 0, 0), packBounceNum(0, a_currDepth - 1), false,
                                a_globals, in_texStorage, in_texStorageAux, &ptlDummy, &matSam, &matOffset);
      const float3 bxdfVal = matSam.color;
      const float cosNext  = fabs(dot(matSam.direction, surfElem.normal));
      if (a_currDepth == 1)
      {
        if (isPureSpecular(matSam))
          pdfArray[d].pdfFwd = 0.0f;
      }
      else
      {
        if (!isPureSpecular(matSam))
        {
          ShadeContext sc;
          sc.wp = surfElem.pos; sc.l = (-1.0f)*ray_dir; sc.v = matSam.direction; sc.n = surfElem.normal; sc.fn = surfElem.flatNormal;
          sc.tg = surfElem.tangent; sc.bn = surfElem.biTangent; sc.tc = surfElem.texCoord; sc.tccp = surfElem.texCoordCamProj; sc.hfi = surfElem.hfi;
          const float pdfFwdW  = materialEval(pHitMaterial, &sc, (EVAL_FLAG_DEFAULT), a_globals, in_texStorage, in_texStorageAux, &ptlDummy).pdfFwd;
          const float pdfFwdWP = pdfFwdW / fmax(cosHere, DEPSILON);
          pdfArray[prevVertexId].pdfFwd = pdfFwdWP*GTerm;
        }
        else
          pdfArray[prevVertexId].pdfFwd = -1.0f*GTerm;
        const float pdfCamPrevWP = misPrev.matSamplePdf / fmax(cosPrev, DEPSILON);
        pdfArray[prevVertexId].pdfRev = misPrev.isSpecular ? -1.0f*GTerm : pdfCamPrevWP*GTerm;
      }
      const bool stopDL = m_splitDLByGrammar ? flagsHaveOnlySpecular(flags) : false;
      factors[nFactors] = (bxdfVal*cosNext / fmax(matSam.pdf, DEPSILON2));
      if (stopDL && haveToHitLight && a_currDepth + 1 == camTraceDepth) zeroFrom = nFactors;
      nFactors++;

      const float3 nextRay_dir = matSam.direction;
      const float3 nextRay_pos = OffsRayPos(surfElem.pos, surfElem.normal, matSam.direction);
      MisData thisBounce       = makeInitialMisData();
      thisBounce.isSpecular    = isPureSpecular(matSam);
      thisBounce.matSamplePdf  = matSam.pdf;
      thisBounce.cosThetaPrev  = dot(nextRay_dir, surfElem.normal);
      flags   = flagsNextBounceLite(flags, matSam, a_globals);
      misPrev = thisBounce;
      ray_pos = nextRay_pos;
      ray_dir = nextRay_dir;
    }
    if (!cv.valid) cv.accColor = make_float3(0, 0, 0);
    for (int k = nFactors - 1; k >= 0; k--)
    {
      cv.accColor *= factors[k];
      if (k == zeroFrom) cv.accColor = make_float3(0, 0, 0);
    }
  }

  /* (2) light sub-path */
  PathVertex lv;
  InitPathVertex(&lv);
  if (lightTraceDepth > 0)
  {
    LightGroup2 lightSelector;
    RndLightMMLT(&gen, rptr0, &lightSelector);
    float lightPickProb = 1.0f;
    const int lightId = SelectRandomLightFwd(lightSelector.group2.z, a_globals, &lightPickProb);
    __global const PlainLight* pLight = lightAt(a_globals, lightId);
    LightSampleFwd sample;
    LightSampleForward(pLight, lightSelector.group1, make_float2(lightSelector.group2.x, lightSelector.group2.y), a_globals, in_texStorage, in_pdfStorage, &sample);
    pdfArray[0].pdfFwd = sample.pdfA*lightPickProb;
    pdfArray[0].pdfRev = 1.0f;
    float3 a_color = (1.0f/lightPickProb)*sample.color/(sample.pdfA*sample.pdfW);
    __global const float* rptr = rptr0 + MMLT_HEAD_TOTAL_SIZE;
    float3 ray_pos = sample.pos, ray_dir = sample.dir;
    float a_prevLightCos = sample.cosTheta, a_prevPdf = sample.pdfW;
    bool a_wasSpecular = false;
    for (int a_currDepth = 1; a_currDepth <= lightTraceDepth; a_currDepth++)
    {
      const Lite_Hit hit = ref_rayTrace(ray_pos, ray_dir, trees);
      if (!HitSome(hit)) break;
      const SurfaceHit surfElem = ref_evalSurface(ray_pos, ray_dir, hit, in_matrices, in_geomStorage, a_globals);
      const float cosCurr = fabs(-dot(ray_dir, surfElem.normal));
      const float dist    = length(surfElem.pos - ray_pos);
      const float GTermPrev = (a_prevLightCos*cosCurr / fmax(dist*dist, DEPSILON2));
      const float prevPdfWP = a_prevPdf / fmax(a_prevLightCos, DEPSILON);
      if (!a_wasSpecular)
        pdfArray[a_currDepth].pdfFwd = prevPdfWP*GTermPrev;
      else
        pdfArray[a_currDepth].pdfFwd = -1.0f*GTermPrev;
      __global const PlainMaterial* pHitMaterial = materialAt(a_globals, in_mtlStorage, surfElem.matId);
      float allRands[MMLT_FLOATS_PER_BOUNCE];
      RndMatAll(&gen, rptr + rndMatOffsetMMLT(a_currDepth - 1), a_currDepth - 1, a_globals->rmQMC, 0, 0, allRands);
      MatSample matSam; int matOffset;
      MaterialSampleAndEvalBxDF(pHitMaterial, allRands, &surfElem, ray_dir, make_float3(0, 0, 0), packBounceNum(0, a_currDepth - 1), true,
                                a_globals, in_texStorage, in_texStorageAux, &ptlDummy, &matSam, &matOffset);
      const float3 nextRay_dir = matSam.direction;
      const float3 nextRay_pos = OffsRayPos(surfElem.pos, surfElem.normal, matSam.direction);
      const float cosNext = fabs(+dot(nextRay_dir, surfElem.normal));
      if (a_currDepth == lightTraceDepth)
      {
        lv.hit = surfElem; lv.ray_dir = ray_dir; lv.accColor = a_color; lv.valid = true; lv.lastGTerm = GTermPrev;
        break;
      }
      if (!isPureSpecular(matSam))
      {
        ShadeContext sc;
        sc.wp = surfElem.pos; sc.l = (-1.0f)*ray_dir; sc.v = (-1.0f)*nextRay_dir; sc.n = surfElem.normal; sc.fn = surfElem.flatNormal;
        sc.tg = surfElem.tangent; sc.bn = surfElem.biTangent; sc.tc = surfElem.texCoord; sc.tccp = surfElem.texCoordCamProj; sc.hfi = surfElem.hfi;
        const float pdfW         = materialEval(pHitMaterial, &sc, (EVAL_FLAG_DEFAULT), a_globals, in_texStorage, in_texStorageAux, &ptlDummy).pdfFwd;
        const float prevPdfRevWP = pdfW / fmax(cosCurr, DEPSILON);
        pdfArray[a_currDepth].pdfRev = prevPdfRevWP*GTermPrev;
      }
      else
        pdfArray[a_currDepth].pdfRev = -1.0f*GTermPrev;
      a_color *= matSam.color*cosNext*(1.0f / fmax(matSam.pdf, DEPSILON2));
      ray_pos = nextRay_pos; ray_dir = nextRay_dir;
      a_prevLightCos = cosNext; a_prevPdf = matSam.pdf; a_wasSpecular = isPureSpecular(matSam);
    }
  }

  /* (3) connect */
  float3 sampleColor = make_float3(0, 0, 0);
  if (lightTraceDepth == -1)
    sampleColor = cv.accColor;
  else
  {
    if (camTraceDepth == 0)
    {
      if (lv.valid)
      {
        float3 camDir; float zDepth;
        const float imageToSurfaceFactor = CameraImageToSurfaceFactor(lv.hit.pos, lv.hit.normal, a_globals, make_float2(0,0), &camDir, &zDepth);
        __global const PlainMaterial* pHitMaterial = materialAt(a_globals, in_mtlStorage, lv.hit.matId);
        float signOfNormal = 1.0f;
        if ((materialGetFlags(pHitMaterial) & PLAIN_MATERIAL_HAVE_BTDF) != 0 && dot(camDir, lv.hit.normal) < -0.01f)
          signOfNormal = -1.0f;
        const Lite_Hit hit = ref_rayTrace(lv.hit.pos + epsilonOfPos(lv.hit.pos)*signOfNormal*lv.hit.normal, camDir, trees);
        if (imageToSurfaceFactor <= 0.0f || (HitSome(hit) && hit.t <= zDepth))
        {
          x = -1; y = -1;
        }
        else
          sampleColor = ConnectEyeP(&lv, mLightSubPathCount, camDir, imageToSurfaceFactor, a_globals, in_mtlStorage, in_texStorage, in_texStorageAux, &ptlDummy,
                                    &pdfArray[lightTraceDepth + 0], &pdfArray[lightTraceDepth + 1], &x, &y);
      }
    }
    else if (lightTraceDepth == 0)
    {
      if (cv.valid && !cv.wasSpecOnly)
      {
        float3 explicitColor = make_float3(0, 0, 0);
        LightGroup2 lightSelector;
        RndLightMMLT(&gen, rptr0, &lightSelector);
        float lightPickProb = 1.0f;
        const int lightOffset = SelectRandomLightRev(lightSelector.group2.z, cv.hit.pos, a_globals, &lightPickProb);
        if (lightOffset >= 0)
        {
          __global const PlainLight* pLight = lightAt(a_globals, lightOffset);
          ShadowSample explicitSam;
          LightSampleRev(pLight, to_float3(lightSelector.group1), cv.hit.pos, a_globals, in_pdfStorage, in_texStorage, &explicitSam);
          const float3 shadowRayDir = normalize(explicitSam.pos - cv.hit.pos);
          const float3 shadowRayPos = OffsRayPos(cv.hit.pos, cv.hit.normal, shadowRayDir);
          const Lite_Hit sh = ref_shadowClosest(shadowRayPos, shadowRayDir, trees);
          const float t_far = explicitSam.maxDist*0.9995f;
          const float3 shadow = (HitSome(sh) && sh.t > 0.0f && sh.t < t_far) ? make_float3(0.0f, 0.0f, 0.0f) : make_float3(1.0f, 1.0f, 1.0f);
          if (dot(shadow, shadow) > 1e-12f)
            explicitColor = shadow*ConnectShadowP(&cv, t, pLight, explicitSam, lightPickProb, a_globals, in_mtlStorage, in_texStorage, in_texStorageAux, in_pdfStorage, &ptlDummy,
                                                  &pdfArray[0], &pdfArray[1], &pdfArray[2]);
        }
        sampleColor = cv.accColor*explicitColor;
      }
    }
    else
    {
      if (cv.valid)
      {
        float3 explicitColor = make_float3(0, 0, 0);
        if (lv.valid)
        {
          const float3 diff = cv.hit.pos - lv.hit.pos;
          const float dist2 = fmax(dot(diff, diff), DEPSILON2);
          const float  dist = sqrt(dist2);
          const float3 lToC = diff / dist;
          const float cosAtLightVertex  = +dot(lv.hit.normal, lToC);
          const float cosAtCameraVertex = -dot(cv.hit.normal, lToC);
          const float GTerm = cosAtLightVertex*cosAtCameraVertex / dist2;
          if (!(GTerm < 0.0f))
          {
            const float3 shadowRayPos = OffsRayPos(lv.hit.pos, lv.hit.normal, lToC);
            const Lite_Hit sh = ref_shadowClosest(shadowRayPos, lToC, trees);
            const float t_far = dist*0.9995f;
            const float3 shadow = (HitSome(sh) && sh.t > 0.0f && sh.t < t_far) ? make_float3(0.0f, 0.0f, 0.0f) : make_float3(1.0f, 1.0f, 1.0f);
            if (!(dot(shadow, shadow) < 1e-12f))
              explicitColor = shadow*ConnectEndPointsP(&lv, &cv, d, a_globals, in_mtlStorage, in_texStorage, in_texStorageAux, &ptlDummy,
                                                       &pdfArray[s - 1], &pdfArray[s + 0], &pdfArray[s + 1]);
          }
        }
        sampleColor = cv.accColor*explicitColor*lv.accColor;
      }
    }
  }

  /* (4) MIS weight */
  float misWeight = 1.0f;
  if (dot(sampleColor, sampleColor) > 1e-12f)
  {
    float pdfThisWay = 1.0f;
    float pdfSumm    = 0.0f;
    for (int split = 0; split <= d; split++)
    {
      const int s1 = split;
      const int t1 = d - split;
      const bool specularMet = (split > 0) && (split < d) && (pdfArray[split].pdfRev < 0.0f || pdfArray[split].pdfFwd < 0.0f);
      float pdfOtherWay = specularMet ? 0.0f : 1.0f;
      if (split == d)
        pdfOtherWay = misHeuristicPower1(pdfArray[d].pdfFwd);
      for (int i = 0; i < s1; i++)
        pdfOtherWay *= misHeuristicPower1(pdfArray[i].pdfFwd);
      for (int i = s1 + 1; i <= d; i++)
        pdfOtherWay *= misHeuristicPower1(pdfArray[i].pdfRev);
      if (s1 == s && t1 == t)
        pdfThisWay = pdfOtherWay;
      pdfSumm += pdfOtherWay;
    }
    misWeight = pdfThisWay / fmax(pdfSumm, DEPSILON2);
  }
  sampleColor *= misWeight;
  if (!(x >= 0 && x < m_width && y >= 0 && y < m_height))
  {
    x = 0; y = 0;
    sampleColor = make_float3(0, 0, 0);
  }
  __global float* o = out8 + (size_t)tid * 8;
  o[0] = sampleColor.x; o[1] = sampleColor.y; o[2] = sampleColor.z; o[3] = (float)x; o[4] = (float)y; o[5] = (float)s;
  o[6] = misWeight; o[7] = contribFunc(sampleColor);
}

/* IntegratorCommon::gbufferEval / gbufferSample (CPUExp_GBuffer.cpp:15-113) with the reference's functions as bodies: MakeEyeRayFromF4Rnd,
 * the traversal and surface evaluation of the path tracer, materialEvalDiffuse (evalDiffuseColor, Common.cpp:244-251), initGBufferAll,
 * gbuffDiff, packGBuffer1 / packGBuffer2.  qmc = PlaneHammersley(64) (host code, globals_sys.cpp:45-61: handed in).  One work-item per pixel
 * of the window [x0, x0+nx) x [y0, y0+ny); out1 / out2 = the packed layers, raw14 = the winning sample unpacked. */
__kernel void ref_gbuffer(__global const float2* qmc, int x0, int y0, int nx, int ny,
                          __global const float4* bvh, __global const float4* tris, int haveInst,
                          __global const float4* in_matrices, __global const int* instLightInstId,
                          __global const float4* in_geomStorage, __global const float4* in_mtlStorage,
                          __global const int4* in_texStorage, __global const float4* in_pdfStorage,
                          __global const EngineGlobals* a_globals, __global float4* out1, __global float4* out2, __global float* raw14,
                          __global const uint2* alpha, __global const float4* bvh1, __global const float4* tris1, __global const uint2* alpha1, int haveInst1,
                          __global const int4* in_texStorageAux)
{
  const int p = get_global_id(0);
  if (p >= nx*ny) return;
  const RefTrees trees = ref_makeTrees(bvh, tris, alpha, haveInst, bvh1, tris1, alpha1, haveInst1, in_texStorage, a_globals);
  const int x = x0 + p % nx, y = y0 + p / nx;
  const int m_width  = (int)(a_globals->varsF[HRT_WIDTH_F]);
  const int m_height = (int)(a_globals->varsF[HRT_HEIGHT_F]);

  const float fov = DEG_TO_RAD*90.0f;
  GBufferAll samples[GBUFFER_SAMPLES];

  const float sizeInvX = 1.0f / (float)(m_width);
  const float sizeInvY = 1.0f / (float)(m_width);

  for (int i = 0; i < GBUFFER_SAMPLES; i++)
  {
    float4 lensOffs = make_float4(qmc[i].x, qmc[i].y, 0, 0);
    lensOffs.x = sizeInvX * (lensOffs.x + (float)x);
    lensOffs.y = sizeInvY * (lensOffs.y + (float)y);

    float  fx, fy;
    float3 ray_pos, ray_dir;
    MakeEyeRayFromF4Rnd(lensOffs, a_globals, &ray_pos, &ray_dir, &fx, &fy);

    /* gbufferSample */
    GBufferAll result;
    initGBufferAll(&result);
    const Lite_Hit liteHit = ref_rayTrace(ray_pos, ray_dir, trees);
    if (HitNone(liteHit))
    {
      result.data1.rgba = make_float4(0, 0, 0, 1);
    }
    else
    {
      const SurfaceHit surfHit = ref_evalSurface(ray_pos, ray_dir, liteHit, in_matrices, in_geomStorage, a_globals);
      __global const PlainMaterial* pHitMaterial = materialAt(a_globals, in_mtlStorage, surfHit.matId);
      ProcTextureList ptl;
      InitProcTextureList(&ptl);
      result.data1.depth    = liteHit.t;
      result.data1.norm     = surfHit.normal;
      result.data1.rgba     = to_float4(materialEvalDiffuse(pHitMaterial, ray_dir, surfHit.normal, surfHit.texCoord, a_globals, in_texStorage, &ptl), 0.0f);
      result.data1.matId    = surfHit.matId;
      result.data1.coverage = 1.0f;
      result.data2.texCoord = surfHit.texCoord;
      result.data2.objId    = liteHit.geomId;
      result.data2.instId   = liteHit.instId;
    }
    samples[i] = result;
  }

  float minDiff   = 100000000.0f;
  int   minDiffId = 0;
  for (int i = 0; i < GBUFFER_SAMPLES; i++)
  {
    float diff     = 0.0f;
    float coverage = 0.0f;
    for (int j = 0; j < GBUFFER_SAMPLES; j++)
    {
      const float thisDiff = gbuffDiff(samples[i], samples[j], fov, (float)(m_width), (float)(m_height));
      diff += thisDiff;
      if (thisDiff < 1.0f)
        coverage += 1.0f;
    }
    coverage *= (1.0f / (float)GBUFFER_SAMPLES);
    samples[i].data1.coverage = coverage;
    if (diff < minDiff)
    {
      minDiff   = diff;
      minDiffId = i;
    }
  }

  const GBufferAll g = samples[minDiffId];
  out1[p] = packGBuffer1(g.data1);
  out2[p] = packGBuffer2(g.data2);
  __global float* o = raw14 + (size_t)p*14;
  o[0] = g.data1.depth; o[1] = g.data1.norm.x; o[2] = g.data1.norm.y; o[3] = g.data1.norm.z;
  o[4] = g.data1.rgba.x; o[5] = g.data1.rgba.y; o[6] = g.data1.rgba.z; o[7] = g.data1.rgba.w;
  o[8] = as_float(g.data1.matId); o[9] = g.data1.coverage; o[10] = g.data2.texCoord.x; o[11] = g.data2.texCoord.y;
  o[12] = as_float(g.data2.objId); o[13] = as_float(g.data2.instId);
}

/* The wavefront layer keeps SurfaceHit records in four planes with the flat normal and the tangent frame packed to 2 x 16 bit
 * (WriteSurfaceHit / ReadSurfaceHit, cglobals.h:2537-2585).  This unpacks a plane set written by the reference's ComputeHit with the
 * reference's own reader into the 24-float record of ref_eval_surface, so that a test can hand the stage kernels' very inputs to the
 * build's stage functions. */
__kernel void ref_read_surface_hit(__global const float4* in_surfaceHit, __global float* out24, int n)
{
  const int i = get_global_id(0);
  if (i >= n) return;
  __global float* r = out24 + i * 24;
  for (int k = 0; k < 24; k++) r[k] = 0.0f;
  SurfaceHit sh;
  ReadSurfaceHit(in_surfaceHit, i, n, &sh);
  r[0] = sh.pos.x; r[1] = sh.pos.y; r[2] = sh.pos.z; r[3] = sh.normal.x; r[4] = sh.normal.y; r[5] = sh.normal.z;
  r[6] = sh.flatNormal.x; r[7] = sh.flatNormal.y; r[8] = sh.flatNormal.z; r[9] = sh.tangent.x; r[10] = sh.tangent.y; r[11] = sh.tangent.z;
  r[12] = sh.biTangent.x; r[13] = sh.biTangent.y; r[14] = sh.biTangent.z; r[15] = sh.texCoord.x; r[16] = sh.texCoord.y;
  r[17] = as_float(sh.matId); r[18] = sh.t; r[19] = sh.sRayOff; r[20] = sh.hfi ? 1.0f : 0.0f;
}
