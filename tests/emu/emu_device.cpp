// emu_device.cpp -- the HIP device functions (hydracore_amd/csrc/hk_*.h) compiled for the HOST with
// -DHK_HOST_EMU -fsanitize=address,undefined.  TEST INFRASTRUCTURE: GPU sanitizers are not available on the pool, so
// out-of-bounds reads, uninitialised values and UB in the device code are hunted here, on CPU, with the same scene
// buffers; the results are also compared with the oracle (tests/test_emu_device.py).  Not part of the product.
#define HK_HOST_EMU 1
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime.h>
#include <vector>
#include "../../hydracore_amd/csrc/hk_common.h"
#include "../../hydracore_amd/csrc/hk_trace.h"
#include "../../hydracore_amd/csrc/hk_shading.h"
#include "../../hydracore_amd/csrc/hk_bidir.h"
#include "../../hydracore_amd/csrc/hk_gbuffer.h"
#include "emu_integrator.h"

struct EmuScene {   // mirrors tests/oracle_lib.OrcScene field for field
  const int* globals; const float* matStorage; const int* texStorage; const float* geomStorage; const float* pdfStorage;
  const float* bvh; const float* tris; int haveInst; const float* instMatrices; const int* instLightInstId; int instNum;
  const int* remapLists; int remapListsSize; const int* remapTable; int remapTableSize; const int* remapInst; int remapInstSize;
  int treesNum; const float* bvhN[3]; const float* trisN[3]; int haveInstN[3]; const unsigned* alpha[4];
  const int* texAuxStorage;
};

static SceneDev to_dev(const EmuScene* e) {
  SceneDev s;
  s.globals = e->globals;
  s.matStorage = reinterpret_cast<const float4*>(e->matStorage);
  s.texStorage = reinterpret_cast<const int4*>(e->texStorage);
  s.geomStorage = reinterpret_cast<const float4*>(e->geomStorage);
  s.pdfStorage = reinterpret_cast<const float4*>(e->pdfStorage);
  s.bvh = reinterpret_cast<const float4*>(e->bvh);
  s.tris = reinterpret_cast<const float4*>(e->tris);
  s.haveInst = e->haveInst;
  s.instMatrices = reinterpret_cast<const float4*>(e->instMatrices);
  s.instLightInstId = e->instLightInstId;
  s.instNum = e->instNum;
  s.remapLists = e->remapListsSize > 0 ? e->remapLists : nullptr; s.remapListsSize = e->remapListsSize;
  s.remapTable = e->remapTableSize > 0 ? e->remapTable : nullptr; s.remapTableSize = e->remapTableSize;
  s.remapInst = e->remapInstSize > 0 ? e->remapInst : nullptr;    s.remapInstSize = e->remapInstSize;
  s.ptlIds = nullptr; s.ptlVals = nullptr; s.ptlStride = 0; s.ptlMax = 0; s.ptlSlot = -1;   // no procedural texture lists in the emulation
  s.srgbLut = nullptr;
  s.alpha = reinterpret_cast<const uint2*>(e->alpha[0]);
  s.matBase = e->matStorage;
  s.matTable = e->globals ? e->globals + e->globals[HG_MAT_TABLE_OFFS] : nullptr;   // traversal-only callers pass no globals
  s.lightsBase = e->globals ? reinterpret_cast<const float*>(e->globals + e->globals[HG_LIGHTS_OFFS]) : nullptr;
  s.texTable = e->globals ? e->globals + e->globals[HG_TEX_TABLE_OFFS] : nullptr;
  s.texAuxStorage = reinterpret_cast<const int4*>(e->texAuxStorage);
  s.texAuxTable = e->globals ? e->globals + e->globals[HG_TEXAUX_TABLE_OFFS] : nullptr;
  s.hdr = e->globals;
  s.lselRev = e->globals ? reinterpret_cast<const float*>(e->globals + e->globals[HG_LSEL_REV_OFFS]) : nullptr;
  return s;
}

extern "C" {

void emu_trace(const EmuScene* e, int n, const float* pos4, const float* dir4, HydraLiteHit* hits, unsigned* counters4, int anyhit, const float* tfar, float* vis) {
  const SceneDev s = to_dev(e);
  std::vector<int> lds(HK_LDS_DEPTH * HK_TRACE_BLOCK, 0);
  for (int i = 0; i < n; i++) {
    HkStack st;
    st.init(lds.data(), i % HK_TRACE_BLOCK);
    TravCounters c = {0, 0, 0, 0, 0};
    const f3 p = mk3(pos4[4 * i], pos4[4 * i + 1], pos4[4 * i + 2]), d = mk3(dir4[4 * i], dir4[4 * i + 1], dir4[4 * i + 2]);
    if (anyhit) {
      HydraLiteHit h = hk_miss_hit();
      h.t = tfar[i];
      h = hk_traverse<true, true>(make_bvh_view(s.bvh, 0, s.tris, 0), s.haveInst != 0, p, d, 0.0f, h, st, c);
      vis[i] = (h.primId != -1) ? 0.0f : 1.0f;
    } else {
      BvhView bv = make_bvh_view(s.bvh, 0, s.tris, 0);
      bv.alpha = s.alpha; bv.texTable = s.texTable; bv.texStorage = s.texStorage; bv.srgbLut = nullptr;
      HydraLiteHit h = (s.alpha != nullptr) ? hk_traverse<false, true, true>(bv, s.haveInst != 0, p, d, 0.0f, hk_miss_hit(), st, c)
                                            : hk_traverse<false, true, false>(bv, s.haveInst != 0, p, d, 0.0f, hk_miss_hit(), st, c);
      for (int tr = 1; tr < e->treesNum && tr < 4; tr++) {   // further trees carry the hit on (IntegratorCommon::rayTrace, Common.cpp:128-150)
        if (e->bvhN[tr - 1] == nullptr) continue;
        BvhView bt = make_bvh_view(reinterpret_cast<const float4*>(e->bvhN[tr - 1]), 0, reinterpret_cast<const float4*>(e->trisN[tr - 1]), 0);
        bt.alpha = reinterpret_cast<const uint2*>(e->alpha[tr]); bt.texTable = s.texTable; bt.texStorage = s.texStorage; bt.srgbLut = nullptr;
        h = (bt.alpha != nullptr) ? hk_traverse<false, true, true>(bt, e->haveInstN[tr - 1] != 0, p, d, 0.0f, h, st, c)
                                  : hk_traverse<false, true, false>(bt, e->haveInstN[tr - 1] != 0, p, d, 0.0f, h, st, c);
      }
      hits[i] = h;
    }
    if (counters4) { counters4[4 * i] = c.quads; counters4[4 * i + 1] = c.insts; counters4[4 * i + 2] = c.tris; counters4[4 * i + 3] = c.leaves; }
  }
}

void emu_path_trace(const EmuScene* e, int n, const float* pos4, const float* dir4, unsigned* rng2, float* color4) {
  const SceneDev s = to_dev(e);
  std::vector<int> lds(HK_LDS_DEPTH * HK_TRACE_BLOCK, 0);
  for (int i = 0; i < n; i++) {
    HkStack st;
    st.init(lds.data(), i % HK_TRACE_BLOCK);
    RandomGen gen; gen.x = rng2[2 * i]; gen.y = rng2[2 * i + 1];
    float rays = 0.0f;
    const f3 c = hk_path_trace_one(s, st, mk3(pos4[4 * i], pos4[4 * i + 1], pos4[4 * i + 2]), mk3(dir4[4 * i], dir4[4 * i + 1], dir4[4 * i + 2]), gen, rays);
    color4[4 * i] = c.x; color4[4 * i + 1] = c.y; color4[4 * i + 2] = c.z; color4[4 * i + 3] = rays;
    rng2[2 * i] = gen.x; rng2[2 * i + 1] = gen.y;
  }
}

void emu_eye_rays(const EmuScene* e, int n, int w, int h, const int* xy, const float* offs4, float* pos4, float* dir4) {
  const SceneDev s = to_dev(e);
  for (int i = 0; i < n; i++) {
    f3 p, d;
    MakeRandEyeRay(xy[2 * i], xy[2 * i + 1], w, h, make_float4(offs4[4 * i], offs4[4 * i + 1], offs4[4 * i + 2], offs4[4 * i + 3]), s, p, d);
    pos4[4 * i] = p.x; pos4[4 * i + 1] = p.y; pos4[4 * i + 2] = p.z; pos4[4 * i + 3] = 0;
    dir4[4 * i] = d.x; dir4[4 * i + 1] = d.y; dir4[4 * i + 2] = d.z; dir4[4 * i + 3] = 0;
  }
}

// row f3 building blocks (hk_bidir.h), same record layouts as the stage calls of the C-ABI
void emu_bidir(const EmuScene* e, int n, const int* lightIds, const float* rands4, const float* cosTheta, const float* pos4, const float* norm4,
               const float* disk2, const float* values, const float* rands2, float p2, float p1, float* fwd16, float* pdf4, float* cam8, float* mut) {
  const SceneDev s = to_dev(e);
  for (int i = 0; i < n; i++) {
    LightSampleFwd sam;
    LightSampleForward(s, lightAt(s, lightIds[i]), make_float4(rands4[4 * i], rands4[4 * i + 1], rands4[4 * i + 2], rands4[4 * i + 3]), 0.0f, sam);
    float* o = fwd16 + 16 * size_t(i);
    o[0] = sam.pos.x; o[1] = sam.pos.y; o[2] = sam.pos.z; o[3] = sam.dir.x; o[4] = sam.dir.y; o[5] = sam.dir.z;
    o[6] = sam.norm.x; o[7] = sam.norm.y; o[8] = sam.norm.z; o[9] = sam.color.x; o[10] = sam.color.y; o[11] = sam.color.z;
    o[12] = sam.pdfA; o[13] = sam.pdfW; o[14] = sam.cosTheta; o[15] = sam.isPoint ? 1.0f : 0.0f;
    const LightPdfFwd p = lightPdfFwd(s, lightAt(s, lightIds[i]), mk3(0.0f, 0.0f, 1.0f), cosTheta[i]);
    pdf4[4 * i] = p.pdfA; pdf4[4 * i + 1] = p.pdfW; pdf4[4 * i + 2] = p.pickProb; pdf4[4 * i + 3] = 0.0f;
    f3 camDir; float zDepth;
    const f3 hp = mk3(pos4[4 * i], pos4[4 * i + 1], pos4[4 * i + 2]);
    const float f = CameraImageToSurfaceFactor(s, hp, mk3(norm4[4 * i], norm4[4 * i + 1], norm4[4 * i + 2]), mk2(disk2[2 * i], disk2[2 * i + 1]), camDir, zDepth);
    const f2 scr = worldPosToScreenSpace(s, hp);
    float* c = cam8 + 8 * size_t(i);
    c[0] = f; c[1] = camDir.x; c[2] = camDir.y; c[3] = camDir.z; c[4] = zDepth; c[5] = scr.x; c[6] = scr.y; c[7] = 0.0f;
    mut[i] = MutateKelemen(values[i], mk2(rands2[2 * i], rands2[2 * i + 1]), p2, p1);
  }
}

// IntegratorMMLT::F through the wavefront functions of hk_bidir.h: the stage order hydra_hip.hip runs on the device, with this file's
// per-ray traversal between the stages.  xvec: n rows of `stride` floats; out8 as hydra_hip_stage_mmlt_f.
void emu_mmlt_f(const EmuScene* e, int n, const int* depth, const float* xvec, int stride, float* out8) {
  const SceneDev s = to_dev(e);
  int maxD = 1;
  for (int i = 0; i < n; i++) maxD = depth[i] > maxD ? depth[i] : maxD;
  std::vector<float> st(size_t(mmltPlanes(maxD)) * n, 0.0f), x(size_t(mmltStride(maxD)) * n, 0.0f);
  for (int i = 0; i < n; i++) for (int j = 0; j < HK_MMLT_HEAD + HK_MMLT_PER_BOUNCE * depth[i]; j++) x[size_t(j) * n + i] = xvec[size_t(i) * stride + j];
  std::vector<float4> rayPos(2 * size_t(n)), rayDir(2 * size_t(n)), eyePos(n), eyeDir(n), shPos(n), shDir(n);
  std::vector<HydraLiteHit> hits(2 * size_t(n)), eyeHit(n);
  std::vector<float> shVis(n), tfar(n);
  MmltView v;
  v.n = n; v.maxD = maxD; v.st = st.data(); v.x = x.data(); v.depth = depth;
  v.eyePos = eyePos.data(); v.eyeDir = eyeDir.data(); v.eyeHit = eyeHit.data(); v.shPos = shPos.data(); v.shDir = shDir.data(); v.shVis = shVis.data(); v.out8 = out8;
  for (int i = 0; i < n; i++) { bool ca, la; mmltBegin(s, v, i, rayPos[i], rayDir[i], ca, rayPos[n + i], rayDir[n + i], la); }
  for (int k = 1; k <= maxD; k++) {   // plain arrays here: a finished sub-path keeps a ray that misses the scene
    emu_trace(e, 2 * n, reinterpret_cast<const float*>(rayPos.data()), reinterpret_cast<const float*>(rayDir.data()), hits.data(), nullptr, 0, nullptr, nullptr);
    for (int i = 0; i < n; i++) {
      float4 np, nd;
      mmltCameraStep(s, v, i, k, rayPos[i], rayDir[i], hits[i], np, nd); rayPos[i] = np; rayDir[i] = nd;
      mmltLightStep(s, v, i, k, rayPos[n + i], rayDir[n + i], hits[n + i], np, nd); rayPos[n + i] = np; rayDir[n + i] = nd;
    }
  }
  for (int i = 0; i < n; i++) { mmltConnectBegin(s, v, i); tfar[i] = shPos[i].w; }
  emu_trace(e, n, reinterpret_cast<const float*>(eyePos.data()), reinterpret_cast<const float*>(eyeDir.data()), eyeHit.data(), nullptr, 0, nullptr, nullptr);
  emu_trace(e, n, reinterpret_cast<const float*>(shPos.data()), reinterpret_cast<const float*>(shDir.data()), nullptr, nullptr, 1, tfar.data(), shVis.data());
  for (int i = 0; i < n; i++) mmltConnectEnd(s, v, i);
}

// the Markov chains through hk_bidir.h's chain functions (what k_mmlt_mutate / k_mmlt_accept run), F through emu_mmlt_f above
void emu_mmlt_run(const EmuScene* e, int n, unsigned* gens4, const int* depth, int mutations, int w, float* image4, float* chains6, float* xrows, int stride, int* accepted) {
  int maxD = 1;
  for (int i = 0; i < n; i++) maxD = depth[i] > maxD ? depth[i] : maxD;
  const int planes = mmltStride(maxD);
  std::vector<float> ch(size_t(CH_PLANES) * n, 0.0f), xCur(size_t(planes) * n, 0.0f), xNew(size_t(planes) * n, 0.0f), rows(size_t(n) * planes, 0.0f), out8(size_t(n) * 8);
  MmltChains c;
  c.n = n; c.maxD = maxD; c.ch = ch.data(); c.depth = depth; c.xCur = xCur.data(); c.xNew = xNew.data();
  for (int i = 0; i < n; i++) {
    mmltInitChain(c, i, 0);
    RandomGen g; g.x = gens4[4 * i]; g.y = gens4[4 * i + 1]; mchSetGen(c, CH_GEN, i, g);
    g.x = gens4[4 * i + 2]; g.y = gens4[4 * i + 3]; mchSetGen(c, CH_GEN2, i, g);
    for (int j = 0; j < mmltStride(depth[i]); j++) { xCur[size_t(j) * n + i] = xrows[size_t(i) * stride + j]; xNew[size_t(j) * n + i] = xCur[size_t(j) * n + i]; }
  }
  auto evalNew = [&]() {
    for (int i = 0; i < n; i++) for (int j = 0; j < planes; j++) rows[size_t(i) * planes + j] = xNew[size_t(j) * n + i];
    emu_mmlt_f(e, n, depth, rows.data(), planes, out8.data());
  };
  evalNew();
  for (int i = 0; i < n; i++) mmltSeedChain(c, i, out8.data());
  for (int k = 0; k < mutations; k++) {
    for (int i = 0; i < n; i++) mmltMutate(c, i);
    evalNew();
    for (int i = 0; i < n; i++) mmltAcceptReject(c, i, out8.data(), 1.0f, image4, w);
  }
  for (int i = 0; i < n; i++) {
    float* c6 = chains6 + 6 * size_t(i);
    c6[0] = mch(c, CH_Y, i); c6[1] = mch(c, CH_COLOR, i); c6[2] = mch(c, CH_COLOR + 1, i); c6[3] = mch(c, CH_COLOR + 2, i); c6[4] = mch(c, CH_XS, i); c6[5] = mch(c, CH_YS, i);
    accepted[i] = int(mch(c, CH_ACCEPTED, i));
    RandomGen g = mchGen(c, CH_GEN, i); gens4[4 * i] = g.x; gens4[4 * i + 1] = g.y;
    g = mchGen(c, CH_GEN2, i); gens4[4 * i + 2] = g.x; gens4[4 * i + 3] = g.y;
    for (int j = 0; j < mmltStride(depth[i]); j++) xrows[size_t(i) * stride + j] = xCur[size_t(j) * n + i];
  }
}

// IHWLayer::EvalGBuffer through hk_gbuffer.h: the rays, the per-sample record and the packing the device kernels use, with the serial
// form of the cluster vote (the device runs it one wavefront per pixel).  Whole frame; out = packGBuffer1, packGBuffer2, raw record.
void emu_gbuffer(const EmuScene* e, int w, int h, float* data1, float* data2, float* raw14) {
  const SceneDev s = to_dev(e);
  std::vector<float4> pos(HK_GBUFFER_SAMPLES), dir(HK_GBUFFER_SAMPLES);
  std::vector<HydraLiteHit> hits(HK_GBUFFER_SAMPLES);
  GBufferSample samples[HK_GBUFFER_SAMPLES];
  for (int pixel = 0; pixel < w * h; pixel++) {
    for (int k = 0; k < HK_GBUFFER_SAMPLES; k++) {
      f3 rp, rd;
      gbufferEyeRay(s, pixel % w, pixel / w, k, w, rp, rd);
      pos[k] = mk4(rp, 0.0f); dir[k] = mk4(rd, 0.0f);
    }
    emu_trace(e, HK_GBUFFER_SAMPLES, reinterpret_cast<const float*>(pos.data()), reinterpret_cast<const float*>(dir.data()), hits.data(), nullptr, 0, nullptr, nullptr);
    for (int k = 0; k < HK_GBUFFER_SAMPLES; k++) samples[k] = gbufferSampleOf(s, xyz(pos[k]), xyz(dir[k]), hits[k]);
    const GBufferSample g = gbufferResolve(samples, float(w), float(h));
    const float4 p1 = packGBuffer1(g), p2 = packGBuffer2(g);
    float* d1 = data1 + 4 * size_t(pixel), *d2 = data2 + 4 * size_t(pixel), *o = raw14 + 14 * size_t(pixel);
    d1[0] = p1.x; d1[1] = p1.y; d1[2] = p1.z; d1[3] = p1.w; d2[0] = p2.x; d2[1] = p2.y; d2[2] = p2.z; d2[3] = p2.w;
    o[0] = g.depth; o[1] = g.norm.x; o[2] = g.norm.y; o[3] = g.norm.z; o[4] = g.rgba.x; o[5] = g.rgba.y; o[6] = g.rgba.z; o[7] = g.rgba.w;
    o[8] = as_float(g.matId); o[9] = g.coverage; o[10] = g.texCoord.x; o[11] = g.texCoord.y; o[12] = as_float(g.objId); o[13] = as_float(g.instId);
  }
}

}  // extern "C"
