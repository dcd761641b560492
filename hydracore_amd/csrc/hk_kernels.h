// hk_kernels.h -- the heavy kernels of libhydra_hip.so as templates, and the launchers that reach them.
//
// The library is built from several translation units so that its device code compiles in parallel (one TU with every
// instantiation took 2.5 minutes; the largest of these takes a fraction of that) and so that a change to the traversal kernels does not
// recompile the 40 instantiations of the bounce kernel:
//   hydra_hip.hip           host side (C-ABI of include/hydra_hip.h), the small non-template kernels, dispatch
//   hk_inst_trace.hip       k_trace / k_shadow / k_trace_dyn (rows a/T1, a/T2)
//   hk_inst_bounce_*.hip    k_bounce<W, F, STG> by feature set (rows a/H1 ... a/S2, a/Q1), k_hit / k_shade (the split form)
//   hk_inst_mmlt_*.hip      k_mmlt_step / k_mmlt_connect_begin / k_mmlt_connect_end by feature set (row f3)
// A kernel template is instantiated only in the TU whose launcher names it; hydra_hip.hip calls the launchers declared at the end of
// this file (plain functions taking one argument record), so no TU ever launches a kernel another TU holds.
#pragma once
#include <string>
#include <vector>
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/hydra_hip.h"
#include "hk_common.h"
#include "hk_trace.h"
#include "hk_shading.h"
#include "hk_bidir.h"

// ================================================================================================ device state
struct PathState {   // S arrays
  float4* pos4; float4* dir4; float4* thr4; float4* acc4; uint2* rng2;
  float4* pend4;      // fused form only: throughput * unoccluded next-event estimate of the previous bounce xyz | unused
};
struct MidState {    // M arrays: survivors of K_hit, consumed by shadow + shade
  float4* dir4; float4* thr4; float4* acc4; uint2* rng2;
  float4* surfA;      // hit position xyz | matId
  float4* surfB;      // shading normal xyz | texCoord.x
  float4* recC;       // direction to the light sample xyz | texCoord.y
  float4* recD;       // light radiance xyz | light pdf (negative when the sample is a point light)
  float4* recE;       // light pick prob | lightOffset | gid | hit from inside (0/1)
  float4* shadowOrg;  // shadow ray origin xyz | t_far
  float*  vis;        // shadow result
};

#define HK_MAX_DEPTH 64
#define HK_TT_ROW 12   // words per bounce in travTotals: 2 kernels x (rays, quads, insts, leaves, tris, out-of-range fetches)
#ifndef HK_TRACE_MIN_BLOCKS
#define HK_TRACE_MIN_BLOCKS 1   // resident 128-thread blocks per CU the traversal kernels are register-budgeted for
#endif

HK_DEV int wave_compact_index(bool alive, uint32_t* counter) {
  const unsigned long long mask = __ballot(alive);
  const int lane = int(__lane_id());
  int base = 0;
  if (mask != 0ull) {
    const int leader = __ffsll((long long)mask) - 1;
    if (lane == leader) base = int(atomicAdd(counter, uint32_t(__popcll(mask))));
    base = __shfl(base, leader);
  }
  return base + __popcll(mask & ((1ull << lane) - 1ull));
}

// SegQ / segq_iter (the segmented path queues): hk_common.h
// T1 -- closest hit for every live path (kernel_RayTrace).  ALPHA: the tree carries an alpha table (BVH4InstTraverseAlpha,
// ctrace.h:1297-1520).  carry: this launch walks one of trees 1..3 and starts from the hit the earlier trees left in `hits`
// (IntegratorCommon::rayTrace loops over the trees with one running Lite_Hit, Common.cpp:128-150; per-ray counters add up).
template <bool COUNT, bool ALPHA>
__global__ void __launch_bounds__(HK_TRACE_BLOCK, HK_TRACE_MIN_BLOCKS) k_trace(SceneDev s, SegQ q,
                                                           const float4* __restrict__ pos4, const float4* __restrict__ dir4,
                                                           HydraLiteHit* __restrict__ hits, uint32_t* __restrict__ counters3,
                                                           unsigned long long* __restrict__ totals5, int carry) {
  __shared__ int ldsStack[HK_LDS_DEPTH * HK_TRACE_BLOCK];
  const SegIter it = segq_iter(q);
  HkStack st;
  st.init(ldsStack, threadIdx.x);
  BvhView bv = make_bvh_view(s.bvh, s.bvhBytes, s.tris, s.trisBytes, s.leafEnc != 0);
  bv.alpha = s.alpha; bv.texTable = s.texTable; bv.texStorage = s.texStorage; bv.srgbLut = s.srgbLut;
  for (int idx = it.first; idx < it.count; idx += it.step) {
    const int i = it.base + idx;
    const f3 pos = xyz(pos4[i]), dir = xyz(dir4[i]);
    TravCounters c = {0, 0, 0, 0, 0};
    HydraLiteHit h0 = hk_miss_hit();
    if (carry) { const float4 p = reinterpret_cast<const float4*>(hits)[i]; h0.t = p.x; h0.primId = as_int(p.y); h0.instId = as_int(p.z); h0.geomId = as_int(p.w); }
    const HydraLiteHit hit = hk_traverse<false, COUNT, ALPHA>(bv, s.haveInst != 0, pos, dir, 0.0f, h0, st, c);
    reinterpret_cast<float4*>(hits)[i] = make_float4(hit.t, as_float(hit.primId), as_float(hit.instId), as_float(hit.geomId));
    if (COUNT && counters3) {
      if (carry) { counters3[3 * i] += c.quads; counters3[3 * i + 1] += c.insts; counters3[3 * i + 2] += c.tris; }
      else { counters3[3 * i] = c.quads; counters3[3 * i + 1] = c.insts; counters3[3 * i + 2] = c.tris; }
    }
    if (COUNT && totals5) {   // algorithmic-work counters for the roofline byte model
      if (!carry) atomicAdd(totals5 + 0, 1ull);
      atomicAdd(totals5 + 1, (unsigned long long)c.quads); atomicAdd(totals5 + 2, (unsigned long long)c.insts);
      atomicAdd(totals5 + 3, (unsigned long long)c.leaves); atomicAdd(totals5 + 4, (unsigned long long)c.tris);
      if (c.oob) atomicAdd(totals5 + 5, (unsigned long long)c.oob);
    }
  }
}

// T2 -- any-hit visibility: origin xyz | t_far, direction xyz (kernel_ShadowTrace)
template <bool COUNT>
__global__ void __launch_bounds__(HK_TRACE_BLOCK, HK_TRACE_MIN_BLOCKS) k_shadow(SceneDev s, SegQ q,
                                                            const float4* __restrict__ org4, const float4* __restrict__ dir4, float* __restrict__ vis,
                                                            unsigned long long* __restrict__ totals5) {
  __shared__ int ldsStack[HK_LDS_DEPTH * HK_TRACE_BLOCK];
  const SegIter it = segq_iter(q);
  HkStack st;
  st.init(ldsStack, threadIdx.x);
  const BvhView bv = make_bvh_view(s.bvh, s.bvhBytes, s.tris, s.trisBytes, s.leafEnc != 0);
  for (int idx = it.first; idx < it.count; idx += it.step) {
    const int i = it.base + idx;
    const float4 o = org4[i];
    float v = 0.0f;
    if (o.w >= 0.0f) {   // t_far < 0 marks "no light sample": shadow = 0 (PT_Loop.cpp:175-178)
      HydraLiteHit h = hk_miss_hit();
      h.t = o.w;
      TravCounters c = {0, 0, 0, 0, 0};
      h = hk_traverse<true, COUNT>(bv, s.haveInst != 0, xyz(o), xyz(dir4[i]), 0.0f, h, st, c);
      v = (h.primId != -1) ? 0.0f : 1.0f;
      if (COUNT && totals5) {
        atomicAdd(totals5 + 0, 1ull); atomicAdd(totals5 + 1, (unsigned long long)c.quads); atomicAdd(totals5 + 2, (unsigned long long)c.insts);
        atomicAdd(totals5 + 3, (unsigned long long)c.leaves); atomicAdd(totals5 + 4, (unsigned long long)c.tris);
      if (c.oob) atomicAdd(totals5 + 5, (unsigned long long)c.oob);
      }
    }
    vis[i] = v;
  }
}

// T1/T2, persistent form: every lane that finishes its ray immediately fetches the next one from a device-side counter
// (one atomic per wave per refill), and a wave whose active-lane count drops below `minActive` suspends traversal to
// refill.  Keeps SIMD lanes busy when path lengths inside a wave diverge (secondary and shadow rays).  Results are
// written by ray index, so they are identical to the one-ray-per-lane kernels above.
// VOTE: the steps of the wave's rays are scheduled by wave vote (trav_run_vote, hk_trace.h) instead of the reference's loop nest;
// wq / wt / wi = the weights of the vote.
template <bool ANYHIT, bool COUNT, bool TOPTRIS = false, bool ALPHA = false, bool VOTE = false, bool UNORD = false>
__global__ void __launch_bounds__(HK_TRACE_BLOCK, (ANYHIT && !COUNT) ? HK_TRACE_MIN_WAVES_SHADOW : HK_TRACE_MIN_BLOCKS) k_trace_dyn(SceneDev s, SegQ q, uint32_t* __restrict__ fetchCounters,
                                                               const float4* __restrict__ a4, const float4* __restrict__ b4,
                                                               float4* __restrict__ outHits, float* __restrict__ outVis,
                                                               unsigned long long* __restrict__ totals5, int minActive, int raysPerLane, int wq, int wt, int wi) {
  constexpr int LDS_DEPTH = (ANYHIT && !COUNT) ? HK_LDS_DEPTH_SHADOW : HK_LDS_DEPTH;
  __shared__ int ldsStack[LDS_DEPTH * HK_TRACE_BLOCK];
  __shared__ float4 ldsTop[HK_TOP_QUADS * HK_TOP_STRIDE];
  __shared__ float4 ldsTri[TOPTRIS ? HK_TOP_TRIS * 3 : 1];   // the LDS-staged triangle packets: the leaves rays visit most (chosen at upload)
  const SegIter it = segq_iter(q);
  const int count = it.count, segBase = it.base;
  // the live count is only known on the device: when it is small, let only the first blocks of the segment take part so
  // that every lane still gets ~raysPerLane rays to refill from (a thinly spread queue degenerates to one ray per lane)
  if ((int(blockIdx.x) / q.nseg) * (HK_TRACE_BLOCK * raysPerLane) >= count) return;
  uint32_t* fetchCounter = fetchCounters + it.seg * HK_CSTRIDE;
  HkStackT<LDS_DEPTH> st;
  st.init(ldsStack, threadIdx.x);
  // the hottest quads of the tree (chosen at upload) go to LDS once per block; the node copy walked here names them by slot
  const bool useTop = (s.topCount > 0);
  if (useTop) {
    for (int i = threadIdx.x; i < s.topCount * 8; i += HK_TRACE_BLOCK) ldsTop[(i >> 3) * HK_TOP_STRIDE + (i & 7)] = s.bvhTop[size_t(s.topQuads[i >> 3]) * 8 + (i & 7)];
    if (TOPTRIS) for (int i = threadIdx.x; i < s.topTriCount * 3; i += HK_TRACE_BLOCK) ldsTri[i] = s.tris[s.topTriF4[i]];
    __syncthreads();
  }
  BvhView bv = make_bvh_view(useTop ? s.bvhTop : s.bvh, s.bvhBytes, s.tris, s.trisBytes, s.leafEnc != 0);
  bv.top = (const hk_lds_f4*)ldsTop;
  bv.topTri = (const hk_lds_f4*)ldsTri;
  if (ALPHA) { bv.alpha = s.alpha; bv.texTable = s.texTable; bv.texStorage = s.texStorage; bv.srgbLut = s.srgbLut; }
  const int rootLink = useTop ? (HK_TOP_FLAG | 0) : 1;
  TravState t;
  TravCounters c = {0, 0, 0, 0, 0};
  int rayIdx = -1;
  bool busy = false, queueEmpty = false;
  const int lane = int(__lane_id());
  const bool haveInst = s.haveInst != 0;
  // The wave's own pool of ray indices: it reserves `chunk` consecutive rays of its segment with ONE returning atomic (1.1-1.3 us with
  // every CU pulling, MI355X_MICROARCH.md "dequeue"; it used to be one per refill) and hands them to its idle lanes with ballot
  // arithmetic on wave-uniform (scalar) bookkeeping; the next chunk is reserved while up to 64 rays of the current one are unused.
  // Which wave traces which ray changes; results are written by ray index as before.  Measured (profiles/r03/ab_ray_pool_*.log): -1 %
  // closest-hit, -5 % shadow traversal.  Two further steps were measured and dropped: touching the next chunk's lines so that the rays'
  // loads hit L2 (8 more registers: one wave per SIMD less, slower), and loading a finished lane's next ray into its dead state
  // registers while the wave keeps stepping (slower: what makes frequent refills expensive is not their latency but that lanes which
  // start at different times want different steps -- see trav_run_vote -- so a wave does best refilling a third of its lanes at once).
  // chunk: 64..256 rays, at least ~8 chunks per participating wave so that the segment's tail stays balanced
  const int wavesOfSeg = (int(gridDim.x) / q.nseg) * (HK_TRACE_BLOCK / 64);
  int chunk = (count / (wavesOfSeg > 0 ? wavesOfSeg * 8 : 8)) & ~63;
  chunk = chunk < 64 ? 64 : (chunk > 256 ? 256 : chunk);
  int poolNext = 0, poolEnd = 0, nextBase = -1;   // [poolNext, poolEnd) is reserved and unused; nextBase >= 0: so is [nextBase, nextBase + chunk)
  while (true) {
    if (!queueEmpty) {
      const unsigned long long mask = __ballot(!busy);
      if (mask != 0ull) {
        const int n = __popcll(mask);
        const int avail = poolEnd - poolNext;
        if (nextBase < 0 && avail <= 64) {            // wave-uniform: look one chunk ahead
          int base = 0;
          if (lane == 0) base = int(atomicAdd(fetchCounter, uint32_t(chunk)));
          nextBase = __builtin_amdgcn_readfirstlane(base);
        }
        if (!busy) {
          const int r = __popcll(mask & ((1ull << lane) - 1ull));
          const int idx = (r < avail) ? poolNext + r : nextBase + (r - avail);
          if (idx < count) {
            const float4 a = a4[segBase + idx];
            HydraLiteHit h = hk_miss_hit();
            bool skip = false;
            if (ANYHIT) { h.t = a.w; skip = (a.w < 0.0f); }   // t_far < 0: no light sample => shadow = 0
            if (skip) outVis[segBase + idx] = 0.0f;
            else {
              trav_init(t, xyz(a), xyz(b4[segBase + idx]), h, rootLink);
              if (COUNT) { c.quads = c.insts = c.tris = c.leaves = c.oob = 0; }
              rayIdx = segBase + idx;
              busy = true;
            }
          }
        }
        if (n >= avail) { poolNext = nextBase + (n - avail); poolEnd = nextBase + chunk; nextBase = -1; }   // (n >= avail implies avail <= 64: the next chunk is reserved)
        else poolNext += n;
        if (poolNext >= count) queueEmpty = true;   // wave-uniform: reservations only grow, so nothing this wave can still reserve exists
      }
    }
    if (__ballot(busy) == 0ull) break;
    bool done = false;
    if (VOTE) {
      trav_run_vote<ANYHIT, COUNT, true, HkStackT<LDS_DEPTH>, TOPTRIS, ALPHA, UNORD>(t, busy, bv, haveInst, 0.0f, st, c, queueEmpty ? 0 : minActive, wq, wt, wi);
      done = busy && t.top < 0 && t.pend < 0;
    } else if (busy) done = trav_run<ANYHIT, COUNT, true, HkStackT<LDS_DEPTH>, TOPTRIS, ALPHA>(t, bv, haveInst, 0.0f, st, c, queueEmpty ? 0 : minActive);
    if (done) {
      if (ANYHIT) outVis[rayIdx] = (t.hit.primId != -1) ? 0.0f : 1.0f;
      else outHits[rayIdx] = make_float4(t.hit.t, as_float(t.hit.primId), as_float(t.hit.instId), as_float(t.hit.geomId));
      if (COUNT && totals5) {
        atomicAdd(totals5 + 0, 1ull); atomicAdd(totals5 + 1, (unsigned long long)c.quads); atomicAdd(totals5 + 2, (unsigned long long)c.insts);
        atomicAdd(totals5 + 3, (unsigned long long)c.leaves); atomicAdd(totals5 + 4, (unsigned long long)c.tris);
        if (c.oob) atomicAdd(totals5 + 5, (unsigned long long)c.oob);
      }
      busy = false;
    }
  }
}

// ---- per-path phases shared by the split (k_hit + k_shade) and the fused (k_bounce) kernels
struct LightPick {
  f3 shadowRayDir, color;
  float pdfSigned;       // light pdf, negative when the sample is a point light
  float pickProb;
  int lightOffset;       // < 0: no light sampled
  float4 shadowOrg;      // shadow ray origin xyz | t_far (t_far < 0: no shadow ray)
};

// H1 + E1 + E2 -- surface, environment/emission with MIS, termination (kernel_HitEnvironment, kernel_EvalSurface,
// kernel_EvalEmission).  Returns true when the path goes on (surf valid); false when it ended with radiance `finalColor`.
// `td` / `instInv`: the triangle record and the instance matrix of the hit, fetched by the caller (valid when HitSome(hit))
// E2 -- what a hit surface sends back along the ray, MIS-weighted against the light's pdf when the surface belongs to a light
// (kernel_EvalEmission, PT_Loop.cpp:86-139).  Returns true when the path ends on an emitter; `currColor` is then its radiance.
template <int F = HK_FEAT_ALL>
HK_DEV bool emission_phase(const SceneDev& s, const f3 ray_pos, const f3 ray_dir, const uint32_t flags, const float prevPdf, const bool prevSpecular,
                           const int hitInstId, const SurfaceHit& surf, const float* mat, f3& currColor) {
  const int lightOffset0 = (s.hdr[HG_LIGHTS_NUM] != 0) ? s.instLightInstId[hitInstId] : -1;
  const float* pLightHit = lightAt(s, lightOffset0);
  const f3 emission = emissionEval<F>(s, ray_pos, ray_dir, surf, flags, prevSpecular, pLightHit, mat);
  if (!(dot(emission, emission) > 1e-3f)) return false;
  if (pLightHit != nullptr) {
    const float lgtPdf = pLightHit[HL_PICK_PROB_REV] * lightEvalPDF<F>(s, pLightHit, ray_pos, ray_dir, surf.pos, surf.normal, surf.texCoord);
    float misWeight = misWeightHeuristic(prevPdf, lgtPdf);
    if (prevSpecular) misWeight = 1.0f;
    currColor = emission * misWeight;
  } else
    currColor = emission;
  return true;
}
// back-plate scenes (hk_shading.h, environmentColorExtended): the pixel a path was generated for, from the path's id (k_raygen: gid = stream * nOwned + index of the pixel
// in this rank's list); ownedPixels == nullptr: path i plays pixel i (hydra_hip_stage_path_trace)
struct ScreenOfPath { const int* ownedPixels; int nOwned, width; };
HK_DEV void screen_of_path(const ScreenOfPath& sp, int gid, int& x, int& y) {
  const int pixel = (sp.ownedPixels != nullptr && sp.nOwned > 0) ? sp.ownedPixels[gid % sp.nOwned] : gid;
  const int w = sp.width > 0 ? sp.width : 1;
  x = pixel % w; y = pixel / w;
}
template <int F = HK_FEAT_ALL>
HK_DEV bool surface_phase_with(const SceneDev& s, int depth, int maxDepth, const float4& pos4, const float4& dir4, const float4& thr4, const float4& acc4,
                               const HydraLiteHit& hit, const TriData& td, const m44& instInv, SurfaceHit& surf, f3& finalColor, const ScreenOfPath sp = {nullptr, 0, 0}) {
  const f3 ray_pos = xyz(pos4), ray_dir = xyz(dir4);
  const uint32_t flags = uint32_t(as_int(dir4.w));
  f3 currColor = mk3(0, 0, 0);
  bool done = false;
  if (!HitSome(hit)) {              // kernel_HitEnvironment, PT_Loop.cpp:23-33
    if ((F & HK_FEAT_RARE_LIGHTS) && haveBackPlate(s)) {
      int sx, sy;
      screen_of_path(sp, as_int(pos4.w), sx, sy);
      currColor = environmentColorExtended<F>(s, ray_pos, ray_dir, thr4.w, acc4.w != 0.0f, flags, sx, sy);
    } else
      currColor = environmentColor<F>(s, ray_dir, thr4.w, acc4.w != 0.0f, flags);
    done = true;
  }
  else {
    surf = evalSurfaceWith(s, ray_pos, ray_dir, hit, td, instInv);
    if (emission_phase<F>(s, ray_pos, ray_dir, flags, thr4.w, acc4.w != 0.0f, hit.instId, surf, materialAt(s, surf.matId), currColor)) done = true;
    else if (depth >= maxDepth - 1) done = true;
  }
  if (done) {
    finalColor = xyz(acc4) + (xyz(thr4) * currColor);   // kernel_AddLastBouceContrib
    return false;
  }
  return true;
}

template <int F = HK_FEAT_ALL>
HK_DEV bool surface_phase(const SceneDev& s, int depth, int maxDepth, const float4& pos4, const float4& dir4, const float4& thr4, const float4& acc4,
                          const HydraLiteHit& hit, SurfaceHit& surf, f3& finalColor, const ScreenOfPath sp = {nullptr, 0, 0}) {
  TriData td;
  m44 instInv;
  if (HitSome(hit)) { instInv = load_m44(s.instMatrices + size_t(hit.instId) * 4); td = fetchTri(s, hit); }
  return surface_phase_with<F>(s, depth, maxDepth, pos4, dir4, thr4, acc4, hit, td, instInv, surf, finalColor, sp);
}

// L1 + L2 -- light pick + sample, shadow ray (kernel_LightSelect, kernel_LightSample) with the random numbers handed in: rl = the
// four numbers of rndLight (crandom.h:404-418), pickRand = the one that picks the light (the CPU path passes rl.z, PT_Loop.cpp:141-151)
template <int F = HK_FEAT_ALL>
HK_DEV void light_phase_with(const SceneDev& s, const SurfaceHit& surf, const float4 rl, const float pickRand, LightPick& lp, ShadowSample& sam) {
  lp.pickProb = 1.0f;
  lp.lightOffset = SelectRandomLightRev(pickRand, s, lp.pickProb);
  lp.shadowRayDir = mk3(0, 0, 0);
  lp.shadowOrg = make_float4(0, 0, 0, -1.0f);
  sam.pos = mk3(0, 0, 0); sam.color = mk3(0, 0, 0); sam.pdf = 0.0f; sam.isPoint = false;
  if (lp.lightOffset >= 0) {
    LightSampleRev<F>(s, lightAt(s, lp.lightOffset), mk3(rl.x, rl.y, rl.z), surf.pos, sam);   // clight.h:1561-1610
    lp.shadowRayDir = normalize(sam.pos - surf.pos);
    const f3 shadowRayPos = OffsShadowRayPos(surf.pos, surf.normal, lp.shadowRayDir, surf.sRayOff);
    lp.shadowOrg = mk4(shadowRayPos, length(shadowRayPos - sam.pos) * 0.995f);
  }
  lp.color = sam.color;
  lp.pdfSigned = sam.isPoint ? -sam.pdf : sam.pdf;
}
template <int F = HK_FEAT_ALL>
HK_DEV void light_phase(const SceneDev& s, const SurfaceHit& surf, RandomGen& gen, LightPick& lp) {
  const float4 rl = rndFloat4_Pseudo(gen);   // rndLight, crandom.h:404-418
  ShadowSample sam;
  light_phase_with<F>(s, surf, rl, rl.z, lp, sam);
}

// S1 -- next-event estimate before visibility (kernel_Shade): explicitColor of PT_Loop.cpp:190-215 is this value * shadow
template <int F = HK_FEAT_ALL>
HK_DEV f3 direct_light_unoccluded(const SceneDev& s, const float* mat, const SurfaceHit& surf, const f3 ray_dir,
                                  const f3 shadowRayDir, const f3 lightColor, const float pdfSigned, const float lightPickProb) {
  const f3 surfNormal = surf.normal;
  ShadeContext sc;
  sc.l = shadowRayDir; sc.v = ray_dir * (-1.0f); sc.n = surfNormal; sc.tc = surf.texCoord;
  if (F & (HK_FEAT_NMAP | HK_FEAT_ANISO)) { sc.fn = surf.flatNormal; sc.tg = surf.tangent; sc.bn = surf.biTangent; }
  const BxDFResult ev = materialEval<F>(mat, sc, s);
  const float cos1 = fmaxf(+dot(shadowRayDir, surfNormal), 0.0f), cos2 = fmaxf(-dot(shadowRayDir, surfNormal), 0.0f);
  const f3 bxdfVal = (ev.brdf * cos1) + (ev.btdf * cos2);
  const float samPdf = fabsf(pdfSigned);
  float misWeight = misWeightHeuristic(samPdf * lightPickProb, ev.pdfFwd);
  if (pdfSigned < 0.0f) misWeight = 1.0f;
  const f3 lc = lightColor * (1.0f / fmaxf(samPdf, HK_DEPSILON2));
  return ((lc * (1.0f / lightPickProb)) * bxdfVal) * misWeight;
}

// S2 -- BSDF sampling of the next bounce (kernel_NextBounce) with the ten random numbers of RndMatAll handed in; `accum` is the radiance carried on
template <int F = HK_FEAT_ALL>
// returns true when the sampled leaf is a shadow catcher whose throughput awaits this bounce's shadow (hk_shading.h, HK_MATTE_PENDING)
HK_DEV bool next_bounce_with(const SceneDev& s, const float* mat, const SurfaceHit& surf, const f3 ray_dir, uint32_t flags, const float* rands,
                             const float4& thr4, const f3 accum, const float gidBits, float4& oPos, float4& oDir, float4& oThr, float4& oAcc) {
  MatSample ms;
  MaterialSampleAndEvalBxDF<F>(mat, rands, surf, ray_dir, flags, s, ms);
  const f3 bxdfVal = ms.color * (1.0f / fmaxf(ms.pdf, 1e-20f));
  const float cosTheta = fabsf(dot(ms.direction, surf.normal));
  const f3 newPos = OffsRayPos(surf.pos, surf.normal, ms.direction);
  const bool isSpec = ((ms.flags & HRE_S) != 0 || (ms.flags & HRE_T) != 0);
  flags = flagsNextBounceLite(flags, ms, s);
  const f3 thr = xyz(thr4) * (bxdfVal * cosTheta);
  oPos = mk4(newPos, gidBits);
  oDir = mk4(ms.direction, as_float(int(flags)));
  oThr = mk4(thr, ms.pdf);
  oAcc = mk4(accum, isSpec ? 1.0f : 0.0f);
  return (ms.flags & HK_MATTE_PENDING) != 0;
}
template <int F = HK_FEAT_ALL>
HK_DEV bool next_bounce_phase(const SceneDev& s, const float* mat, const SurfaceHit& surf, const f3 ray_dir, uint32_t flags, RandomGen& gen,
                              const float4& thr4, const f3 accum, const float gidBits, float4& oPos, float4& oDir, float4& oThr, float4& oAcc) {
  float rands[10];   // RndMatAll, crandom.h:478-494: 1 draw -> 3 floats, then 7 single draws
  {
    const float4 r4 = rndFloat4_Pseudo(gen);
    rands[0] = r4.x; rands[1] = r4.y; rands[2] = r4.z;
    for (int k = 0; k < 7; k++) rands[3 + k] = rndFloat1_Pseudo(gen);
  }
  return next_bounce_with<F>(s, mat, surf, ray_dir, flags, rands, thr4, accum, gidBits, oPos, oDir, oThr, oAcc);
}

// Split form, kernel 1 of 2: hit phase, survivors compacted into M
HK_DEV void k_hit_body(const SceneDev& s, const SegQ& q, uint32_t* __restrict__ nextCounts,
                       uint32_t* __restrict__ shadowCounts, int depth, int maxDepth, const PathState& S,
                       const HydraLiteHit* __restrict__ hits, const MidState& M,
                       float4* __restrict__ contrib, uint2* __restrict__ gens) {
  const SegIter it = segq_iter(q);
  const int count = it.count;
  uint32_t* nextCount = nextCounts + it.seg * HK_CSTRIDE;
  int shadowRaysOfWave = 0;   // statistic only: one atomic per wave at the end instead of one per iteration
  for (int idx0 = it.first - int(threadIdx.x); idx0 < count; idx0 += it.step) {   // idx0 is block-uniform: every wave runs the ballots
    const int idx = idx0 + int(threadIdx.x);
    const int i = it.base + idx;
    bool alive = false;
    float4 pos4 = make_float4(0, 0, 0, 0), dir4 = pos4, thr4 = pos4, acc4 = pos4;
    RandomGen gen; gen.x = gen.y = 0;
    SurfaceHit surf;
    LightPick lp;
    lp.shadowOrg = make_float4(0, 0, 0, -1.0f);
    if (idx < count) {
      pos4 = S.pos4[i]; dir4 = S.dir4[i];
      if (depth > 0) { thr4 = S.thr4[i]; acc4 = S.acc4[i]; }
      else { thr4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f); acc4 = make_float4(0.0f, 0.0f, 0.0f, 1.0f); }   // what every path starts with: not stored by k_raygen, see there
      const uint2 g2 = S.rng2[i];
      gen.x = g2.x; gen.y = g2.y;
      const float4 h4 = reinterpret_cast<const float4*>(hits)[i];
      HydraLiteHit hit; hit.t = h4.x; hit.primId = as_int(h4.y); hit.instId = as_int(h4.z); hit.geomId = as_int(h4.w);
      f3 finalColor;
      alive = surface_phase(s, depth, maxDepth, pos4, dir4, thr4, acc4, hit, surf, finalColor);
      if (!alive) {
        const int gid = as_int(pos4.w);
        contrib[gid] = mk4(finalColor, 0.0f);
        gens[gid] = make_uint2(gen.x, gen.y);
      } else
        light_phase(s, surf, gen, lp);
    }
    const int dst = it.base + wave_compact_index(alive, nextCount);
    shadowRaysOfWave += __popcll(__ballot(alive && lp.shadowOrg.w >= 0.0f));
    if (alive) {
      M.dir4[dst] = dir4; M.thr4[dst] = thr4; M.acc4[dst] = acc4; M.rng2[dst] = make_uint2(gen.x, gen.y);
      M.surfA[dst] = mk4(surf.pos, as_float(surf.matId));
      M.surfB[dst] = mk4(surf.normal, surf.texCoord.x);
      M.recC[dst] = mk4(lp.shadowRayDir, surf.texCoord.y);
      M.recD[dst] = mk4(lp.color, lp.pdfSigned);
      M.recE[dst] = make_float4(lp.pickProb, as_float(lp.lightOffset), pos4.w, surf.hfi ? 1.0f : 0.0f);   // hit-from-inside: the glass BxDF needs it
      M.shadowOrg[dst] = lp.shadowOrg;
    }
  }
  if (shadowRaysOfWave > 0 && __lane_id() == 0) atomicAdd(shadowCounts + it.seg * HK_CSTRIDE, uint32_t(shadowRaysOfWave));
}

// W = minimum waves per SIMD the register allocator must leave room for (256-thread blocks): 3 => 152 VGPRs, no spills;
// 4 => 128 VGPRs with a dozen spilled dwords but a third more waves to hide the dependent gathers (measured in DESIGN.md 6)
template <int W>
__global__ void __launch_bounds__(256, W) k_hit(SceneDev s, SegQ q, uint32_t* __restrict__ nextCount,
                                                 uint32_t* __restrict__ shadowCount, int depth, int maxDepth, PathState S,
                                                 const HydraLiteHit* __restrict__ hits, MidState M,
                                                 float4* __restrict__ contrib, uint2* __restrict__ gens) {
  s.ptlSlot = -1;
  k_hit_body(s, q, nextCount, shadowCount, depth, maxDepth, S, hits, M, contrib, gens);
}

// Split form, kernel 2 of 2 (after the shadow rays): next-event shading and BSDF sampling of the next bounce
HK_DEV void k_shade_body(const SceneDev& s, const SegQ& q, const MidState& M, const PathState& S) {
  const SegIter it = segq_iter(q);
  for (int idx = it.first; idx < it.count; idx += it.step) {
    const int i = it.base + idx;
    const float4 dir4 = M.dir4[i], thr4 = M.thr4[i], acc4 = M.acc4[i];
    const float4 sa = M.surfA[i], sb = M.surfB[i], rc = M.recC[i], rd = M.recD[i], re = M.recE[i];
    const uint2 g2 = M.rng2[i];
    RandomGen gen; gen.x = g2.x; gen.y = g2.y;
    const f3 ray_dir = xyz(dir4);
    SurfaceHit surf;
    surf.pos = xyz(sa); surf.matId = as_int(sa.w); surf.normal = xyz(sb); surf.texCoord = mk2(sb.w, rc.w);
    surf.hfi = (re.w != 0.0f);
    surf.flatNormal = surf.normal; surf.tangent = mk3(0, 0, 0); surf.biTangent = mk3(0, 0, 0); surf.t = 0.0f; surf.sRayOff = 0.0f;   // not read after the hit phase
    const float* mat = materialAt(s, surf.matId);
    f3 explicitColor = mk3(0, 0, 0);
    if (as_int(re.y) >= 0) explicitColor = direct_light_unoccluded<HK_FEAT_CLASSIC>(s, mat, surf, ray_dir, xyz(rc), xyz(rd), rd.w, re.x) * M.vis[i];   // the split form carries no tangent frame: scenes with normal maps use the fused kernel
    const f3 accum = xyz(acc4) + (xyz(thr4) * explicitColor);
    float4 oPos, oDir, oThr, oAcc;
    next_bounce_phase<HK_FEAT_CLASSIC>(s, mat, surf, ray_dir, uint32_t(as_int(dir4.w)), gen, thr4, accum, re.z, oPos, oDir, oThr, oAcc);
    S.pos4[i] = oPos; S.dir4[i] = oDir; S.thr4[i] = oThr; S.acc4[i] = oAcc;
    S.rng2[i] = make_uint2(gen.x, gen.y);
  }
}

template <int W>
__global__ void __launch_bounds__(256, W) k_shade(SceneDev s, SegQ q, MidState M, PathState S) {
  s.ptlSlot = -1;
  k_shade_body(s, q, M, S);
}

// Fused form: one kernel per bounce does hit + light sample + unoccluded next-event estimate + BSDF sampling and writes
// the next path state straight into the other S buffer.  The light term waits as `pend` = throughput * estimate until
// the shadow kernel has produced vis; the NEXT k_bounce (which every survivor passes through) adds pend * vis first.
// vis is 0 or 1, so acc + (thr * X) * vis has the bits of the split form's acc + thr * (X * vis).  Per path-bounce
// this moves 228 B through HBM instead of 504 B and drops one launch (no M record).
struct ShadowQ { float4* org4; float4* dir4; float* vis; };

// LDS staging of the scene's small hot tables (SceneDev::matBase ... texTable): sizes in 16-byte units, all zero = leave them in
// global memory (tables too large for HK_SCENE_LDS_MAX_BYTES per block, or option "scene_tables_in_lds" 0)
struct SceneStage { int matF4, matTabF4, lightsF4, texTabF4, hdrF4, lselF4, triBaseF4, instLightF4, instMatF4; const float4* img; };
#define HK_SORT_BINS 16
#define HK_SCENE_LDS_MAX_BYTES (44 * 1024)   // + 5.3 KB of sort arrays, x 3 resident 256-thread blocks per CU = 148 of the CU's 160 KB
// STG (compile time): bit 0 = the material group is staged (material arena, material-id table, lights, texture-id table, the scalar
// header and the light-selection table), bit 1 = the path group (triBase, per-instance light ids and matrices).  The pointers of a
// staged group are re-pointed UNCONDITIONALLY, so that the compiler sees every access through them start at the LDS array and
// emits ds_read instead of flat_load: a flat load is routed through the CU's vector-memory address path, the very unit this
// kernel saturates, and waits on vmcnt and lgkmcnt together, i.e. for every global load in flight as well.
template <int STG>
HK_DEV void stage_scene_tables(SceneDev& s, const SceneStage st, float4* lds) {
  if (STG == 0) return;
  const int n0 = st.matF4, n1 = n0 + st.matTabF4, n2 = n1 + st.lightsF4, n3 = n2 + st.texTabF4, n3a = n3 + st.hdrF4, n3b = n3a + st.lselF4;
  const int n4 = n3b + st.triBaseF4, n5 = n4 + st.instLightF4, n6 = n5 + st.instMatF4;
  const float4* __restrict__ img = st.img;
  const int B = int(blockDim.x);
  int i = int(threadIdx.x);
  for (; i + 3 * B < n6; i += 4 * B) {
    const float4 v0 = img[i], v1 = img[i + B], v2 = img[i + 2 * B], v3 = img[i + 3 * B];
    lds[i] = v0; lds[i + B] = v1; lds[i + 2 * B] = v2; lds[i + 3 * B] = v3;
  }
  for (; i < n6; i += B) lds[i] = img[i];
  __syncthreads();
  if (STG & 1) {
    s.matBase = reinterpret_cast<const float*>(lds);
    s.matTable = reinterpret_cast<const int*>(lds + n0);
    s.lightsBase = reinterpret_cast<const float*>(lds + n1);   // lightsF4 == 0: never dereferenced (no light ids exist)
    s.texTable = reinterpret_cast<const int*>(lds + n2);
    s.hdr = reinterpret_cast<const int*>(lds + n3);
    s.lselRev = reinterpret_cast<const float*>(lds + n3a);
  }
  if (STG & 2) {
    s.triBase = reinterpret_cast<const int*>(lds + n3b);
    s.instLightInstId = reinterpret_cast<const int*>(lds + n4);
    s.instMatrices = lds + n5;
  }
}
#ifdef HK_EXP_BOUNCE_STAMPS   /* timing experiment (tools/bounce_stamps.py): s_memtime at the phase boundaries of k_bounce, summed per wave */
// one array per translation unit that instantiates k_bounce; hydra_hip_debug_bounce_stamps (hydra_hip.hip) adds them up
static __device__ unsigned long long hk_bounce_stamps[16];
static inline int hk_bounce_stamps_read(unsigned long long* acc16, int reset) {
  unsigned long long v[16];
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(hk_bounce_stamps), sizeof(v)) != hipSuccess) return -1;
  for (int k = 0; k < 16; k++) acc16[k] += v[k];
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(hk_bounce_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#define HK_STAMP(k) { const unsigned long long now_ = __builtin_readcyclecounter(); stampAcc[k] += now_ - stampT; stampT = now_; }
#else
#define HK_STAMP(k)
#endif
#ifndef HK_BOUNCE_BLOCK
#define HK_BOUNCE_BLOCK 256   // threads per block of the fused bounce kernel (only wave-level cooperation inside: any multiple of 64 works)
#endif
template <int W, int F = HK_FEAT_ALL, int STG = 0>   // F: the shading features this instantiation contains (hk_shading.h, HK_FEAT_*); STG: see stage_scene_tables
__global__ void __launch_bounds__(HK_BOUNCE_BLOCK, W) k_bounce(SceneDev sArg, SceneStage stage, SegQ q, uint32_t* __restrict__ nextCounts, uint32_t* __restrict__ shadowCounts,
                                                    int depth, int maxDepth, PathState Sin, PathState Sout, const HydraLiteHit* __restrict__ hits,
                                                    ShadowQ sh, float4* __restrict__ contrib, uint2* __restrict__ gens, int sortPaths, ScreenOfPath screen) {
  extern __shared__ float4 hk_scene_lds[];
  // workgroup-local grouping of the paths by shading class (sortPaths): counts per wave and class, slot offsets, permutation
  constexpr int NW = HK_BOUNCE_BLOCK / 64;
  static_assert(NW * HK_SORT_BINS <= 64, "the offset scan of the path grouping runs in one wave");
  __shared__ int sCnt[NW][HK_SORT_BINS];
  __shared__ int sOff[HK_SORT_BINS][NW];
  __shared__ unsigned short sPerm[HK_BOUNCE_BLOCK];
  __shared__ float4 sHit[HK_BOUNCE_BLOCK];   // the hit records travel with the permutation: no second fetch, and the triangle fetch can leave with the state loads
  SceneDev s = sArg;
  if constexpr ((F & HK_FEAT_PROCTEX) == 0) s.ptlSlot = -1;   // a constant: the procedural-texture look-up of sample2DExt folds away
#ifdef HK_EXP_BOUNCE_STAMPS
  unsigned long long stampAcc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stampT = __builtin_readcyclecounter();
#endif
  const SegIter it = segq_iter(q);
  const int count = it.count;
  if (it.first - int(threadIdx.x) >= count) return;   // block-uniform: nothing in this block's stride (late bounces leave most of the grid idle) -- before the tables are staged
  stage_scene_tables<STG>(s, stage, hk_scene_lds);
  uint32_t* nextCount = nextCounts + it.seg * HK_CSTRIDE;
  int shadowRaysOfWave = 0;
  HK_STAMP(0)
  for (int idx0 = it.first - int(threadIdx.x); idx0 < count; idx0 += it.step) {
    // Which path of this 256-path chunk the thread shades.  Closest hits arrive labelled with the shading class of their
    // material (hk_trace.h, HK_CLASS_SHIFT): a counting sort of the chunk by class through LDS hands every wave paths that
    // run the same code (after the first bounce a wave otherwise holds every material of the scene and pays for each of them
    // in turn).  State is read and survivors are written by path index as before, so nothing but the order inside a chunk changes.
    int src = int(threadIdx.x);
    float4 h4 = make_float4(0.0f, as_float(-1), as_float(-1), 0.0f);
    if (idx0 + src < count) h4 = reinterpret_cast<const float4*>(hits)[it.base + idx0 + src];
    if (sortPaths) {
      const int wave = int(threadIdx.x) >> 6, lane = int(threadIdx.x) & 63;
      int key = HK_SORT_BINS - 1;                             // past the end of the queue
      if (idx0 + src < count) {
        const int cls = HK_GEOM_CLASS(as_int(h4.w));
        key = (as_int(h4.y) == -1) ? 0 : (cls != 0 ? (cls < HK_SORT_BINS - 2 ? cls : HK_SORT_BINS - 2) : HK_SORT_BINS - 2);
      }
      int rank = 0;
      for (int b = 0; b < HK_SORT_BINS; b++) {
        const unsigned long long m = __ballot(key == b);
        if (key == b) rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) sCnt[wave][b] = __popcll(m);
      }
      __syncthreads();
      if (wave == 0) {                                         // exclusive scan in class-major, wave-minor order
        const int b = lane / NW, w = lane % NW;
        const int v = (lane < NW * HK_SORT_BINS) ? sCnt[w][b] : 0;
        int incl = v;
        for (int d = 1; d < 64; d <<= 1) { const int n = __shfl_up(incl, d); if (lane >= d) incl += n; }
        if (lane < NW * HK_SORT_BINS) sOff[b][w] = incl - v;
      }
      __syncthreads();
      const int slot = sOff[key][wave] + rank;
      sPerm[slot] = (unsigned short)threadIdx.x;
      sHit[slot] = h4;
      __syncthreads();
      src = int(sPerm[threadIdx.x]);
      h4 = sHit[threadIdx.x];
    }
    HK_STAMP(1)
    const int idx = idx0 + src;
    const int i = it.base + idx;
    bool alive = false;
    float4 oPos = make_float4(0, 0, 0, 0), oDir = oPos, oThr = oPos, oAcc = oPos, oPend = oPos, oShDir = oPos;
    RandomGen gen; gen.x = gen.y = 0;
    LightPick lp;
    lp.shadowOrg = make_float4(0, 0, 0, -1.0f);
    float4 pos4 = make_float4(0, 0, 0, 0), dir4 = pos4, thr4 = pos4, acc4 = pos4;
    SurfaceHit surf;
    f3 finalColor = mk3(0, 0, 0);
    if (idx < count) {
      pos4 = Sin.pos4[i]; dir4 = Sin.dir4[i];
      // a path starts with throughput 1, MIS pdf 1, no radiance, "previous bounce specular" (kernel_InitAccumData + makeInitialMisData): constants, so k_raygen
      // does not write them and the first bounce does not read them (64 B less per path and sample through HBM)
      thr4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f); acc4 = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
      if (depth > 0) {   // settle the previous bounce's next-event estimate
        thr4 = Sin.thr4[i]; acc4 = Sin.acc4[i];
        const float4 pend = Sin.pend4[i];
        const float vis = sh.vis[i];
        acc4.x = acc4.x + pend.x * vis; acc4.y = acc4.y + pend.y * vis; acc4.z = acc4.z + pend.z * vis;
        if ((F & HK_FEAT_RARE_LIGHTS) && pend.w != 0.0f) { thr4.x *= vis; thr4.y *= vis; thr4.z *= vis; }   // the previous bounce went through a shadow catcher (hk_shading.h, HK_MATTE_PENDING)
      }
      const uint2 g2 = Sin.rng2[i];
      gen.x = g2.x; gen.y = g2.y;
      HydraLiteHit hit; hit.t = h4.x; hit.primId = as_int(h4.y); hit.instId = as_int(h4.z); hit.geomId = as_int(h4.w);
      if constexpr ((F & HK_FEAT_PROCTEX) != 0) s.ptlSlot = (uint32_t(s.ptlIds[i]) != HYDRA_INVALID_TEXTURE) ? i : -1;   // what k_proctex left for this path (hk_proctex_rt.h)
      // the triangle record and the instance matrix are requested here, behind the state loads and before anything waits for those:
      // one round trip to memory for both instead of two in a row
#ifdef HK_EXP_BOUNCE_PRELOAD
      TriData td;
      m44 instInv;
      if (HitSome(hit)) { instInv = load_m44(s.instMatrices + size_t(hit.instId) * 4); td = fetchTri(s, hit); }
#endif
#if defined(HK_EXP_BOUNCE_SKIP) && (HK_EXP_BOUNCE_SKIP & 4)   /* timing experiment (profiles/r01/pass_bounce_phase_cost.log): phases left out, results invalid */
      alive = true; surf.pos = xyz(pos4); surf.normal = mk3(0, 1, 0); surf.flatNormal = surf.normal; surf.tangent = mk3(1, 0, 0); surf.biTangent = mk3(0, 0, 1);
      surf.texCoord = mk2(0, 0); surf.matId = 0; surf.t = hit.t; surf.sRayOff = 0.0f; surf.hfi = false;
#else
#ifdef HK_EXP_BOUNCE_PRELOAD
      alive = surface_phase_with<F>(s, depth, maxDepth, pos4, dir4, thr4, acc4, hit, td, instInv, surf, finalColor, screen);
#else
      alive = surface_phase<F>(s, depth, maxDepth, pos4, dir4, thr4, acc4, hit, surf, finalColor, screen);
#endif
#endif
    }
    // the slot of the survivor is reserved as soon as survival is known: the returning atomic then overlaps the light and
    // material fetches below instead of standing alone at the end of the iteration
    HK_STAMP(2)
    // issued here, read just before the stores: the atomic's round trip runs under the light and material work
    const unsigned long long aliveMask = __ballot(alive);
    const int aliveLeader = aliveMask != 0ull ? __ffsll((long long)aliveMask) - 1 : 0;
    int slotBase = 0;
    if (aliveMask != 0ull && int(__lane_id()) == aliveLeader) slotBase = int(atomicAdd(nextCount, uint32_t(__popcll(aliveMask))));
    HK_STAMP(3)
    if (idx < count) {
      if (!alive) {
        const int gid = as_int(pos4.w);
        contrib[gid] = mk4(finalColor, 0.0f);
        gens[gid] = make_uint2(gen.x, gen.y);
      } else {
#if defined(HK_EXP_BOUNCE_SKIP) && (HK_EXP_BOUNCE_SKIP & 1)
        lp.lightOffset = -1; lp.shadowRayDir = mk3(0, 1, 0); lp.color = mk3(0, 0, 0); lp.pdfSigned = 1.0f; lp.pickProb = 1.0f;
#else
        light_phase<F>(s, surf, gen, lp);
#endif
        HK_STAMP(4)
        const float* mat = materialAt(s, surf.matId);
        const f3 ray_dir = xyz(dir4);
        f3 pend = mk3(0, 0, 0);
        if (lp.lightOffset >= 0)
          pend = xyz(thr4) * direct_light_unoccluded<F>(s, mat, surf, ray_dir, lp.shadowRayDir, lp.color, lp.pdfSigned, lp.pickProb);
        oPend = mk4(pend, 0.0f);
        oShDir = mk4(lp.shadowRayDir, 0.0f);
        HK_STAMP(5)
#if defined(HK_EXP_BOUNCE_SKIP) && (HK_EXP_BOUNCE_SKIP & 2)
        oPos = mk4(surf.pos, pos4.w); oDir = dir4; oThr = thr4; oAcc = acc4;
#else
        if (next_bounce_phase<F>(s, mat, surf, ray_dir, uint32_t(as_int(dir4.w)), gen, thr4, xyz(acc4), pos4.w, oPos, oDir, oThr, oAcc)) oPend.w = 1.0f;
#endif
      }
    }
#ifdef HK_EXP_BOUNCE_EXTRA_VALU   /* timing experiment: N dependent multiply-adds per path, result parked in the unused pend4.w */
    {
      float x = oThr.x;
      for (int rep = 0; rep < HK_EXP_BOUNCE_EXTRA_VALU; rep++) { asm volatile("" : "+v"(x)); x = x * 1.0001f + 0.5f; }
      oPend.w = x;
    }
#endif
    HK_STAMP(6)
    shadowRaysOfWave += __popcll(__ballot(alive && lp.shadowOrg.w >= 0.0f));
    const int dst = it.base + __builtin_amdgcn_readlane(slotBase, aliveLeader) + __popcll(aliveMask & ((1ull << __lane_id()) - 1ull));
    if (alive) {
      Sout.pos4[dst] = oPos; Sout.dir4[dst] = oDir; Sout.thr4[dst] = oThr; Sout.acc4[dst] = oAcc;
      Sout.rng2[dst] = make_uint2(gen.x, gen.y);
      Sout.pend4[dst] = oPend;
      sh.org4[dst] = lp.shadowOrg; sh.dir4[dst] = oShDir;
    }
    HK_STAMP(7)
  }
  if (shadowRaysOfWave > 0 && __lane_id() == 0) atomicAdd(shadowCounts + it.seg * HK_CSTRIDE, uint32_t(shadowRaysOfWave));
#ifdef HK_EXP_BOUNCE_STAMPS
  if (__lane_id() == 0) { for (int k = 0; k < 8; k++) atomicAdd(&hk_bounce_stamps[k], stampAcc[k]); atomicAdd(&hk_bounce_stamps[8], 1ull); }
#endif
}

// ---- IntegratorMMLT::F in wavefront form (hk_bidir.h): one thread per chain between the traversal launches
// The sub-path rays of a level live in segmented, compacted queues like the path tracer's (SegQ): a ray slot holds origin, direction and its
// owner (chain * 2 + side, side 1 = light sub-path); a finished sub-path simply appends nothing, so a level traces only the live rays.
struct MmltRays { float4* pos; float4* dir; int* owner; };
#ifndef HK_MMLT_STEP_W
#define HK_MMLT_STEP_W 3   /* register budget of the MMLT stage kernels in waves per SIMD */
#endif
#ifndef HK_MMLT_CONN_W
#define HK_MMLT_CONN_W 2   /* k_mmlt_connect_end: 256 registers and no spills beat 168 with 69 spilled dwords (533-552 -> 608 M mutations/s, profiles/r02/mmlt_register_budget.log) */
#endif
template <int F>
__global__ void __launch_bounds__(256, HK_MMLT_STEP_W) k_mmlt_step(SceneDev s, MmltView v, int currDepth, SegQ q, MmltRays in, const HydraLiteHit* __restrict__ hits, MmltRays out, uint32_t* __restrict__ outCount) {
  s.ptlSlot = -1;   // MMLT does not run procedural textures (hydra_hip_mmlt_begin refuses such a scene)
  const SegIter it = segq_iter(q);
  uint32_t* counter = outCount + it.seg * HK_CSTRIDE;
  for (int idx = it.first; idx - int(__lane_id()) < it.count; idx += it.step) {   // whole waves iterate together: the compaction is a wave ballot
    bool alive = false;
    float4 npos, ndir;
    int owner = 0;
    if (idx < it.count) {
      const int j = it.base + idx;
      owner = in.owner[j];
      const int chain = owner >> 1;
      alive = (owner & 1) ? mmltLightStep<F>(s, v, chain, currDepth, in.pos[j], in.dir[j], hits[j], npos, ndir)
                          : mmltCameraStep<F>(s, v, chain, currDepth, in.pos[j], in.dir[j], hits[j], npos, ndir);
    }
    const int dst = it.base + wave_compact_index(alive, counter);
    if (alive) { out.pos[dst] = npos; out.dir[dst] = ndir; out.owner[dst] = owner; }
  }
}
template <int F>
__global__ void k_mmlt_connect_begin(SceneDev s, MmltView v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  s.ptlSlot = -1;
  if (i < v.n) mmltConnectBegin<F>(s, v, i);
}
template <int F>
__global__ void __launch_bounds__(256, HK_MMLT_CONN_W) k_mmlt_connect_end(SceneDev s, MmltView v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  s.ptlSlot = -1;
  if (i < v.n) mmltConnectEnd<F>(s, v, i);
}

// ================================================================================================ launchers (defined in hk_inst_*.hip)
struct TraceLaunch {          // one traversal launch: rays a4 (origin | t_far for shadow rays) / b4 (direction), results hits or vis
  int grid; hipStream_t stream; SceneDev s; SegQ q;
  const float4* a4; const float4* b4; HydraLiteHit* hits; float* vis;
  uint32_t* perRay3; unsigned long long* totals5; uint32_t* fetchCounters;
  int carry, minActive, raysPerLane;
  int vote, wq, wt, wi;       // persistent kernels: schedule by wave vote (trav_run_vote) with these weights, or 0 = the reference's loop nest
  int unordered = 0;          // shadow rays under the vote: children of a quad in stored order (k_trace_dyn<..., UNORD>)
};
void hk_launch_trace_static(bool count, bool alpha, const TraceLaunch& a);                          // k_trace<COUNT, ALPHA>
void hk_launch_shadow_static(bool count, const TraceLaunch& a);                                     // k_shadow<COUNT>
void hk_launch_trace_dyn(bool anyhit, bool count, bool toptris, bool alpha, const TraceLaunch& a);  // k_trace_dyn<ANYHIT, COUNT, TOPTRIS, ALPHA>

struct BounceLaunch {         // one launch of the fused bounce kernel
  int grid; size_t ldsBytes; hipStream_t stream; SceneDev s; SceneStage stage; SegQ qIn;
  uint32_t* nextCnt; uint32_t* shCnt; int depth, maxDepth; PathState A, B; const HydraLiteHit* hits; ShadowQ sh;
  float4* contrib; uint2* gens; int sortPaths;
  ScreenOfPath screen;   // back-plate scenes: which pixel a path belongs to
};
// each returns false when (W, F, STG) is not one of its instantiations; hk_launch_bounce tries them in turn
bool hk_launch_bounce_lean(int W, int F, int STG, const BounceLaunch& a);      // W 3; F = 0, SKY, SKY | DELTA_LIGHTS | OREN_NAYAR
bool hk_launch_bounce_classic(int W, int F, int STG, const BounceLaunch& a);   // W 3; F = CLASSIC & ~GLASS, CLASSIC & ~GGX, CLASSIC
bool hk_launch_bounce_nmap(int W, int F, int STG, const BounceLaunch& a);      // W 3; F = CLASSIC | NMAP
bool hk_launch_bounce_all(int W, int F, int STG, const BounceLaunch& a);       // W 3; F = ALL
bool hk_launch_bounce_all45(int W, int F, int STG, const BounceLaunch& a);     // W 4, 5 (experiment switches); F = ALL, STG 0 or 3
#ifdef HK_EXP_BOUNCE_STAMPS
int hk_bounce_stamps_read_lean(unsigned long long*, int); int hk_bounce_stamps_read_classic(unsigned long long*, int); int hk_bounce_stamps_read_nmap(unsigned long long*, int);
int hk_bounce_stamps_read_all(unsigned long long*, int); int hk_bounce_stamps_read_all45(unsigned long long*, int);
#endif

bool hk_launch_bounce_proctex(int W, int F, int STG, const BounceLaunch& a);   // W 3; F = ALL | PROCTEX (scenes whose materials bind procedural textures)

// procedural textures (hydra_proctex.hip): the scene's own texture functions, compiled at load time into k_proctex (hk_proctex_rt.h)
struct HkProcTexProgram;
bool hk_proctex_compile(const char* source, size_t len, std::vector<char>& code, std::string& log, std::string& err);   // text -> gfx950 code object; needs no device
HkProcTexProgram* hk_proctex_build(const char* source, size_t len, std::string& err);   // nullptr + err (with the compiler's log) on failure
void hk_proctex_free(HkProcTexProgram* p);
hipError_t hk_proctex_launch(HkProcTexProgram* p, int grid, hipStream_t stream, const SceneDev& s, const SegQ& q, const float4* pos4, const float4* dir4, const HydraLiteHit* hits,
                             int* ids, uint2* vals, int stride, int maxNum);
hipError_t hk_proctex_launch_points(HkProcTexProgram* p, hipStream_t stream, const SceneDev& s, int n, const float4* pos4, const float4* dir4, const HydraLiteHit* hits,
                                    int* ids, uint2* vals, int stride, int maxNum);

struct SplitLaunch {          // the split form: k_hit<W> / k_shade<W>
  int grid; hipStream_t stream; SceneDev s; SegQ q; uint32_t* nextCnt; uint32_t* shCnt; int depth, maxDepth; PathState S;
  const HydraLiteHit* hits; MidState M; float4* contrib; uint2* gens;
};
void hk_launch_hit(int W, const SplitLaunch& a);
void hk_launch_shade(int W, const SplitLaunch& a);

struct MmltLaunch {           // the stage kernels of IntegratorMMLT::F; F = SKY | DELTA_LIGHTS | OREN_NAYAR, CLASSIC or ALL
  int grid; hipStream_t stream; SceneDev s; MmltView v;
  int currDepth; SegQ q; MmltRays in; const HydraLiteHit* hits; MmltRays out; uint32_t* outCount;   // k_mmlt_step only
};
bool hk_launch_mmlt_lean(int kernel, int F, const MmltLaunch& a);      // kernel: 0 = step, 1 = connect_begin, 2 = connect_end; F = the lean set or CLASSIC
bool hk_launch_mmlt_all(int kernel, int F, const MmltLaunch& a);       // F = ALL
