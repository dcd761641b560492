// hydra_hip.hip -- libhydra_hip.so: wavefront path-tracing core for MI355X (gfx950) behind the C-ABI of
// include/hydra_hip.h.  Stage split (BASELINE.json north_star): ray generation, BVH4 traversal + Moeller-Trumbore,
// hit/emission/light-sample with wave-ballot compaction, any-hit shadow traversal, BSDF shade + next bounce,
// framebuffer accumulate.  Reference behaviour: IntegratorMISPTLoop2 (hydra_drv/CPUExp_Integrators_PT_Loop.cpp:264-321),
// pass driver IntegratorCommon::DoPass (CPUExp_Integrators_Common.cpp:278-316).
//
// HBM layout (SoA, one entry per live path, all float4 so every lane moves 16 B per access):
//   S.pos4  = ray origin xyz | gid (see below)        S.dir4 = ray direction xyz | ray flags
//   S.thr4  = path throughput xyz | prev BSDF pdf  S.acc4 = accumulated radiance xyz | prev-bounce-was-specular
//   S.rng2  = RandomGen state
// K_hit writes survivors densely into the M (mid) arrays via a wave-aggregated atomic, so the shadow and shade
// kernels and the next bounce always read contiguous, fully coalesced arrays; live counts stay on the device
// (live[bounce]) and every kernel is a grid-stride loop over *count, so a whole pass is enqueued without host syncs.
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <queue>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <thread>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library itself is dlopen()ed by hydra_hip_comm_init
#include "../../include/hydra_hip.h"
#include "hk_kernels.h"   // the template kernels and their launchers (other translation units hold the instantiations)
#include "hk_gbuffer.h"

// ================================================================================================ kernels
// counts of a queue of n items cut into nseg segments of `cap` slots, laid out back to back (item r = slot r % cap of segment r / cap)
__global__ void k_fill_seg_counts(int n, int cap, int nseg, uint32_t* __restrict__ counts) {
  const int sg = int(threadIdx.x);
  if (sg >= nseg) return;
  const long long left = (long long)n - (long long)sg * cap;
  counts[sg * HK_CSTRIDE] = uint32_t(left <= 0 ? 0 : (left < cap ? left : cap));
}
// P1 -- ray generation: IntegratorCommon::makeEyeRay (Common.cpp:347-359) for every owned pixel
// Path p of a sub-pass that traces `ns` samples of each of the N owned pixels: stream-major (p = stream * N + pixelIndex, the
// default) or pixel-major (p = pixelIndex * ns + stream: the 64 lanes of a wave are 64 / ns neighbouring pixels x ns samples;
// measured within noise of the default, profiles/r01/pass_path_order.log).  Paths are dealt to the queue segments in
// chunks of 256: chunk c -> segment c % nseg, so slot idx of segment seg holds path ((idx / 256) * nseg + seg) * 256 + idx % 256.
// gid = stream * nOwned + pixelIndex is the index of the path's RandomGen state and of its contribution record: both arrays
// hold only what this rank owns (1/world of the frame), while the generator itself is seeded from the GLOBAL slot
// stream * (w*h) + pixel (k_init_gens), so the image does not depend on the partition.
__global__ void k_raygen(SceneDev s, SegQ q, const int* __restrict__ ownedPixels, int nOwned, int ns, int streamMajor,
                         const uint2* __restrict__ gens, int w, int h, PathState S) {
  const SegIter it = segq_iter(q);
  for (int idx = it.first; idx < it.count; idx += it.step) {
    const int i = it.base + idx;
    const long long p = ((long long)(idx >> 8) * q.nseg + it.seg) * 256 + (idx & 255);
    int pixIdx, stream;
    if (streamMajor) { stream = int(p / nOwned); pixIdx = int(p - (long long)stream * nOwned); }
    else { pixIdx = int(p / ns); stream = int(p - (long long)pixIdx * ns); }
    const int pixel = ownedPixels[pixIdx];
    const int gid = stream * nOwned + pixIdx;   // < 2^31: alloc_render_state checks nOwned * K
    const uint2 g2 = gens[gid];
    RandomGen gen; gen.x = g2.x; gen.y = g2.y;
    const float4 r = rndFloat4_Pseudo(gen);   // rndUniform(gen, -1, 1), crandom.h:617-620
    const float4 offs = make_float4(-1.0f + 2.0f * r.x, -1.0f + 2.0f * r.y, -1.0f + 2.0f * r.z, -1.0f + 2.0f * r.w);
    f3 pos, dir;
    MakeRandEyeRay(pixel % w, pixel / w, w, h, offs, s, pos, dir);
    S.pos4[i] = mk4(pos, as_float(gid));
    S.dir4[i] = mk4(dir, as_float(0));
    // thr4 = (1, 1, 1 | MIS pdf 1) and acc4 = (0, 0, 0 | isSpecular 1) (kernel_InitAccumData + makeInitialMisData) are the same for every path: the bounce
    // kernels put them in at depth 0 instead of reading them, so they are not written here
    S.rng2[i] = make_uint2(gen.x, gen.y);
  }
}

// The staged tables gathered into one array in the order of the LDS copy (k_build_stage_image, once per pass): a block then fills
// its LDS from ONE contiguous range with four loads in flight per thread, instead of choosing among nine sources per float4.
__global__ void k_build_stage_image(SceneDev s, SceneStage st, float4* __restrict__ img) {
  const int n0 = st.matF4, n1 = n0 + st.matTabF4, n2 = n1 + st.lightsF4, n3 = n2 + st.texTabF4, n3a = n3 + st.hdrF4, n3b = n3a + st.lselF4;
  const int n4 = n3b + st.triBaseF4, n5 = n4 + st.instLightF4, n6 = n5 + st.instMatF4;
  const float4* a = reinterpret_cast<const float4*>(s.matBase), *b = reinterpret_cast<const float4*>(s.matTable);
  const float4* c = reinterpret_cast<const float4*>(s.lightsBase), *d = reinterpret_cast<const float4*>(s.texTable);
  const float4* d1 = reinterpret_cast<const float4*>(s.hdr), *d2 = reinterpret_cast<const float4*>(s.lselRev);
  const float4* e = reinterpret_cast<const float4*>(s.triBase), *f = reinterpret_cast<const float4*>(s.instLightInstId), *g = s.instMatrices;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n6; i += gridDim.x * blockDim.x) {
    float4 v;
    if (i < n0) v = a[i]; else if (i < n1) v = b[i - n0]; else if (i < n2) v = c[i - n1]; else if (i < n3) v = d[i - n2];
    else if (i < n3a) v = d1[i - n3]; else if (i < n3b) v = d2[i - n3a];
    else if (i < n4) v = e[i - n3b]; else if (i < n5) v = f[i - n4]; else v = g[i - n5];
    img[i] = v;
  }
}
// F1 -- framebuffer accumulate: sums, mean on readout (SURVEY.md row a/F1; CPU reference keeps a running mean, Common.cpp:283,303)
// one thread per owned pixel adds the `streams` samples of this sub-pass in stream order, so the sum does not depend on
// where the paths lived in the queues
__global__ void k_accumulate(int n, const int* __restrict__ ownedPixels, const float4* __restrict__ contrib, float4* __restrict__ accum, int streams) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int pixel = ownedPixels[i];
    float4 a = accum[pixel];
    for (int k = 0; k < streams; k++) {
      const float4 c = contrib[size_t(k) * n + i];
      a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
    }
    accum[pixel] = a;
  }
}
// P0 readout -- IntegratorCommon::GetImageToLDR (Common.cpp:319-333): mean = sums / spp, ToneMapping4 (clamp to 1, cglobals.h:697-724),
// linearToSRGB (cglobals.h:3032-3038: the pow in float, the scale and offset in double), RealColorToUint32 (truncation)
__global__ void k_ldr(int n, const float4* __restrict__ accum, float invSpp, uint32_t* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 a = accum[i];
    float ch[4] = {fminf(a.x * invSpp, 1.0f), fminf(a.y * invSpp, 1.0f), fminf(a.z * invSpp, 1.0f), fminf(a.w * invSpp, 1.0f)};
    for (int k = 0; k < 3; k++) ch[k] = (ch[k] <= 0.00313066844250063f) ? ch[k] * 12.92f : float(1.055 * double(powf(ch[k], 1.0f / 2.4f)) - 0.055);
    const unsigned char r = (unsigned char)(ch[0] * 255.0f), g = (unsigned char)(ch[1] * 255.0f), b = (unsigned char)(ch[2] * 255.0f), al = (unsigned char)(ch[3] * 255.0f);
    out[i] = uint32_t(r) | (uint32_t(g) << 8) | (uint32_t(b) << 16) | (uint32_t(al) << 24);
  }
}
// multi-GPU exchange (hydra_hip_comm_gather_frame): a rank packs the accumulator values of its own pixels, slot order, for the root ...
__global__ void k_pack_owned(int n, const int* __restrict__ ownedPixels, const float4* __restrict__ accum, float4* __restrict__ packed) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) packed[i] = accum[ownedPixels[i]];
}
// ... and the root drops what it received into its frame (supports are disjoint: plain stores, nothing is added)
__global__ void k_unpack_owned(int n, const int* __restrict__ pixels, const float4* __restrict__ packed, float4* __restrict__ accum) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) accum[pixels[i]] = packed[i];
}
// counters are laid out [bounce][segment] with HK_CSTRIDE words between neighbours; one wave, lane = segment
__global__ void k_tally(const uint32_t* __restrict__ live, const uint32_t* __restrict__ shadowCnt, int maxDepth, int nseg, unsigned long long* totals) {
  const int sg = int(threadIdx.x);
  if (blockIdx.x != 0 || sg >= nseg) return;
  unsigned long long e = 0, sh = 0;
  for (int b = 0; b < maxDepth; b++) {
    e += live[size_t(b) * HK_CROW + sg * HK_CSTRIDE];
    sh += shadowCnt[size_t(b) * HK_CROW + sg * HK_CSTRIDE];
  }
  atomicAdd(&totals[0], e); atomicAdd(&totals[1], sh); atomicAdd(&totals[2], (unsigned long long)live[sg * HK_CSTRIDE]);
}
// ---- geometry re-layout (SceneDev::triRec): count the triangles of every mesh in the geometry table, then copy each
// triangle's vertex data into its record (bit copies of the arena: positions|u, normals|v, tangents, material id, shadow offset)
__global__ void k_geom_count(const int* __restrict__ globals, const float4* __restrict__ geom, int tableSize, int* __restrict__ triCount) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= tableSize) return;
  const int offset = globals[globals[HG_GEOM_TABLE_OFFS] + id];
  int n = 0;
  if (offset >= 0) n = reinterpret_cast<const HydraPlainMesh*>(geom + offset)->tIndicesNum / 3;
  triCount[id] = n;
}
__global__ void k_geom_fill(const int* __restrict__ globals, const float4* __restrict__ geom, int tableSize, const int* __restrict__ triBase, int total,
                            float4* __restrict__ rec, float4* __restrict__ tan) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    int lo = 0, hi = tableSize - 1;                 // last mesh whose base is <= t (empty meshes share a base with their successor)
    while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (triBase[mid] <= t) lo = mid; else hi = mid - 1; }
    HydraLiteHit h;
    h.t = 0.0f; h.geomId = lo; h.instId = 0; h.primId = t - triBase[lo];
    const TriData d = fetchTriFromMesh(h, geom + globals[globals[HG_GEOM_TABLE_OFFS] + lo]);
    float4* r = rec + size_t(t) * 8;
    r[0] = d.A1; r[1] = d.B1; r[2] = d.C1; r[3] = d.A2; r[4] = d.B2; r[5] = d.C2;
    r[6] = make_float4(as_float(d.matId), d.sRayOff, 0.0f, 0.0f);
    r[7] = make_float4(0, 0, 0, 0);
    tan[size_t(t) * 3] = d.At; tan[size_t(t) * 3 + 1] = d.Bt; tan[size_t(t) * 3 + 2] = d.Ct;
  }
}
// upload-time pass over the device copy of the node array: boxes of invalid children become NaN (see trav_run)
__global__ void k_prepare_bvh(int nodes, float4* __restrict__ bvh) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nodes; i += gridDim.x * blockDim.x) {
    float4 a = bvh[2 * size_t(i)], b = bvh[2 * size_t(i) + 1];
    if (uint32_t(as_int(a.w)) == HYDRA_BVH_INVALID && uint32_t(as_int(b.w)) == HYDRA_BVH_INVALID) {
      const float qnan = as_float(0x7fc00000);
      a.x = a.y = a.z = qnan; b.x = b.y = b.z = qnan;
      bvh[2 * size_t(i)] = a; bvh[2 * size_t(i) + 1] = b;
    }
  }
}
// InitRandomGen, shaders/trace.cl:6-13, with slot = stream * (w*h) + pixel: generator `stream` of owned pixel i.  The slot is
// formed in 32-bit wrap-around arithmetic like the reference's `a_seed + tid`; alloc_render_state keeps K * w * h below 2^32
// so that no two slots coincide.
// upload-time pass over the device copy of the triangle lists: label every triangle with the shading class of its material
// (hk_trace.h, HK_CLASS_SHIFT).  One thread per triangle leaf; idempotent (the class bits are replaced, not or-ed).
__global__ void k_tag_triangle_classes(int nLeaves, const int* __restrict__ headers, float4* __restrict__ tris, unsigned trisF4, SceneDev s, int geomTableSize) {
  for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < nLeaves; l += gridDim.x * blockDim.x) {
    const float4 hdr = tris[headers[l]];
    const int first = as_int(hdr.x), count = as_int(hdr.y);
    for (int k = 0; k < count; k++) {
      const unsigned a = unsigned(first + 3 * k);
      if (a + 2 >= trisF4) break;
      float4 d2 = tris[a + 1];
      const int primId = as_int(tris[a].w), geomId = HK_GEOM_ID(as_int(d2.w));
      int cls = 0;
      if (geomId >= 0 && geomId < geomTableSize && primId >= 0 && s.triBase[geomId] + primId < s.triBase[geomId + 1]) {
        const int matId = as_int(s.triRec[(size_t(s.triBase[geomId]) + size_t(primId)) * 8 + 6].x);
        if (matId >= 0 && matId < s.globals[HG_MAT_TABLE_SIZE] && s.matTable[matId] >= 0) cls = shadeClassOfMaterial(materialAt(s, matId));
      }
      d2.w = as_float(geomId | (cls << HK_CLASS_SHIFT));
      tris[a + 1] = d2;
    }
  }
}
__global__ void k_fill_srgb_lut(float* lut) { lut[threadIdx.x] = srgbByteToLinear((unsigned char)threadIdx.x); }   // SceneDev::srgbLut
__global__ void k_init_gens(int nOwned, int streams, const int* __restrict__ ownedPixels, unsigned npix, int seed, uint2* gens) {
  const size_t total = size_t(nOwned) * size_t(streams);
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += size_t(gridDim.x) * blockDim.x) {
    const unsigned stream = unsigned(i / size_t(nOwned)), pixel = unsigned(ownedPixels[i - size_t(stream) * nOwned]);
    const RandomGen g = RandomGenInit(int(unsigned(seed) + stream * npix + pixel));
    gens[i] = make_uint2(g.x, g.y);
  }
}

// the miss shader of a scene with a back-plate on rays handed in (hydra_hip_stage_environment)
__global__ void k_stage_environment(SceneDev s, int n, const float4* __restrict__ dir4, const float* __restrict__ in8, float4* __restrict__ out4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s.ptlSlot = -1;
  const float* in = in8 + size_t(i) * 8;
  const f3 c = haveBackPlate(s) ? environmentColorExtended(s, mk3(in[0], in[1], in[2]), xyz(dir4[i]), in[3], in[4] != 0.0f, uint32_t(as_int(in[5])), as_int(in[6]), as_int(in[7]))
                                : environmentColor(s, xyz(dir4[i]), in[3], in[4] != 0.0f, uint32_t(as_int(in[5])));
  out4[i] = mk4(c, 0.0f);
}


// ---------------------------------------------------------------------------------------- stage kernels (tests)
__global__ void k_stage_eye(SceneDev s, int n, int w, int h, const int* xy, const float4* offs, float4* pos4, float4* dir4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f3 p, d;
  MakeRandEyeRay(xy[2 * i], xy[2 * i + 1], w, h, offs[i], s, p, d);
  pos4[i] = mk4(p, 0.0f);
  dir4[i] = mk4(d, 0.0f);
}
__global__ void k_stage_surface(SceneDev s, int n, const float4* pos4, const float4* dir4, const HydraLiteHit* hits, float* out24) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* r = out24 + size_t(i) * 24;
  for (int k = 0; k < 24; k++) r[k] = 0.0f;
  const HydraLiteHit hit = hits[i];
  if (!HitSome(hit)) { r[17] = as_float(-1); return; }
  const SurfaceHit sh = evalSurface(s, xyz(pos4[i]), xyz(dir4[i]), hit);
  r[0] = sh.pos.x; r[1] = sh.pos.y; r[2] = sh.pos.z; r[3] = sh.normal.x; r[4] = sh.normal.y; r[5] = sh.normal.z;
  r[6] = sh.flatNormal.x; r[7] = sh.flatNormal.y; r[8] = sh.flatNormal.z; r[9] = sh.tangent.x; r[10] = sh.tangent.y; r[11] = sh.tangent.z;
  r[12] = sh.biTangent.x; r[13] = sh.biTangent.y; r[14] = sh.biTangent.z; r[15] = sh.texCoord.x; r[16] = sh.texCoord.y;
  r[17] = as_float(sh.matId); r[18] = sh.t; r[19] = sh.sRayOff; r[20] = sh.hfi ? 1.0f : 0.0f;
}
// one shading point with the random numbers handed in: the product's light pick / light sample / materialEval / BxDF sampling
// device functions exactly as k_bounce calls them (light_phase, direct_light_unoccluded's inputs, next_bounce_phase)
__global__ void k_stage_shade_point(SceneDev s, int n, const float* __restrict__ surf24, const float4* __restrict__ dir4, const int* __restrict__ flagsIn,
                                    const float4* __restrict__ rndLight4, const float* __restrict__ rands10, float* __restrict__ out28) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s.ptlSlot = (s.ptlIds != nullptr && uint32_t(s.ptlIds[i]) != HYDRA_INVALID_TEXTURE) ? i : -1;   // procedural texture lists handed in (hydra_hip_stage_set_proctex)
  const float* r = surf24 + size_t(i) * 24;
  float* o = out28 + size_t(i) * 28;
  for (int k = 0; k < 28; k++) o[k] = 0.0f;
  SurfaceHit surf;
  surf.pos = mk3(r[0], r[1], r[2]); surf.normal = mk3(r[3], r[4], r[5]); surf.flatNormal = mk3(r[6], r[7], r[8]);
  surf.tangent = mk3(r[9], r[10], r[11]); surf.biTangent = mk3(r[12], r[13], r[14]); surf.texCoord = mk2(r[15], r[16]);
  surf.matId = as_int(r[17]); surf.t = r[18]; surf.sRayOff = r[19]; surf.hfi = (r[20] != 0.0f);
  if (surf.matId < 0) { o[8] = as_float(-2); return; }
  const f3 ray_dir = xyz(dir4[i]);
  const uint32_t flags = uint32_t(flagsIn[i]);
  const float* mat = materialAt(s, surf.matId);
  const float4 rl = rndLight4[i];
  float pick = 1.0f;
  const int lightOffset = SelectRandomLightRev(rl.z, s, pick);
  o[7] = pick; o[8] = as_float(lightOffset);
  if (lightOffset >= 0) {
    ShadowSample sam;
    sam.pos = mk3(0, 0, 0); sam.color = mk3(0, 0, 0); sam.pdf = 0.0f; sam.isPoint = false;
    LightSampleRev(s, lightAt(s, lightOffset), mk3(rl.x, rl.y, rl.z), surf.pos, sam);
    const f3 sdir = normalize(sam.pos - surf.pos);
    o[0] = sam.pos.x; o[1] = sam.pos.y; o[2] = sam.pos.z; o[3] = sam.pdf;
    o[4] = sam.color.x; o[5] = sam.color.y; o[6] = sam.color.z; o[9] = sam.isPoint ? 1.0f : 0.0f;
    ShadeContext sc;
    sc.l = sdir; sc.v = ray_dir * (-1.0f); sc.n = surf.normal; sc.tc = surf.texCoord; sc.fn = surf.flatNormal; sc.tg = surf.tangent; sc.bn = surf.biTangent;
    const BxDFResult ev = materialEval(mat, sc, s);
    o[10] = ev.brdf.x; o[11] = ev.brdf.y; o[12] = ev.brdf.z; o[13] = ev.pdfFwd;
    o[14] = ev.btdf.x; o[15] = ev.btdf.y; o[16] = ev.btdf.z;
  }
  float rands[10];
  for (int k = 0; k < 10; k++) rands[k] = rands10[size_t(i) * 10 + k];
  MatSample ms;
  MaterialSampleAndEvalBxDF(mat, rands, surf, ray_dir, flags, s, ms);
  o[17] = ms.color.x; o[18] = ms.color.y; o[19] = ms.color.z; o[20] = ms.pdf;
  o[21] = ms.direction.x; o[22] = ms.direction.y; o[23] = ms.direction.z;
  o[24] = as_float(ms.flags); o[25] = as_float(int(flagsNextBounceLite(flags, ms, s)));
}
// one bounce of one path with every input handed in -- the phases k_bounce strings together (emission_phase, light_phase_with,
// direct_light_unoccluded, next_bounce_with, environmentColor), so that each can be checked against the reference's own stage kernel of
// the same name-sake (HitEnvOrLightKernel, LightSample, Shade, NextBounce; tests/golden/ref_stage_*.npz).  in16 per path: thr xyz, previous
// BSDF pdf, radiance xyz, previous bounce specular (0/1), the light's four random numbers, the number that picks the light, visibility of
// the shadow ray, Lite_Hit.instId (int bits), ray flags (int bits).  out40: see include/hydra_hip.h.
__global__ void k_stage_bounce(SceneDev s, int n, int depth, int maxDepth, const float4* __restrict__ pos4, const float4* __restrict__ dir4, const float* __restrict__ surf24,
                               const float* __restrict__ in16, const float* __restrict__ rands10, float* __restrict__ out40) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s.ptlSlot = (s.ptlIds != nullptr && uint32_t(s.ptlIds[i]) != HYDRA_INVALID_TEXTURE) ? i : -1;
  const float* r = surf24 + size_t(i) * 24;
  const float* in = in16 + size_t(i) * 16;
  float* o = out40 + size_t(i) * 40;
  for (int k = 0; k < 40; k++) o[k] = 0.0f;
  const f3 ray_pos = xyz(pos4[i]), ray_dir = xyz(dir4[i]);
  const f3 thr = mk3(in[0], in[1], in[2]), acc = mk3(in[4], in[5], in[6]);
  const float prevPdf = in[3];
  const bool prevSpec = in[7] != 0.0f;
  const uint32_t flags = uint32_t(as_int(in[15]));
  SurfaceHit surf;
  surf.pos = mk3(r[0], r[1], r[2]); surf.normal = mk3(r[3], r[4], r[5]); surf.flatNormal = mk3(r[6], r[7], r[8]);
  surf.tangent = mk3(r[9], r[10], r[11]); surf.biTangent = mk3(r[12], r[13], r[14]); surf.texCoord = mk2(r[15], r[16]);
  surf.matId = as_int(r[17]); surf.t = r[18]; surf.sRayOff = r[19]; surf.hfi = (r[20] != 0.0f);
  if (surf.matId < 0) {                                   // the ray left the scene: kernel_HitEnvironment, kernel_AddLastBouceContrib
    // in[14] of a ray that left the scene: its pixel, x | y << 16 (the reference's in_packXY), read by the back-plate
    const f3 env = haveBackPlate(s) ? environmentColorExtended(s, ray_pos, ray_dir, prevPdf, prevSpec, flags, as_int(in[14]) & 0xFFFF, (as_int(in[14]) >> 16) & 0xFFFF)
                                    : environmentColor(s, ray_dir, prevPdf, prevSpec, flags);
    const f3 fin = acc + (thr * env);
    o[0] = env.x; o[1] = env.y; o[2] = env.z; o[3] = as_float(1);
    o[34] = fin.x; o[35] = fin.y; o[36] = fin.z;
    return;
  }
  const float* mat = materialAt(s, surf.matId);
  f3 currColor = mk3(0, 0, 0);
  if (emission_phase(s, ray_pos, ray_dir, flags, prevPdf, prevSpec, as_int(in[14]), surf, mat, currColor)) {
    const f3 fin = acc + (thr * currColor);
    o[0] = currColor.x; o[1] = currColor.y; o[2] = currColor.z; o[3] = as_float(2);
    o[34] = fin.x; o[35] = fin.y; o[36] = fin.z;
    return;
  }
  if (depth >= maxDepth - 1) { o[3] = as_float(4); o[34] = acc.x; o[35] = acc.y; o[36] = acc.z; return; }
  LightPick lp;
  ShadowSample sam;
  light_phase_with(s, surf, make_float4(in[8], in[9], in[10], in[11]), in[12], lp, sam);
  o[4] = sam.pos.x; o[5] = sam.pos.y; o[6] = sam.pos.z; o[7] = sam.pdf; o[8] = sam.color.x; o[9] = sam.color.y; o[10] = sam.color.z;
  o[11] = sam.isPoint ? 1.0f : 0.0f; o[12] = lp.pickProb; o[13] = as_float(lp.lightOffset);
  o[14] = lp.shadowOrg.x; o[15] = lp.shadowOrg.y; o[16] = lp.shadowOrg.z; o[17] = lp.shadowOrg.w;
  o[18] = lp.shadowRayDir.x; o[19] = lp.shadowRayDir.y; o[20] = lp.shadowRayDir.z;
  f3 explicitColor = mk3(0, 0, 0);
  if (lp.lightOffset >= 0) explicitColor = direct_light_unoccluded(s, mat, surf, ray_dir, lp.shadowRayDir, lp.color, lp.pdfSigned, lp.pickProb) * in[13];
  o[21] = explicitColor.x; o[22] = explicitColor.y; o[23] = explicitColor.z;
  const f3 accum = acc + (thr * explicitColor);
  float4 oPos, oDir, oThr, oAcc;
  if (next_bounce_with(s, mat, surf, ray_dir, flags, rands10 + size_t(i) * 10, make_float4(thr.x, thr.y, thr.z, prevPdf), accum, 0.0f, oPos, oDir, oThr, oAcc)) {
    oThr.x *= in[13]; oThr.y *= in[13]; oThr.z *= in[13];   // a shadow catcher: what the next bounce of the production kernel does with the shadow ray's result
  }
  o[24] = oPos.x; o[25] = oPos.y; o[26] = oPos.z; o[27] = oDir.x; o[28] = oDir.y; o[29] = oDir.z; o[30] = oDir.w;
  o[31] = oThr.x; o[32] = oThr.y; o[33] = oThr.z; o[34] = oAcc.x; o[35] = oAcc.y; o[36] = oAcc.z; o[37] = oThr.w; o[38] = oAcc.w;
}
// ---- bidirectional building blocks (row f3, first milestone), one call per item with the random numbers handed in
__global__ void k_stage_light_fwd(SceneDev s, int n, const int* __restrict__ lightIds, const float4* __restrict__ rands4, float* __restrict__ out16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  LightSampleFwd sam;
  LightSampleForward(s, lightAt(s, lightIds[i]), rands4[i], 0.0f, sam);   // the reference-side fixture hands in rands2 = (0, 0) as well (oracle/ref_driver.cl)
  float* o = out16 + size_t(i) * 16;
  o[0] = sam.pos.x; o[1] = sam.pos.y; o[2] = sam.pos.z; o[3] = sam.dir.x; o[4] = sam.dir.y; o[5] = sam.dir.z;
  o[6] = sam.norm.x; o[7] = sam.norm.y; o[8] = sam.norm.z; o[9] = sam.color.x; o[10] = sam.color.y; o[11] = sam.color.z;
  o[12] = sam.pdfA; o[13] = sam.pdfW; o[14] = sam.cosTheta; o[15] = sam.isPoint ? 1.0f : 0.0f;
}
__global__ void k_stage_light_pdf_fwd(SceneDev s, int n, const int* __restrict__ lightIds, const float* __restrict__ cosTheta, float4* __restrict__ out4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const LightPdfFwd p = lightPdfFwd(s, lightAt(s, lightIds[i]), mk3(0.0f, 0.0f, 1.0f), cosTheta[i]);   // the direction the reference-side fixture hands in
  out4[i] = make_float4(p.pdfA, p.pdfW, p.pickProb, 0.0f);
}
__global__ void k_stage_camera_connect(SceneDev s, int n, const float4* __restrict__ pos4, const float4* __restrict__ norm4, const float2* __restrict__ disk2, float* __restrict__ out8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f3 camDir; float zDepth;
  const float f = CameraImageToSurfaceFactor(s, xyz(pos4[i]), xyz(norm4[i]), mk2(disk2[i].x, disk2[i].y), camDir, zDepth);
  const f2 scr = worldPosToScreenSpace(s, xyz(pos4[i]));
  float* o = out8 + size_t(i) * 8;
  o[0] = f; o[1] = camDir.x; o[2] = camDir.y; o[3] = camDir.z; o[4] = zDepth; o[5] = scr.x; o[6] = scr.y; o[7] = 0.0f;
}
__global__ void k_stage_mutate_kelemen(int n, const float* __restrict__ values, const float2* __restrict__ rands2, float p2, float p1, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = MutateKelemen(values[i], mk2(rands2[i].x, rands2[i].y), p2, p1);
}
__global__ void __launch_bounds__(256) k_mmlt_begin(SceneDev s, MmltView v, int nseg, int cap, MmltRays out, uint32_t* __restrict__ outCount) {
  const int seg = int(blockIdx.x) % nseg, bis = int(blockIdx.x) / nseg;
  const int i = (bis * nseg + seg) * int(blockDim.x) + int(threadIdx.x);   // chains are dealt to the segments in chunks of one block
  float4 cpos, cdir, lpos, ldir;
  bool ca = false, la = false;
  if (i < v.n) mmltBegin(s, v, i, cpos, cdir, ca, lpos, ldir, la);
  uint32_t* counter = outCount + seg * HK_CSTRIDE;
  const int dc = seg * cap + wave_compact_index(ca, counter);
  if (ca) { out.pos[dc] = cpos; out.dir[dc] = cdir; out.owner[dc] = i * 2; }
  const int dl = seg * cap + wave_compact_index(la, counter);
  if (la) { out.pos[dl] = lpos; out.dir[dl] = ldir; out.owner[dl] = i * 2 + 1; }
}
// ---- the Markov chains of IntegratorMMLT (hk_bidir.h): one thread per chain
__global__ void k_mmlt_init_chains(MmltChains c, int seed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c.n) mmltInitChain(c, i, seed);
}
__global__ void k_mmlt_pick_depth(MmltChains c, int* __restrict__ depth, const float* __restrict__ accum, int accumSize, int fixedDepth) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c.n) return;
  if (fixedDepth > 0) { depth[i] = fixedDepth; return; }
  RandomGen g = mchGen(c, CH_GEN, i);   // DoPassIndirectMLT(float4*), :348-356: d proportional to the average brightness of its paths
  float pdf = 1.0f;
  depth[i] = SelectIndexPropToOpt(rndFloat1_Pseudo(g), accum, accumSize, pdf);
  mchSetGen(c, CH_GEN, i, g);
}
__global__ void k_mmlt_fresh(MmltChains c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c.n) mmltFreshSample(c, i, c.xNew);
}
__global__ void k_mmlt_seed(MmltChains c, const float* __restrict__ out8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c.n) return;
  const int size = mmltStride(c.depth[i]);
  for (int j = 0; j < size; j++) c.xCur[size_t(j) * c.n + i] = c.xNew[size_t(j) * c.n + i];
  mmltSeedChain(c, i, out8);
}
__global__ void k_mmlt_mutate(MmltChains c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c.n) mmltMutate(c, i);
}
__global__ void k_mmlt_accept(MmltChains c, const float* __restrict__ out8, float bkScale, float* __restrict__ image4, int w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c.n) mmltAcceptReject(c, i, out8, bkScale, image4, w);
}
__global__ void k_sbdpt_pick_depth(MmltChains c, int* __restrict__ depth, int maxDepth) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c.n) sbdptPickDepth(c, i, depth, maxDepth);
}
__global__ void k_sbdpt_splat(int n, const int* __restrict__ depth, int maxDepth, const float* __restrict__ out8, float* __restrict__ image4, int w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sbdptSplat(i, depth, maxDepth, out8, image4, w);
}
__global__ void k_mmlt_sum(int n, const float* __restrict__ values, int stride, int offset, double* __restrict__ sum) {   // sum += values[i * stride + offset]
  __shared__ double part[256];
  double acc = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += double(values[size_t(i) * stride + offset]);
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (int(threadIdx.x) < k) part[threadIdx.x] += part[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(sum, part[0]);
}
__global__ void k_mmlt_scale_image(int n, const float4* __restrict__ in, float scale, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = make_float4(in[i].x * scale, in[i].y * scale, in[i].z * scale, in[i].w);
}
__global__ void k_mmlt_transpose_in(int n, int stride, int floats, const float* __restrict__ rows, float* __restrict__ planes) {   // rows[i][j] -> planes[j][i]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int j = 0; j < floats; j++) planes[size_t(j) * n + i] = rows[size_t(i) * stride + j];
}
__global__ void k_stage_random(int n, const int* seeds, int draws, float4* out4, uint2* state2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  RandomGen g = RandomGenInit(seeds[i]);
  for (int d = 0; d < draws; d++) out4[size_t(i) * draws + d] = rndFloat4_Pseudo(g);
  state2[i] = make_uint2(g.x, g.y);
}
// seed the path state from caller-provided primary rays and RandomGen states (stage_path_trace): path i plays pixel i
__global__ void k_stage_seed_paths(int n, const float4* __restrict__ pos4, const float4* __restrict__ dir4, const uint2* __restrict__ rng2, PathState S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  S.pos4[i] = make_float4(pos4[i].x, pos4[i].y, pos4[i].z, as_float(i));
  S.dir4[i] = make_float4(dir4[i].x, dir4[i].y, dir4[i].z, as_float(0));
  S.thr4[i] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
  S.acc4[i] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
  S.rng2[i] = rng2[i];
}

#ifdef HK_EXP_BOUNCE_STAMPS   /* timing experiment (tools/bounce_stamps.py): the phase stamps of k_bounce, summed over the translation units that instantiate it */
extern "C" int hydra_hip_debug_bounce_stamps(unsigned long long* out16, int reset) {
  for (int k = 0; k < 16; k++) out16[k] = 0;
  return (hk_bounce_stamps_read_lean(out16, reset) || hk_bounce_stamps_read_classic(out16, reset) || hk_bounce_stamps_read_nmap(out16, reset) ||
          hk_bounce_stamps_read_all(out16, reset) || hk_bounce_stamps_read_all45(out16, reset)) ? -1 : 0;
}
#endif
// ================================================================================================ host side
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct hydra_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int w = 0, h = 0;
  std::string err;
  char devName[256] = {0};
  int numCU = 256;

  size_t storageBytes[HYDRA_STORAGE_KINDS] = {0, 0, 0, 0, 0};   // bytes uploaded (a DevBuf may be larger)
  int sortPathsWanted = 1, sortPathsFromDepth = 1;   // options "sort_paths" / "sort_paths_from_bounce": group the paths of a workgroup by shading class in k_bounce
  int sceneTablesInLds = 2;          // option "scene_tables_in_lds"
  int lastGbufferUs = 0;             // device time of the last hydra_hip_eval_gbuffer (read-only option "last_gbuffer_device_us")
  DevBuf stageImg;                   // the staged scene tables gathered in LDS order (k_build_stage_image), HK_SCENE_LDS_MAX_BYTES
  DevBuf srgbLut;                    // 256 floats, see SceneDev::srgbLut; option "srgb_table" 0 disables it
  int srgbLutWanted = 1;
  DevBuf leafHeaders[4]; int leafHeadersNum[4] = {0, 0, 0, 0}; bool classDirty = true;   // triangle-leaf headers per tree; the class labels in the device triangle lists must be (re)written
  DevBuf bvhNodesTop, topQuads;      // node copy with tagged links to the cached quads + their indices (tree 0, persistent kernels)
  DevBuf topTriF4; int topTriCount = 0, topTrisWanted = 0;   // option "top_tris_in_lds" (0..HK_TOP_TRIS), read by the next upload_bvh
  int shadowUnordered = 1;           // option "shadow_unordered": any-hit rays take a quad's children in stored order (hk_trace.h, trav_run_vote)
  int topCount = 0, topWanted = HK_TOP_QUADS;   // option "top_quads_in_lds" (0..HK_TOP_QUADS), read by the next upload_bvh
  DevBuf bvhAlpha[4];                // alpha tables of the trees that have one (uint2 per float4 of the triangle list + the opacity samplers)
  DevBuf globals, storage[HYDRA_STORAGE_KINDS], bvhNodes[4], bvhTris[4], instMat, instLight, triRec, triTan, triBase, remapLists, remapTable, remapInst;
  size_t globalsWords = 0;
  int haveInst[4] = {0, 0, 0, 0};
  size_t bvhNodeBytes[4] = {0, 0, 0, 0}, bvhTriBytes[4] = {0, 0, 0, 0};
  int treesNum = 0, instNum = 0;
  int remapListsSize = 0, remapTableSize = 0, remapInstSize = 0;
  std::vector<int32_t> hostHeader;   // copy of the first words of the globals blob (trace depth, ...)
  bool leafEnc[4] = {false, false, false, false};   // device node copy of tree i has triangle counts in its leaf links
  int lightFeatures = HK_FEAT_CLASSIC, matFeatures = HK_FEAT_CLASSIC, sceneFeatures = HK_FEAT_CLASSIC;   // HK_FEAT_* the uploaded scene needs; 0 => the lean k_bounce
  bool matDirty = true;              // material arena or material table changed since validate_materials last passed
  std::vector<float> hostMaterials;  // host copy of the material arena and table, for validate_materials only
  std::vector<int32_t> hostMatTable;
  std::vector<int32_t> hostTexAuxTable;   // aux texture id -> offset in the aux arena (normal maps), for validate_materials
  std::vector<int32_t> hostTexTable;      // texture id -> offset in the texture arena, for the back-plate check of run_bounces
  bool geomDirty = true;             // triRec/triTan/triBase must be rebuilt (geometry arena or geometry table changed)
  // procedural textures: the scene's compiled program (hydra_hip_proctex_compile), how many a material head lists at most (validate_materials), the per-path lists
  // of the bounce in flight (ids: int[ptlMax][slots], colours: uint2[ptlMax][slots]) and the lists handed to the stage entries (hydra_hip_stage_set_proctex)
  HkProcTexProgram* procTex = nullptr;
  int ptlMax = 0;
  DevBuf ptlIds, ptlVals;
  DevBuf stagePtlIds, stagePtlVals; int stagePtlN = 0, stagePtlMax = 0;

  // render state
  int rank = 0, world = 1, tile = 64;
  int N = 0;                         // owned pixels = live paths at bounce 0
  int streamsWanted = 0;             // option "samples_in_flight": samples per pixel traced concurrently, 0 = by resolution
  int streams = 1;
  int leafEncWanted = 1;             // option "leaf_count_links": read by the next upload_bvh
  int streamMajor = 1;               // option "path_order": 1 = stream-major (default), 0 = pixel-major (samples of a pixel share a wave); measured equal
  DevBuf ownedPixels;                // the N owned pixels in slot order (k_accumulate)
  int nsegWanted = 32;               // option "queue_segments" (a multiple of the 8 XCDs: segment s is always worked on by XCD s % 8)
  int nseg = 1, segCap = 0;          // segmented path queues (see SegQ): nseg * segCap slots
  DevBuf liveInit;                   // one counter row holding the initial per-segment path counts
  DevBuf gens, accumInternal, contrib, hits, live, shadowCnt, totals;
  float4* accum = nullptr;           // internal or external
  bool externalAccum = false;
  DevBuf sPos, sDir, sThr, sAcc, sRng;
  DevBuf tPos, tDir, tThr, tAcc, tRng, sPend, tPend, shDir;   // fused form: second S set (ping-pong), pending estimates, shadow directions
  int fusedBounce = 1;               // option "fused_bounce": 1 = k_bounce, 0 = k_hit + k_shade with the M record
  DevBuf mDir, mThr, mAcc, mRng, mSurfA, mSurfB, mRecC, mRecD, mRecE, mShadowOrg, mVis;
  bool stateAllocated = false, gensReady = false;
  int seed = 777;
  float spp = 0.0f;

  bool stageTiming = false;
  bool travCounters = false;
  int traceMode = 1;          // 1 = persistent dynamic fetch (k_trace_dyn, default: 8-35 % faster once the refill counters are per segment), 0 = one ray per lane
  int traceRaysPerLane = 1;   // persistent kernels: blocks beyond count / (128 * this) leave at once
  int traceMinActive = 40;    // suspend-and-refill threshold of k_trace_dyn (lanes of 64): the vote's optimum (profiles/r03/vote_minactive_*.log); the loop nest's was 48
  int traceVote = 1;          // option "trace_vote": the persistent kernels schedule quad / triangle / instance steps by wave vote (hk_trace.h, trav_run_vote); 0 = the reference's loop nest
  int traceVoteW[3] = {1, 1, 2};   // options "trace_vote_wq / _wt / _wi": weights of the vote
  int shadeWaves = 3;         // launch-bounds variant of k_bounce / k_hit / k_shade (3, 4 or 5 waves per SIMD); 3 = no spills, measured fastest for the fused kernel
  int shadeBlocksPerCU = 256;        // grid cap of the bounce kernels: 16 -> 128..1024 takes 5 % off k_bounce (finer tail, pass_sweep_shade_blocks_final.log)
  int staticBlocksPerCU = 16; // grid cap of the one-ray-per-lane traversal kernels (128-thread blocks per CU)   // grid cap of the 256-thread kernels, in blocks per CU
  int traceBlocksPerCU = 12;  // resident 128-thread blocks per CU for the persistent kernels
  // IntegratorMMLT run (hydra_hip_mmlt_*): chain planes, the two x-vector sets, the work buffers of F, the indirect image
  struct MmltRun {
    bool active = false;
    int n = 0, maxD = 0, firstBounce = 0;
    int w = 0, h = 0;                  // frame the run's images were allocated for (row stride of every splat)
    DevBuf ch, depth, xCur, xNew, out8, image, accum, sum, scaled;
    DevBuf sbDepth, sbImage; unsigned long long sbSamples = 0;   // the SBDPT passes of the same run (hydra_hip_sbdpt_pass)
    DevBuf st, rayPos[2], rayDir[2], rayOwner[2], hits, counts, eyePos, eyeDir, eyeHit, shPos, shDir, shVis;
    float avgB[HK_MMLT_MAX_DEPTH + 2] = {0};
    float avgBrightness = 0.0f;
    unsigned long long mutations = 0;
  } mmlt;
  DevBuf fetchCnt;            // refill counters of the persistent kernels: [2*bounce + (shadow ? 1 : 0)], + 1 spare for stage calls
  DevBuf travTotals;   // [bounce][ext|shadow][rays, quads, insts, leaves, tris, out-of-range fetches]
  // RCCL exchange of the accumulator without Python (hydra_hip_comm_*): function table of the dlopen()ed library, communicator, staging
  struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
  } rccl;
  ncclComm_t comm = nullptr;
  int commRank = -1, commWorld = 0;
  DevBuf commPacked, commRecv, commPixels;     // this rank's packed pixels; (root) the other ranks' packed pixels and their pixel indices
  std::vector<long long> commCount;            // (root) owned pixels per rank
  int commW = 0, commH = 0, commTile = 0;      // frame and tile size commCount / commPixels / commRecv were built for
  DevBuf commMeta;                             // [world][4] ints: what every rank is about to send (comm_agree)
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double tTrace = 0, tHit = 0, tShadow = 0, tShade = 0, tRaygen = 0, tAccum = 0, tPass = 0;
  uint64_t nTrace = 0, nShadow = 0;   // launches folded into tTrace / tShadow
  std::vector<hipEvent_t> evPool;
  struct EvSpan { int a, b, kind, depth; };
  double tDepth[HK_MAX_DEPTH][3] = {};   // per bounce: closest-hit traversal, bounce kernel(s), shadow traversal (ms)
  std::vector<EvSpan> spans;
  size_t evCursor = 0;
};

static std::string g_createError;

#define HCHECK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      c->err = std::string(#call) + ": " + hipGetErrorString(e_);                                      \
      return HYDRA_HIP_EDEVICE;                                                                        \
    }                                                                                                  \
  } while (0)

static int fail(hydra_hip_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_createError = msg;
  return code;
}

static int dev_alloc(hydra_hip_ctx* c, DevBuf& b, size_t bytes) {
  if (b.p != nullptr && b.bytes >= bytes && bytes > 0) return HYDRA_HIP_OK;
  if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
  if (bytes == 0) return HYDRA_HIP_OK;
  if (hipMalloc(&b.p, bytes) != hipSuccess) { c->err = "hipMalloc failed for " + std::to_string(bytes) + " bytes"; return HYDRA_HIP_ENOMEM; }
  b.bytes = bytes;
  return HYDRA_HIP_OK;
}
static int dev_upload(hydra_hip_ctx* c, DevBuf& b, const void* src, size_t bytes) {
  const size_t alloc = bytes > 0 ? (bytes + 15) / 16 * 16 : 16;   // never hand a null pointer to a kernel; whole float4s (LDS staging copies in 16-byte pieces)
  int rc = dev_alloc(c, b, alloc);
  if (rc != HYDRA_HIP_OK) return rc;
  if (bytes > 0) HCHECK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
  HCHECK(hipStreamSynchronize(c->stream));       // the caller may free `src` right after the call
  return HYDRA_HIP_OK;
}
static void dev_free(DevBuf& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }

static SceneDev make_scene_tree(const hydra_hip_ctx* c, int tree);
static SceneDev make_scene(const hydra_hip_ctx* c) { return make_scene_tree(c, 0); }
static SceneDev make_scene_tree(const hydra_hip_ctx* c, int tree) {
  SceneDev s;
  s.globals = static_cast<const int*>(c->globals.p);
  s.texStorage = static_cast<const int4*>(c->storage[HYDRA_STORAGE_TEXTURES].p);
  s.geomStorage = static_cast<const float4*>(c->storage[HYDRA_STORAGE_GEOM].p);
  s.triRec = static_cast<const float4*>(c->triRec.p);
  s.triTan = static_cast<const float4*>(c->triTan.p);
  s.triBase = static_cast<const int*>(c->triBase.p);
  s.matStorage = static_cast<const float4*>(c->storage[HYDRA_STORAGE_MATERIALS].p);
  s.pdfStorage = static_cast<const float4*>(c->storage[HYDRA_STORAGE_PDFS].p);
  s.bvh = static_cast<const float4*>(c->bvhNodes[tree].p);
  s.bvhBytes = unsigned(c->bvhNodeBytes[tree]);
  s.leafEnc = c->leafEnc[tree] ? 1 : 0;
  s.alpha = static_cast<const uint2*>(c->bvhAlpha[tree].p);
  s.bvhTop = tree == 0 ? static_cast<const float4*>(c->bvhNodesTop.p) : nullptr;   // the LDS caches of the persistent kernels exist for tree 0 only
  s.topQuads = tree == 0 ? static_cast<const int*>(c->topQuads.p) : nullptr;
  s.topCount = (tree == 0 && c->bvhNodesTop.p && c->topQuads.p) ? c->topCount : 0;
  s.topTriF4 = static_cast<const int*>(c->topTriF4.p);
  s.topTriCount = (s.topCount > 0 && c->topTriF4.p && s.alpha == nullptr) ? c->topTriCount : 0;
  s.trisBytes = unsigned(c->bvhTriBytes[tree]);
  s.tris = static_cast<const float4*>(c->bvhTris[tree].p);
  s.haveInst = c->haveInst[tree];
  s.instMatrices = static_cast<const float4*>(c->instMat.p);
  s.instLightInstId = static_cast<const int*>(c->instLight.p);
  s.instNum = c->instNum;
  s.remapLists = c->remapListsSize > 0 ? static_cast<const int*>(c->remapLists.p) : nullptr; s.remapListsSize = c->remapListsSize;
  s.remapTable = c->remapTableSize > 0 ? static_cast<const int*>(c->remapTable.p) : nullptr; s.remapTableSize = c->remapTableSize;
  s.remapInst = c->remapInstSize > 0 ? static_cast<const int*>(c->remapInst.p) : nullptr;   s.remapInstSize = c->remapInstSize;
  s.srgbLut = c->srgbLutWanted ? static_cast<const float*>(c->srgbLut.p) : nullptr;
  s.matBase = reinterpret_cast<const float*>(s.matStorage);
  const bool hdr = c->hostHeader.size() > size_t(HG_LIGHTS_OFFS) && s.globals != nullptr;
  s.matTable = hdr ? s.globals + c->hostHeader[HG_MAT_TABLE_OFFS] : nullptr;
  s.lightsBase = hdr ? reinterpret_cast<const float*>(s.globals + c->hostHeader[HG_LIGHTS_OFFS]) : nullptr;
  s.texTable = hdr ? s.globals + c->hostHeader[HG_TEX_TABLE_OFFS] : nullptr;
  s.texAuxStorage = static_cast<const int4*>(c->storage[HYDRA_STORAGE_TEXTURES_AUX].p);
  s.texAuxTable = hdr ? s.globals + c->hostHeader[HG_TEXAUX_TABLE_OFFS] : nullptr;
  s.hdr = s.globals;
  s.lselRev = hdr ? reinterpret_cast<const float*>(s.globals + c->hostHeader[HG_LSEL_REV_OFFS]) : nullptr;
  s.ptlIds = nullptr; s.ptlVals = nullptr; s.ptlStride = 0; s.ptlMax = 0; s.ptlSlot = -1;   // procedural textures: set by the callers that run them (trace_pass, the stage entries)
  return s;
}
static bool scene_ready(const hydra_hip_ctx* c) {
  return c->globals.p && c->bvhNodes[0].p && c->bvhTris[0].p && c->instMat.p && c->instLight.p &&
         c->storage[HYDRA_STORAGE_GEOM].p && c->storage[HYDRA_STORAGE_MATERIALS].p && c->treesNum >= 1;
}
static int grid_for(const hydra_hip_ctx* c, int n, int block, int blocksPerCU) {
  int g = (n + block - 1) / block;
  const int cap = c->numCU * blocksPerCU;
  if (g > cap) g = cap;
  return g < 1 ? 1 : g;
}
// grid of a segmented-queue kernel: blocks-per-segment from the busiest possible segment, times nseg
static int seg_grid(const hydra_hip_ctx* c, const SegQ& q, int block, int blocksPerCU) {
  const int perSeg = q.counts ? q.cap : q.countImm;
  int bps = (perSeg + block - 1) / block;
  const int cap = std::max(1, c->numCU * blocksPerCU / q.nseg);
  if (bps > cap) bps = cap;
  return std::max(1, bps) * q.nseg;
}
static SegQ seg_q(const uint32_t* counts, int countImm, int nseg, int cap) { SegQ q; q.counts = counts; q.countImm = countImm; q.nseg = nseg; q.cap = cap; return q; }

// Every material a table entry reaches must be one this layer shades: the device's leaf dispatch would otherwise return
// black for it without a word.  Walks the blend trees (cmaterial.h:2180-2207 addressing) on a host copy of the arena.
static int validate_materials(hydra_hip_ctx* c) {
  if (!c->matDirty) return HYDRA_HIP_OK;
  const size_t floats = c->hostMaterials.size();
  auto word = [](const float* m, int i) { int32_t v; memcpy(&v, m + i, 4); return v; };
  int feat = 0, ptlMax = 0;
  for (size_t id = 0; id < c->hostMatTable.size(); id++) {
    const int32_t offs = c->hostMatTable[id];
    if (offs < 0) continue;
    size_t stack[2 * 8];
    int top = 0, visited = 0, procIds = 0;
    const float* headOfTree = c->hostMaterials.data() + size_t(offs) * 4;
    stack[top++] = size_t(offs) * 4;
    if (size_t(offs) * 4 + HM_NODE_FLOATS <= floats && (word(c->hostMaterials.data() + size_t(offs) * 4, HM_FLAGS) & HMF_HAVE_PROC_TEXTURES) != 0) {
      // the head lists the ids (cglobals.h:2732-2753) and points at its argument table, one node of (id, offset) pairs whose last word is the count, followed by
      // the argument words (RenderDriverRTE_ProcTex.cpp:191-252, PlainMaterialConverter.cpp:1865-1874): all of it must lie inside the arena
      const float* head = c->hostMaterials.data() + size_t(offs) * 4;
      const std::string who = "material " + std::to_string(id);
      int n = 0;
      while (n < 16 && uint32_t(word(head, HM_PROC_TEX_IDS + n)) != HYDRA_INVALID_TEXTURE) n++;
      if (n == 0) return fail(c, HYDRA_HIP_EINVAL, "materials: " + who + " is flagged as having procedural textures but its head lists none");
      const long long tab = word(head, HM_PROC_TEX_TABLE);
      if (tab <= 0 || size_t(offs) * 4 + size_t(tab) + HM_NODE_FLOATS > floats) return fail(c, HYDRA_HIP_EINVAL, "materials: " + who + ": procedural texture argument table outside the material arena");
      const float* table = head + tab;
      const int entries = word(table, HM_NODE_FLOATS - 1);
      if (entries < 0 || entries > (HM_NODE_FLOATS - 1) / 2) return fail(c, HYDRA_HIP_EINVAL, "materials: " + who + ": bad procedural texture argument table");
      for (int e = 0; e < entries; e++) {
        const int o = word(table, 2 * e + 1);
        if (o < -1 || size_t(offs) * 4 + size_t(tab) + HM_NODE_FLOATS + size_t(o < 0 ? 0 : o) > floats) return fail(c, HYDRA_HIP_EINVAL, "materials: " + who + ": procedural texture arguments outside the material arena");
      }
      ptlMax = std::max(ptlMax, n);
      procIds = n;
    }
    while (top > 0) {
      const size_t at = stack[--top];
      const std::string who = "material " + std::to_string(id) + " (node at float " + std::to_string(at) + ")";
      if (at + HM_NODE_FLOATS > floats) return fail(c, HYDRA_HIP_EINVAL, "materials: " + who + " lies outside the material arena");
      if (++visited > 64) return fail(c, HYDRA_HIP_EINVAL, "materials: " + who + ": blend tree with more than 64 nodes (cycle?)");
      const float* m = c->hostMaterials.data() + at;
      const int type = word(m, HM_TYPE);
      bool procNormal = false;   // a procedural normal map: the slot holds the texture id and the value comes from the path's list (PlainMaterialConverter.cpp:1396-1399)
      if (type != HMT_BLEND_MASK && uint32_t(word(m, HM_NORMAL_TEX)) != HYDRA_INVALID_TEXTURE && procIds > 0) {
        const int so = word(m, HM_NORMAL_TEX_MATRIX);
        if (so > 0 && so * 4 + HS_TEXID < HM_NODE_FLOATS)
          for (int k = 0; k < procIds; k++) procNormal = procNormal || (word(headOfTree, HM_PROC_TEX_IDS + k) == word(m, so * 4 + HS_TEXID));
        if (procNormal) feat |= HK_FEAT_NMAP;
      }
      if (!procNormal && type != HMT_BLEND_MASK && uint32_t(word(m, HM_NORMAL_TEX)) != HYDRA_INVALID_TEXTURE) {   // normal-mapped leaf (a blend node may carry the ids too: never read)
        const int auxId = word(m, HM_NORMAL_TEX);
        if (auxId < 0 || size_t(auxId) >= c->hostTexAuxTable.size()) return fail(c, HYDRA_HIP_EINVAL, "materials: " + who + ": normal-map id outside the aux texture table");
        const int offs = c->hostTexAuxTable[size_t(auxId)];
        if (offs < 0 || (size_t(offs) + 1) * 16 > c->storageBytes[HYDRA_STORAGE_TEXTURES_AUX]) return fail(c, HYDRA_HIP_EINVAL, "materials: " + who + ": its normal map is not in the aux texture arena");
        feat |= HK_FEAT_NMAP;
      }
      if (type == HMT_BLEND_MASK) {
        const int o1 = word(m, HM_BLEND_MAT1), o2 = word(m, HM_BLEND_MAT2);
        if (o1 <= 0 || o2 <= 0 || top + 2 > 16) return fail(c, HYDRA_HIP_EINVAL, "materials: " + who + ": bad blend children");
        stack[top++] = at + size_t(o1) * HM_NODE_FLOATS;
        stack[top++] = at + size_t(o2) * HM_NODE_FLOATS;
        continue;
      }
      if (type == HMT_OREN_NAYAR) feat |= HK_FEAT_OREN_NAYAR;
      if (type == HMT_GLASS || type == HMT_THIN_GLASS) feat |= HK_FEAT_GLASS;
      if (type == HMT_GGX) feat |= HK_FEAT_GGX;
      if (type == HMT_TRANSLUCENT) feat |= HK_FEAT_TRANSLUCENT;
      if (type == HMT_BLINN) feat |= HK_FEAT_BLINN;
      if (type == HMT_BECKMANN || type == HMT_TRGGX) feat |= HK_FEAT_ANISO;
      const bool known = (type == HMT_PHONG || type == HMT_MIRROR || type == HMT_THIN_GLASS || type == HMT_GLASS || type == HMT_LAMBERT ||
                          type == HMT_OREN_NAYAR || type == HMT_EMISSIVE || type == HMT_GGX || type == HMT_TRANSLUCENT || type == HMT_BLINN || type == HMT_BECKMANN || type == HMT_TRGGX || type == HMT_SHADOW_MATTE);
      if (!known)
        return fail(c, HYDRA_HIP_EINVAL, "materials: " + who + " has BxDF class " + std::to_string(type) +
                                             "; the HIP layer implements phong, Blinn (Torrance-Sparrow), Beckmann, TRGGX, GGX, mirror, thin glass, glass, translucent, shadow matte (the CPU integrator's: black), lambert, oren-nayar, blend mask and emissive only");
    }
  }
  c->matFeatures = feat;
  c->ptlMax = ptlMax;
  c->sceneFeatures = c->matFeatures | c->lightFeatures;
  c->matDirty = false;
  return HYDRA_HIP_OK;
}

// (re)build the per-triangle records when the geometry arena or the geometry table changed; called by every entry point
// that shades.  Two small kernels and one read-back of the per-mesh triangle counts.
static int prepare_geometry(hydra_hip_ctx* c) {
  if (!c->geomDirty) return HYDRA_HIP_OK;
  const int tableSize = c->hostHeader[HG_GEOM_TABLE_SIZE];
  if (tableSize <= 0 || tableSize > (1 << 24)) return fail(c, HYDRA_HIP_EINVAL, "geometry table size out of range");
  int rc;
  if ((rc = dev_alloc(c, c->triBase, (size_t(tableSize + 1) * 4 + 15) / 16 * 16)) != 0) return rc;   // whole float4s: k_bounce may stage it in LDS
  const int* globals = static_cast<const int*>(c->globals.p);
  const float4* geom = static_cast<const float4*>(c->storage[HYDRA_STORAGE_GEOM].p);
  hipLaunchKernelGGL(k_geom_count, dim3((tableSize + 255) / 256), dim3(256), 0, c->stream, globals, geom, tableSize, static_cast<int*>(c->triBase.p));
  std::vector<int> counts(size_t(tableSize) + 1, 0);
  HCHECK(hipMemcpyAsync(counts.data(), c->triBase.p, size_t(tableSize) * 4, hipMemcpyDeviceToHost, c->stream));
  HCHECK(hipStreamSynchronize(c->stream));
  long long total = 0;
  for (int i = 0; i < tableSize; i++) {
    if (counts[i] < 0) return fail(c, HYDRA_HIP_EINVAL, "geometry table: negative triangle count in mesh " + std::to_string(i));
    const int n = counts[i];
    counts[i] = int(total);
    total += n;
  }
  counts[tableSize] = int(total);
  if (total <= 0 || total >= (1ll << 31)) return fail(c, HYDRA_HIP_EINVAL, "geometry table: no triangles (or 2^31 and more)");
  HCHECK(hipMemcpyAsync(c->triBase.p, counts.data(), size_t(tableSize + 1) * 4, hipMemcpyHostToDevice, c->stream));
  if ((rc = dev_alloc(c, c->triRec, size_t(total) * 128)) != 0) return rc;
  if ((rc = dev_alloc(c, c->triTan, size_t(total) * 48)) != 0) return rc;
  hipLaunchKernelGGL(k_geom_fill, dim3(grid_for(c, int(total), 256, 8)), dim3(256), 0, c->stream, globals, geom, tableSize,
                     static_cast<const int*>(c->triBase.p), int(total), static_cast<float4*>(c->triRec.p), static_cast<float4*>(c->triTan.p));
  HCHECK(hipGetLastError());
  HCHECK(hipStreamSynchronize(c->stream));     // `counts` must outlive the copy
  c->geomDirty = false;
  return HYDRA_HIP_OK;
}

// Tile ownership: the T x T tiles of the image plane are ordered along the Morton (Z) curve of their tile
// coordinates and dealt round-robin, the i-th tile of that order going to rank i % world.  Every rank then owns tiles from
// all over the frame (path-length differences between image regions average out without any exchange of cost figures),
// whatever the number of tiles per row, and walks its own tiles in an order that keeps neighbours together.
static inline uint32_t morton2(uint32_t x, uint32_t y) {
  auto spread = [](uint32_t v) { v &= 0xffffu; v = (v | (v << 8)) & 0x00ff00ffu; v = (v | (v << 4)) & 0x0f0f0f0fu; v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u; return v; };
  return spread(x) | (spread(y) << 1);
}
// tiles (row-major tile index) in Morton order
static std::vector<int> morton_tile_order(int tilesX, int tilesY) {
  std::vector<std::pair<uint32_t, int>> t;
  t.reserve(size_t(tilesX) * tilesY);
  for (int ty = 0; ty < tilesY; ty++) for (int tx = 0; tx < tilesX; tx++) t.push_back({morton2(uint32_t(tx), uint32_t(ty)), ty * tilesX + tx});
  std::sort(t.begin(), t.end());
  std::vector<int> out(t.size());
  for (size_t i = 0; i < t.size(); i++) out[i] = t[i].second;
  return out;
}
// (re)label the triangles of the device triangle lists with their material's shading class; needs the per-triangle records
// (prepare_geometry), the material arena and the globals blob.  Called by every entry point that traces.
static int prepare_classes(hydra_hip_ctx* c) {
  if (!c->classDirty) return HYDRA_HIP_OK;
  if (c->triRec.p && c->storage[HYDRA_STORAGE_MATERIALS].p && c->hostHeader.size() > size_t(HG_GEOM_TABLE_SIZE)) {
    const int tableSize = c->hostHeader[HG_GEOM_TABLE_SIZE];
    for (int tree = 0; tree < 4 && tableSize > 0 && tableSize < (1 << HK_CLASS_SHIFT); tree++) {
      if (c->leafHeadersNum[tree] <= 0 || !c->leafHeaders[tree].p || !c->bvhTris[tree].p) continue;
      const SceneDev s = make_scene(c);
      hipLaunchKernelGGL(k_tag_triangle_classes, dim3(grid_for(c, c->leafHeadersNum[tree], 256, 8)), dim3(256), 0, c->stream, c->leafHeadersNum[tree],
                         static_cast<const int*>(c->leafHeaders[tree].p), static_cast<float4*>(c->bvhTris[tree].p), unsigned(c->bvhTriBytes[tree] / 16), s, tableSize);
      HCHECK(hipGetLastError());
    }
  }
  c->classDirty = false;
  return HYDRA_HIP_OK;
}

// slot -> pixel map: this rank's tiles in Morton order, pixels inside a tile in 8x8 blocks so that one wave = one block
static void build_slot_map(int w, int h, int T, int rank, int world, std::vector<int>* out, long long* count) {
  if (out) out->clear();
  long long n = 0;
  const int tilesX = (w + T - 1) / T, tilesY = (h + T - 1) / T;
  const std::vector<int> order = morton_tile_order(tilesX, tilesY);
  for (size_t i = 0; i < order.size(); i++) {
    if (world > 1 && int(i % size_t(world)) != rank) continue;
    const int tx = order[i] % tilesX, ty = order[i] / tilesX;
    const int x0 = tx * T, y0 = ty * T, x1 = std::min(x0 + T, w), y1 = std::min(y0 + T, h);
    n += (long long)(x1 - x0) * (y1 - y0);
    if (!out) continue;
    for (int by = y0; by < y1; by += 8)
      for (int bx = x0; bx < x1; bx += 8)
        for (int y = by; y < std::min(by + 8, y1); y++)
          for (int x = bx; x < std::min(bx + 8, x1); x++) out->push_back(y * w + x);
  }
  if (count) *count = n;
}

// samples per pixel in flight when the caller does not say: enough paths to keep late bounces (a few percent of the
// paths survive to bounce 8) above the size where kernels stop scaling down, within ~10 GB of path state at 1080p.
// Depends on the resolution only -- never on the rank count -- so every rank draws the same streams.
static int auto_streams(size_t npix) {
  int k = 1;
  while (k < 16 && size_t(2 * k) * npix <= size_t(40) << 20) k *= 2;
  return k;
}

// Sizes of everything a render state holds; pure host arithmetic (no device), shared by alloc_render_state and by
// hydra_hip_plan_render_state so that the limits can be checked without a GPU.
static int plan_render_state(int w, int h, int rank, int world, int tile, int streamsWanted, int nsegWanted, int fused, HydraStatePlan* p, std::string* err) {
  memset(p, 0, sizeof(*p));
  if (w <= 0 || h <= 0 || world < 1 || rank < 0 || rank >= world || tile < 8 || (tile % 8) != 0) { *err = "plan_render_state: bad arguments"; return HYDRA_HIP_EINVAL; }
  const size_t npix = size_t(w) * h;
  long long owned = 0;
  build_slot_map(w, h, tile, rank, world, nullptr, &owned);
  const int K = streamsWanted > 0 ? streamsWanted : auto_streams(npix);
  p->samples_in_flight = K;
  p->owned_pixels = owned;
  // generator slots stream * w * h + pixel are 32-bit (k_init_gens); path/record indices stream * N + pixelIndex are ints
  if (size_t(K) * npix > size_t(0xffffffffu)) { *err = "samples_in_flight * width * height must stay below 2^32"; return HYDRA_HIP_EINVAL; }
  if ((long long)K * owned > 0x7fffffffll) { *err = "samples_in_flight * owned pixels must stay below 2^31 (use more ranks or fewer samples in flight)"; return HYDRA_HIP_EINVAL; }
  // Paths are dealt to the queue segments in 256-path chunks round-robin (see k_raygen), so every segment sees the whole
  // image and path-length differences between image regions do not unbalance them.
  const long long maxPaths = owned * K;
  const long long maxChunks = (maxPaths + 255) / 256;
  const int nseg = int(std::max<long long>(1, std::min<long long>(std::min(nsegWanted, HK_MAX_SEG), maxChunks)));
  const long long cap = std::max<long long>(1, (maxChunks + nseg - 1) / nseg) * 256;
  if (cap * nseg > 0x7fffffffll) { *err = "too many path slots"; return HYDRA_HIP_EINVAL; }
  p->paths = maxPaths; p->segments = nseg; p->segment_capacity = cap;
  const long long slots = cap * nseg;
  // path state: only the arrays of the form in use are held (fused: 13 float4 + 2 uint2 + 1 float, split: 14 float4 + 2 uint2 + 1 float per slot)
  p->path_state_bytes = slots * ((fused ? 13 : 14) * 16 + 2 * 8 + 4);
  p->generator_bytes = std::max<long long>(1, owned) * K * 8;
  p->contrib_bytes = std::max<long long>(1, owned) * K * 16;
  p->owned_map_bytes = owned * 4;
  p->total_bytes = p->path_state_bytes + p->generator_bytes + p->contrib_bytes + p->owned_map_bytes;
  return HYDRA_HIP_OK;
}

static int alloc_render_state(hydra_hip_ctx* c) {
  HydraStatePlan plan;
  { std::string e; const int prc = plan_render_state(c->w, c->h, c->rank, c->world, c->tile, c->streamsWanted, c->nsegWanted, c->fusedBounce, &plan, &e); if (prc) return fail(c, prc, e); }
  std::vector<int> order;
  build_slot_map(c->w, c->h, c->tile, c->rank, c->world, &order, nullptr);
  c->N = int(order.size());
  const size_t npix = size_t(c->w) * c->h;
  const int K = plan.samples_in_flight;
  if (K != c->streams) c->gensReady = false;
  c->streams = K;
  c->nseg = int(plan.segments);
  const long long cap = plan.segment_capacity;
  c->segCap = int(cap);
  std::vector<uint32_t> init(size_t(K) * HK_CROW, 0u);
  for (int ns = 1; ns <= K; ns++) {
    const long long paths = (long long)c->N * ns, chunks = (paths + 255) / 256;
    for (int sg = 0; sg < c->nseg; sg++) {
      if (sg >= chunks) continue;
      const long long mine = (chunks - sg + c->nseg - 1) / c->nseg;            // chunks sg, sg + nseg, ...
      long long count = mine * 256;
      if ((chunks - 1) % c->nseg == sg) count -= chunks * 256 - paths;           // the last chunk may be partial
      init[size_t(ns - 1) * HK_CROW + sg * HK_CSTRIDE] = uint32_t(count);
    }
  }
  const size_t N = size_t(c->nseg) * size_t(cap);
  int rc;
  if ((rc = dev_upload(c, c->ownedPixels, order.data(), order.size() * 4)) != 0) return rc;
  if ((rc = dev_upload(c, c->liveInit, init.data(), init.size() * 4)) != 0) return rc;
  if ((rc = dev_alloc(c, c->gens, std::max<size_t>(1, size_t(c->N)) * K * 8)) != 0) return rc;      // per owned pixel: state scales with 1 / world
  if ((rc = dev_alloc(c, c->contrib, std::max<size_t>(1, size_t(c->N)) * K * 16)) != 0) return rc;
  if (!c->externalAccum) {
    const bool fresh = (c->accumInternal.bytes < npix * 16);
    if ((rc = dev_alloc(c, c->accumInternal, npix * 16)) != 0) return rc;
    if (fresh) HCHECK(hipMemsetAsync(c->accumInternal.p, 0, npix * 16, c->stream));
    c->accum = static_cast<float4*>(c->accumInternal.p);
  }
  // path state: only the arrays of the form in use are held (fused: 228 B per slot, split: 264 B)
  DevBuf* common[] = {&c->sPos, &c->sDir, &c->sThr, &c->sAcc, &c->mShadowOrg, &c->hits};
  DevBuf* fusedOnly[] = {&c->tPos, &c->tDir, &c->tThr, &c->tAcc, &c->sPend, &c->tPend, &c->shDir};
  DevBuf* splitOnly[] = {&c->mDir, &c->mThr, &c->mAcc, &c->mSurfA, &c->mSurfB, &c->mRecC, &c->mRecD, &c->mRecE};
  for (DevBuf* b : common) if ((rc = dev_alloc(c, *b, N * 16)) != 0) return rc;
  for (DevBuf* b : fusedOnly) { if (c->fusedBounce) { if ((rc = dev_alloc(c, *b, N * 16)) != 0) return rc; } else dev_free(*b); }
  for (DevBuf* b : splitOnly) { if (!c->fusedBounce) { if ((rc = dev_alloc(c, *b, N * 16)) != 0) return rc; } else dev_free(*b); }
  if ((rc = dev_alloc(c, c->sRng, N * 8)) != 0) return rc;
  if (c->fusedBounce) { dev_free(c->mRng); if ((rc = dev_alloc(c, c->tRng, N * 8)) != 0) return rc; }
  else { dev_free(c->tRng); if ((rc = dev_alloc(c, c->mRng, N * 8)) != 0) return rc; }
  if ((rc = dev_alloc(c, c->mVis, N * 4)) != 0) return rc;
  if ((rc = dev_alloc(c, c->live, size_t(HK_MAX_DEPTH + 2) * HK_CROW * 4)) != 0) return rc;
  if ((rc = dev_alloc(c, c->shadowCnt, size_t(HK_MAX_DEPTH + 2) * HK_CROW * 4)) != 0) return rc;
  if ((rc = dev_alloc(c, c->fetchCnt, size_t(2 * HK_MAX_DEPTH + 4) * HK_CROW * 4)) != 0) return rc;
  if (c->totals.p == nullptr) {
    if ((rc = dev_alloc(c, c->totals, 4 * 8)) != 0) return rc;
    HCHECK(hipMemsetAsync(c->totals.p, 0, 32, c->stream));
  }
  c->stateAllocated = true;
  return HYDRA_HIP_OK;
}

// ---- traversal launchers: one place decides between the one-ray-per-lane kernels and the persistent dynamic-fetch form
static int ensure_fetch_counters(hydra_hip_ctx* c) { return dev_alloc(c, c->fetchCnt, size_t(2 * HK_MAX_DEPTH + 4) * HK_CROW * 4); }

// `fetchCounters` is one zeroed counter row (HK_CROW words) for the persistent form, or nullptr to force the static form.
// Tree 0 goes through the persistent kernel (or the static one); trees 1..3, if the scene has them, follow with the static kernel,
// each starting from the hit the trees before it left behind (IntegratorCommon::rayTrace, Common.cpp:128-150).
static SceneDev make_scene_tree(const hydra_hip_ctx* c, int tree);
static void launch_closest(hydra_hip_ctx* c, const SceneDev& s, const SegQ& q, const float4* pos4, const float4* dir4,
                           HydraLiteHit* hits, uint32_t* perRay3, unsigned long long* totals5, uint32_t* fetchCounters) {
  TraceLaunch a;
  a.stream = c->stream; a.s = s; a.q = q; a.a4 = pos4; a.b4 = dir4; a.hits = hits; a.vis = nullptr;
  a.perRay3 = perRay3; a.totals5 = totals5; a.fetchCounters = fetchCounters; a.carry = 0; a.minActive = c->traceMinActive; a.raysPerLane = c->traceRaysPerLane;
  a.vote = c->traceVote; a.wq = c->traceVoteW[0]; a.wt = c->traceVoteW[1]; a.wi = c->traceVoteW[2];
  const bool count = (perRay3 != nullptr || totals5 != nullptr);
  if (c->traceMode == 0 || perRay3 != nullptr || fetchCounters == nullptr) {
    a.grid = seg_grid(c, q, HK_TRACE_BLOCK, c->staticBlocksPerCU);
    hk_launch_trace_static(count, s.alpha != nullptr, a);
  } else {
    a.grid = seg_grid(c, q, HK_TRACE_BLOCK, c->traceBlocksPerCU);
    hk_launch_trace_dyn(false, totals5 != nullptr, s.topTriCount > 0, s.alpha != nullptr, a);
  }
  for (int tree = 1; tree < c->treesNum; tree++) {
    if (!c->bvhNodes[tree].p || !c->bvhTris[tree].p) continue;
    a.s = make_scene_tree(c, tree);
    a.grid = seg_grid(c, q, HK_TRACE_BLOCK, c->staticBlocksPerCU);
    a.carry = 1;
    hk_launch_trace_static(count, a.s.alpha != nullptr, a);
  }
}
static void launch_shadow(hydra_hip_ctx* c, const SceneDev& s, const SegQ& q, const float4* org4, const float4* dir4,
                          float* vis, unsigned long long* totals5, uint32_t* fetchCounters) {
  TraceLaunch a;
  a.stream = c->stream; a.s = s; a.q = q; a.a4 = org4; a.b4 = dir4; a.hits = nullptr; a.vis = vis;
  a.perRay3 = nullptr; a.totals5 = totals5; a.fetchCounters = fetchCounters; a.carry = 0; a.minActive = c->traceMinActive; a.raysPerLane = c->traceRaysPerLane;
  a.vote = c->traceVote; a.wq = c->traceVoteW[0]; a.wt = c->traceVoteW[1]; a.wi = c->traceVoteW[2]; a.unordered = c->shadowUnordered;
  if (c->traceMode == 0 || fetchCounters == nullptr) {
    a.grid = seg_grid(c, q, HK_TRACE_BLOCK, c->staticBlocksPerCU);
    hk_launch_shadow_static(totals5 != nullptr, a);
    return;
  }
  a.grid = seg_grid(c, q, HK_TRACE_BLOCK, c->traceBlocksPerCU);
  hk_launch_trace_dyn(true, totals5 != nullptr, s.topTriCount > 0, false, a);
}

static hipEvent_t next_event(hydra_hip_ctx* c, size_t& cursor);

// which of the scene's small tables k_bounce copies into LDS (all or none): the material arena as uploaded, the material-id and
// texture-id tables and the lights of the globals blob
static SceneStage scene_stage(const hydra_hip_ctx* c) {
  SceneStage st = {0, 0, 0, 0, 0, 0, 0, 0, 0, nullptr};
  if (!c->sceneTablesInLds || c->hostHeader.size() <= size_t(HG_LIGHTS_NUM)) return st;
  size_t total = 0;
  // the geometry-id -> first triangle record table and the per-instance light id / inverse matrix arrays: what a path needs
  // between its hit record and its triangle record (option "scene_tables_in_lds" 2, the default, adds them when they fit 16 KB)
  const int geomTab = c->hostHeader[HG_GEOM_TABLE_SIZE];
  if (c->sceneTablesInLds >= 2 && geomTab > 0 && c->triBase.p && c->instNum > 0 && c->instMat.p && c->instLight.p) {
    const size_t need = size_t((geomTab + 1 + 3) / 4 + (c->instNum + 3) / 4 + c->instNum * 4) * 16;
    // the buffers are read in whole float4s: the int arrays must be allocated up to the next multiple of 16 bytes
    if (need <= 16 * 1024 && c->triBase.bytes >= size_t((geomTab + 1 + 3) / 4) * 16 && c->instLight.bytes >= size_t((c->instNum + 3) / 4) * 16) {
      st.triBaseF4 = (geomTab + 1 + 3) / 4; st.instLightF4 = (c->instNum + 3) / 4; st.instMatF4 = c->instNum * 4;
      total += need;
    }
  }
  const size_t matBytes = c->storageBytes[HYDRA_STORAGE_MATERIALS];
  const int matTab = c->hostHeader[HG_MAT_TABLE_SIZE], texTab = c->hostHeader[HG_TEX_TABLE_SIZE], lights = c->hostHeader[HG_LIGHTS_NUM];
  if (matBytes > 0 && (matBytes % 16) == 0 && matTab > 0 && texTab >= 0 && lights >= 0) {
    const int lsel = c->hostHeader[HG_LSEL_REV_SIZE];
    // the tables are read in whole float4s: they end inside the globals blob (the lights follow them), so rounding their length up stays in it
    const size_t need = matBytes + size_t((matTab + 3) / 4 + (texTab + 3) / 4 + HK_HDR_WORDS / 4 + (lsel + 3) / 4) * 16 + size_t(lights) * HL_FLOATS * 4;
    const bool aligned = (c->hostHeader[HG_MAT_TABLE_OFFS] % 4) == 0 && (c->hostHeader[HG_TEX_TABLE_OFFS] % 4) == 0 && (c->hostHeader[HG_LIGHTS_OFFS] % 4) == 0 && (lsel == 0 || (c->hostHeader[HG_LSEL_REV_OFFS] % 4) == 0);
    if (total + need <= HK_SCENE_LDS_MAX_BYTES && aligned && lsel >= 0) {
      st.matF4 = int(matBytes / 16); st.matTabF4 = (matTab + 3) / 4; st.lightsF4 = lights * (HL_FLOATS / 4); st.texTabF4 = (texTab + 3) / 4;
      st.hdrF4 = HK_HDR_WORDS / 4; st.lselF4 = (lsel + 3) / 4;
    }
  }
  return st;
}

// everything one sub-pass reads and writes besides the scene
struct BounceBufs {
  PathState A, B;      // A = current path state; B = the other S set (fused form only)
  MidState M;          // split form only
  ShadowQ sh;
  HydraLiteHit* hits;
};

// the back-plate the header names (HRT_SHADOW_MATTE_BACK) must be a texture of the arena: every kernel that shades a ray leaving the scene fetches it
static int check_back_plate(hydra_hip_ctx* c, bool& have) {
  have = false;
  if (c->hostHeader.size() <= size_t(HG_VARS_I + HV_I_SHADOW_MATTE_BACK_MODE)) return HYDRA_HIP_OK;
  const int32_t backId = c->hostHeader[HG_VARS_I + HV_I_SHADOW_MATTE_BACK];
  if (uint32_t(backId) == HYDRA_INVALID_TEXTURE) return HYDRA_HIP_OK;
  if (backId <= 0 || size_t(backId) >= c->hostTexTable.size() || c->hostTexTable[size_t(backId)] < 0)
    return fail(c, HYDRA_HIP_EINVAL, "HRT_SHADOW_MATTE_BACK names texture " + std::to_string(backId) + ", which is not in the texture arena");
  have = true;
  return HYDRA_HIP_OK;
}
// the per-bounce kernel sequence of one sub-pass.  fused: trace -> k_bounce (+compaction) -> shadow;
// split: trace -> k_hit (+compaction) -> shadow -> k_shade.
// counters: live / shadowCnt / fetch are arrays of counter rows (HK_CROW words), row = bounce (fetch: 2*bounce + shadow)
static int run_bounces(hydra_hip_ctx* c, const SceneDev& s, int nseg, int segCap, int maxDepth, BounceBufs bb, uint32_t* live, uint32_t* shadowCnt,
                       float4* contrib, uint2* gens, uint32_t* fetch, bool timing, const ScreenOfPath& screen) {
  const int gWide = seg_grid(c, seg_q(live, 0, nseg, segCap), 256, c->shadeBlocksPerCU);
  const int gBounce = seg_grid(c, seg_q(live, 0, nseg, segCap), HK_BOUNCE_BLOCK, c->shadeBlocksPerCU * 256 / HK_BOUNCE_BLOCK);
  const bool fused = c->fusedBounce != 0;
  if (!fused && (c->sceneFeatures & (HK_FEAT_TRANSLUCENT | HK_FEAT_BLINN | HK_FEAT_ANISO))) return fail(c, HYDRA_HIP_ESTATE, "trace_pass: the scene has translucent, Blinn, Beckmann or TRGGX materials, which only the fused bounce kernel contains (fused_bounce = 1)");
  if (!fused && (c->sceneFeatures & HK_FEAT_NMAP)) return fail(c, HYDRA_HIP_ESTATE, "trace_pass: the scene has normal-mapped materials; the split bounce form (fused_bounce = 0) carries no tangent frame in its record, use the fused kernel");
  SceneStage stage = scene_stage(c);
  if (stage.matF4 + stage.triBaseF4 > 0) {   // gather the staged tables (the header changes with the camera, so once per pass)
    if (!c->stageImg.p) { const int rc = dev_alloc(c, c->stageImg, HK_SCENE_LDS_MAX_BYTES); if (rc != 0) return rc; }
    stage.img = static_cast<const float4*>(c->stageImg.p);
    const int nF4 = stage.matF4 + stage.matTabF4 + stage.lightsF4 + stage.texTabF4 + stage.hdrF4 + stage.lselF4 + stage.triBaseF4 + stage.instLightF4 + stage.instMatF4;
    hipLaunchKernelGGL(k_build_stage_image, dim3((nF4 + 255) / 256), dim3(256), 0, c->stream, s, stage, static_cast<float4*>(c->stageImg.p));
  }
  // the back-plate (hk_shading.h, environmentColorExtended): named by the header's variables, which arrive with every PrepareEngineGlobals -- checked per pass
  bool backPlate = false;
  { const int brc = check_back_plate(c, backPlate); if (brc) return brc; }
  if (backPlate && (!fused || c->shadeWaves != 3)) return fail(c, HYDRA_HIP_ESTATE, "trace_pass: the back-plate exists in the fused bounce kernel at its default register budget only (fused_bounce = 1, shade_waves = 3)");
  // procedural textures: k_proctex (the scene's program) runs between the traversal and the bounce kernel and leaves every path its list
  SceneDev sPtl = s;
  if (c->ptlMax > 0) {
    if (c->procTex == nullptr) return fail(c, HYDRA_HIP_ESTATE, "trace_pass: materials of the scene bind procedural textures, but no program was compiled for them (hydra_hip_proctex_compile / IHWLayer::RecompileProcTexShaders)");
    if (!fused || c->shadeWaves != 3) return fail(c, HYDRA_HIP_ESTATE, "trace_pass: procedural textures exist in the fused bounce kernel at its default register budget only (fused_bounce = 1, shade_waves = 3)");
    const size_t slots = size_t(nseg) * size_t(segCap);
    int rc = dev_alloc(c, c->ptlIds, slots * size_t(c->ptlMax) * 4); if (rc != 0) return rc;
    rc = dev_alloc(c, c->ptlVals, slots * size_t(c->ptlMax) * 8); if (rc != 0) return rc;
    sPtl.ptlIds = static_cast<const int*>(c->ptlIds.p); sPtl.ptlVals = static_cast<const uint2*>(c->ptlVals.p); sPtl.ptlStride = int(slots); sPtl.ptlMax = c->ptlMax;
  }
  const bool canSort = (c->sortPathsWanted != 0) && (HK_BOUNCE_BLOCK / 64) * HK_SORT_BINS <= 64;
  const size_t stageBytes = size_t(stage.matF4 + stage.matTabF4 + stage.lightsF4 + stage.texTabF4 + stage.hdrF4 + stage.lselF4 + stage.triBaseF4 + stage.instLightF4 + stage.instMatF4) * 16;
  const int stg = (stage.matF4 > 0 ? 1 : 0) | (stage.triBaseF4 > 0 ? 2 : 0);
  HydraLiteHit* hits = bb.hits;
  auto mark = [&]() -> int { if (!timing) return -1; hipEvent_t e = next_event(c, c->evCursor); (void)hipEventRecord(e, c->stream); return int(c->evCursor) - 1; };
  for (int depth = 0; depth < maxDepth; depth++) {
    int a = mark();
    unsigned long long* tt = c->travCounters ? static_cast<unsigned long long*>(c->travTotals.p) + size_t(depth) * HK_TT_ROW : nullptr;
    const SegQ qIn = seg_q(live + size_t(depth) * HK_CROW, 0, nseg, segCap), qOut = seg_q(live + size_t(depth + 1) * HK_CROW, 0, nseg, segCap);
    uint32_t* nextCnt = live + size_t(depth + 1) * HK_CROW, *shCnt = shadowCnt + size_t(depth) * HK_CROW;
    const PathState S = bb.A;
    launch_closest(c, s, qIn, S.pos4, S.dir4, hits, nullptr, tt, fetch ? fetch + size_t(2 * depth) * HK_CROW : nullptr);
    int b = mark();
    if (c->ptlMax > 0)   // timed with the bounce kernel: it is shading work
      HCHECK(hk_proctex_launch(c->procTex, seg_grid(c, qIn, 256, c->shadeBlocksPerCU), c->stream, s, qIn, S.pos4, S.dir4, hits,
                               static_cast<int*>(c->ptlIds.p), static_cast<uint2*>(c->ptlVals.p), sPtl.ptlStride, c->ptlMax));
    if (fused) {
      const int sortPaths = (canSort && depth >= c->sortPathsFromDepth) ? 1 : 0;
      // the leanest instantiation that contains everything the scene uses (register need without spills: 167-168 VGPRs for
      // the lean sets, 16 spilled for the full one); the 4- and 5-wave register budgets (option shade_waves) exist as experiment
      // switches only: every feature, and everything staged or nothing
      const int f = c->sceneFeatures | (backPlate ? HK_FEAT_RARE_LIGHTS : 0);
      int W = 3, F = HK_FEAT_CLASSIC, G = stg;
      if (c->shadeWaves != 3) { W = (c->shadeWaves == 5) ? 5 : 4; F = HK_FEAT_ALL; G = (stg == 3) ? 3 : 0; }
      else if (f == 0) F = 0;
      else if ((f & ~HK_FEAT_SKY) == 0) F = HK_FEAT_SKY;
      else if ((f & ~(HK_FEAT_SKY | HK_FEAT_DELTA_LIGHTS | HK_FEAT_OREN_NAYAR)) == 0) F = HK_FEAT_SKY | HK_FEAT_DELTA_LIGHTS | HK_FEAT_OREN_NAYAR;
      else if ((f & ~(HK_FEAT_CLASSIC | HK_FEAT_NMAP)) == 0 && (f & HK_FEAT_NMAP)) F = HK_FEAT_CLASSIC | HK_FEAT_NMAP;   // normal maps over the classic set: without the rarer lobes
      else if (f & (HK_FEAT_NMAP | HK_FEAT_TRANSLUCENT | HK_FEAT_BLINN | HK_FEAT_ANISO | HK_FEAT_PEREZ | HK_FEAT_RARE_LIGHTS)) F = HK_FEAT_ALL;
      else if (!(f & HK_FEAT_GLASS)) F = HK_FEAT_CLASSIC & ~HK_FEAT_GLASS;
      else if (!(f & HK_FEAT_GGX)) F = HK_FEAT_CLASSIC & ~HK_FEAT_GGX;
      if (c->ptlMax > 0) F = HK_FEAT_ALL | HK_FEAT_PROCTEX;
      BounceLaunch bl;
      bl.grid = gBounce; bl.ldsBytes = (G == 0) ? 0 : stageBytes; bl.stream = c->stream; bl.s = sPtl; bl.stage = stage; bl.qIn = qIn; bl.nextCnt = nextCnt; bl.shCnt = shCnt;
      bl.depth = depth; bl.maxDepth = maxDepth; bl.A = bb.A; bl.B = bb.B; bl.hits = hits; bl.sh = bb.sh; bl.contrib = contrib; bl.gens = gens; bl.sortPaths = sortPaths; bl.screen = screen;
      if (!(hk_launch_bounce_lean(W, F, G, bl) || hk_launch_bounce_classic(W, F, G, bl) || hk_launch_bounce_nmap(W, F, G, bl) || hk_launch_bounce_all(W, F, G, bl) || hk_launch_bounce_all45(W, F, G, bl) || hk_launch_bounce_proctex(W, F, G, bl)))
        return fail(c, HYDRA_HIP_ESTATE, "trace_pass: no k_bounce instantiation for register budget " + std::to_string(W) + ", features " + std::to_string(F) + ", staging " + std::to_string(G));
      std::swap(bb.A, bb.B);
    } else {
      SplitLaunch sl;
      sl.grid = gWide; sl.stream = c->stream; sl.s = s; sl.q = qIn; sl.nextCnt = nextCnt; sl.shCnt = shCnt; sl.depth = depth; sl.maxDepth = maxDepth; sl.S = S;
      sl.hits = hits; sl.M = bb.M; sl.contrib = contrib; sl.gens = gens;
      hk_launch_hit(c->shadeWaves, sl);
    }
    int d = mark();
    if (depth + 1 < maxDepth) {
      if (fused) launch_shadow(c, s, qOut, bb.sh.org4, bb.sh.dir4, bb.sh.vis, tt ? tt + HK_TT_ROW / 2 : nullptr, fetch ? fetch + size_t(2 * depth + 1) * HK_CROW : nullptr);
      else launch_shadow(c, s, qOut, bb.M.shadowOrg, bb.M.recC, bb.M.vis, tt ? tt + HK_TT_ROW / 2 : nullptr, fetch ? fetch + size_t(2 * depth + 1) * HK_CROW : nullptr);
      int e = mark();
      if (timing) c->spans.push_back({d, e, 3, depth});
      if (!fused) {
        SplitLaunch sl;
        sl.grid = gWide; sl.stream = c->stream; sl.s = s; sl.q = qOut; sl.nextCnt = nullptr; sl.shCnt = nullptr; sl.depth = depth; sl.maxDepth = maxDepth; sl.S = S;
        sl.hits = hits; sl.M = bb.M; sl.contrib = contrib; sl.gens = gens;
        hk_launch_shade(c->shadeWaves, sl);
        int f = mark();
        if (timing) c->spans.push_back({e, f, 4, depth});
      }
    }
    if (timing) { c->spans.push_back({a, b, 1, depth}); c->spans.push_back({b, d, 2, depth}); }
  }
  HCHECK(hipGetLastError());
  return HYDRA_HIP_OK;
}

// rows [y0, y1) of a frame on up to 16 host threads (readout only; nothing here is on the hot path)
template <class F>
static void parallel_rows(int height, F body) {
  const int nt = std::max(1, std::min(16, std::min(height / 64, int(std::thread::hardware_concurrency()))));
  if (nt <= 1) { body(0, height); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++) th.emplace_back([=]() { body(int((long long)height * t / nt), int((long long)height * (t + 1) / nt)); });
  for (auto& t : th) t.join();
}
extern "C" {

int hydra_hip_create(int width, int height, int flags, int device_id, hydra_hip_handle* out) {
  (void)flags;
  if (out == nullptr || width <= 0 || height <= 0) return fail(nullptr, HYDRA_HIP_EINVAL, "hydra_hip_create: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, HYDRA_HIP_ENODEV, "hydra_hip_create: no HIP device available (the HIP layer has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, HYDRA_HIP_ENODEV, "hydra_hip_create: bad device id");
  if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, HYDRA_HIP_ENODEV, "hydra_hip_create: hipSetDevice failed");
  hydra_hip_ctx* c = new hydra_hip_ctx();
  c->device = device_id;
  c->w = width; c->h = height;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
    snprintf(c->devName, sizeof(c->devName), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    c->numCU = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  if (const char* e = getenv("HYDRA_HIP_TRACE_MODE")) c->traceMode = atoi(e) ? 1 : 0;
  if (const char* e = getenv("HYDRA_HIP_LEAF_COUNT_LINKS")) c->leafEncWanted = atoi(e) ? 1 : 0;
  if (const char* e = getenv("HYDRA_HIP_TOP_QUADS")) c->topWanted = std::max(0, std::min(atoi(e), HK_TOP_QUADS));
  if (const char* e = getenv("HYDRA_HIP_TOP_TRIS")) c->topTrisWanted = std::max(0, std::min(atoi(e), HK_TOP_TRIS));
  if (const char* e = getenv("HYDRA_HIP_TRACE_MIN_ACTIVE")) c->traceMinActive = std::max(0, std::min(64, atoi(e)));
  if (const char* e = getenv("HYDRA_HIP_TRACE_BLOCKS_PER_CU")) c->traceBlocksPerCU = std::max(1, std::min(64, atoi(e)));
  c->stream = nullptr;   // the null stream: ordered with torch's default stream and with plain hipMemcpy
  if (const char* e = getenv("HYDRA_HIP_PRIVATE_STREAM")) {   // experiment only (tools/overlap_bench.py): two layers sharing one GPU on their own streams
    if (atoi(e) != 0 && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) c->stream = nullptr;
  }
  if (dev_alloc(c, c->srgbLut, 256 * 4) != HYDRA_HIP_OK) { g_createError = c->err; delete c; return HYDRA_HIP_ENOMEM; }
  hipLaunchKernelGGL(k_fill_srgb_lut, dim3(1), dim3(256), 0, c->stream, static_cast<float*>(c->srgbLut.p));
  if (hipStreamSynchronize(c->stream) != hipSuccess) { g_createError = "hydra_hip_create: filling the sRGB table failed"; dev_free(c->srgbLut); delete c; return HYDRA_HIP_EDEVICE; }
  *out = c;
  return HYDRA_HIP_OK;
}

int hydra_hip_destroy(hydra_hip_handle c) {
  if (!c) return HYDRA_HIP_EINVAL;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  (void)hydra_hip_comm_destroy(c);
  (void)hydra_hip_mmlt_end(c);
  DevBuf* all[] = {&c->srgbLut, &c->stageImg, &c->globals, &c->instMat, &c->instLight, &c->triRec, &c->triTan, &c->triBase, &c->remapLists, &c->remapTable, &c->remapInst, &c->ownedPixels, &c->liveInit, &c->gens, &c->accumInternal,
                   &c->contrib, &c->hits, &c->live, &c->shadowCnt, &c->totals, &c->sPos, &c->sDir, &c->sThr, &c->sAcc, &c->sRng, &c->tPos, &c->tDir, &c->tThr, &c->tAcc, &c->tRng, &c->sPend, &c->tPend, &c->shDir, &c->mDir, &c->mThr, &c->mAcc,
                   &c->mRng, &c->travTotals, &c->fetchCnt, &c->mSurfA, &c->mSurfB, &c->mRecC, &c->mRecD, &c->mRecE, &c->mShadowOrg, &c->mVis};
  for (DevBuf* b : all) dev_free(*b);
  for (auto& b : c->storage) dev_free(b);
  for (auto& b : c->bvhNodes) dev_free(b);
  dev_free(c->bvhNodesTop); dev_free(c->topQuads); for (auto& b : c->leafHeaders) dev_free(b); dev_free(c->topTriF4);
  for (auto& b : c->bvhTris) dev_free(b);
  for (auto& b : c->bvhAlpha) dev_free(b);
  dev_free(c->ptlIds); dev_free(c->ptlVals); dev_free(c->stagePtlIds); dev_free(c->stagePtlVals);
  hk_proctex_free(c->procTex); c->procTex = nullptr;
  for (hipEvent_t e : c->evPool) (void)hipEventDestroy(e);
  delete c;
  return HYDRA_HIP_OK;
}

int hydra_hip_comm_destroy(hydra_hip_handle c);
int hydra_hip_mmlt_end(hydra_hip_handle c);
const char* hydra_hip_last_error(hydra_hip_handle c) { return c ? c->err.c_str() : g_createError.c_str(); }

int hydra_hip_device_name(hydra_hip_handle c, char* buf, int n) {
  if (!c || !buf || n <= 0) return HYDRA_HIP_EINVAL;
  strncpy(buf, c->devName, size_t(n - 1));
  buf[n - 1] = 0;
  return HYDRA_HIP_OK;
}

int hydra_hip_resize(hydra_hip_handle c, int width, int height) {
  if (!c || width <= 0 || height <= 0) return HYDRA_HIP_EINVAL;
  if (width == c->w && height == c->h && c->stateAllocated) return HYDRA_HIP_OK;
  HCHECK(hipSetDevice(c->device));
  HCHECK(hipDeviceSynchronize());
  (void)hydra_hip_mmlt_end(c);   // its images and splat stride belong to the old frame
  c->commCount.clear();          // so do the root's pixel lists of the exchange
  c->w = width; c->h = height;
  dev_free(c->accumInternal);
  c->stateAllocated = false; c->gensReady = false;
  c->spp = 0.0f;
  return HYDRA_HIP_OK;
}

int hydra_hip_available_memory(hydra_hip_handle c, size_t* free_bytes, size_t* total_bytes) {
  if (!c) return HYDRA_HIP_EINVAL;
  size_t f = 0, t = 0;
  HCHECK(hipSetDevice(c->device));
  HCHECK(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return HYDRA_HIP_OK;
}

int hydra_hip_finish(hydra_hip_handle c) {
  if (!c) return HYDRA_HIP_EINVAL;
  HCHECK(hipSetDevice(c->device));
  HCHECK(hipStreamSynchronize(c->stream));
  return HYDRA_HIP_OK;
}

int hydra_hip_upload_globals(hydra_hip_handle c, const int32_t* blob, size_t words) {
  if (!c || !blob || words < HG_HEADER_WORDS) return fail(c, HYDRA_HIP_EINVAL, "upload_globals: blob shorter than the EngineGlobals header");
  HCHECK(hipSetDevice(c->device));
  c->globalsWords = words;
  c->hostHeader.assign(blob, blob + HG_TABLES_READY + 1);
  c->geomDirty = true;
  c->classDirty = true;
  {
    const int64_t to = blob[HG_MAT_TABLE_OFFS], ts = blob[HG_MAT_TABLE_SIZE];
    if (to < 0 || ts < 0 || size_t(to + ts) > words) return fail(c, HYDRA_HIP_EINVAL, "upload_globals: material table runs past the blob");
    c->hostMatTable.assign(blob + to, blob + to + ts);
    c->matDirty = true;
  }
  {
    const int64_t to = blob[HG_TEXAUX_TABLE_OFFS], ts = blob[HG_TEXAUX_TABLE_SIZE];
    if (to < 0 || ts < 0 || size_t(to + ts) > words) return fail(c, HYDRA_HIP_EINVAL, "upload_globals: aux texture table runs past the blob");
    c->hostTexAuxTable.assign(blob + to, blob + to + ts);
  }
  {
    const int64_t to = blob[HG_TEX_TABLE_OFFS], ts = blob[HG_TEX_TABLE_SIZE];
    if (to < 0 || ts < 0 || size_t(to + ts) > words) return fail(c, HYDRA_HIP_EINVAL, "upload_globals: texture table runs past the blob");
    c->hostTexTable.assign(blob + to, blob + to + ts);
  }
  // the sky light (if any) must be well-formed: constant colour, lat-long texture or the Perez model
  const int skyId = blob[HG_SKY_LIGHT_ID], lightsNum = blob[HG_LIGHTS_NUM];
  if (skyId != -1 && lightsNum > 0) {   // (before SetAllPODLights the header is still zero-initialised: nothing to check yet)
    const size_t at = size_t(blob[HG_LIGHTS_OFFS]) + size_t(skyId) * HL_FLOATS;
    if (skyId < 0 || skyId >= lightsNum || at + HL_FLOATS > words) return fail(c, HYDRA_HIP_EINVAL, "upload_globals: skyLightId points outside the lights table");
    if (blob[at + HL_TYPE] != HLT_SKY_DOME) return fail(c, HYDRA_HIP_EINVAL, "upload_globals: skyLightId does not name a sky light");
  }
  int lightFeat = (skyId != -1 && lightsNum > 0) ? HK_FEAT_SKY : 0;
  if (lightsNum < 0 || (lightsNum > 0 && (blob[HG_LIGHTS_OFFS] < 0 || size_t(blob[HG_LIGHTS_OFFS]) + size_t(lightsNum) * HL_FLOATS > words)))
    return fail(c, HYDRA_HIP_EINVAL, "upload_globals: lights table runs past the blob");   // checked as a whole first: a portal's record offset names another record of it
  for (int i = 0; i < blob[HG_LIGHTS_NUM]; i++) {   // LightSampleRev knows area (rect/disk/spot cone), sphere, sky-dome, point, spot and directional lights
    const size_t at = size_t(blob[HG_LIGHTS_OFFS]) + size_t(i) * HL_FLOATS;
    if (at + HL_FLOATS > words) return fail(c, HYDRA_HIP_EINVAL, "upload_globals: lights table runs past the blob");
    const int type = blob[at + HL_TYPE];
    const bool known = (type == HLT_AREA || type == HLT_SKY_DOME || type == HLT_POINT_OMNI || type == HLT_POINT_SPOT || type == HLT_DIRECT || type == HLT_SPHERE || type == HLT_MESH || type == HLT_CYLINDER);
    if (!known)
      return fail(c, HYDRA_HIP_EINVAL, "upload_globals: light " + std::to_string(i) + " has type " + std::to_string(type) + "; the HIP layer implements area, sphere, cylinder, mesh, sky-dome, point, spot and directional lights only");
    if (type == HLT_AREA && uint32_t(blob[at + HL_COLOR_TEX]) != HYDRA_INVALID_TEXTURE)   // the reference's converter never makes one either (PlainLightConverter.cpp:206-207)
      return fail(c, HYDRA_HIP_EINVAL, "upload_globals: light " + std::to_string(i) + " is a textured area light, which the HIP layer does not implement");
    if (type == HLT_AREA && (blob[at + HL_FLAGS] & HLF_SKY_PORTAL)) {   // a portal names its sky by a record offset (RenderDriverRTE.cpp:1670-1682)
      const int64_t sky = int64_t(i) + blob[at + HL_AREA_SKY_OFFSET];
      if (sky < 0 || sky >= blob[HG_LIGHTS_NUM] || blob[size_t(blob[HG_LIGHTS_OFFS]) + size_t(sky) * HL_FLOATS + HL_TYPE] != HLT_SKY_DOME)
        return fail(c, HYDRA_HIP_EINVAL, "upload_globals: sky portal " + std::to_string(i) + " does not name a sky light of the table");
    }
    if (type == HLT_CYLINDER || (type == HLT_AREA && (blob[at + HL_FLAGS] & HLF_SKY_PORTAL)) || (type == HLT_MESH && uint32_t(blob[at + HL_MESH_TEXMATRIX_ID]) != HYDRA_INVALID_TEXTURE))
      lightFeat |= HK_FEAT_RARE_LIGHTS;   // only the all-features instantiations carry them
    if (type == HLT_CYLINDER || type == HLT_MESH) {   // the colour sampler sits inside the record, addressed in float4 units
      const uint32_t so = uint32_t(blob[at + (type == HLT_CYLINDER ? HL_CYL_TEXMATRIX_ID : HL_MESH_TEXMATRIX_ID)]);
      if (so != HYDRA_INVALID_TEXTURE && (so * 4u + 12u > uint32_t(HL_FLOATS)))
        return fail(c, HYDRA_HIP_EINVAL, "upload_globals: light " + std::to_string(i) + " places its colour sampler outside its own record");
    }
    if (type == HLT_CYLINDER) {
      const int tab = blob[at + HL_CYL_PDF_TABLE_ID];
      if (tab < 0 || tab >= blob[HG_PDF_TABLE_SIZE])
        return fail(c, HYDRA_HIP_EINVAL, "upload_globals: cylinder light " + std::to_string(i) + " has no sampling table (RenderDriverRTE::UpdateLight always makes one)");
    }
    if (type == HLT_SKY_DOME) lightFeat |= HK_FEAT_SKY;
    if (type == HLT_SKY_DOME && (blob[at + HL_FLAGS] & HLF_SKY_USE_PEREZ)) lightFeat |= HK_FEAT_PEREZ;   // only the all-features instantiation carries the model
    if (type == HLT_POINT_OMNI || type == HLT_POINT_SPOT || type == HLT_DIRECT || type == HLT_SPHERE || type == HLT_MESH || type == HLT_CYLINDER) lightFeat |= HK_FEAT_DELTA_LIGHTS;   // the bit stands for "lights other than area and sky"
    if (blob[at + HL_FLAGS] & HLF_HAS_IES) {   // point and area lights carry a photometric web: an image and a sampling table in the pdf arena
      if (type != HLT_POINT_OMNI && type != HLT_AREA)
        return fail(c, HYDRA_HIP_EINVAL, "upload_globals: light " + std::to_string(i) + " has an IES distribution on a light type that has none in the reference");
      const int tex = blob[at + HL_IES_SPHERE_TEX_ID], tab = blob[at + HL_IES_SPHERE_PDF_ID];
      if (tex < 0 || tex >= blob[HG_PDF_TABLE_SIZE] || tab < 0 || tab >= blob[HG_PDF_TABLE_SIZE])
        return fail(c, HYDRA_HIP_EINVAL, "upload_globals: IES light " + std::to_string(i) + " names tables outside the pdf-table table");
      lightFeat |= HK_FEAT_RARE_LIGHTS;
    }
  }
  if (blob[HG_SUN_NUMBER] < 0 || blob[HG_SUN_NUMBER] > 8) return fail(c, HYDRA_HIP_EINVAL, "upload_globals: sunNumber outside 0..8 (MAX_SUN_NUM, cfetch.h:18)");
  c->lightFeatures = lightFeat;
  c->sceneFeatures = c->matFeatures | c->lightFeatures;
  c->matDirty = true;   // (re)derives sceneFeatures together with the material walk
  return dev_upload(c, c->globals, blob, words * 4);
}
int hydra_hip_update_globals_header(hydra_hip_handle c, const int32_t* blob, size_t words) {
  if (!c || !blob || words == 0 || words > c->globalsWords || c->globals.p == nullptr) return fail(c, HYDRA_HIP_ESTATE, "update_globals_header: upload_globals first");
  HCHECK(hipSetDevice(c->device));
  const size_t keep = std::min<size_t>(words, HG_TABLES_READY + 1);
  c->hostHeader.assign(blob, blob + keep);
  if (words > HG_LIGHTS_NUM && blob[HG_SKY_LIGHT_ID] != -1 && blob[HG_LIGHTS_NUM] > 0) { c->lightFeatures |= HK_FEAT_SKY; c->sceneFeatures |= HK_FEAT_SKY; }
  HCHECK(hipMemcpyAsync(c->globals.p, blob, words * 4, hipMemcpyHostToDevice, c->stream));
  HCHECK(hipStreamSynchronize(c->stream));
  return HYDRA_HIP_OK;
}
int hydra_hip_upload_storage(hydra_hip_handle c, int kind, const void* data, size_t bytes) {
  if (!c || kind < 0 || kind >= HYDRA_STORAGE_KINDS || (bytes > 0 && !data)) return fail(c, HYDRA_HIP_EINVAL, "upload_storage: bad arguments");
  HCHECK(hipSetDevice(c->device));
  if (kind == HYDRA_STORAGE_GEOM) c->geomDirty = true;
  if (kind == HYDRA_STORAGE_GEOM || kind == HYDRA_STORAGE_MATERIALS) c->classDirty = true;
  if (kind == HYDRA_STORAGE_MATERIALS) {
    c->hostMaterials.assign(static_cast<const float*>(data), static_cast<const float*>(data) + bytes / 4);
    c->matDirty = true;
  }
  c->storageBytes[kind] = bytes;
  return dev_upload(c, c->storage[kind], data, bytes);
}
// Device copy of the node array: put the triangle count of every triangle leaf into bits 27..30 of the link that points at it
// (hk_trace.h, HK_LEAF_COUNT_SHIFT).  Which leaves are triangle leaves follows from walking the tree the way BVH4InstTraverse
// does (ctrace.h:841-1062): below the root every leaf is an instance quad when the tree is instanced, and the quad an
// instance names (or a leaf link stored there directly) starts an object tree whose leaves are triangle lists.
// Returns false (and leaves `nodes` untouched) when an index does not fit next to the count.
static bool encode_leaf_counts(std::vector<HydraBVHNode>& nodes, const float* tri_f4, int tri_f4_num, bool haveInst) {
  const size_t quads = nodes.size() / 4;
  if (quads >= (size_t(1) << HK_LEAF_COUNT_SHIFT) || size_t(tri_f4_num) >= (size_t(1) << HK_LEAF_COUNT_SHIFT)) return false;
  std::vector<HydraBVHNode> out = nodes;
  auto encode = [&](uint32_t& link) {   // link has the leaf bit and names a triangle list header
    const uint32_t off = link & 0x7fffffffu;
    if (off + 1 >= uint32_t(tri_f4_num)) return;
    int32_t hdr[2];
    memcpy(hdr, tri_f4 + size_t(off) * 4, 8);
    if (hdr[0] == int32_t(off) + 1 && hdr[1] >= 1 && hdr[1] <= 15 && size_t(hdr[0]) + size_t(hdr[1]) * 3 <= size_t(tri_f4_num))
      link = HYDRA_BVH_LEAF | (uint32_t(hdr[1]) << HK_LEAF_COUNT_SHIFT) | off;
  };
  std::vector<uint8_t> seen(quads * 2, 0);   // [quad][level]
  std::vector<std::pair<uint32_t, int>> stack;
  stack.push_back({1u, haveInst ? 0 : 1});   // level 0 = scene tree of an instanced BVH, 1 = tree whose leaves are triangles
  while (!stack.empty()) {
    const uint32_t q = stack.back().first;
    const int level = stack.back().second;
    stack.pop_back();
    if (q >= quads || seen[size_t(q) * 2 + level]) continue;
    seen[size_t(q) * 2 + level] = 1;
    for (int k = 0; k < 4; k++) {
      const HydraBVHNode& n = nodes[size_t(q) * 4 + k];
      if (n.leftOffsetAndLeaf == HYDRA_BVH_INVALID && n.escapeIndex == HYDRA_BVH_INVALID) continue;
      const uint32_t off = n.leftOffsetAndLeaf & 0x7fffffffu;
      if (!(n.leftOffsetAndLeaf & HYDRA_BVH_LEAF)) { stack.push_back({off, level}); continue; }
      if (level == 1) { encode(out[size_t(q) * 4 + k].leftOffsetAndLeaf); continue; }
      if (off >= quads) continue;                                  // instance quad: word 3 of its first float4 = where the object tree starts
      const uint32_t next = nodes[size_t(off) * 4].leftOffsetAndLeaf;
      if (next & HYDRA_BVH_LEAF) {
        if (!seen[size_t(off) * 2 + 1]) { seen[size_t(off) * 2 + 1] = 1; encode(out[size_t(off) * 4].leftOffsetAndLeaf); }
      } else stack.push_back({next, 1});
    }
  }
  // a quad reached both as part of the scene tree and of an object tree would need two readings of the same link
  for (size_t q = 0; q < quads; q++) if (seen[q * 2] && seen[q * 2 + 1]) return false;
  nodes.swap(out);
  return true;
}

// float4 index of the header of every triangle leaf of the tree, found by the same walk as encode_leaf_counts (k_tag_triangle_classes)
static std::vector<int> collect_leaf_headers(const std::vector<HydraBVHNode>& nodes, const float* tri_f4, int tri_f4_num, bool haveInst) {
  std::vector<int> out;
  const size_t quads = nodes.size() / 4;
  auto leaf = [&](uint32_t link) {
    const uint32_t off = link & 0x7fffffffu;
    if (off + 1 >= uint32_t(tri_f4_num)) return;
    int32_t hdr[2];
    memcpy(hdr, tri_f4 + size_t(off) * 4, 8);
    if (hdr[0] >= 0 && hdr[1] >= 1 && size_t(hdr[0]) + size_t(hdr[1]) * 3 <= size_t(tri_f4_num)) out.push_back(int(off));
  };
  std::vector<uint8_t> seen(quads * 2, 0);
  std::vector<std::pair<uint32_t, int>> stack;
  stack.push_back({1u, haveInst ? 0 : 1});
  while (!stack.empty()) {
    const uint32_t q = stack.back().first;
    const int level = stack.back().second;
    stack.pop_back();
    if (q >= quads || seen[size_t(q) * 2 + level]) continue;
    seen[size_t(q) * 2 + level] = 1;
    for (int k = 0; k < 4; k++) {
      const HydraBVHNode& n = nodes[size_t(q) * 4 + k];
      if (n.leftOffsetAndLeaf == HYDRA_BVH_INVALID && n.escapeIndex == HYDRA_BVH_INVALID) continue;
      const uint32_t off = n.leftOffsetAndLeaf & 0x7fffffffu;
      if (!(n.leftOffsetAndLeaf & HYDRA_BVH_LEAF)) { stack.push_back({off, level}); continue; }
      if (level == 1) { leaf(n.leftOffsetAndLeaf); continue; }
      if (off >= quads) continue;
      const uint32_t next = nodes[size_t(off) * 4].leftOffsetAndLeaf;
      if (next & HYDRA_BVH_LEAF) { if (!seen[size_t(off) * 2 + 1]) { seen[size_t(off) * 2 + 1] = 1; leaf(next); } }
      else stack.push_back({next, 1});
    }
  }
  std::sort(out.begin(), out.end());
  out.erase(std::unique(out.begin(), out.end()), out.end());
  return out;
}

// Choose the triangle leaves to keep in LDS (the "triangle packets" of the persistent traversal kernels): the same best-first
// walk and priorities as for the quads below, continued for a bounded number of quads; a triangle leaf's priority is its parent
// quad's times its box-area ratio.  Leaves are taken in priority order while their triangles fit the pool.  Needs the triangle
// counts in the leaf links (encode_leaf_counts).  Returns, per chosen leaf, where its link sits (node index) and the float4
// indices of its triangles; `poolF4` receives the source index of every float4 of the pool.
struct TopLeaf { size_t node; int count, poolStart; };
static std::vector<TopLeaf> choose_top_leaves(const std::vector<HydraBVHNode>& nodes, bool haveInst, int triF4Num, int wantedTris, std::vector<int>& poolF4) {
  std::vector<TopLeaf> out;
  poolF4.clear();
  const size_t quads = nodes.size() / 4;
  if (wantedTris <= 0 || quads < 2 || triF4Num >= int(HK_LEAF_LDS_FLAG)) return out;
  auto valid = [](const HydraBVHNode& n) { return !(n.leftOffsetAndLeaf == HYDRA_BVH_INVALID && n.escapeIndex == HYDRA_BVH_INVALID); };
  auto area = [](const float* lo, const float* hi) {
    const double dx = std::max(0.0f, hi[0] - lo[0]), dy = std::max(0.0f, hi[1] - lo[1]), dz = std::max(0.0f, hi[2] - lo[2]);
    return 2.0 * (dx * dy + dy * dz + dx * dz);
  };
  struct Item { double w; uint32_t quad; int level; bool operator<(const Item& o) const { return w < o.w; } };
  struct Cand { double w; size_t node; uint32_t link; };
  std::priority_queue<Item> heap;
  std::vector<Cand> cands;
  heap.push({1.0, 1u, haveInst ? 0 : 1});
  std::vector<uint8_t> seen(quads * 2, 0);
  for (int pops = 0; !heap.empty() && pops < 4096; ) {
    const Item it = heap.top();
    heap.pop();
    if (it.quad >= quads || seen[size_t(it.quad) * 2 + it.level]) continue;
    seen[size_t(it.quad) * 2 + it.level] = 1;
    pops++;
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int k = 0; k < 4; k++) {
      const HydraBVHNode& n = nodes[size_t(it.quad) * 4 + k];
      if (!valid(n)) continue;
      for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], n.boxMin[a]); hi[a] = std::max(hi[a], n.boxMax[a]); }
    }
    const double qa = std::max(area(lo, hi), 1e-30);
    for (int k = 0; k < 4; k++) {
      const size_t at = size_t(it.quad) * 4 + k;
      const HydraBVHNode& n = nodes[at];
      if (!valid(n)) continue;
      const double w = it.w * std::min(area(n.boxMin, n.boxMax) / qa, 1.0);
      const uint32_t off = n.leftOffsetAndLeaf & 0x7fffffffu;
      if (!(n.leftOffsetAndLeaf & HYDRA_BVH_LEAF)) heap.push({w, off, it.level});
      else if (it.level == 1) cands.push_back({w, at, n.leftOffsetAndLeaf});
      else if (off < quads) {
        const uint32_t next = nodes[size_t(off) * 4].leftOffsetAndLeaf;
        if (!(next & HYDRA_BVH_LEAF)) heap.push({w, next, 1});
        else cands.push_back({w, size_t(off) * 4, next});      // an object that is one leaf: the instance quad's own link
      }
    }
  }
  std::stable_sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.w > b.w; });
  std::vector<uint8_t> taken(nodes.size(), 0);
  int used = 0;
  for (const Cand& cd : cands) {
    const int count = int((cd.link >> HK_LEAF_COUNT_SHIFT) & 15u);
    const int first = int(cd.link & HK_LEAF_OFFSET_MASK) + 1;
    if (count == 0 || used + count > wantedTris || taken[cd.node] || first + 3 * count > triF4Num) continue;
    taken[cd.node] = 1;
    out.push_back({cd.node, count, used});
    for (int i = 0; i < 3 * count; i++) poolF4.push_back(first + i);
    used += count;
    if (used == wantedTris) break;
  }
  return out;
}

// Choose the quads to keep in LDS: best-first walk from the root, a child's priority = its parent's times the ratio of the
// child's box area to the area of the quad's union box (the surface-area estimate of how often a ray that visits the parent
// goes on to the child); an instance leaf hands its priority to the root of the object tree it names.  On the reference's
// Cornell-style scene the first 21 quads in this order take 63-71 % of all quad visits of bounce and shadow rays, on the
// instanced atrium 46-53 % (profiles/r01/top_quad_visit_share.log).  Then tag every link of `nodes` that names a chosen quad.
static std::vector<int> choose_and_tag_top_quads(std::vector<HydraBVHNode>& nodes, bool haveInst, int wanted) {
  std::vector<int> top;
  const size_t quads = nodes.size() / 4;
  if (wanted <= 0 || quads < 2 || quads >= (size_t(1) << HK_LEAF_COUNT_SHIFT)) return top;
  auto valid = [](const HydraBVHNode& n) { return !(n.leftOffsetAndLeaf == HYDRA_BVH_INVALID && n.escapeIndex == HYDRA_BVH_INVALID); };
  auto area = [](const float* lo, const float* hi) {
    const double dx = std::max(0.0f, hi[0] - lo[0]), dy = std::max(0.0f, hi[1] - lo[1]), dz = std::max(0.0f, hi[2] - lo[2]);
    return 2.0 * (dx * dy + dy * dz + dx * dz);
  };
  struct Item { double w; uint32_t quad; int level; bool operator<(const Item& o) const { return w < o.w; } };
  std::priority_queue<Item> heap;
  heap.push({1.0, 1u, haveInst ? 0 : 1});
  std::vector<uint8_t> seen(quads * 2, 0);
  std::vector<int> slotOf(quads, -1);
  while (!heap.empty() && int(top.size()) < wanted) {
    const Item it = heap.top();
    heap.pop();
    if (it.quad >= quads || seen[size_t(it.quad) * 2 + it.level]) continue;
    seen[size_t(it.quad) * 2 + it.level] = 1;
    if (slotOf[it.quad] >= 0) continue;   // reached on both levels: one slot is enough
    slotOf[it.quad] = int(top.size());
    top.push_back(int(it.quad));
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int k = 0; k < 4; k++) {
      const HydraBVHNode& n = nodes[size_t(it.quad) * 4 + k];
      if (!valid(n)) continue;
      for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], n.boxMin[a]); hi[a] = std::max(hi[a], n.boxMax[a]); }
    }
    const double qa = std::max(area(lo, hi), 1e-30);
    for (int k = 0; k < 4; k++) {
      const HydraBVHNode& n = nodes[size_t(it.quad) * 4 + k];
      if (!valid(n)) continue;
      const double w = it.w * std::min(area(n.boxMin, n.boxMax) / qa, 1.0);
      const uint32_t off = n.leftOffsetAndLeaf & 0x7fffffffu;
      if (!(n.leftOffsetAndLeaf & HYDRA_BVH_LEAF)) heap.push({w, off, it.level});
      else if (it.level == 0 && off < quads) {
        const uint32_t next = nodes[size_t(off) * 4].leftOffsetAndLeaf;
        if (!(next & HYDRA_BVH_LEAF)) heap.push({w, next, 1});
      }
    }
  }
  if (top.empty() || top[0] != 1) { top.clear(); return top; }
  // tag: every inner link (bit 31 clear) naming a chosen quad, in tree quads and in the "next" word of instance quads alike.
  // Words that are not links (box floats, matrices) are never touched: only word 3 of a node is rewritten, and for nodes of
  // an instance quad other than its first that word is matrix data -- so instance quads are handled one by one below.
  std::vector<uint8_t> isTreeQuad(quads, 0), isInstQuad(quads, 0);
  {
    std::vector<std::pair<uint32_t, int>> stack;
    std::vector<uint8_t> vis(quads * 2, 0);
    stack.push_back({1u, haveInst ? 0 : 1});
    while (!stack.empty()) {
      const uint32_t q = stack.back().first;
      const int level = stack.back().second;
      stack.pop_back();
      if (q >= quads || vis[size_t(q) * 2 + level]) continue;
      vis[size_t(q) * 2 + level] = 1;
      isTreeQuad[q] = 1;
      for (int k = 0; k < 4; k++) {
        const HydraBVHNode& n = nodes[size_t(q) * 4 + k];
        if (!valid(n)) continue;
        const uint32_t off = n.leftOffsetAndLeaf & 0x7fffffffu;
        if (!(n.leftOffsetAndLeaf & HYDRA_BVH_LEAF)) stack.push_back({off, level});
        else if (level == 0 && off < quads) {
          isInstQuad[off] = 1;
          const uint32_t next = nodes[size_t(off) * 4].leftOffsetAndLeaf;
          if (!(next & HYDRA_BVH_LEAF)) stack.push_back({next, 1});
        }
      }
    }
  }
  for (size_t q = 0; q < quads; q++) if (isTreeQuad[q] && isInstQuad[q]) { top.clear(); return top; }   // ambiguous layout: no cache
  auto tag = [&](uint32_t& link) {
    if (link & HYDRA_BVH_LEAF) return;
    if (link < quads && slotOf[link] >= 0) link = uint32_t(HK_TOP_FLAG) | uint32_t(slotOf[link]);
  };
  for (size_t q = 0; q < quads; q++) {
    if (isTreeQuad[q]) { for (int k = 0; k < 4; k++) if (valid(nodes[q * 4 + k])) tag(nodes[q * 4 + k].leftOffsetAndLeaf); }
    else if (isInstQuad[q]) tag(nodes[q * 4].leftOffsetAndLeaf);
  }
  return top;
}

int hydra_hip_upload_bvh(hydra_hip_handle c, int tree, const HydraBVHNode* nodes, int nodes_num, const float* tri_f4, int tri_f4_num,
                         const uint32_t* alpha, int alpha_num, int have_inst) {
  if (!c || tree < 0 || tree >= 4 || !nodes || nodes_num < 8 || !tri_f4 || tri_f4_num <= 0) return fail(c, HYDRA_HIP_EINVAL, "upload_bvh: bad arguments");
  // alpha table: one uint2 per float4 of the triangle list, then the opacity samplers (RenderDriverRTE_AlphaTestTable.cpp:65-224)
  if (alpha != nullptr && alpha_num > 0 && alpha_num < tri_f4_num) return fail(c, HYDRA_HIP_EINVAL, "upload_bvh: alpha table shorter than the triangle list");
  if (alpha != nullptr && alpha_num > 0 && !have_inst) return fail(c, HYDRA_HIP_EINVAL, "upload_bvh: an alpha table on a tree without instances (the reference has no such traversal, Common.cpp:140-146)");
  HCHECK(hipSetDevice(c->device));
  if (alpha != nullptr && alpha_num > 0) {
    // every sampler position a triangle names must lie inside the table (the traversal reads 6 uint2 from there)
    for (int i = 0; i < tri_f4_num; i++) {
      const uint32_t x = alpha[2 * size_t(i)];
      int32_t w2, w3;
      memcpy(&w2, tri_f4 + size_t(i) * 4 + 2, 4); memcpy(&w3, tri_f4 + size_t(i) * 4 + 3, 4);
      const bool header = (w2 == -1 && w3 == -1);
      if (header) { i += 0; continue; }
      // only the first float4 of a triangle carries a sampler position; the other two carry flags (0 / 1)
      if (x != 0xFFFFFFFFu && x != HYDRA_INVALID_TEXTURE && int32_t(x) > 1 && size_t(x) + 6 > size_t(alpha_num))
        return fail(c, HYDRA_HIP_EINVAL, "upload_bvh: alpha table entry " + std::to_string(i) + " points outside the table");
    }
    const int arc = dev_upload(c, c->bvhAlpha[tree], alpha, size_t(alpha_num) * 8);
    if (arc) return arc;
  } else dev_free(c->bvhAlpha[tree]);
  // the traversal kernels address both arrays as raw buffers with 32-bit byte offsets
  if (size_t(nodes_num) * sizeof(HydraBVHNode) >= (size_t(1) << 32) || size_t(tri_f4_num) * 16 >= (size_t(1) << 32))
    return fail(c, HYDRA_HIP_EINVAL, "upload_bvh: node or triangle arrays of 4 GiB and more are not supported");
  std::vector<HydraBVHNode> devNodes(nodes, nodes + nodes_num);
  {
    const std::vector<int> headers = collect_leaf_headers(devNodes, tri_f4, tri_f4_num, have_inst != 0);
    c->leafHeadersNum[tree] = int(headers.size());
    const int hrc = dev_upload(c, c->leafHeaders[tree], headers.data(), headers.size() * sizeof(int));
    if (hrc) return hrc;
    c->classDirty = true;
  }
  c->leafEnc[tree] = (c->leafEncWanted != 0) && encode_leaf_counts(devNodes, tri_f4, tri_f4_num, have_inst != 0);
  int rc = dev_upload(c, c->bvhNodes[tree], devNodes.data(), size_t(nodes_num) * sizeof(HydraBVHNode));
  if (rc) return rc;
  rc = dev_upload(c, c->bvhTris[tree], tri_f4, size_t(tri_f4_num) * 16);
  if (rc) return rc;
  c->bvhNodeBytes[tree] = size_t(nodes_num) * sizeof(HydraBVHNode);
  hipLaunchKernelGGL(k_prepare_bvh, dim3(grid_for(c, nodes_num, 256, 8)), dim3(256), 0, c->stream, nodes_num, static_cast<float4*>(c->bvhNodes[tree].p));
  HCHECK(hipGetLastError());
  if (tree == 0) {   // the copy the persistent kernels walk (tree 0 only): same nodes, links to the quads kept in LDS tagged with their slot
    std::vector<int> poolF4;
    // no leaf goes to LDS when the tree has an alpha table: the alpha-tested kernels are instantiated without TOPTRIS (the alpha test
    // indexes the table with the triangle's address in the list, which a pool index is not), and a tagged link read without TOPTRIS
    // decodes its LDS flag as part of the list offset
    const bool treeHasAlpha = (alpha != nullptr && alpha_num > 0);
    const std::vector<TopLeaf> topLeaves = (c->leafEnc[tree] && c->topWanted > 0 && !treeHasAlpha) ? choose_top_leaves(devNodes, have_inst != 0, tri_f4_num, std::min(c->topTrisWanted, HK_TOP_TRIS), poolF4)
                                                                                   : std::vector<TopLeaf>();
    const std::vector<int> top = choose_and_tag_top_quads(devNodes, have_inst != 0, std::min(c->topWanted, HK_TOP_QUADS));
    c->topCount = int(top.size());
    c->topTriCount = 0;
    if (c->topCount > 0 && !topLeaves.empty()) {   // leaf links are not touched by the quad tagging: rewrite them in the same copy
      for (const TopLeaf& l : topLeaves)
        devNodes[l.node].leftOffsetAndLeaf = HYDRA_BVH_LEAF | (uint32_t(l.count) << HK_LEAF_COUNT_SHIFT) | uint32_t(HK_LEAF_LDS_FLAG) | uint32_t(l.poolStart);
      rc = dev_upload(c, c->topTriF4, poolF4.data(), poolF4.size() * sizeof(int));
      if (rc) return rc;
      c->topTriCount = int(poolF4.size() / 3);
    }
    if (c->topCount > 0) {
      rc = dev_upload(c, c->bvhNodesTop, devNodes.data(), size_t(nodes_num) * sizeof(HydraBVHNode));
      if (rc) return rc;
      rc = dev_upload(c, c->topQuads, top.data(), top.size() * sizeof(int));
      if (rc) return rc;
      hipLaunchKernelGGL(k_prepare_bvh, dim3(grid_for(c, nodes_num, 256, 8)), dim3(256), 0, c->stream, nodes_num, static_cast<float4*>(c->bvhNodesTop.p));
      HCHECK(hipGetLastError());
    } else { dev_free(c->bvhNodesTop); dev_free(c->topQuads); }
  }
  c->bvhTriBytes[tree] = size_t(tri_f4_num) * 16;
  c->haveInst[tree] = have_inst ? 1 : 0;
  if (c->treesNum < tree + 1) c->treesNum = tree + 1;
  return HYDRA_HIP_OK;
}
int hydra_hip_set_bvh_trees_num(hydra_hip_handle c, int n) {
  if (!c || n < 1 || n > 4) return HYDRA_HIP_EINVAL;
  for (int t = 0; t < n; t++) if (!c->bvhNodes[t].p) return fail(c, HYDRA_HIP_ESTATE, "set_bvh_trees_num: tree " + std::to_string(t) + " was not uploaded");
  c->treesNum = n;
  for (int t = n; t < 4; t++) { dev_free(c->bvhNodes[t]); dev_free(c->bvhTris[t]); dev_free(c->bvhAlpha[t]); dev_free(c->leafHeaders[t]); c->leafHeadersNum[t] = 0; }
  return HYDRA_HIP_OK;
}
int hydra_hip_upload_instances(hydra_hip_handle c, const float* inv16, const int32_t* lightInstId, int n) {
  if (!c || !inv16 || !lightInstId || n <= 0) return fail(c, HYDRA_HIP_EINVAL, "upload_instances: bad arguments");
  HCHECK(hipSetDevice(c->device));
  int rc = dev_upload(c, c->instMat, inv16, size_t(n) * 64);
  if (rc) return rc;
  rc = dev_upload(c, c->instLight, lightInstId, size_t(n) * 4);
  if (rc) return rc;
  c->instNum = n;
  return HYDRA_HIP_OK;
}
int hydra_hip_upload_remap_lists(hydra_hip_handle c, const int32_t* all_lists, int all_size, const int32_t* table_int2, int table_size,
                                 const int32_t* inst_to_remap, int inst_num) {
  if (!c || all_size < 0 || table_size < 0 || inst_num < 0) return HYDRA_HIP_EINVAL;
  HCHECK(hipSetDevice(c->device));
  int rc;
  if ((rc = dev_upload(c, c->remapLists, all_lists, size_t(all_size) * 4))) return rc;
  if ((rc = dev_upload(c, c->remapTable, table_int2, size_t(table_size) * 8))) return rc;
  if ((rc = dev_upload(c, c->remapInst, inst_to_remap, size_t(inst_num) * 4))) return rc;
  c->remapListsSize = all_size; c->remapTableSize = table_size; c->remapInstSize = inst_num;
  return HYDRA_HIP_OK;
}

int hydra_hip_set_tile_partition(hydra_hip_handle c, int rank, int world, int tile) {
  if (!c || world < 1 || rank < 0 || rank >= world || tile < 8 || (tile % 8) != 0) return fail(c, HYDRA_HIP_EINVAL, "set_tile_partition: need 0 <= rank < world and tile a multiple of 8");
  if (rank != c->rank || world != c->world || tile != c->tile) c->gensReady = false;   // generator states are kept per OWNED pixel: a new partition restarts the image
  c->rank = rank; c->world = world; c->tile = tile;
  c->stateAllocated = false;
  c->commCount.clear();   // the root's pixel lists of the exchange follow the partition
  return HYDRA_HIP_OK;
}
int hydra_hip_plan_render_state(int width, int height, int rank, int world, int tile_size, int samples_in_flight, int queue_segments, int fused_bounce,
                                HydraStatePlan* out, char* err, int err_len) {
  if (!out) return HYDRA_HIP_EINVAL;
  std::string e;
  const int rc = plan_render_state(width, height, rank, world, tile_size, samples_in_flight, queue_segments > 0 ? queue_segments : 32, fused_bounce, out, &e);
  if (rc && err && err_len > 0) { strncpy(err, e.c_str(), size_t(err_len - 1)); err[err_len - 1] = 0; }
  return rc;
}
int hydra_hip_tile_owners(int width, int height, int world, int tile_size, int32_t* owner_of_tile, int tiles) {
  if (width <= 0 || height <= 0 || world < 1 || tile_size < 8 || !owner_of_tile) return HYDRA_HIP_EINVAL;
  const int tilesX = (width + tile_size - 1) / tile_size, tilesY = (height + tile_size - 1) / tile_size;
  if (tiles != tilesX * tilesY) return HYDRA_HIP_EINVAL;
  const std::vector<int> order = morton_tile_order(tilesX, tilesY);
  for (size_t i = 0; i < order.size(); i++) owner_of_tile[order[i]] = int(i % size_t(world));
  return HYDRA_HIP_OK;
}
int hydra_hip_set_external_accumulator(hydra_hip_handle c, void* dev, size_t bytes) {
  if (!c) return HYDRA_HIP_EINVAL;
  if (dev == nullptr) { c->externalAccum = false; c->accum = static_cast<float4*>(c->accumInternal.p); c->stateAllocated = false; return HYDRA_HIP_OK; }
  if (bytes < size_t(c->w) * c->h * 16) return fail(c, HYDRA_HIP_EINVAL, "set_external_accumulator: buffer smaller than width*height float4");
  c->externalAccum = true;
  c->accum = static_cast<float4*>(dev);
  return HYDRA_HIP_OK;
}
int hydra_hip_init_path_tracing(hydra_hip_handle c, int seed) {
  if (!c) return HYDRA_HIP_EINVAL;
  HCHECK(hipSetDevice(c->device));
  if (!c->stateAllocated) { int rc = alloc_render_state(c); if (rc) return rc; }
  c->seed = seed;
  if (c->N > 0) {
    const size_t ngen = size_t(c->N) * c->streams;
    hipLaunchKernelGGL(k_init_gens, dim3(grid_for(c, int(std::min<size_t>(ngen, size_t(1) << 30)), 256, 8)), dim3(256), 0, c->stream, c->N, c->streams,
                       static_cast<const int*>(c->ownedPixels.p), unsigned(c->w) * unsigned(c->h), seed, static_cast<uint2*>(c->gens.p));
  }
  HCHECK(hipGetLastError());
  c->gensReady = true;
  return hydra_hip_clear_accumulated_color(c);
}
int hydra_hip_clear_accumulated_color(hydra_hip_handle c) {
  if (!c) return HYDRA_HIP_EINVAL;
  HCHECK(hipSetDevice(c->device));
  if (!c->stateAllocated) { int rc = alloc_render_state(c); if (rc) return rc; }
  HCHECK(hipMemsetAsync(c->accum, 0, size_t(c->w) * c->h * 16, c->stream));
  c->spp = 0.0f;
  return HYDRA_HIP_OK;
}

// resolve the recorded stage events into per-stage totals (needs the stream to be idle => one sync)
static int fold_stage_events(hydra_hip_ctx* c) {
  if (c->spans.empty()) { c->evCursor = 0; return HYDRA_HIP_OK; }
  HCHECK(hipStreamSynchronize(c->stream));
  for (const auto& sp : c->spans) {
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, c->evPool[sp.a], c->evPool[sp.b]);
    switch (sp.kind) { case 0: c->tRaygen += ms; break; case 1: c->tTrace += ms; c->nTrace++; break; case 2: c->tHit += ms; break; case 3: c->tShadow += ms; c->nShadow++; break;
                       case 4: c->tShade += ms; break; case 5: c->tAccum += ms; break; default: c->tPass += ms; break; }
    if (sp.depth >= 0 && sp.depth < HK_MAX_DEPTH && sp.kind >= 1 && sp.kind <= 4) c->tDepth[sp.depth][sp.kind == 4 ? 1 : sp.kind - 1] += ms;
  }
  c->spans.clear();
  c->evCursor = 0;
  return HYDRA_HIP_OK;
}

static hipEvent_t next_event(hydra_hip_ctx* c, size_t& cursor) {
  if (cursor >= c->evPool.size()) { hipEvent_t e; (void)hipEventCreate(&e); c->evPool.push_back(e); }
  return c->evPool[cursor++];
}

int hydra_hip_trace_pass(hydra_hip_handle c, int spp) {
  if (!c || spp < 1) return HYDRA_HIP_EINVAL;
  if (!scene_ready(c)) return fail(c, HYDRA_HIP_ESTATE, "trace_pass: scene is not completely uploaded (globals, storages, BVH, instances)");
  HCHECK(hipSetDevice(c->device));
  { int rc = prepare_geometry(c); if (rc) return rc; }
  { int rc = validate_materials(c); if (rc) return rc; }
  { int rc = prepare_classes(c); if (rc) return rc; }
  if (!c->stateAllocated) { int rc = alloc_render_state(c); if (rc) return rc; }
  if (!c->gensReady) { int rc = hydra_hip_init_path_tracing(c, c->seed); if (rc) return rc; }
  if (c->N == 0) { c->spp += float(spp); return HYDRA_HIP_OK; }
  const int maxDepth = c->hostHeader[HG_VARS_I + HV_I_TRACE_DEPTH];
  if (maxDepth < 1 || maxDepth > HK_MAX_DEPTH) return fail(c, HYDRA_HIP_EINVAL, "trace_pass: HRT_TRACE_DEPTH out of range");

  const SceneDev s = make_scene(c);
  auto f4 = [](const DevBuf& b) { return static_cast<float4*>(b.p); };
  const PathState S = {f4(c->sPos), f4(c->sDir), f4(c->sThr), f4(c->sAcc), static_cast<uint2*>(c->sRng.p), f4(c->sPend)};
  BounceBufs bb;
  bb.A = S;
  bb.B = {f4(c->tPos), f4(c->tDir), f4(c->tThr), f4(c->tAcc), static_cast<uint2*>(c->tRng.p), f4(c->tPend)};
  bb.M = {f4(c->mDir), f4(c->mThr), f4(c->mAcc), static_cast<uint2*>(c->mRng.p), f4(c->mSurfA), f4(c->mSurfB), f4(c->mRecC), f4(c->mRecD), f4(c->mRecE),
          f4(c->mShadowOrg), static_cast<float*>(c->mVis.p)};
  bb.sh = {f4(c->mShadowOrg), f4(c->shDir), static_cast<float*>(c->mVis.p)};
  bb.hits = static_cast<HydraLiteHit*>(c->hits.p);
  uint32_t* live = static_cast<uint32_t*>(c->live.p);
  uint32_t* shadowCnt = static_cast<uint32_t*>(c->shadowCnt.p);
  const SegQ q0 = seg_q(live, 0, c->nseg, c->segCap);
  const int gWide = seg_grid(c, q0, 256, 8);
  const bool timing = c->stageTiming;
  auto mark = [&]() -> int { if (!timing) return -1; hipEvent_t e = next_event(c, c->evCursor); (void)hipEventRecord(e, c->stream); return int(c->evCursor) - 1; };

  for (int done = 0; done < spp;) {
    const int ns = std::min(c->streams, spp - done);   // samples per pixel traced concurrently in this sub-pass: streams 0..ns-1
    done += ns;
    HCHECK(hipMemsetAsync(live + HK_CROW, 0, size_t(maxDepth + 1) * HK_CROW * 4, c->stream));
    HCHECK(hipMemsetAsync(shadowCnt, 0, size_t(maxDepth + 1) * HK_CROW * 4, c->stream));
    if (c->traceMode != 0) HCHECK(hipMemsetAsync(c->fetchCnt.p, 0, size_t(2 * maxDepth + 2) * HK_CROW * 4, c->stream));
    HCHECK(hipMemcpyAsync(live, static_cast<const uint32_t*>(c->liveInit.p) + size_t(ns - 1) * HK_CROW, size_t(HK_CROW) * 4, hipMemcpyDeviceToDevice, c->stream));
    int e0 = mark();
    hipLaunchKernelGGL(k_raygen, dim3(gWide), dim3(256), 0, c->stream, s, q0, static_cast<const int*>(c->ownedPixels.p), c->N, ns, c->streamMajor, static_cast<const uint2*>(c->gens.p), c->w, c->h, S);
    int e1 = mark();
    if (timing) c->spans.push_back({e0, e1, 0, -1});
    { int rc = run_bounces(c, s, c->nseg, c->segCap, maxDepth, bb, live, shadowCnt, static_cast<float4*>(c->contrib.p), static_cast<uint2*>(c->gens.p),
                           static_cast<uint32_t*>(c->fetchCnt.p), timing, ScreenOfPath{static_cast<const int*>(c->ownedPixels.p), c->N, c->w}); if (rc) return rc; }
    int g0 = mark();
    hipLaunchKernelGGL(k_accumulate, dim3(grid_for(c, c->N, 256, 8)), dim3(256), 0, c->stream, c->N, static_cast<const int*>(c->ownedPixels.p), static_cast<const float4*>(c->contrib.p), c->accum, ns);
    hipLaunchKernelGGL(k_tally, dim3(1), dim3(64), 0, c->stream, live, shadowCnt, maxDepth, c->nseg, static_cast<unsigned long long*>(c->totals.p));
    int g1 = mark();
    if (timing) { c->spans.push_back({g0, g1, 5, -1}); c->spans.push_back({e0, g1, 6, -1}); }
    HCHECK(hipGetLastError());
    if (timing && c->evCursor > 200000) { int rc = fold_stage_events(c); if (rc) return rc; }   // bound the event pool
  }
  c->spp += float(spp);
  return HYDRA_HIP_OK;
}

int hydra_hip_set_spp(hydra_hip_handle c, float spp) { if (!c) return HYDRA_HIP_EINVAL; c->spp = spp; return HYDRA_HIP_OK; }
float hydra_hip_get_spp(hydra_hip_handle c) { return c ? c->spp : 0.0f; }

int hydra_hip_get_accumulator(hydra_hip_handle c, float* rgba_sums, int width, int height) {
  if (!c || !rgba_sums) return HYDRA_HIP_EINVAL;
  if (width != c->w || height != c->h) return fail(c, HYDRA_HIP_EINVAL, "get_accumulator: bad input resolution");
  if (!c->stateAllocated || c->accum == nullptr) return fail(c, HYDRA_HIP_ESTATE, "get_accumulator: nothing rendered yet");
  HCHECK(hipSetDevice(c->device));
  HCHECK(hipStreamSynchronize(c->stream));
  HCHECK(hipMemcpy(rgba_sums, c->accum, size_t(width) * height * 16, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}
int hydra_hip_get_hdr_image(hydra_hip_handle c, float* rgba, int width, int height) {
  if (!c || !rgba) return HYDRA_HIP_EINVAL;
  if (width != c->w || height != c->h) return fail(c, HYDRA_HIP_EINVAL, "get_hdr_image: bad input resolution");
  if (!c->stateAllocated || c->accum == nullptr) return fail(c, HYDRA_HIP_ESTATE, "get_hdr_image: nothing rendered yet");
  const int rc = hydra_hip_get_accumulator(c, rgba, width, height);
  if (rc) return rc;
  const float inv = c->spp > 0.0f ? 1.0f / c->spp : 0.0f;
  parallel_rows(height, [=](int y0, int y1) { for (size_t i = size_t(y0) * width * 4; i < size_t(y1) * width * 4; i++) rgba[i] *= inv; });
  return HYDRA_HIP_OK;
}
// IntegratorCommon::GetImageToLDR (Common.cpp:319-333) on the device (k_ldr): 4 bytes per pixel cross PCIe instead of 16
int hydra_hip_get_ldr_image(hydra_hip_handle c, uint32_t* out, int width, int height) {
  if (!c || !out) return HYDRA_HIP_EINVAL;
  if (width != c->w || height != c->h) return fail(c, HYDRA_HIP_EINVAL, "get_ldr_image: bad input resolution");
  if (!c->stateAllocated || c->accum == nullptr) return fail(c, HYDRA_HIP_ESTATE, "get_ldr_image: nothing rendered yet");
  HCHECK(hipSetDevice(c->device));
  const int npix = width * height;
  DevBuf d;
  const int rc = dev_alloc(c, d, size_t(npix) * 4);
  if (rc) return rc;
  hipLaunchKernelGGL(k_ldr, dim3(grid_for(c, npix, 256, 8)), dim3(256), 0, c->stream, npix, (const float4*)c->accum, c->spp > 0.0f ? 1.0f / c->spp : 0.0f, static_cast<uint32_t*>(d.p));
  const hipError_t e1 = hipGetLastError();
  const hipError_t e2 = e1 == hipSuccess ? hipMemcpyAsync(out, d.p, size_t(npix) * 4, hipMemcpyDeviceToHost, c->stream) : e1;
  const hipError_t e3 = hipStreamSynchronize(c->stream);
  dev_free(d);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(c, HYDRA_HIP_EDEVICE, "get_ldr_image: a HIP call failed");
  return HYDRA_HIP_OK;
}

int hydra_hip_get_rays_stat(hydra_hip_handle c, HydraRaysStat* out) {
  if (!c || !out) return HYDRA_HIP_EINVAL;
  memset(out, 0, sizeof(*out));
  HCHECK(hipSetDevice(c->device));
  { int rc = fold_stage_events(c); if (rc) return rc; }
  if (c->totals.p) {
    unsigned long long t[4] = {0, 0, 0, 0};
    HCHECK(hipMemcpy(t, c->totals.p, 32, hipMemcpyDeviceToHost));
    out->extensionRays = t[0]; out->shadowRays = t[1]; out->samples = t[2];
  }
  out->raygenTimeMs = float(c->tRaygen); out->traversalTimeMs = float(c->tTrace); out->evalHitMs = float(c->tHit);
  out->samLightTimeMs = float(c->tHit); out->shadowTimeMs = float(c->tShadow); out->shadeTimeMs = float(c->tShade); out->nextBounceMs = float(c->tShade);
  out->accumTimeMs = float(c->tAccum); out->passTimeMs = float(c->tPass);
  out->traceLaunches = c->nTrace; out->shadowLaunches = c->nShadow; out->bounceTimeMs = float(c->tTrace + c->tHit + c->tShadow + c->tShade);
  if (c->tPass > 0) {
    out->traceTimePerCent = int(100.0 * (c->tTrace + c->tShadow) / c->tPass);
    out->raysPerSec = float(double(out->extensionRays + out->shadowRays) / (c->tPass * 1e-3));
  }
  return HYDRA_HIP_OK;
}
int hydra_hip_get_stage_times_per_bounce(hydra_hip_handle c, float* out, int max_depth) {
  if (!c || !out || max_depth < 1 || max_depth > HK_MAX_DEPTH) return HYDRA_HIP_EINVAL;
  HCHECK(hipSetDevice(c->device));
  { int rc = fold_stage_events(c); if (rc) return rc; }
  for (int d = 0; d < max_depth; d++) for (int k = 0; k < 3; k++) out[d * 3 + k] = float(c->tDepth[d][k]);
  return HYDRA_HIP_OK;
}
int hydra_hip_reset_perf_counters(hydra_hip_handle c) {
  if (!c) return HYDRA_HIP_EINVAL;
  { int rc = fold_stage_events(c); if (rc) return rc; }
  c->tTrace = c->tHit = c->tShadow = c->tShade = c->tRaygen = c->tAccum = c->tPass = 0;
  memset(c->tDepth, 0, sizeof(c->tDepth));
  c->nTrace = c->nShadow = 0;
  if (c->totals.p) { HCHECK(hipSetDevice(c->device)); HCHECK(hipMemsetAsync(c->totals.p, 0, 32, c->stream)); }
  return HYDRA_HIP_OK;
}
int hydra_hip_set_option(hydra_hip_handle c, const char* name, int value) {
  if (!c || !name) return HYDRA_HIP_EINVAL;
  const std::string n(name);
  if (n == "trace_mode") { if (value < 0 || value > 1) return fail(c, HYDRA_HIP_EINVAL, "trace_mode: 0 or 1"); c->traceMode = value; }
  else if (n == "trace_min_active") { if (value < 0 || value > 64) return fail(c, HYDRA_HIP_EINVAL, "trace_min_active: 0..64"); c->traceMinActive = value; }
  else if (n == "trace_vote") c->traceVote = value ? 1 : 0;
  else if (n == "trace_vote_wq" || n == "trace_vote_wt" || n == "trace_vote_wi") { if (value < 1 || value > 64) return fail(c, HYDRA_HIP_EINVAL, n + ": 1..64"); c->traceVoteW[n == "trace_vote_wq" ? 0 : (n == "trace_vote_wt" ? 1 : 2)] = value; }
  else if (n == "shade_waves") { if (value < 3 || value > 5) return fail(c, HYDRA_HIP_EINVAL, "shade_waves: 3, 4 or 5"); c->shadeWaves = value; }
  else if (n == "shade_blocks_per_cu") { if (value < 1 || value > 4096) return fail(c, HYDRA_HIP_EINVAL, "shade_blocks_per_cu: 1..4096"); c->shadeBlocksPerCU = value; }
  else if (n == "static_blocks_per_cu") { if (value < 1 || value > 64) return fail(c, HYDRA_HIP_EINVAL, "static_blocks_per_cu: 1..64"); c->staticBlocksPerCU = value; }
  else if (n == "trace_blocks_per_cu") { if (value < 1 || value > 64) return fail(c, HYDRA_HIP_EINVAL, "trace_blocks_per_cu: 1..64"); c->traceBlocksPerCU = value; }
  else if (n == "trace_rays_per_lane") { if (value < 1 || value > 64) return fail(c, HYDRA_HIP_EINVAL, "trace_rays_per_lane: 1..64"); c->traceRaysPerLane = value; }
  else if (n == "path_order") { if (value < 0 || value > 1) return fail(c, HYDRA_HIP_EINVAL, "path_order: 0 or 1"); c->streamMajor = value; }
  else if (n == "leaf_count_links") { if (value < 0 || value > 1) return fail(c, HYDRA_HIP_EINVAL, "leaf_count_links: 0 or 1"); c->leafEncWanted = value; }
  else if (n == "top_tris_in_lds") { if (value < 0 || value > HK_TOP_TRIS) return fail(c, HYDRA_HIP_EINVAL, "top_tris_in_lds: 0.." + std::to_string(HK_TOP_TRIS)); c->topTrisWanted = value; }
  else if (n == "shadow_unordered") { if (value < 0 || value > 1) return fail(c, HYDRA_HIP_EINVAL, "shadow_unordered: 0 or 1"); c->shadowUnordered = value; }
  else if (n == "top_quads_in_lds") { if (value < 0 || value > HK_TOP_QUADS) return fail(c, HYDRA_HIP_EINVAL, "top_quads_in_lds: 0.." + std::to_string(HK_TOP_QUADS)); c->topWanted = value; }
  else if (n == "scene_tables_in_lds") { if (value < 0 || value > 2) return fail(c, HYDRA_HIP_EINVAL, "scene_tables_in_lds: 0, 1 or 2"); c->sceneTablesInLds = value; }
  else if (n == "sort_paths") { if (value < 0 || value > 1) return fail(c, HYDRA_HIP_EINVAL, "sort_paths: 0 or 1"); c->sortPathsWanted = value; }
  else if (n == "sort_paths_from_bounce") { if (value < 0 || value > HK_MAX_DEPTH) return fail(c, HYDRA_HIP_EINVAL, "sort_paths_from_bounce: 0..64"); c->sortPathsFromDepth = value; }
  else if (n == "srgb_table") { if (value < 0 || value > 1) return fail(c, HYDRA_HIP_EINVAL, "srgb_table: 0 or 1"); c->srgbLutWanted = value; }
  else if (n == "fused_bounce") {
    if (value < 0 || value > 1) return fail(c, HYDRA_HIP_EINVAL, "fused_bounce: 0 or 1");
    if (value != c->fusedBounce) { (void)hipStreamSynchronize(c->stream); c->fusedBounce = value; c->stateAllocated = false; }
  }
  else if (n == "samples_in_flight") {
    if (value < 0 || value > 512) return fail(c, HYDRA_HIP_EINVAL, "samples_in_flight: 0 (by resolution) or 1..512");
    if (value != c->streamsWanted) { c->streamsWanted = value; c->stateAllocated = false; }   // generators are re-seeded by the next init_path_tracing
  }
  else if (n == "queue_segments") {
    if (value < 1 || value > HK_MAX_SEG) return fail(c, HYDRA_HIP_EINVAL, "queue_segments: 1..64");
    if (value != c->nsegWanted) { c->nsegWanted = value; c->stateAllocated = false; }   // slot map is rebuilt by the next pass; accumulated image stays
  }
  else return fail(c, HYDRA_HIP_EINVAL, "set_option: unknown option " + n);
  return HYDRA_HIP_OK;
}
int hydra_hip_get_option(hydra_hip_handle c, const char* name, int* value) {
  if (!c || !name || !value) return HYDRA_HIP_EINVAL;
  const std::string n(name);
  if (n == "trace_mode") *value = c->traceMode;
  else if (n == "trace_min_active") *value = c->traceMinActive;
  else if (n == "trace_vote") *value = c->traceVote;
  else if (n == "trace_vote_wq") *value = c->traceVoteW[0];
  else if (n == "trace_vote_wt") *value = c->traceVoteW[1];
  else if (n == "trace_vote_wi") *value = c->traceVoteW[2];
  else if (n == "shadow_unordered") *value = c->shadowUnordered;
  else if (n == "trace_rays_per_lane") *value = c->traceRaysPerLane;
  else if (n == "shade_waves") *value = c->shadeWaves;
  else if (n == "shade_blocks_per_cu") *value = c->shadeBlocksPerCU;
  else if (n == "static_blocks_per_cu") *value = c->staticBlocksPerCU;
  else if (n == "trace_blocks_per_cu") *value = c->traceBlocksPerCU;
  else if (n == "queue_segments") *value = c->nsegWanted;
  else if (n == "fused_bounce") *value = c->fusedBounce;
  else if (n == "srgb_table") *value = c->srgbLutWanted;
  else if (n == "last_gbuffer_device_us") *value = c->lastGbufferUs;
  else if (n == "sort_paths") *value = c->sortPathsWanted;
  else if (n == "sort_paths_from_bounce") *value = c->sortPathsFromDepth;
  else if (n == "scene_tables_in_lds") *value = c->sceneTablesInLds;
  else if (n == "path_order") *value = c->streamMajor;
  else if (n == "leaf_count_links") *value = c->leafEncWanted;
  else if (n == "top_quads_in_lds") *value = c->topWanted;
  else if (n == "top_tris_in_lds") *value = c->topTrisWanted;
  else if (n == "samples_in_flight") *value = c->streamsWanted > 0 ? c->streamsWanted : auto_streams(size_t(c->w) * c->h);
  else return fail(c, HYDRA_HIP_EINVAL, "get_option: unknown option " + n);
  return HYDRA_HIP_OK;
}
int hydra_hip_enable_traversal_counters(hydra_hip_handle c, int enable) {
  if (!c) return HYDRA_HIP_EINVAL;
  HCHECK(hipSetDevice(c->device));
  if (enable) {
    int rc = dev_alloc(c, c->travTotals, size_t(HK_MAX_DEPTH) * HK_TT_ROW * 8);
    if (rc) return rc;
    HCHECK(hipMemsetAsync(c->travTotals.p, 0, size_t(HK_MAX_DEPTH) * HK_TT_ROW * 8, c->stream));
  }
  c->travCounters = (enable != 0);
  return HYDRA_HIP_OK;
}
int hydra_hip_get_traversal_counters(hydra_hip_handle c, uint64_t* out, int max_depth) {
  if (!c || !out || max_depth < 1 || max_depth > HK_MAX_DEPTH) return HYDRA_HIP_EINVAL;
  if (!c->travTotals.p) return fail(c, HYDRA_HIP_ESTATE, "get_traversal_counters: enable_traversal_counters first");
  HCHECK(hipSetDevice(c->device));
  std::vector<uint64_t> raw(size_t(max_depth) * HK_TT_ROW);
  HCHECK(hipMemcpy(raw.data(), c->travTotals.p, raw.size() * 8, hipMemcpyDeviceToHost));
  for (int d = 0; d < max_depth; d++) for (int k = 0; k < 2; k++) for (int j = 0; j < 5; j++) out[(size_t(d) * 2 + k) * 5 + j] = raw[size_t(d) * HK_TT_ROW + k * 6 + j];
  return HYDRA_HIP_OK;
}
int hydra_hip_get_traversal_oob(hydra_hip_handle c, uint64_t* out) {
  if (!c || !out) return HYDRA_HIP_EINVAL;
  if (!c->travTotals.p) return fail(c, HYDRA_HIP_ESTATE, "get_traversal_oob: enable_traversal_counters first");
  HCHECK(hipSetDevice(c->device));
  std::vector<uint64_t> raw(size_t(HK_MAX_DEPTH) * HK_TT_ROW);
  HCHECK(hipMemcpy(raw.data(), c->travTotals.p, raw.size() * 8, hipMemcpyDeviceToHost));
  *out = 0;
  for (int d = 0; d < HK_MAX_DEPTH; d++) *out += raw[size_t(d) * HK_TT_ROW + 5] + raw[size_t(d) * HK_TT_ROW + 11];
  return HYDRA_HIP_OK;
}
int hydra_hip_enable_stage_timing(hydra_hip_handle c, int enable) { if (!c) return HYDRA_HIP_EINVAL; c->stageTiming = (enable != 0); return HYDRA_HIP_OK; }

// ------------------------------------------------------------------------------------------------ stage entry points
struct TmpBufs {
  std::vector<void*> ptrs;
  ~TmpBufs() { for (void* p : ptrs) (void)hipFree(p); }
  void* up(hydra_hip_ctx* c, const void* src, size_t bytes, int& rc) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes > 0 ? bytes : 16) != hipSuccess) { c->err = "hipMalloc failed in stage call"; rc = HYDRA_HIP_ENOMEM; return nullptr; }
    ptrs.push_back(p);
    if (src && bytes) { if (hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { c->err = "hipMemcpy H2D failed"; rc = HYDRA_HIP_EDEVICE; } }
    return p;
  }
};
#define STAGE_PROLOG(needScene)                                                                         \
  if (!c || n <= 0) return HYDRA_HIP_EINVAL;                                                           \
  if ((needScene) && !scene_ready(c)) return fail(c, HYDRA_HIP_ESTATE, "stage call: scene is not uploaded"); \
  HCHECK(hipSetDevice(c->device));                                                                      \
  if (needScene) { const int prc_ = prepare_geometry(c); if (prc_) return prc_; }                       \
  if (needScene) { const int vrc_ = validate_materials(c); if (vrc_) return vrc_; }                      \
  if (needScene) { const int crc_ = prepare_classes(c); if (crc_) return crc_; }                         \
  TmpBufs tb; int rc = HYDRA_HIP_OK;
#define STAGE_EPILOG()                                                                                  \
  HCHECK(hipGetLastError());                                                                            \
  HCHECK(hipStreamSynchronize(c->stream));

// One accept / reject step of n Markov chains through the PRODUCTION kernel (k_mmlt_accept -> mmltAcceptReject, hk_bidir.h): chain i holds old8[i] (colour xyz, -, -, -, -, contribFunc)
// at pixel (2i, 0), the proposal new8[i] lands on pixel (2i + 1, 0) of a 2n x 1 image, so the image IS the pair of contributions; xCur = 0, xNew = 1: accepted chains show 1.
int hydra_hip_stage_mmlt_accept(hydra_hip_handle c, int n, const float* old8, const float* new8, uint32_t* gen2, float bk_scale, float* out12) {
  STAGE_PROLOG(false);
  if (!old8 || !new8 || !gen2 || !out12) return fail(c, HYDRA_HIP_EINVAL, "stage_mmlt_accept: null argument");
  std::vector<float> ch(size_t(CH_PLANES) * n, 0.0f), o8(new8, new8 + size_t(n) * 8);
  for (int i = 0; i < n; i++) {
    ch[size_t(CH_Y) * n + i] = old8[size_t(i) * 8 + 7];
    for (int k = 0; k < 3; k++) ch[size_t(CH_COLOR + k) * n + i] = old8[size_t(i) * 8 + k];
    ch[size_t(CH_XS) * n + i] = float(2 * i); ch[size_t(CH_YS) * n + i] = 0.0f;
    memcpy(&ch[size_t(CH_GEN2) * n + i], &gen2[2 * i], 4); memcpy(&ch[size_t(CH_GEN2 + 1) * n + i], &gen2[2 * i + 1], 4);
    o8[size_t(i) * 8 + 3] = float(2 * i + 1); o8[size_t(i) * 8 + 4] = 0.0f;
  }
  const int stride = mmltStride(2);
  std::vector<int> depth(size_t(n), 2);
  std::vector<float> xc(size_t(stride) * n, 0.0f), xn(size_t(stride) * n, 1.0f);
  float* dch = (float*)tb.up(c, ch.data(), ch.size() * 4, rc);
  int* ddepth = (int*)tb.up(c, depth.data(), depth.size() * 4, rc);
  float* dxc = (float*)tb.up(c, xc.data(), xc.size() * 4, rc);
  float* dxn = (float*)tb.up(c, xn.data(), xn.size() * 4, rc);
  float* do8 = (float*)tb.up(c, o8.data(), o8.size() * 4, rc);
  float* dimg = (float*)tb.up(c, nullptr, size_t(2 * n) * 16, rc);
  if (rc) return rc;
  HCHECK(hipMemsetAsync(dimg, 0, size_t(2 * n) * 16, c->stream));
  MmltChains mc; mc.n = n; mc.maxD = 2; mc.ch = dch; mc.depth = ddepth; mc.xCur = dxc; mc.xNew = dxn;
  hipLaunchKernelGGL(k_mmlt_accept, dim3((n + 255) / 256), dim3(256), 0, c->stream, mc, do8, bk_scale, dimg, 2 * n);
  STAGE_EPILOG();
  std::vector<float> img(size_t(2 * n) * 4);
  HCHECK(hipMemcpy(img.data(), dimg, img.size() * 4, hipMemcpyDeviceToHost));
  HCHECK(hipMemcpy(ch.data(), dch, ch.size() * 4, hipMemcpyDeviceToHost));
  HCHECK(hipMemcpy(xc.data(), dxc, size_t(n) * 4, hipMemcpyDeviceToHost));   // plane 0 of xCur
  for (int i = 0; i < n; i++) {
    float* r = out12 + size_t(i) * 12;
    memcpy(r, &img[size_t(2 * i) * 4], 16); memcpy(r + 4, &img[size_t(2 * i + 1) * 4], 16);
    r[8] = xc[size_t(i)]; r[9] = ch[size_t(CH_ACCEPTED) * n + i]; r[10] = 0.0f; r[11] = 0.0f;
    memcpy(&gen2[2 * i], &ch[size_t(CH_GEN2) * n + i], 4); memcpy(&gen2[2 * i + 1], &ch[size_t(CH_GEN2 + 1) * n + i], 4);
  }
  return HYDRA_HIP_OK;
}

// the miss shader with a back-plate: environmentColorExtended (hk_shading.h) for n rays that left the scene; in8 = ray origin xyz, previous BSDF pdf, previous bounce
// specular (0/1), ray flags (int bits), pixel x, y (int bits)
int hydra_hip_stage_environment(hydra_hip_handle c, int n, const float* ray_dir4, const float* in8, float* out4) {
  STAGE_PROLOG(true);
  if (!ray_dir4 || !in8 || !out4) return fail(c, HYDRA_HIP_EINVAL, "stage_environment: null argument");
  { bool have; const int brc = check_back_plate(c, have); if (brc) return brc; }
  float4* ddir = (float4*)tb.up(c, ray_dir4, size_t(n) * 16, rc);
  float* din = (float*)tb.up(c, in8, size_t(n) * 32, rc);
  float4* dout = (float4*)tb.up(c, nullptr, size_t(n) * 16, rc);
  if (rc) return rc;
  hipLaunchKernelGGL(k_stage_environment, dim3((n + 255) / 256), dim3(256), 0, c->stream, make_scene(c), n, ddir, din, dout);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(out4, dout, size_t(n) * 16, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

// ---- procedural textures (hydra_proctex.hip, hk_proctex_rt.h)
int hydra_hip_proctex_compile(hydra_hip_handle c, const char* source, size_t length) {
  if (!c) return HYDRA_HIP_EINVAL;
  HCHECK(hipSetDevice(c->device));
  HCHECK(hipStreamSynchronize(c->stream));
  hk_proctex_free(c->procTex); c->procTex = nullptr;
  if (source == nullptr || length == 0) return HYDRA_HIP_OK;   // no procedural textures: the program is dropped
  std::string err;
  c->procTex = hk_proctex_build(source, length, err);
  if (c->procTex == nullptr) return fail(c, HYDRA_HIP_EINVAL, err);
  return HYDRA_HIP_OK;
}
// the same compilation without a device or a context: does this text build?  (the build log, or the error, through hydra_hip_last_error(NULL))
int hydra_hip_proctex_check(const char* source, size_t length) {
  if (source == nullptr || length == 0) return fail(nullptr, HYDRA_HIP_EINVAL, "proctex_check: no text");
  std::vector<char> code;
  std::string log, err;
  if (!hk_proctex_compile(source, length, code, log, err)) return fail(nullptr, HYDRA_HIP_EINVAL, err);
  g_createError = log;
  return HYDRA_HIP_OK;
}
// the lists the next stage_shade_point / stage_bounce calls of the same n consult: ids[max_num][n], colours as halfs [max_num][n][4]; n = 0 drops them
int hydra_hip_stage_set_proctex(hydra_hip_handle c, int n, int max_num, const int32_t* ids, const uint16_t* halfs4) {
  if (!c) return HYDRA_HIP_EINVAL;
  HCHECK(hipSetDevice(c->device));
  c->stagePtlN = 0; c->stagePtlMax = 0;
  if (n <= 0) return HYDRA_HIP_OK;
  if (max_num < 1 || max_num > 16 || !ids || !halfs4) return fail(c, HYDRA_HIP_EINVAL, "stage_set_proctex: bad arguments");
  int rc = dev_upload(c, c->stagePtlIds, ids, size_t(n) * size_t(max_num) * 4); if (rc) return rc;
  rc = dev_upload(c, c->stagePtlVals, halfs4, size_t(n) * size_t(max_num) * 8); if (rc) return rc;
  c->stagePtlN = n; c->stagePtlMax = max_num;
  return HYDRA_HIP_OK;
}
static void stage_proctex_lists(hydra_hip_ctx* c, int n, SceneDev& s) {
  if (c->stagePtlN != n || c->stagePtlMax <= 0) return;
  s.ptlIds = static_cast<const int*>(c->stagePtlIds.p); s.ptlVals = static_cast<const uint2*>(c->stagePtlVals.p); s.ptlStride = n; s.ptlMax = c->stagePtlMax;
}
// the compiled program on n hits handed in: ids[max_num][n] (HYDRA_INVALID_TEXTURE ends a point's list), colours as halfs [max_num][n][4]
int hydra_hip_stage_proctex(hydra_hip_handle c, int n, int max_num, const float* ray_pos4, const float* ray_dir4, const HydraLiteHit* hits, int32_t* ids, uint16_t* halfs4) {
  STAGE_PROLOG(true);
  if (!ray_pos4 || !ray_dir4 || !hits || !ids || !halfs4 || max_num < 1 || max_num > 16) return fail(c, HYDRA_HIP_EINVAL, "stage_proctex: bad arguments");
  if (c->procTex == nullptr) return fail(c, HYDRA_HIP_ESTATE, "stage_proctex: no program compiled (hydra_hip_proctex_compile)");
  float4* dpos = (float4*)tb.up(c, ray_pos4, size_t(n) * 16, rc);
  float4* ddir = (float4*)tb.up(c, ray_dir4, size_t(n) * 16, rc);
  HydraLiteHit* dh = (HydraLiteHit*)tb.up(c, hits, size_t(n) * 16, rc);
  int* dids = (int*)tb.up(c, nullptr, size_t(n) * size_t(max_num) * 4, rc);
  uint2* dvals = (uint2*)tb.up(c, nullptr, size_t(n) * size_t(max_num) * 8, rc);
  if (rc) return rc;
  HCHECK(hipMemsetAsync(dids, 0xff, size_t(n) * size_t(max_num) * 4, c->stream));
  HCHECK(hipMemsetAsync(dvals, 0, size_t(n) * size_t(max_num) * 8, c->stream));
  HCHECK(hk_proctex_launch_points(c->procTex, c->stream, make_scene(c), n, dpos, ddir, dh, dids, dvals, n, max_num));
  STAGE_EPILOG();
  HCHECK(hipMemcpy(ids, dids, size_t(n) * size_t(max_num) * 4, hipMemcpyDeviceToHost));
  HCHECK(hipMemcpy(halfs4, dvals, size_t(n) * size_t(max_num) * 8, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

int hydra_hip_stage_make_eye_rays(hydra_hip_handle c, int n, const int32_t* xy, const float* offs4, float* ray_pos4, float* ray_dir4) {
  if (!c || n <= 0 || !c->globals.p) return fail(c, HYDRA_HIP_ESTATE, "stage_make_eye_rays: globals are not uploaded");
  HCHECK(hipSetDevice(c->device));
  TmpBufs tb; int rc = HYDRA_HIP_OK;
  int* dxy = (int*)tb.up(c, xy, size_t(n) * 8, rc);
  float4* doffs = (float4*)tb.up(c, offs4, size_t(n) * 16, rc);
  float4* dpos = (float4*)tb.up(c, nullptr, size_t(n) * 16, rc);
  float4* ddir = (float4*)tb.up(c, nullptr, size_t(n) * 16, rc);
  if (rc) return rc;
  SceneDev s = make_scene(c);
  hipLaunchKernelGGL(k_stage_eye, dim3((n + 255) / 256), dim3(256), 0, c->stream, s, n, c->w, c->h, dxy, doffs, dpos, ddir);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(ray_pos4, dpos, size_t(n) * 16, hipMemcpyDeviceToHost));
  HCHECK(hipMemcpy(ray_dir4, ddir, size_t(n) * 16, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

int hydra_hip_stage_trace(hydra_hip_handle c, int n, const float* ray_pos4, const float* ray_dir4, HydraLiteHit* hits, uint32_t* counters3) {
  STAGE_PROLOG(true);
  float4* dpos = (float4*)tb.up(c, ray_pos4, size_t(n) * 16, rc);
  float4* ddir = (float4*)tb.up(c, ray_dir4, size_t(n) * 16, rc);
  HydraLiteHit* dh = (HydraLiteHit*)tb.up(c, nullptr, size_t(n) * 16, rc);
  uint32_t* dc = counters3 ? (uint32_t*)tb.up(c, nullptr, size_t(n) * 12, rc) : nullptr;
  if (rc) return rc;
  SceneDev s = make_scene(c);
  if ((rc = ensure_fetch_counters(c))) return rc;
  uint32_t* fetch = static_cast<uint32_t*>(c->fetchCnt.p) + size_t(2 * HK_MAX_DEPTH + 2) * HK_CROW;
  HCHECK(hipMemsetAsync(fetch, 0, 4, c->stream));
  launch_closest(c, s, seg_q(nullptr, n, 1, n), dpos, ddir, dh, dc, nullptr, fetch);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(hits, dh, size_t(n) * 16, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) if (hits[i].primId != -1) hits[i].geomId = HK_GEOM_ID(hits[i].geomId);   // the device's class label is not part of Lite_Hit
  if (counters3) HCHECK(hipMemcpy(counters3, dc, size_t(n) * 12, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

int hydra_hip_stage_shadow_trace(hydra_hip_handle c, int n, const float* ray_pos4, const float* ray_dir4, const float* t_far, float* visibility) {
  STAGE_PROLOG(true);
  std::vector<float> org(size_t(n) * 4);
  for (int i = 0; i < n; i++) { org[4 * i] = ray_pos4[4 * i]; org[4 * i + 1] = ray_pos4[4 * i + 1]; org[4 * i + 2] = ray_pos4[4 * i + 2]; org[4 * i + 3] = t_far[i]; }
  float4* dorg = (float4*)tb.up(c, org.data(), size_t(n) * 16, rc);
  float4* ddir = (float4*)tb.up(c, ray_dir4, size_t(n) * 16, rc);
  float* dv = (float*)tb.up(c, nullptr, size_t(n) * 4, rc);
  if (rc) return rc;
  SceneDev s = make_scene(c);
  if ((rc = ensure_fetch_counters(c))) return rc;
  uint32_t* fetch = static_cast<uint32_t*>(c->fetchCnt.p) + size_t(2 * HK_MAX_DEPTH + 2) * HK_CROW;
  HCHECK(hipMemsetAsync(fetch, 0, 4, c->stream));
  launch_shadow(c, s, seg_q(nullptr, n, 1, n), dorg, ddir, dv, nullptr, fetch);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(visibility, dv, size_t(n) * 4, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

int hydra_hip_stage_eval_surface(hydra_hip_handle c, int n, const float* ray_pos4, const float* ray_dir4, const HydraLiteHit* hits, float* surf24) {
  STAGE_PROLOG(true);
  float4* dpos = (float4*)tb.up(c, ray_pos4, size_t(n) * 16, rc);
  float4* ddir = (float4*)tb.up(c, ray_dir4, size_t(n) * 16, rc);
  HydraLiteHit* dh = (HydraLiteHit*)tb.up(c, hits, size_t(n) * 16, rc);
  float* dout = (float*)tb.up(c, nullptr, size_t(n) * 96, rc);
  if (rc) return rc;
  SceneDev s = make_scene(c);
  hipLaunchKernelGGL(k_stage_surface, dim3((n + 255) / 256), dim3(256), 0, c->stream, s, n, dpos, ddir, dh, dout);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(surf24, dout, size_t(n) * 96, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

int hydra_hip_stage_shade_point(hydra_hip_handle c, int n, const float* surf24, const float* ray_dir4, const int32_t* flags, const float* rnd_light4,
                                const float* rands10, float* out28) {
  STAGE_PROLOG(true);
  if (!surf24 || !ray_dir4 || !flags || !rnd_light4 || !rands10 || !out28) return fail(c, HYDRA_HIP_EINVAL, "stage_shade_point: null argument");
  float* dsurf = (float*)tb.up(c, surf24, size_t(n) * 96, rc);
  float4* ddir = (float4*)tb.up(c, ray_dir4, size_t(n) * 16, rc);
  int* dfl = (int*)tb.up(c, flags, size_t(n) * 4, rc);
  float4* drl = (float4*)tb.up(c, rnd_light4, size_t(n) * 16, rc);
  float* drn = (float*)tb.up(c, rands10, size_t(n) * 40, rc);
  float* dout = (float*)tb.up(c, nullptr, size_t(n) * 112, rc);
  if (rc) return rc;
  SceneDev s = make_scene(c);
  stage_proctex_lists(c, n, s);
  hipLaunchKernelGGL(k_stage_shade_point, dim3((n + 255) / 256), dim3(256), 0, c->stream, s, n, dsurf, ddir, dfl, drl, drn, dout);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(out28, dout, size_t(n) * 112, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

int hydra_hip_stage_bounce(hydra_hip_handle c, int n, int depth, int max_depth, const float* ray_pos4, const float* ray_dir4, const float* surf24, const float* in16,
                           const float* rands10, float* out40) {
  STAGE_PROLOG(true);
  if (!ray_pos4 || !ray_dir4 || !surf24 || !in16 || !rands10 || !out40) return fail(c, HYDRA_HIP_EINVAL, "stage_bounce: null argument");
  { bool have; const int brc = check_back_plate(c, have); if (brc) return brc; }
  float4* dpos = (float4*)tb.up(c, ray_pos4, size_t(n) * 16, rc);
  float4* ddir = (float4*)tb.up(c, ray_dir4, size_t(n) * 16, rc);
  float* dsurf = (float*)tb.up(c, surf24, size_t(n) * 96, rc);
  float* din = (float*)tb.up(c, in16, size_t(n) * 64, rc);
  float* dr = (float*)tb.up(c, rands10, size_t(n) * 40, rc);
  float* dout = (float*)tb.up(c, nullptr, size_t(n) * 160, rc);
  if (rc) return rc;
  SceneDev s = make_scene(c);
  stage_proctex_lists(c, n, s);
  hipLaunchKernelGGL(k_stage_bounce, dim3((n + 127) / 128), dim3(128), 0, c->stream, s, n, depth, max_depth, dpos, ddir, dsurf, din, dr, dout);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(out40, dout, size_t(n) * 160, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}
int hydra_hip_stage_path_trace(hydra_hip_handle c, int n, const float* ray_pos4, const float* ray_dir4, uint32_t* rng_state2, float* color4) {
  // n caller-provided primary rays + RandomGen states run through the PRODUCTION wavefront kernels (path i plays pixel i)
  STAGE_PROLOG(true);
  const int maxDepth = c->hostHeader[HG_VARS_I + HV_I_TRACE_DEPTH];
  if (maxDepth < 1 || maxDepth > HK_MAX_DEPTH) return fail(c, HYDRA_HIP_EINVAL, "stage_path_trace: HRT_TRACE_DEPTH out of range");
  float4* dpos = (float4*)tb.up(c, ray_pos4, size_t(n) * 16, rc);
  float4* ddir = (float4*)tb.up(c, ray_dir4, size_t(n) * 16, rc);
  uint2* drng = (uint2*)tb.up(c, rng_state2, size_t(n) * 8, rc);
  float4* bufs[22];
  for (auto& b : bufs) b = (float4*)tb.up(c, nullptr, size_t(n) * 16, rc);
  uint2* sRng = (uint2*)tb.up(c, nullptr, size_t(n) * 8, rc);
  uint2* mRng = (uint2*)tb.up(c, nullptr, size_t(n) * 8, rc);
  uint2* gensOut = (uint2*)tb.up(c, nullptr, size_t(n) * 8, rc);
  float* vis = (float*)tb.up(c, nullptr, size_t(n) * 4, rc);
  uint32_t* counters = (uint32_t*)tb.up(c, nullptr, size_t(4 * maxDepth + 8) * HK_CROW * 4, rc);
  if (rc) return rc;
  uint2* tRng = (uint2*)tb.up(c, nullptr, size_t(n) * 8, rc);
  if (rc) return rc;
  const PathState S = {bufs[0], bufs[1], bufs[2], bufs[3], sRng, bufs[15]};
  BounceBufs bb;
  bb.A = S;
  bb.B = {bufs[16], bufs[17], bufs[18], bufs[19], tRng, bufs[20]};
  bb.M = {bufs[4], bufs[5], bufs[6], mRng, bufs[7], bufs[8], bufs[9], bufs[10], bufs[11], bufs[12], vis};
  bb.sh = {bufs[12], bufs[21], vis};
  bb.hits = reinterpret_cast<HydraLiteHit*>(bufs[13]);
  float4* contrib = bufs[14];
  uint32_t* live = counters, *shadowCnt = counters + size_t(maxDepth + 2) * HK_CROW, *fetch = counters + size_t(2 * maxDepth + 4) * HK_CROW;
  HCHECK(hipMemsetAsync(counters, 0, size_t(4 * maxDepth + 8) * HK_CROW * 4, c->stream));
  HCHECK(hipMemcpyAsync(live, &n, 4, hipMemcpyHostToDevice, c->stream));
  HCHECK(hipMemsetAsync(contrib, 0, size_t(n) * 16, c->stream));
  SceneDev s = make_scene(c);
  hipLaunchKernelGGL(k_stage_seed_paths, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dpos, ddir, drng, S);
  rc = run_bounces(c, s, 1, n, maxDepth, bb, live, shadowCnt, contrib, gensOut, fetch, false, ScreenOfPath{nullptr, 0, c->w});   // path i plays pixel i
  if (rc) return rc;
  STAGE_EPILOG();
  HCHECK(hipMemcpy(color4, contrib, size_t(n) * 16, hipMemcpyDeviceToHost));
  HCHECK(hipMemcpy(rng_state2, gensOut, size_t(n) * 8, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

// The device buffers of one evaluation of F for n chains (hk_bidir.h MmltView), owned by the caller.
struct MmltBufs {
  MmltView v;
  MmltRays rays[2];        // ping-pong ray queues, nseg segments of cap slots
  HydraLiteHit* hits;      // per ray slot
  uint32_t* counts;        // (HK_MMLT_MAX_DEPTH + 2) counter rows of HK_CROW words: live rays per level and segment
  int nseg, cap;
  HydraLiteHit* eyeHit; float* shVis;   // the writable twins of v.eyeHit / v.shVis
};
static void mmlt_queue_shape(int n, int& nseg, int& cap) {
  const int blocks = (n + 255) / 256;
  nseg = std::max(1, std::min(32, blocks));
  cap = ((blocks + nseg - 1) / nseg) * 256 * 2;      // every chain of a segment may start a camera and a light sub-path; a level never grows
}
static int mmlt_alloc(hydra_hip_ctx* c, TmpBufs& tb, int n, int maxD, MmltBufs& b) {
  int rc = HYDRA_HIP_OK;
  b.v.n = n; b.v.maxD = maxD;
  b.v.st = (float*)tb.up(c, nullptr, size_t(mmltPlanes(maxD)) * n * 4, rc);
  mmlt_queue_shape(n, b.nseg, b.cap);
  const size_t slots = size_t(b.nseg) * b.cap;
  for (int k = 0; k < 2; k++) {
    b.rays[k].pos = (float4*)tb.up(c, nullptr, slots * 16, rc); b.rays[k].dir = (float4*)tb.up(c, nullptr, slots * 16, rc); b.rays[k].owner = (int*)tb.up(c, nullptr, slots * 4, rc);
  }
  b.hits = (HydraLiteHit*)tb.up(c, nullptr, slots * 16, rc);
  b.counts = (uint32_t*)tb.up(c, nullptr, size_t(HK_MMLT_MAX_DEPTH + 2) * HK_CROW * 4, rc);
  b.v.eyePos = (float4*)tb.up(c, nullptr, size_t(n) * 16, rc); b.v.eyeDir = (float4*)tb.up(c, nullptr, size_t(n) * 16, rc);
  b.eyeHit = (HydraLiteHit*)tb.up(c, nullptr, size_t(n) * 16, rc);
  b.v.shPos = (float4*)tb.up(c, nullptr, size_t(n) * 16, rc); b.v.shDir = (float4*)tb.up(c, nullptr, size_t(n) * 16, rc);
  b.shVis = (float*)tb.up(c, nullptr, size_t(n) * 4, rc);
  b.v.eyeHit = b.eyeHit; b.v.shVis = b.shVis;
  return rc;
}
// F for every chain of the view: v.x, v.depth and v.out8 are set by the caller; maxDepth = the largest d among the chains.
// The stage kernels exist in three feature sets like k_bounce (hk_shading.h, HK_FEAT_*): the sky / delta-light / Oren-Nayar subset, the classic set,
// everything -- with every BxDF inlined a stage kernel is 47 k vector instructions, most of which a scene like test_42 never runs
static int mmlt_launch(hydra_hip_ctx* c, int kernel, MmltLaunch& a) {   // kernel: 0 = k_mmlt_step, 1 = k_mmlt_connect_begin, 2 = k_mmlt_connect_end
  const int f_ = c->sceneFeatures;
  const int F = ((f_ & ~(HK_FEAT_SKY | HK_FEAT_DELTA_LIGHTS | HK_FEAT_OREN_NAYAR)) == 0) ? (HK_FEAT_SKY | HK_FEAT_DELTA_LIGHTS | HK_FEAT_OREN_NAYAR) : (((f_ & ~HK_FEAT_CLASSIC) == 0) ? HK_FEAT_CLASSIC : HK_FEAT_ALL);
  a.stream = c->stream;
  if (!(hk_launch_mmlt_lean(kernel, F, a) || hk_launch_mmlt_all(kernel, F, a))) return fail(c, HYDRA_HIP_ESTATE, "mmlt: no stage-kernel instantiation for features " + std::to_string(F));
  return HYDRA_HIP_OK;
}
static int mmlt_eval(hydra_hip_ctx* c, const SceneDev& s, const MmltBufs& b, int maxDepth) {
  const MmltView& v = b.v;
  const int n = v.n;
  int rc = ensure_fetch_counters(c);
  if (rc) return rc;
  uint32_t* fetch = static_cast<uint32_t*>(c->fetchCnt.p) + size_t(2 * HK_MAX_DEPTH + 2) * HK_CROW;
  HCHECK(hipMemsetAsync(b.counts, 0, size_t(maxDepth + 2) * HK_CROW * 4, c->stream));
  const int bps = ((n + 255) / 256 + b.nseg - 1) / b.nseg;
  hipLaunchKernelGGL(k_mmlt_begin, dim3(bps * b.nseg), dim3(256), 0, c->stream, s, v, b.nseg, b.cap, b.rays[0], b.counts);
  MmltLaunch ml;
  ml.s = s; ml.v = v; ml.currDepth = 0; ml.q = seg_q(nullptr, 0, 1, 0); ml.in = b.rays[0]; ml.hits = b.hits; ml.out = b.rays[1]; ml.outCount = b.counts;
  for (int k = 1; k <= maxDepth; k++) {
    const MmltRays in = b.rays[(k - 1) & 1], out = b.rays[k & 1];
    const SegQ q = seg_q(b.counts + size_t(k - 1) * HK_CROW, 0, b.nseg, b.cap);
    HCHECK(hipMemsetAsync(fetch, 0, size_t(HK_CROW) * 4, c->stream));
    launch_closest(c, s, q, in.pos, in.dir, b.hits, nullptr, nullptr, fetch);
    ml.grid = seg_grid(c, q, 256, 64); ml.currDepth = k; ml.q = q; ml.in = in; ml.hits = b.hits; ml.out = out; ml.outCount = b.counts + size_t(k) * HK_CROW;
    if ((rc = mmlt_launch(c, 0, ml))) return rc;
  }
  ml.grid = (n + 255) / 256;
  if ((rc = mmlt_launch(c, 1, ml))) return rc;
  // the two connection rays of every chain, as a segmented queue over the chain order (one fetch counter for all persistent waves
  // saturates at ~88 fetches per microsecond: 0.19 ms per launch of 1 M rays, more than their traversal takes)
  const int capC = ((n + b.nseg - 1) / b.nseg + 63) / 64 * 64;
  uint32_t* connCounts = b.counts + size_t(maxDepth + 1) * HK_CROW;
  hipLaunchKernelGGL(k_fill_seg_counts, dim3(1), dim3(64), 0, c->stream, n, capC, b.nseg, connCounts);
  const SegQ qc = seg_q(connCounts, 0, b.nseg, capC);
  HCHECK(hipMemsetAsync(fetch, 0, size_t(HK_CROW) * 4, c->stream));
  launch_closest(c, s, qc, v.eyePos, v.eyeDir, b.eyeHit, nullptr, nullptr, fetch);
  HCHECK(hipMemsetAsync(fetch, 0, size_t(HK_CROW) * 4, c->stream));
  launch_shadow(c, s, qc, v.shPos, v.shDir, b.shVis, nullptr, fetch);
  if ((rc = mmlt_launch(c, 2, ml))) return rc;
  HCHECK(hipGetLastError());
  return HYDRA_HIP_OK;
}
// ------------------------------------------------------------------------------------------------ IntegratorMMLT, the product path of row f3
static MmltChains mmlt_chains(hydra_hip_ctx* c) {
  MmltChains ch;
  ch.n = c->mmlt.n; ch.maxD = c->mmlt.maxD;
  ch.ch = (float*)c->mmlt.ch.p; ch.depth = (const int*)c->mmlt.depth.p; ch.xCur = (float*)c->mmlt.xCur.p; ch.xNew = (float*)c->mmlt.xNew.p;
  return ch;
}
static MmltBufs mmlt_run_bufs(hydra_hip_ctx* c) {
  MmltBufs b;
  auto& m = c->mmlt;
  b.v.n = m.n; b.v.maxD = m.maxD; b.v.st = (float*)m.st.p; b.v.x = (const float*)m.xNew.p; b.v.depth = (const int*)m.depth.p;
  mmlt_queue_shape(m.n, b.nseg, b.cap);
  for (int k = 0; k < 2; k++) { b.rays[k].pos = (float4*)m.rayPos[k].p; b.rays[k].dir = (float4*)m.rayDir[k].p; b.rays[k].owner = (int*)m.rayOwner[k].p; }
  b.hits = (HydraLiteHit*)m.hits.p; b.counts = (uint32_t*)m.counts.p;
  b.v.eyePos = (float4*)m.eyePos.p; b.v.eyeDir = (float4*)m.eyeDir.p; b.eyeHit = (HydraLiteHit*)m.eyeHit.p;
  b.v.shPos = (float4*)m.shPos.p; b.v.shDir = (float4*)m.shDir.p; b.shVis = (float*)m.shVis.p;
  b.v.eyeHit = b.eyeHit; b.v.shVis = b.shVis; b.v.out8 = (float*)m.out8.p;
  return b;
}
static bool mmlt_camera_ready(const hydra_hip_ctx* c) {   // F projects light-path vertices with varsF[HRT_WIDTH_F / HEIGHT_F] and mProj / mWorldView
  if (c->hostHeader.size() <= size_t(HG_VARS_F + HV_F_HEIGHT_F)) return false;
  float wf, hf;
  memcpy(&wf, &c->hostHeader[HG_VARS_F + HV_F_WIDTH_F], 4); memcpy(&hf, &c->hostHeader[HG_VARS_F + HV_F_HEIGHT_F], 4);
  return wf >= 1.0f && hf >= 1.0f;
}
// the on-screen test of a splat uses the header's WIDTH_F x HEIGHT_F, the image it lands in is c->w x c->h: they must be the same frame
static bool header_frame_matches(const hydra_hip_ctx* c) {
  float wf, hf;
  memcpy(&wf, &c->hostHeader[HG_VARS_F + HV_F_WIDTH_F], 4); memcpy(&hf, &c->hostHeader[HG_VARS_F + HV_F_HEIGHT_F], 4);
  return wf == float(c->w) && hf == float(c->h);
}
static int mmlt_frame_check(hydra_hip_ctx* c, const char* who) {
  if (c->mmlt.w != c->w || c->mmlt.h != c->h) return fail(c, HYDRA_HIP_ESTATE, std::string(who) + ": the frame was resized after mmlt_begin");
  if (!header_frame_matches(c)) return fail(c, HYDRA_HIP_ESTATE, std::string(who) + ": the globals header's HRT_WIDTH_F x HRT_HEIGHT_F is not the layer's frame");
  return HYDRA_HIP_OK;
}
int hydra_hip_mmlt_end(hydra_hip_handle c) {
  if (!c) return HYDRA_HIP_EINVAL;
  auto& m = c->mmlt;
  dev_free(m.sbDepth); dev_free(m.sbImage); m.sbSamples = 0;
  DevBuf* all[] = {&m.ch, &m.depth, &m.xCur, &m.xNew, &m.out8, &m.image, &m.accum, &m.sum, &m.scaled, &m.st, &m.rayPos[0], &m.rayPos[1], &m.rayDir[0], &m.rayDir[1], &m.rayOwner[0], &m.rayOwner[1], &m.hits, &m.counts, &m.eyePos, &m.eyeDir, &m.eyeHit, &m.shPos, &m.shDir, &m.shVis};
  for (DevBuf* b : all) dev_free(*b);
  m.active = false; m.n = 0; m.mutations = 0; m.w = 0; m.h = 0;
  return HYDRA_HIP_OK;
}
// DoPassEstimateAvgBrightness (:463-520) + the start of every chain (DoPassIndirectMLT :348-356, :371-377)
int hydra_hip_mmlt_begin(hydra_hip_handle c, int chains, int seed, int first_bounce, int max_depth, int estimate_passes) {
  int n = chains;
  STAGE_PROLOG(true);
  if (c->w <= 0 || c->h <= 0) return fail(c, HYDRA_HIP_ESTATE, "mmlt_begin: set the image size first");
  if (c->ptlMax > 0) return fail(c, HYDRA_HIP_ESTATE, "mmlt_begin: materials of the scene bind procedural textures; this layer runs them in the path tracer only (trace_pass)");
  if (c->hostHeader.size() > size_t(HG_VARS_I + HV_I_SHADOW_MATTE_BACK) && uint32_t(c->hostHeader[HG_VARS_I + HV_I_SHADOW_MATTE_BACK]) != HYDRA_INVALID_TEXTURE)
    return fail(c, HYDRA_HIP_ESTATE, "mmlt_begin: the header names a back-plate (HRT_SHADOW_MATTE_BACK); the camera-visible second environment and its shadow catchers exist in the path tracer only");
  if (!mmlt_camera_ready(c)) return fail(c, HYDRA_HIP_ESTATE, "mmlt_begin: the globals header holds no camera yet (IHWLayer::SetCamMatrices + PrepareEngineGlobals: the caller's Draw does both)");
  if (!header_frame_matches(c)) return fail(c, HYDRA_HIP_ESTATE, "mmlt_begin: the globals header's HRT_WIDTH_F x HRT_HEIGHT_F is not the layer's frame (ResizeScreen and the header must agree: splats are tested against one and written into the other)");
  const int maxD = max_depth > 0 ? max_depth : c->hostHeader[HG_VARS_I + HV_I_TRACE_DEPTH];
  int first = first_bounce > 0 ? first_bounce : c->hostHeader[HG_VARS_I + HV_I_MMLT_FIRST_BOUNCE];
  if (first > 3) first = 3;   // :481-483
  if (first < 2) first = 2;
  if (maxD < first || maxD > HK_MMLT_MAX_DEPTH) return fail(c, HYDRA_HIP_EINVAL, "mmlt_begin: max depth must lie in [first bounce, 16]");
  if (estimate_passes <= 0) estimate_passes = 4;
  hydra_hip_mmlt_end(c);
  auto& m = c->mmlt;
  m.n = n; m.maxD = maxD; m.firstBounce = first; m.w = c->w; m.h = c->h;
  const size_t N = size_t(n);
  int qseg = 1, qcap = 0;
  mmlt_queue_shape(n, qseg, qcap);
  const size_t slots = size_t(qseg) * qcap;
  struct { DevBuf* b; size_t bytes; } allocs[] = {
    {&m.ch, N * CH_PLANES * 4}, {&m.depth, N * 4}, {&m.xCur, N * mmltStride(maxD) * 4}, {&m.xNew, N * mmltStride(maxD) * 4}, {&m.out8, N * 32},
    {&m.image, size_t(c->w) * c->h * 16}, {&m.scaled, size_t(c->w) * c->h * 16}, {&m.accum, size_t(maxD + 2) * 4}, {&m.sum, 8},
    {&m.st, N * mmltPlanes(maxD) * 4}, {&m.rayPos[0], slots * 16}, {&m.rayPos[1], slots * 16}, {&m.rayDir[0], slots * 16}, {&m.rayDir[1], slots * 16},
    {&m.rayOwner[0], slots * 4}, {&m.rayOwner[1], slots * 4}, {&m.hits, slots * 16}, {&m.counts, size_t(HK_MMLT_MAX_DEPTH + 2) * HK_CROW * 4},
    {&m.eyePos, N * 16}, {&m.eyeDir, N * 16}, {&m.eyeHit, N * 16},
    {&m.shPos, N * 16}, {&m.shDir, N * 16}, {&m.shVis, N * 4}};
  for (auto& a : allocs) if ((rc = dev_alloc(c, *a.b, a.bytes))) { hydra_hip_mmlt_end(c); return rc; }
  HCHECK(hipMemsetAsync(m.image.p, 0, size_t(c->w) * c->h * 16, c->stream));
  HCHECK(hipMemsetAsync(m.xCur.p, 0, N * mmltStride(maxD) * 4, c->stream));
  HCHECK(hipMemsetAsync(m.xNew.p, 0, N * mmltStride(maxD) * 4, c->stream));
  const MmltChains ch = mmlt_chains(c);
  const MmltBufs b = mmlt_run_bufs(c);
  const SceneDev s = make_scene(c);
  const dim3 grid((n + 255) / 256), block(256);
  hipLaunchKernelGGL(k_mmlt_init_chains, grid, block, 0, c->stream, ch, seed);
  // average brightness of the paths of every length: estimate_passes x n fresh samples of F per d, each weighted by its selector's 1/pdf = d + 1
  for (int d = 0; d <= maxD + 1; d++) m.avgB[d] = 0.0f;
  for (int pass = 0; pass < estimate_passes; pass++) {
    for (int d = first; d <= maxD; d++) {
      hipLaunchKernelGGL(k_mmlt_pick_depth, grid, block, 0, c->stream, ch, (int*)m.depth.p, (const float*)nullptr, 0, d);
      hipLaunchKernelGGL(k_mmlt_fresh, grid, block, 0, c->stream, ch);
      if ((rc = mmlt_eval(c, s, b, d))) return rc;
      HCHECK(hipMemsetAsync(m.sum.p, 0, 8, c->stream));
      hipLaunchKernelGGL(k_mmlt_sum, dim3(256), dim3(256), 0, c->stream, n, (const float*)m.out8.p, 8, 7, (double*)m.sum.p);
      double sum = 0.0;
      HCHECK(hipMemcpyAsync(&sum, m.sum.p, 8, hipMemcpyDeviceToHost, c->stream));
      HCHECK(hipStreamSynchronize(c->stream));
      m.avgB[d] += float(sum * double(d + 1));
    }
  }
  m.avgBrightness = 0.0f;
  std::vector<float> accum(size_t(maxD) + 2, 0.0f);   // PrefixSumm(m_avgBPerBounce), :317-328
  float acc = 0.0f;
  for (int d = 0; d <= maxD; d++) {
    m.avgB[d] *= 1.0f / float(size_t(estimate_passes) * N);
    m.avgBrightness += m.avgB[d];
    accum[d] = acc;
    acc += m.avgB[d];
  }
  accum[size_t(maxD) + 1] = acc;
  if (!(m.avgBrightness > 0.0f)) {
    std::string per = "";
    for (int d = first; d <= maxD; d++) per += (d > first ? ", " : "") + std::to_string(m.avgB[d]);
    hydra_hip_mmlt_end(c);
    return fail(c, HYDRA_HIP_ESTATE, "mmlt_begin: no light reaches the camera on paths of the requested lengths (average brightness per length: " + per + ")");
  }
  HCHECK(hipMemcpyAsync(m.accum.p, accum.data(), accum.size() * 4, hipMemcpyHostToDevice, c->stream));
  // every chain: its path length, a fresh sample, F of it
  hipLaunchKernelGGL(k_mmlt_pick_depth, grid, block, 0, c->stream, ch, (int*)m.depth.p, (const float*)m.accum.p, maxD + 2, 0);
  hipLaunchKernelGGL(k_mmlt_fresh, grid, block, 0, c->stream, ch);
  if ((rc = mmlt_eval(c, s, b, maxD))) return rc;
  hipLaunchKernelGGL(k_mmlt_seed, grid, block, 0, c->stream, ch, (const float*)m.out8.p);
  STAGE_EPILOG();
  m.active = true; m.mutations = 0;
  return HYDRA_HIP_OK;
}
// `mutations` steps of every chain: propose, F, accept / reject, two contributions to the indirect image (DoPassIndirectMLT :379-447)
int hydra_hip_mmlt_pass(hydra_hip_handle c, int mutations) {
  if (!c || mutations <= 0) return HYDRA_HIP_EINVAL;
  if (!c->mmlt.active) return fail(c, HYDRA_HIP_ESTATE, "mmlt_pass: call mmlt_begin first");
  { const int frc = mmlt_frame_check(c, "mmlt_pass"); if (frc) return frc; }
  HCHECK(hipSetDevice(c->device));
  auto& m = c->mmlt;
  const MmltChains ch = mmlt_chains(c);
  const MmltBufs b = mmlt_run_bufs(c);
  const SceneDev s = make_scene(c);
  const dim3 grid((m.n + 255) / 256), block(256);
  for (int k = 0; k < mutations; k++) {
    hipLaunchKernelGGL(k_mmlt_mutate, grid, block, 0, c->stream, ch);
    const int rc = mmlt_eval(c, s, b, m.maxD);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mmlt_accept, grid, block, 0, c->stream, ch, (const float*)m.out8.p, 1.0f, (float*)m.image.p, c->w);
  }
  HCHECK(hipGetLastError());
  m.mutations += (unsigned long long)mutations * (unsigned long long)m.n;
  return HYDRA_HIP_OK;
}
// image4 = kScale x the indirect image (GetImageHDR :616-635 without the direct part; kScale = EstimateScaleCoeff :548-552);
// info8 = average brightness, kScale, acceptance rate, mutations so far, chains, first bounce, max depth, 0
int hydra_hip_mmlt_get_image(hydra_hip_handle c, float* image4, int width, int height, float* info8) {
  if (!c) return HYDRA_HIP_EINVAL;
  if (!c->mmlt.active) return fail(c, HYDRA_HIP_ESTATE, "mmlt_get_image: call mmlt_begin first");
  if (image4 && (width != c->mmlt.w || height != c->mmlt.h || width != c->w || height != c->h)) return fail(c, HYDRA_HIP_EINVAL, "mmlt_get_image: bad input resolution");
  HCHECK(hipSetDevice(c->device));
  auto& m = c->mmlt;
  const int npix = c->w * c->h;
  double sums[3] = {0, 0, 0};
  for (int k = 0; k < 3; k++) {
    HCHECK(hipMemsetAsync(m.sum.p, 0, 8, c->stream));
    hipLaunchKernelGGL(k_mmlt_sum, dim3(256), dim3(256), 0, c->stream, npix, (const float*)m.image.p, 4, k, (double*)m.sum.p);
    HCHECK(hipMemcpyAsync(&sums[k], m.sum.p, 8, hipMemcpyDeviceToHost, c->stream));
    HCHECK(hipStreamSynchronize(c->stream));
  }
  float avgB = fmaxf(0.33334f * float((sums[0] + sums[1] + sums[2]) / double(npix)), 0.0f);   // EstimateAverageBrightness2, :26-33
  if (avgB < 1e-20f) avgB = 1e-20f;
  const float kScale = m.avgBrightness / avgB;
  double accepted = 0.0;
  HCHECK(hipMemsetAsync(m.sum.p, 0, 8, c->stream));
  hipLaunchKernelGGL(k_mmlt_sum, dim3(256), dim3(256), 0, c->stream, m.n, (const float*)m.ch.p + size_t(CH_ACCEPTED) * m.n, 1, 0, (double*)m.sum.p);
  HCHECK(hipMemcpyAsync(&accepted, m.sum.p, 8, hipMemcpyDeviceToHost, c->stream));
  if (image4) {
    hipLaunchKernelGGL(k_mmlt_scale_image, dim3((npix + 255) / 256), dim3(256), 0, c->stream, npix, (const float4*)m.image.p, kScale, (float4*)m.scaled.p);
    HCHECK(hipMemcpyAsync(image4, m.scaled.p, size_t(npix) * 16, hipMemcpyDeviceToHost, c->stream));
  }
  HCHECK(hipStreamSynchronize(c->stream));
  if (info8) {
    info8[0] = m.avgBrightness; info8[1] = kScale; info8[2] = m.mutations ? float(accepted / double(m.mutations)) : 0.0f; info8[3] = float(m.mutations);
    info8[4] = float(m.n); info8[5] = float(m.firstBounce); info8[6] = float(m.maxD); info8[7] = 0.0f;
  }
  return HYDRA_HIP_OK;
}
// the indirect image restarts from zero, the chains go on: what a contribution to a shared accumulation image needs (the reference's
// ClearAccumulatedColor zeroes the image its MMLT pass splats into and leaves the chain state alone, GPUOCLLayer.cpp:1288-1297)
int hydra_hip_mmlt_reset_image(hydra_hip_handle c) {
  if (!c) return HYDRA_HIP_EINVAL;
  if (!c->mmlt.active) return fail(c, HYDRA_HIP_ESTATE, "mmlt_reset_image: call mmlt_begin first");
  HCHECK(hipSetDevice(c->device));
  HCHECK(hipMemsetAsync(c->mmlt.image.p, 0, size_t(c->mmlt.w) * c->mmlt.h * 16, c->stream));
  return HYDRA_HIP_OK;
}
// IntegratorSBDPT::DoPass (CPUExp_Integrators_SBDPT.cpp:11-216) on the buffers of the MMLT run: `passes` x chains samples, each a path
// length drawn uniformly from 2..max_depth, a fresh primary-sample vector, F, a splat weighted by (d + 1)(max_depth - 1)
int hydra_hip_sbdpt_pass(hydra_hip_handle c, int passes) {
  if (!c || passes <= 0) return HYDRA_HIP_EINVAL;
  if (!c->mmlt.active) return fail(c, HYDRA_HIP_ESTATE, "sbdpt_pass: call mmlt_begin first (it owns the buffers and the generators)");
  { const int frc = mmlt_frame_check(c, "sbdpt_pass"); if (frc) return frc; }
  HCHECK(hipSetDevice(c->device));
  auto& m = c->mmlt;
  int rc;
  if (m.sbImage.p == nullptr) {
    if ((rc = dev_alloc(c, m.sbDepth, size_t(m.n) * 4)) || (rc = dev_alloc(c, m.sbImage, size_t(c->w) * c->h * 16))) return rc;
    HCHECK(hipMemsetAsync(m.sbImage.p, 0, size_t(c->w) * c->h * 16, c->stream));
    m.sbSamples = 0;
  }
  MmltChains ch = mmlt_chains(c);
  ch.depth = (const int*)m.sbDepth.p;            // fresh samples are as long as this pass's d, not the chain's
  MmltBufs b = mmlt_run_bufs(c);
  b.v.depth = (const int*)m.sbDepth.p;
  const SceneDev s = make_scene(c);
  const dim3 grid((m.n + 255) / 256), block(256);
  for (int k = 0; k < passes; k++) {
    hipLaunchKernelGGL(k_sbdpt_pick_depth, grid, block, 0, c->stream, ch, (int*)m.sbDepth.p, m.maxD);
    hipLaunchKernelGGL(k_mmlt_fresh, grid, block, 0, c->stream, ch);
    if ((rc = mmlt_eval(c, s, b, m.maxD))) return rc;
    hipLaunchKernelGGL(k_sbdpt_splat, grid, block, 0, c->stream, m.n, (const int*)m.sbDepth.p, m.maxD, (const float*)m.out8.p, (float*)m.sbImage.p, c->w);
  }
  HCHECK(hipGetLastError());
  m.sbSamples += (unsigned long long)passes * (unsigned long long)m.n;
  return HYDRA_HIP_OK;
}
// image4 = splats x width*height / samples (the reference's m_hdrData / spp with width*height samples per pass, :18-19, :192); samples so far
int hydra_hip_sbdpt_get_image(hydra_hip_handle c, float* image4, int width, int height, double* samples) {
  if (!c) return HYDRA_HIP_EINVAL;
  if (!c->mmlt.active || c->mmlt.sbImage.p == nullptr) return fail(c, HYDRA_HIP_ESTATE, "sbdpt_get_image: no SBDPT pass has run");
  if (image4 && (width != c->mmlt.w || height != c->mmlt.h || width != c->w || height != c->h)) return fail(c, HYDRA_HIP_EINVAL, "sbdpt_get_image: bad input resolution");
  HCHECK(hipSetDevice(c->device));
  auto& m = c->mmlt;
  const int npix = c->w * c->h;
  if (image4) {
    const float scale = m.sbSamples ? float(double(npix) / double(m.sbSamples)) : 0.0f;
    hipLaunchKernelGGL(k_mmlt_scale_image, dim3((npix + 255) / 256), dim3(256), 0, c->stream, npix, (const float4*)m.sbImage.p, scale, (float4*)m.scaled.p);
    HCHECK(hipMemcpyAsync(image4, m.scaled.p, size_t(npix) * 16, hipMemcpyDeviceToHost, c->stream));
  }
  HCHECK(hipStreamSynchronize(c->stream));
  if (samples) *samples = double(m.sbSamples);
  return HYDRA_HIP_OK;
}
// ---------------------------------------------------------------------------------------- G-buffer (IHWLayer::EvalGBuffer)
__global__ void k_gbuffer_rays(SceneDev s, int w, int pix0, int nPix, float4* __restrict__ pos4, float4* __restrict__ dir4) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nPix * HK_GBUFFER_SAMPLES) return;
  const int pixel = pix0 + r / HK_GBUFFER_SAMPLES, k = r % HK_GBUFFER_SAMPLES;
  f3 pos, dir;
  gbufferEyeRay(s, pixel % w, pixel / w, k, w, pos, dir);
  pos4[r] = mk4(pos, 0.0f);
  dir4[r] = mk4(dir, 0.0f);
}
// one wavefront per pixel, lane = sample: every lane compares its sample with the 64 of the wave (v_readlane broadcasts), in the order
// of the reference's inner loop so that the float sums are the same; the first lane with the smallest sum writes the pixel
__global__ void __launch_bounds__(256) k_gbuffer_resolve(SceneDev s, int w, int h, int pix0, int nPix, const float4* __restrict__ pos4, const float4* __restrict__ dir4,
                                                         const HydraLiteHit* __restrict__ hits, const int* __restrict__ remap, int remapSize,
                                                         float4* __restrict__ out1, float4* __restrict__ out2, float* __restrict__ raw14) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = int(__lane_id());
  if (wave >= nPix) return;   // wave-uniform
  const int r = wave * HK_GBUFFER_SAMPLES + lane;
  s.ptlSlot = (s.ptlIds != nullptr && uint32_t(s.ptlIds[r]) != HYDRA_INVALID_TEXTURE) ? r : -1;   // the ray's procedural texture list (eval_gbuffer runs the scene's program first)
  HydraLiteHit hit = hits[r];
  if (hit.primId != -1) hit.geomId = HK_GEOM_ID(hit.geomId);   // the device's class label is not part of Lite_Hit
  GBufferSample g = gbufferSampleOf(s, xyz(pos4[r]), xyz(dir4[r]), hit);
  const GBufferKey mine = gbufferKey(g);
  float diff = 0.0f, coverage = 0.0f;
  for (int j = 0; j < HK_GBUFFER_SAMPLES; j++) {
    GBufferKey o;
    o.norm.x = as_float(__builtin_amdgcn_readlane(as_int(mine.norm.x), j)); o.norm.y = as_float(__builtin_amdgcn_readlane(as_int(mine.norm.y), j));
    o.norm.z = as_float(__builtin_amdgcn_readlane(as_int(mine.norm.z), j)); o.depth = as_float(__builtin_amdgcn_readlane(as_int(mine.depth), j));
    o.instId = __builtin_amdgcn_readlane(mine.instId, j); o.objId = __builtin_amdgcn_readlane(mine.objId, j);
    o.matId = __builtin_amdgcn_readlane(mine.matId, j); o.alpha = as_float(__builtin_amdgcn_readlane(as_int(mine.alpha), j));
    const float thisDiff = gbuffDiff(mine, o, HK_GBUFFER_FOV, float(w), float(h));
    diff += thisDiff;
    if (thisDiff < 1.0f) coverage += 1.0f;
  }
  g.coverage = coverage * (1.0f / float(HK_GBUFFER_SAMPLES));
  const bool cand = diff < 100000000.0f;          // `diff < minDiff` against the initial minimum (NaN never wins)
  float m = cand ? diff : 3.0e38f;
  for (int d = 32; d >= 1; d >>= 1) m = fminf(m, __shfl_xor(m, d));
  const unsigned long long winners = __ballot(cand && diff == m);
  const int winner = winners != 0ull ? __ffsll((long long)winners) - 1 : 0;
  if (lane == winner) {
    const int pixel = pix0 + wave;
    if (remap != nullptr && g.instId >= 0 && g.instId < remapSize) g.instId = remap[g.instId];   // a_instIdByInstId, GPUOCLLayerOther.cpp:846-853
    out1[pixel] = packGBuffer1(g);
    out2[pixel] = packGBuffer2(g);
    if (raw14) {
      float* o = raw14 + size_t(pixel) * 14;
      o[0] = g.depth; o[1] = g.norm.x; o[2] = g.norm.y; o[3] = g.norm.z; o[4] = g.rgba.x; o[5] = g.rgba.y; o[6] = g.rgba.z; o[7] = g.rgba.w;
      o[8] = as_float(g.matId); o[9] = g.coverage; o[10] = g.texCoord.x; o[11] = g.texCoord.y; o[12] = as_float(g.objId); o[13] = as_float(g.instId);
    }
  }
}
int hydra_hip_eval_gbuffer(hydra_hip_handle c, float* data1, float* data2, int width, int height, const int32_t* inst_remap, int inst_remap_size, float* raw14) {
  if (!c || !data1 || !data2 || inst_remap_size < 0) return HYDRA_HIP_EINVAL;
  if (width != c->w || height != c->h) return fail(c, HYDRA_HIP_EINVAL, "eval_gbuffer: bad input resolution");
  if (!scene_ready(c)) return fail(c, HYDRA_HIP_ESTATE, "eval_gbuffer: scene is not uploaded");
  if (!mmlt_camera_ready(c)) return fail(c, HYDRA_HIP_ESTATE, "eval_gbuffer: the globals header holds no camera yet (SetCamMatrices + PrepareEngineGlobals)");
  if (!header_frame_matches(c)) return fail(c, HYDRA_HIP_ESTATE, "eval_gbuffer: the globals header's HRT_WIDTH_F x HRT_HEIGHT_F is not the layer's frame");
  HCHECK(hipSetDevice(c->device));
  { const int prc = prepare_geometry(c); if (prc) return prc; }
  { const int vrc = validate_materials(c); if (vrc) return vrc; }
  if (c->ptlMax > 0 && c->procTex == nullptr) return fail(c, HYDRA_HIP_ESTATE, "eval_gbuffer: materials of the scene bind procedural textures, but no program was compiled for them (hydra_hip_proctex_compile)");
  { const int crc = prepare_classes(c); if (crc) return crc; }
  int rc = HYDRA_HIP_OK;
  if ((rc = ensure_fetch_counters(c))) return rc;
  const int npix = c->w * c->h;
  // rays in blocks of pixels (GPUOCLLayer works in MEGABLOCKSIZE lines the same way, :743-757): through the path tracer's own ray and hit
  // arrays when a render state is allocated (no path is alive between passes), else through 4 M-ray temporaries
  TmpBufs tb;
  size_t cap = std::min(std::min(c->sPos.bytes, c->sDir.bytes), c->hits.bytes) / 16 / HK_GBUFFER_SAMPLES;
  float4* dpos = static_cast<float4*>(c->sPos.p), *ddir = static_cast<float4*>(c->sDir.p);
  HydraLiteHit* dh = static_cast<HydraLiteHit*>(c->hits.p);
  if (cap > size_t(1) << 24) cap = size_t(1) << 24;   // n = pixels x 64 stays below 2^31
  if (cap < 65536 && cap < size_t(npix)) {
    cap = npix < 65536 ? size_t(npix) : 65536;
    dpos = (float4*)tb.up(c, nullptr, cap * HK_GBUFFER_SAMPLES * 16, rc);
    ddir = (float4*)tb.up(c, nullptr, cap * HK_GBUFFER_SAMPLES * 16, rc);
    dh = (HydraLiteHit*)tb.up(c, nullptr, cap * HK_GBUFFER_SAMPLES * 16, rc);
  }
  const int pixPerBlock = size_t(npix) < cap ? npix : int(cap);
  float4* d1 = (float4*)tb.up(c, nullptr, size_t(npix) * 16, rc);
  float4* d2 = (float4*)tb.up(c, nullptr, size_t(npix) * 16, rc);
  float* draw = raw14 ? (float*)tb.up(c, nullptr, size_t(npix) * 14 * 4, rc) : nullptr;
  int* dremap = (inst_remap && inst_remap_size > 0) ? (int*)tb.up(c, inst_remap, size_t(inst_remap_size) * 4, rc) : nullptr;
  // procedural textures: the scene's program on the hits of every block of rays, lists by ray index (GetGBufferSample reads the same per-ray lists, material.cl:1347)
  int* dptlIds = nullptr; uint2* dptlVals = nullptr;
  if (c->ptlMax > 0) {
    dptlIds = (int*)tb.up(c, nullptr, size_t(pixPerBlock) * HK_GBUFFER_SAMPLES * size_t(c->ptlMax) * 4, rc);
    dptlVals = (uint2*)tb.up(c, nullptr, size_t(pixPerBlock) * HK_GBUFFER_SAMPLES * size_t(c->ptlMax) * 8, rc);
  }
  if (rc) return rc;
  const SceneDev s = make_scene(c);
  // the rays go through the traversal kernel as a 32-segment queue like the path tracer's: its persistent waves take their rays through one
  // counter per segment, and a single counter saturates at ~88 fetches per microsecond (5.6 G rays/s; measured here before the split)
  uint32_t* fetch = static_cast<uint32_t*>(c->fetchCnt.p) + size_t(2 * HK_MAX_DEPTH + 2) * HK_CROW;
  uint32_t* segCounts = fetch + HK_CROW;
  const int nsegG = 32;
  hipEvent_t ev0, ev1;
  HCHECK(hipEventCreate(&ev0)); HCHECK(hipEventCreate(&ev1));
  HCHECK(hipEventRecord(ev0, c->stream));
  for (int pix0 = 0; pix0 < npix; pix0 += pixPerBlock) {
    const int nPix = (npix - pix0 < pixPerBlock) ? npix - pix0 : pixPerBlock;
    const int n = nPix * HK_GBUFFER_SAMPLES;
    hipLaunchKernelGGL(k_gbuffer_rays, dim3((n + 255) / 256), dim3(256), 0, c->stream, s, c->w, pix0, nPix, dpos, ddir);
    const int capG = ((n + nsegG - 1) / nsegG + 63) / 64 * 64;
    HCHECK(hipMemsetAsync(fetch, 0, size_t(HK_CROW) * 4, c->stream));
    hipLaunchKernelGGL(k_fill_seg_counts, dim3(1), dim3(64), 0, c->stream, n, capG, nsegG, segCounts);
    launch_closest(c, s, seg_q(segCounts, 0, nsegG, capG), dpos, ddir, dh, nullptr, nullptr, fetch);
    SceneDev sr = s;
    if (c->ptlMax > 0) {
      HCHECK(hk_proctex_launch_points(c->procTex, c->stream, s, n, dpos, ddir, dh, dptlIds, dptlVals, n, c->ptlMax));
      sr.ptlIds = dptlIds; sr.ptlVals = dptlVals; sr.ptlStride = n; sr.ptlMax = c->ptlMax;
    }
    hipLaunchKernelGGL(k_gbuffer_resolve, dim3((n + 255) / 256), dim3(256), 0, c->stream, sr, c->w, c->h, pix0, nPix, dpos, ddir, dh, dremap, inst_remap_size, d1, d2, draw);
  }
  HCHECK(hipGetLastError());
  HCHECK(hipEventRecord(ev1, c->stream));
  HCHECK(hipMemcpyAsync(data1, d1, size_t(npix) * 16, hipMemcpyDeviceToHost, c->stream));
  HCHECK(hipMemcpyAsync(data2, d2, size_t(npix) * 16, hipMemcpyDeviceToHost, c->stream));
  if (raw14) HCHECK(hipMemcpyAsync(raw14, draw, size_t(npix) * 14 * 4, hipMemcpyDeviceToHost, c->stream));
  HCHECK(hipStreamSynchronize(c->stream));
  { float ms = 0.0f; if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) c->lastGbufferUs = int(ms * 1000.0f); }   // device time of the rays, the traversal and the vote (option "last_gbuffer_device_us")
  (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
  return HYDRA_HIP_OK;
}

// test hook: chain planes (CH_PLANES x n), path lengths (n) and current x vectors (n rows of 12 + 10 * maxD), average brightness per length (maxD + 1)
int hydra_hip_mmlt_get_state(hydra_hip_handle c, float* chains, int32_t* depth, float* xrows, float* avg_b) {
  if (!c) return HYDRA_HIP_EINVAL;
  if (!c->mmlt.active) return fail(c, HYDRA_HIP_ESTATE, "mmlt_get_state: call mmlt_begin first");
  HCHECK(hipSetDevice(c->device));
  auto& m = c->mmlt;
  HCHECK(hipStreamSynchronize(c->stream));
  if (chains) HCHECK(hipMemcpy(chains, m.ch.p, size_t(m.n) * CH_PLANES * 4, hipMemcpyDeviceToHost));
  if (depth) HCHECK(hipMemcpy(depth, m.depth.p, size_t(m.n) * 4, hipMemcpyDeviceToHost));
  if (xrows) {
    const int stride = mmltStride(m.maxD);
    std::vector<float> planes(size_t(m.n) * stride);
    HCHECK(hipMemcpy(planes.data(), m.xCur.p, planes.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < m.n; i++) for (int j = 0; j < stride; j++) xrows[size_t(i) * stride + j] = planes[size_t(j) * m.n + i];
  }
  if (avg_b) for (int d = 0; d <= m.maxD; d++) avg_b[d] = m.avgB[d];
  return HYDRA_HIP_OK;
}

int hydra_hip_stage_mmlt_f(hydra_hip_handle c, int n, const int32_t* depth, const float* xvec, int stride, float* out8) {
  STAGE_PROLOG(true);
  if (!depth || !xvec || !out8) return fail(c, HYDRA_HIP_EINVAL, "stage_mmlt_f: null argument");
  if (!mmlt_camera_ready(c)) return fail(c, HYDRA_HIP_ESTATE, "stage_mmlt_f: the globals header holds no camera yet (SetCamMatrices + PrepareEngineGlobals)");
  int maxD = 0;
  for (int i = 0; i < n; i++) {
    if (depth[i] < 1 || depth[i] > HK_MMLT_MAX_DEPTH || stride < HK_MMLT_HEAD + HK_MMLT_PER_BOUNCE * depth[i]) return fail(c, HYDRA_HIP_EINVAL, "stage_mmlt_f: depth must be 1..16 and stride >= 12 + 10 * depth");
    maxD = depth[i] > maxD ? depth[i] : maxD;
  }
  MmltBufs b;
  if ((rc = mmlt_alloc(c, tb, n, maxD, b))) return rc;
  float* drows = (float*)tb.up(c, xvec, size_t(n) * stride * 4, rc);
  float* dx = (float*)tb.up(c, nullptr, size_t(mmltStride(maxD)) * n * 4, rc);
  int* dd = (int*)tb.up(c, depth, size_t(n) * 4, rc);
  float* dout = (float*)tb.up(c, nullptr, size_t(n) * 32, rc);
  if (rc) return rc;
  HCHECK(hipMemsetAsync(dx, 0, size_t(mmltStride(maxD)) * n * 4, c->stream));
  const int floats = (stride < mmltStride(maxD)) ? stride : mmltStride(maxD);
  hipLaunchKernelGGL(k_mmlt_transpose_in, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, stride, floats, drows, dx);
  b.v.x = dx; b.v.depth = dd; b.v.out8 = dout;
  if ((rc = mmlt_eval(c, make_scene(c), b, maxD))) return rc;
  STAGE_EPILOG();
  HCHECK(hipMemcpy(out8, dout, size_t(n) * 32, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

// f3 building blocks: LightSampleForward, lightPdfFwd, camera connection, Kelemen mutation
int hydra_hip_stage_light_sample_forward(hydra_hip_handle c, int n, const int32_t* light_ids, const float* rands4, float* out16) {
  if (!c || n <= 0 || !light_ids || !rands4 || !out16 || !c->globals.p) return fail(c, HYDRA_HIP_ESTATE, "stage_light_sample_forward: globals are not uploaded / null argument");
  HCHECK(hipSetDevice(c->device));
  const int lightsNum = c->hostHeader[HG_LIGHTS_NUM];
  for (int i = 0; i < n; i++) if (light_ids[i] < 0 || light_ids[i] >= lightsNum) return fail(c, HYDRA_HIP_EINVAL, "stage_light_sample_forward: light id out of range");
  TmpBufs tb; int rc = HYDRA_HIP_OK;
  int* dl = (int*)tb.up(c, light_ids, size_t(n) * 4, rc);
  float4* dr = (float4*)tb.up(c, rands4, size_t(n) * 16, rc);
  float* dout = (float*)tb.up(c, nullptr, size_t(n) * 64, rc);
  if (rc) return rc;
  hipLaunchKernelGGL(k_stage_light_fwd, dim3((n + 255) / 256), dim3(256), 0, c->stream, make_scene(c), n, dl, dr, dout);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(out16, dout, size_t(n) * 64, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}
int hydra_hip_stage_light_pdf_fwd(hydra_hip_handle c, int n, const int32_t* light_ids, const float* cos_theta, float* out4) {
  if (!c || n <= 0 || !light_ids || !cos_theta || !out4 || !c->globals.p) return fail(c, HYDRA_HIP_ESTATE, "stage_light_pdf_fwd: globals are not uploaded / null argument");
  HCHECK(hipSetDevice(c->device));
  const int lightsNum = c->hostHeader[HG_LIGHTS_NUM];
  for (int i = 0; i < n; i++) if (light_ids[i] < 0 || light_ids[i] >= lightsNum) return fail(c, HYDRA_HIP_EINVAL, "stage_light_pdf_fwd: light id out of range");
  TmpBufs tb; int rc = HYDRA_HIP_OK;
  int* dl = (int*)tb.up(c, light_ids, size_t(n) * 4, rc);
  float* dc = (float*)tb.up(c, cos_theta, size_t(n) * 4, rc);
  float4* dout = (float4*)tb.up(c, nullptr, size_t(n) * 16, rc);
  if (rc) return rc;
  hipLaunchKernelGGL(k_stage_light_pdf_fwd, dim3((n + 255) / 256), dim3(256), 0, c->stream, make_scene(c), n, dl, dc, dout);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(out4, dout, size_t(n) * 16, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}
int hydra_hip_stage_camera_connect(hydra_hip_handle c, int n, const float* pos4, const float* norm4, const float* disk2, float* out8) {
  if (!c || n <= 0 || !pos4 || !norm4 || !disk2 || !out8 || !c->globals.p) return fail(c, HYDRA_HIP_ESTATE, "stage_camera_connect: globals are not uploaded / null argument");
  HCHECK(hipSetDevice(c->device));
  TmpBufs tb; int rc = HYDRA_HIP_OK;
  float4* dp = (float4*)tb.up(c, pos4, size_t(n) * 16, rc);
  float4* dn = (float4*)tb.up(c, norm4, size_t(n) * 16, rc);
  float2* dd = (float2*)tb.up(c, disk2, size_t(n) * 8, rc);
  float* dout = (float*)tb.up(c, nullptr, size_t(n) * 32, rc);
  if (rc) return rc;
  hipLaunchKernelGGL(k_stage_camera_connect, dim3((n + 255) / 256), dim3(256), 0, c->stream, make_scene(c), n, dp, dn, dd, dout);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(out8, dout, size_t(n) * 32, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}
int hydra_hip_stage_mutate_kelemen(hydra_hip_handle c, int n, const float* values, const float* rands2, float p2, float p1, float* out) {
  if (!c || n <= 0 || !values || !rands2 || !out) return HYDRA_HIP_EINVAL;
  HCHECK(hipSetDevice(c->device));
  TmpBufs tb; int rc = HYDRA_HIP_OK;
  float* dv = (float*)tb.up(c, values, size_t(n) * 4, rc);
  float2* dr = (float2*)tb.up(c, rands2, size_t(n) * 8, rc);
  float* dout = (float*)tb.up(c, nullptr, size_t(n) * 4, rc);
  if (rc) return rc;
  hipLaunchKernelGGL(k_stage_mutate_kelemen, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dv, dr, p2, p1, dout);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(out, dout, size_t(n) * 4, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

int hydra_hip_stage_random(hydra_hip_handle c, int n, const int32_t* seeds, int draws, float* out4, uint32_t* state2) {
  if (!c || n <= 0 || draws <= 0) return HYDRA_HIP_EINVAL;
  HCHECK(hipSetDevice(c->device));
  TmpBufs tb; int rc = HYDRA_HIP_OK;
  int* ds = (int*)tb.up(c, seeds, size_t(n) * 4, rc);
  float4* dout = (float4*)tb.up(c, nullptr, size_t(n) * draws * 16, rc);
  uint2* dst = (uint2*)tb.up(c, nullptr, size_t(n) * 8, rc);
  if (rc) return rc;
  hipLaunchKernelGGL(k_stage_random, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, ds, draws, dout, dst);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(out4, dout, size_t(n) * draws * 16, hipMemcpyDeviceToHost));
  HCHECK(hipMemcpy(state2, dst, size_t(n) * 8, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

int hydra_hip_bench_trace(hydra_hip_handle c, int n, const float* ray_pos4, const float* ray_dir4, int iters, int shadow, float* avg_ms) {
  STAGE_PROLOG(true);
  if (iters < 1 || !avg_ms) return HYDRA_HIP_EINVAL;
  float4* dpos = (float4*)tb.up(c, ray_pos4, size_t(n) * 16, rc);   // for shadow: w of pos = t_far
  float4* ddir = (float4*)tb.up(c, ray_dir4, size_t(n) * 16, rc);
  HydraLiteHit* dh = (HydraLiteHit*)tb.up(c, nullptr, size_t(n) * 16, rc);
  if (rc) return rc;
  SceneDev s = make_scene(c);
  if ((rc = ensure_fetch_counters(c))) return rc;
  uint32_t* fetch = static_cast<uint32_t*>(c->fetchCnt.p) + size_t(2 * HK_MAX_DEPTH + 2) * HK_CROW;
  hipEvent_t e0, e1;
  HCHECK(hipEventCreate(&e0));
  HCHECK(hipEventCreate(&e1));
  auto launch = [&]() {
    (void)hipMemsetAsync(fetch, 0, 4, c->stream);
    if (shadow) launch_shadow(c, s, seg_q(nullptr, n, 1, n), dpos, ddir, reinterpret_cast<float*>(dh), nullptr, fetch);
    else launch_closest(c, s, seg_q(nullptr, n, 1, n), dpos, ddir, dh, nullptr, nullptr, fetch);
  };
  launch();   // warm-up
  HCHECK(hipEventRecord(e0, c->stream));
  for (int i = 0; i < iters; i++) launch();
  HCHECK(hipEventRecord(e1, c->stream));
  HCHECK(hipEventSynchronize(e1));
  float ms = 0.0f;
  HCHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  HCHECK(hipGetLastError());
  *avg_ms = ms / float(iters);
  return HYDRA_HIP_OK;
}

// the persistent COUNTING kernels (the ones bench.py prices its roofline bytes with) on caller-provided rays
int hydra_hip_stage_trace_totals(hydra_hip_handle c, int n, const float* ray_pos4, const float* ray_dir4, const float* t_far, uint64_t* totals6) {
  STAGE_PROLOG(true);
  if (!ray_pos4 || !ray_dir4 || !totals6) return fail(c, HYDRA_HIP_EINVAL, "stage_trace_totals: null argument");
  if (c->traceMode == 0) return fail(c, HYDRA_HIP_ESTATE, "stage_trace_totals: needs trace_mode 1 (the persistent kernels)");
  std::vector<float> org(ray_pos4, ray_pos4 + size_t(n) * 4);
  if (t_far) for (int i = 0; i < n; i++) org[4 * size_t(i) + 3] = t_far[i];
  float4* dpos = (float4*)tb.up(c, org.data(), size_t(n) * 16, rc);
  float4* ddir = (float4*)tb.up(c, ray_dir4, size_t(n) * 16, rc);
  HydraLiteHit* dh = (HydraLiteHit*)tb.up(c, nullptr, size_t(n) * 16, rc);
  unsigned long long* dt = (unsigned long long*)tb.up(c, nullptr, 6 * 8, rc);
  if (rc) return rc;
  HCHECK(hipMemsetAsync(dt, 0, 6 * 8, c->stream));
  SceneDev s = make_scene(c);
  if ((rc = ensure_fetch_counters(c))) return rc;
  uint32_t* fetch = static_cast<uint32_t*>(c->fetchCnt.p) + size_t(2 * HK_MAX_DEPTH + 2) * HK_CROW;
  HCHECK(hipMemsetAsync(fetch, 0, 4, c->stream));
  if (t_far) launch_shadow(c, s, seg_q(nullptr, n, 1, n), dpos, ddir, reinterpret_cast<float*>(dh), dt, fetch);
  else launch_closest(c, s, seg_q(nullptr, n, 1, n), dpos, ddir, dh, nullptr, dt, fetch);
  STAGE_EPILOG();
  HCHECK(hipMemcpy(totals6, dt, 6 * 8, hipMemcpyDeviceToHost));
  return HYDRA_HIP_OK;
}

// ------------------------------------------------------------------------------------------------ multi-GPU exchange (RCCL)
// tiles of the image plane are partitioned over the GPUs of one node and the only exchange per frame is the float4
// accumulator.  These entry points do that exchange for a host without Python or torch (a RenderDriverRTE-style C++ process per
// GPU; precedent for "every process adds its frame into one image": hydra_drv/GPUOCLLayerOther.cpp:365-429): RCCL is dlopen()ed,
// the communicator lives in the context, the collective runs on the context's stream behind the frame's kernels.
static bool load_rccl(hydra_hip_ctx* c) {
  if (c->rccl.lib) return true;
  void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);   // an already loaded RCCL (e.g. torch's) is reused
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) { c->err = std::string("hydra_hip_comm: cannot load librccl.so: ") + dlerror(); return false; }
  auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) c->err = std::string("hydra_hip_comm: librccl.so lacks ") + n; return p; };
#define HK_RCCL_SYM(field, name) c->rccl.field = reinterpret_cast<decltype(c->rccl.field)>(sym(name)); if (!c->rccl.field) return false;
  HK_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") HK_RCCL_SYM(CommInitRank, "ncclCommInitRank") HK_RCCL_SYM(CommDestroy, "ncclCommDestroy")
  HK_RCCL_SYM(Reduce, "ncclReduce") HK_RCCL_SYM(Send, "ncclSend") HK_RCCL_SYM(Recv, "ncclRecv") HK_RCCL_SYM(GroupStart, "ncclGroupStart")
  HK_RCCL_SYM(GroupEnd, "ncclGroupEnd") HK_RCCL_SYM(GetErrorString, "ncclGetErrorString") HK_RCCL_SYM(AllGather, "ncclAllGather")
#undef HK_RCCL_SYM
  c->rccl.lib = lib;
  return true;
}
#define NCHECK(call)                                                                                   \
  do {                                                                                                 \
    ncclResult_t r_ = (call);                                                                          \
    if (r_ != ncclSuccess) { c->err = std::string(#call) + ": " + c->rccl.GetErrorString(r_); return HYDRA_HIP_EDEVICE; } \
  } while (0)

int hydra_hip_comm_unique_id(hydra_hip_handle c, char* id128) {
  if (!c || !id128) return HYDRA_HIP_EINVAL;
  if (!load_rccl(c)) return HYDRA_HIP_EDEVICE;
  ncclUniqueId id;
  NCHECK(c->rccl.GetUniqueId(&id));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, 128);
  return HYDRA_HIP_OK;
}
int hydra_hip_comm_init(hydra_hip_handle c, const char* id128, int rank, int world) {
  if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(c, HYDRA_HIP_EINVAL, "comm_init: bad arguments");
  if (rank != c->rank || world != c->world) return fail(c, HYDRA_HIP_ESTATE, "comm_init: rank / world differ from set_tile_partition");
  if (c->comm) return fail(c, HYDRA_HIP_ESTATE, "comm_init: already initialised (comm_destroy first)");
  HCHECK(hipSetDevice(c->device));
  if (!load_rccl(c)) return HYDRA_HIP_EDEVICE;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  NCHECK(c->rccl.CommInitRank(&c->comm, world, id, rank));
  c->commRank = rank; c->commWorld = world;
  c->commCount.clear();
  return HYDRA_HIP_OK;
}
int hydra_hip_comm_destroy(hydra_hip_handle c) {
  if (!c) return HYDRA_HIP_EINVAL;
  if (c->comm) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); (void)c->rccl.CommDestroy(c->comm); c->comm = nullptr; }
  dev_free(c->commPacked); dev_free(c->commRecv); dev_free(c->commPixels); dev_free(c->commMeta);
  c->commCount.clear(); c->commRank = -1; c->commWorld = 0;
  return HYDRA_HIP_OK;
}
// the staging a gather needs: this rank's packed buffer; on the root also the receive buffer and every other rank's pixel list.
// The root's lists are a function of (frame, tile size, world): they are rebuilt whenever one of them is not what they were built for
// (hydra_hip_resize and hydra_hip_set_tile_partition also drop them).
static int comm_prepare(hydra_hip_ctx* c, int root) {
  if (!c->stateAllocated) { int rc = alloc_render_state(c); if (rc) return rc; }
  int rc;
  if ((rc = dev_alloc(c, c->commPacked, std::max<size_t>(1, size_t(c->N)) * 16)) != 0) return rc;
  if (c->commRank != root) return HYDRA_HIP_OK;
  if (!c->commCount.empty() && c->commW == c->w && c->commH == c->h && c->commTile == c->tile && int(c->commCount.size()) == c->commWorld) return HYDRA_HIP_OK;
  std::vector<int> all, one;
  c->commCount.assign(size_t(c->commWorld), 0);
  for (int r = 0; r < c->commWorld; r++) {
    if (r == root) continue;
    build_slot_map(c->w, c->h, c->tile, r, c->commWorld, &one, nullptr);
    c->commCount[size_t(r)] = (long long)one.size();
    all.insert(all.end(), one.begin(), one.end());
  }
  if ((rc = dev_upload(c, c->commPixels, all.data(), all.size() * 4)) != 0) { c->commCount.clear(); return rc; }
  if ((rc = dev_alloc(c, c->commRecv, std::max<size_t>(1, all.size()) * 16)) != 0) { c->commCount.clear(); return rc; }
  c->commW = c->w; c->commH = c->h; c->commTile = c->tile;
  return HYDRA_HIP_OK;
}
// Before any rank posts a send or a receive, all of them learn what every other one is about to do: one ncclAllGather of
// {pixels to send or -1 when this rank could not prepare, width, height, tile}.  A rank that failed, or whose frame or tile size is not
// the others', makes EVERY rank return an error instead of leaving the root waiting in a receive nobody will match; with equal
// (frame, tile, world) the per-rank counts are the same function on every rank, so the root's table agrees by construction (checked).
static int comm_agree(hydra_hip_ctx* c, int root, int prepareRc) {
  const int W = c->commWorld;
  int rc;
  if ((rc = dev_alloc(c, c->commMeta, size_t(W + 1) * 16)) != 0) return rc;   // [0..W) gathered rows, row W = this rank's
  const int mine[4] = {prepareRc == HYDRA_HIP_OK ? c->N : -1, c->w, c->h, c->tile};
  int* meta = static_cast<int*>(c->commMeta.p);
  HCHECK(hipMemcpyAsync(meta + 4 * W, mine, 16, hipMemcpyHostToDevice, c->stream));
  NCHECK(c->rccl.AllGather(meta + 4 * W, meta, 4, ncclInt32, c->comm, c->stream));
  std::vector<int> rows(size_t(W) * 4);
  HCHECK(hipMemcpyAsync(rows.data(), meta, size_t(W) * 16, hipMemcpyDeviceToHost, c->stream));
  HCHECK(hipStreamSynchronize(c->stream));
  const std::string keep = c->err;
  for (int r = 0; r < W; r++) {
    const int* q = &rows[size_t(r) * 4];
    if (q[0] < 0) return fail(c, HYDRA_HIP_ESTATE, r == c->commRank ? ("comm_gather_frame: this rank could not prepare the exchange: " + keep) : ("comm_gather_frame: rank " + std::to_string(r) + " could not prepare the exchange"));
    if (q[1] != c->w || q[2] != c->h || q[3] != c->tile) return fail(c, HYDRA_HIP_ESTATE, "comm_gather_frame: rank " + std::to_string(r) + " renders another frame or tile size than rank " + std::to_string(c->commRank));
    if (c->commRank == root && r != root && (long long)q[0] != c->commCount[size_t(r)]) return fail(c, HYDRA_HIP_ESTATE, "comm_gather_frame: rank " + std::to_string(r) + " owns another number of pixels than the root's table says");
  }
  return HYDRA_HIP_OK;
}
int hydra_hip_comm_gather_frame(hydra_hip_handle c, int root) {
  if (!c) return HYDRA_HIP_EINVAL;
  if (c->world == 1 && !c->comm) return HYDRA_HIP_OK;     // one rank owns the whole frame
  if (!c->comm) return fail(c, HYDRA_HIP_ESTATE, "comm_gather_frame: comm_init first");
  if (root < 0 || root >= c->commWorld) return fail(c, HYDRA_HIP_EINVAL, "comm_gather_frame: bad root");
  if (c->rank != c->commRank || c->world != c->commWorld) return fail(c, HYDRA_HIP_ESTATE, "comm_gather_frame: the tile partition changed after comm_init");
  HCHECK(hipSetDevice(c->device));
  { const int prc = comm_prepare(c, root); const int arc = comm_agree(c, root, prc); if (arc) return arc; if (prc) return prc; }
  if (c->world == 1) return HYDRA_HIP_OK;                 // a one-rank communicator: the agreement ran, nothing to move
  if (c->commRank != root) {
    if (c->N > 0) {
      hipLaunchKernelGGL(k_pack_owned, dim3(grid_for(c, c->N, 256, 8)), dim3(256), 0, c->stream, c->N, static_cast<const int*>(c->ownedPixels.p), c->accum, static_cast<float4*>(c->commPacked.p));
      HCHECK(hipGetLastError());
      NCHECK(c->rccl.Send(c->commPacked.p, size_t(c->N) * 4, ncclFloat, root, c->comm, c->stream));
    }
    return HYDRA_HIP_OK;
  }
  long long total = 0;
  NCHECK(c->rccl.GroupStart());
  for (int r = 0; r < c->commWorld; r++) {
    if (r == root || c->commCount[size_t(r)] == 0) continue;
    const ncclResult_t rr = c->rccl.Recv(static_cast<float4*>(c->commRecv.p) + total, size_t(c->commCount[size_t(r)]) * 4, ncclFloat, r, c->comm, c->stream);
    if (rr != ncclSuccess) { (void)c->rccl.GroupEnd(); c->err = std::string("ncclRecv: ") + c->rccl.GetErrorString(rr); return HYDRA_HIP_EDEVICE; }
    total += c->commCount[size_t(r)];
  }
  NCHECK(c->rccl.GroupEnd());
  if (total > 0) {
    hipLaunchKernelGGL(k_unpack_owned, dim3(grid_for(c, int(total), 256, 8)), dim3(256), 0, c->stream, int(total), static_cast<const int*>(c->commPixels.p),
                       static_cast<const float4*>(c->commRecv.p), c->accum);
    HCHECK(hipGetLastError());
  }
  return HYDRA_HIP_OK;
}
int hydra_hip_comm_reduce_frame(hydra_hip_handle c, int root) {
  if (!c) return HYDRA_HIP_EINVAL;
  if (c->world == 1) return HYDRA_HIP_OK;
  if (!c->comm) return fail(c, HYDRA_HIP_ESTATE, "comm_reduce_frame: comm_init first");
  if (root < 0 || root >= c->commWorld) return fail(c, HYDRA_HIP_EINVAL, "comm_reduce_frame: bad root");
  if (!c->stateAllocated || c->accum == nullptr) return fail(c, HYDRA_HIP_ESTATE, "comm_reduce_frame: nothing rendered yet");
  HCHECK(hipSetDevice(c->device));
  NCHECK(c->rccl.Reduce(c->accum, c->accum, size_t(c->w) * c->h * 4, ncclFloat, ncclSum, root, c->comm, c->stream));
  return HYDRA_HIP_OK;
}
// the pack / unpack kernels of comm_gather_frame without a second GPU: this rank's pixels are packed and dropped into a zeroed
// frame, which is returned (test entry point: the result must equal the accumulator on the rank's tiles and be zero elsewhere)
int hydra_hip_stage_pack_unpack(hydra_hip_handle c, float* rgba_frame, int width, int height) {
  if (!c || !rgba_frame) return HYDRA_HIP_EINVAL;
  if (width != c->w || height != c->h) return fail(c, HYDRA_HIP_EINVAL, "stage_pack_unpack: bad resolution");
  if (!c->stateAllocated || c->accum == nullptr) return fail(c, HYDRA_HIP_ESTATE, "stage_pack_unpack: nothing rendered yet");
  HCHECK(hipSetDevice(c->device));
  int rc;
  if ((rc = dev_alloc(c, c->commPacked, std::max<size_t>(1, size_t(c->N)) * 16)) != 0) return rc;
  DevBuf frame;
  if ((rc = dev_alloc(c, frame, size_t(width) * height * 16)) != 0) return rc;
  HCHECK(hipMemsetAsync(frame.p, 0, size_t(width) * height * 16, c->stream));
  if (c->N > 0) {
    hipLaunchKernelGGL(k_pack_owned, dim3(grid_for(c, c->N, 256, 8)), dim3(256), 0, c->stream, c->N, static_cast<const int*>(c->ownedPixels.p), c->accum, static_cast<float4*>(c->commPacked.p));
    hipLaunchKernelGGL(k_unpack_owned, dim3(grid_for(c, c->N, 256, 8)), dim3(256), 0, c->stream, c->N, static_cast<const int*>(c->ownedPixels.p),
                       static_cast<const float4*>(c->commPacked.p), static_cast<float4*>(frame.p));
  }
  const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(c->stream);
  const hipError_t e3 = (e1 == hipSuccess && e2 == hipSuccess) ? hipMemcpy(rgba_frame, frame.p, size_t(width) * height * 16, hipMemcpyDeviceToHost) : hipSuccess;
  dev_free(frame);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(c, HYDRA_HIP_EDEVICE, "stage_pack_unpack: a HIP call failed");
  return HYDRA_HIP_OK;
}

}  // extern "C"
