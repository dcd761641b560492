// hk_shading.h -- device functions for the non-traversal stages: RNG, camera, surface reconstruction, software
// textures, BxDFs, area lights.  Each function cites the reference inline kernel it reproduces (hydra_drv/...).
#pragma once
#include "hk_common.h"
#include "hk_trace.h"

// ================================================================================================ R1: RandomGen
// crandom.h:10-83.  2 x u32 state, integer-exact.
struct RandomGen { uint32_t x, y; };

HK_DEV uint32_t NextState(RandomGen& g) {
  const uint32_t x = g.x * 17u + g.y * 13123u;
  g.x = (x << 13) ^ x;
  g.y ^= (x << 7);
  return x;
}
HK_DEV RandomGen RandomGenInit(int a_seed) {
  RandomGen g;
  const uint32_t s = uint32_t(a_seed);
  g.x = (s * (s * s * 15731u + 74323u) + 871483u);
  g.y = (s * (s * s * 13734u + 37828u) + 234234u);
  for (int i = 0; i < (a_seed % 7); i++) NextState(g);
  return g;
}
HK_DEV float4 rndFloat4_Pseudo(RandomGen& g) {
  const uint32_t x = NextState(g);
  const uint32_t x1 = (x * (x * x * 15731u + 74323u) + 871483u);
  const uint32_t y1 = (x * (x * x * 13734u + 37828u) + 234234u);
  const uint32_t z1 = (x * (x * x * 11687u + 26461u) + 137589u);
  const uint32_t w1 = (x * (x * x * 15707u + 789221u) + 1376312589u);
  const float scale = (1.0f / 4294967296.0f);
  return make_float4(float(x1) * scale, float(y1) * scale, float(z1) * scale, float(w1) * scale);
}
HK_DEV float rndFloat1_Pseudo(RandomGen& g) {
  const uint32_t x = NextState(g);
  const uint32_t tmp = (x * (x * x * 15731u + 74323u) + 871483u);
  return float(tmp) * (1.0f / 4294967296.0f);
}

// ================================================================================================ small helpers
HK_DEV float epsilonOfPos(f3 p) { return fmaxf(fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))), 2.0f * HK_GEPSILON) * HK_GEPSILON; }   // cglobals.h:737
HK_DEV float misHeuristicPower1(float p) { return isfinite(p) ? fabsf(p) : 0.0f; }
HK_DEV float misWeightHeuristic(float a, float b) {   // cglobals.h:741-745 (power 1 = balance heuristic)
  const float w = misHeuristicPower1(a) / fmaxf(misHeuristicPower1(a) + misHeuristicPower1(b), HK_DEPSILON2);
  return isfinite(w) ? w : 0.0f;
}
HK_DEV f3 OffsRayPos(f3 hitPos, f3 n, f3 sampleDir) {   // cglobals.h:764-769
  const float sgn = dot(sampleDir, n) < 0.0f ? -1.0f : 1.0f;
  return hitPos + n * (sgn * epsilonOfPos(hitPos));
}
HK_DEV f3 OffsShadowRayPos(f3 hitPos, f3 n, f3 sampleDir, float aux) {   // cglobals.h:779-784
  const float sgn = dot(sampleDir, n) < 0.0f ? -1.0f : 1.0f;
  return hitPos + n * (sgn * (epsilonOfPos(hitPos) + aux));
}
HK_DEV f3 reflect3(f3 dir, f3 n) { return normalize(((n * dot(dir, n)) * (-2.0f)) + dir); }   // cglobals.h:686-691

HK_DEV void CoordinateSystem(f3 v1, f3& v2, f3& v3) {   // cglobals.h:1502-1518
  if (fabsf(v1.x) > fabsf(v1.y)) {
    const float invLen = 1.0f / sqrtf(v1.x * v1.x + v1.z * v1.z);
    v2 = mk3((-1.0f) * v1.z * invLen, 0.0f, v1.x * invLen);
  } else {
    const float invLen = 1.0f / sqrtf(v1.y * v1.y + v1.z * v1.z);
    v2 = mk3(0.0f, v1.z * invLen, (-1.0f) * v1.y * invLen);
  }
  v3 = cross(v1, v2);
}
HK_DEV f3 MapSampleToCosineDistribution(float r1, float r2, f3 direction, f3 hit_norm, float power) {   // cglobals.h:1521-1559
  if (power >= 1e6f) return direction;
  const float sin_phi = sinf(2.0f * r1 * 3.141592654f), cos_phi = cosf(2.0f * r1 * 3.141592654f);
  const float cos_theta = powf(1.0f - r2, 1.0f / (power + 1.0f));
  const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
  const f3 dev = mk3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
  f3 nx, nzz;
  CoordinateSystem(direction, nx, nzz);
  const f3 ny = nzz, nz = direction;   // the reference swaps ny and nz after building the frame
  f3 res = ((nx * dev.x) + (ny * dev.y)) + (nz * dev.z);
  const float invSign = dot(direction, hit_norm) > 0.0f ? 1.0f : -1.0f;
  if (invSign * dot(res, hit_norm) < 0.0f) res = (((nx * (-1.0f)) * dev.x) + (ny * dev.y)) - (nz * dev.z);
  return res;
}
HK_DEV f3 MapSampleToModifiedCosineDistribution(float r1, float r2, f3 direction, f3 hit_norm, float power, bool& under) {   // :1563-1601
  if (power >= 1e6f) return direction;
  const float sin_phi = sinf(2.0f * r1 * 3.141592654f), cos_phi = cosf(2.0f * r1 * 3.141592654f);
  const float sin_theta = sqrtf(1.0f - powf(r2, 2.0f / (power + 1.0f)));
  f3 dev;
  dev.x = sin_theta * cos_phi;
  dev.y = sin_theta * sin_phi;
  dev.z = sqrtf(1.0f - dev.x * dev.x - dev.y * dev.y);
  f3 nx, nzz;
  CoordinateSystem(direction, nx, nzz);
  const f3 ny = nzz, nz = direction;
  f3 res = ((nx * dev.x) + (ny * dev.y)) + (nz * dev.z);
  under = false;
  const float invSign = dot(direction, hit_norm) >= 0.0f ? 1.0f : -1.0f;
  if (invSign * dot(res, hit_norm) < 0.0f) {
    res = (((nx * (-1.0f)) * dev.x) - (ny * dev.y)) + (nz * dev.z);
    under = true;
  }
  return res;
}
HK_DEV f2 MapSamplesToDisc(f2 xy) {   // cglobals.h:1609-1652
  const float x = xy.x, y = xy.y;
  float r = 0, phi = 0;
  if (x > y && x > -y) { r = x; phi = 0.25f * 3.141592654f * (y / x); }
  if (x < y && x > -y) { r = y; phi = 0.25f * 3.141592654f * (2.0f - x / y); }
  if (x < y && x < -y) { r = -x; phi = 0.25f * 3.141592654f * (4.0f + y / x); }
  if (x > y && x < -y) { r = -y; phi = 0.25f * 3.141592654f * (6 - x / y); }
  return mk2(r * sinf(phi), r * cosf(phi));
}
HK_DEV float sRGBToLinear(float s) {   // cglobals.h:3024-3030
  if (s <= 0.0404482362771082f) return s * 0.077399381f;
  return powf((s + 0.055f) * 0.947867299f, 2.4f);
}
HK_DEV float linearToSRGB(float l) {   // cglobals.h:3032-3038 (double constants on the CPU path)
  if (l <= 0.00313066844250063f) return l * 12.92f;
  return float(1.055 * double(powf(l, 1.0f / 2.4f)) - 0.055);
}

// ================================================================================================ P1: camera
HK_DEV f3 EyeRayDirNormalized(float x, float y, const m44& projInv) {   // cglobals.h:1069-1078
  float4 pos = make_float4(2.0f * x - 1.0f, 2.0f * y - 1.0f, 0.0f, 1.0f);
  pos = mul4x4x4(projInv, pos);
  return normalize(mk3(pos.x / pos.w, pos.y / pos.w, pos.z / pos.w));
}
HK_DEV f3 tiltCorrection(f3 ray_pos, f3 ray_dir, const SceneDev& s) {   // cfetch.h:832-863
  const float tiltX = g_varsF(s)[HV_F_TILT_ROT_X], tiltY = g_varsF(s)[HV_F_TILT_ROT_Y];
  if ((fabsf(tiltX) > 0.0f || fabsf(tiltY) > 0.0f) && fabsf(ray_dir.z) > 0.0f) {
    const float t = (-1.0f - ray_pos.z) / ray_dir.z;
    f3 p = ray_pos + ray_dir * t;
    p.z += 1.0f;
    if (fabsf(tiltY) > 0.0f) { const float sn = sinf(-tiltY), cs = cosf(-tiltY); p = mk3(p.x * cs + p.z * sn, p.y, p.x * (-sn) + p.z * cs); }
    if (fabsf(tiltX) > 0.0f) { const float sn = sinf(-tiltX), cs = cosf(-tiltX); p = mk3(p.x, p.y * cs + p.z * (-sn), p.y * sn + p.z * cs); }
    p.z -= 1.0f;
    ray_dir = normalize(p - ray_pos);
  }
  return ray_dir;
}
HK_DEV void MakeRandEyeRay(int x, int y, int w, int h, float4 offsets, const SceneDev& s, f3& outPos, f3& outDir) {   // cfetch.h:877-930
  const m44 projInv = load_m44(reinterpret_cast<const float4*>(s.globals + HG_MPROJ_INV));
  const m44 wvInv = load_m44(reinterpret_cast<const float4*>(s.globals + HG_MWORLDVIEW_INV));
  f3 ray_pos = mk3(0.0f, 0.0f, 0.0f);
  f3 ray_dir = EyeRayDirNormalized((float(x) + 0.5f) / float(w), (float(y) + 0.5f) / float(h), projInv);
  {
    const float sinFov = sinf(0.5f * g_varsF(s)[HV_F_CAM_FOV]);
    const float pxSizeX = sinFov * (1.0f / float(w)), pxSizeY = sinFov * (1.0f / float(h));
    ray_dir.x += pxSizeX * offsets.x;
    ray_dir.y += pxSizeY * offsets.y;
    ray_dir.z = -sqrtf(1.0f - (ray_dir.x * ray_dir.x + ray_dir.y * ray_dir.y));
  }
  ray_dir = tiltCorrection(ray_pos, ray_dir, s);
  if (g_varsI(s)[HV_I_ENABLE_DOF] == 1) {
    const float tFocus = g_varsF(s)[HV_F_DOF_FOCAL_PLANE_DIST] / (-ray_dir.z);
    const f3 focusPosition = ray_pos + ray_dir * tFocus;
    const f2 d = MapSamplesToDisc(mk2(1.0f * offsets.z, 1.0f * offsets.w));
    const float R = g_varsF(s)[HV_F_DOF_LENS_RADIUS];
    ray_pos.x += R * d.x;
    ray_pos.y += R * d.y;
    ray_dir = normalize(focusPosition - ray_pos);
  }
  const f3 pos = mul4x3(wvInv, ray_pos);   // matrix4x4f_mult_ray3, cglobals.h:1080-1087
  const f3 pos2 = mul4x3(wvInv, ray_pos + ray_dir * 100.0f);
  outPos = pos;
  outDir = normalize(pos2 - pos);
}

// ================================================================================================ H1: surface
struct SurfaceHit {   // cglobals.h:2514-2528
  f3 pos, normal, flatNormal, tangent, biTangent;
  f2 texCoord;
  int matId; float t, sRayOff; bool hfi;
};

HK_DEV int remapMaterialId(int mId, int instId, const SceneDev& s) {   // cglobals.h:2931-2983
  if (mId < 0 || instId < 0 || instId >= s.remapInstSize || s.remapInst == nullptr || s.remapLists == nullptr || s.remapTable == nullptr) return mId;
  const int listId = s.remapInst[instId];
  if (listId < 0 || listId >= s.remapTableSize) return mId;
  const int offs = s.remapTable[2 * listId], size = s.remapTable[2 * listId + 1];
  int low = 0, high = size - 1;
  while (low <= high) {
    const int mid = low + ((high - low) / 2);
    if (s.remapLists[offs + mid * 2] >= mId) high = mid - 1; else low = mid + 1;
  }
  if (high + 1 < size) {
    const int from = s.remapLists[offs + (high + 1) * 2], to = s.remapLists[offs + (high + 1) * 2 + 1];
    return (from == mId) ? to : mId;
  }
  return mId;
}

// The vertex data of one triangle, however it is stored
struct TriData { float4 A1, B1, C1, A2, B2, C2, At, Bt, Ct; int matId; float sRayOff; };
HK_DEV TriData fetchTriFromMesh(const HydraLiteHit& hit, const float4* __restrict__ mesh) {   // the reference's arena layout, cfetch.h:1038-1119
  const HydraPlainMesh* hdr = reinterpret_cast<const HydraPlainMesh*>(mesh);
  const float4* vertPos = mesh + hdr->vPosOffset;
  const float4* vertNorm = mesh + hdr->vNormOffset;
  const float4* vertTang = mesh + hdr->vTangentOffset;
  const int* vertIndices = reinterpret_cast<const int*>(mesh + hdr->vIndicesOffset);
  const int* matIndices = reinterpret_cast<const int*>(mesh + hdr->mIndicesOffset);
  const float* shadowRayOff = reinterpret_cast<const float*>(mesh + hdr->polyShadowOffset);
  TriData d;
  d.matId = matIndices[hit.primId];
  const int o = hit.primId * 3;
  const int iA = vertIndices[o], iB = vertIndices[o + 1], iC = vertIndices[o + 2];
  d.A1 = vertPos[iA]; d.B1 = vertPos[iB]; d.C1 = vertPos[iC];
  d.A2 = vertNorm[iA]; d.B2 = vertNorm[iB]; d.C2 = vertNorm[iC];
  d.At = vertTang[iA]; d.Bt = vertTang[iB]; d.Ct = vertTang[iC];
  d.sRayOff = shadowRayOff[hit.primId];
  return d;
}
// Per-triangle records (SceneDev::triRec): 7 loads from ONE aligned 128-byte line instead of ~25 loads from ~15 lines.
// k_bounce is bound by the CU's address path, where the first touch of a line costs ~2.3 clk per lane and every further
// load from it ~1 (tools/micro/ta_bench.hip), and this fetch was two thirds of its divergent loads.
HK_DEV TriData fetchTriFromRecords(const HydraLiteHit& hit, const SceneDev& s) {
  const size_t t = size_t(s.triBase[HK_GEOM_ID(hit.geomId)]) + size_t(hit.primId);
  const float4* r = s.triRec + t * 8;
  const float4* tg = s.triTan + t * 3;
  TriData d;
  d.A1 = r[0]; d.B1 = r[1]; d.C1 = r[2]; d.A2 = r[3]; d.B2 = r[4]; d.C2 = r[5];
  const float4 misc = r[6];
  d.matId = as_int(misc.x); d.sRayOff = misc.y;
  d.At = tg[0]; d.Bt = tg[1]; d.Ct = tg[2];
  return d;
}

HK_DEV SurfaceHit surfaceEvalLS(f3 a_rpos, f3 a_rdir, const HydraLiteHit& hit, const TriData& td) {   // ctrace.h:1988-2109
  const float4 A1 = td.A1, B1 = td.B1, C1 = td.C1, A2 = td.A2, B2 = td.B2, C2 = td.C2;
  SurfaceHit sh;
  sh.matId = td.matId;
  const f3 A_pos = xyz(A1), B_pos = xyz(B1), C_pos = xyz(C1);
  const f3 A_norm = xyz(A2), B_norm = xyz(B2), C_norm = xyz(C2);

  float u, v;   // triBaricentrics
  {
    const f3 edge1 = B_pos - A_pos, edge2 = C_pos - A_pos;
    const f3 pvec = cross(a_rdir, edge2);
    const float inv_det = 1.0f / dot(edge1, pvec);
    const f3 tvec = a_rpos - A_pos;
    v = dot(tvec, pvec) * inv_det;
    const f3 qvec = cross(tvec, edge1);
    u = dot(a_rdir, qvec) * inv_det;
  }
  const float w0 = (1.0f - u - v);
  sh.pos = ((A_pos * w0) + (B_pos * v)) + (C_pos * u);
  sh.texCoord.x = w0 * A1.w + v * B1.w + u * C1.w;
  sh.texCoord.y = w0 * A2.w + v * B2.w + u * C2.w;
  sh.normal = ((A_norm * w0) + (B_norm * v)) + (C_norm * u);
  sh.t = hit.t;
  sh.sRayOff = td.sRayOff;

  const float4 At = td.At, Bt = td.Bt, Ct = td.Ct;
  sh.flatNormal = normalize(cross(A_pos - B_pos, A_pos - C_pos));
  if (dot(a_rdir, sh.flatNormal) > 0.025f) sh.flatNormal = sh.flatNormal * (-1.0f);
  const float maxEdge = fmaxf(fmaxf(length(A_pos - B_pos), length(A_pos - C_pos)), length(B_pos - C_pos));
  if (sh.sRayOff > 1e-5f * maxEdge) {   // smooth low-poly surfaces
    if (dot(a_rdir, sh.normal) > 0.120f) { sh.normal = sh.normal * (-1.0f); sh.hfi = true; }
    else if (dot(a_rdir, sh.normal) > 0.0f) { sh.normal = sh.flatNormal; sh.hfi = false; }
    else sh.hfi = false;
  } else {
    if (dot(a_rdir, sh.normal) > 0.0f) { sh.normal = sh.normal * (-1.0f); sh.hfi = true; }
    else sh.hfi = false;
  }
  const float handed = (At.w < 0.0f || Bt.w < 0.0f || Ct.w < 0.0f) ? -1.0f : 1.0f;
  sh.tangent = normalize(((xyz(At) * w0) + (xyz(Bt) * v)) + (xyz(Ct) * u));
  sh.biTangent = normalize(handed > 0.0f ? cross(sh.normal, sh.tangent) : cross(sh.tangent, sh.normal));
  const bool badTangent = !finite3(sh.biTangent);
  if (fabsf(fabsf(dot(sh.normal, sh.tangent)) - 1.0f) < 1e-4f || badTangent) CoordinateSystem(sh.normal, sh.tangent, sh.biTangent);
  return sh;
}

// the two fetches of evalSurface, separately, so that a kernel can issue them together with its other loads (k_bounce)
HK_DEV TriData fetchTri(const SceneDev& s, const HydraLiteHit& hit) {
#ifdef HK_HOST_EMU
  return fetchTriFromMesh(hit, s.geomStorage + s.globals[s.hdr[HG_GEOM_TABLE_OFFS] + hit.geomId]);
#else
  return fetchTriFromRecords(hit, s);
#endif
}
HK_DEV SurfaceHit evalSurfaceWith(const SceneDev& s, f3 ray_pos, f3 ray_dir, const HydraLiteHit& hit, const TriData& td, const m44& instInv) {   // CPUExp_Integrators_PT_Loop.cpp:35-84
  const f3 posLS = mul4x3(instInv, ray_pos), dirLS = mul3x3(instInv, ray_dir);
  const SurfaceHit ls = surfaceEvalLS(posLS, dirLS, hit, td);
  const m44 inst = inverse_affine(instInv);
  SurfaceHit ws = ls;
  const float multInv = 1.0f / sqrtf(3.0f);
  const f3 shadowStart = mul3x3(inst, mk3(multInv * ls.sRayOff, multInv * ls.sRayOff, multInv * ls.sRayOff));
  const m44 nm = transpose44(instInv);
  ws.pos = mul4x3(inst, ls.pos);
  ws.normal = normalize(mul3x3(nm, ls.normal));
  ws.flatNormal = normalize(mul3x3(nm, ls.flatNormal));
  ws.tangent = normalize(mul3x3(nm, ls.tangent));
  ws.biTangent = normalize(mul3x3(nm, ls.biTangent));
  ws.t = length(ws.pos - ray_pos);
  ws.sRayOff = length(shadowStart);
  ws.matId = remapMaterialId(ws.matId, hit.instId, s);
  return ws;
}
HK_DEV SurfaceHit evalSurface(const SceneDev& s, f3 ray_pos, f3 ray_dir, const HydraLiteHit& hit) {
  const m44 instInv = load_m44(s.instMatrices + size_t(hit.instId) * 4);
  const TriData td = fetchTri(s, hit);
  return evalSurfaceWith(s, ray_pos, ray_dir, hit, td, instInv);
}

// ================================================================================================ textures
HK_DEV int4 bilinearOffsets(float ffx, float ffy, int flags, int w, int h) {   // cfetch.h:312-362
  const int sx = (ffx > 0.0f) ? 1 : -1, sy = (ffy > 0.0f) ? 1 : -1;
  const int px = int(ffx), py = int(ffy);
  int px_w0, px_w1, py_w0, py_w1;
  if (flags & HTEX_CLAMP_U) {
    px_w0 = (px >= w) ? w - 1 : px; px_w1 = (px + 1 >= w) ? w - 1 : px + 1;
    px_w0 = (px_w0 < 0) ? 0 : px_w0; px_w1 = (px_w1 < 0) ? 0 : px_w1;
  } else {
    px_w0 = px % w; px_w1 = (px + sx) % w;
    px_w0 = (px_w0 < 0) ? px_w0 + w : px_w0; px_w1 = (px_w1 < 0) ? px_w1 + w : px_w1;
  }
  if (flags & HTEX_CLAMP_V) {
    py_w0 = (py >= h) ? h - 1 : py; py_w1 = (py + 1 >= h) ? h - 1 : py + 1;
    py_w0 = (py_w0 < 0) ? 0 : py_w0; py_w1 = (py_w1 < 0) ? 0 : py_w1;
  } else {
    py_w0 = py % h; py_w1 = (py + sy) % h;
    py_w0 = (py_w0 < 0) ? py_w0 + h : py_w0; py_w1 = (py_w1 < 0) ? py_w1 + h : py_w1;
  }
  return make_int4(py_w0 * w + px_w0, py_w0 * w + px_w1, py_w1 * w + px_w0, py_w1 * w + px_w1);
}
HK_DEV float srgbByteToLinear(unsigned char c) { return sRGBToLinear(0.003921568f * float(c)); }   // what read_uchar4 computes per channel
HK_DEV float4 read_uchar4(const uchar4* data, int offset, bool srgb, const float* lut) {   // cfetch.h:298-303
  const float mult = 0.003921568f;
  const uchar4 c = data[offset];
  // a tap decodes 4 channels through powf; the 256 possible results come from SceneDev::srgbLut instead (same bits, see hk_common.h)
  if (srgb && lut != nullptr) return make_float4(lut[c.x], lut[c.y], lut[c.z], lut[c.w]);
  float4 r = make_float4(mult * float(c.x), mult * float(c.y), mult * float(c.z), mult * float(c.w));
  if (srgb) r = make_float4(sRGBToLinear(r.x), sRGBToLinear(r.y), sRGBToLinear(r.z), sRGBToLinear(r.w));
  return r;
}
HK_DEV float4 read_imagef_sw4(const int4* tex, f2 tc, int flags, bool srgb, const float* lut) {   // cfetch.h:461-584 (4-channel textures)
  const int4 header = tex[0];
  const int w = header.x, h = header.y, bpp = header.w;
  float ffx = tc.x * float(w) - 0.5f, ffy = tc.y * float(h) - 0.5f;
  if ((flags & HTEX_CLAMP_U) != 0 && ffx < 0) ffx = 0.0f;
  if ((flags & HTEX_CLAMP_V) != 0 && ffy < 0) ffy = 0.0f;
  const uchar4* bytes = reinterpret_cast<const uchar4*>(tex + 1);
  const float4* fdata = reinterpret_cast<const float4*>(tex + 1);
  if (flags & HTEX_POINT_SAM) {
    int px = int(ffx + 0.5f), py = int(ffy + 0.5f);
    if (flags & HTEX_CLAMP_U) { px = (px >= w) ? w - 1 : px; px = (px < 0) ? 0 : px; } else { px = px % w; px = (px < 0) ? px + w : px; }
    if (flags & HTEX_CLAMP_V) { py = (py >= h) ? h - 1 : py; py = (py < 0) ? 0 : py; } else { py = py % h; py = (py < 0) ? py + h : py; }
    const int offset = py * w + px;
    if (bpp == 4) return read_uchar4(bytes, offset, srgb, lut);
    if (bpp == 16) return fdata[offset];
    return make_float4(0, 0, 0, 0);
  }
  const int px = int(ffx), py = int(ffy);
  const float fx = fabsf(ffx - float(px)), fy = fabsf(ffy - float(py));
  const float fx1 = 1.0f - fx, fy1 = 1.0f - fy;
  const float w1 = fx1 * fy1, w2 = fx * fy1, w3 = fx1 * fy, w4 = fx * fy;
  const int4 offs = bilinearOffsets(ffx, ffy, flags, w, h);
  float4 f1, f2_, f3_, f4_;
  if (bpp == 4) {
    f1 = read_uchar4(bytes, offs.x, srgb, lut); f2_ = read_uchar4(bytes, offs.y, srgb, lut);
    f3_ = read_uchar4(bytes, offs.z, srgb, lut); f4_ = read_uchar4(bytes, offs.w, srgb, lut);
  } else {
    f1 = fdata[offs.x]; f2_ = fdata[offs.y]; f3_ = fdata[offs.z]; f4_ = fdata[offs.w];
  }
  return make_float4(f1.x * w1 + f2_.x * w2 + f3_.x * w3 + f4_.x * w4, f1.y * w1 + f2_.y * w2 + f3_.y * w3 + f4_.y * w4,
                     f1.z * w1 + f2_.z * w2 + f3_.z * w3 + f4_.z * w4, f1.w * w1 + f2_.w * w2 + f3_.w * w3 + f4_.w * w4);
}
// sample2DExt, cfetch.h:677-709, without procedural textures.  blob = owning material/light node, int4-addressed.
// A real call (HK_DEV_CALL) with everything it needs passed by value: the body appears once per kernel.
HK_DEV_CALL f3 sample2DExtCall(int samplerOffset, f2 texCoord, const float* blob, const int* texTable, const int4* texStorage, const float* srgbLut) {
  if (uint32_t(samplerOffset) == HYDRA_INVALID_TEXTURE || samplerOffset < 0) return mk3(1, 1, 1);
  const float* sm = blob + size_t(samplerOffset) * 4;
  const int flags = as_int(sm[HS_FLAGS]);
  const float gamma = sm[HS_GAMMA];
  const int texId = as_int(sm[HS_TEXID]);
  if (texId <= 0) return mk3(1, 1, 1);
  const f2 tct = mk2(sm[HS_ROW0] * texCoord.x + sm[HS_ROW0 + 1] * texCoord.y + sm[HS_ROW0 + 3],
                     sm[HS_ROW1] * texCoord.x + sm[HS_ROW1 + 1] * texCoord.y + sm[HS_ROW1 + 3]);
  const int offset = texTable[texId];
  float4 c = make_float4(1, 1, 1, 1);
  if (offset >= 0) c = read_imagef_sw4(texStorage + offset, tct, flags, (gamma != 1.0f), srgbLut);
  if (flags & HTEX_ALPHASRC_W) { c.x = c.w; c.y = c.w; c.z = c.w; }
  return mk3(c.x, c.y, c.z);
}
// ---- procedural textures: readProcTex (cglobals.h:2402-2441) over the per-path list the run-time compiled kernel wrote (SceneDev::ptl*, hk_proctex_rt.h).
// Returns w = -1 when the path's list does not hold the id (the reference's "not found" marker), else the colour with w = 0.
#ifdef HK_HOST_EMU
HK_DEV float hk_half_to_float(unsigned h) {
  const unsigned sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
  if (e == 0) return as_float(int(sign)) * 0.0f + (sign ? -1.0f : 1.0f) * float(m) * 5.9604644775390625e-08f;   // zero and subnormals: m * 2^-24
  if (e == 31) return as_float(int(sign | 0x7f800000u | (m << 13)));
  return as_float(int(sign | ((e + 112u) << 23) | (m << 13)));
}
#else
HK_DEV float hk_half_to_float(unsigned h) { return float(__builtin_bit_cast(_Float16, (unsigned short)h)); }
#endif
HK_DEV float4 readProcTexAt(int texId, const int* ids, const uint2* vals, int stride, int maxNum) {   // ids / vals: already offset to the path's slot
  float4 r = make_float4(1.0f, 1.0f, 1.0f, -1.0f);
  for (int k = 0; k < maxNum; k++) {
    const int id = ids[size_t(k) * stride];
    if (uint32_t(id) == HYDRA_INVALID_TEXTURE) break;
    if (id == texId) {
      const uint2 v = vals[size_t(k) * stride];
      r = make_float4(hk_half_to_float(v.x & 0xffffu), hk_half_to_float(v.x >> 16), hk_half_to_float(v.y & 0xffffu), 0.0f);
      break;
    }
  }
  return r;
}
HK_DEV float4 readProcTex(int texId, const SceneDev& s) { return readProcTexAt(texId, s.ptlIds + s.ptlSlot, s.ptlVals + s.ptlSlot, s.ptlStride, s.ptlMax); }
// sample2DExt with the path's procedural textures (cfetch.h:677-709 as written): the stored texture is fetched when the id has one, the procedural colour wins.
// A real call like sample2DExtCall, so everything by value (a reference to the scene view would force the caller's copy of it into memory).
HK_DEV_CALL f3 sample2DExtProc(int samplerOffset, f2 texCoord, const float* blob, const int* texTable, const int4* texStorage, const float* srgbLut,
                               const int* ids, const uint2* vals, int stride, int maxNum) {
  const float* sm = blob + size_t(samplerOffset) * 4;
  const int flags = as_int(sm[HS_FLAGS]);
  const int texId = as_int(sm[HS_TEXID]);
  if (texId <= 0) return mk3(1, 1, 1);
  const f2 tct = mk2(sm[HS_ROW0] * texCoord.x + sm[HS_ROW0 + 1] * texCoord.y + sm[HS_ROW0 + 3],
                     sm[HS_ROW1] * texCoord.x + sm[HS_ROW1 + 1] * texCoord.y + sm[HS_ROW1 + 3]);
  float4 c = readProcTexAt(texId, ids, vals, stride, maxNum);
  if (fabsf(c.w + 1.0f) < 1e-5f) {
    const int offset = texTable[texId];
    c = make_float4(1, 1, 1, 1);
    if (offset >= 0) c = read_imagef_sw4(texStorage + offset, tct, flags, (sm[HS_GAMMA] != 1.0f), srgbLut);
  }
  if (flags & HTEX_ALPHASRC_W) { c.x = c.w; c.y = c.w; c.z = c.w; }
  return mk3(c.x, c.y, c.z);
}
HK_DEV f3 sample2DExt(int samplerOffset, f2 texCoord, const float* blob, const SceneDev& s) {
  if (uint32_t(samplerOffset) == HYDRA_INVALID_TEXTURE || samplerOffset < 0) return mk3(1, 1, 1);   // the common untextured node: no call
  if (s.ptlSlot >= 0) return sample2DExtProc(samplerOffset, texCoord, blob, s.texTable, s.texStorage, s.srgbLut, s.ptlIds + s.ptlSlot, s.ptlVals + s.ptlSlot, s.ptlStride, s.ptlMax);   // only in kernels that set a slot (HK_FEAT_PROCTEX); the constant -1 elsewhere
  return sample2DExtCall(samplerOffset, texCoord, blob, s.texTable, s.texStorage, s.srgbLut);
}

// Compile-time feature sets of the shading code.  A kernel instantiated with a subset does not contain (nor keep registers for)
// the rest; the host picks the instantiation from the material classes and light types the uploaded scene really has.
enum { HK_FEAT_SKY = 1, HK_FEAT_DELTA_LIGHTS = 2, HK_FEAT_OREN_NAYAR = 4, HK_FEAT_GLASS = 8, HK_FEAT_GGX = 16, HK_FEAT_NMAP = 32, HK_FEAT_TRANSLUCENT = 64, HK_FEAT_BLINN = 128, HK_FEAT_ANISO = 256, HK_FEAT_PEREZ = 512, HK_FEAT_RARE_LIGHTS = 1024 /* sky portals, cylinder lights, textured mesh lights */, HK_FEAT_ALL = 2047,
       HK_FEAT_PROCTEX = 2048 /* the kernel reads the per-path procedural texture lists (SceneDev::ptl*); NOT part of HK_FEAT_ALL: one more instantiation on top of it */,
       HK_FEAT_CLASSIC = 31 /* everything but normal maps, translucent, Blinn and the anisotropic (Beckmann, TRGGX) nodes */ };   // DELTA_LIGHTS stands for "lights other than area and sky": point, spot, directional, sphere

// ================================================================================================ materials
// a back-plate named by the header (a sky light's or a shadow catcher's <back>): the miss shader and the shadow catcher then follow the OpenCL layer (environmentColorExtended below)
HK_DEV bool haveBackPlate(const SceneDev& s) { return uint32_t(g_varsI(s)[HV_I_SHADOW_MATTE_BACK]) != HYDRA_INVALID_TEXTURE; }
struct MatSample { f3 color; f3 direction; float pdf; int flags; };          // cglobals.h:394-402
#define HK_MATTE_PENDING (1 << 30)   /* internal bit of MatSample::flags: a shadow catcher sample whose throughput still has to be multiplied by this bounce's shadow (never reaches the ray flags) */
struct BxDFResult { f3 brdf; float pdfFwd; f3 btdf; float pdfRev; bool diffuse; };   // cmaterial.h:2374-2386
struct ShadeContext { f3 l, v, n; f2 tc; f3 fn, tg, bn; };                  // cglobals.h:2282-2301 (fn: light-tracing form of materialEval and normal maps; tg, bn: normal maps and the anisotropic lobes)

// ---- normal maps (cmaterial.h:2208-2243; sample2DAuxExt cfetch.h:795-820 without procedural textures) ----
HK_DEV bool hasNormalMap(const float* m) { return uint32_t(as_int(m[HM_NORMAL_TEX])) != HYDRA_INVALID_TEXTURE; }
HK_DEV f3 sample2DAuxExt(int auxTexId, int samplerOffset, f2 texCoord, const float* blob, const SceneDev& s) {
  if (uint32_t(samplerOffset) == HYDRA_INVALID_TEXTURE) return mk3(1, 1, 1);
  const float* sm = blob + size_t(samplerOffset) * 4;
  const int flags = as_int(sm[HS_FLAGS]);
  const f2 tct = mk2(sm[HS_ROW0] * texCoord.x + sm[HS_ROW0 + 1] * texCoord.y + sm[HS_ROW0 + 3],
                     sm[HS_ROW1] * texCoord.x + sm[HS_ROW1 + 1] * texCoord.y + sm[HS_ROW1 + 3]);
  if (as_int(sm[HS_TEXID]) == 0) return mk3(1, 1, 1);
  float4 c = make_float4(1.0f, 1.0f, 1.0f, -1.0f);
  if (s.ptlSlot >= 0) c = readProcTex(as_int(sm[HS_TEXID]), s);   // a procedural normal map: its slot holds the texture id, not an aux id (PlainMaterialConverter.cpp:1396-1399) -- the aux arena is not read for it
  if (fabsf(c.w + 1.0f) < 1e-5f) c = read_imagef_sw4(s.texAuxStorage + s.texAuxTable[auxTexId], tct, flags, (sm[HS_GAMMA] != 1.0f), s.srgbLut);
  if (flags & HTEX_ALPHASRC_W) { c.x = c.w; c.y = c.w; c.z = c.w; }
  return mk3(c.x, c.y, c.z);
}
HK_DEV f3 materialNormalMapFetch(const float* m, f2 tc, const SceneDev& s) {   // cmaterial.h:2208-2233
  const int flags = as_int(m[HM_FLAGS]);
  const f3 t = sample2DAuxExt(as_int(m[HM_NORMAL_TEX]), as_int(m[HM_NORMAL_TEX_MATRIX]), tc, m, s);
  f3 normalTS = mk3(2.0f * t.x - 1.0f, 2.0f * t.y - 1.0f, t.z);
  if (flags & HMF_INVERT_NMAP_Y) normalTS.y *= (-1.0f);
  if (flags & HMF_INVERT_NMAP_X) normalTS.x *= (-1.0f);
  if (flags & HMF_INVERT_SWAP_NMAP_XY) { const float tmp = normalTS.x; normalTS.x = normalTS.y; normalTS.y = tmp; }
  return normalize(normalTS);
}
HK_DEV f3 BumpMapping(f3 tangent, f3 bitangent, f3 normal, f2 tc, const float* m, const SceneDev& s) {   // cmaterial.h:2235-2243, inverse cglobals.h:893-918
  const f3 nts = materialNormalMapFetch(m, tc, s);
  const f3 r0 = tangent, r1 = bitangent, r2 = normal;   // make_float3x3: rows
  const float det = r0.x * (r1.y * r2.z - r1.z * r2.y) - r0.y * (r1.x * r2.z - r1.z * r2.x) + r0.z * (r1.x * r2.y - r1.y * r2.x);
  f3 b0 = mk3((r1.y * r2.z - r1.z * r2.y), -(r0.y * r2.z - r0.z * r2.y), (r0.y * r1.z - r0.z * r1.y));
  f3 b1 = mk3(-(r1.x * r2.z - r1.z * r2.x), (r0.x * r2.z - r0.z * r2.x), -(r0.x * r1.z - r0.z * r1.x));
  f3 b2 = mk3((r1.x * r2.y - r1.y * r2.x), -(r0.x * r2.y - r0.y * r2.x), (r0.x * r1.y - r0.y * r1.x));
  const float sc = 1.0f / det;
  b0 = b0 * sc; b1 = b1 * sc; b2 = b2 * sc;
  return normalize(mk3(b0.x * nts.x + b0.y * nts.y + b0.z * nts.z, b1.x * nts.x + b1.y * nts.y + b1.z * nts.z, b2.x * nts.x + b2.y * nts.y + b2.z * nts.z));
}

HK_DEV const float* materialAt(const SceneDev& s, int matId) {   // cfetch.h:192-213
  return s.matBase + size_t(s.matTable[matId]) * 4;
}
HK_DEV int matType(const float* m) { return as_int(m[HM_TYPE]); }
// Shading class of a material root, 1..15 (0 is kept for "no hit"): paths of one class run the same code in k_bounce.  Only an
// ordering hint -- nothing reads it for shading -- so per-instance remap lists that swap a material do not matter.
HK_DEV int shadeClassOfMaterial(const float* m) {
  const bool textured = (uint32_t(as_int(m[HM_TEXMATRIXID])) != HYDRA_INVALID_TEXTURE) && as_int(m[HM_TEXMATRIXID]) >= 0;
  switch (as_int(m[HM_TYPE])) {
    case HMT_EMISSIVE: return 1;
    case HMT_LAMBERT: return textured ? 3 : 2;
    case HMT_OREN_NAYAR: return 4;
    case HMT_PHONG: return textured ? 6 : 5;
    case HMT_MIRROR: return 7;
    case HMT_GGX: return 8;
    case HMT_BLEND_MASK: return 9 + (matType(m + size_t(as_int(m[HM_BLEND_MAT1])) * HM_NODE_FLOATS) == HMT_BLEND_MASK || matType(m + size_t(as_int(m[HM_BLEND_MAT2])) * HM_NODE_FLOATS) == HMT_BLEND_MASK ? 1 : 0);
    case HMT_THIN_GLASS: return 11;
    case HMT_GLASS: return 12;
    case HMT_TRANSLUCENT: return 14;
    case HMT_BLINN: return 5;      // shaded next to phong
    default: return 13;
  }
}
HK_DEV int matFlags(const float* m) { return as_int(m[HM_FLAGS]); }
HK_DEV f3 matColor(const float* m) { return mk3(m[HM_COLOR], m[HM_COLOR + 1], m[HM_COLOR + 2]); }

// cmaterial.h:435-450, as compile-time constants (see cosPowerFromGlosiness)
HK_DEV constexpr float hk_glosscoeff_row(int r, int c) {
  constexpr float T[10][4] = {
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
  return T[r][c];
}
// translucent (diffuse transmission), cmaterial.h:1852-1909; colour and sampler sit at the lambert offsets
HK_DEV f3 lambertColorFwd(const float* m, f2 tc, const SceneDev& s);
HK_DEV float translucentEvalPDF(f3 l, f3 v, f3 n) {
  const float sign1 = dot(l, n) > 0 ? 1.0f : -1.0f, sign2 = dot(v, n) > 0 ? 1.0f : -1.0f;
  const float coeff = (sign1 * sign2 < 0.0f) ? 1.0f : 0.0f;
  return fabsf(dot(l, n)) * HK_INV_PI * coeff;
}
HK_DEV float cosPowerFromGlosiness(float x) {   // cmaterial.h:453-466
  const int k = (fabsf(x - 1.0f) < 1e-5f) ? 10 : int(x * 10.0f);
  const float x1 = (x - float(k) * 0.1f);
  if (k == 10 || x >= 0.99f) return 1000000.0f;
  // the row by a chain of selects over literal constants instead of a load from the table in global memory: the lobe functions call this three to
  // four times per path and bounce, each time in the middle of a dependent chain (a cache hit is still several hundred cycles; 40 selects are not)
  float c0 = hk_glosscoeff_row(9, 0), c1 = hk_glosscoeff_row(9, 1), c2 = hk_glosscoeff_row(9, 2), c3 = hk_glosscoeff_row(9, 3);
#pragma unroll
  for (int r = 8; r >= 0; r--)
    if (k == r) { c0 = hk_glosscoeff_row(r, 0); c1 = hk_glosscoeff_row(r, 1); c2 = hk_glosscoeff_row(r, 2); c3 = hk_glosscoeff_row(r, 3); }
  return c3 + c2 * x1 + c1 * x1 * x1 + c0 * x1 * x1 * x1;
}

// ---- lambert, cmaterial.h:219-263
HK_DEV f3 lambertColor(const float* m, f2 tc, const SceneDev& s) {
  return clamp3(sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s) * matColor(m), 0.0f, 1.0f);
}
HK_DEV f3 lambertColorFwd(const float* m, f2 tc, const SceneDev& s) { return lambertColor(m, tc, s); }
HK_DEV f3 translucentEvalBxDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const SceneDev& s) {
  const float sign1 = dot(l, n) > 0 ? 1.0f : -1.0f, sign2 = dot(v, n) > 0 ? 1.0f : -1.0f;
  const float coeff = (sign1 * sign2 < 0.0f) ? 1.0f : 0.0f;
  return (lambertColor(m, tc, s) * coeff) * HK_INV_PI;
}
HK_DEV void TranslucentSampleAndEvalBRDF(const float* m, float r1, float r2, f3 n, f2 tc, const SceneDev& s, MatSample& out) {
  const f3 kd = lambertColor(m, tc, s);
  const f3 nn = n * (-1.0f);
  const f3 newDir = MapSampleToCosineDistribution(r1, r2, nn, nn, 1.0f);
  const float cosTheta = dot(newDir, nn);
  out.direction = newDir;
  out.pdf = cosTheta * HK_INV_PI;
  out.color = kd * HK_INV_PI;
  if (cosTheta <= 1e-6f) out.color = mk3(0, 0, 0);
  out.flags = (HRE_D | HRE_T);
}
HK_DEV void LambertSampleAndEvalBRDF(const float* m, float r1, float r2, f3 n, f2 tc, const SceneDev& s, MatSample& out) {
  const f3 color = lambertColor(m, tc, s);
  const f3 newDir = MapSampleToCosineDistribution(r1, r2, n, n, 1.0f);
  const float cosTheta = dot(newDir, n);
  out.direction = newDir;
  out.pdf = cosTheta * HK_INV_PI;
  out.color = color * HK_INV_PI;
  if (cosTheta <= HK_DEPSILON) out.color = mk3(0, 0, 0);
  out.flags = HRE_D;
}
// ---- oren-nayar, cmaterial.h:264-371 (CosPhiPBRT1 / SinPhiPBRT1: cmatpbrt.h:17-31)
HK_DEV float orennayarFunc(f3 l, f3 v, f3 n, float A, float B) {
  const float cosTheta_wi = dot(l, n), cosTheta_wo = dot(v, n);
  const float sinTheta_wi = sqrtf(fmaxf(0.0f, 1.0f - cosTheta_wi * cosTheta_wi));
  const float sinTheta_wo = sqrtf(fmaxf(0.0f, 1.0f - cosTheta_wo * cosTheta_wo));
  f3 nx, ny;
  CoordinateSystem(n, nx, ny);
  const f3 wo = mk3(-dot(v, nx), -dot(v, ny), -dot(v, n));
  const f3 wi = mk3(-dot(l, nx), -dot(l, ny), -dot(l, n));
  float maxcos = 0.f;
  if (sinTheta_wi > 1e-4f && sinTheta_wo > 1e-4f) {
    const float sinphii = clampf(wi.y / sinTheta_wi, -1.f, 1.f), cosphii = clampf(wi.x / sinTheta_wi, -1.f, 1.f);
    const float sinphio = clampf(wo.y / sinTheta_wo, -1.f, 1.f), cosphio = clampf(wo.x / sinTheta_wo, -1.f, 1.f);
    const float dcos = cosphii * cosphio + sinphii * sinphio;
    maxcos = fmaxf(0.f, dcos);
  }
  float sinalpha, tanbeta;
  if (fabsf(cosTheta_wi) > fabsf(cosTheta_wo)) { sinalpha = sinTheta_wo; tanbeta = sinTheta_wi / fmaxf(fabsf(cosTheta_wi), HK_DEPSILON); }
  else { sinalpha = sinTheta_wi; tanbeta = sinTheta_wo / fmaxf(fabsf(cosTheta_wo), HK_DEPSILON); }
  return (A + B * maxcos * sinalpha * tanbeta);
}
HK_DEV f3 orennayarEvalBxDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const SceneDev& s) {
  return (lambertColor(m, tc, s) * HK_INV_PI) * orennayarFunc(l, v, n, m[HM_ORENNAYAR_A], m[HM_ORENNAYAR_B]);   // same colour/sampler offsets as lambert
}
HK_DEV void OrennayarSampleAndEvalBRDF(const float* m, float r1, float r2, f3 ray_dir, f3 n, f2 tc, const SceneDev& s, MatSample& out) {
  const f3 color = lambertColor(m, tc, s);
  const f3 newDir = MapSampleToCosineDistribution(r1, r2, n, n, 1.0f);
  const float cosTheta = dot(newDir, n);
  out.direction = newDir;
  out.pdf = cosTheta * HK_INV_PI;
  out.color = (color * HK_INV_PI) * orennayarFunc(newDir, ray_dir * (-1.0f), n, m[HM_ORENNAYAR_A], m[HM_ORENNAYAR_B]);
  if (cosTheta <= HK_DEPSILON) out.color = mk3(0, 0, 0);
  out.flags = HRE_D;
}
// ---- phong, cmaterial.h:915-1033
HK_DEV float phongGlosiness(const float* m, f2 tc, const SceneDev& s) {
  if (uint32_t(as_int(m[HM_PHONG_GLOSS_TEXID])) != HYDRA_INVALID_TEXTURE) {
    const f3 g = sample2DExt(as_int(m[HM_PHONG_GLOSS_TEXMATRIXID]), tc, m, s);
    return clampf(m[HM_PHONG_GLOSINESS] * fmaxf(g.x, fmaxf(g.y, g.z)), 0.0f, 0.99f);
  }
  return m[HM_PHONG_GLOSINESS];
}
HK_DEV float phongEvalPDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const SceneDev& s) {
  if (dot(n, v) < 1e-6f || dot(n, l) < 1e-6f) return 1.0f;
  const float cosPower = cosPowerFromGlosiness(phongGlosiness(m, tc, s));
  const f3 r = reflect3(v * (-1.0f), n);
  const float cosTheta = clampf(fabsf(dot(l, r)), 0.0f, 1.0f);
  return powf(cosTheta, cosPower) * (cosPower + 1.0f) * HK_INV_TWOPI;
}
HK_DEV f3 phongEvalBxDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const SceneDev& s) {
  if (dot(n, v) < 1e-6f || dot(n, l) < 1e-6f) return mk3(0, 0, 0);
  const f3 color = clamp3(matColor(m) * sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s), 0.0f, 1.0f);
  const float cosPower = cosPowerFromGlosiness(phongGlosiness(m, tc, s));
  const f3 r = reflect3(v * (-1.0f), n);
  const float cosAlpha = clampf(dot(l, r), 0.0f, 1.0f);
  const float fix = (matFlags(m) & HMF_ENERGY_FIX) ? (cosAlpha / fmaxf(dot(n, l), 1e-6f)) : 1.0f;
  return (((color * (cosPower + 2.0f)) * HK_INV_TWOPI) * powf(cosAlpha, cosPower)) * fix;
}
HK_DEV void PhongSampleAndEvalBRDF(const float* m, float r1, float r2, f3 ray_dir, f3 n, f2 tc, const SceneDev& s, MatSample& out) {
  const f3 color = clamp3(matColor(m) * sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s), 0.0f, 1.0f);
  const float gloss = phongGlosiness(m, tc, s);
  const float cosPower = cosPowerFromGlosiness(gloss);
  bool under = false;
  const f3 r = reflect3(ray_dir, n);
  const f3 newDir = MapSampleToModifiedCosineDistribution(r1, r2, r, n, cosPower, under);
  const f3 v = ray_dir * (-1.0f);
  if (dot(n, v) < 1e-6f || dot(n, newDir) < 1e-6f || under) { out.color = mk3(0, 0, 0); out.pdf = 1.0f; }
  else {
    const float cosAlpha = clampf(dot(newDir, r), 0.0f, 1.0f);
    const float eqTemp = powf(cosAlpha, cosPower) * HK_INV_TWOPI;
    const float fix = (matFlags(m) & HMF_ENERGY_FIX) ? (cosAlpha / fmaxf(dot(n, newDir), 1e-6f)) : 1.0f;
    out.pdf = eqTemp * (cosPower + 1.0f);
    out.color = (color * (eqTemp * (cosPower + 2.0f))) * fix;
  }
  out.direction = newDir;
  out.flags = (gloss >= 0.99f) ? HRE_S : HRE_G;
}
// ---- mirror, cmaterial.h:395-430
// ---- Blinn distribution in a Torrance-Sparrow microfacet model (brdf_type "torranse_sparrow"), cmaterial.h:1020-1168, cmatpbrt.h:33-103;
// colour, gloss and the two samplers sit at the phong offsets
HK_DEV float TorranceSparrowG1(f3 wo, f3 wi, f3 wh) {   // PBRT frame
  const float NdotWh = fabsf(wh.z), NdotWo = fabsf(wo.z), NdotWi = fabsf(wi.z);
  const float WOdotWh = fmaxf(fabsf(dot(wo, wh)), HK_DEPSILON);
  return fminf(1.f, fminf((2.f * NdotWh * NdotWo / WOdotWh), (2.f * NdotWh * NdotWi / WOdotWh)));
}
HK_DEV float TorranceSparrowGF1(f3 wo, f3 wi) {   // PBRT frame; the Fresnel factor of the reference is the constant 1
  const float cosThetaO = fabsf(wo.z), cosThetaI = fabsf(wi.z);
  if (cosThetaI == 0.0f || cosThetaO == 0.0f) return 0.0f;
  f3 wh = wi + wo;
  if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return 0.0f;
  wh = normalize(wh);
  return fminf(TorranceSparrowG1(wo, wi, wh) * 1.0f / fmaxf(4.0f * cosThetaI * cosThetaO, HK_DEPSILON), 250.0f);
}
HK_DEV float TorranceSparrowGF2(f3 wo, f3 wi, f3 n) {   // world frame
  const float cosThetaO = fabsf(dot(wo, n)), cosThetaI = fabsf(dot(wi, n));
  if (cosThetaI == 0.f || cosThetaO == 0.0f) return 0.0f;
  f3 wh = wi + wo;
  if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return 0.0f;
  wh = normalize(wh);
  const float NdotWh = fabsf(dot(wh, n)), WOdotWh = fmaxf(fabsf(dot(wo, wh)), HK_DEPSILON);
  const float G = fminf(1.f, fminf((2.f * NdotWh * cosThetaO / WOdotWh), (2.f * NdotWh * cosThetaI / WOdotWh)));
  return fminf(G * 1.0f / fmaxf(4.0f * cosThetaI * cosThetaO, HK_DEPSILON), 10.0f);
}
HK_DEV float blinnEvalPDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const SceneDev& s) {
  if (dot(n, v) < 1e-6f || dot(n, l) < 1e-6f) return 1.0f;
  const float exponent = cosPowerFromGlosiness(phongGlosiness(m, tc, s));
  const f3 wh = normalize(l + v);
  const float costheta = fabsf(dot(wh, n));
  return ((exponent + 1.0f) * powf(costheta, exponent)) / (HK_TWOPI * 4.0f * dot(l, wh));
}
HK_DEV f3 blinnEvalBxDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const SceneDev& s) {
  if (dot(n, v) < 1e-6f || dot(n, l) < 1e-6f) return mk3(0, 0, 0);
  const f3 color = clamp3(matColor(m) * sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s), 0.0f, 1.0f);
  const float exponent = cosPowerFromGlosiness(phongGlosiness(m, tc, s));
  const f3 wh = normalize(l + v);
  const float D = (exponent + 2.0f) * HK_INV_TWOPI * powf(fabsf(dot(wh, n)), exponent);
  return (color * D) * TorranceSparrowGF2(l, v, n);
}
HK_DEV void BlinnSampleAndEvalBRDF(const float* m, float r1, float r2, f3 ray_dir, f3 n, f2 tc, const SceneDev& s, MatSample& out) {
  const f3 color = clamp3(matColor(m) * sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s), 0.0f, 1.0f);
  const float gloss = phongGlosiness(m, tc, s);
  f3 nx, ny;
  const f3 nz = n;
  CoordinateSystem(nz, nx, ny);
  const f3 wo = mk3(-dot(ray_dir, nx), -dot(ray_dir, ny), -dot(ray_dir, nz));
  const float exponent = cosPowerFromGlosiness(gloss);
  const float costheta = powf(r1, 1.0f / (exponent + 1.0f));
  const float sintheta = sqrtf(fmaxf(0.0f, 1.0f - costheta * costheta));
  const float phi = r2 * HK_TWOPI;
  const f3 wh = mk3(sintheta * cosf(phi), sintheta * sinf(phi), costheta);   // SphericalDirectionPBRT
  const f3 wi = (wh * (2.0f * dot(wo, wh))) - wo;
  const f3 newDir = normalize(((nx * wi.x) + (ny * wi.y)) + (nz * wi.z));
  const f3 v = ray_dir * (-1.0f);
  if (dot(n, v) < 1e-6f || dot(n, newDir) < 1e-6f) { out.color = mk3(0, 0, 0); out.pdf = 1.0f; }
  else {
    const float D = ((exponent + 2.0f) * HK_INV_TWOPI * powf(costheta, exponent));
    out.color = (color * D) * TorranceSparrowGF1(wo, wi);
    out.pdf = ((exponent + 1.0f) * powf(costheta, exponent)) / fmaxf(HK_TWOPI * 4.0f * dot(wo, wh), HK_DEPSILON);
  }
  out.direction = newDir;
  out.flags = (gloss >= 0.99f) ? HRE_S : HRE_G;
}

// ---- anisotropic microfacet lobes: Beckmann and Trowbridge-Reitz (GGX) distributions as in PBRT v3 (cmatpbrt.h:105-540) behind the material
// wrappers of cmaterial.h:1558-1846; both node classes share one layout (BECKMANN_* offsets, :1531-1556): colour 10..12, colour sampler ids
// 13/14, glossiness 16 with sampler ids 17/18, anisotropy 19, rotation 68, anisotropy sampler ids 69/70, rotation sampler ids 71/72
HK_DEV float Cos2ThetaPBRT(f3 w) { return w.z * w.z; }
HK_DEV float AbsCosThetaPBRT(f3 w) { return fabsf(w.z); }
HK_DEV float Sin2ThetaPBRT(f3 w) { return fmaxf(0.0f, 1.0f - Cos2ThetaPBRT(w)); }
HK_DEV float SinThetaPBRT(f3 w) { return sqrtf(Sin2ThetaPBRT(w)); }
HK_DEV float TanThetaPBRT(f3 w) { return (fabsf(w.z) < 1e-6f) ? 0.0f : SinThetaPBRT(w) / w.z; }
HK_DEV float Tan2ThetaPBRT(f3 w) { return Sin2ThetaPBRT(w) / fmaxf(Cos2ThetaPBRT(w), 1e-6f); }
HK_DEV float CosPhiPBRT(f3 w) { const float st = SinThetaPBRT(w); return (st == 0.0f) ? 1.0f : clampf(w.x / st, -1.0f, 1.0f); }
HK_DEV float SinPhiPBRT(f3 w) { const float st = SinThetaPBRT(w); return (st == 0.0f) ? 0.0f : clampf(w.y / st, -1.0f, 1.0f); }
HK_DEV float Cos2PhiPBRT(f3 w) { return CosPhiPBRT(w) * CosPhiPBRT(w); }
HK_DEV float Sin2PhiPBRT(f3 w) { return SinPhiPBRT(w) * SinPhiPBRT(w); }
HK_DEV float ErfPBRT(float x) {   // cmatpbrt.h:139-160
  const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
  int sign = 1;
  if (x < 0.0f) sign = -1;
  x = fabsf(x);
  const float t = 1.0f / (1.0f + p * x);
  const float y = 1.0f - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * expf(-x * x);
  return float(sign) * y;
}
HK_DEV float ErfInvPBRT(float x) {   // cmatpbrt.h:162-193
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
HK_DEV float BeckmannDistributionD(f3 wh, float ax, float ay) {
  const float tan2Theta = Tan2ThetaPBRT(wh), cos4Theta = Cos2ThetaPBRT(wh) * Cos2ThetaPBRT(wh);
  return expf((-1.0f) * tan2Theta * (Cos2PhiPBRT(wh) / fmaxf(ax * ax, 1e-6f) + Sin2PhiPBRT(wh) / fmaxf(ay * ay, 1e-6f))) / fmaxf(HK_PI * ax * ay * cos4Theta, 1e-6f);
}
HK_DEV float BeckmannDistributionLambda(f3 w, float ax, float ay) {
  const float absTanTheta = fabsf(TanThetaPBRT(w));
  if (!isfinite(absTanTheta) || absTanTheta == 0.0f) return 0.0f;
  const float alpha = sqrtf(fmaxf(Cos2PhiPBRT(w) * ax * ax + Sin2PhiPBRT(w) * ay * ay, 1e-6f));
  const float a = 1.0f / fmaxf(alpha * absTanTheta, 1e-6f);
  if (a >= 1.6f) return 0.0f;
  return (1.0f - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
}
HK_DEV void BeckmannSample11(float cosThetaI, float U1, float U2, float& slope_x, float& slope_y) {   // cmatpbrt.h:219-295
  if (cosThetaI > 0.9999f) {
    const float r = sqrtf(logf(1.0f - U1) * (-1.0f));
    const float sinPhi = sinf(HK_TWOPI * U2), cosPhi = cosf(HK_TWOPI * U2);
    slope_x = r * cosPhi; slope_y = r * sinPhi;
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
  const float SQRT_PI_INV = 1.0f / sqrtf(HK_PI);
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
  slope_x = ErfInvPBRT(b);
  slope_y = ErfInvPBRT(2.0f * fmaxf(U2, 1e-6f) - 1.0f);
}
HK_DEV void TrowbridgeReitzSample11(float cosTheta, float U1, float U2, float& slope_x, float& slope_y) {   // cmatpbrt.h:397-448
  if (cosTheta > 0.9999f) {
    const float r = sqrtf(U1 / fmaxf(1.0f - U1, 1e-6f));
    const float phi = HK_TWOPI * U2;
    slope_x = r * cosf(phi); slope_y = r * sinf(phi);
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
  slope_x = (A < 0.0f || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
  float S;
  if (U2 > 0.5f) { S = 1.0f; U2 = 2.0f * (U2 - 0.5f); } else { S = -1.0f; U2 = 2.0f * (0.5f - U2); }
  const float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) / (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
  slope_y = S * z * sqrtf(1.0f + slope_x * slope_x);
}
// BeckmannSample / TrowbridgeReitzSample (:297-317, :450-472) and ...DistributionSampleWH (:319-327, :474-482)
template <bool TR>
HK_DEV f3 microfacetSampleWH(f3 wo, float u1, float u2, float ax, float ay) {
  const bool flip = (wo.z < 0.0f);
  const f3 wi = flip ? wo * (-1.0f) : wo;
  const f3 wiStretched = normalize(mk3(ax * wi.x, ay * wi.y, wi.z));
  float slope_x, slope_y;
  if (TR) TrowbridgeReitzSample11(wiStretched.z, u1, u2, slope_x, slope_y); else BeckmannSample11(wiStretched.z, u1, u2, slope_x, slope_y);
  const float tmp = CosPhiPBRT(wiStretched) * slope_x - SinPhiPBRT(wiStretched) * slope_y;
  slope_y = SinPhiPBRT(wiStretched) * slope_x + CosPhiPBRT(wiStretched) * slope_y;
  slope_x = tmp;
  slope_x = ax * slope_x;
  slope_y = ay * slope_y;
  f3 wh = normalize(mk3(slope_x * (-1.0f), slope_y * (-1.0f), 1.0f));
  if (flip) wh = wh * (-1.0f);
  return wh;
}
HK_DEV float TrowbridgeReitzDistributionD(f3 wh, float ax, float ay) {
  const float tan2Theta = Tan2ThetaPBRT(wh);
  if (!isfinite(tan2Theta)) return 0.0f;
  const float cos4Theta = Cos2ThetaPBRT(wh) * Cos2ThetaPBRT(wh);
  const float e = (Cos2PhiPBRT(wh) / (ax * ax) + Sin2PhiPBRT(wh) / (ay * ay)) * tan2Theta;
  return 1.0f / (HK_PI * ax * ay * cos4Theta * (1.0f + e) * (1.0f + e));
}
HK_DEV float TrowbridgeReitzDistributionLambda(f3 w, float ax, float ay) {
  const float absTanTheta = fabsf(TanThetaPBRT(w));
  if (!isfinite(absTanTheta)) return 0.0f;
  const float alpha = sqrtf(Cos2PhiPBRT(w) * ax * ax + Sin2PhiPBRT(w) * ay * ay);
  const float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
  return (-1.0f + sqrtf(1.0f + alpha2Tan2Theta)) / 2.0f;
}
template <bool TR> HK_DEV float microfacetD(f3 wh, float ax, float ay) { return TR ? TrowbridgeReitzDistributionD(wh, ax, ay) : BeckmannDistributionD(wh, ax, ay); }
template <bool TR> HK_DEV float microfacetLambda(f3 w, float ax, float ay) { return TR ? TrowbridgeReitzDistributionLambda(w, ax, ay) : BeckmannDistributionLambda(w, ax, ay); }
template <bool TR> HK_DEV float microfacetPdf(f3 wo, f3 wh, float ax, float ay) {   // ...DistributionPdf, :335-338, :490-493
  return microfacetD<TR>(wh, ax, ay) * (1.0f / (1.0f + microfacetLambda<TR>(wo, ax, ay))) / fmaxf(4.0f * AbsCosThetaPBRT(wo), 1e-6f);
}
template <bool TR> HK_DEV float microfacetBRDF_PBRT(f3 wo, f3 wi, float ax, float ay) {   // BeckmannBRDF_PBRT :351-371, TrowbridgeReitzBRDF_PBRT :506-526
  const float cosThetaO = AbsCosThetaPBRT(wo), cosThetaI = AbsCosThetaPBRT(wi);
  f3 wh = wi + wo;
  if (cosThetaI <= 1e-6f || cosThetaO <= 1e-6f) return 0.0f;
  if (fabsf(wh.x) <= 1e-6f && fabsf(wh.y) <= 1e-6f && fabsf(wh.z) <= 1e-6f) return 0.0f;
  wh = normalize(wh);
  const float G = 1.0f / (1.0f + microfacetLambda<TR>(wo, ax, ay) + microfacetLambda<TR>(wi, ax, ay));
  if (TR) return microfacetD<TR>(wh, ax, ay) * G / fmaxf(4.0f * cosThetaI * cosThetaO, 1e-6f);
  return microfacetD<TR>(wh, ax, ay) * G * 1.0f / fmaxf(4.0f * cosThetaI * cosThetaO, 1e-6f);   // the Beckmann form carries its F = 1 factor
}
HK_DEV float BeckmannRoughnessToAlpha(float roughness) {   // cmatpbrt.h:340-344
  const float x = logf(fmaxf(roughness, 1.0e-4f));
  return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
HK_DEV float beckmannGlosiness(const float* m, f2 tc, const SceneDev& s) {   // cmaterial.h:1568-1581 (offsets = phong's)
  return phongGlosiness(m, tc, s);
}
HK_DEV f2 beckmannAlphaXY(const float* m, f2 tc, const SceneDev& s) {   // :1583-1611
  const float roughness = 0.5f - 0.5f * beckmannGlosiness(m, tc, s);
  const f3 ac = sample2DExt(as_int(m[HM_BECKMANN_ANISO_TEXMATRIXID]), tc, m, s);
  const float anisoMult = 1.0f - clampf(m[HM_BECKMANN_ANISOTROPY] * fmaxf(ac.x, fmaxf(ac.y, ac.z)), 0.0f, 1.0f);
  return mk2(BeckmannRoughnessToAlpha(roughness * roughness), BeckmannRoughnessToAlpha(roughness * roughness * anisoMult * anisoMult));
}
HK_DEV void BeckmanTangentSpace(const float* m, f2 alpha, f3 nz, f3 a_tan, f3 a_bitan, f2 tc, const SceneDev& s, f3& nx, f3& ny) {   // :1613-1635
  if (fabsf(alpha.x - alpha.y) > 1e-5f) {
    nx = a_bitan; ny = a_tan;
    const f3 rc = sample2DExt(as_int(m[HM_BECKMANN_ROT_TEXMATRIXID]), tc, m, s);
    const float rotVal = clampf(m[HM_BECKMANN_ANISO_ROT] * fmaxf(rc.x, fmaxf(rc.y, rc.z)), 0.0f, 1.0f);
    const float rotAngle = rotVal * HK_TWOPI;
    const float cos_t = cosf(rotAngle), sin_t = sinf(rotAngle);
    const f3 v = nz;   // RotateAroundVector4x4, cglobals.h:1122-1150, applied with mul3x3
    const f3 r0 = mk3((1.0f - cos_t) * v.x * v.x + cos_t, (1.0f - cos_t) * v.x * v.y - sin_t * v.z, (1.0f - cos_t) * v.x * v.z + sin_t * v.y);
    const f3 r1 = mk3((1.0f - cos_t) * v.y * v.x + sin_t * v.z, (1.0f - cos_t) * v.y * v.y + cos_t, (1.0f - cos_t) * v.y * v.z - sin_t * v.x);
    const f3 r2 = mk3((1.0f - cos_t) * v.x * v.z - sin_t * v.y, (1.0f - cos_t) * v.z * v.y + sin_t * v.x, (1.0f - cos_t) * v.z * v.z + cos_t);
    const f3 px = nx, py = ny;
    nx = mk3(r0.x * px.x + r0.y * px.y + r0.z * px.z, r1.x * px.x + r1.y * px.y + r1.z * px.z, r2.x * px.x + r2.y * px.y + r2.z * px.z);
    ny = mk3(r0.x * py.x + r0.y * py.y + r0.z * py.z, r1.x * py.x + r1.y * py.y + r1.z * py.z, r2.x * py.x + r2.y * py.y + r2.z * py.z);
  } else
    CoordinateSystem(nz, nx, ny);
  if ((matFlags(m) & HMF_FLIP_TANGENT) != 0) { const f3 t = nx; nx = ny; ny = t; }
}
template <bool TR>
HK_DEV float anisoEvalPDF(const float* m, f3 l, f3 v, f3 n, f3 a_tan, f3 a_bitan, f2 tc, const SceneDev& s) {   // beckmannEvalPDF :1637-1660, trggxEvalPDF :1743-1767
  if (dot(n, v) < 1e-6f || dot(n, l) < 1e-6f) return 1.0f;
  const f2 alpha = beckmannAlphaXY(m, tc, s);
  f3 nx, ny;
  BeckmanTangentSpace(m, alpha, n, a_tan, a_bitan, tc, s, nx, ny);
  const f3 wo = mk3(-dot(v, nx), -dot(v, ny), -dot(v, n));
  const f3 wh = normalize(l + v);
  return microfacetPdf<TR>(wo, wh, alpha.x, alpha.y);
}
template <bool TR>
HK_DEV f3 anisoEvalBxDF(const float* m, f3 l, f3 v, f3 n, f3 a_tan, f3 a_bitan, f2 tc, const SceneDev& s) {   // :1662-1687, :1769-1794
  if (dot(n, v) < 1e-6f || dot(n, l) < 1e-6f) return mk3(0, 0, 0);
  const f3 color = clamp3(sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s) * matColor(m), 0.0f, 1.0f);
  const f2 alpha = beckmannAlphaXY(m, tc, s);
  f3 nx, ny;
  BeckmanTangentSpace(m, alpha, n, a_tan, a_bitan, tc, s, nx, ny);
  const f3 wo = mk3(-dot(v, nx), -dot(v, ny), -dot(v, n)), wi = mk3(-dot(l, nx), -dot(l, ny), -dot(l, n));
  return color * microfacetBRDF_PBRT<TR>(wo, wi, alpha.x, alpha.y);
}
template <bool TR>
HK_DEV void AnisoSampleAndEvalBRDF(const float* m, float r1, float r2, f3 ray_dir, f3 a_normal, f2 tc, f3 a_tan, f3 a_bitan, const SceneDev& s, MatSample& out) {   // :1689-1731, :1796-1838
  const f3 color = clamp3(sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s) * matColor(m), 0.0f, 1.0f);
  const f2 alpha = beckmannAlphaXY(m, tc, s);
  const float gloss = beckmannGlosiness(m, tc, s);
  f3 nx, ny;
  const f3 nz = a_normal;
  BeckmanTangentSpace(m, alpha, nz, a_tan, a_bitan, tc, s, nx, ny);
  const f3 wo = mk3(-dot(ray_dir, nx), -dot(ray_dir, ny), -dot(ray_dir, nz));
  const f3 wh = microfacetSampleWH<TR>(wo, r1, r2, alpha.x, alpha.y);
  const f3 wi = (wh * (2.0f * dot(wo, wh))) - wo;
  const f3 newDir = normalize(((nx * wi.x) + (ny * wi.y)) + (nz * wi.z));
  const f3 v = ray_dir * (-1.0f), l = newDir;
  if (dot(a_normal, v) < 1e-6f || dot(a_normal, l) < 1e-6f) { out.color = mk3(0, 0, 0); out.pdf = 1.0f; }
  else { out.color = color * microfacetBRDF_PBRT<TR>(wo, wi, alpha.x, alpha.y); out.pdf = microfacetPdf<TR>(wo, wh, alpha.x, alpha.y); }
  out.direction = newDir;
  out.flags = (gloss >= 0.99f) ? HRE_S : HRE_G;
}
HK_DEV void MirrorSampleAndEvalBRDF(const float* m, f3 ray_dir, f3 n, f2 tc, const SceneDev& s, MatSample& out) {
  const f3 tex = sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s);
  f3 newDir = reflect3(ray_dir, n);
  if (dot(ray_dir, n) > 0.0f) newDir = ray_dir;
  const float cosOut = dot(newDir, n);
  out.direction = newDir;
  out.pdf = 1.0f;
  out.color = (matColor(m) * tex) * (1.0f / fmaxf(cosOut, 1e-6f));
  if (cosOut <= 1e-6f) out.color = mk3(0, 0, 0);
  out.flags = HRE_S;
}
// ---- multi-scattering energy tables of the globals header (cfetch.h:78-79), cmaterial.h:61-196.  Read only when a node carries
// PLAIN_MATERIAL_ENERGY_FIX_OR_MULTISCATTER; whoever assembles the blob fills them (IHWLayerDataAssembler.cpp via cfetch.h:84-93).
HK_DEV float BilinearFrom2dTable(const uint16_t* t, float x, float y, int w, int h) {   // :61-90
  x = clampf(x, 0.0f, float(w) - 1.0001f);
  y = clampf(y, 0.0f, float(h) - 1.0001f);
  const int fy = int(floorf(y)), fx = int(floorf(x));
  const int d1 = fy * w + fx, d2 = d1 + 1, d3 = (fy + 1) * w + fx, d4 = d3 + 1;
  const float dx = x - float(fx), dy = y - float(fy);
  const float m1 = (1.0f - dx) * (1.0f - dy), m2 = dx * (1.0f - dy), m3 = dy * (1.0f - dx), m4 = dx * dy;
  if (fy >= 0 && fx >= 0 && fy <= h - 2 && fx <= w - 2) return float(t[d1]) * m1 + float(t[d2]) * m2 + float(t[d3]) * m3 + float(t[d4]) * m4;
  return 1.0f;
}
HK_DEV float BilinearFrom3dTable(const uint16_t* t, float x, float y, float z, int w, int h, int d, int size2d) {   // :93-150
  x = clampf(x, 0.0f, float(w) - 1.0001f);
  y = clampf(y, 0.0f, float(h) - 1.0001f);
  z = clampf(z, 0.0f, float(d) - 1.0001f);
  const int fx = int(floorf(x)), fy = int(floorf(y)), fz = int(floorf(z));
  const int zo = fz * size2d, zo2 = (fz + 1) * size2d;
  const int d1 = zo + fy * w + fx, d3 = zo + (fy + 1) * w + fx, d2 = d1 + 1, d4 = d3 + 1;
  const int d5 = zo2 + fy * w + fx, d7 = zo2 + (fy + 1) * w + fx, d6 = d5 + 1, d8 = d7 + 1;
  const float dx = x - float(fx), dy = y - float(fy), dz = z - float(fz);
  const float m1 = (1.0f - dx) * (1.0f - dy), m2 = dx * (1.0f - dy), m3 = dy * (1.0f - dx), m4 = dx * dy;
  if (fy >= 0 && fx >= 0 && fy <= h - 2 && fx <= w - 2) {
    const float p1 = float(t[d1]) * m1 + float(t[d2]) * m2 + float(t[d3]) * m3 + float(t[d4]) * m4;
    const float p2 = float(t[d5]) * m1 + float(t[d6]) * m2 + float(t[d7]) * m3 + float(t[d8]) * m4;
    return p1 + dz * (p2 - p1);
  }
  return 1.0f;
}
HK_DEV f3 multiscatterFromEss(float Ess, f3 color) {
  const float a = 1.0f - Ess, b = fmaxf(Ess, 1e-6f);
  return mk3(1.0f + (color.x * a) / b, 1.0f + (color.y * a) / b, 1.0f + (color.z * a) / b);
}
HK_DEV f3 GetMultiscatteringFrom2dTable(const SceneDev& s, float roughness, float dotNV, f3 color) {   // :152-159, 64 x 64
  const uint16_t* t = reinterpret_cast<const uint16_t*>(s.globals + HG_ESS_GGX_TABLE);
  return multiscatterFromEss(BilinearFrom2dTable(t, dotNV * 64.0f, roughness * 64.0f, 64, 64) * (1.0f / 65535.0f), color);
}
HK_DEV f3 GetMultiscatteringFrom3dTable(const SceneDev& s, float roughness, float dotNV, float ior, f3 color) {   // :161-196, 64^3
  if (!(ior >= 0.4166f && ior <= 2.4f)) return mk3(1, 1, 1);
  const uint16_t* t = reinterpret_cast<const uint16_t*>(s.globals + HG_ESS_TRANSP_TABLE);
  const float iorNormal = (ior - 0.4166f) / (2.4f - 0.4166f);
  return multiscatterFromEss(BilinearFrom3dTable(t, dotNV * 64.0f, roughness * 64.0f, iorNormal * 64.0f, 64, 64, 64, 64 * 64) * (1.0f / 65536.0f), color);
}
// ---- thin glass, cmaterial.h:472-556.  Eval is zero (:512-520): shadow rays never sample it.
HK_DEV float transpGloss(const float* m, int glossMult, int glossTexMatrixId, f2 tc, const SceneDev& s) {   // thinglassCosPower :496-505, glassGloss :610-618
  const f3 g = sample2DExt(as_int(m[glossTexMatrixId]), tc, m, s);
  return clampf(m[glossMult] * fmaxf(g.x, fmaxf(g.y, g.z)), 0.0f, 1.0f);
}
HK_DEV void ThinglassSampleAndEvalBRDF(const float* m, float r1, float r2, f3 ray_dir, f3 n, f2 tc, const SceneDev& s, MatSample& out) {   // :522-556
  const f3 tex = sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s);
  const float cosPower = cosPowerFromGlosiness(transpGloss(m, HM_THINGLASS_GLOSINESS, HM_THINGLASS_GLOSS_TEXMATRIXID, tc, s));
  float pdf = 1.0f, fVal = 1.0f;
  if (cosPower < 1e6f) {
    bool under = false;
    const f3 oldDir = ray_dir;
    ray_dir = MapSampleToModifiedCosineDistribution(r1, r2, ray_dir, n * (-1.0f), cosPower, under);
    const float cosTheta = clampf(dot(oldDir, ray_dir), 0.0f, (HK_PI * 0.499995f));
    fVal = (cosPower + 2.0f) * HK_INV_TWOPI * powf(cosTheta, cosPower);
    if (under) fVal = 0.0f;
    pdf = powf(cosTheta, cosPower) * (cosPower + 1.0f) * (0.5f * HK_INV_PI);
  }
  const float cosOut = dot(ray_dir, n);
  const float cosMult = 1.0f / fmaxf(fabsf(cosOut), 1e-6f);
  out.direction = ray_dir;
  out.pdf = pdf;
  out.color = ((matColor(m) * fVal) * tex) * cosMult;
  if (cosOut >= -1e-6f) out.color = mk3(0, 0, 0);   // transparency must leave on the far side
  out.flags = (HRE_S | HRE_T | HRE_THINGLASS);
}
// ---- glass, cmaterial.h:566-867: the GGX form the reference dispatches to (:2298-2301)
struct RefractResult { f3 ray_dir; bool success; float eta; };   // :634-641
HK_DEV RefractResult myRefractGgx(f3 ray_dir, f3 n, float matIOR, float outsideIOR) {   // :684-714
  RefractResult res;
  res.eta = outsideIOR / matIOR;
  float cosTheta = dot(n, ray_dir) * (-1.0f);
  if (cosTheta < 0.0f) { cosTheta = cosTheta * (-1.0f); n = n * (-1.0f); res.eta = 1.0f / res.eta; }
  const float dotVN = cosTheta * (-1.0f);
  const float k = 1.0f - res.eta * res.eta * (1.0f - cosTheta * cosTheta);
  if (k > 0.0f) {
    res.ray_dir = normalize((ray_dir * res.eta) + (n * (res.eta * cosTheta - sqrtf(k))));
    res.success = true;
  } else {
    res.ray_dir = normalize((n * (dotVN * (-2.0f))) + ray_dir);
    res.success = false;
    res.eta = 1.0f;
  }
  return res;
}
HK_DEV float SmithGGXMasking(float dotNV, float roughSqr) {   // :1214-1218
  const float denomC = sqrtf(roughSqr + (1.0f - roughSqr) * dotNV * dotNV) + dotNV;
  return 2.0f * dotNV / fmaxf(denomC, 1e-6f);
}
HK_DEV float SmithGGXMaskingShadowing(float dotNL, float dotNV, float roughSqr) {   // :1247-1252
  const float denomA = dotNV * sqrtf(roughSqr + (1.0f - roughSqr) * dotNL * dotNL);
  const float denomB = dotNL * sqrtf(roughSqr + (1.0f - roughSqr) * dotNV * dotNV);
  return 2.0f * dotNL * dotNV / fmaxf(denomA + denomB, 1e-6f);
}
HK_DEV f3 GgxVndf(f3 wo, float roughness, float u1, float u2) {   // :1220-1245 (Heitz 2017, as the reference wrote it)
  const f3 v = normalize(mk3(wo.x * roughness, wo.y * roughness, wo.z));
  const f3 t1 = (v.z < 0.999f) ? normalize(cross(v, mk3(0, 0, 1))) : mk3(1, 0, 0);
  const f3 t2 = cross(t1, v);
  const float a = 1.0f / (1.0f + v.z);
  const float r = sqrtf(u1);
  const float phi = (u2 < a) ? (u2 / a) * HK_PI : HK_PI + (u2 - a) / (1.0f - a) * HK_PI;
  const float p1 = r * cosf(phi);
  const float p2 = r * sinf(phi) * ((u2 < a) ? 1.0f : v.z);
  const f3 n = ((t1 * p1) + (t2 * p2)) + (v * sqrtf(fmaxf(0.0f, 1.0f - p1 * p1 - p2 * p2)));
  return normalize(mk3(roughness * n.x, roughness * n.y, fmaxf(0.0f, n.z)));
}
HK_DEV void GlassGGXSampleAndEvalBRDF(const float* m, const float* rands, f3 ray_dir, f3 n, f2 tc, bool hitFromInside, const SceneDev& s, MatSample& out, bool isFwdDir = false) {   // :775-882
  const f3 tex = sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s);
  const f3 color = clamp3(matColor(m) * tex, 0.0f, 1.0f);
  const float gloss = transpGloss(m, HM_GLASS_GLOSINESS, HM_GLASS_GLOSS_TEXMATRIXID, tc, s);
  const float roughness = clampf(1.0f - gloss, 0.0f, 1.0f);
  const float roughSqr = roughness * roughness;
  const float IOR = m[HM_GLASS_IOR];
  const f3 normal2 = hitFromInside ? n * (-1.0f) : n;
  bool spec = true;
  float Pss = 1.0f;
  f3 Pms = mk3(1, 1, 1);
  out.pdf = 1.0f;
  RefractResult refr = myRefractGgx(ray_dir, normal2, IOR, 1.0f);
  if (gloss < 0.999f) {
    spec = false;
    float eta = 1.0f / IOR;
    const float cosTheta = dot(normal2, ray_dir) * (-1.0f);
    if (cosTheta < 0.0f) eta = 1.0f / eta;
    f3 nx, ny;
    const f3 nz = n;
    CoordinateSystem(nz, nx, ny);
    const f3 wo = mk3(-dot(ray_dir, nx), -dot(ray_dir, ny), -dot(ray_dir, nz));
    const f3 wh = GgxVndf(wo, roughSqr, rands[0], rands[1]);
    const float dotWoWh = dot(wo, wh);
    f3 newDir;
    const float radicand = 1.0f + eta * eta * (dotWoWh * dotWoWh - 1.0f);
    if (radicand > 0.0f) { newDir = (wh * (eta * dotWoWh - sqrtf(radicand))) - (wo * eta); refr.success = true; refr.eta = eta; }
    else { newDir = (wh * (2.0f * dotWoWh)) - wo; refr.success = false; refr.eta = 1.0f; }
    refr.ray_dir = normalize(((nx * newDir.x) + (ny * newDir.y)) + (nz * newDir.z));
    const f3 v = ray_dir * (-1.0f);
    const float dotNV = fabsf(dot(n, v)), dotNL = fabsf(dot(n, refr.ray_dir));
    const float G1 = SmithGGXMasking(dotNV, roughSqr);
    const float G2 = SmithGGXMaskingShadowing(dotNL, dotNV, roughSqr);
    Pss = G2 / fmaxf(G1, 1e-6f);
    if (matFlags(m) & HMF_ENERGY_FIX) Pms = GetMultiscatteringFrom3dTable(s, roughness, dotNV, 1.0f / eta, color);   // :858-860
  }
  const float cosOut = dot(refr.ray_dir, n);
  const float cosMult = 1.0f / fmaxf(fabsf(cosOut), 1e-6f);
  out.direction = refr.ray_dir;
  const float adjointBtdfMult = isFwdDir ? 1.0f : (refr.eta * refr.eta);   // camera paths: radiance flows against the walk (:867-869)
  if (refr.success) out.color = (((color * adjointBtdfMult) * Pss) * Pms) * cosMult;
  else out.color = ((mk3(1, 1, 1) * Pss) * Pms) * cosMult;
  out.flags = spec ? (HRE_S | HRE_T) : (HRE_G | HRE_T);
  if (refr.success && cosOut >= -1e-6f) out.color = mk3(0, 0, 0);
  else if (!refr.success && cosOut < 1e-6f) out.color = mk3(0, 0, 0);
}
// ---- GGX reflection, cmaterial.h:1165-1212, 1285-1291, 1317-1381, 1454-1520 (the 2017 VNDF form the reference dispatches to)
HK_DEV float ggxGlosiness(const float* m, f2 tc, const SceneDev& s) {
  if (uint32_t(as_int(m[HM_GGX_GLOSS_TEXID])) != HYDRA_INVALID_TEXTURE) {
    const f3 g = sample2DExt(as_int(m[HM_GGX_GLOSS_TEXMATRIXID]), tc, m, s);
    return clampf(m[HM_GGX_GLOSINESS] * fmaxf(g.x, fmaxf(g.y, g.z)), 0.0f, 0.99f);
  }
  return m[HM_GGX_GLOSINESS];
}
HK_DEV float GGX_Distribution(float cosThetaNH, float alpha) {
  const float alpha2 = alpha * alpha;
  const float NH_sqr = clampf(cosThetaNH * cosThetaNH, 0.0f, 1.0f);
  const float den = NH_sqr * alpha2 + (1.0f - NH_sqr);
  return alpha2 / fmaxf(HK_PI * den * den, 1e-6f);
}
HK_DEV float ggx2EvalPDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const SceneDev& s) {
  const float dotNV = dot(n, v), dotNL = dot(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) return 1.0f;
  const float roughness = 1.0f - ggxGlosiness(m, tc, s);
  const float roughSqr = roughness * roughness;
  const f3 h = normalize(v + l);
  const float dotNH = dot(n, h), dotHV = dot(h, v);
  const float G1 = SmithGGXMasking(dotNV, roughSqr);
  const float D = GGX_Distribution(dotNH, roughSqr);
  const float Dv = D * G1 * dotHV / fmaxf(dotNV, 1e-6f);
  const float jacob = 1.0f / fmaxf(4.0f * dotHV, 1e-6f);
  return Dv * jacob;
}
HK_DEV f3 ggxEvalBxDF(const float* m, f3 l, f3 v, f3 n, f2 tc, const SceneDev& s) {
  const float dotNV = dot(n, v), dotNL = dot(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) return mk3(0, 0, 0);
  const f3 color = clamp3(matColor(m) * sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s), 0.0f, 1.0f);
  const float roughness = 1.0f - ggxGlosiness(m, tc, s);
  const float roughSqr = roughness * roughness;
  const f3 h = normalize(v + l);
  const float dotNH = dot(n, h);
  const float D = GGX_Distribution(dotNH, roughSqr);
  const float G = SmithGGXMaskingShadowing(dotNL, dotNV, roughSqr);
  const float Pss = D * G / fmaxf(4.0f * dotNV * dotNL, 1e-6f);
  f3 Pms = mk3(1, 1, 1);
  if (matFlags(m) & HMF_ENERGY_FIX) Pms = GetMultiscatteringFrom2dTable(s, roughness, dotNV, color);
  return (color * Pss) * Pms;
}
HK_DEV void GGXSample2AndEvalBRDF(const float* m, float r1, float r2, f3 ray_dir, f3 n, f2 tc, const SceneDev& s, MatSample& out) {
  const f3 color = clamp3(matColor(m) * sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s), 0.0f, 1.0f);
  const float gloss = ggxGlosiness(m, tc, s);
  const float roughness = 1.0f - gloss;
  const float roughSqr = roughness * roughness;
  f3 nx, ny;
  const f3 nz = n;
  CoordinateSystem(nz, nx, ny);
  float Pss = 1.0f;
  f3 Pms = mk3(1, 1, 1);
  const f3 wo = mk3(-dot(ray_dir, nx), -dot(ray_dir, ny), -dot(ray_dir, nz));
  const f3 wh = GgxVndf(wo, roughSqr, r1, r2);
  const f3 wi = (wh * (2.0f * dot(wo, wh))) - wo;
  const f3 newDir = normalize(((nx * wi.x) + (ny * wi.y)) + (nz * wi.z));
  const f3 v = ray_dir * (-1.0f), l = newDir;
  const float dotNV = dot(n, v), dotNL = dot(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) { Pss = 0.0f; out.pdf = 1.0f; }
  else {
    const f3 h = normalize(v + l);
    const float dotNH = dot(n, h), dotHV = dot(h, v);
    const float D = GGX_Distribution(dotNH, roughSqr);
    const float G1 = SmithGGXMasking(dotNV, roughSqr);
    const float G2 = SmithGGXMaskingShadowing(dotNL, dotNV, roughSqr);
    Pss = D * G2 / fmaxf(4.0f * dotNV, 1e-6f);
    const float Dv = D * G1 * dotHV / fmaxf(dotNV, 1e-6f);
    const float jacob = 1.0f / fmaxf(4.0f * dotHV, 1e-6f);
    out.pdf = Dv * jacob;
    if (matFlags(m) & HMF_ENERGY_FIX) Pms = GetMultiscatteringFrom2dTable(s, roughness, dotNV, color);
  }
  out.direction = newDir;
  const f3 c = (color * Pss) * Pms;
  const float d = fmaxf(dotNL, 1e-6f);
  out.color = mk3(c.x / d, c.y / d, c.z / d);
  out.flags = (gloss >= 0.99f) ? HRE_S : HRE_G;
}
// ---- blend mask, cmaterial.h:2008-2137; fresnel cglobals.h:1879-1926
HK_DEV float fresnelDielectric(float c1, float c2, float etaExt, float etaInt) {
  const float Rs = (etaExt * c1 - etaInt * c2) / (etaExt * c1 + etaInt * c2);
  const float Rp = (etaInt * c1 - etaExt * c2) / (etaInt * c1 + etaExt * c2);
  return (Rs * Rs + Rp * Rp) / 2.0f;
}
HK_DEV float fresnelReflectionCoeff(float cosTheta1, float etaExt, float etaInt) {
  if (cosTheta1 < 0.0f) { const float t = etaInt; etaInt = etaExt; etaExt = t; }
  const float sinTheta2 = etaExt / etaInt * sqrtf(fmaxf(0.0f, 1.0f - cosTheta1 * cosTheta1));
  if (sinTheta2 > 1.0f) return 1.0f;
  const float cosTheta2 = sqrtf(fmaxf(0.0f, 1.0f - sinTheta2 * sinTheta2));
  return fresnelDielectric(fabsf(cosTheta1), cosTheta2, etaInt, etaExt);
}
HK_DEV float hermiteSplineEvalT(float t, const float* points, const float* tangents, int numPoints) {
  int ps = int(t * float(numPoints - 1));
  if (ps == numPoints - 1) ps--;
  const int pe = ps + 1;
  const float tStart = float(ps) / float(numPoints - 1), tEnd = float(pe) / float(numPoints - 1);
  const float sx = fabsf(t - tStart) / (tEnd - tStart);
  const float s2 = sx * sx, s3 = s2 * sx;
  const float h1 = 2.0f * s3 - 3.0f * s2 + 1.0f, h2 = -2.0f * s3 + 3.0f * s2, h3 = s3 - 2.0f * s2 + sx, h4 = s3 - s2;
  return 1.0f - clampf(h1 * points[2 * ps + 1] + h2 * points[2 * pe + 1] + h3 * tangents[2 * ps + 1] + h4 * tangents[2 * pe + 1], 0.0f, 1.0f);
}
HK_DEV float blendMaskAlpha2(const float* m, f3 v, f3 n, f2 tc, const SceneDev& s) {
  const f3 lum1 = clamp3(sample2DExt(as_int(m[HM_TEXMATRIXID]), tc, m, s) * matColor(m), 0.0f, 1.0f);
  const int bflags = as_int(m[HM_BLEND_FLAGS]);
  float lum = (bflags & HBF_EXTRUSION_LUMINANCE) ? dot(mk3(0.2126f, 0.7152f, 0.0722f), lum1) : fmaxf(lum1.x, fmaxf(lum1.y, lum1.z));
  const float normAngle = fabsf(dot(v, n));
  float faloff = 0.0f;
  if (bflags & HBF_FALOFF) {
    const int start = as_int(m[HM_BLEND_FALOFF_OFFSET]), size = as_int(m[HM_BLEND_FALOFF_SIZE]);
    const float* points = reinterpret_cast<const float*>(s.globals + s.hdr[HG_FLOAT_ARRAYS_OFFS]) + start;
    const float param = (as_int(m[HM_BLEND_FLAGS2]) & 1) ? normAngle : 1.0f - normAngle;
    faloff = hermiteSplineEvalT(param, points, points + size / 2, size / 4);
  }
  if (as_int(m[HM_BLEND_TYPE]) == 4) {   // BLEND_SIGMOID; unreachable with converter-made blobs (sampler aliasing)
    const float x2 = -5.0f + 10.0f * lum;
    lum = 1.04f / (1.0f + expf(-m[HM_BLEND_SIGMOID_EXP] * x2)) - 0.02f;
  }
  if (bflags & HBF_FALOFF) return clampf(faloff, 0.0f, 1.0f);
  if (bflags & HBF_FRESNEL) return clampf(lum * fresnelReflectionCoeff(fabsf(normAngle), 1.0f, m[HM_BLEND_FRESNEL_IOR]), 0.0f, 1.0f);
  return clampf(lum, 0.0f, 1.0f);
}
struct BRDFSelector { float w; int localOffs; };
HK_DEV BRDFSelector blendSelectBRDF(const float* m, float r3, f3 rayDir, f3 n, f2 tc, bool reflOnly, const SceneDev& s) {
  float alpha = blendMaskAlpha2(m, rayDir, n, tc, s);
  BRDFSelector m1, m2;
  m1.localOffs = as_int(m[HM_BLEND_MAT1]); m2.localOffs = as_int(m[HM_BLEND_MAT2]);
  m1.w = 1.0f; m2.w = 1.0f;
  const int bflags = as_int(m[HM_BLEND_FLAGS]);
  const bool comp1IsLeaf = matType(m + size_t(m1.localOffs) * HM_NODE_FLOATS) != HMT_BLEND_MASK;
  if ((bflags & HBF_REFLECTION_WEIGHT_IS_ONE) && comp1IsLeaf) { m1.w = alpha; m2.w = 1.0f; }
  if ((bflags & HBF_FRESNEL) != 0 && reflOnly) { m1.w = alpha; alpha = 1.0f; }
  return (r3 <= alpha) ? m1 : m2;
}
HK_DEV bool isEyeRay(uint32_t flags) {   // cglobals.h:1366-1376
  const uint32_t other = flags >> 16;
  const bool nonSpec = (other & HRE_D) || (other & HRE_G);
  return (((flags >> 8) & 0xFFu) == 0) || !nonSpec;
}
// MaterialSampleAndEvalBxDF, cmaterial.h:2345-2371 (random walk :2180-2207, leaf dispatch :2245-2335)
template <int F = HK_FEAT_ALL>
HK_DEV void MaterialSampleAndEvalBxDF(const float* m, const float* rands, const SurfaceHit& sh, f3 rayDir, uint32_t rayFlags, const SceneDev& s, MatSample& out, bool isFwdDir = false) {
  const bool reflOnly = (((rayFlags >> 16) & 64u /*RAY_GRAMMAR_DIRECT_LIGHT*/) != 0) && ((matFlags(m) & HMF_CAN_SAMPLE_REFL_ONLY) != 0);
  float mixW = 1.0f;
  const float* node = m;
  for (int i = 0; matType(node) == HMT_BLEND_MASK && i < 7; i++) {
    const BRDFSelector sel = blendSelectBRDF(node, rands[3 + i], rayDir, sh.normal, sh.texCoord, reflOnly && (i == 0), s);
    mixW = mixW * sel.w;
    node = node + size_t(sel.localOffs) * HM_NODE_FLOATS;
  }
  out.color = mk3(0, 0, 0); out.direction = mk3(0, 1, 0); out.pdf = 1.0f; out.flags = 0;
  f3 hitNorm = sh.normal;
  const bool nmap = (F & HK_FEAT_NMAP) && hasNormalMap(node);
  if (nmap) {   // :2252-2259: the bumped normal is built on the flat normal (turned over when hit from inside, except for glass)
    const f3 flatNorm = (sh.hfi && matType(node) != HMT_GLASS) ? sh.flatNormal * (-1.0f) : sh.flatNormal;
    hitNorm = BumpMapping(sh.tangent, sh.biTangent, flatNorm, sh.texCoord, node, s);
  }
  switch (matType(node)) {
    case HMT_PHONG: PhongSampleAndEvalBRDF(node, rands[0], rands[1], rayDir, hitNorm, sh.texCoord, s, out); break;
    case HMT_MIRROR: MirrorSampleAndEvalBRDF(node, rayDir, hitNorm, sh.texCoord, s, out); break;
    case HMT_LAMBERT: LambertSampleAndEvalBRDF(node, rands[0], rands[1], hitNorm, sh.texCoord, s, out); break;
    case HMT_OREN_NAYAR: if (F & HK_FEAT_OREN_NAYAR) OrennayarSampleAndEvalBRDF(node, rands[0], rands[1], rayDir, hitNorm, sh.texCoord, s, out); break;
    case HMT_GGX: if (F & HK_FEAT_GGX) GGXSample2AndEvalBRDF(node, rands[0], rands[1], rayDir, hitNorm, sh.texCoord, s, out); break;
    case HMT_THIN_GLASS: if (F & HK_FEAT_GLASS) ThinglassSampleAndEvalBRDF(node, rands[0], rands[1], rayDir, hitNorm, sh.texCoord, s, out); break;
    case HMT_GLASS: if (F & HK_FEAT_GLASS) GlassGGXSampleAndEvalBRDF(node, rands, rayDir, hitNorm, sh.texCoord, sh.hfi, s, out, isFwdDir); break;
    case HMT_TRANSLUCENT: if (F & HK_FEAT_TRANSLUCENT) TranslucentSampleAndEvalBRDF(node, rands[0], rands[1], hitNorm, sh.texCoord, s, out); break;
    case HMT_BLINN: if (F & HK_FEAT_BLINN) BlinnSampleAndEvalBRDF(node, rands[0], rands[1], rayDir, hitNorm, sh.texCoord, s, out); break;
    case HMT_SHADOW_MATTE: {   // ShadowmatteSampleAndEvalBRDF, cmaterial.h:1929-1942.  The CPU integrator hands in the shadow value (0, 0, 0) (PT_Loop.cpp:240): a black
      // pass-through surface.  With a back-plate in the header (hk_shading.h, haveBackPlate) the surface is the OpenCL layer's shadow catcher instead: NextBounce hands in the
      // traced shadow of this bounce's light sample (material.cl:812, 897), so the ray goes on to the back-plate darkened by it.  The fused kernel does not know that value
      // yet: the sample is made with shadow 1 and marked, and the next bounce multiplies the throughput by the shadow ray's result (k_bounce; 0 or 1: the same bits).
      const bool catcher = (F & HK_FEAT_RARE_LIGHTS) && haveBackPlate(s);
      const float sv = catcher ? 1.0f : 0.0f;
      out.direction = rayDir; out.pdf = 1.0f; out.color = mk3(sv, sv, sv) * (1.0f / fmaxf(fabsf(dot(rayDir, hitNorm)), 1e-5f)); out.flags = HRE_S | HRE_T | (catcher ? HK_MATTE_PENDING : 0);
      break;
    }
    case HMT_BECKMANN: if (F & HK_FEAT_ANISO) AnisoSampleAndEvalBRDF<false>(node, rands[0], rands[1], rayDir, hitNorm, sh.texCoord, sh.tangent, sh.biTangent, s, out); break;
    case HMT_TRGGX: if (F & HK_FEAT_ANISO) AnisoSampleAndEvalBRDF<true>(node, rands[0], rands[1], rayDir, hitNorm, sh.texCoord, sh.tangent, sh.biTangent, s, out); break;
    default: break;
  }
  if (nmap) {   // :2322-2327: the caller multiplies by the cosine to the shading normal
    const float cosThetaOut1 = fabsf(dot(out.direction, sh.normal)), cosThetaOut2 = fabsf(dot(out.direction, hitNorm));
    out.color = out.color * (cosThetaOut2 / fmaxf(cosThetaOut1, HK_DEPSILON2));
  }
  if (out.pdf <= 0.0f) out.color = mk3(0, 0, 0);
  out.color = out.color * (1.0f / fmaxf(mixW, 0.015625f));
  if ((matFlags(node) & HMF_SKIP_SKY_PORTAL) && isEyeRay(rayFlags)) { out.color = mk3(1, 1, 1); out.pdf = 1.0f; }
}
// adjointBsdfShadeNormalFix, cmaterial.h:2398-2417 (Veach 5.3.2: shading normals make light tracing non-symmetric)
HK_DEV float adjointBsdfShadeNormalFix(f3 toLightWo, f3 toCamWi, f3 shadeNorm, f3 geomNorm, float maxVal) {
  if (dot(shadeNorm, geomNorm) < 0) geomNorm = geomNorm * (-1.0f);
  if (1.0f - fabsf(dot(shadeNorm, geomNorm)) <= 1e-6f) return 1.0f;
  else if (dot(toCamWi, geomNorm) * dot(toCamWi, shadeNorm) <= 0 || dot(toLightWo, geomNorm) * dot(toLightWo, shadeNorm) <= 0) return 1.0f;
  const float k1 = dot(toLightWo, shadeNorm), k2 = dot(toCamWi, geomNorm), k3 = dot(toLightWo, geomNorm), k4 = dot(toCamWi, shadeNorm);
  const float res = (k1 * k2) / fmaxf(k3 * k4, HK_DEPSILON2);
  return fminf(fmaxf(res, 0.1f), maxVal);
}
// materialEval, cmaterial.h:2554-2628 (leaf: :2425-2551); fwdDir = EVAL_FLAG_FWD_DIR (light tracing: the adjoint fix of the leaf, needs sc.fn)
template <int F = HK_FEAT_ALL>
HK_DEV BxDFResult materialEval(const float* a_m, const ShadeContext& sc, const SceneDev& s, bool fwdDir = false) {
  BxDFResult val;
  val.brdf = mk3(0, 0, 0); val.btdf = mk3(0, 0, 0); val.pdfFwd = 0.0f; val.pdfRev = 0.0f; val.diffuse = true;
  float stackW[7]; int stackO[7];
  int top = 0, currOffset = 0;
  float currW = 1.0f;
  do {
    if (top > 0) { top--; currOffset = stackO[top]; currW = stackW[top]; }
    const float* m = a_m + size_t(currOffset) * HM_NODE_FLOATS;
    if (matType(m) == HMT_BLEND_MASK) {
      const float alpha = blendMaskAlpha2(m, sc.v, sc.n, sc.tc, s);
      const int o1 = as_int(m[HM_BLEND_MAT1]), o2 = as_int(m[HM_BLEND_MAT2]);
      float w1 = alpha;
      const float w2 = 1.0f - alpha;
      if ((as_int(m[HM_BLEND_FLAGS]) & HBF_REFLECTION_WEIGHT_IS_ONE) && matType(m + size_t(o1) * HM_NODE_FLOATS) != HMT_BLEND_MASK) w1 = 1.0f;
      if (top < 7) { stackW[top] = currW * w1; stackO[top] = currOffset + o1; top++; }
      if (top < 7) { stackW[top] = currW * w2; stackO[top] = currOffset + o2; top++; }
    } else {
      f3 brdf = mk3(0, 0, 0);
      float pf = 0.0f, pr = 0.0f;
      bool diffuse = false;
      const int type = matType(m);
      f3 n = sc.n;
      f3 btdf = mk3(0, 0, 0);
      float cosMult = 1.0f, cosMult2 = 1.0f;
      if ((F & HK_FEAT_NMAP) && hasNormalMap(m)) {   // :2431-2459: the cosine the caller applies is to sc.n, the lobe sees the bumped normal
        n = BumpMapping(sc.tg, sc.bn, sc.fn, sc.tc, m, s);
        const f3 lDir = fwdDir ? sc.v : sc.l;
        const float clampVal = fwdDir ? 0.15f : 1e-6f;
        const float cosThetaOut1 = fmaxf(dot(lDir, sc.n), 0.0f), cosThetaOut2 = fmaxf(dot(lDir, n), 0.0f);
        cosMult = (cosThetaOut2 / fmaxf(cosThetaOut1, clampVal));
        if (cosThetaOut1 <= 0.0f) cosMult = 0.0f;
        const float cosThetaOut3 = fmaxf(-dot(lDir, sc.n), 0.0f), cosThetaOut4 = fmaxf(-dot(lDir, n), 0.0f);
        cosMult2 = (cosThetaOut4 / fmaxf(cosThetaOut3, clampVal));
        if (cosThetaOut3 <= 0.0f) cosMult2 = 0.0f;
        if (fwdDir && dot(sc.l, sc.fn) <= 0.0f) { cosMult = 0.0f; cosMult2 = 0.0f; }
      }
      if (type == HMT_PHONG) {
        brdf = phongEvalBxDF(m, sc.l, sc.v, n, sc.tc, s) * cosMult;
        pf = phongEvalPDF(m, sc.l, sc.v, n, sc.tc, s);
        pr = phongEvalPDF(m, sc.v, sc.l, n, sc.tc, s);
      } else if ((F & HK_FEAT_GGX) && type == HMT_GGX) {
        brdf = ggxEvalBxDF(m, sc.l, sc.v, n, sc.tc, s) * cosMult;
        pf = ggx2EvalPDF(m, sc.l, sc.v, n, sc.tc, s);
        pr = ggx2EvalPDF(m, sc.v, sc.l, n, sc.tc, s);
      } else if (type == HMT_LAMBERT) {
        brdf = (lambertColor(m, sc.tc, s) * HK_INV_PI) * cosMult;
        pf = fabsf(dot(sc.l, n)) * HK_INV_PI;
        pr = fabsf(dot(sc.v, n)) * HK_INV_PI;
        diffuse = true;
      } else if ((F & HK_FEAT_OREN_NAYAR) && type == HMT_OREN_NAYAR) {
        brdf = orennayarEvalBxDF(m, sc.l, sc.v, n, sc.tc, s) * cosMult;
        pf = fabsf(dot(sc.l, n)) * HK_INV_PI;
        pr = fabsf(dot(sc.v, n)) * HK_INV_PI;
        diffuse = true;
      } else if ((F & HK_FEAT_BLINN) && type == HMT_BLINN) {
        brdf = blinnEvalBxDF(m, sc.l, sc.v, n, sc.tc, s) * cosMult;
        pf = blinnEvalPDF(m, sc.l, sc.v, n, sc.tc, s);
        pr = blinnEvalPDF(m, sc.v, sc.l, n, sc.tc, s);
      } else if ((F & HK_FEAT_ANISO) && type == HMT_BECKMANN) {
        brdf = anisoEvalBxDF<false>(m, sc.l, sc.v, n, sc.tg, sc.bn, sc.tc, s) * cosMult;
        pf = anisoEvalPDF<false>(m, sc.l, sc.v, n, sc.tg, sc.bn, sc.tc, s);
        pr = anisoEvalPDF<false>(m, sc.v, sc.l, n, sc.tg, sc.bn, sc.tc, s);
      } else if ((F & HK_FEAT_ANISO) && type == HMT_TRGGX) {
        brdf = anisoEvalBxDF<true>(m, sc.l, sc.v, n, sc.tg, sc.bn, sc.tc, s) * cosMult;
        pf = anisoEvalPDF<true>(m, sc.l, sc.v, n, sc.tg, sc.bn, sc.tc, s);
        pr = anisoEvalPDF<true>(m, sc.v, sc.l, n, sc.tg, sc.bn, sc.tc, s);
      } else if ((F & HK_FEAT_TRANSLUCENT) && type == HMT_TRANSLUCENT) {
        btdf = translucentEvalBxDF(m, sc.l, sc.v, n, sc.tc, s) * cosMult2;
        pf = translucentEvalPDF(sc.l, sc.v, n);
        pr = translucentEvalPDF(sc.v, sc.l, n);
        diffuse = true;
      }
      if (fwdDir) brdf = brdf * adjointBsdfShadeNormalFix(sc.v, sc.l, sc.n, sc.fn, diffuse ? 20.0f : 2.0f);
      val.brdf = val.brdf + (brdf * currW);
      if (F & HK_FEAT_TRANSLUCENT) val.btdf = val.btdf + (btdf * currW);
      val.pdfFwd += currW * pf;
      val.pdfRev += currW * pr;
      val.diffuse = val.diffuse && diffuse;
    }
  } while (top > 0);
  return val;
}
// materialEvalEmission, cmaterial.h:2918-2978
HK_DEV f3 materialEvalEmission(const float* a_m, f3 v, f3 n, f2 tc, const SceneDev& s) {
  f3 val = mk3(0, 0, 0);
  float stackW[7]; int stackO[7];
  int top = 0, currOffset = 0;
  float currW = 1.0f;
  do {
    if (top > 0) { top--; currOffset = stackO[top]; currW = stackW[top]; }
    const float* m = a_m + size_t(currOffset) * HM_NODE_FLOATS;
    if (matType(m) == HMT_BLEND_MASK) {
      const float alpha = blendMaskAlpha2(m, v, n, tc, s);
      const int o1 = as_int(m[HM_BLEND_MAT1]), o2 = as_int(m[HM_BLEND_MAT2]);
      if (top < 7) { stackW[top] = currW * alpha; stackO[top] = currOffset + o1; top++; }
      if (top < 7) { stackW[top] = currW * (1.0f - alpha); stackO[top] = currOffset + o2; top++; }
    }
    const f3 e = mk3(m[HM_EMISSIVE_COLOR], m[HM_EMISSIVE_COLOR + 1], m[HM_EMISSIVE_COLOR + 2]) * sample2DExt(as_int(m[HM_EMISSIVE_TEXMATRIXID]), tc, m, s);
    val = val + (e * currW);
  } while (top > 0);
  return val;
}
HK_DEV uint32_t flagsNextBounceLite(uint32_t flags, const MatSample& ms, const SceneDev& s) {   // cmaterial.h:3262-3293
  const uint32_t bounce = (flags >> 8) & 0xFFu, diff = flags & 0xFFu;
  uint32_t other = flags >> 16;
  flags = (flags & 0xFFFF00FFu) | ((bounce + 1) << 8);
  if (ms.flags & HRE_D) flags = (flags & 0xFFFFFF00u) | (diff + 1);
  const uint32_t diff2 = flags & 0xFFu;
  if (((bounce + 1) >= uint32_t(g_varsI(s)[HV_I_TRACE_DEPTH])) || (diff2 >= uint32_t(g_varsI(s)[HV_I_DIFFUSE_TRACE_DEPTH]) + 1)) other |= HRF_IS_DEAD;
  if (ms.flags & HRE_G) other |= HRE_G;
  if ((ms.flags & HRE_S) || (ms.flags & HRE_T)) other |= HRE_S;
  if (ms.flags & HRE_D) other |= HRE_D;
  if (ms.flags & HRE_T) other |= HRE_T;
  return (flags & 0x0000FFFFu) | (other << 16);
}

// ================================================================================================ lights
HK_DEV const float* lightAt(const SceneDev& s, int id) {   // clight.h:1739-1749
  if (id < 0) return nullptr;
  return s.lightsBase + size_t(id) * HL_FLOATS;
}
HK_DEV f3 lightPos(const float* L) { return mk3(L[HL_POS], L[HL_POS + 1], L[HL_POS + 2]); }
HK_DEV f3 lightNorm(const float* L) { return mk3(L[HL_NORM], L[HL_NORM + 1], L[HL_NORM + 2]); }
HK_DEV f3 lightColor(const float* L) { return mk3(L[HL_COLOR], L[HL_COLOR + 1], L[HL_COLOR + 2]); }

HK_DEV float areaDiffuseLightEvalPDF(const float* L, f3 rayDir, float hitDist) {   // clight.h:524-530, PdfAtoW cglobals.h:1754-1757
  const float pdfA = 1.0f / fmaxf(L[HL_SURFACE_AREA], HK_DEPSILON);
  const float d = dot(rayDir, lightNorm(L) * (-1.0f));
  const float cosVal = (as_int(L[HL_FLAGS]) & HLF_HAS_IES) ? fabsf(d) : fmaxf(d, 0.0f);
  return (pdfA * hitDist * hitDist) / fmaxf(cosVal, HK_DEPSILON2);
}
HK_DEV f3 areaDiffuseLightGetIntensity(const float* L, f3 rayDir, bool eyeRay) {   // clight.h:542-611 (plain / spot distributions)
  f3 color = lightColor(L);
  if (as_int(L[HL_AREA_SPOT_DISTR]) != 0) {
    const float cos1 = L[HL_AREA_SPOT_COS1], cos2 = L[HL_AREA_SPOT_COS2];
    const float cos_theta = fmaxf(dot(rayDir * (-1.0f), lightNorm(L)), 0.0f);
    const float tt = fminf(fmaxf((cos_theta - cos2) / (cos1 - cos2), 0.0f), 1.0f);
    const float atten = tt * tt * (3.0f - 2.0f * tt);
    if (!eyeRay) color = color * clampf(atten, 0.0f, 1.0f);
    else color = color * (1.0f / fmaxf(color.x, fmaxf(color.y, color.z)));
  }
  return color;
}
struct ShadowSample { f3 pos, color; float pdf, maxDist, cosAtLight; bool isPoint; };   // cglobals.h:2448-2456
// sky portals (AREA_LIGHT_SKY_PORTAL): an area light whose colour is multiplied by what the sky light it names shows in the ray's direction; defined below, after the sky
template <int F> HK_DEV f3 areaLightSkyPortalCustomColor(const SceneDev& s, const float* L, f3 rayDir);
// an area light with an IES distribution (LIGHT_HAS_IES): the colour is scaled by the photometric table in the ray's direction; defined below, after the table helpers
template <int F> HK_DEV f3 areaLightIntensity(const SceneDev& s, const float* L, f3 rayDir, bool eyeRay);
template <int F>
HK_DEV void AreaLightSampleRev(const SceneDev& s, const float* L, f3 rands, f3 illum, ShadowSample& out) {   // clight.h:1180-1229
  const float offsetX = rands.x * 2.0f - 1.0f, offsetY = rands.y * 2.0f - 1.0f;
  f3 sp = mk3(offsetX * L[HL_AREA_SIZE_X], 0.0f, offsetY * L[HL_AREA_SIZE_Y]);
  if (as_int(L[HL_AREA_IS_DISK]) != 0) {
    const f2 xz = MapSamplesToDisc(mk2(offsetX, offsetY));
    sp = mk3(xz.x * L[HL_AREA_SIZE_X], 0, xz.y * L[HL_AREA_SIZE_X]);
  }
  const float* M = L + HL_AREA_MATRIX;
  sp = mk3(M[0] * sp.x + M[1] * sp.y + M[2] * sp.z, M[3] * sp.x + M[4] * sp.y + M[5] * sp.z, M[6] * sp.x + M[7] * sp.y + M[8] * sp.z);
  sp = sp + lightPos(L);
  const f3 rayDir = normalize(sp - illum);
  const float hitDist = length(sp - illum);
  const f3 ln = lightNorm(L);
  out.isPoint = false;
  out.pos = sp + ln * epsilonOfPos(sp);
  if ((F & HK_FEAT_RARE_LIGHTS) && (as_int(L[HL_FLAGS]) & HLF_SKY_PORTAL)) out.color = areaLightSkyPortalCustomColor<F>(s, L, rayDir);
  else if ((F & HK_FEAT_RARE_LIGHTS) && (as_int(L[HL_FLAGS]) & HLF_IES_POINT_AREA)) out.color = areaLightIntensity<F>(s, L, normalize(lightPos(L) - illum), false);   // the table is read from the light's centre (:1215-1217)
  else out.color = areaLightIntensity<F>(s, L, rayDir, false);
  out.pdf = areaDiffuseLightEvalPDF(L, rayDir, hitDist);
  out.maxDist = hitDist;
  out.cosAtLight = -dot(rayDir, ln);
}
HK_DEV int SelectIndexPropToOpt(float a_r, const float* a_accum, int N, float& pPDF) {   // cglobals.h:2808-2859
  int leftBound = 0, rightBound = N - 2, counter = 0, currPos = -1;
  const float x = a_r * a_accum[N - 1];
  while (rightBound - leftBound > 1 && counter < 50) {
    const int currSize = rightBound + leftBound;
    const int currPos1 = (currSize % 2 == 0) ? (currSize + 1) / 2 : (currSize + 0) / 2;
    const float a = a_accum[currPos1], b = a_accum[currPos1 + 1];
    if (a < x && x <= b) { currPos = currPos1; break; }
    else if (x <= a) rightBound = currPos1;
    else if (x > b) leftBound = currPos1;
    counter++;
  }
  if (currPos < 0) {
    if (a_accum[leftBound] < x && x <= a_accum[leftBound + 1]) currPos = leftBound;
    if (a_accum[rightBound] < x && x <= a_accum[rightBound + 1]) currPos = rightBound;
  }
  if (x == 0.0f) currPos = 0;
  else if (currPos < 0) currPos = (rightBound + leftBound + 1) / 2;
  pPDF = (a_accum[currPos + 1] - a_accum[currPos]) / a_accum[N - 1];
  return currPos;
}
HK_DEV int SelectRandomLightRev(float r, const SceneDev& s, float& pickProb) {   // clight.h:1774-1793
  const int tableSize = s.hdr[HG_LSEL_REV_SIZE];
  pickProb = 1.0f;
  if (tableSize == 0) return -1;
  if (tableSize <= 2) return 0;
  return SelectIndexPropToOpt(r, s.lselRev, tableSize, pickProb);
}
// ---- sky dome light: constant colour or lat-long texture (clight.h:285-306, 308-364, 384-462; cfetch.h:259-296) ----
HK_DEV f2 sphereMapTo2DTexCoord(f3 ray_dir, float& sinTheta) {   // cfetch.h:259-281
  const float x = ray_dir.z, y = ray_dir.x, z = -ray_dir.y;
  const float theta = acosf(z);
  float phi = atan2f(y, x);
  if (phi < 0.0f) phi += 2.0f * HK_PI;
  const float texX = clampf(phi * 0.5f * HK_INV_PI, 0.0f, 1.0f);
  const float texY = clampf(theta * HK_INV_PI, 0.0f, 1.0f);
  sinTheta = sqrtf(1.0f - ray_dir.y * ray_dir.y);
  return mk2(texX, texY);
}
HK_DEV f3 texCoord2DToSphereMap(f2 tc, float& sinThetaOut) {   // cfetch.h:283-296
  const float phi = tc.x * 2.f * HK_PI, theta = tc.y * HK_PI;
  const float sinTheta = sinf(theta);
  const float x = sinTheta * cosf(phi), y = sinTheta * sinf(phi), z = cosf(theta);
  sinThetaOut = sinTheta;
  return mk3(y, -z, x);
}
HK_DEV const float* pdfTableHeader(const SceneDev& s, int tableId) {   // cfetch.h:153-163
  const int offset = s.globals[s.hdr[HG_PDF_TABLE_OFFS] + tableId];
  return reinterpret_cast<const float*>(s.pdfStorage + offset);
}
HK_DEV float evalMap2DPdf(f2 tc, const float* intervals, const int sizeX, const int sizeY) {   // clight.h:308-337
  const float fw = float(sizeX), fh = float(sizeY);
  if (tc.x < 0.0f || tc.x > 1.0f) tc.x -= float(int(tc.x));
  if (tc.y < 0.0f || tc.x > 1.0f) tc.y -= float(int(tc.y));   // sic: the reference tests x twice
  int pixelX = int(fw * tc.x - 0.5f), pixelY = int(fh * tc.y - 0.5f);
  if (pixelX >= sizeX) pixelX = sizeX - 1;
  if (pixelY >= sizeY) pixelY = sizeY - 1;
  if (pixelX < 0) pixelX += sizeX;
  if (pixelY < 0) pixelY += sizeY;
  const int pixelOffset = pixelY * sizeX + pixelX, maxSize = sizeX * sizeY;
  const int offset0 = (pixelOffset + 0 < maxSize + 0) ? pixelOffset + 0 : maxSize - 1;
  const int offset1 = (pixelOffset + 1 < maxSize + 1) ? pixelOffset + 1 : maxSize;
  return (intervals[offset1] - intervals[offset0]) * (fw * fh) / intervals[sizeX * sizeY];
}
HK_DEV float skyLightEvalPDF(const SceneDev& s, const float* L, f3 rayDir) {   // clight.h:339-364
  const float* hdr = pdfTableHeader(s, as_int(L[HL_SKY_PDF_TABLE0]));
  const int sizeX = as_int(hdr[0]), sizeY = as_int(hdr[1]);
  float sintheta = 0.0f;
  const f2 tc = sphereMapTo2DTexCoord(rayDir, sintheta);
  if (sintheta == 0.f) return 0.f;
  const float* r0 = L + HL_SKY_MATRIX0;
  const f2 tcT = mk2(r0[0] * tc.x + r0[1] * tc.y + r0[3], r0[4] * tc.x + r0[5] * tc.y + r0[7]);   // mul2x4, cfetch.h:642-648
  const float mapPdf = evalMap2DPdf(tcT, hdr + 4, sizeX, sizeY);
  return (mapPdf * 1.0f) / (2.f * HK_PI * HK_PI * fmaxf(fabsf(sintheta), HK_DEPSILON));
}
// ---- IES distributions (clight.h:405-426, 465-495; cfetch.h:364-462): the photometric web as a one-channel float lat-long image in the pdf arena
// ({w, h, 1, 4} + w * h floats, RenderDriverRTE_PdfTables.cpp:385-478) and a 2-D sampling table over the same pixels
HK_DEV float read_imagef_sw1(const float4* tex, f2 tc, int flags) {   // the float (bpp == 4) branch; IES images are never 8-bit
  const int4 header = *reinterpret_cast<const int4*>(tex);
  const int w = header.x, h = header.y;
  float ffx = tc.x * float(w) - 0.5f, ffy = tc.y * float(h) - 0.5f;
  if ((flags & HTEX_CLAMP_U) != 0 && ffx < 0) ffx = 0.0f;
  if ((flags & HTEX_CLAMP_V) != 0 && ffy < 0) ffy = 0.0f;
  const float* fdata = reinterpret_cast<const float*>(tex + 1);
  const int px = int(ffx), py = int(ffy);
  const float fx = fabsf(ffx - float(px)), fy = fabsf(ffy - float(py));
  const float fx1 = 1.0f - fx, fy1 = 1.0f - fy;
  const float w1 = fx1 * fy1, w2 = fx * fy1, w3 = fx1 * fy, w4 = fx * fy;
  const int4 offs = bilinearOffsets(ffx, ffy, flags, w, h);
  return ((fdata[offs.x] * w1 + fdata[offs.y] * w2) + fdata[offs.z] * w3) + fdata[offs.w] * w4;
}
HK_DEV f3 lightMatrixMul3(const float* M, f3 v) {   // matrix3x3f_mult_float3, cglobals.h:1091-1098
  return mk3(M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z);
}
HK_DEV float lightDistributionMask(const SceneDev& s, const float* L, f3 rayDir) {   // clight.h:465-484, for a light that has the table (1 otherwise)
  rayDir = normalize(lightMatrixMul3(L + HL_IES_LIGHT_MATRIX, rayDir));
  float sintheta = 0.0f;
  const f2 tc = sphereMapTo2DTexCoord(rayDir * (-1.0f), sintheta);
  const float4* tex = s.pdfStorage + s.globals[s.hdr[HG_PDF_TABLE_OFFS] + as_int(L[HL_IES_SPHERE_TEX_ID])];
  return read_imagef_sw1(tex, tc, HTEX_CLAMP_U | HTEX_CLAMP_V);
}
template <int F>
HK_DEV f3 areaLightIntensity(const SceneDev& s, const float* L, f3 rayDir, bool eyeRay) {   // areaDiffuseLightGetIntensity with its IES branch (clight.h:563-577)
  if ((F & HK_FEAT_RARE_LIGHTS) && (as_int(L[HL_FLAGS]) & HLF_HAS_IES)) {
    f3 color = lightColor(L);
    const float atten = lightDistributionMask(s, L, rayDir);
    if (!eyeRay) color = color * atten;
    else color = color * (1.0f / fmaxf(color.x, fmaxf(color.y, color.z)));
    return color;
  }
  return areaDiffuseLightGetIntensity(L, rayDir, eyeRay);
}
// ---- Perez all-weather sky (Preetham's fit), clight.h:178-282: zenith colour in Yxy, the distribution function, Yxy -> linear RGB,
// and the sun disc blended in over the last 0.05-0.15 % of the cosine
HK_DEV f3 perezZenith(float t, float thetaSun) {
  const float pi = 3.1415926f;   // the reference's own constant here
  const float t2 = t * t;
  const float chi = (4.0f / 9.0f - t / 120.0f) * (pi - 2.0f * thetaSun);
  const float th1 = thetaSun, th2 = thetaSun * thetaSun, th3 = thetaSun * thetaSun * thetaSun;
  // dot(float4 coefficients, (1, th, th^2, th^3)) in LiteMath's order x*x + y*y + z*z + w*w
  #define HK_DOT4(a, b, c, d) ((a) * 1.0f + (b) * th1 + (c) * th2 + (d) * th3)
  const float Y = (4.0453f * t - 4.9710f) * tanf(chi) - 0.2155f * t + 2.4192f;
  const float x = t2 * HK_DOT4(0.0f, 0.00209f, -0.00375f, 0.00165f) + t * HK_DOT4(0.00394f, -0.03202f, 0.06377f, -0.02903f) + HK_DOT4(0.25886f, 0.06052f, -0.21196f, 0.11693f);
  const float y = t2 * HK_DOT4(0.0f, 0.00317f, -0.00610f, 0.00275f) + t * HK_DOT4(0.00516f, -0.04153f, 0.08970f, -0.04214f) + HK_DOT4(0.26688f, 0.06670f, -0.26756f, 0.15346f);
  #undef HK_DOT4
  return mk3(Y, x, y);
}
HK_DEV f3 perezFunc(float t, float cosTheta, float cosGamma) {
  const float gamma = acosf(cosGamma), cosGammaSq = cosGamma * cosGamma;
  const float aY = 0.17872f * t - 1.46303f, bY = -0.35540f * t + 0.42749f, cY = -0.02266f * t + 5.32505f, dY = 0.12064f * t - 2.57705f, eY = -0.06696f * t + 0.37027f;
  const float ax = -0.01925f * t - 0.25922f, bx = -0.06651f * t + 0.00081f, cx = -0.00041f * t + 0.21247f, dx = -0.06409f * t - 0.89887f, ex = -0.00325f * t + 0.04517f;
  const float ay = -0.01669f * t - 0.26078f, by = -0.09495f * t + 0.00921f, cy = -0.00792f * t + 0.21023f, dy = -0.04405f * t - 1.65369f, ey = -0.01092f * t + 0.05291f;
  return mk3((1.0f + aY * expf(bY / cosTheta)) * (1.0f + cY * expf(dY * gamma) + eY * cosGammaSq),
             (1.0f + ax * expf(bx / cosTheta)) * (1.0f + cx * expf(dx * gamma) + ex * cosGammaSq),
             (1.0f + ay * expf(by / cosTheta)) * (1.0f + cy * expf(dy * gamma) + ey * cosGammaSq));
}
HK_DEV f3 perezSky(float turbidity, float cosTheta, float cosGamma, float cosThetaSun) {
  const f3 a = perezZenith(turbidity, acosf(cosThetaSun)) * perezFunc(turbidity, cosTheta, cosGamma), b = perezFunc(turbidity, 1.0f, cosThetaSun);
  return mk3(a.x / b.x, a.y / b.y, a.z / b.z);
}
HK_DEV f3 perezConvertColor(f3 clrYxy) {   // convertColor, clight.h:235-252
  clrYxy.x = 1.0f - expf(-clrYxy.x / 20.0f);
  const float ratio = clrYxy.x / fmaxf(clrYxy.z, 1e-10f);
  f3 XYZ;
  XYZ.x = clrYxy.y * ratio;
  XYZ.y = clrYxy.x;
  XYZ.z = ratio - XYZ.x - XYZ.y;
  return clamp3(mk3(dot(mk3(3.240479f, -1.53715f, -0.49853f), XYZ), dot(mk3(-0.969256f, 1.875991f, 0.041556f), XYZ), dot(mk3(0.055684f, -0.204043f, 1.057311f), XYZ)), 0.0f, 1.0f);
}
HK_DEV f3 skyLightPerezColor(const float* L, f3 ray_dir) {   // clight.h:255-282
  const f3 sunDir = mk3(L[HL_SKY_SUN_DIR], L[HL_SKY_SUN_DIR + 1], L[HL_SKY_SUN_DIR + 2]);
  const float turbidity = L[HL_SKY_TURBIDITY];
  const f3 colorYxy = perezSky(turbidity, fmaxf(ray_dir.y, 0.0f) + 0.05f, fmaxf(dot(sunDir, ray_dir * (-1.0f)), 0.0f), fmaxf(-sunDir.y, 0.0f));
  f3 rgb = perezConvertColor(colorYxy);
  rgb.x = powf(rgb.x, 2.2f); rgb.y = powf(rgb.y, 2.2f); rgb.z = powf(rgb.z, 2.2f);
  const float tSunAngle = fmaxf(-sunDir.y, 0.0f);
  const float threshold = 0.9985f + tSunAngle * (0.9995f - 0.9985f);
  const float tSun = dot(sunDir, ray_dir * (-1.0f));
  if (tSun >= threshold) {
    const f3 sunColor = mk3(L[HL_SKY_SUN_COLOR], L[HL_SKY_SUN_COLOR + 1], L[HL_SKY_SUN_COLOR + 2]) * (2.0f + 2.0f * tSunAngle);
    float tSun2 = (tSun - threshold) / (1.0f - threshold);
    tSun2 = tSun2 * tSun2;
    rgb = (sunColor * tSun2) + (rgb * (1.0f - tSun2));
  }
  return rgb;
}
// skyLightGetIntensityTexturedENV, clight.h:285-306: the lat-long texture is fetched in both branches there, used in one
template <int F = HK_FEAT_ALL>
HK_DEV f3 skyLightIntensity(const SceneDev& s, const float* L, f3 dir) {
  if ((F & HK_FEAT_PEREZ) && (as_int(L[HL_FLAGS]) & HLF_SKY_USE_PEREZ)) return lightColor(L) * skyLightPerezColor(L, dir);
  float sintheta = 0.0f;
  const f2 tc = sphereMapTo2DTexCoord(dir, sintheta);
  return lightColor(L) * sample2DExt(as_int(L[HL_COLOR_TEX_MATRIX]), tc, L + HL_SKY_SAMPLER0, s);
}
// areaLightSkyPortalCustomColor, clight.h:614-629, and the tail of areaDiffuseLightGetIntensity (:590-607): the sky record sits AREA_LIGHT_SKY_OFFSET records
// away from the portal's (RenderDriverRTE.cpp:1670-1682); a Perez sky counts half
template <int F>
HK_DEV f3 portalSkyColor(const SceneDev& s, const float* L, f3 rayDir) {
  const float* sky = L + ptrdiff_t(as_int(L[HL_AREA_SKY_OFFSET])) * HL_FLOATS;
  if ((F & HK_FEAT_PEREZ) && (as_int(sky[HL_FLAGS]) & HLF_SKY_USE_PEREZ)) return skyLightPerezColor(sky, rayDir) * 0.5f;
  return skyLightIntensity<F & ~HK_FEAT_PEREZ>(s, sky, rayDir);
}
template <int F>
HK_DEV f3 areaLightSkyPortalCustomColor(const SceneDev& s, const float* L, f3 rayDir) { return lightColor(L) * portalSkyColor<F>(s, L, rayDir); }
HK_DEV void SkyLightSampleRev(const SceneDev& s, const float* L, f3 rands, f3 illum, ShadowSample& out) {   // clight.h:427-462
  const float* hdr = pdfTableHeader(s, as_int(L[HL_SKY_PDF_TABLE0]));
  const int sizeX = as_int(hdr[0]), sizeY = as_int(hdr[1]);
  const float fw = float(sizeX), fh = float(sizeY);
  float pdf = 1.0f;   // sampleMap2D, clight.h:378-403
  int pixelOffset = SelectIndexPropToOpt(rands.z, hdr + 4, sizeX * sizeY + 1, pdf);
  if (pixelOffset >= sizeX * sizeY) pixelOffset = sizeX * sizeY - 1;
  const int yPos = pixelOffset / sizeX, xPos = pixelOffset - yPos * sizeX;
  const float texX = (1.0f / fw) * ((float(xPos) + 0.5f) + (rands.x * 2.0f - 1.0f) * 0.5f);
  const float texY = (1.0f / fh) * ((float(yPos) + 0.5f) + (rands.y * 2.0f - 1.0f) * 0.5f);
  const float mapPdf = pdf * (fw * fh);
  const m44 inv = load_m44(reinterpret_cast<const float4*>(L + HL_SKY_INV_MATRIX0));
  const f3 tcT = mul4x3(inv, mk3(texX, texY, 0.0f));
  float sintheta = 0.0f;
  const f3 sampleDir = texCoord2DToSphereMap(mk2(tcT.x, tcT.y), sintheta);
  const f3 samplePos = illum + sampleDir * g_varsF(s)[HV_F_BSPHERE_RADIUS];
  const f3 txClr = sample2DExt(as_int(L[HL_COLOR_TEX_MATRIX]), mk2(tcT.x, tcT.y), L + HL_SKY_SAMPLER0, s);
  out.isPoint = false;
  out.pos = samplePos;
  out.color = lightColor(L) * txClr;
  out.pdf = (mapPdf * 1.0f) / (2.f * HK_PI * HK_PI * fmaxf(fabsf(sintheta), HK_DEPSILON));
  out.maxDist = length(illum - samplePos);
  out.cosAtLight = 1.0f;
}
// ---- delta lights: point (omni, no IES), spot, directional (clight.h:1394-1506) ----
HK_DEV float mylocalsmoothstep(float edge0, float edge1, float x) {   // clight.h:7-12
  const float tVal = (x - edge0) / (edge1 - edge0);
  const float t = fminf(fmaxf(tVal, 0.0f), 1.0f);
  return t * t * (3.0f - 2.0f * t);
}
HK_DEV float PdfAtoW(float aPdfA, float aDist, float aCosThere) { return (aPdfA * aDist * aDist) / fmaxf(aCosThere, HK_DEPSILON2); }   // cglobals.h:1754-1757
template <int F = HK_FEAT_ALL>
HK_DEV void PointLightSampleRev(const SceneDev& s, const float* L, f3 illum, ShadowSample& out) {   // clight.h:1394-1407; lightDistributionMask = 1 without IES (:465-484)
  const f3 samplePos = lightPos(L);
  const float hitDist = length(samplePos - illum);
  out.isPoint = true;
  out.pos = samplePos;
  float mask = 1.0f;
  if ((F & HK_FEAT_RARE_LIGHTS) && (as_int(L[HL_FLAGS]) & HLF_HAS_IES)) mask = lightDistributionMask(s, L, normalize(samplePos - illum));
  out.color = mk3(mask, mask, mask) * lightColor(L);
  out.pdf = PdfAtoW(1.0f, hitDist, 1.0f);
  out.maxDist = hitDist;
  out.cosAtLight = 1.0f;
}
HK_DEV void SpotLightSampleRev(const float* L, f3 illum, ShadowSample& out) {   // clight.h:1416-1450
  const f3 samplePos = lightPos(L), norm = lightNorm(L);
  const float hitDist = length(samplePos - illum);
  const f3 rayDir = normalize(samplePos - illum);
  const float cos_theta = fmaxf(dot(rayDir * (-1.0f), norm), 0.0f);
  const float atten = mylocalsmoothstep(L[HL_POINT_SPOT_COS2], L[HL_POINT_SPOT_COS1], cos_theta);
  out.isPoint = true;
  out.pos = samplePos;
  out.color = lightColor(L) * atten;
  out.pdf = PdfAtoW(1.0f, hitDist, 1.0f);
  out.maxDist = hitDist;
  out.cosAtLight = fmaxf(-dot(rayDir, norm), 0.0f);
}
HK_DEV f3 MapSamplesToCone(float cosCutoff, f2 sample, f3 direction) {   // cglobals.h:1655-1681
  const float cosTheta = (1.0f - sample.x) + sample.x * cosCutoff;
  const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
  const float sinPhi = sinf(2.0f * HK_PI * sample.y), cosPhi = cosf(2.0f * HK_PI * sample.y);
  const f3 deviation = mk3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
  f3 nx, nz;
  CoordinateSystem(direction, nx, nz);
  const f3 ny = nz, nz2 = direction;   // the reference swaps ny and nz
  return ((nx * deviation.x) + (ny * deviation.y)) + (nz2 * deviation.z);
}
HK_DEV void DirectLightSampleRev(const float* L, f3 rands, f3 illum, ShadowSample& out) {   // clight.h:1478-1506, attenuation :892-912
  const f3 lpos = lightPos(L);
  f3 norm = lightNorm(L);
  const float pdfW = 1.0f;
  if (L[HL_DIRECT_SSOFTNESS] > 1e-5f) norm = MapSamplesToCone(L[HL_DIRECT_ALPHA_COS], mk2(rands.x, rands.y), norm);
  const f3 AC = illum - lpos;
  const float CBLen = dot(normalize(AC), norm) * length(AC);
  float atten = 0.0f;
  {
    const f3 n0 = lightNorm(L);
    const float cos_alpha = dot(normalize(illum - lpos), n0);
    if (cos_alpha > 0.0f) {
      const float sinAlpha = sqrtf(1.0f - cos_alpha * cos_alpha);
      const float d = length(illum - lpos) * sinAlpha;
      const float r1 = L[HL_DIRECT_RADIUS1], r2 = L[HL_DIRECT_RADIUS2];
      atten = mylocalsmoothstep(fmaxf(r2, r1), fminf(r2, r1), d);
    }
  }
  out.isPoint = true;
  out.pos = illum - norm * CBLen;
  out.color = (lightColor(L) * atten) * pdfW;
  out.pdf = pdfW;
  out.maxDist = CBLen;
  out.cosAtLight = 1.0f;
}
// LightSampleRev, clight.h:1561-1610: the light types this layer accepts (upload_globals rejects the others)
// sphere lights, clight.h:1287-1332: uniform points on the sphere, area pdf turned into a solid-angle pdf at the shaded point
HK_DEV float sphereLightEvalPDF(const float* L, f3 illum, f3 lpos, f3 lnorm) {
  const float lradius = L[HL_SPHERE_RADIUS];
  const f3 lcenter = lightPos(L);
  const f3 dc = lcenter - illum;
  if (dot(dc, dc) - lradius * lradius <= 0.0f) return 1.0f;
  const float pdfA = 1.0f / L[HL_SURFACE_AREA];
  const float dist = length(lpos - illum);
  const f3 dirToV = normalize(lpos - illum);
  return PdfAtoW(pdfA, dist, fabsf(dot(dirToV, lnorm)));
}
HK_DEV f3 sphereLightUnitSample(float r1, float r2) {   // the direction both samplers build (:1312-1316, :726-730)
  const float theta = 2.0f * HK_PI * r1;
  const float phi = acosf(1.0f - 2.0f * r2);
  return mk3(sinf(phi) * cosf(theta), sinf(phi) * sinf(theta), cosf(phi));
}
HK_DEV void SphereLightSampleRev(const float* L, f3 rands, f3 illum, ShadowSample& out) {
  const f3 lcenter = lightPos(L);
  const f3 samplePos = lcenter + (sphereLightUnitSample(rands.x, rands.y) * L[HL_SPHERE_RADIUS]);
  const f3 lnorm = normalize(samplePos - lcenter);
  const f3 dirToV = normalize(samplePos - illum);
  out.isPoint = false;
  out.pos = samplePos;
  out.color = lightColor(L);
  out.pdf = sphereLightEvalPDF(L, illum, samplePos, lnorm);
  out.maxDist = length(samplePos - illum);
  out.cosAtLight = fabsf(dot(lnorm, dirToV));
}
// mesh lights, clight.h:966-1062, 1513-1546: a copy of the light's mesh and the prefix sums of its triangle areas live in the pdf arena
// (MeshLight, PlainLightConverter.cpp:724-835); a triangle by area, a point in it by uniform barycentrics; the colour texture (meshLightGetIntensity, :957-963) is looked up at the
// texture coordinates interpolated the same way
HK_DEV void MeshLightSamplePos(const SceneDev& s, const float* L, f3 rands, f3& pos, f3& norm, f2& texCoord, float& pdfA) {
  const int meshId = as_int(L[HL_MESH_MESH_ID]), pdftId = as_int(L[HL_MESH_TABLE_ID]), triNum = as_int(L[HL_MESH_TRI_NUM]);
  const int tabOffs = s.hdr[HG_PDF_TABLE_OFFS];
  const float4* mesh = s.pdfStorage + s.globals[tabOffs + meshId];
  const float* table = reinterpret_cast<const float*>(s.pdfStorage + s.globals[tabOffs + pdftId]);
  const HydraPlainMesh* hdr = reinterpret_cast<const HydraPlainMesh*>(mesh);
  const float4* vpos = mesh + hdr->vPosOffset;
  const float4* vnorm = mesh + hdr->vNormOffset;
  const int* indices = reinterpret_cast<const int*>(mesh + hdr->vIndicesOffset);
  float pickProb = 1.0f;
  const int triangleId = SelectIndexPropToOpt(rands.z, table, triNum + 1, pickProb);
  const int iA = indices[triangleId * 3 + 0], iB = indices[triangleId * 3 + 1], iC = indices[triangleId * 3 + 2];
  const float4 dA = vpos[iA], dB = vpos[iB], dC = vpos[iC], dnA = vnorm[iA], dnB = vnorm[iB], dnC = vnorm[iC];
  const f3 A = xyz(dA), B = xyz(dB), C = xyz(dC);
  const f3 nA = xyz(dnA), nB = xyz(dnB), nC = xyz(dnC);
  float u = rands.x, v = rands.y;
  if (u + v > 1.0f) { u = 1.0f - u; v = 1.0f - v; }
  const float w = 1.0f - u - v;
  pos = ((A * u) + (B * v)) + (C * w);
  norm = ((nA * u) + (nB * v)) + (nC * w);
  texCoord = mk2((dA.w * u + dB.w * v) + dC.w * w, (dnA.w * u + dnB.w * v) + dnC.w * w);   // u rides in pos.w, v in norm.w (:1003-1005)
  pdfA = 1.0f / L[HL_SURFACE_AREA];
}
HK_DEV f3 meshLightMatrixMul(const float* M, f3 v) {   // matrix3x3f_mult_float3, cglobals.h:1091-1098
  return mk3(M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z);
}
HK_DEV f3 meshLightGetIntensity(const SceneDev& s, const float* L, f2 tc) { return sample2DExt(as_int(L[HL_MESH_TEXMATRIX_ID]), tc, L, s) * lightColor(L); }
template <int F = HK_FEAT_ALL>
HK_DEV void MeshLightSampleRev(const SceneDev& s, const float* L, f3 rands, f3 illum, ShadowSample& out) {
  f3 samplePos, sampleNorm; f2 tc; float pdfA;
  MeshLightSamplePos(s, L, rands, samplePos, sampleNorm, tc, pdfA);
  samplePos = meshLightMatrixMul(L + HL_MESH_MATRIX, samplePos);
  sampleNorm = normalize(meshLightMatrixMul(L + HL_MESH_MATRIX, sampleNorm));
  samplePos = samplePos + lightPos(L);
  const f3 rayDir = normalize(samplePos - illum);
  const float hitDist = length(samplePos - illum);
  const float cosVal = fmaxf(-dot(rayDir, sampleNorm), 0.0f);
  out.isPoint = false;
  out.pos = samplePos + sampleNorm * epsilonOfPos(samplePos);
  out.color = (F & HK_FEAT_RARE_LIGHTS) ? meshLightGetIntensity(s, L, tc) : lightColor(L);   // an untextured light's sampler returns white
  out.pdf = PdfAtoW(pdfA, hitDist, cosVal);
  out.maxDist = hitDist;
  out.cosAtLight = cosVal;
}
HK_DEV float meshLightEvalPDF(const float* L, f3 rayDir, f3 lnorm, float hitDist) {
  const float pdfA = 1.0f / fmaxf(L[HL_SURFACE_AREA], HK_DEPSILON);
  return PdfAtoW(pdfA, hitDist, fmaxf(dot(rayDir, lnorm * (-1.0f)), 0.0f));
}
// cylinder lights, clight.h:753-830, 1338-1385: (z, phi) over the 2-D table the front end makes from the colour texture (2 x 2 uniform without one), mapped onto the
// open cylinder of the light's local frame; the area pdf carries the table's density
struct Map2DSample { f2 texCoord; float mapPdf; };
HK_DEV Map2DSample sampleMap2D(f3 rands, const float* intervals, const int sizeX, const int sizeY) {   // clight.h:378-403
  const float fw = float(sizeX), fh = float(sizeY);
  float pdf = 1.0f;
  int pixelOffset = SelectIndexPropToOpt(rands.z, intervals, sizeX * sizeY + 1, pdf);
  if (pixelOffset >= sizeX * sizeY) pixelOffset = sizeX * sizeY - 1;
  const int yPos = pixelOffset / sizeX, xPos = pixelOffset - yPos * sizeX;
  Map2DSample r;
  r.texCoord = mk2((1.0f / fw) * ((float(xPos) + 0.5f) + (rands.x * 2.0f - 1.0f) * 0.5f), (1.0f / fh) * ((float(yPos) + 0.5f) + (rands.y * 2.0f - 1.0f) * 0.5f));
  r.mapPdf = pdf * (fw * fh);
  return r;
}
HK_DEV f3 cylinderLightGetIntensity(const SceneDev& s, const float* L, f2 tc) { return sample2DExt(as_int(L[HL_CYL_TEXMATRIX_ID]), tc, L, s) * lightColor(L); }
HK_DEV void CylinderLightSamplePos(const SceneDev& s, const float* L, f3 rands, f3& pos, f3& norm, f2& texCoord, float& pdfA) {
  Map2DSample sample;
  sample.texCoord = mk2(rands.x, rands.y);
  sample.mapPdf = 1.0f;
  const int texId = as_int(L[HL_CYL_PDF_TABLE_ID]);
  if (texId > 0) {
    const float* hdr = pdfTableHeader(s, texId);
    sample = sampleMap2D(rands, hdr + 4, as_int(hdr[0]), as_int(hdr[1]));
  }
  pdfA = sample.mapPdf / L[HL_SURFACE_AREA];
  const float zMin = L[HL_CYL_ZMIN], zMax = L[HL_CYL_ZMAX], radius = L[HL_CYL_RADIUS], phiMax = L[HL_CYL_PHIMAX];
  const float z = zMin + sample.texCoord.x * (zMax - zMin);
  const float phi = sample.texCoord.y * phiMax;
  const float sinPhi = sinf(phi), cosPhi = cosf(phi);
  f3 pObj = mk3(radius * cosPhi, radius * sinPhi, z);
  f3 n = normalize(mk3(pObj.x, pObj.y, 0.0f));
  const float hitRad = sqrtf(pObj.x * pObj.x + pObj.y * pObj.y);
  pObj.x *= radius / hitRad;
  pObj.y *= radius / hitRad;
  n = normalize(meshLightMatrixMul(L + HL_CYL_MATRIX, n));
  const f3 center = lightPos(L);
  pos = (center + meshLightMatrixMul(L + HL_CYL_MATRIX, pObj)) + (n * epsilonOfPos(center));
  norm = n;
  texCoord = sample.texCoord;
}
HK_DEV float cylinderLightEvalPDF(const SceneDev& s, const float* L, f3 illum, f3 lpos, f3 lnorm, f2 texCoord) {
  float mapPdf = 1.0f;
  const int texId = as_int(L[HL_CYL_PDF_TABLE_ID]);
  if (texId != 0) {
    const float* hdr = pdfTableHeader(s, texId);
    mapPdf = evalMap2DPdf(texCoord, hdr + 4, as_int(hdr[0]), as_int(hdr[1]));
  }
  const float hitDist = length(lpos - illum);
  const f3 rayDir = normalize(lpos - illum);
  const float pdfA = mapPdf / fmaxf(L[HL_SURFACE_AREA], HK_DEPSILON);
  return PdfAtoW(pdfA, hitDist, fmaxf(dot(rayDir, lnorm * (-1.0f)), 0.0f));
}
HK_DEV void CylinderLightSampleRev(const SceneDev& s, const float* L, f3 rands, f3 illum, ShadowSample& out) {
  f3 samplePos, n; f2 tc; float pdfA;
  CylinderLightSamplePos(s, L, rands, samplePos, n, tc, pdfA);
  const float hitDist = length(samplePos - illum);
  const f3 rayDir = normalize(samplePos - illum);
  const float cosVal = fmaxf(dot(rayDir, n * (-1.0f)), 0.0f);
  out.isPoint = false;
  out.pos = samplePos;
  out.color = cylinderLightGetIntensity(s, L, tc);
  out.pdf = PdfAtoW(pdfA, hitDist, cosVal);
  out.maxDist = hitDist;
  out.cosAtLight = cosVal;
}
// lightEvalPDF, clight.h:1613-1633, for the light a path has run into: the types that have a surface (area rectangles / disks, spheres, meshes)
template <int F = HK_FEAT_ALL>
HK_DEV float lightEvalPDF(const SceneDev& s, const float* L, f3 illum, f3 rayDir, f3 lpos, f3 lnorm, f2 texCoord) {
  if (as_int(L[HL_TYPE]) == HLT_SPHERE) return sphereLightEvalPDF(L, illum, lpos, lnorm);
  if ((F & HK_FEAT_RARE_LIGHTS) && as_int(L[HL_TYPE]) == HLT_CYLINDER) return cylinderLightEvalPDF(s, L, illum, lpos, lnorm, texCoord);
  if (as_int(L[HL_TYPE]) == HLT_MESH) return meshLightEvalPDF(L, rayDir, lnorm, length(illum - lpos));
  return areaDiffuseLightEvalPDF(L, rayDir, length(illum - lpos));
}
template <int F = HK_FEAT_ALL>
HK_DEV void LightSampleRev(const SceneDev& s, const float* L, f3 rands, f3 illum, ShadowSample& out) {
  const int type = as_int(L[HL_TYPE]);
  if ((F & HK_FEAT_SKY) && type == HLT_SKY_DOME) SkyLightSampleRev(s, L, rands, illum, out);
  else if ((F & HK_FEAT_DELTA_LIGHTS) && type == HLT_DIRECT) DirectLightSampleRev(L, rands, illum, out);
  else if ((F & HK_FEAT_DELTA_LIGHTS) && type == HLT_POINT_SPOT) SpotLightSampleRev(L, illum, out);
  else if ((F & HK_FEAT_DELTA_LIGHTS) && type == HLT_POINT_OMNI) PointLightSampleRev<F>(s, L, illum, out);
  else if ((F & HK_FEAT_DELTA_LIGHTS) && type == HLT_SPHERE) SphereLightSampleRev(L, rands, illum, out);
  else if ((F & HK_FEAT_DELTA_LIGHTS) && type == HLT_MESH) MeshLightSampleRev<F>(s, L, rands, illum, out);
  else if ((F & HK_FEAT_RARE_LIGHTS) && type == HLT_CYLINDER) CylinderLightSampleRev(s, L, rands, illum, out);
  else AreaLightSampleRev<F>(s, L, rands, illum, out);
}
// environmentColor, cbidir.h:492-533 (misPrev.prevMaterialOffset stays -1 on this path: PT_Loop.cpp:247-249)
template <int F = HK_FEAT_ALL>
HK_DEV f3 environmentColor(const SceneDev& s, f3 rayDir, float prevPdf, bool prevSpecular, uint32_t flags) {
  if (!(F & HK_FEAT_SKY)) return mk3(0, 0, 0);   // a scene without a sky light: skyLightId == -1 (cbidir.h:498-499)
  const int skyId = s.hdr[HG_SKY_LIGHT_ID];
  if (skyId == -1) return mk3(0, 0, 0);
  const float* L = lightAt(s, skyId);
  f3 envColor = skyLightIntensity<F>(s, L, rayDir);
  const uint32_t rayBounceNum = (flags >> 8) & 0xFFu;   // unpackBounceNum, cglobals.h:1330-1340
  if (rayBounceNum > 0 && !(uint32_t(s.hdr[HG_FLAGS]) & HF_STUPID_PT_MODE) && !prevSpecular) {
    const float lgtPdf = L[HL_PICK_PROB_REV] * skyLightEvalPDF(s, L, rayDir);
    envColor = envColor * misWeightHeuristic(prevPdf, lgtPdf);
  }
  return envColor;
}

// hitDirectLight, clight.h:1636-1657: the first sun of the header whose cone holds the ray
HK_DEV int hitDirectLight(const SceneDev& s, f3 ray_dir) {
  const float* gf = reinterpret_cast<const float*>(s.globals);
  const int sunNumber = s.globals[HG_SUN_NUMBER];   // beyond the words a kernel may hold staged in LDS (HK_HDR_WORDS): read from the buffer itself
  for (int sunId = 0; sunId < sunNumber; sunId++) {
    const float* sun = gf + HG_SUNS + sunId * HL_FLOATS;
    if (-dot(ray_dir, lightNorm(sun)) > sun[HL_DIRECT_ALPHA_COS]) return sunId;
  }
  return -1;
}
HK_DEV float directLightAttenuation(const float* L, f3 illum) {   // clight.h:892-912
  const f3 lpos = lightPos(L);
  const float cos_alpha = dot(normalize(illum - lpos), lightNorm(L));
  if (cos_alpha > 0.0f) {
    const float sinAlpha = sqrtf(1.0f - cos_alpha * cos_alpha);
    const float d = length(illum - lpos) * sinAlpha;
    const float r1 = L[HL_DIRECT_RADIUS1], r2 = L[HL_DIRECT_RADIUS2];
    return mylocalsmoothstep(fmaxf(r2, r1), fminf(r2, r1), d);
  }
  return 0.0f;
}
HK_DEV float directLightEvalPDF(const float* L, f3 ray_dir) {   // clight.h:1462-1476
  if (L[HL_DIRECT_SSOFTNESS] > 1e-5f) {
    const float tanAlpha = L[HL_DIRECT_ALPHA_TAN], cosTheta = -dot(ray_dir, lightNorm(L));
    return HK_PI * (tanAlpha * tanAlpha) * (cosTheta * cosTheta * cosTheta);
  }
  return 1.0f;
}
// ---- the back-plate: what the CAMERA sees where a ray leaves the scene, when the scene names one (a sky light's or a shadow catcher's <back>: HRT_SHADOW_MATTE_BACK).
// environmentColorExtended, cbidir.h:593-629, the OpenCL layer's form of the miss shader (HitEnvOrLightKernel, material.cl:354): a ray into a sun of the header's table
// returns that sun; any other ray the environment as before, replaced by backColorOfSecondEnv (:543-573) for camera rays and for rays that only passed through
// transparent surfaces.  The CPU integrator has no such branch (kernel_HitEnvironment calls environmentColor, PT_Loop.cpp:28); this layer takes it when, and only when,
// the header names a back texture, so every scene without one keeps the CPU path's values.
HK_DEV f3 backColorOfSecondEnv(const SceneDev& s, f3 ray_dir, float screenX, float screenY) {
  const float* vf = g_varsF(s);
  const int offset = s.texTable[g_varsI(s)[HV_I_SHADOW_MATTE_BACK]];
  const f3 mult = mk3(vf[HV_F_SHADOW_MATTE_BACK_COLOR_X], vf[HV_F_SHADOW_MATTE_BACK_COLOR_X + 1], vf[HV_F_SHADOW_MATTE_BACK_COLOR_X + 2]);
  f2 tc = mk2(screenX / vf[HV_F_WIDTH_F], screenY / vf[HV_F_HEIGHT_F]);
  if (g_varsI(s)[HV_I_SHADOW_MATTE_BACK_MODE] == 1) { float sintheta = 0.0f; tc = sphereMapTo2DTexCoord(ray_dir, sintheta); }
  const float4 c = read_imagef_sw4(s.texStorage + offset, tc, HTEX_CLAMP_U | HTEX_CLAMP_V, true, s.srgbLut);
  f3 env = mult * mk3(c.x, c.y, c.z);
  if (vf[HV_F_BACK_TEXINPUT_GAMMA] != 1.0f) env = mk3(sRGBToLinear(env.x), sRGBToLinear(env.y), sRGBToLinear(env.z));   // on top of the fetch's own decode, as the reference has it
  return env;
}
template <int F = HK_FEAT_ALL>
HK_DEV f3 environmentColorExtended(const SceneDev& s, f3 ray_pos, f3 ray_dir, float prevPdf, bool prevSpecular, uint32_t flags, int screenX, int screenY) {
  const int hitId = hitDirectLight(s, ray_dir);
  if (hitId >= 0) {
    const float* sun = reinterpret_cast<const float*>(s.globals) + HG_SUNS + hitId * HL_FLOATS;
    f3 envColor = lightColor(sun) * directLightAttenuation(sun, ray_pos);
    const float pdfW = directLightEvalPDF(sun, ray_dir);
    const uint32_t gflags = uint32_t(s.hdr[HG_FLAGS]);
    if (((flags >> 8) & 0xFFu) > 0 && !(gflags & HF_STUPID_PT_MODE) && !prevSpecular) envColor = mk3(0, 0, 0);
    else if ((prevSpecular && (gflags & HF_ENABLE_PT_CAUSTICS)) || (gflags & HF_STUPID_PT_MODE)) envColor = envColor * (1.0f / pdfW);
    if (gflags & HF_3WAY_MIS_WEIGHTS) envColor = mk3(0, 0, 0);
    return envColor;
  }
  f3 envColor = environmentColor<F>(s, ray_dir, prevPdf, prevSpecular, flags);
  const uint32_t rayBounce = (flags >> 8) & 0xFFu, other = flags >> 16;
  const bool transparent = (other & 8u) != 0 && (other & 2u) == 0 && (other & 4u) == 0;   // RAY_EVENT_T without _D and _G, cglobals.h:1333-1336
  if (rayBounce == 0 || transparent) envColor = backColorOfSecondEnv(s, ray_dir, float(screenX) + 0.5f, float(screenY) + 0.5f);
  return envColor;
}
// lightGetIntensity, clight.h:1661-1706: what the light a path has run into sends back along the ray
template <int F = HK_FEAT_ALL>
HK_DEV f3 lightGetIntensity(const SceneDev& s, const float* L, f3 ray_pos, f3 ray_dir, f2 texCoord, uint32_t flags, bool wasSpecular) {
  const int type = as_int(L[HL_TYPE]);
  if ((F & HK_FEAT_RARE_LIGHTS) && (as_int(L[HL_FLAGS]) & HLF_SKY_PORTAL) && (flags & 0xFFu) > 0) {
    const int hitId = hitDirectLight(s, ray_dir);
    if (hitId >= 0) {   // the ray looks into a sun through the portal
      const float* sun = reinterpret_cast<const float*>(s.globals) + HG_SUNS + hitId * HL_FLOATS;
      f3 sunColor = lightColor(sun) * directLightAttenuation(sun, ray_pos);
      const float pdfW = directLightEvalPDF(sun, ray_dir);
      const uint32_t gflags = uint32_t(s.hdr[HG_FLAGS]);
      if (((flags >> 8) & 0xFFu) > 0 && !(gflags & HF_STUPID_PT_MODE) && !wasSpecular) sunColor = mk3(0, 0, 0);
      else if ((wasSpecular && (gflags & HF_ENABLE_PT_CAUSTICS)) || (gflags & HF_STUPID_PT_MODE)) sunColor = sunColor * (1.0f / pdfW);
      return sunColor;
    }
    return areaLightSkyPortalCustomColor<F>(s, L, ray_dir);
  }
  if (type == HLT_AREA) {
    f3 customDir = ray_dir;
    if ((F & HK_FEAT_RARE_LIGHTS) && (as_int(L[HL_FLAGS]) & HLF_IES_POINT_AREA)) customDir = normalize(lightPos(L) - ray_pos);
    f3 color = areaLightIntensity<F>(s, L, customDir, (flags & 0xFFu) == 0);
    if ((F & HK_FEAT_RARE_LIGHTS) && (as_int(L[HL_FLAGS]) & HLF_SKY_PORTAL)) color = color * portalSkyColor<F>(s, L, ray_dir);
    return color;
  }
  if ((F & HK_FEAT_RARE_LIGHTS) && type == HLT_CYLINDER) return cylinderLightGetIntensity(s, L, texCoord);
  if ((F & HK_FEAT_RARE_LIGHTS) && type == HLT_MESH) return meshLightGetIntensity(s, L, texCoord);
  return lightColor(L);
}
// emissionEval, cbidir.h:653-678
template <int F = HK_FEAT_ALL>
HK_DEV f3 emissionEval(const SceneDev& s, f3 ray_pos, f3 ray_dir, const SurfaceHit& sh, uint32_t flags, bool wasSpecular, const float* pLight, const float* mat) {
  const f3 normal = sh.hfi ? sh.normal * (-1.0f) : sh.normal;
  const int lightsNum = s.hdr[HG_LIGHTS_NUM];
  bool hasIES = false;
  if (lightsNum > 0 && pLight != nullptr) hasIES = (as_int(pLight[HL_FLAGS]) & HLF_HAS_IES) != 0;
  if (dot(ray_dir, normal) >= 0.0f && !hasIES) return mk3(0, 0, 0);
  f3 out = materialEvalEmission(mat, ray_dir, normal, sh.texCoord, s);
  if ((matFlags(mat) & HMF_FORBID_EMISSIVE_GI) && (flags & 0xFFu) > 0) out = mk3(0, 0, 0);
  if (lightsNum > 0 && pLight != nullptr) out = lightGetIntensity<F>(s, pLight, ray_pos, ray_dir, sh.texCoord, flags, wasSpecular);
  return out;
}
