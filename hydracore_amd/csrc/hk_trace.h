// hk_trace.h -- BVH4 (two-level, instanced) traversal + Moeller-Trumbore for one ray per lane.
//
// Behaviour contract (rows a/T1, a/T2): closest hit exactly as BVH4InstTraverse / BVH4Traverse
// (hydra_drv/ctrace.h:841-1062, :669-838) over the reference's flattened layout: children tested with
// RayBoxIntersectionLite2 (:32-53), ordered near->far by the same 5 compare-swaps (:906-960) so that equal-t ties
// resolve to the same triangle, <=3 pushes per quad, 80-entry stack with silent drop, instance leaves re-express the
// ray in object space with the direction left un-normalised; leaf test :124-182 (u,v > -1e-6, u+v < 1+1e-6,
// t_min < t < best).  The shadow form answers "any triangle with t_min < t < t_far" which is what
// IntegratorCommon::shadowTrace computes from a closest hit (CPUExp_Integrators_Common.cpp:156-180).
//
// MI355X mapping: one ray per lane, the 4 child boxes of a quad are one 128-byte line; the traversal stack lives in
// LDS, transposed ([entry][lane]) so a wave's push/pop touches 64 consecutive banks; entries beyond HK_LDS_DEPTH
// spill to scratch, keeping the reference's 80-entry semantics without paying for them in LDS occupancy.
#pragma once
#include "hk_common.h"

#define HK_STACK_SIZE 80
#define HK_LDS_DEPTH 24
#define HK_TRACE_BLOCK 128

struct TravCounters { uint32_t quads, insts, tris, leaves; };

HK_DEV f3 SafeInverse(f3 d) {   // hydra_drv/cglobals.h:726-735
  const float ooeps = 1.0e-36f;
  f3 r;
  r.x = 1.0f / (fabsf(d.x) > ooeps ? d.x : copysignf(ooeps, d.x));
  r.y = 1.0f / (fabsf(d.y) > ooeps ? d.y : copysignf(ooeps, d.y));
  r.z = 1.0f / (fabsf(d.z) > ooeps ? d.z : copysignf(ooeps, d.z));
  return r;
}

HK_DEV float2 RayBox(f3 o, f3 inv, float4 lo4, float4 hi4) {
  const float lo = inv.x * (lo4.x - o.x), hi = inv.x * (hi4.x - o.x);
  const float lo1 = inv.y * (lo4.y - o.y), hi1 = inv.y * (hi4.y - o.y);
  const float lo2 = inv.z * (lo4.z - o.z), hi2 = inv.z * (hi4.z - o.z);
  float tmin = fminf(lo, hi), tmax = fmaxf(lo, hi);
  tmin = fmaxf(tmin, fminf(lo1, hi1)); tmax = fminf(tmax, fmaxf(lo1, hi1));
  tmin = fmaxf(tmin, fminf(lo2, hi2)); tmax = fminf(tmax, fmaxf(lo2, hi2));
  return make_float2(tmin, tmax);
}

struct HkStack {
  int* lds;       // this lane's column in the block's LDS stack
  int  stride;    // = blockDim.x
  int  spill[HK_STACK_SIZE - HK_LDS_DEPTH];
  HK_DEV void put(int top, int v) {
    if (top < HK_LDS_DEPTH) lds[top * stride] = v; else spill[top - HK_LDS_DEPTH] = v;
  }
  HK_DEV int get(int top) const {
    if (top < 0) return 0;   // the reference reads an unused slot here; the value is never acted on
    return (top < HK_LDS_DEPTH) ? lds[top * stride] : spill[top - HK_LDS_DEPTH];
  }
};

template <bool ANYHIT, bool COUNT>
HK_DEV HydraLiteHit IntersectLeaf(f3 ray_pos, f3 ray_dir, int leaf_offset, float t_min, HydraLiteHit res,
                                  const float4* __restrict__ tris, int instId, bool useInstId, TravCounters& cnt) {
  const float4 hdr = tris[leaf_offset];
  const int first = as_int(hdr.x), count = as_int(hdr.y);
  const int end = first + count * 3;
  if (COUNT) { cnt.tris += uint32_t(count); cnt.leaves++; }
  for (int a = first; a < end; a += 3) {
    const float4 d1 = tris[a], d2 = tris[a + 1], d3 = tris[a + 2];
    const f3 A = xyz(d1), B = xyz(d2), C = xyz(d3);
    const f3 edge1 = B - A, edge2 = C - A;
    const f3 pvec = cross(ray_dir, edge2);
    const f3 tvec = ray_pos - A;
    const f3 qvec = cross(tvec, edge1);
    const float invDet = 1.0f / dot(edge1, pvec);
    const float v = dot(tvec, pvec) * invDet;
    const float u = dot(qvec, ray_dir) * invDet;
    const float t = dot(edge2, qvec) * invDet;
    if (v > -1e-6f && u > -1e-6f && (u + v < 1.0f + 1e-6f) && t > t_min && t < res.t) {
      res.t = t;
      res.primId = as_int(d1.w);
      res.geomId = as_int(d2.w);
      res.instId = useInstId ? instId : as_int(d3.w);
      if (ANYHIT) return res;
    }
  }
  return res;
}

template <bool ANYHIT, bool COUNT>
HK_DEV HydraLiteHit hk_traverse(const float4* __restrict__ bvh, const float4* __restrict__ tris, const bool haveInst,
                                f3 ray_pos, f3 ray_dir, const float t_rayMin, HydraLiteHit hit, HkStack& stack, TravCounters& cnt) {
  f3 invDir = SafeInverse(ray_dir);
  int top = 0, leftNodeOffset = 1;
  bool searchingForLeaf = true;
  int instDeep = 0, instTop = 0, instId = -1;
  f3 old_pos = mk3(0, 0, 0), old_dir = mk3(0, 0, 0);

  while (top >= 0) {
    while (searchingForLeaf) {
      const float4* q = bvh + size_t(leftNodeOffset) * 8;
      const float4 n0a = q[0], n0b = q[1], n1a = q[2], n1b = q[3], n2a = q[4], n2b = q[5], n3a = q[6], n3b = q[7];
      if (COUNT) cnt.quads++;
      int c0 = as_int(n0a.w), c1 = as_int(n1a.w), c2 = as_int(n2a.w), c3 = as_int(n3a.w);
      const bool v0 = !((uint32_t(c0) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n0b.w)) == HYDRA_BVH_INVALID));
      const bool v1 = !((uint32_t(c1) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n1b.w)) == HYDRA_BVH_INVALID));
      const bool v2 = !((uint32_t(c2) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n2b.w)) == HYDRA_BVH_INVALID));
      const bool v3 = !((uint32_t(c3) == HYDRA_BVH_INVALID) && (uint32_t(as_int(n3b.w)) == HYDRA_BVH_INVALID));
      const float2 t0 = RayBox(ray_pos, invDir, n0a, n0b), t1 = RayBox(ray_pos, invDir, n1a, n1b);
      const float2 t2 = RayBox(ray_pos, invDir, n2a, n2b), t3 = RayBox(ray_pos, invDir, n3a, n3b);
      float k0 = ((t0.x <= t0.y) && (t0.y >= t_rayMin) && (t0.x <= hit.t) && v0) ? t0.x : HK_MAXFLOAT;
      float k1 = ((t1.x <= t1.y) && (t1.y >= t_rayMin) && (t1.x <= hit.t) && v1) ? t1.x : HK_MAXFLOAT;
      float k2 = ((t2.x <= t2.y) && (t2.y >= t_rayMin) && (t2.x <= hit.t) && v2) ? t2.x : HK_MAXFLOAT;
      float k3 = ((t3.x <= t3.y) && (t3.y >= t_rayMin) && (t3.x <= hit.t) && v3) ? t3.x : HK_MAXFLOAT;
#define HK_CSWAP(ka, kb, ca, cb) { const bool sw = (kb < ka); const float tk = sw ? kb : ka; kb = sw ? ka : kb; ka = tk; const int tc = sw ? cb : ca; cb = sw ? ca : cb; ca = tc; }
      HK_CSWAP(k0, k1, c0, c1) HK_CSWAP(k2, k3, c2, c3) HK_CSWAP(k0, k2, c0, c2) HK_CSWAP(k1, k3, c1, c3) HK_CSWAP(k1, k2, c1, c2)
#undef HK_CSWAP
      const bool stackHaveSpace = (top < HK_STACK_SIZE);
      if (k3 < HK_MAXFLOAT && stackHaveSpace) { stack.put(top, c3); top++; }
      if (k2 < HK_MAXFLOAT && stackHaveSpace) { stack.put(top, c2); top++; }
      if (k1 < HK_MAXFLOAT && stackHaveSpace) { stack.put(top, c1); top++; }
      if (k0 < HK_MAXFLOAT) leftNodeOffset = c0;
      else if (top >= 0) { top--; leftNodeOffset = stack.get(top); }
      searchingForLeaf = !(leftNodeOffset & int(HYDRA_BVH_LEAF)) && (top >= 0);
      leftNodeOffset = leftNodeOffset & 0x7fffffff;
      if (haveInst && top < instTop && instDeep == 1) {
        ray_pos = old_pos; ray_dir = old_dir; invDir = SafeInverse(ray_dir); instDeep = 0;
      }
    }
    if (!haveInst) {
      if (top >= 0) {
        hit = IntersectLeaf<ANYHIT, COUNT>(ray_pos, ray_dir, leftNodeOffset, t_rayMin, hit, tris, 0, false, cnt);
        if (ANYHIT && hit.primId != -1) return hit;
      }
      top--;
      leftNodeOffset = stack.get(top);
    } else if (top >= 0 && instDeep == 1) {
      hit = IntersectLeaf<ANYHIT, COUNT>(ray_pos, ray_dir, leftNodeOffset, t_rayMin, hit, tris, instId, true, cnt);
      if (ANYHIT && hit.primId != -1) return hit;
      top--;
      leftNodeOffset = stack.get(top);
    } else if (top >= 0 && instDeep == 0) {
      instDeep = 1;
      old_pos = ray_pos; old_dir = ray_dir;
      const float4* q = bvh + size_t(leftNodeOffset) * 8;
      const int nextOffset = as_int(q[0].w);
      const m44 matrix = load_m44(q + 2);
      instId = as_int(q[6].x);
      if (COUNT) cnt.insts++;
      ray_pos = mul4x3(matrix, ray_pos);
      ray_dir = mul3x3(matrix, ray_dir);   // stays un-normalised so t keeps world units
      invDir = SafeInverse(ray_dir);
      instTop = top;
      leftNodeOffset = nextOffset;
    }
    searchingForLeaf = !(leftNodeOffset & int(HYDRA_BVH_LEAF));
    leftNodeOffset = leftNodeOffset & 0x7fffffff;
    if (haveInst && top < instTop && instDeep == 1) {
      ray_pos = old_pos; ray_dir = old_dir; invDir = SafeInverse(ray_dir); instDeep = 0;
    }
  }
  return hit;
}

HK_DEV HydraLiteHit hk_miss_hit() {   // Make_Lite_Hit(MAXFLOAT, -1), hydra_drv/cglobals.h:1256-1266
  HydraLiteHit h;
  h.t = HK_MAXFLOAT; h.primId = -1; h.instId = -1; h.geomId = int(0xC0000000u);
  return h;
}
HK_DEV bool HitSome(const HydraLiteHit& h) { return (h.primId != -1) && isfinite(h.t); }
