// bvh4_builder.cpp -- binned-SAH BVH4 build + emission of the reference's flattened layout.
// Layout citations: bvh_builder/bvh_access_dll2.cpp:264-386 (triangle lists), :388-545 (quads, instance quads),
// :604-717 (root quad, mesh subtrees shared between instances).  The build algorithm itself is ours.
#include <cstdlib>
#include "bvh4_builder.h"
#include "../../include/hydra_hip.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace hydra_host {

const char* BVH4Builder::kTypeObject = "object";

namespace {
const float kInf = std::numeric_limits<float>::infinity();
inline void box_reset(float3& mn, float3& mx) { mn = float3(kInf, kInf, kInf); mx = float3(-kInf, -kInf, -kInf); }
inline float box_area(const float3& mn, const float3& mx) {
  const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
  if (dx < 0.0f || dy < 0.0f || dz < 0.0f) return 0.0f;
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}
inline float axis_of(const float3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

inline HydraBVHNode default_node() {
  HydraBVHNode n;
  n.boxMin[0] = n.boxMin[1] = n.boxMin[2] = kInf;
  n.boxMax[0] = n.boxMax[1] = n.boxMax[2] = -kInf;
  n.leftOffsetAndLeaf = HYDRA_BVH_INVALID;
  n.escapeIndex = HYDRA_BVH_INVALID;
  return n;
}
inline void set_box(HydraBVHNode& n, const float3& mn, const float3& mx) {
  n.boxMin[0] = mn.x; n.boxMin[1] = mn.y; n.boxMin[2] = mn.z;
  n.boxMax[0] = mx.x; n.boxMax[1] = mx.y; n.boxMax[2] = mx.z;
}
inline void set_link(HydraBVHNode& n, bool leaf, uint32_t offset) {
  n.leftOffsetAndLeaf = (leaf ? HYDRA_BVH_LEAF : 0u) | (offset & 0x7fffffffu);
}
}  // namespace

void BVH4Builder::ClearScene() {
  m_meshes.clear(); m_insts.clear(); m_nodes.clear(); m_primIds.clear(); m_conns.clear();
  m_outNodes.clear(); m_outTris.clear();
  m_topRoot = -1;
  statInnerQuads = statLeaves = statTriangles = 0;
}

int BVH4Builder::InstanceTriangleMeshes(InstanceInputData d, int a_treeId, int a_realInstIdBase) {
  (void)a_treeId;   // one builder object = one tree; the driver keeps a second builder for a second tree (render_driver_lite.cpp, EndScene)
  int slot = -1;
  for (size_t i = 0; i < m_meshes.size(); i++)
    if (m_meshes[i].meshId == d.meshId) slot = int(i);
  if (slot < 0) {
    MeshRec m;
    m.meshId = d.meshId;
    m.vert4f.assign(d.vert4f, d.vert4f + size_t(d.numVert) * 4);
    m.indices.assign(d.indices, d.indices + d.numIndices);
    m_meshes.push_back(std::move(m));
    slot = int(m_meshes.size()) - 1;
  }
  for (int i = 0; i < d.numInst; i++) {
    InstRec r;
    r.meshSlot = slot;
    r.realInstId = a_realInstIdBase + i;
    memcpy(r.matrix.c, d.matrices + size_t(i) * 16, 64);
    m_insts.push_back(r);
  }
  return slot;
}

// ------------------------------------------------------------------------------------------ build
int BVH4Builder::SplitSAH(std::vector<PrimRef>& prims, int begin, int end, float* a_bestCost, bool a_evalOnly) const {
  static constexpr int NBMAX = 64;
  static const int NB = [] { const char* e = getenv("HYDRA_BVH_BINS"); const int v = e ? atoi(e) : 32; return std::max(4, std::min(NBMAX, v)); }();   // tuning sweeps only
  float3 cmn, cmx;
  box_reset(cmn, cmx);
  for (int i = begin; i < end; i++) { cmn = vmin(cmn, prims[i].centroid); cmx = vmax(cmx, prims[i].centroid); }

  float bestCost = kInf;
  int bestAxis = -1, bestBin = -1;
  for (int axis = 0; axis < 3; axis++) {
    const float lo = axis_of(cmn, axis), hi = axis_of(cmx, axis);
    if (!(hi > lo)) continue;
    const float scale = float(NB) / (hi - lo);
    float cnt[NBMAX]; float3 bmn[NBMAX], bmx[NBMAX];
    for (int b = 0; b < NB; b++) { cnt[b] = 0.0f; box_reset(bmn[b], bmx[b]); }
    for (int i = begin; i < end; i++) {
      int b = int((axis_of(prims[i].centroid, axis) - lo) * scale);
      b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
      cnt[b] += prims[i].weight;
      bmn[b] = vmin(bmn[b], prims[i].box.mn); bmx[b] = vmax(bmx[b], prims[i].box.mx);
    }
    float rightArea[NBMAX], rightCnt[NBMAX];
    float3 amn, amx; box_reset(amn, amx);
    float c = 0.0f;
    for (int b = NB - 1; b > 0; b--) {
      amn = vmin(amn, bmn[b]); amx = vmax(amx, bmx[b]); c += cnt[b];
      rightArea[b] = box_area(amn, amx); rightCnt[b] = c;
    }
    box_reset(amn, amx); c = 0.0f;
    for (int b = 0; b < NB - 1; b++) {
      amn = vmin(amn, bmn[b]); amx = vmax(amx, bmx[b]); c += cnt[b];
      if (c == 0.0f || rightCnt[b + 1] == 0.0f) continue;
      static const int leafUnit = [] { const char* e = getenv("HYDRA_BVH_SAH_UNIT"); return e ? std::max(1, atoi(e)) : 1; }();   // experiment: count triangles in units of a leaf
      const float cl = (leafUnit > 1) ? ceilf(c / float(leafUnit)) : c, cr = (leafUnit > 1) ? ceilf(rightCnt[b + 1] / float(leafUnit)) : rightCnt[b + 1];
      const float cost = box_area(amn, amx) * cl + rightArea[b + 1] * cr;
      if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
    }
  }
  if (a_bestCost) *a_bestCost = bestCost;
  if (a_evalOnly) return begin;
  if (bestAxis < 0) return (begin + end) / 2;   // all centroids coincide: split in the middle
  const float lo = axis_of(cmn, bestAxis), hi = axis_of(cmx, bestAxis);
  const float scale = float(NB) / (hi - lo);
  auto mid = std::partition(prims.begin() + begin, prims.begin() + end, [&](const PrimRef& p) {
    int b = int((axis_of(p.centroid, bestAxis) - lo) * scale);
    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
    return b <= bestBin;
  });
  int m = int(mid - prims.begin());
  if (m == begin || m == end) m = (begin + end) / 2;
  return m;
}

int BVH4Builder::BuildRecursive(std::vector<PrimRef>& prims, int begin, int end, int leafMax) {
  TmpNode node;
  box_reset(node.box.mn, node.box.mx);
  for (int i = begin; i < end; i++) { node.box.mn = vmin(node.box.mn, prims[i].box.mn); node.box.mx = vmax(node.box.mx, prims[i].box.mx); }
  const int idx = int(m_nodes.size());
  m_nodes.push_back(node);

  if (end - begin <= leafMax) {
    m_nodes[idx].first = int(m_primIds.size());
    m_nodes[idx].count = end - begin;
    for (int i = begin; i < end; i++) m_primIds.push_back(prims[i].id);
    return idx;
  }
  // open the node into up to four ranges, always splitting the range with the largest box
  struct Range { int b, e; };
  Range r[4];
  int nr = 1;
  r[0] = {begin, end};
  while (nr < 4) {
    int pick = -1;
    float best = -1.0f;
    for (int i = 0; i < nr; i++) {
      const int cnt = r[i].e - r[i].b;
      if (cnt <= leafMax) continue;
      float3 mn, mx; box_reset(mn, mx);
      for (int k = r[i].b; k < r[i].e; k++) { mn = vmin(mn, prims[k].box.mn); mx = vmax(mx, prims[k].box.mx); }
      // which range to split next: the one with the largest box (profiles/r03/ab_open_node_rule.log: closest-hit traversal -2.5 % on atrium250k, shadow traversal -3 % on
      // test_224, +0.9 % on both passes against area x count, the rule of rounds 1-3a); HYDRA_BVH_OPEN = product | count selects the others for A/B
      static const char* const openEnv = getenv("HYDRA_BVH_OPEN");
      float w = (openEnv && openEnv[0] == 'p') ? box_area(mn, mx) * float(cnt) + 1e-30f * float(cnt)
              : (openEnv && openEnv[0] == 'c') ? float(cnt)
              : box_area(mn, mx) + 1e-30f * float(cnt);
      if (openEnv && openEnv[0] == 'g') {   // experiment: the range whose best split takes the most off the SAH cost
        float splitCost = kInf;
        (void)SplitSAH(prims, r[i].b, r[i].e, &splitCost, true);
        w = (splitCost < kInf) ? box_area(mn, mx) * float(cnt) - splitCost : 0.0f;
      }
      if (w > best) { best = w; pick = i; }
    }
    if (pick < 0) break;
    const int m = SplitSAH(prims, r[pick].b, r[pick].e);
    const Range left = {r[pick].b, m}, right = {m, r[pick].e};
    r[pick] = left;
    r[nr++] = right;
  }
  // Order of the children inside the quad.  Closest-hit rays sort them by distance anyway; any-hit (shadow) rays take them as stored (hk_trace.h, UNORD), so the
  // order decides how soon an occluder is found.  HYDRA_BVH_CHILD_ORDER = area: largest box first; count: most triangles first; unset: the split order.
  static const char* const orderEnv = getenv("HYDRA_BVH_CHILD_ORDER");
  if (orderEnv != nullptr && (orderEnv[0] == 'a' || orderEnv[0] == 'c')) {
    float key[4];
    for (int i = 0; i < nr; i++) {
      if (orderEnv[0] == 'c') { key[i] = float(r[i].e - r[i].b); continue; }
      float3 mn, mx; box_reset(mn, mx);
      for (int k = r[i].b; k < r[i].e; k++) { mn = vmin(mn, prims[k].box.mn); mx = vmax(mx, prims[k].box.mx); }
      key[i] = box_area(mn, mx);
    }
    for (int i = 1; i < nr; i++)
      for (int j = i; j > 0 && key[j] > key[j - 1]; j--) { std::swap(key[j], key[j - 1]); std::swap(r[j], r[j - 1]); }
  }
  for (int i = 0; i < nr; i++) {
    const int c = BuildRecursive(prims, r[i].b, r[i].e, leafMax);
    m_nodes[idx].child[i] = c;
  }
  return idx;
}

int BVH4Builder::BuildTree(std::vector<PrimRef>& prims, int leafMax) {
  if (prims.empty()) return -1;
  return BuildRecursive(prims, 0, int(prims.size()), leafMax);
}

void BVH4Builder::CommitScene() {
  m_nodes.clear(); m_primIds.clear();
  statGpuBuildMs = 0.0f;
  // (1) one tree per mesh, over its non-degenerate triangles (zero-area triangles are dropped, bvh_access_dll2.cpp:354-355)
  for (auto& mesh : m_meshes) {
    const int triNum = int(mesh.indices.size() / 3);
    std::vector<PrimRef> prims;
    prims.reserve(triNum);
    const float* v = mesh.vert4f.data();
    for (int t = 0; t < triNum; t++) {
      const int ia = mesh.indices[t * 3 + 0], ib = mesh.indices[t * 3 + 1], ic = mesh.indices[t * 3 + 2];
      const float3 A(v[ia * 4], v[ia * 4 + 1], v[ia * 4 + 2]), B(v[ib * 4], v[ib * 4 + 1], v[ib * 4 + 2]), C(v[ic * 4], v[ic * 4 + 1], v[ic * 4 + 2]);
      const float area = 0.5f * length(cross(B - A, C - A));
      if (!(area > 0.0f)) continue;
      PrimRef p;
      p.box.mn = vmin(A, vmin(B, C));
      p.box.mx = vmax(A, vmax(B, C));
      p.centroid = (p.box.mn + p.box.mx) * 0.5f;
      p.id = t;
      prims.push_back(p);
    }
    if (prims.empty()) RunTimeError("BVH4Builder::CommitScene: mesh without valid triangles");
    // Experiment (HYDRA_BVH_PRESPLIT = extra references in percent, off by default): early split clipping.  The references with the largest boxes are cut in two along their
    // longest axis, each half bounded by the part of the triangle inside it, until the budget is spent: a large triangle then sits in several small leaves instead of
    // inflating one (the leaf lists hold triangles by value, a triangle may appear in more than one; a second hit on it has the same t and is not nearer).
    if (const char* e = getenv("HYDRA_BVH_PRESPLIT")) {
      const int budget = int(double(prims.size()) * std::max(0, std::min(400, atoi(e))) / 100.0);
      auto areaOf = [](const PrimRef& r) { return box_area(r.box.mn, r.box.mx); };
      auto cmp = [&](int a, int b) { return areaOf(prims[size_t(a)]) < areaOf(prims[size_t(b)]); };
      std::vector<int> heap(prims.size());
      for (size_t i = 0; i < prims.size(); i++) heap[i] = int(i);
      std::make_heap(heap.begin(), heap.end(), cmp);
      auto clipBox = [&](int tri, int axis, float lo, float hi, const Box& within, Box& out) -> bool {   // the triangle cut to lo <= x[axis] <= hi, inside `within`
        const int ia = mesh.indices[tri * 3 + 0], ib = mesh.indices[tri * 3 + 1], ic = mesh.indices[tri * 3 + 2];
        float3 poly[8], tmp[8];
        int n = 3;
        poly[0] = float3(v[ia * 4], v[ia * 4 + 1], v[ia * 4 + 2]); poly[1] = float3(v[ib * 4], v[ib * 4 + 1], v[ib * 4 + 2]); poly[2] = float3(v[ic * 4], v[ic * 4 + 1], v[ic * 4 + 2]);
        for (int side = 0; side < 2 && n > 0; side++) {
          const float plane = side == 0 ? lo : hi, sgn = side == 0 ? 1.0f : -1.0f;
          int m = 0;
          for (int k = 0; k < n; k++) {
            const float3 P = poly[k], Q = poly[(k + 1) % n];
            const float dp = sgn * (axis_of(P, axis) - plane), dq = sgn * (axis_of(Q, axis) - plane);
            if (dp >= 0.0f) tmp[m++] = P;
            if ((dp >= 0.0f) != (dq >= 0.0f)) { const float t = dp / (dp - dq); tmp[m++] = P + (Q - P) * t; }
          }
          n = m;
          for (int k = 0; k < n; k++) poly[k] = tmp[k];
        }
        if (n == 0) return false;
        box_reset(out.mn, out.mx);
        for (int k = 0; k < n; k++) { out.mn = vmin(out.mn, poly[k]); out.mx = vmax(out.mx, poly[k]); }
        out.mn = vmax(out.mn, within.mn); out.mx = vmin(out.mx, within.mx);
        return out.mn.x <= out.mx.x && out.mn.y <= out.mx.y && out.mn.z <= out.mx.z;
      };
      int added = 0;
      while (added < budget && !heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        const int at = heap.back(); heap.pop_back();
        const PrimRef r = prims[size_t(at)];
        const float3 ext = r.box.mx - r.box.mn;
        const int axis = (ext.x >= ext.y && ext.x >= ext.z) ? 0 : (ext.y >= ext.z ? 1 : 2);
        if (!(axis_of(ext, axis) > 1e-5f)) continue;
        const float mid = 0.5f * (axis_of(r.box.mn, axis) + axis_of(r.box.mx, axis));
        PrimRef a = r, b = r;
        if (!clipBox(r.id, axis, axis_of(r.box.mn, axis), mid, r.box, a.box) || !clipBox(r.id, axis, mid, axis_of(r.box.mx, axis), r.box, b.box)) continue;
        a.centroid = (a.box.mn + a.box.mx) * 0.5f; b.centroid = (b.box.mn + b.box.mx) * 0.5f;
        prims[size_t(at)] = a;
        prims.push_back(b);
        heap.push_back(at); std::push_heap(heap.begin(), heap.end(), cmp);
        heap.push_back(int(prims.size()) - 1); std::push_heap(heap.begin(), heap.end(), cmp);
        added++;
      }
    }
    int leafMax = maxLeafSize;
    if (const char* e = getenv("HYDRA_BVH_MAX_LEAF")) leafMax = std::max(1, std::min(16, atoi(e)));   // tuning sweeps only
    if (gpuBuildDevice >= 0) {   // the tree of this mesh comes from the device in build form; node and primitive indices are rebased into the shared arrays
      std::vector<HydraBuildNode> gn(size_t(triNum) * 2);
      std::vector<int32_t> order(static_cast<size_t>(triNum));
      int32_t nodeCount = 0, primCount = 0;
      float ms = 0.0f;
      int method = HYDRA_BVH_PLOC, radius = 128;                                     // A/B switches: HYDRA_GPU_BVH_METHOD=lbvh|ploc, HYDRA_GPU_BVH_RADIUS=1..128
      if (const char* e = getenv("HYDRA_GPU_BVH_METHOD")) method = (std::string(e) == "lbvh") ? HYDRA_BVH_LBVH : HYDRA_BVH_PLOC;
      if (const char* e = getenv("HYDRA_GPU_BVH_RADIUS")) radius = std::max(1, std::min(128, atoi(e)));
      const int rc = hydra_hip_bvh_build_mesh_ex(gpuBuildDevice, mesh.vert4f.data(), int(mesh.vert4f.size() / 4), mesh.indices.data(), int(mesh.indices.size()), leafMax, method, radius,
                                                 gn.data(), &nodeCount, order.data(), &primCount, &ms);
      if (rc != HYDRA_HIP_OK) RunTimeError(std::string("BVH4Builder::CommitScene: GPU build failed: ") + hydra_hip_bvh_last_error());
      statGpuBuildMs += ms;
      const int nodeBase = int(m_nodes.size()), primBase = int(m_primIds.size());
      for (int k = 0; k < nodeCount; k++) {
        TmpNode tn;
        tn.box.mn = float3(gn[k].boxMin[0], gn[k].boxMin[1], gn[k].boxMin[2]);
        tn.box.mx = float3(gn[k].boxMax[0], gn[k].boxMax[1], gn[k].boxMax[2]);
        for (int c = 0; c < 4; c++) tn.child[c] = gn[k].child[c] >= 0 ? nodeBase + gn[k].child[c] : -1;
        tn.first = gn[k].count > 0 ? primBase + gn[k].first : 0;
        tn.count = gn[k].count;
        m_nodes.push_back(tn);
      }
      m_primIds.insert(m_primIds.end(), order.begin(), order.begin() + primCount);
      mesh.rootNode = nodeBase;
    } else
      mesh.rootNode = BuildTree(prims, leafMax);
    mesh.bounds = m_nodes[mesh.rootNode].box;
  }
  // (2) top level over instances
  std::vector<PrimRef> iprims;
  box_reset(m_sceneBox.mn, m_sceneBox.mx);
  for (size_t i = 0; i < m_insts.size(); i++) {
    InstRec& in = m_insts[i];
    const Box& mb = m_meshes[in.meshSlot].bounds;
    float3 mn, mx; box_reset(mn, mx);
    // the world box of an instance: of its transformed vertices -- for a rotated object much tighter than the box of the eight transformed corners of its local box, and
    // every ray that misses the tighter box skips an instance entry (a third of the traversal turns on atrium250k).  HYDRA_BVH_INST_BOX=corners: the old bound, for A/B;
    // also the fallback when vertices x instances would make the exact bound expensive.
    static const bool cornersOnly = [] { const char* e = getenv("HYDRA_BVH_INST_BOX"); return e != nullptr && e[0] == 'c'; }();
    const std::vector<float>& mv = m_meshes[in.meshSlot].vert4f;
    if (!cornersOnly && double(mv.size() / 4) * double(m_insts.size()) <= 4.0e8) {
      for (size_t k = 0; k + 3 < mv.size(); k += 4) {
        const float3 w = mul_point(in.matrix, float3(mv[k], mv[k + 1], mv[k + 2]));
        mn = vmin(mn, w); mx = vmax(mx, w);
      }
    } else
      for (int k = 0; k < 8; k++) {
        const float3 c((k & 1) ? mb.mx.x : mb.mn.x, (k & 2) ? mb.mx.y : mb.mn.y, (k & 4) ? mb.mx.z : mb.mn.z);
        const float3 w = mul_point(in.matrix, c);
        mn = vmin(mn, w); mx = vmax(mx, w);
      }
    // guard against rounding of the 8-corner bound: pad by a few ulp of the extent
    const float3 ext = mx - mn;
    const float pad = 1e-5f * fmaxf(fmaxf(ext.x, ext.y), fmaxf(ext.z, 1e-20f));
    in.worldBox.mn = mn - float3(pad, pad, pad);
    in.worldBox.mx = mx + float3(pad, pad, pad);
    PrimRef p;
    p.box = in.worldBox;
    p.centroid = (p.box.mn + p.box.mx) * 0.5f;
    p.id = int(i);
    // experiment (HYDRA_BVH_INST_WEIGHT=1): an instance weighs what walking its mesh tree costs, ~log2 of its triangles, instead of 1
    static const bool instWeight = getenv("HYDRA_BVH_INST_WEIGHT") != nullptr;
    if (instWeight) p.weight = log2f(float(m_meshes[in.meshSlot].indices.size() / 3) + 2.0f);
    iprims.push_back(p);
    m_sceneBox.mn = vmin(m_sceneBox.mn, in.worldBox.mn);
    m_sceneBox.mx = vmax(m_sceneBox.mx, in.worldBox.mx);
  }
  if (iprims.empty()) RunTimeError("BVH4Builder::CommitScene: no instances in the scene");
  m_topRoot = BuildTree(iprims, 1);
}

void BVH4Builder::GetBounds(float a_bMin[3], float a_bMax[3]) const {
  a_bMin[0] = m_sceneBox.mn.x; a_bMin[1] = m_sceneBox.mn.y; a_bMin[2] = m_sceneBox.mn.z;
  a_bMax[0] = m_sceneBox.mx.x; a_bMax[1] = m_sceneBox.mx.y; a_bMax[2] = m_sceneBox.mx.z;
}

// ------------------------------------------------------------------------------------------ emission
size_t BVH4Builder::Alloc4Nodes() {
  const size_t o = m_outNodes.size();
  for (int i = 0; i < 4; i++) m_outNodes.push_back(default_node());
  return o;
}

size_t BVH4Builder::EmitTriangleLeaf(const MeshRec& mesh, const TmpNode& leaf) {
  const size_t listOffset = m_outTris.size() / 4;
  m_outTris.resize(m_outTris.size() + 4);
  const float* v = mesh.vert4f.data();
  for (int k = 0; k < leaf.count; k++) {
    const int t = m_primIds[leaf.first + k];
    const int ia = mesh.indices[t * 3 + 0], ib = mesh.indices[t * 3 + 1], ic = mesh.indices[t * 3 + 2];
    const int32_t ids[3] = {t, mesh.meshId, -1};
    const int iv[3] = {ia, ib, ic};
    for (int c = 0; c < 3; c++) {
      float w;
      memcpy(&w, &ids[c], 4);
      m_outTris.push_back(v[iv[c] * 4 + 0]); m_outTris.push_back(v[iv[c] * 4 + 1]); m_outTris.push_back(v[iv[c] * 4 + 2]);
      m_outTris.push_back(w);
    }
  }
  const int32_t hdr[4] = {int32_t(listOffset + 1), leaf.count, -1, -1};
  memcpy(&m_outTris[listOffset * 4], hdr, 16);
  statLeaves++;
  statTriangles += leaf.count;
  return listOffset;
}

size_t BVH4Builder::EmitMeshSubtree(const MeshRec& mesh, int tmp, size_t curr) {
  const TmpNode node = m_nodes[tmp];
  if (node.count > 0) {
    const size_t list = EmitTriangleLeaf(mesh, node);
    set_link(m_outNodes[curr], true, uint32_t(list));
    m_outNodes[curr].escapeIndex = 0;
    return size_t(-1);
  }
  const size_t quad = Alloc4Nodes();
  statInnerQuads++;
  for (int i = 0; i < 4; i++)
    if (node.child[i] >= 0) set_box(m_outNodes[quad + i], m_nodes[node.child[i]].box.mn, m_nodes[node.child[i]].box.mx);
  set_link(m_outNodes[curr], false, uint32_t(quad / 4));
  for (int i = 0; i < 4; i++)
    if (node.child[i] >= 0) EmitMeshSubtree(mesh, node.child[i], quad + i);
  return quad;
}

void BVH4Builder::EmitTop(int tmp, size_t curr) {
  const TmpNode node = m_nodes[tmp];
  if (node.count > 0) {  // instance leaf
    const InstRec& in = m_insts[m_primIds[node.first]];
    if (curr == 0) {     // single-instance scene: synthetic top quad (bvh_access_dll2.cpp:458-484)
      const size_t q = Alloc4Nodes();
      set_link(m_outNodes[0], false, uint32_t(q / 4));
      set_box(m_outNodes[q], in.worldBox.mn, in.worldBox.mx);
      curr = q;
    }
    const size_t q = Alloc4Nodes();
    set_link(m_outNodes[curr], true, uint32_t(q / 4));
    m_outNodes[curr].escapeIndex = 1;  // SetInstance(1)
    const MeshRec& mesh = m_meshes[in.meshSlot];
    set_box(m_outNodes[q], mesh.bounds.mn, mesh.bounds.mx);
    const float4x4 inv = inverse4x4(in.matrix);
    memcpy(&m_outNodes[q + 1], inv.c, 64);
    const int32_t ids[4] = {in.realInstId, mesh.meshId, 0, 0};
    memcpy(&m_outNodes[q + 3], ids, 16);
    m_conns.push_back({q, in.meshSlot});
    return;
  }
  const size_t quad = Alloc4Nodes();
  statInnerQuads++;
  for (int i = 0; i < 4; i++)
    if (node.child[i] >= 0) set_box(m_outNodes[quad + i], m_nodes[node.child[i]].box.mn, m_nodes[node.child[i]].box.mx);
  set_link(m_outNodes[curr], false, uint32_t(quad / 4));
  for (int i = 0; i < 4; i++)
    if (node.child[i] >= 0) EmitTop(node.child[i], quad + i);
}

ConvertionResult BVH4Builder::ConvertMap() {
  if (m_topRoot < 0) RunTimeError("BVH4Builder::ConvertMap: CommitScene was not called");
  m_outNodes.clear(); m_outTris.clear(); m_conns.clear();
  statInnerQuads = statLeaves = statTriangles = 0;

  const size_t root = Alloc4Nodes();               // quad 0: root box + identity matrix in nodes 1-2
  set_box(m_outNodes[root], m_sceneBox.mn, m_sceneBox.mx);
  float4x4 ident;
  memcpy(&m_outNodes[root + 1], ident.c, 64);
  EmitTop(m_topRoot, root);

  std::vector<uint32_t> meshLink(m_meshes.size(), 0u);
  std::vector<char> meshDone(m_meshes.size(), 0);
  for (const Conn& c : m_conns) {
    if (!meshDone[c.meshSlot]) {
      EmitMeshSubtree(m_meshes[c.meshSlot], m_meshes[c.meshSlot].rootNode, c.instNode0);
      meshLink[c.meshSlot] = m_outNodes[c.instNode0].leftOffsetAndLeaf;
      meshDone[c.meshSlot] = 1;
    } else
      m_outNodes[c.instNode0].leftOffsetAndLeaf = meshLink[c.meshSlot];   // all instances share one mesh subtree
    m_outNodes[c.instNode0].escapeIndex = HYDRA_BVH_INVALID;
  }

  ConvertionResult res;
  res.treesNum = 1;
  res.bvhType[0] = kTypeObject;
  res.pBVH[0] = m_outNodes.data();
  res.pTriangleData[0] = m_outTris.data();
  res.nodesNum[0] = int(m_outNodes.size());
  res.trif4Num[0] = int(m_outTris.size() / 4);
  return res;
}

void BVH4Builder::ConvertUnmap() {
  m_outNodes = std::vector<HydraBVHNode>();
  m_outTris = std::vector<float>();
  m_conns.clear();
}

}  // namespace hydra_host
