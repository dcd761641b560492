// hydra_bvh.hip -- GPU-side build of the per-mesh BVH4 trees (row f2 of SURVEY.md 8: "BVH build / converted-layout compatibility").
//
// The reference gets its flattened BVH4 from Embree 2.17 through IBVHBuilder2 (hydra_drv/IBVHBuilderAPI.h:35-68,
// bvh_builder/bvh_access_dll2.cpp:388-717); this repository's default builder is a binned-SAH build on the host
// (hydracore_amd/host/bvh4_builder.cpp).  This file is the MI355X-native alternative for the part that scales with the triangle count:
//   k_bvh_prims      triangle boxes + centroid bounds (degenerate triangles dropped, bvh_access_dll2.cpp:354-355)
//   k_bvh_morton     30-bit Morton code of the centroid
//   radix sort       stable LSD sort of (code, triangle) pairs, 4 bits per pass, wave-ballot ranking (hand-written: no library sort)
//   k_ploc_*         the binary tree over the sorted triangles by parallel locally-ordered clustering (Meister & Bittner 2018): rounds of
//                    nearest-neighbour search in a window of the Morton order, merging of mutual pairs, order-preserving compaction
//   (k_bvh_hierarchy + k_bvh_refit: the LBVH alternative -- binary radix tree of the codes alone, Karras 2012 -- kept for A/B, HYDRA_BVH_LBVH)
//   k_bvh_collapse   level-synchronous collapse of the binary tree into 4-wide nodes with leaves of <= leafMax triangles
// The result is the builder's build-form node array (box, 4 children, leaf range) + the triangle order; emission into the reference's
// quad / triangle-list layout stays in bvh4_builder.cpp, which the SAH path shares.  Measured on atrium250k (profiles/r03/bvh_ab_atrium250k.log):
// whole-pass rate on PLOC trees (window 128) 9 % below the host's binned-SAH trees, on LBVH trees 22 % below; build 45-80 Mtris/s of device time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <algorithm>
#include "../../include/hydra_hip.h"

namespace {

#define BCHECK(call)                                                                                        \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) { g_bvhError = std::string(#call) + ": " + hipGetErrorString(e_); return HYDRA_HIP_EDEVICE; } \
  } while (0)
thread_local std::string g_bvhError;

struct Box3 { float mn[3], mx[3]; };

__device__ __forceinline__ void atomicMinF(float* addr, float v) {   // IEEE order trick: non-negative floats order as ints, negative ones reversed as unsigned
  if (v >= 0.0f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v)); else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomicMaxF(float* addr, float v) {
  if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v)); else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// per triangle: box, validity; bounds6 = min xyz, max xyz of the centroids of the valid triangles
__global__ void k_bvh_prims(int triNum, const float4* __restrict__ vert, const int* __restrict__ indices, Box3* __restrict__ boxes, int* __restrict__ valid, float* __restrict__ bounds6) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= triNum) return;
  const float4 A = vert[indices[3 * t]], B = vert[indices[3 * t + 1]], C = vert[indices[3 * t + 2]];
  const float e1x = B.x - A.x, e1y = B.y - A.y, e1z = B.z - A.z, e2x = C.x - A.x, e2y = C.y - A.y, e2z = C.z - A.z;
  const float cx = e1y * e2z - e1z * e2y, cy = e1z * e2x - e1x * e2z, cz = e1x * e2y - e1y * e2x;
  const float area = 0.5f * sqrtf(cx * cx + cy * cy + cz * cz);
  Box3 b;
  b.mn[0] = fminf(A.x, fminf(B.x, C.x)); b.mn[1] = fminf(A.y, fminf(B.y, C.y)); b.mn[2] = fminf(A.z, fminf(B.z, C.z));
  b.mx[0] = fmaxf(A.x, fmaxf(B.x, C.x)); b.mx[1] = fmaxf(A.y, fmaxf(B.y, C.y)); b.mx[2] = fmaxf(A.z, fmaxf(B.z, C.z));
  boxes[t] = b;
  const bool ok = (area > 0.0f);
  valid[t] = ok ? 1 : 0;
  if (ok) for (int a = 0; a < 3; a++) { const float c = 0.5f * (b.mn[a] + b.mx[a]); atomicMinF(bounds6 + a, c); atomicMaxF(bounds6 + 3 + a, c); }
}
__device__ __forceinline__ uint32_t expandBits10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu; v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
// invalid triangles get the largest key and sort behind the valid ones
__global__ void k_bvh_morton(int triNum, const Box3* __restrict__ boxes, const int* __restrict__ valid, const float* __restrict__ bounds6, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= triNum) return;
  uint32_t code = 0xFFFFFFFFu;
  if (valid[t]) {
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
      const float lo = bounds6[a], ext = bounds6[3 + a] - lo;
      const float c = 0.5f * (boxes[t].mn[a] + boxes[t].mx[a]);
      const float u = ext > 0.0f ? (c - lo) / ext : 0.0f;
      q[a] = uint32_t(fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f));
    }
    code = (expandBits10(q[0]) << 2) | (expandBits10(q[1]) << 1) | expandBits10(q[2]);
  }
  keys[t] = code; vals[t] = uint32_t(t);
}

// ---- stable LSD radix sort, 4 bits per pass, 256-thread tiles; ranking inside a tile by wave ballots
#define RS_BLOCK 256
#define RS_WAVES (RS_BLOCK / 64)
__device__ __forceinline__ void rs_rank(uint32_t digit, bool live, int& rankInTile, int (*sWave)[16], int* sTile) {
  const int lane = int(__lane_id()), wave = int(threadIdx.x) / 64;
  int rankInWave = 0;
  for (uint32_t d = 0; d < 16; d++) {
    const unsigned long long m = __ballot(live && digit == d);
    if (live && digit == d) rankInWave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) sWave[wave][d] = __popcll(m);
  }
  __syncthreads();
  int before = 0;
  if (live) for (int w = 0; w < wave; w++) before += sWave[w][digit];
  rankInTile = before + rankInWave;
  if (threadIdx.x < 16) { int s = 0; for (int w = 0; w < RS_WAVES; w++) s += sWave[w][threadIdx.x]; sTile[threadIdx.x] = s; }
  __syncthreads();
}
__global__ void __launch_bounds__(RS_BLOCK) k_rs_hist(int n, const uint32_t* __restrict__ keys, int shift, int* __restrict__ hist, int numTiles) {
  __shared__ int sWave[RS_WAVES][16]; __shared__ int sTile[16];
  const int i = blockIdx.x * RS_BLOCK + threadIdx.x;
  const bool live = i < n;
  const uint32_t digit = live ? (keys[i] >> shift) & 15u : 0u;
  int r;
  rs_rank(digit, live, r, sWave, sTile);
  if (threadIdx.x < 16) hist[threadIdx.x * numTiles + blockIdx.x] = sTile[threadIdx.x];   // digit-major: the scan then yields the start of (digit, tile)
}
__global__ void __launch_bounds__(256) k_rs_scan(int count, int* __restrict__ data) {   // exclusive scan in place, one block
  __shared__ int part[256];
  const int per = (count + 255) / 256, b = threadIdx.x * per, e = min(b + per, count);
  int s = 0;
  for (int i = b; i < e; i++) s += data[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int acc = 0; for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = acc; acc += v; } }
  __syncthreads();
  int acc = part[threadIdx.x];
  for (int i = b; i < e; i++) { const int v = data[i]; data[i] = acc; acc += v; }
}
__global__ void __launch_bounds__(RS_BLOCK) k_rs_scatter(int n, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int shift, const int* __restrict__ hist, int numTiles,
                                                        uint32_t* __restrict__ keysOut, uint32_t* __restrict__ valsOut) {
  __shared__ int sWave[RS_WAVES][16]; __shared__ int sTile[16];
  const int i = blockIdx.x * RS_BLOCK + threadIdx.x;
  const bool live = i < n;
  const uint32_t key = live ? keys[i] : 0u, digit = (key >> shift) & 15u;
  int r;
  rs_rank(digit, live, r, sWave, sTile);
  if (live) { const int dst = hist[digit * numTiles + blockIdx.x] + r; keysOut[dst] = key; valsOut[dst] = vals[i]; }
}

// ---- binary radix tree over the n sorted keys (Karras 2012).  Leaves are addressed as n - 1 + i, internal nodes 0 .. n - 2, root 0.
__device__ __forceinline__ int delta(const uint32_t* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint32_t a = keys[i], b = keys[j];
  return (a == b) ? 32 + __clz(uint32_t(i) ^ uint32_t(j)) : __clz(a ^ b);
}
__global__ void k_bvh_hierarchy(int n, const uint32_t* __restrict__ keys, int2* __restrict__ children, int* __restrict__ parent, int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2) if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) / 2; ; t = (t + 1) / 2) {
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t <= 1) break;
  }
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  const int left = (lo == gamma) ? (n - 1 + gamma) : gamma, right = (hi == gamma + 1) ? (n - 1 + gamma + 1) : (gamma + 1);
  children[i] = make_int2(left, right);
  parent[left] = i; parent[right] = i;
  count[i] = hi - lo + 1;          // leaves below node i (count[n - 1 + k] = 1 is set by the host)
  if (i == 0) parent[0] = -1;
}
__device__ __forceinline__ Box3 loadBoxCoherent(const Box3* p) {
  Box3 b;
  const float* f = reinterpret_cast<const float*>(p);
  for (int k = 0; k < 3; k++) { b.mn[k] = __hip_atomic_load(f + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b.mx[k] = __hip_atomic_load(f + 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  return b;
}
__global__ void k_bvh_refit(int n, const uint32_t* __restrict__ sortedTri, const Box3* __restrict__ triBoxes, const int2* __restrict__ children, const int* __restrict__ parent,
                            Box3* __restrict__ nodeBoxes, int* __restrict__ arrived, int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  nodeBoxes[n - 1 + i] = triBoxes[sortedTri[i]];
  count[n - 1 + i] = 1;
  __threadfence();
  int node = parent[n - 1 + i];
  while (node >= 0) {
    if (atomicAdd(&arrived[node], 1) == 0) return;      // the first child to arrive leaves; the second one sees both boxes
    __threadfence();
    const int2 ch = children[node];
    const Box3 a = loadBoxCoherent(nodeBoxes + ch.x), b = loadBoxCoherent(nodeBoxes + ch.y);   // written by another CU a moment ago: not through this CU's vector cache
    Box3 r;
    for (int k = 0; k < 3; k++) { r.mn[k] = fminf(a.mn[k], b.mn[k]); r.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
    nodeBoxes[node] = r;
    __threadfence();
    node = parent[node];
  }
}

// ---- PLOC (parallel locally-ordered clustering, Meister & Bittner 2018): agglomerative build over the Morton order.  Every round each cluster
// looks `radius` places to either side for the neighbour whose union with it has the smallest surface area; clusters that chose each other merge,
// the array is compacted in order, until one cluster is left.  Trees come out close to a top-down SAH build's (the LBVH above splits by code bits
// alone).  Node ids as above: leaf k = n - 1 + k; inner nodes are handed out from n - 2 downwards, so that the last merge makes node 0, the root.
__global__ void k_ploc_init(int n, const uint32_t* __restrict__ sortedTri, const Box3* __restrict__ triBoxes, int* __restrict__ clusterId, Box3* __restrict__ clusterBox,
                            Box3* __restrict__ nodeBoxes, int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Box3 b = triBoxes[sortedTri[i]];
  clusterId[i] = n - 1 + i; clusterBox[i] = b;
  nodeBoxes[n - 1 + i] = b; count[n - 1 + i] = 1;
}
__device__ __forceinline__ float unionArea(const Box3& a, const Box3& b) {
  const float dx = fmaxf(a.mx[0], b.mx[0]) - fminf(a.mn[0], b.mn[0]), dy = fmaxf(a.mx[1], b.mx[1]) - fminf(a.mn[1], b.mn[1]), dz = fmaxf(a.mx[2], b.mx[2]) - fminf(a.mn[2], b.mn[2]);
  return dx * dy + dy * dz + dz * dx;
}
#define PLOC_BLOCK 256
#define PLOC_MAX_RADIUS 128
__global__ void __launch_bounds__(PLOC_BLOCK) k_ploc_nn(int m, int radius, const Box3* __restrict__ clusterBox, int* __restrict__ nn) {
  __shared__ Box3 tile[PLOC_BLOCK + 2 * PLOC_MAX_RADIUS];
  const int base = blockIdx.x * PLOC_BLOCK - radius;
  for (int k = threadIdx.x; k < PLOC_BLOCK + 2 * radius; k += PLOC_BLOCK) { const int g = base + k; if (g >= 0 && g < m) tile[k] = clusterBox[g]; }
  __syncthreads();
  const int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
  if (i >= m) return;
  const Box3 me = tile[threadIdx.x + radius];
  float best = 3.0e38f;
  int bestJ = -1;
  // equal areas: the pair with the smaller (min index, max index) wins on both sides, so that the globally best pair always chooses each other
  for (int o = -radius; o <= radius; o++) {
    const int j = i + o;
    if (o == 0 || j < 0 || j >= m) continue;
    const float d = unionArea(me, tile[threadIdx.x + radius + o]);
    if (d < best) { best = d; bestJ = j; }
    else if (d == best && bestJ >= 0) {
      const int a0 = min(i, j), a1 = max(i, j), b0 = min(i, bestJ), b1 = max(i, bestJ);
      if (a0 < b0 || (a0 == b0 && a1 < b1)) bestJ = j;
    }
  }
  nn[i] = bestJ;
}
__global__ void k_ploc_merge(int m, const int* __restrict__ nn, int* __restrict__ clusterId, Box3* __restrict__ clusterBox, int* __restrict__ nextNode,
                             int2* __restrict__ children, Box3* __restrict__ nodeBoxes, int* __restrict__ count, int* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int j = nn[i];
  const bool mutual = (j >= 0) && (nn[j] == i);
  if (!mutual) { keep[i] = 1; return; }
  if (i > j) { keep[i] = 0; return; }                    // the lower place keeps the merged cluster
  const int id = atomicSub(nextNode, 1);
  const int a = clusterId[i], b = clusterId[j];
  const Box3 ba = clusterBox[i], bb = clusterBox[j];
  Box3 u;
  for (int k = 0; k < 3; k++) { u.mn[k] = fminf(ba.mn[k], bb.mn[k]); u.mx[k] = fmaxf(ba.mx[k], bb.mx[k]); }
  children[id] = make_int2(a, b);
  nodeBoxes[id] = u;
  count[id] = count[a] + count[b];
  clusterId[i] = id; clusterBox[i] = u;
  keep[i] = 1;
}
// order-preserving compaction: per-block counts, one-block scan, scatter by ballot rank
__global__ void __launch_bounds__(PLOC_BLOCK) k_ploc_count(int m, const int* __restrict__ keep, int* __restrict__ blockSums) {
  __shared__ int part[PLOC_BLOCK / 64];
  const int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
  const bool k = (i < m) && keep[i] != 0;
  const unsigned long long bal = __ballot(k);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < PLOC_BLOCK / 64; w++) t += part[w]; blockSums[blockIdx.x] = t; }
}
__global__ void __launch_bounds__(PLOC_BLOCK) k_ploc_scatter(int m, const int* __restrict__ keep, const int* __restrict__ blockOffs, const int* __restrict__ idIn, const Box3* __restrict__ boxIn,
                                                            int* __restrict__ idOut, Box3* __restrict__ boxOut) {
  __shared__ int part[PLOC_BLOCK / 64];
  const int i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
  const bool k = (i < m) && keep[i] != 0;
  const unsigned long long bal = __ballot(k);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = __popcll(bal);
  __syncthreads();
  int before = blockOffs[blockIdx.x];
  for (int w = 0; w < wave; w++) before += part[w];
  if (k) { const int dst = before + __popcll(bal & ((1ull << lane) - 1ull)); idOut[dst] = idIn[i]; boxOut[dst] = boxIn[i]; }
}

// ---- collapse into 4-wide nodes.  A work item = (output node, binary node, first place of its triangles in the output order); children with more
// than leafMax triangles go to the next level.  Works on any binary tree given as children / count / boxes: a leaf of the 4-wide tree lists the
// triangles below its binary node in depth-first order.
struct WorkItem { int outNode, binNode, first; };
__device__ __forceinline__ float boxArea(const Box3& b) {
  const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}
__device__ __forceinline__ void listTriangles(int node, int n, const int2* __restrict__ children, const uint32_t* __restrict__ sortedTri, int first, int* __restrict__ primOut) {
  int stack[24];
  int top = 0, w = first;
  stack[top++] = node;
  while (top > 0) {
    const int nd = stack[--top];
    if (nd >= n - 1) primOut[w++] = int(sortedTri[nd - (n - 1)]);
    else if (top + 2 <= 24) { const int2 ch = children[nd]; stack[top++] = ch.y; stack[top++] = ch.x; }
  }
}
__global__ void k_bvh_collapse(int n, int leafMax, const WorkItem* __restrict__ in, int inCount, const int2* __restrict__ children, const int* __restrict__ count, const Box3* __restrict__ nodeBoxes,
                               const uint32_t* __restrict__ sortedTri, HydraBuildNode* __restrict__ out, int* __restrict__ outCount, WorkItem* __restrict__ next, int* __restrict__ nextCount, int* __restrict__ primOut, int openByProduct) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= inCount) return;
  const WorkItem item = in[w];
  int c[4];
  int nc = 2;
  { const int2 ch = children[item.binNode]; c[0] = ch.x; c[1] = ch.y; }
  while (nc < 4) {            // open the child with the largest box that still has more than leafMax triangles (the host builder's rule, bvh4_builder.cpp; openByProduct: area x count, the rule before)
    int pick = -1;
    float best = -1.0f;
    for (int k = 0; k < nc; k++) {
      const int cnt = count[c[k]];
      if (cnt <= leafMax) continue;
      const float wgt = openByProduct ? boxArea(nodeBoxes[c[k]]) * float(cnt) : boxArea(nodeBoxes[c[k]]);
      if (wgt > best) { best = wgt; pick = k; }
    }
    if (pick < 0) break;
    const int2 ch = children[c[pick]];
    c[pick] = ch.x; c[nc++] = ch.y;
  }
  const int base = atomicAdd(outCount, nc);
  HydraBuildNode me = out[item.outNode];
  for (int k = 0; k < 4; k++) me.child[k] = k < nc ? base + k : -1;
  me.first = 0; me.count = 0;
  out[item.outNode] = me;
  int first = item.first;
  for (int k = 0; k < nc; k++) {
    HydraBuildNode ch;
    const Box3 b = nodeBoxes[c[k]];
    for (int a = 0; a < 3; a++) { ch.boxMin[a] = b.mn[a]; ch.boxMax[a] = b.mx[a]; }
    ch.child[0] = ch.child[1] = ch.child[2] = ch.child[3] = -1;
    const int cnt = count[c[k]];
    if (cnt <= leafMax) { ch.first = first; ch.count = cnt; listTriangles(c[k], n, children, sortedTri, first, primOut); }
    else { ch.first = 0; ch.count = 0; const int q = atomicAdd(nextCount, 1); next[q] = WorkItem{base + k, c[k], first}; }
    out[base + k] = ch;
    first += cnt;
  }
}
__global__ void k_bvh_root(int n, int leafMax, const int2* __restrict__ children, const Box3* __restrict__ nodeBoxes, const uint32_t* __restrict__ sortedTri, HydraBuildNode* __restrict__ out, int* __restrict__ outCount,
                           WorkItem* __restrict__ next, int* __restrict__ nextCount, int* __restrict__ primOut) {
  const int root = (n == 1) ? 0 /* the only leaf: n - 1 + 0 */ : 0;
  HydraBuildNode r;
  const Box3 b = nodeBoxes[root];
  for (int a = 0; a < 3; a++) { r.boxMin[a] = b.mn[a]; r.boxMax[a] = b.mx[a]; }
  r.child[0] = r.child[1] = r.child[2] = r.child[3] = -1;
  if (n <= leafMax) { r.first = 0; r.count = n; listTriangles(root, n, children, sortedTri, 0, primOut); }
  else { r.first = 0; r.count = 0; next[0] = WorkItem{0, 0, 0}; *nextCount = 1; }
  out[0] = r;
  *outCount = 1;
}

struct DevMem {
  std::vector<void*> ptrs;
  ~DevMem() { for (void* p : ptrs) (void)hipFree(p); }
  template <typename T> T* get(size_t count) {
    void* p = nullptr;
    if (hipMalloc(&p, (count > 0 ? count : 1) * sizeof(T)) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return static_cast<T*>(p);
  }
};

}  // namespace

extern "C" {

const char* hydra_hip_bvh_last_error(void) { return g_bvhError.c_str(); }

// nodes_out: capacity 2 * triangles (a 4-wide tree over T leaves of >= 1 triangle has fewer than 2T nodes); prim_order_out: capacity triangles
int hydra_hip_bvh_build_mesh(int device, const float* vert4f, int num_vert, const int32_t* indices, int num_indices, int leaf_max,
                             HydraBuildNode* nodes_out, int32_t* node_count_out, int32_t* prim_order_out, int32_t* prim_count_out, float* build_ms_out) {
  return hydra_hip_bvh_build_mesh_ex(device, vert4f, num_vert, indices, num_indices, leaf_max, HYDRA_BVH_PLOC, 128, nodes_out, node_count_out, prim_order_out, prim_count_out, build_ms_out);
}
int hydra_hip_bvh_build_mesh_ex(int device, const float* vert4f, int num_vert, const int32_t* indices, int num_indices, int leaf_max, int method, int radius,
                                HydraBuildNode* nodes_out, int32_t* node_count_out, int32_t* prim_order_out, int32_t* prim_count_out, float* build_ms_out) {
  if (!vert4f || !indices || !nodes_out || !node_count_out || !prim_order_out || !prim_count_out || num_vert <= 0 || num_indices < 3 || leaf_max < 1 || leaf_max > 16 ||
      (method != HYDRA_BVH_LBVH && method != HYDRA_BVH_PLOC) || (method == HYDRA_BVH_PLOC && (radius < 1 || radius > PLOC_MAX_RADIUS))) {
    g_bvhError = "bvh_build_mesh: bad argument";
    return HYDRA_HIP_EINVAL;
  }
  const int triNum = num_indices / 3;
  for (int i = 0; i < triNum * 3; i++) if (indices[i] < 0 || indices[i] >= num_vert) { g_bvhError = "bvh_build_mesh: vertex index out of range"; return HYDRA_HIP_EINVAL; }
  BCHECK(hipSetDevice(device));
  DevMem dm;
  float4* dVert = dm.get<float4>(size_t(num_vert)); int* dIdx = dm.get<int>(size_t(triNum) * 3);
  Box3* dTriBox = dm.get<Box3>(size_t(triNum)); int* dValid = dm.get<int>(size_t(triNum)); float* dBounds = dm.get<float>(6);
  uint32_t* dKeys[2] = {dm.get<uint32_t>(size_t(triNum)), dm.get<uint32_t>(size_t(triNum))};
  uint32_t* dVals[2] = {dm.get<uint32_t>(size_t(triNum)), dm.get<uint32_t>(size_t(triNum))};
  const int numTiles = (triNum + RS_BLOCK - 1) / RS_BLOCK;
  int* dHist = dm.get<int>(size_t(16) * numTiles);
  int2* dChildren = dm.get<int2>(size_t(triNum)); int* dParent = dm.get<int>(size_t(triNum) * 2); int* dCount = dm.get<int>(size_t(triNum) * 2);
  int* dPrimOut = dm.get<int>(size_t(triNum));
  int* dClusterId[2] = {dm.get<int>(size_t(triNum)), dm.get<int>(size_t(triNum))}; Box3* dClusterBox[2] = {dm.get<Box3>(size_t(triNum)), dm.get<Box3>(size_t(triNum))};
  int* dNN = dm.get<int>(size_t(triNum)); int* dKeep = dm.get<int>(size_t(triNum)); int* dBlockSums = dm.get<int>(size_t((triNum + PLOC_BLOCK - 1) / PLOC_BLOCK) + 1);
  Box3* dNodeBox = dm.get<Box3>(size_t(triNum) * 2); int* dArrived = dm.get<int>(size_t(triNum));
  HydraBuildNode* dOut = dm.get<HydraBuildNode>(size_t(triNum) * 2);
  WorkItem* dWork[2] = {dm.get<WorkItem>(size_t(triNum)), dm.get<WorkItem>(size_t(triNum))};
  int* dCounts = dm.get<int>(4);   // [0] output nodes, [1] / [2] work items of the two frontiers
  if (!dVert || !dIdx || !dTriBox || !dValid || !dBounds || !dKeys[0] || !dKeys[1] || !dVals[0] || !dVals[1] || !dHist || !dChildren || !dParent || !dCount || !dPrimOut || !dClusterId[0] || !dClusterId[1] || !dClusterBox[0] || !dClusterBox[1] || !dNN || !dKeep || !dBlockSums || !dNodeBox || !dArrived || !dOut ||
      !dWork[0] || !dWork[1] || !dCounts) { g_bvhError = "bvh_build_mesh: hipMalloc failed"; return HYDRA_HIP_ENOMEM; }
  BCHECK(hipMemcpy(dVert, vert4f, size_t(num_vert) * 16, hipMemcpyHostToDevice));
  BCHECK(hipMemcpy(dIdx, indices, size_t(triNum) * 12, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  BCHECK(hipEventCreate(&e0)); BCHECK(hipEventCreate(&e1));
  BCHECK(hipEventRecord(e0, nullptr));
  const float inf = 3.0e38f;
  const float initBounds[6] = {inf, inf, inf, -inf, -inf, -inf};
  BCHECK(hipMemcpy(dBounds, initBounds, 24, hipMemcpyHostToDevice));
  const dim3 blk(256), grdT((triNum + 255) / 256);
  hipLaunchKernelGGL(k_bvh_prims, grdT, blk, 0, nullptr, triNum, dVert, dIdx, dTriBox, dValid, dBounds);
  hipLaunchKernelGGL(k_bvh_morton, grdT, blk, 0, nullptr, triNum, dTriBox, dValid, dBounds, dKeys[0], dVals[0]);
  int cur = 0;
  for (int shift = 0; shift < 32; shift += 4) {
    hipLaunchKernelGGL(k_rs_hist, dim3(numTiles), dim3(RS_BLOCK), 0, nullptr, triNum, dKeys[cur], shift, dHist, numTiles);
    hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(256), 0, nullptr, 16 * numTiles, dHist);
    hipLaunchKernelGGL(k_rs_scatter, dim3(numTiles), dim3(RS_BLOCK), 0, nullptr, triNum, dKeys[cur], dVals[cur], shift, dHist, numTiles, dKeys[cur ^ 1], dVals[cur ^ 1]);
    cur ^= 1;
  }
  BCHECK(hipGetLastError());
  // the number of valid triangles = keys below 0xFFFFFFFF; the validity flags are summed on the host (one small copy)
  std::vector<int> valid((size_t(triNum)), 0);
  BCHECK(hipMemcpy(valid.data(), dValid, size_t(triNum) * 4, hipMemcpyDeviceToHost));
  int n = 0;
  for (int v : valid) n += v;
  if (n == 0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); g_bvhError = "bvh_build_mesh: mesh without valid triangles"; return HYDRA_HIP_EINVAL; }
  BCHECK(hipMemsetAsync(dArrived, 0, size_t(triNum) * 4, nullptr));
  const dim3 grdN((n + 255) / 256);
  if (method == HYDRA_BVH_LBVH) {
    if (n > 1) hipLaunchKernelGGL(k_bvh_hierarchy, grdN, blk, 0, nullptr, n, dKeys[cur], dChildren, dParent, dCount);
    else { const int minus1 = -1; BCHECK(hipMemcpy(dParent, &minus1, 4, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_bvh_refit, grdN, blk, 0, nullptr, n, dVals[cur], dTriBox, dChildren, dParent, dNodeBox, dArrived, dCount);
  } else {
    hipLaunchKernelGGL(k_ploc_init, grdN, blk, 0, nullptr, n, dVals[cur], dTriBox, dClusterId[0], dClusterBox[0], dNodeBox, dCount);
    const int nextNode0 = n - 2;
    BCHECK(hipMemcpy(dCounts + 3, &nextNode0, 4, hipMemcpyHostToDevice));
    int m = n, pc = 0;
    for (int round = 0; m > 1; round++) {
      if (round > 4096) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); g_bvhError = "bvh_build_mesh: PLOC does not converge"; return HYDRA_HIP_EDEVICE; }
      const int blocks = (m + PLOC_BLOCK - 1) / PLOC_BLOCK;
      hipLaunchKernelGGL(k_ploc_nn, dim3(blocks), dim3(PLOC_BLOCK), 0, nullptr, m, radius, dClusterBox[pc], dNN);
      hipLaunchKernelGGL(k_ploc_merge, dim3(blocks), dim3(PLOC_BLOCK), 0, nullptr, m, dNN, dClusterId[pc], dClusterBox[pc], dCounts + 3, dChildren, dNodeBox, dCount, dKeep);
      hipLaunchKernelGGL(k_ploc_count, dim3(blocks), dim3(PLOC_BLOCK), 0, nullptr, m, dKeep, dBlockSums);
      hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(256), 0, nullptr, blocks + 1, dBlockSums);      // exclusive; entry [blocks] (zero before) becomes the total
      hipLaunchKernelGGL(k_ploc_scatter, dim3(blocks), dim3(PLOC_BLOCK), 0, nullptr, m, dKeep, dBlockSums, dClusterId[pc], dClusterBox[pc], dClusterId[pc ^ 1], dClusterBox[pc ^ 1]);
      int total = 0;
      BCHECK(hipMemcpy(&total, dBlockSums + blocks, 4, hipMemcpyDeviceToHost));
      if (total <= 0 || total >= m) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); g_bvhError = "bvh_build_mesh: PLOC round without a merge"; return HYDRA_HIP_EDEVICE; }
      m = total;
      pc ^= 1;
    }
  }
  BCHECK(hipMemsetAsync(dCounts, 0, 16, nullptr));
  hipLaunchKernelGGL(k_bvh_root, dim3(1), dim3(1), 0, nullptr, n, leaf_max, dChildren, dNodeBox, dVals[cur], dOut, dCounts, dWork[0], dCounts + 1, dPrimOut);
  static const int openByProduct = [] { const char* e = getenv("HYDRA_GPU_BVH_OPEN"); return (e != nullptr && e[0] == 'p') ? 1 : 0; }();   // A/B switch: area x count, the rule before
  int frontier = 0;
  for (int level = 0; level < 128; level++) {
    int counts[4];
    BCHECK(hipMemcpy(counts, dCounts, 16, hipMemcpyDeviceToHost));
    const int inCount = counts[1 + frontier];
    if (inCount == 0) break;
    BCHECK(hipMemsetAsync(dCounts + 1 + (frontier ^ 1), 0, 4, nullptr));
    hipLaunchKernelGGL(k_bvh_collapse, dim3((inCount + 255) / 256), blk, 0, nullptr, n, leaf_max, dWork[frontier], inCount, dChildren, dCount, dNodeBox, dVals[cur], dOut, dCounts,
                       dWork[frontier ^ 1], dCounts + 1 + (frontier ^ 1), dPrimOut, openByProduct);
    BCHECK(hipMemsetAsync(dCounts + 1 + frontier, 0, 4, nullptr));
    frontier ^= 1;
  }
  BCHECK(hipEventRecord(e1, nullptr));
  BCHECK(hipEventSynchronize(e1));
  float ms = 0.0f;
  BCHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  int outCount = 0;
  BCHECK(hipMemcpy(&outCount, dCounts, 4, hipMemcpyDeviceToHost));
  if (outCount <= 0 || outCount > 2 * triNum) { g_bvhError = "bvh_build_mesh: internal error, node count out of range"; return HYDRA_HIP_EDEVICE; }
  BCHECK(hipMemcpy(nodes_out, dOut, size_t(outCount) * sizeof(HydraBuildNode), hipMemcpyDeviceToHost));
  BCHECK(hipMemcpy(prim_order_out, dPrimOut, size_t(n) * 4, hipMemcpyDeviceToHost));
  *node_count_out = outCount; *prim_count_out = n;
  if (build_ms_out) *build_ms_out = ms;
  return HYDRA_HIP_OK;
}

}  // extern "C"
