// hydra_bake.hip -- the two multi-scattering energy tables of the globals header, baked on the device.
//
// What they are: EngineGlobals::m_essGgx2017Table (64 x 64 u16: rows = roughness, columns = dot(N, V)) and m_essTranspTable (64^3 u16:
// slices = index of refraction mapped from [0.4166, 2.4], then the same rows and columns) (hydra_drv/cfetch.h:77-79), read by GGX
// reflection and GGX glass nodes that carry PLAIN_MATERIAL_ENERGY_FIX (cmaterial.h:152-196, 858-863, 1400-1405).  A cell holds Ess, the
// mean of the single-scattering weight G2 / G1 of a visible-normal sample over the cell's range of roughness, dot(N, V) (and ior) and
// over the sample's two random numbers, times 65535, truncated.
// The reference ships them as data baked offline by a Monte-Carlo program (bakeBrdfEnergy/bakeBrdf.cpp: Ggx2017 :186-214, TranspGgx
// :259-316, BakeBrdfEnergyTable :321-437 -- ~1 M Sobol samples per GGX cell, ~16 k per transparency cell -- written out by SaveTableToFile
// :440-511) and its layers copy them into the header when they are constructed (IHWLayer.h:101, cfetch.h:83-93).  That data is not part
// of this repository; HipHWLayer bakes its own at construction: the same integrand over the same cells, integrated with one fixed
// low-discrepancy point set per cell (Halton 2-3-5-7-11) instead of a random stream, so that the table is a function of nothing but this
// file.  tests/test_energy_tables.py bounds the difference to the reference's literals (read as text where the reference tree is present).
// One block per cell, a lane per slice of the point set, a block-wide sum; 64^3 cells x 2 048 points = 0.5 G integrand calls, milliseconds.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <mutex>
#include "../../include/hydra_hip.h"
#include "hk_common.h"
#include "hk_trace.h"
#include "hk_shading.h"

namespace {

thread_local std::string g_bakeError;
#define BCHECK(call)                                                                                        \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) { g_bakeError = std::string(#call) + ": " + hipGetErrorString(e_); return HYDRA_HIP_EDEVICE; } \
  } while (0)

__device__ __forceinline__ float halton(uint32_t i, uint32_t base) {   // radical inverse of i in `base`
  float f = 1.0f, r = 0.0f;
  const float inv = 1.0f / float(base);
  while (i > 0) { f *= inv; r += f * float(i % base); i /= base; }
  return r;
}
// the view direction the baker builds from a cosine (bakeBrdf.cpp:361-364): in the x-z plane, coming down onto n = +z
__device__ __forceinline__ f3 bake_view(float cosTheta) {
  const float theta = clampf(1.57079632679489661923f - acosf(cosTheta), 0.0f, 1.57079632679489661923f);
  return normalize(mk3(cosf(theta), 0.0f, sinf(theta))) * (-1.0f);
}
// Ggx2017, bakeBrdf.cpp:186-214
__device__ float bake_ggx(const f3 rayDir, const f3 n, const float roughSqr, const float u1, const float u2) {
  if (roughSqr < 1e-6f) return 1.0f;
  const float dotNV = dot(n, rayDir * (-1.0f));
  f3 nx, ny;
  CoordinateSystem(n, nx, ny);
  const f3 wo = normalize(mk3(-dot(rayDir, nx), -dot(rayDir, ny), -dot(rayDir, n)));
  const f3 wh = GgxVndf(wo, roughSqr, u1, u2);
  const f3 wi = (wh * (2.0f * dot(wo, wh))) - wo;
  const f3 l = normalize(((nx * wi.x) + (ny * wi.y)) + (n * wi.z));
  const float dotNL = dot(n, l);
  if (dotNL < 1e-6f) return 0.0f;
  const float G1 = SmithGGXMasking(dotNV, roughSqr), G2 = SmithGGXMaskingShadowing(dotNL, dotNV, roughSqr);
  return G2 / fmaxf(G1, 1e-6f);
}
// TranspGgx, bakeBrdf.cpp:259-316
__device__ float bake_transp(const f3 rayDir, const f3 n, const float roughSqr, const bool inside, const float ior, const float u1, const float u2) {
  const f3 normal2 = inside ? n * (-1.0f) : n;
  RefractResult refr = myRefractGgx(rayDir, normal2, ior, 1.0f);
  float Pss = 1.0f;
  if (roughSqr > 0.001f) {
    float eta = 1.0f / ior;
    const float cosTheta = dot(normal2, rayDir) * (-1.0f);
    if (cosTheta < 0.0f) eta = 1.0f / eta;
    f3 nx, ny;
    CoordinateSystem(n, nx, ny);
    const f3 wo = mk3(-dot(rayDir, nx), -dot(rayDir, ny), -dot(rayDir, n));
    const f3 wh = GgxVndf(wo, roughSqr, u1, u2);
    const float dotWoWh = dot(wo, wh);
    f3 newDir;
    const float radicand = 1.0f + eta * eta * (dotWoWh * dotWoWh - 1.0f);
    if (radicand > 0.0f) { newDir = (wh * (eta * dotWoWh - sqrtf(radicand))) - (wo * eta); refr.success = true; refr.eta = eta; }
    else { newDir = (wh * (2.0f * dotWoWh)) - wo; refr.success = false; refr.eta = 1.0f; }
    refr.ray_dir = normalize(((nx * newDir.x) + (ny * newDir.y)) + (n * newDir.z));
    const float dotNV = fabsf(dot(n, rayDir)), dotNL = fabsf(dot(n, refr.ray_dir));
    const float G1 = SmithGGXMasking(dotNV, roughSqr), G2 = SmithGGXMaskingShadowing(dotNL, dotNV, roughSqr);
    Pss = G2 / fmaxf(G1, 1e-6f);
  }
  const float cosThetaOut = dot(refr.ray_dir, n);
  if (refr.success && cosThetaOut >= -1e-6f) return 0.0f;      // a refracted ray must leave on the far side
  if (!refr.success && cosThetaOut < 1e-6f) return 0.0f;       // a reflected one on the near side
  return Pss;
}

// cell = (z * 64 + y) * 64 + x: x = dot(N, V) column, y = roughness row, z = ior slice (the 2-D table: z = 0, TRANSP false).
// A sample's position inside the cell comes from the first two (three) Halton dimensions, the visible-normal sample from the next two.
template <bool TRANSP>
__global__ void __launch_bounds__(256) k_bake_ess(int points, uint16_t* __restrict__ out) {
  const int cell = int(blockIdx.x);
  const int x = cell & 63, y = (cell >> 6) & 63, z = cell >> 12;
  const f3 n = mk3(0.0f, 0.0f, 1.0f);
  float sum = 0.0f;
  for (int i = int(threadIdx.x); i < points; i += 256) {
    const uint32_t k = uint32_t(i) + 1u;
    const float roughness = (float(y) + halton(k, 2)) * (1.0f / 64.0f);
    const float cosTheta = (float(x) + halton(k, 3)) * (1.0f / 64.0f);
    const f3 v = bake_view(cosTheta);
    const float roughSqr = roughness * roughness;
    float res;
    if (TRANSP) {
      const float ior = ((float(z) + halton(k, 5)) * (1.0f / 64.0f)) * (2.4f - 0.4166f) + 0.4166f;   // [0.4166, 2.4], bakeBrdf.cpp:374-375
      const bool inside = ior < 1.0f;
      res = bake_transp(v, n, roughSqr, inside, inside ? 1.0f / ior : ior, halton(k, 7), halton(k, 11));
    } else
      res = bake_ggx(v, n, roughSqr, halton(k, 5), halton(k, 7));
    if (res == res) sum += res;                                   // the reference skips a NaN sample too (:410-424)
  }
  __shared__ float part[256];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (int(threadIdx.x) < s) part[threadIdx.x] += part[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const float mean = fminf(fmaxf(part[0] / float(points), 0.0f), 1.0f);
    out[cell] = uint16_t(mean * 65535.0f);                        // SaveTableToFile: (ushort)(value * USHRT_MAX), :490
  }
}

std::mutex g_bakeLock;
std::vector<uint16_t> g_tables;   // both tables of this process, baked once: [4096 | 262144]

}  // namespace

extern "C" {

const char* hydra_hip_bake_last_error(void) { return g_bakeError.c_str(); }

int hydra_hip_bake_energy_tables(int device, uint16_t* ggx4096, uint16_t* transp262144, float* device_ms_out) {
  if (!ggx4096 || !transp262144) { g_bakeError = "bake_energy_tables: null argument"; return HYDRA_HIP_EINVAL; }
  std::lock_guard<std::mutex> lock(g_bakeLock);
  if (device_ms_out) *device_ms_out = 0.0f;
  if (g_tables.empty()) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { g_bakeError = "bake_energy_tables: no HIP device"; return HYDRA_HIP_ENODEV; }
    BCHECK(hipSetDevice(device));
    const size_t n2 = 64 * 64, n3 = 64 * 64 * 64;
    uint16_t* d = nullptr;
    BCHECK(hipMalloc(&d, (n2 + n3) * sizeof(uint16_t)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(k_bake_ess<false>, dim3(unsigned(n2)), dim3(256), 0, nullptr, 16384, d);
    hipLaunchKernelGGL(k_bake_ess<true>, dim3(unsigned(n3)), dim3(256), 0, nullptr, 2048, d + n2);
    (void)hipEventRecord(e1, nullptr);
    std::vector<uint16_t> host(n2 + n3);
    const hipError_t e = hipMemcpy(host.data(), d, host.size() * sizeof(uint16_t), hipMemcpyDeviceToHost);
    float ms = 0.0f;
    if (e == hipSuccess && e0 && e1 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && device_ms_out) *device_ms_out = ms;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d);
    if (e != hipSuccess) { g_bakeError = std::string("bake_energy_tables: ") + hipGetErrorString(e); return HYDRA_HIP_EDEVICE; }
    g_tables.swap(host);
  }
  memcpy(ggx4096, g_tables.data(), 4096 * sizeof(uint16_t));
  memcpy(transp262144, g_tables.data() + 4096, 262144 * sizeof(uint16_t));
  return HYDRA_HIP_OK;
}

}  // extern "C"
