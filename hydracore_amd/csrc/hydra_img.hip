// hydra_img.hip -- IHWLayer::NormalMapFromDisplacement on the device: a height map (RGBA8) becomes the normal map the shading's BumpMapping reads.
//
// Behaviour contract: CPUSharedData::NormalMapFromDisplacement + BilateralFilter (hydra_drv/CPUBilateralFilter2D.cpp:15-246), the host form of
// what GPUOCLLayer::NormalMapFromDisplacement (GPUOCLData.cpp:549-640) runs as image kernels:
//   height   = 255 - max(r, g, b) per texel
//   normal   = mean of eight difference vectors to the neighbours (wrap-around with the reference's `<= 0` / `>= size` index rules), the
//              differences scaled by bumpAmt^2 and signed by invHeight, z = 2000 / min(w, h); x mirrored, normalised, z floored at 0.65
//              and normalised again; w = the texel's own height back in 0..1
//   filter   (smoothLvl >= 1) a bilateral filter over an 11 x 11 window clipped at the image border: range weight on the float4 distance,
//              spatial sigma 1/50, then a lerp with the unfiltered texel by how many taps passed the weight threshold
//   pack     x, y to 0.5 + 0.5 v, z and w as they are, times 255, clamped, truncated to bytes
// One entry point, no layer handle needed (like the BVH builder): the front end calls it while it packs materials.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/hydra_hip.h"

namespace {

thread_local std::string g_imgError;
#define ICHECK(call)                                                                                        \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) { g_imgError = std::string(#call) + ": " + hipGetErrorString(e_); return HYDRA_HIP_EDEVICE; } \
  } while (0)

__global__ void k_nm_height(int n, const uchar4* __restrict__ in, float* __restrict__ height) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uchar4 c = in[i];
  height[i] = 255.0f - fmaxf(float(c.x), fmaxf(float(c.y), float(c.z)));
}

__global__ void k_nm_normals(int w, int h, const float* __restrict__ hd, float bumpAmt, int invHeight, float4* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const float kScale = 2000.0f / fminf(float(w), float(h));
  const int offsetY = y * w;
  int offsetYPlusOne = (y + 1) * w, offsetYMinusOne = (y - 1) * w;
  if (y + 1 >= h) offsetYPlusOne = 0;
  if (y - 1 <= 0) offsetYMinusOne = (h - 1) * w;          // sic: row 1 also looks at the last row
  int offsetXPlusOne = x + 1, offsetXMinusOne = x - 1;
  if (x + 1 >= w) offsetXPlusOne = 0;
  if (x - 1 <= 0) offsetXMinusOne = w - 1;
  const float c = hd[offsetY + x];
  float diff[8];
  diff[0] = c - hd[offsetYMinusOne + offsetXMinusOne];
  diff[1] = c - hd[offsetYMinusOne + x];
  diff[2] = c - hd[offsetYMinusOne + offsetXPlusOne];
  diff[3] = c - hd[offsetY + offsetXMinusOne];
  diff[4] = c - hd[offsetY + offsetXPlusOne];
  diff[5] = c - hd[offsetYPlusOne + offsetXMinusOne];
  diff[6] = c - hd[offsetYPlusOne + x];
  diff[7] = c - hd[offsetYPlusOne + offsetXPlusOne];
  if (!invHeight)
    for (int i = 0; i < 8; i++) diff[i] *= -1.0f;
  for (int i = 0; i < 8; i++) diff[i] *= (bumpAmt * bumpAmt);
  const float scale = kScale;
  const float vx[8] = {-diff[0], 0.f, diff[2], -diff[3], diff[4], -diff[5], 0.f, diff[7]};
  const float vy[8] = {-diff[0], -diff[1], -diff[2], 0.f, 0.f, diff[5], diff[6], diff[7]};
  float rx = 0.0f, ry = 0.0f, rz = 0.0f;
  for (int i = 0; i < 8; i++) { rx += vx[i]; ry += vy[i]; rz += scale; }
  rx = rx * (1.0f / 8.0f); ry = ry * (1.0f / 8.0f); rz = rz * (1.0f / 8.0f);
  rx *= -1.0f;
  // normalize / lerp / dot3f come from HydraAPI's LiteMath, which is not part of the reference tree: taken as u / length(u), u + t (v - u), x^2 + y^2 + z^2
  float len = sqrtf(rx * rx + ry * ry + rz * rz);
  rx /= len; ry /= len; rz /= len;
  if (rz < 0.65f) {
    rz = 0.65f;
    len = sqrtf(rx * rx + ry * ry + rz * rz);
    rx /= len; ry /= len; rz /= len;
  }
  out[offsetY + x] = make_float4(rx, ry, rz, (255.0f - c) / 255.0f);
}

// 16 x 16 output texels per block, the tile and its halo of `radius` texels staged in LDS (texels outside the image are marked and skipped:
// the reference clips its window at the border)
#define NM_TILE 16
#define NM_MAXR 7
__global__ void __launch_bounds__(NM_TILE * NM_TILE) k_nm_bilateral(int w, int h, const float4* __restrict__ in, int radius, float smoothLvl, float4* __restrict__ out) {
  __shared__ float4 tile[(NM_TILE + 2 * NM_MAXR) * (NM_TILE + 2 * NM_MAXR)];
  const int span = NM_TILE + 2 * radius;
  const int x0 = blockIdx.x * NM_TILE - radius, y0 = blockIdx.y * NM_TILE - radius;
  for (int i = threadIdx.y * NM_TILE + threadIdx.x; i < span * span; i += NM_TILE * NM_TILE) {
    const int tx = x0 + i % span, ty = y0 + i / span;
    tile[i] = (tx >= 0 && tx < w && ty >= 0 && ty < h) ? in[ty * w + tx] : make_float4(0, 0, 0, 0);
  }
  __syncthreads();
  const int x = blockIdx.x * NM_TILE + threadIdx.x, y = blockIdx.y * NM_TILE + threadIdx.y;
  if (x >= w || y >= h) return;
  const float g_NoiseLevel = 1.0f / (smoothLvl * smoothLvl), g_GaussianSigma = 1.0f / 50.0f, g_WeightThreshold = 0.03f, g_LerpCoefficeint = 0.80f, g_CounterThreshold = 0.05f;
  const float windowArea = (2.0f * float(radius) + 1.0f) * (2.0f * float(radius) + 1.0f);
  const int minX = max(x - radius, 0), maxX = min(x + radius, w - 1), minY = max(y - radius, 0), maxY = min(y + radius, h - 1);
  const float4 c0 = tile[(threadIdx.y + radius) * span + threadIdx.x + radius];
  int counterPass = 0;
  float fSum = 0.0f;
  float4 result = make_float4(0, 0, 0, 0);
  for (int y1 = minY; y1 <= maxY; y1++)
    for (int x1 = minX; x1 <= maxX; x1++) {
      const float4 c1 = tile[(y1 - y0) * span + (x1 - x0)];
      const float dx = c1.x - c0.x, dy = c1.y - c0.y, dz = c1.z - c0.z;
      const int i = x1 - x, j = y1 - y;
      const float w1 = dx * dx + dy * dy + dz * dz;                                  // dot3f
      const float w2 = expf(-(w1 * g_NoiseLevel + float(i * i + j * j) * g_GaussianSigma));
      if (w2 > g_WeightThreshold) counterPass++;
      fSum += w2;
      result.x += c1.x * w2; result.y += c1.y * w2; result.z += c1.z * w2; result.w += c1.w * w2;
    }
  const float inv = 1.0f / fSum;
  result.x *= inv; result.y *= inv; result.z *= inv; result.w *= inv;
  const float lerpQ = (float(counterPass) > (g_CounterThreshold * windowArea)) ? 1.0f - g_LerpCoefficeint : g_LerpCoefficeint;
  // lerp(u, v, t) = u + t * (v - u), hydra_drv/LiteMath-style helper
  result.x = result.x + lerpQ * (c0.x - result.x); result.y = result.y + lerpQ * (c0.y - result.y);
  result.z = result.z + lerpQ * (c0.z - result.z); result.w = result.w + lerpQ * (c0.w - result.w);
  out[y * w + x] = result;
}

__global__ void k_nm_pack(int n, const float4* __restrict__ in, uchar4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 r = in[i];
  float cx = 0.5f * r.x + 0.5f, cy = 0.5f * r.y + 0.5f, cz = r.z, cw = r.w;
  cx = fminf(fmaxf(cx * 255.0f, 0.0f), 255.0f); cy = fminf(fmaxf(cy * 255.0f, 0.0f), 255.0f);
  cz = fminf(fmaxf(cz * 255.0f, 0.0f), 255.0f); cw = fminf(fmaxf(cw * 255.0f, 0.0f), 255.0f);
  out[i] = make_uchar4((unsigned char)cx, (unsigned char)cy, (unsigned char)cz, (unsigned char)cw);
}

}  // namespace

extern "C" {

const char* hydra_hip_image_last_error(void) { return g_imgError.c_str(); }

int hydra_hip_normal_map_from_displacement(int device, int w, int h, const uint8_t* rgba_in, float bump_amt, int inv_height, float smooth_lvl, uint8_t* rgba_out, float* device_ms_out) {
  if (w <= 0 || h <= 0 || !rgba_in || !rgba_out || (long long)w * h > (1ll << 28)) { g_imgError = "normal_map_from_displacement: bad size or null image"; return HYDRA_HIP_EINVAL; }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) { g_imgError = "normal_map_from_displacement: no such HIP device"; return HYDRA_HIP_EDEVICE; }
  ICHECK(hipSetDevice(device));
  const int n = w * h;
  uchar4* dIn = nullptr; uchar4* dOut = nullptr; float* dH = nullptr; float4* dN = nullptr; float4* dF = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = HYDRA_HIP_OK;
  auto fin = [&]() { (void)hipFree(dIn); (void)hipFree(dOut); (void)hipFree(dH); (void)hipFree(dN); (void)hipFree(dF); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); };
#define ICHECK_F(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_imgError = std::string(#call) + ": " + hipGetErrorString(e_); fin(); return HYDRA_HIP_EDEVICE; } } while (0)
  ICHECK_F(hipMalloc(&dIn, size_t(n) * 4)); ICHECK_F(hipMalloc(&dOut, size_t(n) * 4)); ICHECK_F(hipMalloc(&dH, size_t(n) * 4));
  ICHECK_F(hipMalloc(&dN, size_t(n) * 16));
  ICHECK_F(hipEventCreate(&e0)); ICHECK_F(hipEventCreate(&e1));
  ICHECK_F(hipMemcpy(dIn, rgba_in, size_t(n) * 4, hipMemcpyHostToDevice));
  ICHECK_F(hipEventRecord(e0, nullptr));
  hipLaunchKernelGGL(k_nm_height, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, dIn, dH);
  hipLaunchKernelGGL(k_nm_normals, dim3((w + 15) / 16, (h + 15) / 16), dim3(16, 16), 0, nullptr, w, h, dH, bump_amt, inv_height ? 1 : 0, dN);
  const float4* packed = dN;
  if (smooth_lvl >= 1.0f) {                       // CPUBilateralFilter2D.cpp:201-210: radius 5, the level capped at 10 and scaled by 0.1
    const int radius = 5;
    float lvl = smooth_lvl;
    if (lvl > 10.0f) lvl = 10.0f;
    ICHECK_F(hipMalloc(&dF, size_t(n) * 16));
    hipLaunchKernelGGL(k_nm_bilateral, dim3((w + NM_TILE - 1) / NM_TILE, (h + NM_TILE - 1) / NM_TILE), dim3(NM_TILE, NM_TILE), 0, nullptr, w, h, dN, radius, lvl * 0.1f, dF);
    packed = dF;
  }
  hipLaunchKernelGGL(k_nm_pack, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, packed, dOut);
  ICHECK_F(hipEventRecord(e1, nullptr));
  ICHECK_F(hipGetLastError());
  ICHECK_F(hipMemcpy(rgba_out, dOut, size_t(n) * 4, hipMemcpyDeviceToHost));
  if (device_ms_out) { float ms = 0.0f; ICHECK_F(hipEventElapsedTime(&ms, e0, e1)); *device_ms_out = ms; }
  fin();
  return rc;
}

}  // extern "C"
