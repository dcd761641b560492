// math_util.h -- small host-side vector/matrix helpers for the scene front end.
//
// Matrices are stored as four consecutive float4 COLUMNS, which is what the reference kernels
// consume (make_float4x4 hydra_drv/cglobals.h:792-800, mul4x3/mul3x3 :288-304).  HydraAPI scene
// files store 4x4 matrices row-major with the translation in the 4th column.
#pragma once
#include <cmath>
#include <cstring>

namespace hydra_host {

struct float3 {
  float x, y, z;
  float3() : x(0), y(0), z(0) {}
  float3(float a, float b, float c) : x(a), y(b), z(c) {}
};
inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, float3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline float  dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float  length(float3 a) { return sqrtf(dot(a, a)); }
inline float3 normalize(float3 a) { float l = length(a); return a * (1.0f / l); }
inline float3 vmin(float3 a, float3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
inline float3 vmax(float3 a, float3 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }

struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uchar4 { unsigned char x, y, z, w; };
struct ushort2 { unsigned short x, y; };

struct float4x4 {
  float c[4][4];  // c[col][row]
  float4x4() { identity(); }
  void identity() {
    memset(c, 0, sizeof(c));
    c[0][0] = c[1][1] = c[2][2] = c[3][3] = 1.0f;
  }
  float& at(int row, int col) { return c[col][row]; }
  float  at(int row, int col) const { return c[col][row]; }
  // 16 floats, row-major text order as in the scene XML
  static float4x4 from_row_major(const float* m) {
    float4x4 r;
    for (int row = 0; row < 4; row++)
      for (int col = 0; col < 4; col++) r.c[col][row] = m[row * 4 + col];
    return r;
  }
  const float* data() const { return &c[0][0]; }
};

inline float4x4 mul(const float4x4& a, const float4x4& b) {
  float4x4 r;
  for (int col = 0; col < 4; col++)
    for (int row = 0; row < 4; row++) {
      float s = 0.0f;
      for (int k = 0; k < 4; k++) s += a.at(row, k) * b.at(k, col);
      r.at(row, col) = s;
    }
  return r;
}
inline float3 mul_point(const float4x4& m, float3 v) {
  return {v.x * m.c[0][0] + v.y * m.c[1][0] + v.z * m.c[2][0] + m.c[3][0],
          v.x * m.c[0][1] + v.y * m.c[1][1] + v.z * m.c[2][1] + m.c[3][1],
          v.x * m.c[0][2] + v.y * m.c[1][2] + v.z * m.c[2][2] + m.c[3][2]};
}
inline float3 mul_vec(const float4x4& m, float3 v) {
  return {v.x * m.c[0][0] + v.y * m.c[1][0] + v.z * m.c[2][0],
          v.x * m.c[0][1] + v.y * m.c[1][1] + v.z * m.c[2][1],
          v.x * m.c[0][2] + v.y * m.c[1][2] + v.z * m.c[2][2]};
}
inline float4x4 transpose(const float4x4& m) {
  float4x4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) r.c[i][j] = m.c[j][i];
  return r;
}

// general 4x4 inverse by cofactors, evaluated in double and rounded once
inline float4x4 inverse4x4(const float4x4& m) {
  double a[16], inv[16];
  for (int col = 0; col < 4; col++)
    for (int row = 0; row < 4; row++) a[col * 4 + row] = m.c[col][row];
  inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
  inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
  inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
  inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
  inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
  inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
  inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
  inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
  inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
  inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
  inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
  inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
  inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
  inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
  inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
  inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
  double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
  double id = 1.0 / det;
  float4x4 r;
  for (int col = 0; col < 4; col++)
    for (int row = 0; row < 4; row++) r.c[col][row] = (float)(inv[col * 4 + row] * id);
  return r;
}

// OpenGL-style camera matrices; the consumer side (EyeRayDirNormalized, hydra_drv/cglobals.h:1069-1078,
// MakeRandEyeRay cfetch.h:877-930) unprojects NDC (2x-1, 2y-1, 0) and looks down -z.
inline float4x4 perspective_matrix(float fovYdeg, float aspect, float zNear, float zFar) {
  const float ymax = zNear * tanf(fovYdeg * 3.14159265358979323846f / 360.0f);
  const float xmax = ymax * aspect;
  const float l = -xmax, r = xmax, b = -ymax, t = ymax;
  float4x4 m;
  memset(m.c, 0, sizeof(m.c));
  m.at(0, 0) = 2.0f * zNear / (r - l);
  m.at(1, 1) = 2.0f * zNear / (t - b);
  m.at(0, 2) = (r + l) / (r - l);
  m.at(1, 2) = (t + b) / (t - b);
  m.at(2, 2) = -(zFar + zNear) / (zFar - zNear);
  m.at(3, 2) = -1.0f;
  m.at(2, 3) = -2.0f * zFar * zNear / (zFar - zNear);
  return m;
}
inline float4x4 look_at(float3 eye, float3 center, float3 up) {
  const float3 f = normalize(center - eye);
  const float3 s = normalize(cross(f, up));
  const float3 u = cross(s, f);
  float4x4 m;
  m.at(0, 0) = s.x; m.at(0, 1) = s.y; m.at(0, 2) = s.z; m.at(0, 3) = -dot(s, eye);
  m.at(1, 0) = u.x; m.at(1, 1) = u.y; m.at(1, 2) = u.z; m.at(1, 3) = -dot(u, eye);
  m.at(2, 0) = -f.x; m.at(2, 1) = -f.y; m.at(2, 2) = -f.z; m.at(2, 3) = dot(f, eye);
  return m;
}

}  // namespace hydra_host
