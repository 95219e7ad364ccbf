// mt_shade.h — per-hit arithmetic of the reference, restated for the device:
// V3D (math3d.h), Triangle::GetNormal/GetUVW (primitive_triangle.cc:27-79),
// Texture::GetColorAt (texture.cc:11-58), V3DtoRGB (mythtracer.cc:235-241).
// Operand order and association follow the reference expression by
// expression; the translation unit is built with -ffp-contract=off.
#pragma once
#include "mt_trace.h"

namespace mt {

struct V3 {
  double x, y, z;
};

__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 v3_load(const double *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, double n) { return V3{a.x * n, a.y * n, a.z * n}; }
__device__ __forceinline__ V3 operator/(V3 a, double n) { return V3{a.x / n, a.y / n, a.z / n}; }
// a.Dot(b) == b.Dot(a): products commute, the sum order is x, y, z.
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double sqr_distance(V3 self, V3 a) {  // math3d.h:105-110
  const double dx = a.x - self.x, dy = a.y - self.y, dz = a.z - self.z;
  return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ double distance(V3 self, V3 a) { return __builtin_sqrt(sqr_distance(self, a)); }
__device__ __forceinline__ V3 normalized(V3 a) {  // Norm(), math3d.h:128-131
  const double l = __builtin_sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  return V3{a.x / l, a.y / l, a.z / l};
}

__device__ __forceinline__ double area_of_triangle(double a, double b, double c) {
  const double p = (a + b + c) / 2.0;  // primitive_triangle.cc:27-40
  const double area_sqr = p * (p - a) * (p - b) * (p - c);
  if (area_sqr < 0.0) return 0.0;
  return __builtin_sqrt(area_sqr);
}

// Barycentric weights by Heron areas, shared by GetNormal and GetUVW.
struct Bary {
  double n0, n1, n2, n;
};
__device__ __forceinline__ Bary barycentric(const double *vtx, V3 point) {
  const V3 v0 = v3_load(vtx), v1 = v3_load(vtx + 3), v2 = v3_load(vtx + 6);
  const double a = distance(v0, v1);
  const double b = distance(v1, v2);
  const double c = distance(v2, v0);
  const double p0 = distance(point, v0);
  const double p1 = distance(point, v1);
  const double p2 = distance(point, v2);
  Bary w;
  w.n0 = area_of_triangle(b, p2, p1);
  w.n1 = area_of_triangle(c, p0, p2);
  w.n2 = area_of_triangle(a, p1, p0);
  w.n = w.n0 + w.n1 + w.n2;
  return w;
}
__device__ __forceinline__ V3 interpolate(const double *attr, const Bary &w) {
  return (v3_load(attr) * w.n0 + v3_load(attr + 3) * w.n1 + v3_load(attr + 6) * w.n2) / w.n;
}

// fmod(x, 1.0) is exact: x - trunc(x), sign of x.
__device__ __forceinline__ double fmod1(double x) { return ::fmod(x, 1.0); }

__device__ __forceinline__ V3 texel(const DevTexture &t, size_t idx) {
  if (t.format == MT_TEX_RGB8) {
    const uint8_t *p = (const uint8_t *)t.texels + idx * 3;
    return V3{(double)p[0] / 255.0, (double)p[1] / 255.0, (double)p[2] / 255.0};  // texture.cc:100-104
  }
  const double *p = (const double *)t.texels + idx * 3;
  return V3{p[0], p[1], p[2]};
}

__device__ __forceinline__ V3 texture_color_at(const DevTexture &t, double u, double v) {
  // NaN coordinates (degenerate triangles): the reference's (size_t)NaN index makes
  // colors.at() throw; defined here, like every out-of-range texel, as a NaN colour.
  if (u != u || v != v) return V3{__builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
  u = fmod1(u);
  v = fmod1(v);
  if (u < 0.0) u += 1.0;
  if (v < 0.0) v += 1.0;
  v = 1.0 - v;
  const double x = u * (double)(t.width - 1);
  const double y = v * (double)(t.height - 1);
  const size_t w = (size_t)t.width, h = (size_t)t.height;
  const size_t bx = (size_t)x, by = (size_t)y;
  const size_t x1 = (bx + 1 == w) ? bx : bx + 1;
  const size_t y1 = (by + 1 == h) ? by : by + 1;
  const size_t i0 = bx + by * w, i1 = x1 + by * w, i2 = bx + y1 * w, i3 = x1 + y1 * w;
  const size_t n = w * h;
  if (i0 >= n || i1 >= n || i2 >= n || i3 >= n) {
    // colors.at() would throw in the reference; keep the pixel defined.
    return V3{__builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
  }
  const double dxf = fmod1(x), dyf = fmod1(y);
  const double a0 = (1.0 - dxf) * (1.0 - dyf), a1 = dxf * (1.0 - dyf), a2 = (1.0 - dxf) * dyf,
               a3 = dxf * dyf;
  return texel(t, i0) * a0 + texel(t, i1) * a1 + texel(t, i2) * a2 + texel(t, i3) * a3;
}

__device__ __forceinline__ uint8_t channel_to_u8(double v) {  // mythtracer.cc:235-241
  if (v > 1.0) return 255;
  if (v < 0.0) return 0;
  if (v != v) return 0;  // NaN: the reference's cast is UB; x86-64 yields 0
  return (uint8_t)(int)(v * 255);
}

}  // namespace mt
