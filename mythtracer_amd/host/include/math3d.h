// math3d.h — host-side vector/matrix types of the drop-in facade.
//
// Source-compatible with the public surface of the reference's
// VerStarting/math3d.h (namespace math3d, V3D with a public `v[3]`, accessor
// aliases, Dot/Cross/Norm/..., M4D rotations, Deg2Rad) so that drivers written
// against the reference (main_local.cc:72-110) compile unchanged.  Written from
// scratch; every expression keeps the reference's operand order because the
// camera set-up (camera.cc:27-63) must come out bit-identical.
#pragma once
#include <cmath>
#include <cstddef>
#include <iomanip>
#include <ostream>
#include <sstream>
#include <string>

namespace math3d {

template <typename T>
struct V3D_Base {
  using basetype = T;
  T v[3]{};

  // element access (aliases: xyz for positions, rgb for colours)
  T& x() { return v[0]; }
  T& y() { return v[1]; }
  T& z() { return v[2]; }
  T& r() { return v[0]; }
  T& g() { return v[1]; }
  T& b() { return v[2]; }
  const T& x() const { return v[0]; }
  const T& y() const { return v[1]; }
  const T& z() const { return v[2]; }
  const T& r() const { return v[0]; }
  const T& g() const { return v[1]; }
  const T& b() const { return v[2]; }

  // component-wise arithmetic between vectors
  friend V3D_Base operator+(const V3D_Base& a, const V3D_Base& b) {
    return {a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]};
  }
  friend V3D_Base operator-(const V3D_Base& a, const V3D_Base& b) {
    return {a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]};
  }
  friend V3D_Base operator*(const V3D_Base& a, const V3D_Base& b) {
    return {a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2]};
  }
  friend V3D_Base operator/(const V3D_Base& a, const V3D_Base& b) {
    return {a.v[0] / b.v[0], a.v[1] / b.v[1], a.v[2] / b.v[2]};
  }
  V3D_Base operator-() const { return {-v[0], -v[1], -v[2]}; }
  V3D_Base operator+() const { return *this; }
  V3D_Base& operator+=(const V3D_Base& o) { return *this = *this + o; }
  V3D_Base& operator-=(const V3D_Base& o) { return *this = *this - o; }
  V3D_Base& operator*=(const V3D_Base& o) { return *this = *this * o; }
  V3D_Base& operator/=(const V3D_Base& o) { return *this = *this / o; }

  // scalar on the right-hand side
  friend V3D_Base operator*(const V3D_Base& a, T n) { return {a.v[0] * n, a.v[1] * n, a.v[2] * n}; }
  friend V3D_Base operator/(const V3D_Base& a, T n) { return {a.v[0] / n, a.v[1] / n, a.v[2] / n}; }
  V3D_Base& operator*=(T n) { return *this = *this * n; }
  V3D_Base& operator/=(T n) { return *this = *this / n; }

  T SqrLength() const { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; }
  T Length() const { return std::sqrt(SqrLength()); }
  T SqrDistance(const V3D_Base& o) const {
    const T dx = o.v[0] - v[0], dy = o.v[1] - v[1], dz = o.v[2] - v[2];
    return dx * dx + dy * dy + dz * dz;
  }
  T Distance(const V3D_Base& o) const { return std::sqrt(SqrDistance(o)); }
  T Dot(const V3D_Base& o) const { return o.v[0] * v[0] + o.v[1] * v[1] + o.v[2] * v[2]; }
  V3D_Base Cross(const V3D_Base& o) const {
    return {v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2],
            v[0] * o.v[1] - v[1] * o.v[0]};
  }
  V3D_Base DupNorm() const {
    const T len = Length();
    return {v[0] / len, v[1] / len, v[2] / len};
  }
  void Norm() { *this = DupNorm(); }
};

template <typename T>
std::ostream& operator<<(std::ostream& os, const V3D_Base<T>& a) {
  return os << std::fixed << std::setprecision(5) << a.v[0] << ", " << a.v[1] << ", " << a.v[2];
}

template <typename T>
std::string ToStr(const T& a) {
  std::ostringstream s;
  s << a;
  return s.str();
}
#define V3DStr(a) math3d::ToStr(a).c_str()
#define M4DStr(a) math3d::ToStr(a).c_str()

inline double Deg2Rad(double angle) { return (angle * M_PI) / 180.0; }

// 4x4 matrix; only rotations are ever built, the translation column stays 0.
template <typename T>
struct M4D_Base {
  using basetype = T;
  T m[4][4]{};

  M4D_Base operator*(const M4D_Base& a) const {
    M4D_Base out;
    for (size_t row = 0; row < 4; row++) {
      for (size_t col = 0; col < 4; col++) {
        out.m[row][col] = m[row][0] * a.m[0][col] + m[row][1] * a.m[1][col] +
                          m[row][2] * a.m[2][col] + m[row][3] * a.m[3][col];
      }
    }
    return out;
  }
  M4D_Base& operator*=(const M4D_Base& a) { return *this = *this * a; }

  // The 4th vector element is taken as 1.  The reference adds m[0][3] on all
  // three rows (math3d.h:212-214); it is always 0, and kept for bit parity.
  template <typename U>
  V3D_Base<U> operator*(const V3D_Base<U>& a) const {
    return {m[0][0] * a.v[0] + m[0][1] * a.v[1] + m[0][2] * a.v[2] + m[0][3],
            m[1][0] * a.v[0] + m[1][1] * a.v[1] + m[1][2] * a.v[2] + m[0][3],
            m[2][0] * a.v[0] + m[2][1] * a.v[1] + m[2][2] * a.v[2] + m[0][3]};
  }

  void ResetIdentity() {
    for (size_t row = 0; row < 4; row++)
      for (size_t col = 0; col < 4; col++) m[row][col] = (row == col) ? 1.0 : 0.0;
  }
  void ResetRotationXRad(T a) {
    const T c = std::cos(a), s = std::sin(a);
    *this = {{{1.0, 0.0, 0.0, 0.0}, {0.0, c, -s, 0.0}, {0.0, s, c, 0.0}, {0.0, 0.0, 0.0, 1.0}}};
  }
  void ResetRotationYRad(T a) {
    const T c = std::cos(a), s = std::sin(a);
    *this = {{{c, 0.0, s, 0.0}, {0.0, 1.0, 0.0, 0.0}, {-s, 0.0, c, 0.0}, {0.0, 0.0, 0.0, 1.0}}};
  }
  void ResetRotationZRad(T a) {
    const T c = std::cos(a), s = std::sin(a);
    *this = {{{c, -s, 0.0, 0.0}, {s, c, 0.0, 0.0}, {0.0, 0.0, 1.0, 0.0}, {0.0, 0.0, 0.0, 1.0}}};
  }
  static M4D_Base RotationXRad(T a) { M4D_Base r; r.ResetRotationXRad(a); return r; }
  static M4D_Base RotationYRad(T a) { M4D_Base r; r.ResetRotationYRad(a); return r; }
  static M4D_Base RotationZRad(T a) { M4D_Base r; r.ResetRotationZRad(a); return r; }
  static M4D_Base RotationXDeg(T a) { return RotationXRad(Deg2Rad(a)); }
  static M4D_Base RotationYDeg(T a) { return RotationYRad(Deg2Rad(a)); }
  static M4D_Base RotationZDeg(T a) { return RotationZRad(Deg2Rad(a)); }
};

template <typename T>
std::ostream& operator<<(std::ostream& os, const M4D_Base<T>& a) {
  os << std::fixed << std::setprecision(5);
  for (int row = 0; row < 4; row++) {
    os << (row == 0 ? "[  " : "   ") << a.m[row][0] << ", " << a.m[row][1] << ", " << a.m[row][2]
       << ", " << a.m[row][3] << (row == 3 ? "  ]\n" : "   \n");
  }
  return os;
}

typedef V3D_Base<double> V3D;
typedef M4D_Base<double> M4D;

}  // namespace math3d
