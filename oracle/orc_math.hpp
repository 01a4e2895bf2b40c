// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product path;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// CPU restatement (dependency-free C++17, double precision exactly where the reference uses double)
// of the small SO(3) / dense-matrix helpers of hku-mars/LiDAR_IMU_Init.
//   Exp / Log / RotMtoEuler ...... reference include/so3_math.h:18-129
//   24-state boxplus / boxminus .. reference include/common_lib.h:68-169
// Parity status: PINNED - held bit for bit to the reference's own so3_math.h / common_lib.h compiled unmodified against a
// minimal matrix shim (oracle/ref_shim_math -> oracle/_ref/libref_math.so, tests/test_oracle_math_pinned.py), and cross-checked
// against scipy.spatial.transform (tests/test_oracle_core.py).  Note the Rodrigues term: the reference writes
// `(1.0 - cos) * K * K`, which C++ evaluates as ((1 - cos) K) K - the scalar is rounded into K before the product.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

namespace orc {

struct V3 {
  double x = 0, y = 0, z = 0;
  V3() = default;
  V3(double a, double b, double c) : x(a), y(b), z(c) {}
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(const V3& a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double s, const V3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(const V3& a, double s) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(const V3& a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(const V3& a) { return std::sqrt(dot(a, a)); }
inline V3 cross(const V3& a, const V3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Row-major 3x3.
struct M3 {
  double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double operator()(int r, int c) const { return m[3 * r + c]; }
  double& operator()(int r, int c) { return m[3 * r + c]; }
  static M3 identity() {
    M3 r;
    r.m[0] = r.m[4] = r.m[8] = 1.0;
    return r;
  }
  static M3 from(const double* p) {
    M3 r;
    std::memcpy(r.m, p, sizeof(r.m));
    return r;
  }
};
inline M3 operator*(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += a(i, k) * b(k, j);
      r(i, j) = s;
    }
  return r;
}
inline V3 operator*(const M3& a, const V3& v) {
  return {a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z, a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z,
          a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z};
}
inline M3 operator+(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 9; i++) r.m[i] = a.m[i] + b.m[i];
  return r;
}
inline M3 operator*(double s, const M3& a) {
  M3 r;
  for (int i = 0; i < 9; i++) r.m[i] = s * a.m[i];
  return r;
}
inline M3 transpose(const M3& a) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r(i, j) = a(j, i);
  return r;
}
// SKEW_SYM_MATRX — reference include/so3_math.h:8
inline M3 skew(const V3& v) {
  M3 r;
  r(0, 1) = -v.z; r(0, 2) = v.y;
  r(1, 0) = v.z;  r(1, 2) = -v.x;
  r(2, 0) = -v.y; r(2, 1) = v.x;
  return r;
}

// Exp(ang) — reference include/so3_math.h:18-35  (identity iff |ang| <= 1e-7)
inline M3 Exp(const V3& ang) {
  double n = norm(ang);
  if (n > 0.0000001) {
    M3 K = skew(ang / n);
    return M3::identity() + std::sin(n) * K + ((1.0 - std::cos(n)) * K) * K;
  }
  return M3::identity();
}
// Exp(ang_vel, dt) — reference include/so3_math.h:37-59
inline M3 Exp(const V3& w, double dt) {
  double n = norm(w);
  if (n > 0.0000001) {
    M3 K = skew(w / n);
    double r = n * dt;
    return M3::identity() + std::sin(r) * K + ((1.0 - std::cos(r)) * K) * K;
  }
  return M3::identity();
}
// Exp(v1,v2,v3) — reference include/so3_math.h:61-79  (threshold 1e-5; used by StatesGroup::operator+)
inline M3 Exp3(double v1, double v2, double v3) {
  double n = std::sqrt(v1 * v1 + v2 * v2 + v3 * v3);
  if (n > 0.00001) {
    M3 K = skew(V3(v1 / n, v2 / n, v3 / n));
    return M3::identity() + std::sin(n) * K + ((1.0 - std::cos(n)) * K) * K;
  }
  return M3::identity();
}
// Log(R) — reference include/so3_math.h:100-107
inline V3 Log(const M3& R) {
  double tr = R(0, 0) + R(1, 1) + R(2, 2);
  double theta = (tr > 3.0 - 1e-6) ? 0.0 : std::acos(0.5 * (tr - 1));
  V3 K(R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1));
  return (std::fabs(theta) < 0.001) ? (0.5 * K) : ((0.5 * theta / std::sin(theta)) * K);
}
// RotMtoEuler — reference include/so3_math.h:109-129
inline V3 RotMtoEuler(const M3& rot) {
  double sy = std::sqrt(rot(0, 0) * rot(0, 0) + rot(1, 0) * rot(1, 0));
  bool singular = sy < 1e-6;
  double x, y, z;
  if (!singular) {
    x = std::atan2(rot(2, 1), rot(2, 2));
    y = std::atan2(-rot(2, 0), sy);
    z = std::atan2(rot(1, 0), rot(0, 0));
  } else {
    x = std::atan2(-rot(1, 2), rot(1, 1));
    y = std::atan2(-rot(2, 0), sy);
    z = 0;
  }
  return {x, y, z};
}

// ---------------------------------------------------------------------------------------------
// Small dense matrices (row-major, runtime n) for the 24x24 filter algebra.
struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() = default;
  Mat(int r_, int c_) : r(r_), c(c_), a(size_t(r_) * c_, 0.0) {}
  double operator()(int i, int j) const { return a[size_t(i) * c + j]; }
  double& operator()(int i, int j) { return a[size_t(i) * c + j]; }
  static Mat identity(int n) {
    Mat m(n, n);
    for (int i = 0; i < n; i++) m(i, i) = 1.0;
    return m;
  }
};
inline Mat matmul(const Mat& A, const Mat& B) {
  Mat C(A.r, B.c);
  for (int i = 0; i < A.r; i++)
    for (int k = 0; k < A.c; k++) {
      double aik = A(i, k);
      if (aik == 0.0) continue;
      for (int j = 0; j < B.c; j++) C(i, j) += aik * B(k, j);
    }
  return C;
}
// General inverse by LU with partial pivoting (what Eigen's .inverse() does for n > 4:
// PartialPivLU — reference call sites src/laserMapping.cpp:1081).
inline Mat inverse(const Mat& A) {
  int n = A.r;
  Mat LU = A;
  std::vector<int> piv(n);
  for (int i = 0; i < n; i++) piv[i] = i;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::fabs(LU(k, k));
    for (int i = k + 1; i < n; i++)
      if (std::fabs(LU(i, k)) > best) { best = std::fabs(LU(i, k)); p = i; }
    if (p != k) {
      for (int j = 0; j < n; j++) std::swap(LU(k, j), LU(p, j));
      std::swap(piv[k], piv[p]);
    }
    double d = LU(k, k);
    for (int i = k + 1; i < n; i++) {
      LU(i, k) /= d;
      double l = LU(i, k);
      if (l == 0.0) continue;
      for (int j = k + 1; j < n; j++) LU(i, j) -= l * LU(k, j);
    }
  }
  Mat inv(n, n);
  std::vector<double> y(n);
  for (int col = 0; col < n; col++) {
    // solve L U x = P e_col
    for (int i = 0; i < n; i++) {
      double s = (piv[i] == col) ? 1.0 : 0.0;
      for (int j = 0; j < i; j++) s -= LU(i, j) * y[j];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = y[i];
      for (int j = i + 1; j < n; j++) s -= LU(i, j) * inv(j, col);
      inv(i, col) = s / LU(i, i);
    }
  }
  return inv;
}

// ---------------------------------------------------------------------------------------------
// StatesGroup — reference include/common_lib.h:68-169.  POD mirror shared with the ctypes layer:
// 9+3+9+3+3+3+3+3 = 36 doubles followed by the 24x24 covariance (row-major) = 612 doubles.
constexpr int DIM_STATE = 24;
struct State {
  M3 rot_end = M3::identity();
  V3 pos_end;
  M3 offset_R_L_I = M3::identity();
  V3 offset_T_L_I;
  V3 vel_end;
  V3 bias_g;
  V3 bias_a;
  V3 gravity;
  Mat cov;
  State() : cov(Mat::identity(DIM_STATE)) {
    for (int i = 15; i < 24; i++) cov(i, i) = 0.00001;  // common_lib.h:79-80
  }
};
// operator+= — common_lib.h:126-137
inline void boxplus(State& s, const double* d) {
  s.rot_end = s.rot_end * Exp3(d[0], d[1], d[2]);
  s.pos_end = s.pos_end + V3(d[3], d[4], d[5]);
  s.offset_R_L_I = s.offset_R_L_I * Exp3(d[6], d[7], d[8]);
  s.offset_T_L_I = s.offset_T_L_I + V3(d[9], d[10], d[11]);
  s.vel_end = s.vel_end + V3(d[12], d[13], d[14]);
  s.bias_g = s.bias_g + V3(d[15], d[16], d[17]);
  s.bias_a = s.bias_a + V3(d[18], d[19], d[20]);
  s.gravity = s.gravity + V3(d[21], d[22], d[23]);
}
// operator- (a ⊟ b) — common_lib.h:139-154
inline void boxminus(const State& a, const State& b, double* out) {
  V3 r = Log(transpose(b.rot_end) * a.rot_end);
  V3 p = a.pos_end - b.pos_end;
  V3 ro = Log(transpose(b.offset_R_L_I) * a.offset_R_L_I);
  V3 to = a.offset_T_L_I - b.offset_T_L_I;
  V3 v = a.vel_end - b.vel_end;
  V3 bg = a.bias_g - b.bias_g;
  V3 ba = a.bias_a - b.bias_a;
  V3 g = a.gravity - b.gravity;
  const V3* blk[8] = {&r, &p, &ro, &to, &v, &bg, &ba, &g};
  for (int k = 0; k < 8; k++)
    for (int i = 0; i < 3; i++) out[3 * k + i] = (*blk[k])[i];
}

constexpr int STATE_DOUBLES = 36 + DIM_STATE * DIM_STATE;
inline void state_to_pod(const State& s, double* p) {
  std::memcpy(p, s.rot_end.m, 72);
  p[9] = s.pos_end.x; p[10] = s.pos_end.y; p[11] = s.pos_end.z;
  std::memcpy(p + 12, s.offset_R_L_I.m, 72);
  const V3* v[5] = {&s.offset_T_L_I, &s.vel_end, &s.bias_g, &s.bias_a, &s.gravity};
  for (int k = 0; k < 5; k++)
    for (int i = 0; i < 3; i++) p[21 + 3 * k + i] = (*v[k])[i];
  std::memcpy(p + 36, s.cov.a.data(), sizeof(double) * DIM_STATE * DIM_STATE);
}
inline State state_from_pod(const double* p) {
  State s;
  s.rot_end = M3::from(p);
  s.pos_end = V3(p[9], p[10], p[11]);
  s.offset_R_L_I = M3::from(p + 12);
  V3* v[5] = {&s.offset_T_L_I, &s.vel_end, &s.bias_g, &s.bias_a, &s.gravity};
  for (int k = 0; k < 5; k++)
    for (int i = 0; i < 3; i++) (*v[k])[i] = p[21 + 3 * k + i];
  std::memcpy(s.cov.a.data(), p + 36, sizeof(double) * DIM_STATE * DIM_STATE);
  return s;
}

// pcl::PointXYZINormal layout (48 B) — reference include/common_lib.h:37 ; verified in SURVEY §8(c).
struct alignas(16) PointXYZINormal {
  float x, y, z, pad0;
  float normal_x, normal_y, normal_z, pad1;
  float intensity, curvature, pad2, pad3;
};
static_assert(sizeof(PointXYZINormal) == 48, "PointXYZINormal must be 48 bytes");

}  // namespace orc
