// Host-side small algebra of libliinit_hip: SO(3) maps, the 24-state boxplus/boxminus and dense
// square-matrix helpers used around the device kernels.  (Product code: independent of oracle/.)
//   Exp3 / Log ............. reference include/so3_math.h:61-79, :100-107
//   state_plus / state_minus  reference include/common_lib.h:126-154
#pragma once
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/liinit_hip.h"

namespace lii {

constexpr int kDim = 24;

inline void m3_mul(const double* A, const double* B, double* C) {
  double t[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) t[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
  std::memcpy(C, t, sizeof(t));
}
inline void m3t_mul(const double* A, const double* B, double* C) {  // A^T B
  double t[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) t[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
  std::memcpy(C, t, sizeof(t));
}
inline void m3_vec(const double* A, const double* v, double* o) {
  double t[3];
  for (int r = 0; r < 3; r++) t[r] = A[3 * r] * v[0] + A[3 * r + 1] * v[1] + A[3 * r + 2] * v[2];
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
inline void m3_identity(double* R) {
  for (int e = 0; e < 9; e++) R[e] = (e % 4 == 0) ? 1.0 : 0.0;
}

// Rodrigues with an explicit angle about the normalised axis; identity below `thr`.
inline void so3_exp(double v1, double v2, double v3, double thr, double* R) {
  double n = std::sqrt(v1 * v1 + v2 * v2 + v3 * v3);
  m3_identity(R);
  if (n > thr) {
    double a[3] = {v1 / n, v2 / n, v3 / n};
    double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
    double cK[9], KK[9];
    double s = std::sin(n), c1 = 1.0 - std::cos(n);
    for (int e = 0; e < 9; e++) cK[e] = c1 * K[e];  // `(1.0 - cos) * K * K` = ((1 - cos) K) K (so3_math.h:73)
    m3_mul(cK, K, KK);
    for (int e = 0; e < 9; e++) R[e] = (R[e] + s * K[e]) + KK[e];
  }
}
inline void so3_log(const double* R, double* out) {
  double tr = R[0] + R[4] + R[8];
  double theta = (tr > 3.0 - 1e-6) ? 0.0 : std::acos(0.5 * (tr - 1));
  double K[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double f = (std::fabs(theta) < 0.001) ? 0.5 : (0.5 * theta / std::sin(theta));
  out[0] = f * K[0]; out[1] = f * K[1]; out[2] = f * K[2];
}

// x (+) d  — StatesGroup::operator+= (cov untouched)
inline void state_plus(lii_state& s, const double* d) {
  double E[9];
  so3_exp(d[0], d[1], d[2], 0.00001, E);
  m3_mul(s.rot_end, E, s.rot_end);
  so3_exp(d[6], d[7], d[8], 0.00001, E);
  m3_mul(s.offset_R_L_I, E, s.offset_R_L_I);
  for (int i = 0; i < 3; i++) {
    s.pos_end[i] += d[3 + i];
    s.offset_T_L_I[i] += d[9 + i];
    s.vel_end[i] += d[12 + i];
    s.bias_g[i] += d[15 + i];
    s.bias_a[i] += d[18 + i];
    s.gravity[i] += d[21 + i];
  }
}
// a (-) b — StatesGroup::operator-
inline void state_minus(const lii_state& a, const lii_state& b, double* out) {
  double R[9];
  m3t_mul(b.rot_end, a.rot_end, R);
  so3_log(R, out);
  m3t_mul(b.offset_R_L_I, a.offset_R_L_I, R);
  so3_log(R, out + 6);
  for (int i = 0; i < 3; i++) {
    out[3 + i] = a.pos_end[i] - b.pos_end[i];
    out[9 + i] = a.offset_T_L_I[i] - b.offset_T_L_I[i];
    out[12 + i] = a.vel_end[i] - b.vel_end[i];
    out[15 + i] = a.bias_g[i] - b.bias_g[i];
    out[18 + i] = a.bias_a[i] - b.bias_a[i];
    out[21 + i] = a.gravity[i] - b.gravity[i];
  }
}

// In-place LU (partial pivoting) inverse of an n x n row-major matrix; returns false if singular.
inline bool mat_inverse(const double* A, int n, double* inv) {
  std::vector<double> lu(A, A + size_t(n) * n);
  std::vector<int> piv(n);
  for (int i = 0; i < n; i++) piv[i] = i;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::fabs(lu[size_t(k) * n + k]);
    for (int i = k + 1; i < n; i++) {
      double v = std::fabs(lu[size_t(i) * n + k]);
      if (v > best) { best = v; p = i; }
    }
    if (best == 0.0) return false;
    if (p != k) {
      for (int j = 0; j < n; j++) std::swap(lu[size_t(k) * n + j], lu[size_t(p) * n + j]);
      std::swap(piv[k], piv[p]);
    }
    const double d = lu[size_t(k) * n + k];
    for (int i = k + 1; i < n; i++) {
      double l = lu[size_t(i) * n + k] / d;
      lu[size_t(i) * n + k] = l;
      if (l != 0.0)
        for (int j = k + 1; j < n; j++) lu[size_t(i) * n + j] -= l * lu[size_t(k) * n + j];
    }
  }
  std::vector<double> y(n);
  for (int col = 0; col < n; col++) {
    for (int i = 0; i < n; i++) {
      double s = (piv[i] == col) ? 1.0 : 0.0;
      for (int j = 0; j < i; j++) s -= lu[size_t(i) * n + j] * y[j];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = y[i];
      for (int j = i + 1; j < n; j++) s -= lu[size_t(i) * n + j] * inv[size_t(j) * n + col];
      inv[size_t(i) * n + col] = s / lu[size_t(i) * n + i];
    }
  }
  return true;
}

// Symmetric positive-definite solve A x = b by Cholesky (n <= 16); returns false if not SPD.
inline bool spd_solve(const double* A, const double* b, int n, double* x) {
  double L[16 * 16];
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      if (i == j) {
        if (s <= 0) return false;
        L[i * n + i] = std::sqrt(s);
      } else {
        L[i * n + j] = s / L[j * n + j];
      }
    }
  double y[16];
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k];
    y[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
  return true;
}

}  // namespace lii
