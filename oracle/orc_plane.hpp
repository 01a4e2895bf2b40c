// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
//
// esti_plane<double> — reference include/common_lib.h:236-269 (call site src/laserMapping.cpp:997).
// The reference solves the 5x3 system  A n = -1  with Eigen's  A.colPivHouseholderQr().solve(b).
// Eigen is a third-party dependency that is NOT vendored in /root/reference (README.md:59 pins
// Eigen >= 3.3.4); its published algorithm (Eigen/src/QR/ColPivHouseholderQR.h, 3.3.x:
// LAPACK-working-note-176 column-norm down-dating, Householder reflectors, solve() over
// nonzeroPivots()) is restated here from the Eigen documentation/source structure.
// Parity status: UNPINNED by the reference (no golden vectors, Eigen not installed here).  Cross-checked against
// numpy.linalg.lstsq (tests/test_oracle_core.py) and against an independently written numpy implementation of the same
// published algorithm on well-conditioned AND rank-deficient neighbourhoods - coplanar through the origin, collinear,
// duplicated points (tests/test_oracle_plane_degenerate.py).  What cannot be derived without the library is the summation
// order inside Eigen's dynamic-size applyHouseholderOnTheLeft (a gemv kernel): the plane coefficients agree with Eigen's to
// rounding (~1e-15 relative), not bit for bit.
#pragma once
#include <cmath>
#include <limits>

namespace orc {

// Least-squares solve of A(5x3) x = b(5) by column-pivoted Householder QR.  Row-major A.
inline void colpiv_qr_solve_5x3(const double Ain[15], const double bin[5], double x[3]) {
  constexpr int R = 5, C = 3;
  double qr[R][C];
  for (int i = 0; i < R; i++)
    for (int j = 0; j < C; j++) qr[i][j] = Ain[i * C + j];
  double hcoef[C];
  int colsT[C];
  double normsUpdated[C], normsDirect[C];
  const double eps = std::numeric_limits<double>::epsilon();
  double maxnorm = 0;
  for (int k = 0; k < C; k++) {
    double s = 0;
    for (int i = 0; i < R; i++) s += qr[i][k] * qr[i][k];
    normsDirect[k] = normsUpdated[k] = std::sqrt(s);
    maxnorm = std::max(maxnorm, normsUpdated[k]);
  }
  // Eigen 3.3 ColPivHouseholderQR::computeInPlace: threshold_helper = abs2(m_colNormsUpdated.maxCoeff() * epsilon) / rows
  // (round 1 had (maxnorm eps / rows)^2 - a factor `rows` smaller; it only matters for exactly rank-deficient neighbourhoods)
  const double th = maxnorm * eps;
  const double threshold_helper = th * th / double(R);
  const double norm_downdate_threshold = std::sqrt(eps);
  int nonzero_pivots = C;
  double maxpivot = 0;
  for (int k = 0; k < C; k++) {
    int big = k;
    double bigv = normsUpdated[k];
    for (int j = k + 1; j < C; j++)
      if (normsUpdated[j] > bigv) { bigv = normsUpdated[j]; big = j; }
    double big_sq = bigv * bigv;
    if (nonzero_pivots == C && big_sq < threshold_helper * double(R - k)) nonzero_pivots = k;
    colsT[k] = big;
    if (k != big) {
      for (int i = 0; i < R; i++) std::swap(qr[i][k], qr[i][big]);
      std::swap(normsUpdated[k], normsUpdated[big]);
      std::swap(normsDirect[k], normsDirect[big]);
    }
    // makeHouseholderInPlace on qr[k..R-1][k]
    double tailSq = 0;
    for (int i = k + 1; i < R; i++) tailSq += qr[i][k] * qr[i][k];
    double c0 = qr[k][k], beta, tau;
    if (tailSq <= std::numeric_limits<double>::min()) {
      tau = 0;
      beta = c0;
      for (int i = k + 1; i < R; i++) qr[i][k] = 0;
    } else {
      beta = std::sqrt(c0 * c0 + tailSq);
      if (c0 >= 0) beta = -beta;
      for (int i = k + 1; i < R; i++) qr[i][k] /= (c0 - beta);
      tau = (beta - c0) / beta;
    }
    hcoef[k] = tau;
    qr[k][k] = beta;
    if (std::fabs(beta) > maxpivot) maxpivot = std::fabs(beta);
    // apply H = I - tau v v^T (v = [1; essential]) to the trailing columns
    if (tau != 0) {
      for (int j = k + 1; j < C; j++) {
        double tmp = qr[k][j];
        for (int i = k + 1; i < R; i++) tmp += qr[i][k] * qr[i][j];
        qr[k][j] -= tau * tmp;
        for (int i = k + 1; i < R; i++) qr[i][j] -= tau * qr[i][k] * tmp;
      }
    }
    // column-norm down-date (LAPACK WN 176)
    for (int j = k + 1; j < C; j++) {
      if (normsUpdated[j] != 0) {
        double temp = std::fabs(qr[k][j]) / normsUpdated[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0 ? 0 : temp;
        double ratio = normsUpdated[j] / normsDirect[j];
        double temp2 = temp * ratio * ratio;
        if (temp2 <= norm_downdate_threshold) {
          double s = 0;
          for (int i = k + 1; i < R; i++) s += qr[i][j] * qr[i][j];
          normsDirect[j] = std::sqrt(s);
          normsUpdated[j] = normsDirect[j];
        } else {
          normsUpdated[j] *= std::sqrt(temp);
        }
      }
    }
  }
  // solve(): c = Q^T b over the first nonzero_pivots reflectors, back-substitute, un-permute.
  double c[R];
  for (int i = 0; i < R; i++) c[i] = bin[i];
  for (int k = 0; k < nonzero_pivots; k++) {
    double tau = hcoef[k];
    if (tau == 0) continue;
    double tmp = c[k];
    for (int i = k + 1; i < R; i++) tmp += qr[i][k] * c[i];
    c[k] -= tau * tmp;
    for (int i = k + 1; i < R; i++) c[i] -= tau * qr[i][k] * tmp;
  }
  double y[C] = {0, 0, 0};
  for (int i = nonzero_pivots - 1; i >= 0; i--) {
    double s = c[i];
    for (int j = i + 1; j < nonzero_pivots; j++) s -= qr[i][j] * y[j];
    y[i] = s / qr[i][i];
  }
  // colsPermutation = product of transpositions: indices[] built by applying them in order
  int perm[C] = {0, 1, 2};
  for (int k = 0; k < C; k++) std::swap(perm[k], perm[colsT[k]]);
  for (int i = 0; i < C; i++) x[i] = 0;
  for (int i = 0; i < nonzero_pivots; i++) x[perm[i]] = y[i];
}

// esti_plane<double>(pca_result, point, threshold) — common_lib.h:236-269.
// pts: 5 neighbours as float xyz (15 floats).  Returns validity; pabcd = (n̂, d).
inline bool esti_plane(double pabcd[4], const float pts[15], double threshold) {
  double A[15], b[5];
  for (int j = 0; j < 5; j++) {
    A[3 * j + 0] = pts[3 * j + 0];
    A[3 * j + 1] = pts[3 * j + 1];
    A[3 * j + 2] = pts[3 * j + 2];
    b[j] = -1.0;
  }
  double nv[3];
  colpiv_qr_solve_5x3(A, b, nv);
  double n = std::sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
  pabcd[0] = nv[0] / n;
  pabcd[1] = nv[1] / n;
  pabcd[2] = nv[2] / n;
  pabcd[3] = 1.0 / n;
  for (int j = 0; j < 5; j++) {
    if (std::fabs(pabcd[0] * pts[3 * j] + pabcd[1] * pts[3 * j + 1] + pabcd[2] * pts[3 * j + 2] + pabcd[3]) >
        threshold)
      return false;
  }
  return true;
}

}  // namespace orc
