// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
//
// Scan-to-map registration: the iterated error-state Kalman update with a point-to-plane
// measurement, restated from the inline body of main() in the reference:
//   pointBodyToWorld ................. src/laserMapping.cpp:209-220
//   residual / selection loop ........ src/laserMapping.cpp:964-1012 (OpenMP loop :964-968)
//   compaction ....................... src/laserMapping.cpp:1013-1020
//   Jacobian rows .................... src/laserMapping.cpp:1035-1071 (calcBodyVar :1041 is dead code)
//   gain / update / convergence ...... src/laserMapping.cpp:1073-1106
//   covariance update ................ src/laserMapping.cpp:1109-1131
//   map_incremental .................. src/laserMapping.cpp:516-559
// Documented deviations:
//   * the fixed 100 000-element arrays (quirk A1, :108-109,117-119) are lifted to std::vector;
//   * K(24 x m) is never materialised by default: K z = K1[:, :12](H^T R^-1 z) and
//     K H = K1[:, :12](H^T R^-1 H) exactly; `literal_gain` forms K as the reference
//     does, for the equivalence test.
// Parity status: PINNED TO THE REFERENCE'S OWN TEXT around the linear algebra (round 6): `make -C oracle ref` cuts
// src/laserMapping.cpp:936-1134 and :516-559 out of the reference at build time and compiles them, with the unmodified
// include/common_lib.h, so3_math.h and ikd-Tree, into oracle/_ref/libref_iekf.so; tests/test_oracle_iekf_pinned.py holds this file -
// in `literal_gain` mode - BIT FOR BIT to it over consecutive LO and LIO scans with map_incremental between them (state, covariance,
// schedule, selected set, normvec, the tree), and measures the default sums form against it (<= 1e-12 LO, <= 1e-10 LIO).  Still
// unpinned (no Eigen on disk): the insides of Eigen's inverse(), ColPivHouseholderQR and matrix products, which that library takes
// from this oracle.  Independent checks: closed-form synthetic ground truth (tests/test_oracle_core.py) and an independent numpy
// iteration with the literal gain (tests/test_oracle_iekf_independent.py).
#pragma once
#include <cstdint>
#include <vector>

#include "orc_kdtree.hpp"
#include "orc_math.hpp"
#include "orc_plane.hpp"
#include "orc_scan.hpp"

namespace orc {

constexpr int NUM_MATCH_POINTS = 5;

struct IekfParams {
  int max_iterations = 4;        // NUM_MAX_ITERATIONS (launch files set 5)
  int imu_en = 0;                // 0: LO mode (cols 6..11 of H are zero), 1: LIO mode
  int num_threads = 1;           // MP_PROC_NUM analogue
  int literal_gain = 0;          // 1: form the 24 x m gain like the reference
  double plane_threshold = 0.1;  // esti_plane threshold
  double max_dist = 5.0;         // Nearest_Search max_dist argument (quirk A5)
  double laser_point_cov_inv = 1000.0;  // R_inv(i) = 1000 = 1 / LASER_POINT_COV
};

// Per-iteration record for parity checks against the HIP path.
struct IekfIterLog {
  int searched = 0;
  int effect_num = 0;
  double HTH[78];  // upper triangle (row-major, i <= j) of H^T R^-1 H (12 x 12)
  double HTz[12];  // H^T R^-1 z
  double solution[24];
};

struct IekfScratch {
  std::vector<float> world;       // n x 3 (float world coordinates)
  std::vector<float> nearest;     // n x 5 x 3
  std::vector<float> nearest_d2;  // n x 5
  std::vector<int32_t> nearest_n; // n
  std::vector<uint8_t> selected;  // n
  std::vector<float> normvec;     // n x 4: n̂ (float) + pd2 (float) — PointCloud `normvec`
  std::vector<double> pabcd;      // n x 4 (debug)
  std::vector<float> res_last;    // n
};

inline void point_body_to_world(const State& s, const float* pb, float* pw) {
  V3 p_body(pb[0], pb[1], pb[2]);
  V3 g = s.rot_end * (s.offset_R_L_I * p_body + s.offset_T_L_I) + s.pos_end;
  pw[0] = float(g.x);
  pw[1] = float(g.y);
  pw[2] = float(g.z);
}

// One pass of the per-point loop (:964-1012).
inline void residual_pass(const KdTree& tree, const std::vector<P4>& body, const State& st, bool search,
                          const IekfParams& prm, IekfScratch& sc) {
  const int n = int(body.size());
#ifdef _OPENMP
#pragma omp parallel for num_threads(prm.num_threads) schedule(static)
#endif
  for (int i = 0; i < n; i++) {
    const float pb[3] = {body[i].x, body[i].y, body[i].z};
    float* pw = &sc.world[3 * size_t(i)];
    point_body_to_world(st, pb, pw);
    float* near = &sc.nearest[15 * size_t(i)];
    if (search) {
      KPoint np[NUM_MATCH_POINTS];
      float nd[NUM_MATCH_POINTS];
      int found = tree.nearest_search(pw, NUM_MATCH_POINTS, np, nd, prm.max_dist);
      sc.nearest_n[i] = found;
      for (int k = 0; k < found; k++) {
        near[3 * k] = np[k].x; near[3 * k + 1] = np[k].y; near[3 * k + 2] = np[k].z;
        sc.nearest_d2[5 * size_t(i) + k] = nd[k];
      }
      if (found < NUM_MATCH_POINTS) sc.selected[i] = 0;
      else sc.selected[i] = !(nd[NUM_MATCH_POINTS - 1] > 5);
    }
    sc.res_last[i] = -1000.0f;
    if (!sc.selected[i] || sc.nearest_n[i] < NUM_MATCH_POINTS) {
      sc.selected[i] = 0;
      continue;
    }
    sc.selected[i] = 0;
    double pabcd[4] = {0, 0, 0, 0};
    if (esti_plane(pabcd, near, prm.plane_threshold)) {
      float pd2 = pabcd[0] * pw[0] + pabcd[1] * pw[1] + pabcd[2] * pw[2] + pabcd[3];
      V3 p_body(pb[0], pb[1], pb[2]);
      float s = 1 - 0.9 * std::fabs(pd2) / std::sqrt(norm(p_body));
      if (s > 0.9) {
        sc.selected[i] = 1;
        sc.normvec[4 * size_t(i) + 0] = float(pabcd[0]);
        sc.normvec[4 * size_t(i) + 1] = float(pabcd[1]);
        sc.normvec[4 * size_t(i) + 2] = float(pabcd[2]);
        sc.normvec[4 * size_t(i) + 3] = pd2;
        sc.res_last[i] = std::fabs(pd2);
      }
    }
    for (int k = 0; k < 4; k++) sc.pabcd[4 * size_t(i) + k] = pabcd[k];
  }
}

// Jacobian row of one effective point (:1035-1071).  h[12], z.
inline void jacobian_row(const State& st, const float* pb, const float* nv4, int imu_en, double h[12], double& z) {
  V3 pL(pb[0], pb[1], pb[2]);
  V3 p_this = st.offset_R_L_I * pL + st.offset_T_L_I;
  M3 cm = skew(p_this);
  V3 nvec(nv4[0], nv4[1], nv4[2]);
  M3 Rt = transpose(st.rot_end);
  // (Eigen evaluates a chain of products left to right, every inner product into a temporary: (cm * Rt) * nvec, :1059 - the
  // association is part of the rounding, and in LIO mode the normal matrix amplifies a last-bit difference of a row to ~1e-8 in
  // the extrinsic states: tests/test_oracle_iekf_pinned.py holds this function to the reference's text)
  V3 A = (cm * Rt) * nvec;
  h[0] = A.x; h[1] = A.y; h[2] = A.z;
  h[3] = nv4[0]; h[4] = nv4[1]; h[5] = nv4[2];
  if (imu_en) {
    V3 H_R_LI = ((skew(pL) * transpose(st.offset_R_L_I)) * Rt) * nvec;  // :1055-1056, left to right
    V3 H_T_LI = Rt * nvec;
    h[6] = H_R_LI.x; h[7] = H_R_LI.y; h[8] = H_R_LI.z;
    h[9] = H_T_LI.x; h[10] = H_T_LI.y; h[11] = H_T_LI.z;
  } else {
    for (int k = 6; k < 12; k++) h[k] = 0;
  }
  z = -double(nv4[3]);
}

// The full per-scan update.  `st` is state (in/out), `st_prop` = state_propagat.
// Returns the number of iterations executed; logs[it] is filled for each.
inline int iekf_update(const KdTree& tree, const std::vector<P4>& body, State& st, const State& st_prop,
                       const IekfParams& prm, IekfScratch& sc, std::vector<IekfIterLog>& logs) {
  const int n = int(body.size());
  sc.world.assign(3 * size_t(n), 0.f);
  sc.nearest.resize(15 * size_t(n));
  sc.nearest_d2.resize(5 * size_t(n));
  sc.nearest_n.assign(n, 0);
  sc.selected.assign(n, 1);
  sc.normvec.assign(4 * size_t(n), 0.f);
  sc.pabcd.assign(4 * size_t(n), 0.0);
  sc.res_last.assign(n, -1000.f);
  logs.clear();
  int rematch_num = 0;
  bool search = true;
  bool stop = false;
  Mat HTH24(DIM_STATE, DIM_STATE);
  for (int it = 0; it < prm.max_iterations; it++) {
    residual_pass(tree, body, st, search, prm, sc);
    IekfIterLog lg;
    lg.searched = search ? 1 : 0;
    // compaction + rows
    std::vector<double> H;  // m x 12
    std::vector<double> zv;
    for (int i = 0; i < n; i++)
      if (sc.selected[i]) {
        double h[12], z;
        const float pb[3] = {body[i].x, body[i].y, body[i].z};
        jacobian_row(st, pb, &sc.normvec[4 * size_t(i)], prm.imu_en, h, z);
        H.insert(H.end(), h, h + 12);
        zv.push_back(z);
      }
    const int m = int(zv.size());
    lg.effect_num = m;
    const double Rinv = prm.laser_point_cov_inv;
    double G12[12][12] = {{0}}, g12[12] = {0};
    for (int p = 0; p < m; p++) {
      const double* h = &H[12 * size_t(p)];
      for (int i = 0; i < 12; i++) {
        double hi = h[i] * Rinv;
        g12[i] += hi * zv[p];
        for (int j = 0; j < 12; j++) G12[i][j] += hi * h[j];
      }
    }
    int t = 0;
    for (int i = 0; i < 12; i++)
      for (int j = i; j < 12; j++) lg.HTH[t++] = G12[i][j];
    for (int i = 0; i < 12; i++) lg.HTz[i] = g12[i];
    for (int i = 0; i < 12; i++)
      for (int j = 0; j < 12; j++) HTH24(i, j) = G12[i][j];
    Mat sum = inverse(st.cov);
    for (int i = 0; i < DIM_STATE; i++)
      for (int j = 0; j < DIM_STATE; j++) sum(i, j) += HTH24(i, j);
    Mat K1 = inverse(sum);
    double vec[24];
    boxminus(st_prop, st, vec);
    double sol[24];
    Mat KH(DIM_STATE, 12);  // K * Hsub
    if (prm.literal_gain) {
      // K = K1[:, :12] * Hsub^T R^-1  (24 x m), then K*z and K*Hsub as the reference forms them
      Mat K(DIM_STATE, m);
      for (int r = 0; r < DIM_STATE; r++)
        for (int p = 0; p < m; p++) {
          double s = 0;
          for (int c = 0; c < 12; c++) s += K1(r, c) * (H[12 * size_t(p) + c] * Rinv);
          K(r, p) = s;
        }
      for (int r = 0; r < DIM_STATE; r++) {
        double kz = 0;
        for (int p = 0; p < m; p++) kz += K(r, p) * zv[p];
        for (int c = 0; c < 12; c++) {
          double s = 0;
          for (int p = 0; p < m; p++) s += K(r, p) * H[12 * size_t(p) + c];
          KH(r, c) = s;
        }
        double khv = 0;
        for (int c = 0; c < 12; c++) khv += KH(r, c) * vec[c];
        sol[r] = kz + vec[r] - khv;
      }
    } else {
      for (int r = 0; r < DIM_STATE; r++) {
        double kz = 0;
        for (int c = 0; c < 12; c++) kz += K1(r, c) * g12[c];
        for (int c = 0; c < 12; c++) {
          double s = 0;
          for (int k = 0; k < 12; k++) s += K1(r, k) * G12[k][c];
          KH(r, c) = s;
        }
        double khv = 0;
        for (int c = 0; c < 12; c++) khv += KH(r, c) * vec[c];
        sol[r] = kz + vec[r] - khv;
      }
    }
    boxplus(st, sol);
    for (int i = 0; i < 24; i++) lg.solution[i] = sol[i];
    logs.push_back(lg);
    double rn = std::sqrt(sol[0] * sol[0] + sol[1] * sol[1] + sol[2] * sol[2]);
    double tn = std::sqrt(sol[3] * sol[3] + sol[4] * sol[4] + sol[5] * sol[5]);
    bool converged = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);
    search = false;
    if (converged || ((rematch_num == 0) && (it == (prm.max_iterations - 2)))) {
      search = true;
      rematch_num++;
    }
    if (!stop && (rematch_num >= 2 || (it == prm.max_iterations - 1))) {
      // cov = (I - G) cov, G[:, :12] = K Hsub
      Mat IG = Mat::identity(DIM_STATE);
      for (int r = 0; r < DIM_STATE; r++)
        for (int c = 0; c < 12; c++) IG(r, c) -= KH(r, c);
      st.cov = matmul(IG, st.cov);
      stop = true;
    }
    if (stop) break;
  }
  return int(logs.size());
}

// map_incremental (:516-559).  Splits the down-sampled world points into PointToAdd
// (inserted with voxel down-sampling) and PointNoNeedDownsample, then applies both to the tree.
inline void map_incremental(KdTree& tree, const std::vector<P4>& body, const State& st, const IekfScratch& sc,
                            float filter_size_map, std::vector<KPoint>& to_add, std::vector<KPoint>& no_down,
                            bool apply) {
  const int n = int(body.size());
  to_add.clear();
  no_down.clear();
  const float fs = filter_size_map;
  for (int i = 0; i < n; i++) {
    const float pb[3] = {body[i].x, body[i].y, body[i].z};
    float pw[3];
    point_body_to_world(st, pb, pw);
    KPoint wp{pw[0], pw[1], pw[2], i};
    const int nn = sc.nearest_n.empty() ? 0 : sc.nearest_n[i];
    if (nn > 0) {
      const float* near = &sc.nearest[15 * size_t(i)];
      bool need_add = true;
      // NOTE: filter_size_map_min is a double in the reference; the expression is evaluated in
      // double and stored to the float fields of mid_point (:529-534).
      const double fsd = double(fs);
      float mid[3];
      for (int a = 0; a < 3; a++) mid[a] = float(std::floor(pw[a] / fsd) * fsd + 0.5 * fsd);
      float dist = (pw[0] - mid[0]) * (pw[0] - mid[0]) + (pw[1] - mid[1]) * (pw[1] - mid[1]) +
                   (pw[2] - mid[2]) * (pw[2] - mid[2]);
      if (std::fabs(near[0] - mid[0]) > 0.5 * fsd && std::fabs(near[1] - mid[1]) > 0.5 * fsd &&
          std::fabs(near[2] - mid[2]) > 0.5 * fsd) {
        no_down.push_back(wp);
        continue;
      }
      for (int k = 0; k < NUM_MATCH_POINTS; k++) {
        if (nn < NUM_MATCH_POINTS) break;
        float dk = (near[3 * k] - mid[0]) * (near[3 * k] - mid[0]) + (near[3 * k + 1] - mid[1]) * (near[3 * k + 1] - mid[1]) +
                   (near[3 * k + 2] - mid[2]) * (near[3 * k + 2] - mid[2]);
        if (dk < dist) { need_add = false; break; }
      }
      if (need_add) to_add.push_back(wp);
    } else {
      to_add.push_back(wp);
    }
  }
  if (apply) {
    tree.add_points(to_add, true);
    tree.add_points(no_down, false);
  }
}

}  // namespace orc
