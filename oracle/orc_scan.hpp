// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).
//
// Per-scan point processing that precedes registration:
//   undistort_imu .... ImuProcess::propagation_and_undist, back-propagation loop
//                      reference src/IMU_Processing.hpp:287 (time sort), :390-414 (loop; quirk A3)
//   undistort_cv ..... ImuProcess::Forward_propagation_without_imu, CV de-skew
//                      reference src/IMU_Processing.hpp:209 (time sort), :246-266 (loop; quirk A3)
//   voxel_grid ....... pcl::VoxelGrid<PointXYZINormal>::filter as called at
//                      reference src/laserMapping.cpp:917-919.  PCL is third-party and NOT vendored
//                      (README.md:57 pins PCL >= 1.8); its documented algorithm (voxel_grid.hpp,
//                      PCL 1.8: getMinMax3D, int64 overflow guard -> identity copy, linear voxel index,
//                      sort by index, per-voxel centroid of all fields, ascending-index output) is
//                      restated here.  PARITY UNPINNED for this function (no PCL source, no vectors); held bit for
//                      bit to an independently written numpy version in tests/test_oracle_scan_independent.py
//                      (which also checks the two de-skew loops against per-point numpy / scipy formulations).
// Points are carried as float4 records (x, y, z, t) where t is the reference's `curvature` field:
// the per-point time offset from the scan start in MILLISECONDS (src/preprocess.cpp).
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "orc_math.hpp"

namespace orc {

struct P4 {
  float x, y, z, t;
};

// Pose6D — reference msg/Pose6D.msg:1-7, filled by set_pose6d (include/common_lib.h:183-199).
// 22 doubles: offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9] (row-major).
struct Pose6D {
  double offset_time;
  double acc[3], gyr[3], vel[3], pos[3], rot[9];
};
static_assert(sizeof(Pose6D) == 22 * 8, "Pose6D must be 22 doubles");

// Stable time sort — the reference uses std::sort (unstable) with time_list
// (src/IMU_Processing.hpp:33); ties are resolved here by original index so results are reproducible.
inline void sort_by_time(std::vector<P4>& pts) {
  std::stable_sort(pts.begin(), pts.end(), [](const P4& a, const P4& b) { return a.t < b.t; });
}

// IMU-mode back-propagation.  `pts` must already be time-sorted (ascending).
// end_R/end_p: state.rot_end/pos_end AFTER forward propagation; R_LI/T_LI: extrinsic.
inline void undistort_imu(std::vector<P4>& pts, const std::vector<Pose6D>& imupose, const M3& end_R, const V3& end_p,
                          const M3& R_LI, const V3& T_LI) {
  if (pts.empty() || imupose.size() < 2) return;
  const M3 R_LI_t = transpose(R_LI), end_R_t = transpose(end_R);
  long it_pcl = long(pts.size()) - 1;
  for (long kp = long(imupose.size()) - 1; kp != 0; kp--) {
    const Pose6D& head = imupose[kp - 1];
    M3 R_imu = M3::from(head.rot);
    V3 acc_imu(head.acc[0], head.acc[1], head.acc[2]);
    V3 vel_imu(head.vel[0], head.vel[1], head.vel[2]);
    V3 pos_imu(head.pos[0], head.pos[1], head.pos[2]);
    V3 angvel(head.gyr[0], head.gyr[1], head.gyr[2]);
    for (; pts[it_pcl].t / double(1000) > head.offset_time; it_pcl--) {
      double dt = pts[it_pcl].t / double(1000) - head.offset_time;
      M3 R_i = R_imu * Exp(angvel, dt);
      V3 P_i = pos_imu + vel_imu * dt + 0.5 * acc_imu * dt * dt;
      V3 p_in(pts[it_pcl].x, pts[it_pcl].y, pts[it_pcl].z);
      V3 Pc = R_LI_t * (end_R_t * (R_i * (R_LI * p_in + T_LI) + P_i - end_p) - T_LI);
      pts[it_pcl].x = float(Pc.x);
      pts[it_pcl].y = float(Pc.y);
      pts[it_pcl].z = float(Pc.z);
      if (it_pcl == 0) break;
    }
  }
}

// CV-mode de-skew.  `pts` time-sorted.  omega = state.bias_g, vel = state.vel_end (CV slots,
// IMU_Processing.hpp:226-231), end_R = state.rot_end after propagation.
inline void undistort_cv(std::vector<P4>& pts, const V3& omega, const V3& vel, const M3& end_R) {
  if (pts.empty()) return;
  const double end_off = pts.back().t / double(1000);
  const M3 end_R_t = transpose(end_R);
  for (long it = long(pts.size()) - 1; it != 0; it--) {
    double dt_j = end_off - pts[it].t / double(1000);
    M3 R_jk = Exp(omega, -dt_j);
    V3 P_j(pts[it].x, pts[it].y, pts[it].z);
    V3 p_jk = -1.0 * (end_R_t * vel) * dt_j;
    V3 Pc = R_jk * P_j + p_jk;
    pts[it].x = float(Pc.x);
    pts[it].y = float(Pc.y);
    pts[it].z = float(Pc.z);
  }
}

// pcl::VoxelGrid restatement (see header).  Returns false when PCL's int32 index-overflow guard
// triggers (output = input).  Within a voxel, points are accumulated in ascending input order
// (PCL's std::sort is unstable, so its order is unspecified; this fixes one).
inline bool voxel_grid(const std::vector<P4>& in, float leaf, std::vector<P4>& out) {
  out.clear();
  if (in.empty()) return true;
  const float inv = 1.0f / leaf;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (const P4& p : in) {
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    mn[0] = std::min(mn[0], p.x); mx[0] = std::max(mx[0], p.x);
    mn[1] = std::min(mn[1], p.y); mx[1] = std::max(mx[1], p.y);
    mn[2] = std::min(mn[2], p.z); mx[2] = std::max(mx[2], p.z);
  }
  int64_t dx = int64_t((mx[0] - mn[0]) * inv) + 1;
  int64_t dy = int64_t((mx[1] - mn[1]) * inv) + 1;
  int64_t dz = int64_t((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > int64_t(std::numeric_limits<int32_t>::max())) {
    out = in;
    return false;
  }
  int min_b[3], max_b[3], div_b[3];
  for (int a = 0; a < 3; a++) {
    min_b[a] = int(std::floor(mn[a] * inv));
    max_b[a] = int(std::floor(mx[a] * inv));
    div_b[a] = max_b[a] - min_b[a] + 1;
  }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  std::vector<std::pair<int, int>> idx;  // (voxel index, point index)
  idx.reserve(in.size());
  for (int i = 0; i < int(in.size()); i++) {
    const P4& p = in[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    int i0 = int(std::floor(p.x * inv) - float(min_b[0]));
    int i1 = int(std::floor(p.y * inv) - float(min_b[1]));
    int i2 = int(std::floor(p.z * inv) - float(min_b[2]));
    idx.emplace_back(i0 * mul[0] + i1 * mul[1] + i2 * mul[2], i);
  }
  std::stable_sort(idx.begin(), idx.end(),
                   [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
  size_t i = 0;
  while (i < idx.size()) {
    size_t j = i;
    float sx = 0, sy = 0, sz = 0, st = 0;
    while (j < idx.size() && idx[j].first == idx[i].first) {
      const P4& p = in[idx[j].second];
      sx += p.x; sy += p.y; sz += p.z; st += p.t;
      j++;
    }
    float n = float(j - i);
    out.push_back(P4{sx / n, sy / n, sz / n, st / n});
    i = j;
  }
  return true;
}

}  // namespace orc
