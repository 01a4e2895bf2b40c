// C entry points around the UNMODIFIED reference header src/IMU_Processing.hpp (ImuProcess: IMU forward propagation + the
// backward-propagation de-skew, and the constant-velocity variant), compiled where it lies under /root/reference against
// oracle/ref_shim_imu + oracle/ref_shim_math (plain-struct stand-ins for the ROS messages and pcl::PointCloud, a minimal
// fixed-size matrix standing in for Eigen) into oracle/_ref/libref_imu.so.  Test infrastructure:
// tests/test_oracle_deskew_pinned.py holds oracle/orc_scan.hpp (the de-skew loops) bit for bit, and the numpy IMU forward
// propagation of harness/lio_harness.py to rounding, to these functions.  Nothing of the reference is copied here: the code
// below fills the reference's own structures from flat arrays, calls ImuProcess::Process and reads the results back.
// (`private` is lifted for this translation unit only: ImuProcess keeps the hand-over between consecutive scans - the last IMU
// sample, the previous scan's end time - in private members that its constructor leaves uninitialised.)
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <csignal>
#include <deque>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
// everything IMU_Processing.hpp includes that has include guards comes in first, so that the access change below touches
// nothing but the class ImuProcess itself
#include <ros/ros.h>
#include <so3_math.h>
#include <Eigen/Eigen>
#include <common_lib.h>
#define private public
#include <IMU_Processing.hpp>
#undef private

namespace {
StatesGroup from_pod(const double* p) {
  StatesGroup s;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { s.rot_end(i, j) = p[3 * i + j]; s.offset_R_L_I(i, j) = p[12 + 3 * i + j]; }
  for (int i = 0; i < 3; i++) { s.pos_end(i) = p[9 + i]; s.offset_T_L_I(i) = p[21 + i]; s.vel_end(i) = p[24 + i]; s.bias_g(i) = p[27 + i]; s.bias_a(i) = p[30 + i]; s.gravity(i) = p[33 + i]; }
  for (int i = 0; i < DIM_STATE; i++) for (int j = 0; j < DIM_STATE; j++) s.cov(i, j) = p[36 + DIM_STATE * i + j];
  return s;
}
void to_pod(const StatesGroup& s, double* p) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { p[3 * i + j] = s.rot_end(i, j); p[12 + 3 * i + j] = s.offset_R_L_I(i, j); }
  for (int i = 0; i < 3; i++) { p[9 + i] = s.pos_end(i); p[21 + i] = s.offset_T_L_I(i); p[24 + i] = s.vel_end(i); p[27 + i] = s.bias_g(i); p[30 + i] = s.bias_a(i); p[33 + i] = s.gravity(i); }
  for (int i = 0; i < DIM_STATE; i++) for (int j = 0; j < DIM_STATE; j++) p[36 + DIM_STATE * i + j] = s.cov(i, j);
}
sensor_msgs::Imu::ConstPtr imu_of(const double* r) {  // t, gyr[3], acc[3]
  auto m = std::make_shared<sensor_msgs::Imu>();
  m->header.stamp.sec = r[0];
  m->angular_velocity.x = r[1]; m->angular_velocity.y = r[2]; m->angular_velocity.z = r[3];
  m->linear_acceleration.x = r[4]; m->linear_acceleration.y = r[5]; m->linear_acceleration.z = r[6];
  return m;
}
void fill_cloud(PointCloudXYZI& c, const float* pts, int n) {  // x, y, z, curvature [ms]
  c.points.resize(n);
  for (int i = 0; i < n; i++) {
    PointType p{};
    p.x = pts[4 * i]; p.y = pts[4 * i + 1]; p.z = pts[4 * i + 2]; p.curvature = pts[4 * i + 3];
    c.points[i] = p;
  }
}
void read_cloud(const PointCloudXYZI& c, float* out) {
  for (size_t i = 0; i < c.points.size(); i++) {
    out[4 * i] = c.points[i].x; out[4 * i + 1] = c.points[i].y; out[4 * i + 2] = c.points[i].z; out[4 * i + 3] = c.points[i].curvature;
  }
}
}  // namespace

extern "C" {
// One LIO-mode call of ImuProcess::Process (propagation_and_undist, src/IMU_Processing.hpp:271-417) with the hand-over from the
// previous scan given explicitly.  imu: n_imu rows (t, gyr, acc) of THIS scan; last_imu: the previous scan's last sample;
// carry: acc_s_last[3], angvel_last[3]; cov6: cov_gyr[3], cov_acc[3] (the *_scale values Process() installs);
// state: lii_state POD, in: the state after the previous update, out: the propagated state;  pts: n_pts x (x, y, z, t_ms), out: the
// de-skewed cloud IN THE REFERENCE'S ORDER (sorted by time);  poses_out: up to max_poses rows of 22 doubles (Pose6D), *n_poses.
int ref_imu_process_lio(const double* imu, int n_imu, const double* last_imu, double last_lidar_end_time, const double* carry,
                        const double* cov6, double mean_acc_norm, double lidar_beg_time, int lidar_type, double* state, float* pts,
                        int n_pts, double* poses_out, int max_poses, int* n_poses, double* carry_out) {
  mkdir("/tmp/lii_ref_imu", 0700);
  ImuProcess p;
  p.imu_en = true;
  p.LI_init_done = true;
  p.lidar_type = lidar_type;
  p.imu_need_init_ = false;
  p.b_first_frame_ = false;
  p.IMU_mean_acc_norm = mean_acc_norm;
  p.cov_gyr = V3D(cov6[0], cov6[1], cov6[2]);
  p.cov_acc = V3D(cov6[3], cov6[4], cov6[5]);
  p.last_imu_ = imu_of(last_imu);
  p.last_lidar_end_time_ = last_lidar_end_time;
  p.acc_s_last = V3D(carry[0], carry[1], carry[2]);
  p.angvel_last = V3D(carry[3], carry[4], carry[5]);
  MeasureGroup meas;
  meas.lidar_beg_time = lidar_beg_time;
  fill_cloud(*meas.lidar, pts, n_pts);
  for (int i = 0; i < n_imu; i++) meas.imu.push_back(imu_of(imu + 7 * i));
  StatesGroup s = from_pod(state);
  PointCloudXYZI::Ptr un(new PointCloudXYZI());
  p.Process(meas, s, un);
  to_pod(s, state);
  if ((int)un->points.size() != n_pts) return -1;
  read_cloud(*un, pts);
  *n_poses = (int)p.IMUpose.size();
  for (int k = 0; k < *n_poses && k < max_poses; k++) {
    const Pose6D& q = p.IMUpose[k];
    double* o = poses_out + 22 * k;
    o[0] = q.offset_time;
    for (int i = 0; i < 3; i++) { o[1 + i] = q.acc[i]; o[4 + i] = q.gyr[i]; o[7 + i] = q.vel[i]; o[10 + i] = q.pos[i]; }
    for (int i = 0; i < 9; i++) o[13 + i] = q.rot[i];
  }
  if (carry_out) {
    for (int i = 0; i < 3; i++) { carry_out[i] = p.acc_s_last(i); carry_out[3 + i] = p.angvel_last(i); }
    carry_out[6] = p.last_lidar_end_time_;
  }
  return 0;
}
// One LO-mode call (imu_en = false: Forward_propagation_without_imu, :207-269).  first_frame != 0: dt = 0.1 as for the very first
// scan; else dt = lidar_beg_time - time_last_scan.  cov6: cov_gyr_scale[3], cov_acc_scale[3].
int ref_imu_process_cv(double lidar_beg_time, double time_last_scan, int first_frame, const double* cov6, int lidar_type,
                       double* state, float* pts, int n_pts) {
  mkdir("/tmp/lii_ref_imu", 0700);
  ImuProcess p;
  p.imu_en = false;
  p.lidar_type = lidar_type;
  p.b_first_frame_ = first_frame != 0;
  p.time_last_scan = time_last_scan;
  p.cov_gyr_scale = V3D(cov6[0], cov6[1], cov6[2]);
  p.cov_acc_scale = V3D(cov6[3], cov6[4], cov6[5]);
  MeasureGroup meas;
  meas.lidar_beg_time = lidar_beg_time;
  fill_cloud(*meas.lidar, pts, n_pts);
  StatesGroup s = from_pod(state);
  PointCloudXYZI::Ptr un(new PointCloudXYZI());
  p.Process(meas, s, un);
  to_pod(s, state);
  if ((int)un->points.size() != n_pts) return -1;
  read_cloud(*un, pts);
  return 0;
}
}
