// A ROS-free `laserMapping` over the C-ABI, in the reference's host language (BASELINE.json north_star: "the C++ host (ROS
// callbacks, IMU_Processing state propagation, ikd-Tree incremental insert) calls HIP through a thin C-ABI").
//
// What the reference's node does between its subscribers and its result file (src/laserMapping.cpp), with every piece of the
// accelerated path behind include/liinit_hip.h and everything else - buffers, synchronisation, IMU / constant-velocity forward
// propagation, the LO -> LI-Init -> LIO hand-over, the result file - as plain C++ here:
//   lii_replay_imu / _pcl2 / _livox   imu_cbk :395-433, standard_pcl_cbk :345-379, livox_pcl_cbk :310-343 (driver messages as
//                                     byte arrays; the clouds are decoded, filtered, time-sorted and cut on the device:
//                                     lii_ingest_*)
//   sync()                            sync_packages :436-480
//   cv_propagate / imu_propagate      ImuProcess::Forward_propagation_without_imu, src/IMU_Processing.hpp:204-244, and the forward
//                                     part of propagation_and_undist, :296-382 (checked against the unmodified header compiled
//                                     as oracle/_ref/libref_imu.so: tests/test_replay_host.py)
//   process()                         the body of the main loop :895-1234: de-skew + voxel grid + iterated update
//                                     (lii_scan_register), map_incremental (lii_map_incremental), movement detection :1151-1155,
//                                     push_Lidar_CalibState + data_sufficiency_assess :1169-1177 (lii_data_sufficiency),
//                                     LI_Initialization :1179 (lii_li_init_interpolate + lii_li_init_run), the switch to LIO
//                                     :1183-1212, the refinement result :1164-1178
//   write_result()                    fileout_calib_result :708-725 -> result/Initialization_result.txt in the reference's format
// Parameters come from the reference's own launch + yaml files through lii_params_load_launch.
// Harness code, not product: it only includes include/liinit_hip.h.  Two deliberate differences from the node, both stated where
// they occur: ImuProcess::time_last_scan is initialised (the reference reads it uninitialised on its second frame), and nothing
// is published, plotted or logged beyond the odometry rows the tests read.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "liinit_hip.h"

namespace {

constexpr double kG = 9.81;  // G_m_s2, include/common_lib.h:25

// ---- 3 x 3 helpers (row-major), evaluation order of the reference's Eigen expressions
void m3_mul(const double* A, const double* B, double* C) {
  double t[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) t[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
  std::memcpy(C, t, sizeof(t));
}
void m3_vec(const double* A, const double* v, double* o) {
  double t[3];
  for (int r = 0; r < 3; r++) t[r] = A[3 * r] * v[0] + A[3 * r + 1] * v[1] + A[3 * r + 2] * v[2];
  std::memcpy(o, t, sizeof(t));
}
void m3_t(const double* A, double* T) {
  double t[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) t[3 * r + c] = A[3 * c + r];
  std::memcpy(T, t, sizeof(t));
}
// Exp(ang_vel, dt) - include/so3_math.h:37-59: identity below 1e-7, `(1 - cos) * K * K` = ((1 - cos) K) K
void so3_exp(const double w[3], double dt, double R[9]) {
  const double n = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  for (int e = 0; e < 9; e++) R[e] = (e % 4 == 0) ? 1.0 : 0.0;
  if (n > 0.0000001) {
    const double ax[3] = {w[0] / n, w[1] / n, w[2] / n};
    const double K[9] = {0.0, -ax[2], ax[1], ax[2], 0.0, -ax[0], -ax[1], ax[0], 0.0};
    const double ang = n * dt, s = std::sin(ang), c1 = 1.0 - std::cos(ang);
    double cK[9], KK[9];
    for (int e = 0; e < 9; e++) cK[e] = c1 * K[e];
    m3_mul(cK, K, KK);
    for (int e = 0; e < 9; e++) R[e] = (R[e] + s * K[e]) + KK[e];
  }
}
// RotMtoEuler - include/so3_math.h:109-129
void rot_to_euler(const double* R, double e[3]) {
  const double sy = std::sqrt(R[0] * R[0] + R[3] * R[3]);
  if (sy >= 1e-6) {
    e[0] = std::atan2(R[7], R[8]); e[1] = std::atan2(-R[6], sy); e[2] = std::atan2(R[3], R[0]);
  } else {
    e[0] = std::atan2(-R[5], R[4]); e[1] = std::atan2(-R[6], sy); e[2] = 0.0;
  }
}
// cov <- F cov F^T + Q with F = I + (sparse blocks): dense 24 x 24 products as Eigen forms them (F * cov, then * F^T)
void cov_propagate(double* P, const double* F, const double* Q) {
  static thread_local double T[24 * 24], U[24 * 24];
  for (int r = 0; r < 24; r++)
    for (int c = 0; c < 24; c++) {
      double s = 0;
      for (int k = 0; k < 24; k++) s += F[24 * r + k] * P[24 * k + c];
      T[24 * r + c] = s;
    }
  for (int r = 0; r < 24; r++)
    for (int c = 0; c < 24; c++) {
      double s = 0;
      for (int k = 0; k < 24; k++) s += T[24 * r + k] * F[24 * c + k];
      U[24 * r + c] = s + Q[24 * r + c];
    }
  std::memcpy(P, U, sizeof(U));
}

struct ImuMsg { double t, gyr[3], acc[3]; };
struct LidarMsg {
  double stamp;
  int kind;  // 0: sensor_msgs/PointCloud2, 1: livox CustomMsg
  int32_t n_points, scan_count;
  lii_pc2_fields f2;
  lii_livox_fields fl;
  std::vector<unsigned char> data;
};

}  // namespace

extern "C" {

typedef struct lii_replay_config {
  uint32_t struct_size;
  int32_t device, max_scan_points, max_map_points;
  const char* launch_file;  // reference-format launch file (its rosparam yaml is looked up in config_dir, else <launch dir>/../config)
  const char* config_dir;   // may be NULL
  const char* result_path;  // result/Initialization_result.txt of this run; may be NULL (nothing written)
  int32_t stop_after_init;  // 1: stop feeding the loop once the initialization result exists (tests)
  int32_t reserved0;
} lii_replay_config;

typedef struct lii_replay_status {
  uint32_t struct_size;
  int32_t imu_en, data_accum_start, data_accum_finished, refine_done;
  int32_t scans_processed, frames_pending, imu_pending, cut_frame_num;
  double move_start_time, time_lag_imu_wrt_lidar, timediff_imu_wrt_lidar, mean_acc_norm, lidar_end_time;
  lii_state state;
  lii_calib_result init;  // LI_Initialization's result (valid once data_accum_finished)
  double init_time_lag_1, init_total_time_lag;
} lii_replay_status;

// one row per processed scan: [0] lidar_end_time, [1] imu_en, [2] iterations, [3] effect_num, [4 .. 39] the state without its covariance
enum { LII_REPLAY_ROW = 40 };

struct lii_replay {
  lii_handle h = nullptr;
  lii_params prm{};
  lii_ingest_opts ing{};
  lii_iekf_opts opts{};
  float leaf = 0.f;
  std::string result_path;
  bool stop_after_init = false;
  std::string err;

  // ---- buffers of the callbacks (src/laserMapping.cpp:102-150 globals)
  std::deque<ImuMsg> imu_buffer;
  std::deque<LidarMsg> lidar_msgs;          // driver messages not yet cut (the reference cuts in the callback; the device holds the
                                            // frames of ONE message, so a message is cut when it reaches the head of the queue)
  std::vector<lii_frame_info> frames;       // sub-frames of the message at the head (lidar_buffer + time_buffer)
  size_t frame_next = 0;
  double last_timestamp_lidar = -1.0, last_timestamp_imu = -1.0;
  double timediff_imu_wrt_lidar = 0.0, time_lag_IMU_wtr_lidar = 0.0;
  bool timediff_set_flg = false;
  int scan_count = 0;
  // imu_cbk's running mean of the accelerometer norm (:400-415)
  int imu_cnt = 0;
  double mean_acc[3] = {0, 0, 0};
  double mean_acc_norm = kG;
  // ---- main loop
  bool lidar_pushed = false, have_scan = false;
  double lidar_beg_time = 0, lidar_end_time = 0;
  std::vector<ImuMsg> meas_imu;
  lii_state state{}, state_propagat{};
  bool map_built = false, imu_en = false, data_accum_start = false, data_accum_finished = false, refine_done = false;
  double move_start_time = 0, online_calib_starts_time = 0;
  int frame_num = 0, cut_frame_num = 1;
  std::vector<lii_calib_state> imu_all, lidar_states;  // Init_LI: IMU_state_group_ALL, Lidar_state_group
  std::vector<double> omg;                              // LiDAR angular velocity of every LO frame (data_sufficiency_assess)
  lii_calib_result init{};
  double init_lag1 = 0, init_total = 0;
  // ---- ImuProcess
  bool b_first_frame = true, imu_need_init = true, li_init_done = false;
  double time_last_scan = 0, last_lidar_end_time_ = 0;
  ImuMsg last_imu{};
  double angvel_last[3] = {0, 0, 0}, acc_s_last[3] = {0, 0, 0};
  double cov_gyr[3], cov_acc[3], cov_bias_gyr[3], cov_bias_acc[3], cov_R_LI[3], cov_T_LI[3];
  double imu_mean_acc_norm = kG;
  std::vector<lii_pose6d> imu_pose;
  std::vector<double> log;  // LII_REPLAY_ROW doubles per processed scan
};

static int fail(lii_replay* r, int code, const std::string& msg) {
  if (r) r->err = msg;
  return code;
}

// StatesGroup() - include/common_lib.h:70-81: identity rotations, zero vectors (gravity included), cov = I * INIT_COV (1) with the
// velocity / bias / gravity block set to 1e-5 I
static void state_init(lii_state* s) {
  std::memset(s, 0, sizeof(*s));
  for (int e = 0; e < 3; e++) { s->rot_end[4 * e] = 1.0; s->offset_R_L_I[4 * e] = 1.0; }
  for (int e = 0; e < 24; e++) s->cov[25 * e] = e < 15 ? 1.0 : 0.00001;
}

int lii_replay_create(const lii_replay_config* cfg, lii_replay** out) {
  if (!cfg || !out || cfg->struct_size != sizeof(lii_replay_config) || !cfg->launch_file) return LII_ERR_INVALID;
  auto* r = new lii_replay();
  int rc = lii_params_defaults(&r->prm);
  if (rc == LII_OK) rc = lii_params_load_launch(cfg->launch_file, cfg->config_dir, &r->prm);
  lii_config lc{};
  if (rc == LII_OK) rc = lii_params_apply(&r->prm, cfg->device, cfg->max_scan_points, cfg->max_map_points, &lc, &r->ing, &r->opts, &r->leaf);
  if (rc == LII_OK) rc = lii_create(&lc, &r->h);
  if (rc != LII_OK) { delete r; return rc; }
  if (cfg->result_path) r->result_path = cfg->result_path;
  r->stop_after_init = cfg->stop_after_init != 0;
  r->cut_frame_num = r->prm.cut_frame ? r->prm.cut_frame_num : 1;
  r->mean_acc_norm = r->prm.mean_acc_norm;
  state_init(&r->state);
  // main() :851-867: the IMU processor's noise scales come from the parameter file; the extrinsic covariance from Rot_LI_cov / Trans_LI_cov
  for (int a = 0; a < 3; a++) {
    r->cov_gyr[a] = r->prm.gyr_cov; r->cov_acc[a] = r->prm.acc_cov;
    r->cov_bias_gyr[a] = r->prm.b_gyr_cov; r->cov_bias_acc[a] = r->prm.b_acc_cov;
    r->cov_R_LI[a] = r->prm.n_Rot_LI_cov > a ? r->prm.Rot_LI_cov[a] : 0.00001;
    r->cov_T_LI[a] = r->prm.n_Trans_LI_cov > a ? r->prm.Trans_LI_cov[a] : 0.00001;
  }
  *out = r;
  return LII_OK;
}
void lii_replay_destroy(lii_replay* r) {
  if (!r) return;
  if (r->h) lii_destroy(r->h);
  delete r;
}
const char* lii_replay_last_error(lii_replay* r) { return r ? r->err.c_str() : "null handle"; }
lii_handle lii_replay_handle(lii_replay* r) { return r ? r->h : nullptr; }

// imu_cbk, src/laserMapping.cpp:395-433
int lii_replay_imu(lii_replay* r, double stamp, const double gyr[3], const double acc[3]) {
  if (!r || !gyr || !acc) return LII_ERR_INVALID;
  if (r->imu_cnt < 100) {
    r->imu_cnt++;
    for (int a = 0; a < 3; a++) r->mean_acc[a] += (acc[a] - r->mean_acc[a]) / r->imu_cnt;
  }
  ImuMsg m;
  m.t = stamp - r->timediff_imu_wrt_lidar - r->time_lag_IMU_wtr_lidar;  // IMU time compensation
  std::memcpy(m.gyr, gyr, 24);
  std::memcpy(m.acc, acc, 24);
  if (m.t < r->last_timestamp_imu) {  // "IMU loop back, clear IMU buffer"
    r->imu_buffer.clear();
    r->imu_all.clear();
  }
  r->last_timestamp_imu = m.t;
  r->imu_buffer.push_back(m);
  if (!r->imu_en && !r->data_accum_finished) {  // Init_LI->push_ALL_IMU_CalibState (include/LI_init/LI_init.cpp:54-62)
    lii_calib_state c{};
    for (int a = 0; a < 3; a++) { c.ang_vel[a] = gyr[a]; c.linear_acc[a] = acc[a] / r->mean_acc_norm * kG; }
    for (int e = 0; e < 3; e++) c.rot_end[4 * e] = 1.0;
    c.timestamp = m.t;
    r->imu_all.push_back(c);
  }
  return LII_OK;
}

static int push_lidar(lii_replay* r, LidarMsg&& m) {
  r->scan_count++;
  m.scan_count = r->scan_count;
  if (m.stamp < r->last_timestamp_lidar) {  // "lidar loop back, clear buffer"
    r->lidar_msgs.clear();
    r->frames.clear();
    r->frame_next = 0;
    r->lidar_pushed = false;
  }
  r->last_timestamp_lidar = m.stamp;
  if (std::fabs(r->last_timestamp_imu - r->last_timestamp_lidar) > 1.0 && !r->timediff_set_flg && !r->imu_buffer.empty()) {
    r->timediff_set_flg = true;  // "Self sync IMU and LiDAR, HARD time lag"
    r->timediff_imu_wrt_lidar = r->last_timestamp_imu - r->last_timestamp_lidar;
  }
  r->lidar_msgs.push_back(std::move(m));
  return LII_OK;
}
// standard_pcl_cbk :345-379 (data = sensor_msgs/PointCloud2::data as received)
int lii_replay_pcl2(lii_replay* r, double stamp, const void* data, int32_t n_points, const lii_pc2_fields* f) {
  if (!r || !data || n_points < 0 || !f) return LII_ERR_INVALID;
  LidarMsg m;
  m.stamp = stamp; m.kind = 0; m.n_points = n_points; m.f2 = *f;
  m.data.assign(static_cast<const unsigned char*>(data), static_cast<const unsigned char*>(data) + size_t(n_points) * size_t(f->point_step));
  return push_lidar(r, std::move(m));
}
// livox_pcl_cbk :310-343 (points = the CustomPoint array)
int lii_replay_livox(lii_replay* r, double stamp, const void* points, int32_t n_points, const lii_livox_fields* f) {
  if (!r || !points || n_points < 0 || !f) return LII_ERR_INVALID;
  LidarMsg m;
  m.stamp = stamp; m.kind = 1; m.n_points = n_points; m.fl = *f;
  m.data.assign(static_cast<const unsigned char*>(points), static_cast<const unsigned char*>(points) + size_t(n_points) * size_t(f->point_step));
  return push_lidar(r, std::move(m));
}

// ImuProcess::Forward_propagation_without_imu without its de-skew loop (src/IMU_Processing.hpp:204-244): constant-velocity
// model - bias_g holds the angular velocity, vel_end the linear velocity.  `first`: b_first_frame_ (dt = 0.1).
// (exported for the test that holds it to the unmodified header)
void lii_replay_cv_propagate(lii_state* st, double dt, const double cov_gyr_scale[3], const double cov_acc_scale[3]) {
  static thread_local double F[24 * 24], Q[24 * 24];
  std::memset(F, 0, sizeof(F));
  std::memset(Q, 0, sizeof(Q));
  for (int e = 0; e < 24; e++) F[25 * e] = 1.0;
  double Exp_f[9], Exp_m[9];
  so3_exp(st->bias_g, dt, Exp_f);
  so3_exp(st->bias_g, -dt, Exp_m);
  for (int rr = 0; rr < 3; rr++)
    for (int c = 0; c < 3; c++) F[24 * rr + c] = Exp_m[3 * rr + c];
  for (int a = 0; a < 3; a++) {
    F[24 * a + 15 + a] = dt;
    F[24 * (3 + a) + 12 + a] = dt;
    Q[25 * (15 + a)] = cov_gyr_scale[a] * dt * dt;
    Q[25 * (12 + a)] = cov_acc_scale[a] * dt * dt;
  }
  cov_propagate(st->cov, F, Q);
  m3_mul(st->rot_end, Exp_f, st->rot_end);
  for (int a = 0; a < 3; a++) st->pos_end[a] += st->vel_end[a] * dt;
}

// The forward part of ImuProcess::propagation_and_undist (src/IMU_Processing.hpp:271-382): mid-point integration over the scan's
// IMU samples, covariance propagation, the IMUpose table for the back-propagation.  carry = acc_s_last[3], angvel_last[3],
// last_lidar_end_time (in / out); poses: room for n_imu + 1 records.  (exported for the same test)
void lii_replay_imu_propagate(lii_state* st, const double* imu7 /* n x (t, gyr, acc) */, int32_t n_imu, const double last_imu7[7],
                              double carry[7], const double cov6x3[18] /* gyr, acc, bias_gyr, bias_acc, R_LI, T_LI */,
                              double mean_acc_norm, double pcl_beg_time, double pcl_end_time, lii_pose6d* poses, int32_t* n_poses) {
  const double* cov_gyr = cov6x3; const double* cov_acc = cov6x3 + 3; const double* cov_bg = cov6x3 + 6;
  const double* cov_ba = cov6x3 + 9; const double* cov_RLI = cov6x3 + 12; const double* cov_TLI = cov6x3 + 15;
  double* acc_s_last = carry; double* angvel_last = carry + 3; double& last_end = carry[6];
  std::vector<const double*> v;
  v.push_back(last_imu7);
  for (int i = 0; i < n_imu; i++) v.push_back(imu7 + 7 * i);
  const double imu_end_time = v.back()[0];
  int K = 0;
  auto set_pose = [&](double t, const double* acc, const double* gyr, const double* vel, const double* pos, const double* R) {
    lii_pose6d& p = poses[K++];
    p.offset_time = t;
    std::memcpy(p.acc, acc, 24); std::memcpy(p.gyr, gyr, 24); std::memcpy(p.vel, vel, 24); std::memcpy(p.pos, pos, 24);
    std::memcpy(p.rot, R, 72);
  };
  set_pose(0.0, acc_s_last, angvel_last, st->vel_end, st->pos_end, st->rot_end);
  double acc_imu[3] = {0, 0, 0}, angvel_avr[3] = {0, 0, 0}, acc_avr[3], vel[3], pos[3], R[9];
  std::memcpy(vel, st->vel_end, 24); std::memcpy(pos, st->pos_end, 24); std::memcpy(R, st->rot_end, 72);
  static thread_local double F[24 * 24], Q[24 * 24];
  double dt = 0;
  for (size_t i = 0; i + 1 < v.size(); i++) {
    const double* head = v[i];
    const double* tail = v[i + 1];
    if (tail[0] < last_end) continue;
    for (int a = 0; a < 3; a++) {
      angvel_avr[a] = 0.5 * (head[1 + a] + tail[1 + a]);
      acc_avr[a] = 0.5 * (head[4 + a] + tail[4 + a]);
    }
    for (int a = 0; a < 3; a++) {
      angvel_avr[a] -= st->bias_g[a];
      acc_avr[a] = acc_avr[a] / mean_acc_norm * kG - st->bias_a[a];
    }
    dt = head[0] < last_end ? tail[0] - last_end : tail[0] - head[0];
    double Exp_f[9], Exp_m[9];
    so3_exp(angvel_avr, dt, Exp_f);
    so3_exp(angvel_avr, -dt, Exp_m);
    const double Ks[9] = {0.0, -acc_avr[2], acc_avr[1], acc_avr[2], 0.0, -acc_avr[0], -acc_avr[1], acc_avr[0], 0.0};
    std::memset(F, 0, sizeof(F));
    std::memset(Q, 0, sizeof(Q));
    for (int e = 0; e < 24; e++) F[25 * e] = 1.0;
    double RK[9];
    m3_mul(R, Ks, RK);
    for (int rr = 0; rr < 3; rr++)
      for (int c = 0; c < 3; c++) {
        F[24 * rr + c] = Exp_m[3 * rr + c];
        F[24 * (12 + rr) + c] = -RK[3 * rr + c] * dt;
        F[24 * (12 + rr) + 18 + c] = -R[3 * rr + c] * dt;
      }
    for (int a = 0; a < 3; a++) {
      F[24 * a + 15 + a] = -dt;
      F[24 * (3 + a) + 12 + a] = dt;
      F[24 * (12 + a) + 21 + a] = dt;
      Q[25 * a] = cov_gyr[a] * dt * dt;
      Q[25 * (6 + a)] = cov_RLI[a] * dt * dt;
      Q[25 * (9 + a)] = cov_TLI[a] * dt * dt;
      Q[25 * (15 + a)] = cov_bg[a] * dt * dt;
      Q[25 * (18 + a)] = cov_ba[a] * dt * dt;
    }
    {  // R cov_acc.asDiagonal() R^T dt dt
      double RD[9], Rt[9], RDRt[9];
      for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) RD[3 * rr + c] = R[3 * rr + c] * cov_acc[c];
      m3_t(R, Rt);
      m3_mul(RD, Rt, RDRt);
      for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 3; c++) Q[24 * (12 + rr) + 12 + c] = RDRt[3 * rr + c] * dt * dt;
    }
    cov_propagate(st->cov, F, Q);
    m3_mul(R, Exp_f, R);
    double Ra[3];
    m3_vec(R, acc_avr, Ra);
    for (int a = 0; a < 3; a++) acc_imu[a] = Ra[a] + st->gravity[a];
    for (int a = 0; a < 3; a++) pos[a] = pos[a] + vel[a] * dt + 0.5 * acc_imu[a] * dt * dt;
    for (int a = 0; a < 3; a++) vel[a] = vel[a] + acc_imu[a] * dt;
    std::memcpy(angvel_last, angvel_avr, 24);
    std::memcpy(acc_s_last, acc_imu, 24);
    set_pose(tail[0] - pcl_beg_time, acc_imu, angvel_avr, vel, pos, R);
  }
  const double note = pcl_end_time > imu_end_time ? 1.0 : -1.0;
  dt = note * (pcl_end_time - imu_end_time);
  double w_end[3] = {note * angvel_avr[0], note * angvel_avr[1], note * angvel_avr[2]}, E[9];
  so3_exp(w_end, dt, E);
  for (int a = 0; a < 3; a++) {
    st->vel_end[a] = vel[a] + note * acc_imu[a] * dt;
    st->pos_end[a] = pos[a] + note * vel[a] * dt + note * 0.5 * acc_imu[a] * dt * dt;
  }
  m3_mul(R, E, st->rot_end);
  last_end = pcl_end_time;
  *n_poses = K;
}

// fileout_calib_result, src/laserMapping.cpp:708-725 (Eigen's default stream format under fixed / setprecision(6): the coefficients of
// ONE matrix are right-aligned to its widest one, separated by one space)
static std::string eigen_fmt(const double* m, int rows, int cols) {
  std::vector<std::string> cells;
  size_t w = 0;
  char buf[64];
  for (int i = 0; i < rows * cols; i++) {
    std::snprintf(buf, sizeof(buf), "%.6f", m[i]);
    cells.emplace_back(buf);
    w = std::max(w, cells.back().size());
  }
  std::string out;
  for (int rr = 0; rr < rows; rr++) {
    for (int c = 0; c < cols; c++) {
      const std::string& s = cells[size_t(rr * cols + c)];
      if (c) out += " ";
      out += std::string(w - s.size(), ' ') + s;
    }
    if (rr + 1 < rows) out += "\n";
  }
  return out;
}
static int write_result(lii_replay* r, const char* title, bool append) {
  if (r->result_path.empty()) return LII_OK;
  FILE* f = std::fopen(r->result_path.c_str(), append ? "a" : "w");
  if (!f) return fail(r, LII_ERR_INVALID, "cannot open " + r->result_path);
  const lii_state& s = r->state;
  double e[3];
  rot_to_euler(s.offset_R_L_I, e);
  for (double& v : e) v *= 57.3;
  std::fprintf(f, "%s\n", title);
  std::fprintf(f, "Rotation LiDAR to IMU (degree)     = %s\n", eigen_fmt(e, 1, 3).c_str());
  std::fprintf(f, "Translation LiDAR to IMU (meter)   = %s\n", eigen_fmt(s.offset_T_L_I, 1, 3).c_str());
  std::fprintf(f, "Time Lag IMU to LiDAR (second)     = %.6f\n", r->time_lag_IMU_wtr_lidar + r->timediff_imu_wrt_lidar);
  std::fprintf(f, "Bias of Gyroscope  (rad/s)         = %s\n", eigen_fmt(s.bias_g, 1, 3).c_str());
  std::fprintf(f, "Bias of Accelerometer (meters/s^2) = %s\n", eigen_fmt(s.bias_a, 1, 3).c_str());
  std::fprintf(f, "Gravity in World Frame(meters/s^2) = %s\n\n", eigen_fmt(s.gravity, 1, 3).c_str());
  double T[16] = {0};
  for (int rr = 0; rr < 3; rr++) {
    for (int c = 0; c < 3; c++) T[4 * rr + c] = s.offset_R_L_I[3 * rr + c];
    T[4 * rr + 3] = s.offset_T_L_I[rr];
  }
  T[15] = 1.0;
  std::fprintf(f, "Homogeneous Transformation Matrix from LiDAR to IMU: \n%s\n\n\n", eigen_fmt(T, 4, 4).c_str());
  std::fclose(f);
  return LII_OK;
}

// sync_packages, src/laserMapping.cpp:436-480.  The message at the head of the queue is cut into its sub-frames on the device when
// the previous message's frames are used up.
static int sync(lii_replay* r, bool* ready) {
  *ready = false;
  if (r->frame_next >= r->frames.size()) {  // lidar_buffer is empty: cut the next driver message
    if (r->lidar_msgs.empty()) return LII_OK;
    LidarMsg& m = r->lidar_msgs.front();
    lii_ingest_opts io = r->ing;
    io.struct_size = sizeof(io);
    io.stamp_s = m.stamp;
    io.cut_frame_num = r->prm.cut_frame ? r->cut_frame_num : 0;  // (cut_frame: false -> Preprocess::process, laserMapping.cpp:337-342, :374-379)
    io.scan_count = m.scan_count;
    r->frames.assign(64, lii_frame_info{});
    int32_t nf = 0;
    const int rc = m.kind == 0 ? lii_ingest_pcl2(r->h, m.data.data(), m.n_points, &m.f2, &io, r->frames.data(), 64, &nf)
                               : lii_ingest_livox(r->h, m.data.data(), m.n_points, &m.fl, &io, r->frames.data(), 64, &nf);
    r->lidar_msgs.pop_front();
    if (rc != LII_OK) return fail(r, rc, std::string("ingest: ") + lii_last_error(r->h));
    r->frames.resize(size_t(nf));
    r->frame_next = 0;
    if (nf == 0) return LII_OK;
  }
  if (r->imu_buffer.empty()) return LII_OK;
  const lii_frame_info& fr = r->frames[r->frame_next];
  if (!r->lidar_pushed) {
    if (fr.count <= 1) {  // "Too few input point cloud!"
      r->frame_next++;
      return LII_OK;
    }
    r->lidar_beg_time = fr.begin_time_s;
    r->lidar_end_time = fr.begin_time_s + fr.last_offset_ms / 1000.0;
    r->lidar_pushed = true;
  }
  if (r->last_timestamp_imu < r->lidar_end_time) return LII_OK;
  double imu_time = r->imu_buffer.front().t;
  r->meas_imu.clear();
  while (!r->imu_buffer.empty() && imu_time < r->lidar_end_time) {
    imu_time = r->imu_buffer.front().t;
    if (imu_time > r->lidar_end_time) break;
    r->meas_imu.push_back(r->imu_buffer.front());
    r->imu_buffer.pop_front();
  }
  r->lidar_pushed = false;
  *ready = true;
  return LII_OK;
}

static void log_row(lii_replay* r, int iterations, int effect_num) {
  const size_t at = r->log.size();
  r->log.resize(at + LII_REPLAY_ROW);
  double* row = r->log.data() + at;
  row[0] = r->lidar_end_time; row[1] = r->imu_en ? 1.0 : 0.0; row[2] = iterations; row[3] = effect_num;
  std::memcpy(row + 4, &r->state, sizeof(double) * 36);
}

// One pass of the main loop's body for the frame sync() has just completed (src/laserMapping.cpp:895-1234).
static int process(lii_replay* r) {
  const int frame = int(r->frame_next);
  r->frame_next++;
  lii_state& st = r->state;
  lii_scan_job job;
  std::memset(&job, 0, sizeof(job));
  job.struct_size = sizeof(job);
  job.leaf = r->leaf;
  job.opts = r->opts;
  job.opts.imu_en = r->imu_en ? 1 : 0;
  job.scan_sorted = r->prm.cut_frame ? 1 : 0;  // lii_ingest_* hands cut frames over in ascending time order, as process_cut_frame_* does; a whole message keeps the driver's order
  bool select = true;
  // ---- p_imu->Process(Measures, state, feats_undistort), src/IMU_Processing.hpp:419-462
  if (r->imu_en) {
    if (r->meas_imu.empty()) return LII_OK;
    if (r->imu_need_init) {
      // LI_init_done: "[Refinement] Switch to LIO mode" - the first frame after the hand-over only re-arms the processor and
      // RETURNS: the state is not propagated and feats_undistort keeps the PREVIOUS scan, which the loop below registers once
      // more (at the re-expressed state).  The handle's current scan is that previous, de-skewed scan: it is not replaced.
      r->last_imu = r->meas_imu.back();
      r->imu_need_init = false;
      for (int a = 0; a < 3; a++) { r->cov_acc[a] = 0.1; r->cov_gyr[a] = 0.1; }  // cov_acc_scale / cov_gyr_scale as set at :1205-1206
      select = false;
      job.undistort = 0;
    } else {
      r->imu_pose.assign(r->meas_imu.size() + 2, lii_pose6d{});
      std::vector<double> imu7(r->meas_imu.size() * 7);
      for (size_t i = 0; i < r->meas_imu.size(); i++) {
        imu7[7 * i] = r->meas_imu[i].t;
        std::memcpy(&imu7[7 * i + 1], r->meas_imu[i].gyr, 24);
        std::memcpy(&imu7[7 * i + 4], r->meas_imu[i].acc, 24);
      }
      double last7[7] = {r->last_imu.t};
      std::memcpy(last7 + 1, r->last_imu.gyr, 24);
      std::memcpy(last7 + 4, r->last_imu.acc, 24);
      double carry[7];
      std::memcpy(carry, r->acc_s_last, 24); std::memcpy(carry + 3, r->angvel_last, 24); carry[6] = r->last_lidar_end_time_;
      double cov[18];
      std::memcpy(cov, r->cov_gyr, 24); std::memcpy(cov + 3, r->cov_acc, 24); std::memcpy(cov + 6, r->cov_bias_gyr, 24);
      std::memcpy(cov + 9, r->cov_bias_acc, 24); std::memcpy(cov + 12, r->cov_R_LI, 24); std::memcpy(cov + 15, r->cov_T_LI, 24);
      int32_t K = 0;
      lii_replay_imu_propagate(&st, imu7.data(), int32_t(r->meas_imu.size()), last7, carry, cov, r->imu_mean_acc_norm, r->lidar_beg_time,
                               r->lidar_end_time, r->imu_pose.data(), &K);
      std::memcpy(r->acc_s_last, carry, 24); std::memcpy(r->angvel_last, carry + 3, 24); r->last_lidar_end_time_ = carry[6];
      r->last_imu = r->meas_imu.back();
      job.undistort = 1;
      job.imu_poses = r->imu_pose.data();
      job.n_imu_poses = K;
    }
  } else {
    // Forward_propagation_without_imu.  (The reference leaves time_last_scan unset on its first frame and reads it on the second:
    // dt of that frame is then the absolute stamp.  Here the first frame records its begin time.)
    double dt;
    if (r->b_first_frame) { dt = 0.1; r->b_first_frame = false; }
    else dt = r->lidar_beg_time - r->time_last_scan;
    r->time_last_scan = r->lidar_beg_time;
    lii_replay_cv_propagate(&st, dt, r->cov_gyr, r->cov_acc);
    job.undistort = 2;
  }
  r->state_propagat = st;
  if (select) {
    const int rc = lii_frame_select(r->h, frame);
    if (rc != LII_OK) return fail(r, rc, std::string("lii_frame_select: ") + lii_last_error(r->h));
  }
  r->have_scan = true;
  // ---- the first scan seeds the map (:921-931): de-skew + voxel grid, pointBodyToWorld on the host, ikdtree.Build
  if (!r->map_built) {
    int rc = LII_OK;
    if (job.undistort == 2) rc = lii_undistort_cv(r->h, st.bias_g, st.vel_end, st.rot_end);
    else if (job.undistort == 1) rc = lii_undistort_imu(r->h, job.imu_poses, job.n_imu_poses, st.rot_end, st.pos_end, st.offset_R_L_I, st.offset_T_L_I);
    int32_t n_down = 0;
    if (rc == LII_OK) rc = lii_downsample(r->h, r->leaf, &n_down, nullptr);
    if (rc != LII_OK) return fail(r, rc, std::string("first scan: ") + lii_last_error(r->h));
    if (n_down > 5) {
      std::vector<float> body(size_t(n_down) * 4), world(size_t(n_down) * 3);
      int32_t n = 0;
      rc = lii_scan_download(r->h, 1, body.data(), n_down, &n);
      if (rc != LII_OK) return fail(r, rc, std::string("lii_scan_download: ") + lii_last_error(r->h));
      for (int i = 0; i < n; i++) {  // pointBodyToWorld :209-220
        const double pb[3] = {body[4 * size_t(i)], body[4 * size_t(i) + 1], body[4 * size_t(i) + 2]};
        double pi[3], pw[3];
        m3_vec(st.offset_R_L_I, pb, pi);
        for (int a = 0; a < 3; a++) pi[a] += st.offset_T_L_I[a];
        m3_vec(st.rot_end, pi, pw);
        for (int a = 0; a < 3; a++) world[3 * size_t(i) + a] = float(pw[a] + st.pos_end[a]);
      }
      rc = lii_map_build(r->h, world.data(), n, 12);
      if (rc != LII_OK) return fail(r, rc, std::string("lii_map_build: ") + lii_last_error(r->h));
      r->map_built = true;
    }
    return LII_OK;
  }
  // ---- ICP + iterated Kalman filter update (:957-1134) and map_incremental (:1146)
  // (map_incremental rides in the job - lii_scan_job::map_update: its launches are enqueued behind the update's passes)
  lii_iekf_report rep{};
  job.map_update = 1;
  int rc = lii_scan_register(r->h, &job, &st, &r->state_propagat, &rep);
  if (rc != LII_OK) return fail(r, rc, std::string("lii_scan_register (+ map_incremental): ") + lii_last_error(r->h));
  // ---- "Device starts to move, data accumulation begins" (:1151-1155)
  const double pn = std::sqrt(st.pos_end[0] * st.pos_end[0] + st.pos_end[1] * st.pos_end[1] + st.pos_end[2] * st.pos_end[2]);
  if (!r->imu_en && !r->data_accum_start && pn > 0.05) {
    r->data_accum_start = true;
    r->move_start_time = r->lidar_end_time;
  }
  r->frame_num++;
  log_row(r, rep.iterations, rep.effect_num);
  // ---- refinement result after online_refine_time of LIO (:1164-1178)
  if (r->imu_en && !r->refine_done) {
    double done = r->lidar_end_time - r->online_calib_starts_time;
    if (done > r->prm.online_refine_time - 1e-6) {
      r->refine_done = true;
      rc = write_result(r, "Refinement result:", true);
      if (rc != LII_OK) return rc;
    }
  }
  // ---- accumulation, excitation appraisal, LI_Initialization, switch to LIO (:1169-1212)
  if (!r->imu_en && !r->data_accum_finished && r->data_accum_start) {
    lii_calib_state c{};  // Init_LI->push_Lidar_CalibState(state.rot_end, state.bias_g, state.vel_end, lidar_end_time)
    std::memcpy(c.rot_end, st.rot_end, 72);
    std::memcpy(c.ang_vel, st.bias_g, 24);
    std::memcpy(c.linear_vel, st.vel_end, 24);
    c.timestamp = r->lidar_end_time;
    r->lidar_states.push_back(c);
    r->omg.insert(r->omg.end(), st.bias_g, st.bias_g + 3);
    double ev[3], pct[3];
    int32_t sufficient = 0;
    // "Give a Data Appraisal every second": `frame_num % orig_odom_freq * cut_frame_num == 0` (include/LI_init/LI_init.cpp:513 -
    // by C++ precedence (frame_num % orig_odom_freq) * cut_frame_num, i.e. every orig_odom_freq-th frame)
    if ((r->frame_num % r->prm.orig_odom_freq) * r->cut_frame_num == 0) {
      rc = lii_data_sufficiency(r->omg.data(), int32_t(r->omg.size() / 3), r->prm.data_accum_length, ev, pct, &sufficient);
      if (rc != LII_OK) return fail(r, rc, "lii_data_sufficiency");
    }
    if (sufficient) {
      r->data_accum_finished = true;
      std::vector<lii_calib_state> oi(r->lidar_states.size()), ol(r->lidar_states.size());
      int32_t n = 0;
      rc = lii_li_init_interpolate(r->imu_all.data(), int32_t(r->imu_all.size()), r->lidar_states.data(), int32_t(r->lidar_states.size()),
                                   r->move_start_time, oi.data(), ol.data(), &n);
      if (rc == LII_OK) rc = lii_li_init_run(r->h, oi.data(), ol.data(), n, r->prm.orig_odom_freq, r->cut_frame_num, &r->init, &r->init_lag1, &r->init_total);
      if (rc != LII_OK) return fail(r, rc, std::string("LI_Initialization: ") + lii_last_error(r->h));
      r->online_calib_starts_time = r->lidar_end_time;
      // Transfer to FAST-LIO2 (:1183-1212): the body frame becomes the IMU frame
      r->imu_en = true;
      double RLIt[9], v[3], w[3], Rn[9];
      std::memcpy(st.offset_R_L_I, r->init.R_LI, 72);
      std::memcpy(st.offset_T_L_I, r->init.T_LI, 24);
      m3_t(st.offset_R_L_I, RLIt);
      m3_vec(RLIt, st.offset_T_L_I, v);     // R_LI^T T_LI
      m3_vec(st.rot_end, v, w);             // rot_end R_LI^T T_LI
      for (int a = 0; a < 3; a++) st.pos_end[a] = -w[a] + st.pos_end[a];
      m3_mul(st.rot_end, RLIt, Rn);
      std::memcpy(st.rot_end, Rn, 72);
      std::memcpy(st.gravity, r->init.grav_L0, 24);
      std::memcpy(st.bias_g, r->init.gyro_bias, 24);
      std::memcpy(st.bias_a, r->init.acc_bias, 24);
      if (r->prm.lidar_type != LII_LIDAR_AVIA) r->cut_frame_num = 2;
      r->time_lag_IMU_wtr_lidar = r->init_total;  // get_total_time_lag(): compensate the IMU stamps in the buffer
      for (ImuMsg& m : r->imu_buffer) m.t -= r->time_lag_IMU_wtr_lidar;
      if (!r->imu_buffer.empty()) r->last_timestamp_imu = r->imu_buffer.back().t;
      r->li_init_done = true;
      r->imu_need_init = true;
      r->imu_mean_acc_norm = r->mean_acc_norm;
      for (int a = 0; a < 3; a++) { r->cov_gyr[a] = 0.1; r->cov_acc[a] = 0.1; r->cov_bias_gyr[a] = 0.0001; r->cov_bias_acc[a] = 0.0001; }
      rc = write_result(r, "Initialization result:", false);
      if (rc != LII_OK) return rc;
    }
  }
  return LII_OK;
}

// ros::spinOnce() + the while loop of main(): processes every complete (scan, IMU) package the buffers hold.
// Returns the number of scans processed (>= 0) or a negative lii_status.
int lii_replay_spin(lii_replay* r) {
  if (!r) return LII_ERR_INVALID;
  int n = 0;
  for (;;) {
    if (r->stop_after_init && r->data_accum_finished) break;
    bool ready = false;
    const size_t before = r->frame_next, msgs_before = r->lidar_msgs.size();
    int rc = sync(r, &ready);
    if (rc != LII_OK) return rc;
    if (!ready) {
      if (r->frame_next != before || r->lidar_msgs.size() != msgs_before) continue;  // a frame was dropped / a message cut: look again
      break;
    }
    rc = process(r);
    if (rc != LII_OK) return rc;
    n++;
  }
  return n;
}

int lii_replay_get_status(lii_replay* r, lii_replay_status* out) {
  if (!r || !out || out->struct_size != sizeof(lii_replay_status)) return LII_ERR_INVALID;
  out->imu_en = r->imu_en; out->data_accum_start = r->data_accum_start; out->data_accum_finished = r->data_accum_finished;
  out->refine_done = r->refine_done;
  out->scans_processed = r->frame_num;
  out->frames_pending = int32_t(r->frames.size() - std::min(r->frame_next, r->frames.size())) + int32_t(r->lidar_msgs.size());
  out->imu_pending = int32_t(r->imu_buffer.size());
  out->cut_frame_num = r->cut_frame_num;
  out->move_start_time = r->move_start_time;
  out->time_lag_imu_wrt_lidar = r->time_lag_IMU_wtr_lidar;
  out->timediff_imu_wrt_lidar = r->timediff_imu_wrt_lidar;
  out->mean_acc_norm = r->mean_acc_norm;
  out->lidar_end_time = r->lidar_end_time;
  out->state = r->state;
  out->init = r->init;
  out->init_time_lag_1 = r->init_lag1;
  out->init_total_time_lag = r->init_total;
  return LII_OK;
}
// rows of LII_REPLAY_ROW doubles, one per processed scan, oldest first
int lii_replay_log(lii_replay* r, double* out, int32_t capacity_rows, int32_t* n_rows) {
  if (!r || !n_rows) return LII_ERR_INVALID;
  const int32_t n = int32_t(r->log.size() / LII_REPLAY_ROW);
  *n_rows = n;
  if (!out) return LII_OK;
  if (capacity_rows < n) return LII_ERR_CAPACITY;
  std::memcpy(out, r->log.data(), r->log.size() * sizeof(double));
  return LII_OK;
}

}  // extern "C"
