// LI_Init::LI_Initialization (include/LI_init/LI_init.cpp:586-632) — the host-side signal-conditioning chain of the batch
// calibration in C++, wrapped around the HIP residual evaluators (lii_calib_set_buffers / lii_calib_solve_stage).
// §8(f)4 of SURVEY.md: sequential filters over N_s ~ 10^3 records stay on the host; the three least-squares problems
// evaluate on the GPU.
//   downsample_interpolate_IMU :82-125   IMU_time_compensate :195-221   cut_sequence_tail :223-238
//   Butter_filt :260-304 (coefficients LI_init.h:218-224, incl. Coeff_b[4] = 0.0011)   zero_phase_filt :306-315
//   normalize_acc :494-504   xcorr_temporal_init :160-193   central_diff :127-158   acc_interpolate :240-258
//   set_IMU_state / set_Lidar_state :27-33 (drop the last element)   set_states_2nd_filter :35-41
// Quirks kept on purpose: CalibState::operator= copies only the four 3-vectors, so the filters
// leave timeStamp / rot_end of the padded samples untouched; IMU_time_compensate does not shift the last stamp and discards
// ten pairs on its first call; the tail cut removes 20.
// One documented deviation: the reference's mean filter reads one element past the end of its copy (UB, A10); here that
// read returns the last raw sample.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/liinit_hip.h"

struct lii_context;
int lii_internal_li_init_on_device(lii_context* h);  // lii_capi.cpp: lii_li_init_set_device

namespace {

using Seq = std::vector<lii_calib_state>;
constexpr double kG = 9.81;
const double kB[7] = {0.000076, 0.000457, 0.001143, 0.001524, 0.0011, 0.000457, 0.000076};
const double kA[7] = {1.0000, -4.182389, 7.491611, -7.313596, 4.089349, -1.238525, 0.158428};

inline double* vec4(lii_calib_state& s, int f) {  // the four fields CalibState arithmetic touches
  return f == 0 ? s.ang_vel : (f == 1 ? s.linear_vel : (f == 2 ? s.ang_acc : s.linear_acc));
}
inline double norm3(const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

void align(Seq& imu, Seq& lidar) {
  size_t li = 0, ii = 0;
  while (li < lidar.size() && lidar[li].timestamp < imu[ii].timestamp) li++;
  lidar.erase(lidar.begin(), lidar.begin() + li);
  while (ii + 1 < imu.size() && lidar.front().timestamp > imu[ii + 1].timestamp) ii++;
  imu.erase(imu.begin(), imu.begin() + ii);
  while (imu.size() > lidar.size()) imu.pop_back();
  while (imu.size() < lidar.size()) lidar.pop_back();
}
void time_compensate(Seq& imu, Seq& lidar, double lag, bool discard) {
  if (discard) {
    imu.erase(imu.begin(), imu.begin() + 10);
    lidar.erase(lidar.begin(), lidar.begin() + 10);
  }
  for (size_t i = 0; i + 1 < imu.size(); i++) imu[i].timestamp -= lag;
  align(imu, lidar);
}
void cut_tail(Seq& imu, Seq& lidar) {
  imu.resize(imu.size() - 20);
  lidar.resize(lidar.size() - 20);
  align(imu, lidar);
}
Seq butter(const Seq& in) {
  const int ext = 60, n = int(in.size()), nb = 7;
  Seq x;
  x.reserve(n + 2 * ext);
  for (int i = ext; i >= 1; i--) x.push_back(in[i]);
  for (int i = 0; i < n; i++) x.push_back(in[i]);
  for (int i = n - 2; i >= n - 1 - ext; i--) x.push_back(in[i]);
  Seq y = x;
  const int m = int(x.size());
  for (int i = nb; i < m - ext; i++) {
    double acc[4][3] = {{0}};
    for (int j = 0; j < nb; j++)
      for (int f = 0; f < 4; f++) {
        const double* v = vec4(x[i - j], f);
        for (int a = 0; a < 3; a++) acc[f][a] += v[a] * kB[j];
      }
    for (int j = 1; j < nb; j++)
      for (int f = 0; f < 4; f++) {
        const double* v = vec4(y[i - j], f);
        for (int a = 0; a < 3; a++) acc[f][a] -= v[a] * kA[j];
      }
    for (int f = 0; f < 4; f++) std::memcpy(vec4(y[i], f), acc[f], 24);
  }
  return Seq(y.begin() + ext, y.end() - ext);
}
Seq zero_phase(const Seq& in) {
  Seq a = butter(in);
  std::reverse(a.begin(), a.end());
  Seq b = butter(a);
  std::reverse(b.begin(), b.end());
  return b;
}
void normalize_acc(Seq& s) {
  double mean[3] = {0, 0, 0};
  for (int i = 1; i < 10; i++)
    for (int a = 0; a < 3; a++) mean[a] += (s[i].linear_acc[a] - mean[a]) / i;
  const double nrm = norm3(mean);
  for (auto& e : s)
    for (int a = 0; a < 3; a++) e.linear_acc[a] = e.linear_acc[a] / nrm * kG;
}
int xcorr(const Seq& imu, const Seq& lidar) {  // returns lag_IMU_wtr_Lidar
  const int n = int(imu.size());
  std::vector<double> a(n), b(n);
  double ma = 0, mb = 0;
  for (int i = 0; i < n; i++) {
    a[i] = norm3(imu[i].ang_vel);
    b[i] = norm3(lidar[i].ang_vel);
    ma += (a[i] - ma) / (i + 1);
    mb += (b[i] - mb) / (i + 1);
  }
  double best = -1.7976931348623157e308;
  int best_lag = 0;
  for (int lag = -n + 1; lag < n; lag++) {
    double c = 0;
    const int i0 = std::max(0, -lag), i1 = std::min(n, n - lag);
    for (int i = i0; i < i1; i++) c += (a[i] - ma) * (b[i + lag] - mb);
    if (c > best) { best = c; best_lag = -lag; }
  }
  return best_lag;
}
void central_diff(Seq& imu, Seq& lidar) {
  const int n = int(imu.size());
  std::vector<double> ia(size_t(n) * 3, 0.0), la(size_t(n) * 3, 0.0), ll(size_t(n) * 3, 0.0);
  for (int i = 1; i <= n - 3; i++) {
    const double dti = imu[i + 1].timestamp - imu[i - 1].timestamp, dtl = lidar[i + 1].timestamp - lidar[i - 1].timestamp;
    for (int a = 0; a < 3; a++) {
      ia[3 * i + a] = (imu[i + 1].ang_vel[a] - imu[i - 1].ang_vel[a]) / dti;
      la[3 * i + a] = (lidar[i + 1].ang_vel[a] - lidar[i - 1].ang_vel[a]) / dtl;
      ll[3 * i + a] = (lidar[i + 1].linear_vel[a] - lidar[i - 1].linear_vel[a]) / dtl;
    }
  }
  for (int i = 1; i <= n - 3; i++)
    for (int a = 0; a < 3; a++) {
      imu[i].ang_acc[a] = ia[3 * i + a];
      lidar[i].ang_acc[a] = la[3 * i + a];
      lidar[i].linear_acc[a] = ll[3 * i + a];
    }
}
void acc_interpolate(Seq& imu, const Seq& lidar) {
  for (size_t i = 1; i + 1 < lidar.size(); i++) {
    const double d = lidar[i].timestamp - imu[i].timestamp;
    if (d > 0) {
      const double s = d / (imu[i + 1].timestamp - imu[i].timestamp);
      for (int a = 0; a < 3; a++) imu[i].linear_acc[a] = s * imu[i + 1].linear_acc[a] + (1 - s) * imu[i].linear_acc[a];
    } else {
      const double s = -d / (imu[i].timestamp - imu[i - 1].timestamp);
      for (int a = 0; a < 3; a++) imu[i].linear_acc[a] = s * imu[i - 1].linear_acc[a] + (1 - s) * imu[i].linear_acc[a];
    }
    imu[i].timestamp += d;
  }
}

}  // namespace

extern "C" {

// LI_Init::data_sufficiency_assess (:506-556) without its terminal UI: Hessian_rot = Jacobian_rot^T Jacobian_rot with one
// 3x3 block [w]x per LO frame = sum (|w|^2 I - w w^T); its eigenvalues (cyclic Jacobi, symmetric 3x3) scaled by
// data_accum_length; Rot_percent = pairwise products; sufficient iff all three exceed 0.99.  The reference's EigenSolver
// returns the eigenvalues in no particular order and maps them to axes only for the progress bars; here they come ascending.
int lii_data_sufficiency(const double* omg, int32_t n_frames, double data_accum_length, double eigenvalues[3],
                         double rot_percent[3], int32_t* sufficient) {
  if ((!omg && n_frames > 0) || n_frames < 0 || !(data_accum_length > 0) || !eigenvalues || !rot_percent || !sufficient) return LII_ERR_INVALID;
  double Hm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int f = 0; f < n_frames; f++) {
    const double* w = omg + 3 * (size_t)f;
    // [w]x^T [w]x, entry by entry as the matrix product forms it
    const double K[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += K[k][i] * K[k][j];
        Hm[i][j] += s;
      }
  }
  for (int sweep = 0; sweep < 60; sweep++) {
    const double off = std::fabs(Hm[0][1]) + std::fabs(Hm[0][2]) + std::fabs(Hm[1][2]);
    if (off < 1e-300 || off < 1e-18 * (std::fabs(Hm[0][0]) + std::fabs(Hm[1][1]) + std::fabs(Hm[2][2]))) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (Hm[p][q] == 0.0) continue;
        const double theta = (Hm[q][q] - Hm[p][p]) / (2.0 * Hm[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 3; k++) {  // A <- A J
          const double akp = Hm[k][p], akq = Hm[k][q];
          Hm[k][p] = c * akp - sn * akq;
          Hm[k][q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {  // A <- J^T A
          const double apk = Hm[p][k], aqk = Hm[q][k];
          Hm[p][k] = c * apk - sn * aqk;
          Hm[q][k] = sn * apk + c * aqk;
        }
      }
  }
  double ev[3] = {Hm[0][0], Hm[1][1], Hm[2][2]};
  std::sort(ev, ev + 3);
  for (int a = 0; a < 3; a++) eigenvalues[a] = ev[a];
  const double s0 = ev[0] / data_accum_length, s1 = ev[1] / data_accum_length, s2 = ev[2] / data_accum_length;
  rot_percent[0] = s1 * s2;
  rot_percent[1] = s0 * s2;
  rot_percent[2] = s0 * s1;
  *sufficient = (rot_percent[0] > 0.99 && rot_percent[1] > 0.99 && rot_percent[2] > 0.99) ? 1 : 0;
  return LII_OK;
}

// downsample_interpolate_IMU (:82-125).  imu_all: raw IMU states (ang_vel, linear_acc already scaled to m/s^2, timestamp);
// lidar: LiDAR-odometry states.  Writes the interpolated IMU sequence (one per retained LiDAR state) and the retained
// LiDAR states; returns their count through n_out (capacity = n_lidar).
int lii_li_init_interpolate(const lii_calib_state* imu_all, int32_t n_imu, const lii_calib_state* lidar, int32_t n_lidar,
                            double move_start_time, lii_calib_state* imu_out, lii_calib_state* lidar_out, int32_t* n_out) {
  if (!imu_all || !lidar || !imu_out || !lidar_out || !n_out || n_imu < 6 || n_lidar < 1) return LII_ERR_INVALID;
  Seq all(imu_all, imu_all + n_imu), lid(lidar, lidar + n_lidar);
  size_t k = 0;
  while (k < all.size() && all[k].timestamp < move_start_time - 3.0) k++;
  all.erase(all.begin(), all.begin() + k);
  k = 0;
  while (k < lid.size() && lid[k].timestamp < move_start_time - 3.0) k++;
  lid.erase(lid.begin(), lid.begin() + k);
  if (all.size() < 6) return LII_ERR_INVALID;
  const Seq origin = all;  // the reference copies all but the last element and then reads one past it (UB): see header
  for (size_t i = 2; i + 2 < all.size(); i++) {
    double acc[3] = {0, 0, 0};
    for (int d = -2; d <= 2; d++)
      for (int a = 0; a < 3; a++) acc[a] += (origin[i + d].linear_acc[a] - acc[a]) / (d + 3);
    std::memcpy(all[i].linear_acc, acc, 24);
  }
  int cnt = 0;
  for (const auto& L : lid) {
    for (size_t j = 1; j < all.size(); j++) {
      if (all[j - 1].timestamp <= L.timestamp && all[j].timestamp > L.timestamp) {
        const double s = (all[j].timestamp - L.timestamp) / (all[j].timestamp - all[j - 1].timestamp);
        lii_calib_state o{};
        o.rot_end[0] = o.rot_end[4] = o.rot_end[8] = 1.0;
        for (int a = 0; a < 3; a++) {
          o.ang_vel[a] = s * all[j - 1].ang_vel[a] + (1 - s) * all[j].ang_vel[a];
          o.linear_acc[a] = s * all[j - 1].linear_acc[a] + (1 - s) * all[j].linear_acc[a];
        }
        o.timestamp = L.timestamp;
        imu_out[cnt] = o;
        lidar_out[cnt] = L;
        cnt++;
        break;
      }
    }
  }
  *n_out = cnt;
  return LII_OK;
}

// LI_Initialization from `IMU_time_compensate(0.0, true)` on (:593-625): conditioning on the host, the three solves on the
// GPU evaluators.  imu / lidar: the aligned sequences that fout_before_filter dumps (output of lii_li_init_interpolate).
int lii_li_init_run(lii_handle h, const lii_calib_state* imu_in, const lii_calib_state* lidar_in, int32_t n,
                    int32_t orig_odom_freq, int32_t cut_frame_num, lii_calib_result* out, double* time_lag_1,
                    double* total_time_lag) {
  if (!h || !imu_in || !lidar_in || !out || n < 200 || orig_odom_freq < 1 || cut_frame_num < 1) return LII_ERR_INVALID;
  Seq imu(imu_in, imu_in + n), lidar(lidar_in, lidar_in + n);
  // lii_li_init_set_device(h, 1): the zero-phase Butterworth passes and the O(N^2) cross-correlation run on the device
  // (lii_li_init_dev.hip - bit-identical to the host functions above); everything else of the chain is O(N) bookkeeping
  const bool dev = lii_internal_li_init_on_device(h) != 0;
  auto zero_phase_pair = [&](const Seq& a, const Seq& b, Seq& fa, Seq& fb) -> int {
    if (!dev || a.size() != b.size() || a.size() < 62) { fa = zero_phase(a); fb = zero_phase(b); return LII_OK; }
    Seq both(a);
    both.insert(both.end(), b.begin(), b.end());
    Seq out(both.size());
    const int rc_ = lii_zero_phase_filter(h, both.data(), 2, int32_t(a.size()), out.data());
    if (rc_ != LII_OK) return rc_;
    fa.assign(out.begin(), out.begin() + a.size());
    fb.assign(out.begin() + a.size(), out.end());
    return LII_OK;
  };
  time_compensate(imu, lidar, 0.0, true);
  Seq imu_f, lidar_f;
  int rc = zero_phase_pair(imu, lidar, imu_f, lidar_f);
  if (rc != LII_OK) return rc;
  normalize_acc(imu_f);
  imu.assign(imu_f.begin(), imu_f.end() - 1);
  lidar.assign(lidar_f.begin(), lidar_f.end() - 1);
  cut_tail(imu, lidar);
  int lag = 0;
  if (dev) {
    int32_t l = 0;
    if ((rc = lii_xcorr_lag(h, imu.data(), lidar.data(), int32_t(imu.size()), &l)) != LII_OK) return rc;
    lag = l;
  } else {
    lag = xcorr(imu, lidar);
  }
  const double lag1 = double(lag) / double(orig_odom_freq * cut_frame_num);
  time_compensate(imu, lidar, lag1, false);
  central_diff(imu, lidar);
  {
    Seq imu2, lidar2;
    if ((rc = zero_phase_pair(imu, lidar, imu2, lidar2)) != LII_OK) return rc;
    for (size_t i = 0; i < imu.size(); i++) {
      std::memcpy(imu[i].ang_acc, imu2[i].ang_acc, 24);
      std::memcpy(lidar[i].ang_acc, lidar2[i].ang_acc, 24);
      std::memcpy(lidar[i].linear_acc, lidar2[i].linear_acc, 24);
    }
  }
  std::memset(out, 0, sizeof(*out));
  out->R_LI[0] = out->R_LI[4] = out->R_LI[8] = 1.0;
  rc = lii_calib_set_buffers(h, imu.data(), lidar.data(), int32_t(imu.size()));
  if (rc != LII_OK) return rc;
  if ((rc = lii_calib_solve_stage(h, 1, out)) != LII_OK) return rc;
  if ((rc = lii_calib_solve_stage(h, 2, out)) != LII_OK) return rc;
  time_compensate(imu, lidar, out->time_lag_2, false);
  acc_interpolate(imu, lidar);
  if ((rc = lii_calib_set_buffers(h, imu.data(), lidar.data(), int32_t(imu.size()))) != LII_OK) return rc;
  if ((rc = lii_calib_solve_stage(h, 3, out)) != LII_OK) return rc;
  if (time_lag_1) *time_lag_1 = lag1;
  if (total_time_lag) *total_time_lag = lag1 + out->time_lag_2;
  return LII_OK;
}

}  // extern "C"
