// LI-Init batch calibration: host Levenberg-Marquardt around the HIP residual/Jacobian evaluator
// (lii_calib_eval -> k_calib_eval).  Replaces the three ceres::Solve calls of the reference:
//   solve_Rotation_only        include/LI_init/LI_init.cpp:317-343
//   solve_Rot_bias_gyro        include/LI_init/LI_init.cpp:345-401
//   solve_trans_biasacc_grav   include/LI_init/LI_init.cpp:403-492
// Ceres 2.0.0 (third-party, not vendored; docker/Dockerfile:16-21) is driven there with default
// Solver::Options.  The trust-region loop below follows Ceres' documented Levenberg-Marquardt:
// Jacobi column scaling 1/(1+||col||) fixed at iteration 0, (J^T J + D^2/radius) step with D^2 = clamp(diag J^T J,
// 1e-6, 1e32), radius 1e4 initially, step quality rho = actual / model decrease, accept if rho > 1e-3,
// radius /= max(1/3, 1-(2 rho-1)^3) on accept, radius /= 2,4,8.. on reject, tolerances function 1e-6 /
// gradient 1e-10 / parameter 1e-8 tested in Ceres' order (parameter & function tolerance on the candidate BEFORE
// it is accepted), at most 50 iterations, quaternion local parameterisation q <- [cos|d|, sinc|d| d] (x) q
// (so the tangent step is a rotation by 2|d|), box constraints by projection inside Plus.
// The Armijo projected line search Ceres adds for bounded problems is not reproduced (documented in DESIGN.md).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/liinit_hip.h"
#include "lii_hostmath.h"

namespace {

struct Quat { double w, x, y, z; };

void quat_to_rot(const Quat& q, double* R) {  // Eigen::Quaternion::toRotationMatrix
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
Quat rot_to_quat(const double* m) {  // Eigen::Quaternion(Matrix3) (Shoemake)
  Quat q;
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m[3 * k + j] - m[3 * j + k]) * t;
    v[j] = (m[3 * j + i] + m[3 * i + j]) * t;
    v[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}
// ceres::QuaternionParameterization::Plus
Quat quat_plus(const Quat& q, const double* d) {
  const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n <= 0.0) return q;
  const double s = std::sin(n) / n;
  const double a = std::cos(n), b = s * d[0], c = s * d[1], e = s * d[2];
  Quat r;  // (a,b,c,e) (x) q
  r.w = a * q.w - b * q.x - c * q.y - e * q.z;
  r.x = a * q.x + b * q.w + c * q.z - e * q.y;
  r.y = a * q.y - b * q.z + c * q.w + e * q.x;
  r.z = a * q.z + b * q.y - c * q.x + e * q.w;
  return r;
}

struct Problem {
  lii_handle h;
  int stage;
  int dof;         // tangent size: 3 / 7 / 9
  Quat q;          // rotation block
  double v[6];     // stage 2: b_g[3], t_d ; stage 3: b_a[3], T_IL[3]
  double R_LI[9];  // stage 3 constant
  double lo[6], hi[6];
  bool bounded[6];
  int n_vec() const { return dof - 3; }
};

int eval(const Problem& p, double* JtJ, double* Jtr, double* cost) {
  double params[24];
  quat_to_rot(p.q, params);
  if (p.stage == 2) {
    std::memcpy(params + 9, p.v, sizeof(double) * 4);
  } else if (p.stage == 3) {
    std::memcpy(params + 9, p.v, sizeof(double) * 6);
    std::memcpy(params + 15, p.R_LI, sizeof(double) * 9);
  }
  int rc = lii_calib_eval(p.h, p.stage, params, JtJ, Jtr, cost);
  if (rc != LII_OK || !JtJ) return rc;
  // device tangent is R <- Exp(delta) R; Ceres' quaternion tangent doubles the angle: scale the 3 rotation columns by 2
  const int n = p.dof;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double s = ((i < 3) ? 2.0 : 1.0) * ((j < 3) ? 2.0 : 1.0);
      JtJ[i * n + j] *= s;
    }
  for (int i = 0; i < 3; i++) Jtr[i] *= 2.0;
  return LII_OK;
}

Problem plus(const Problem& p, const double* delta) {
  Problem c = p;
  c.q = quat_plus(p.q, delta);
  for (int i = 0; i < p.n_vec(); i++) {
    double x = p.v[i] + delta[3 + i];
    if (p.bounded[i]) x = std::min(std::max(x, p.lo[i]), p.hi[i]);  // ParameterBlock::Plus projects onto the box
    c.v[i] = x;
  }
  return c;
}
double ambient_norm(const Problem& p) {
  double s = p.q.w * p.q.w + p.q.x * p.q.x + p.q.y * p.q.y + p.q.z * p.q.z;
  for (int i = 0; i < p.n_vec(); i++) s += p.v[i] * p.v[i];
  return std::sqrt(s);
}
double ambient_dist(const Problem& a, const Problem& b) {
  double s = (a.q.w - b.q.w) * (a.q.w - b.q.w) + (a.q.x - b.q.x) * (a.q.x - b.q.x) + (a.q.y - b.q.y) * (a.q.y - b.q.y) +
             (a.q.z - b.q.z) * (a.q.z - b.q.z);
  for (int i = 0; i < a.n_vec(); i++) s += (a.v[i] - b.v[i]) * (a.v[i] - b.v[i]);
  return std::sqrt(s);
}
// max-norm of the projected gradient step  x - Plus(x, -g)  (Ceres' gradient_max_norm)
double gradient_max_norm(const Problem& p, const double* g) {
  double neg[9];
  for (int i = 0; i < p.dof; i++) neg[i] = -g[i];
  Problem c = plus(p, neg);
  double m = std::max(std::max(std::fabs(p.q.w - c.q.w), std::fabs(p.q.x - c.q.x)),
                      std::max(std::fabs(p.q.y - c.q.y), std::fabs(p.q.z - c.q.z)));
  for (int i = 0; i < p.n_vec(); i++) m = std::max(m, std::fabs(p.v[i] - c.v[i]));
  return m;
}

// Returns the number of iterations (successful + unsuccessful); the final point is left in `p`.
int minimize(Problem& p, double* final_cost) {
  const int n = p.dof;
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double min_relative_decrease = 1e-3, min_diag = 1e-6, max_diag = 1e32, max_radius = 1e16, min_radius = 1e-32;
  const int max_iterations = 50, max_invalid = 5;
  double radius = 1e4, decrease_factor = 2.0;
  if (true) {  // Ceres projects the initial point onto the bounds
    double zero[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    p = plus(p, zero);
  }
  double JtJ[81], g[9], cost = 0;
  if (eval(p, JtJ, g, &cost) != LII_OK) return -1;
  double scale[9];
  for (int i = 0; i < n; i++) scale[i] = 1.0 / (1.0 + std::sqrt(JtJ[i * n + i]));
  if (gradient_max_norm(p, g) <= gradient_tolerance) { *final_cost = cost; return 0; }
  double x_norm = ambient_norm(p);
  int iter = 0, invalid = 0;
  double diag[9];
  bool reuse_diag = false;
  while (true) {
    if (iter >= max_iterations) break;
    if (radius < min_radius) break;
    iter++;
    // scaled system
    double A[81], gs[9], Js[81];
    for (int i = 0; i < n; i++) {
      gs[i] = g[i] * scale[i];
      for (int j = 0; j < n; j++) Js[i * n + j] = JtJ[i * n + j] * scale[i] * scale[j];
    }
    if (!reuse_diag)
      for (int i = 0; i < n; i++) diag[i] = std::min(std::max(Js[i * n + i], min_diag), max_diag);
    std::memcpy(A, Js, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) A[i * n + i] += diag[i] / radius;
    double y[9], step[9];
    bool ok = lii::spd_solve(A, gs, n, y);
    reuse_diag = true;
    double model_change = 0;
    if (ok) {
      for (int i = 0; i < n; i++) step[i] = -y[i];
      // model_cost_change = -(J s)^T (r + J s / 2) = -s^T g - s^T J^T J s / 2
      double sg = 0, sJs = 0;
      for (int i = 0; i < n; i++) {
        sg += step[i] * gs[i];
        double t = 0;
        for (int j = 0; j < n; j++) t += Js[i * n + j] * step[j];
        sJs += step[i] * t;
      }
      model_change = -sg - 0.5 * sJs;
    }
    if (!ok || !(model_change > 0.0)) {  // invalid step
      if (++invalid >= max_invalid) break;
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      continue;
    }
    invalid = 0;
    double delta[9];
    for (int i = 0; i < n; i++) delta[i] = step[i] * scale[i];
    Problem cand = plus(p, delta);
    double cand_cost = 0;
    if (eval(cand, nullptr, nullptr, &cand_cost) != LII_OK) return -1;
    // parameter tolerance, then function tolerance — both on the candidate, before acceptance
    const double step_norm = ambient_dist(p, cand);
    if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) break;
    const double cost_change = cost - cand_cost;
    if (std::fabs(cost_change) <= function_tolerance * cost) break;
    const double rho = cost_change / model_change;
    if (rho > min_relative_decrease) {
      p = cand;
      x_norm = ambient_norm(p);
      if (eval(p, JtJ, g, &cost) != LII_OK) return -1;
      radius = std::min(max_radius, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3)));
      decrease_factor = 2.0;
      reuse_diag = false;
      if (gradient_max_norm(p, g) <= gradient_tolerance) break;
    } else {
      radius /= decrease_factor;
      decrease_factor *= 2.0;
    }
  }
  *final_cost = cost;
  return iter;
}

}  // namespace

extern "C" int lii_calib_solve_stage(lii_handle h, int32_t stage, lii_calib_result* io) {
  if (!h || !io || stage < 1 || stage > 3) return LII_ERR_INVALID;
  Problem p;
  p.h = h;
  p.stage = stage;
  p.dof = stage == 1 ? 3 : (stage == 2 ? 7 : 9);
  for (int i = 0; i < 6; i++) { p.v[i] = 0; p.bounded[i] = false; p.lo[i] = p.hi[i] = 0; }
  if (stage == 1) {
    p.q = Quat{1, 0, 0, 0};  // LI_init.cpp:318-322
  } else if (stage == 2) {
    p.q = rot_to_quat(io->R_LI);  // Eigen::Quaterniond quat(Rot_Lidar_wrt_IMU), :346
  } else {
    p.q = Quat{1, 0, 0, 0};  // Rot_Init = I, :404-411
    std::memcpy(p.R_LI, io->R_LI, sizeof(p.R_LI));
    for (int i = 0; i < 3; i++) { p.bounded[i] = true; p.lo[i] = -0.01; p.hi[i] = 0.01; }  // :458-461
  }
  double cost = 0;
  int it = minimize(p, &cost);
  if (it < 0) return LII_ERR_HIP;
  io->iterations[stage - 1] = it;
  io->final_cost[stage - 1] = cost;
  double R[9];
  quat_to_rot(p.q, R);
  if (stage == 1) {
    std::memcpy(io->R_LI, R, sizeof(R));
  } else if (stage == 2) {
    std::memcpy(io->R_LI, R, sizeof(R));
    for (int i = 0; i < 3; i++) io->gyro_bias[i] = p.v[i];
    io->time_lag_2 = p.v[3];
  } else {
    const double g[3] = {0, 0, -9.81};
    lii::m3_vec(R, g, io->grav_L0);               // Grav_L0 = R_GL0 * STD_GRAV (:469)
    lii::m3_vec(io->R_LI, p.v, io->acc_bias);     // acc_bias = R_LI * b_aL (:472)
    double t[3];
    lii::m3_vec(io->R_LI, p.v + 3, t);            // T_LI = -R_LI * T_IL (:475)
    for (int i = 0; i < 3; i++) io->T_LI[i] = -t[i];
  }
  return LII_OK;
}
