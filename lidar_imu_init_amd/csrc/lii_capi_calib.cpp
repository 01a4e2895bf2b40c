// libliinit_hip — entry points of the LI_init residual / Jacobian evaluators (include/LI_init/LI_init.h:91-205); the host
// Levenberg-Marquardt around them: lii_calib.cpp, the conditioning chain: lii_li_init.cpp / lii_li_init_dev.hip.
#include "lii_context.h"

using namespace lii_impl;

extern "C" {

// ------------------------------------------------------------------------------------------------ calibration
int lii_calib_set_buffers(lii_handle h, const lii_calib_state* imu, const lii_calib_state* lidar, int32_t n) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !imu || !lidar || n <= 0) return fail(h, LII_ERR_INVALID, "lii_calib_set_buffers: bad arguments");
  static_assert(sizeof(lii_calib_state) == 22 * sizeof(double), "lii_calib_state layout");
  if (n > h->cal.n_cal || !h->cal.d_cal_imu) {
    if (h->cal.d_cal_imu) (void)hipFree(h->cal.d_cal_imu);
    if (h->cal.d_cal_lidar) (void)hipFree(h->cal.d_cal_lidar);
    h->cal.d_cal_imu = h->cal.d_cal_lidar = nullptr;
    HIPCHK(h, dmalloc(&h->cal.d_cal_imu, size_t(n) * 22));
    HIPCHK(h, dmalloc(&h->cal.d_cal_lidar, size_t(n) * 22));
  }
  HIPCHK(h, hipMemcpyAsync(h->cal.d_cal_imu, imu, sizeof(lii_calib_state) * size_t(n), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->cal.d_cal_lidar, lidar, sizeof(lii_calib_state) * size_t(n), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->cal.n_cal = n;
  return LII_OK;
}

int lii_calib_eval(lii_handle h, int32_t stage, const double* params, double* JtJ, double* Jtr, double* cost) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !params || stage < 1 || stage > 3) return fail(h, LII_ERR_INVALID, "lii_calib_eval: bad arguments");
  if (h->cal.n_cal <= 0) return fail(h, LII_ERR_STATE, "lii_calib_eval: no buffers uploaded");
  const int np = stage == 1 ? 9 : (stage == 2 ? 13 : 24);
  const int dof = stage == 1 ? 3 : (stage == 2 ? 7 : 9);
  std::memcpy(h->h_small, params, sizeof(double) * np);
  HIPCHK(h, hipMemcpyAsync(h->cal.d_cal_params, h->h_small, sizeof(double) * np, hipMemcpyHostToDevice, h->stream));
  launch_calib_eval(stage, h->cal.d_cal_imu, h->cal.d_cal_lidar, h->cal.n_cal, h->cal.d_cal_params, h->cal.d_cal_out, h->stream);
  const int n_out = dof * dof + dof + 1;
  HIPCHK(h, hipMemcpyAsync(h->h_small + 64, h->cal.d_cal_out, sizeof(double) * n_out, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const double* o = h->h_small + 64;
  if (JtJ) std::memcpy(JtJ, o, sizeof(double) * dof * dof);
  if (Jtr) std::memcpy(Jtr, o + dof * dof, sizeof(double) * dof);
  if (cost) *cost = o[dof * dof + dof];
  return LII_OK;
}

int lii_li_init_set_device(lii_handle h, int32_t on_device) {
  if (!h) return LII_ERR_INVALID;
  h->cal.li_init_device = on_device != 0;
  return LII_OK;
}

}  // extern "C"
