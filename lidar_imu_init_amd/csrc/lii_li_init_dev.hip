// SURVEY.md section 8(f)4 on the device: the two pieces of the LI-Init conditioning chain with real arithmetic volume,
//   k_zero_phase   LI_Init::zero_phase_filt = Butter_filt forward, reversed, again, reversed (include/LI_init/LI_init.cpp:260-315;
//                  6th-order Butterworth, coefficients LI_init.h:218-224 incl. the asymmetric Coeff_b[4] = 0.0011; 60-sample
//                  reflection padding; recursion from index 7, stopping 60 before the end - SURVEY.md Appendix A8): the four
//                  3-vectors of a CalibState are 12 independent channels; a batch of S sequences is S x 12 lanes, each lane
//                  serial in time (an IIR recursion is a dependent chain), every lane evaluating the sums in the host's order,
//                  so the result is bit-identical to lii_li_init.cpp's butter().
//   k_xcorr        LI_Init::xcorr_temporal_init (:160-193): O(N^2) cross-correlation of the |omega| series; one lane per lag
//                  (2N - 1 lanes), each summing its products in the host's index order; the first maximum in lag order wins,
//                  as the reference's strict `>` does.
// Both exist so that the calibration can be re-run on device-resident buffers during accumulation without a round trip of the
// sequences; at N_s ~ 10^3 the filter is latency (a 1 000-step dependent chain), the correlation is where the device wins.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/liinit_hip.h"
#include "lii_device.h"
#include "lii_launch.h"

namespace lii {
namespace {

constexpr int kExt = 60, kNb = 7;
__constant__ double c_B[7] = {0.000076, 0.000457, 0.001143, 0.001524, 0.0011, 0.000457, 0.000076};
__constant__ double c_A[7] = {1.0000, -4.182389, 7.491611, -7.313596, 4.089349, -1.238525, 0.158428};

// channel ch (0..11) of record r: ang_vel (9..11), linear_vel (12..14), ang_acc (15..17), linear_acc (18..20) of the 22 doubles
__device__ __forceinline__ int chan_off(int ch) { return 9 + ch; }

// One Butter_filt pass of one channel: in[0..n) (stride s_in) -> out[0..n) (stride s_out); x / y = scratch of n + 2 ext.
__device__ void butter_pass(const double* __restrict__ in, int s_in, int n, double* __restrict__ out, int s_out, double* __restrict__ x,
                            double* __restrict__ y, bool reversed_in, bool reversed_out) {
  auto at = [&](int i) { return in[(size_t)(reversed_in ? n - 1 - i : i) * s_in]; };
  const int m = n + 2 * kExt;
  int w = 0;
  for (int i = kExt; i >= 1; i--) x[w++] = at(i);
  for (int i = 0; i < n; i++) x[w++] = at(i);
  for (int i = n - 2; i >= n - 1 - kExt; i--) x[w++] = at(i);
  for (int i = 0; i < m; i++) y[i] = x[i];
  for (int i = kNb; i < m - kExt; i++) {
    double acc = 0;
#pragma unroll
    for (int j = 0; j < kNb; j++) acc += x[i - j] * c_B[j];
#pragma unroll
    for (int j = 1; j < kNb; j++) acc -= y[i - j] * c_A[j];
    y[i] = acc;
  }
  for (int i = 0; i < n; i++) out[(size_t)(reversed_out ? n - 1 - i : i) * s_out] = y[kExt + i];
}

// seqs: n_seq sequences of n records (22 doubles each), back to back; filtered in place (the non-channel members - rot_end,
// timestamp - stay as they are: CalibState::operator= copies only the four 3-vectors, Appendix A7).
__global__ __launch_bounds__(64) void k_zero_phase(double* __restrict__ seqs, int n_seq, int n, double* __restrict__ scratch) {
  const int lane = blockIdx.x * 64 + threadIdx.x;
  if (lane >= n_seq * 12) return;
  const int sq = lane / 12, ch = lane % 12;
  double* base = seqs + (size_t)sq * n * 22 + chan_off(ch);
  const int m = n + 2 * kExt;
  double* x = scratch + (size_t)lane * (2 * m + n);
  double* y = x + m;
  double* tmp = y + m;  // the once-filtered, reversed sequence
  // zero_phase_filt: y1 = butter(in); reverse; y2 = butter(reversed y1); reverse
  butter_pass(base, 22, n, tmp, 1, x, y, false, true);   // tmp = reverse(butter(in))
  butter_pass(tmp, 1, n, base, 22, x, y, false, true);   // base = reverse(butter(tmp))
}

__device__ __forceinline__ double norm3_at(const double* __restrict__ rec) {
  return sqrt(rec[9] * rec[9] + rec[10] * rec[10] + rec[11] * rec[11]);
}
// a[i] = |w_imu|, b[i] = |w_lidar| and their running means (sequential, as the host forms them): one lane
__global__ void k_xcorr_prepare(const double* __restrict__ imu, const double* __restrict__ lidar, int n, double* __restrict__ a,
                                double* __restrict__ b, double* __restrict__ means) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double ma = 0, mb = 0;
  for (int i = 0; i < n; i++) {
    const double va = norm3_at(imu + (size_t)i * 22), vb = norm3_at(lidar + (size_t)i * 22);
    a[i] = va;
    b[i] = vb;
    ma += (va - ma) / (i + 1);
    mb += (vb - mb) / (i + 1);
  }
  means[0] = ma;
  means[1] = mb;
}
// corr[k] for lag = k - (n - 1), one lane per lag, products summed in ascending i
__global__ __launch_bounds__(256) void k_xcorr(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ means,
                                               int n, double* __restrict__ corr) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= 2 * n - 1) return;
  const int lag = k - (n - 1);
  const double ma = means[0], mb = means[1];
  const int i0 = max(0, -lag), i1 = min(n, n - lag);
  double c = 0;
  for (int i = i0; i < i1; i++) c += (a[i] - ma) * (b[i + lag] - mb);
  corr[k] = c;
}
// first maximum in ascending lag order (the reference: `if (c > best)`), result = lag_IMU_wtr_Lidar = -lag
__global__ __launch_bounds__(256) void k_xcorr_argmax(const double* __restrict__ corr, int n, int* __restrict__ out_lag) {
  __shared__ double s_v[256];
  __shared__ int s_k[256];
  const int m = 2 * n - 1;
  double best = -1.7976931348623157e308;
  int bk = 0x7FFFFFFF;
  for (int k = threadIdx.x; k < m; k += 256) {
    const double v = corr[k];
    if (v > best) { best = v; bk = k; }  // a lane walks its lags in ascending order: strict '>' keeps its first maximum
  }
  s_v[threadIdx.x] = best;
  s_k[threadIdx.x] = bk;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double v = s_v[threadIdx.x + s];
      const int k = s_k[threadIdx.x + s];
      if (v > s_v[threadIdx.x] || (v == s_v[threadIdx.x] && k < s_k[threadIdx.x])) { s_v[threadIdx.x] = v; s_k[threadIdx.x] = k; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *out_lag = s_k[0] == 0x7FFFFFFF ? 0 : -(s_k[0] - (n - 1));
}

}  // namespace
}  // namespace lii

using namespace lii;

extern "C" {

int lii_zero_phase_filter(lii_handle h, const lii_calib_state* in, int32_t n_seq, int32_t n, lii_calib_state* out) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !in || !out || n_seq < 1 || n < 62) return lii_internal_fail(h, LII_ERR_INVALID, "lii_zero_phase_filter: bad arguments (n >= 62: the 60-sample reflection)");
  hipStream_t s = lii_internal_stream(h);
  const size_t rec = size_t(n_seq) * size_t(n) * 22;
  const size_t scr = size_t(n_seq) * 12 * (2 * size_t(n + 2 * 60) + size_t(n));
  double *d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(double) * (rec + scr)) != hipSuccess) return lii_internal_fail(h, LII_ERR_HIP, "lii_zero_phase_filter: hipMalloc");
  hipError_t e = hipMemcpyAsync(d, in, sizeof(double) * rec, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_zero_phase, dim3((n_seq * 12 + 63) / 64), dim3(64), 0, s, d, n_seq, n, d + rec);
    e = hipMemcpyAsync(out, d, sizeof(double) * rec, hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d);
  if (e != hipSuccess) return lii_internal_fail(h, LII_ERR_HIP, std::string("lii_zero_phase_filter: ") + hipGetErrorString(e));
  return LII_OK;
}

int lii_xcorr_lag(lii_handle h, const lii_calib_state* imu, const lii_calib_state* lidar, int32_t n, int32_t* lag_imu_wrt_lidar) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !imu || !lidar || !lag_imu_wrt_lidar || n < 1) return lii_internal_fail(h, LII_ERR_INVALID, "lii_xcorr_lag: bad arguments");
  hipStream_t s = lii_internal_stream(h);
  const size_t rec = size_t(n) * 22;
  double* d = nullptr;
  const size_t total = 2 * rec + 2 * size_t(n) + 2 + (2 * size_t(n) - 1) + 2;
  if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(double) * total) != hipSuccess) return lii_internal_fail(h, LII_ERR_HIP, "lii_xcorr_lag: hipMalloc");
  double *d_imu = d, *d_lid = d + rec, *d_a = d + 2 * rec, *d_b = d_a + n, *d_means = d_b + n, *d_corr = d_means + 2;
  int* d_lag = reinterpret_cast<int*>(d_corr + (2 * size_t(n) - 1));
  hipError_t e = hipMemcpyAsync(d_imu, imu, sizeof(double) * rec, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_lid, lidar, sizeof(double) * rec, hipMemcpyHostToDevice, s);
  int lag = 0;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_xcorr_prepare, dim3(1), dim3(64), 0, s, d_imu, d_lid, n, d_a, d_b, d_means);
    hipLaunchKernelGGL(k_xcorr, dim3((2 * n - 1 + 255) / 256), dim3(256), 0, s, d_a, d_b, d_means, n, d_corr);
    hipLaunchKernelGGL(k_xcorr_argmax, dim3(1), dim3(256), 0, s, d_corr, n, d_lag);
    e = hipMemcpyAsync(&lag, d_lag, sizeof(int), hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d);
  if (e != hipSuccess) return lii_internal_fail(h, LII_ERR_HIP, std::string("lii_xcorr_lag: ") + hipGetErrorString(e));
  *lag_imu_wrt_lidar = lag;
  return LII_OK;
}

}  // extern "C"
