// On-device 24-state solve of the iterated Kalman update — the host algebra of lii_iekf_update moved into one
// single-wavefront kernel per iteration so that a scan registration needs ONE host synchronisation.
// Reference code replaced (src/laserMapping.cpp): K_1 = (H^T R^-1 H (+) 0 + P^-1)^-1 and the state update :1080-1087,
// convergence test :1093-1096, rematch schedule :1102-1106, covariance update :1109-1131;
// StatesGroup boxplus/boxminus include/common_lib.h:126-154; Exp/Log include/so3_math.h:61-107.
// The gain is evaluated in an algebraically equal but better-conditioned form than the reference's (see below); the
// host-driven update (lii_hostmath.h, LII_TEST=host_solve) keeps the literal form, and both are held to the oracle in the tests.
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <math.h>

#include "lii_device.h"
#include "lii_launch.h"

namespace lii {

constexpr int N = 24;
constexpr int H = 12;    // columns of the measurement Jacobian (pose + extrinsic)
constexpr int LDH = 13;  // padded leading dimension of the 12-column LDS tiles

// The gain needs only the first 12 columns of K_1 = (H^T R^-1 H (+) 0 + P^-1)^-1.  With G = H^T R^-1 H (12 x 12, PSD),
// E = [I_12; 0] and P = [P11 P12; P21 P22]:
//     K_1 = (P^-1 + E G E^T)^-1 = (I + P E G E^T)^-1 P,     I + P E G E^T = [ I + P11 G   0 ]
//                                                                            [   P21 G     I ]
//  => K_1[:, :12] = [ M P11 ; P21 - P21 G M P11 ] = P[:, :12] (I + G P11)^-1,   M = (I + P11 G)^-1
//     (eigenvalues of P11 G are >= 0: always regular), i.e.  K_1[:, :12]^T = (I + P11 G)^-1 P[:12, :]  for symmetric P, G.
// This is the reference's formula (src/laserMapping.cpp:1081) in exact arithmetic, with ONE 12-step elimination instead of
// two 24 x 24 inversions (the reference's own route loses ~cond(P) eps).
//
// Gauss-Jordan with partial pivoting on a 12-row system, register resident: every lane owns one column of the augmented
// matrix (12 of A + the right-hand sides); the multiplier column is broadcast with v_readlane, all register indices are
// static (fully unrolled).
__device__ __forceinline__ double readlane_f64(double v, int lane) {  // lane is wave-uniform (a constant after unrolling)
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}
// Pivoting (round 4): THRESHOLD pivoting.  Row k stays the pivot row whenever |a_kk| >= 1/4 of the largest magnitude below it in
// its column and only otherwise the largest entry is searched and the rows exchanged (partial pivoting's choice).  Stable like
// partial pivoting (element growth per step bounded by 1 + 1/tau = 5 instead of 2).
// Round 5: this routine - twelve unrolled steps with the exchange code behind every one of them, ~30 KB of straight-line code that
// a launch executes once, every line of it an instruction-cache miss - is now the FALLBACK of gj12_loop below.
__device__ __forceinline__ bool gj12_pivoting(double (&col)[H]) {
#pragma unroll
  for (int k = 0; k < H; k++) {
    // every lane: largest magnitude of its own column below the diagonal (a tree over registers; lane k's is the one that counts)
    double mx = 0.0;
    {
      double t[H];
#pragma unroll
      for (int r = 0; r < H; r++) t[r] = r > k ? fabs(col[r]) : 0.0;
#pragma unroll
      for (int w = 1; w < 16; w <<= 1) {
#pragma unroll
        for (int r = 0; r + w < H; r += 2 * w) t[r] = fmax(t[r], t[r + w]);
      }
      mx = t[0];
    }
    const int keep = __builtin_amdgcn_readlane((fabs(col[k]) >= 0.25 * mx) ? 1 : 0, k);
    double m[H];
#pragma unroll
    for (int r = 0; r < H; r++) m[r] = readlane_f64(col[r], k);  // v_readlane: the multiplier column lands in SGPRs
    if (!keep) {  // (wave-uniform, rare) partial pivoting's choice: the largest magnitude from row k down; rows k and p change places
      int p = k;
      double best = fabs(m[k]);
#pragma unroll
      for (int r = k + 1; r < H; r++) {
        const double v = fabs(m[r]);
        if (v > best) { best = v; p = r; }
      }
      double colp = col[k], mp = m[k];
#pragma unroll
      for (int r = k + 1; r < H; r++)
        if (r == p) { colp = col[r]; mp = m[r]; col[r] = col[k]; m[r] = m[k]; }
      col[k] = colp;
      m[k] = mp;
    }
    if (!(fabs(m[k]) > 0.0)) return false;  // singular (or not a number)
    // 1 / pivot: hardware reciprocal seed + two Newton steps (full double precision, a third of the divide's latency)
    double inv = __builtin_amdgcn_rcp(m[k]);
    inv = fma(fma(-m[k], inv, 1.0), inv, inv);
    inv = fma(fma(-m[k], inv, 1.0), inv, inv);
    const double rowk = col[k] * inv;
#pragma unroll
    for (int r = 0; r < H; r++)
      if (r != k) col[r] -= m[r] * rowk;
    col[k] = rowk;
  }
  return true;
}

// The same elimination as a LOOP of twelve identical steps (round 5).  What made the unrolled form necessary were the register
// indices: step k reads row k as the pivot row.  Here the rows ROTATE instead: position 0 always holds the pivot row, the update of
// row p lands in position p - 1 (the fused multiply-add writes its result one register pair down: the rotation costs no move) and
// the finished pivot row enters at position 11 - after twelve steps every row is back where it started.  Only the lane that
// holds the pivot column changes from step to step, and v_readlane takes the lane from a scalar register.  The body is ~60
// instructions (~0.5 KB): fetched once, it runs out of the instruction cache, where the unrolled form paid a memory round trip for
// every eight instructions (phase stamps: 3.0 us for ~700 instructions - ten cycles per instruction on a wavefront that can
// issue one every four or five).
// Pivoting: none inside the loop - a data-dependent row exchange is what cannot be expressed with static register indices - but
// the elimination WATCHES ITS OWN GROWTH: every lane keeps the largest magnitude its column takes on the way (twelve maxima per
// step, off the critical chain readlane -> reciprocal -> multiply-add), and the result stands only if the columns of A never
// outgrew 2^8 x max(1, max |A|) - the quantity the backward error of an elimination is proportional to (Wilkinson) - and
// every pivot was a number.  Otherwise the saved input goes through gj12_pivoting.  I + P11 G with P11, G positive semi-definite
// has its spectrum in [1, inf): measured on LIO sequences (hall, corridor: tools/gj_growth.py) the pivot-free growth stays
// below 10 where the 1/4-threshold test of round 4 would have exchanged rows on nine scans of ten.
// (kGrowthMax = 2^8: eight bits of the 53 - <= 3e-14 x cond(A) on the gain, far inside the 1e-4 the covariance is held to)
// The watch works on the HIGH WORDS of the doubles read as SINGLE-precision numbers (sign | the exponent field's upper eight bits | its
// lower three and twenty mantissa bits): their magnitudes order like the doubles' - the word is a monotone function of |x| - so |.| is the operand modifier of v_max3_f32 and a step's twelve magnitudes cost six instructions (round 5, first form: twelve
// v_and_b32 + six v_max3_u32; twelve v_max_f64 before that: tools/ubench/solve_ubench.hip - a double-precision max or multiply-add
// occupies the pipe for 7 - 10 cycles, a 32-bit operation for 4), and "x 2^8" is an addition to the double's exponent field.  A word
// whose upper exponent bits are all ones - a double beyond 2^1016, infinity, not-a-number - reads as a single-precision NaN, which the
// maximum SKIPS: such a value cannot pass the elimination unnoticed all the same, a pivot that was zero or not a number turns its
// whole row - every lane's last entry - into not-a-number for good, and gj12_loop looks at that entry once at the end.
__device__ __forceinline__ float max3_abs(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max3_pos(float a, float b, float c) {  // (operands >= 0)
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// max(g, the largest magnitude of the column) on the high words: six instructions
__device__ __forceinline__ float col_absmax_hi(const double (&col)[H], float g) {
  float t[H];
#pragma unroll
  for (int r = 0; r < H; r++) t[r] = __int_as_float(__double2hiint(col[r]));
  const float a = max3_abs(t[0], t[1], t[2]), b = max3_abs(t[3], t[4], t[5]), c = max3_abs(t[6], t[7], t[8]), d = max3_abs(t[9], t[10], t[11]);
  return max3_pos(max3_pos(a, b, c), d, g);
}
// Every lane watches ITS column against ITS column's own scale, max(1, largest magnitude at the start): stricter than the growth
// factor of the whole matrix (a column's scale is at most the matrix's) and needs no reduction over the lanes - the verdict is one
// ballot at the end.  (The columns of the right-hand side are watched as well: they only ride along, but a column that outgrows
// its start by 2^8 says the multipliers were large.)
__device__ __forceinline__ bool gj12_loop(double (&col)[H]) {
  float g = col_absmax_hi(col, 0.f);
  const unsigned int bound = max(__float_as_uint(g), 0x3FF00000u /* 1.0 */) + (8u << 20);  // kGrowthMax = 2^8
#pragma nounroll
  for (int k = 0; k < H; k++) {
    double m[H];
#pragma unroll
    for (int r = 0; r < H; r++) m[r] = readlane_f64(col[r], k);  // the pivot column (lane k), pivot row first
    // 1 / pivot: hardware reciprocal seed + two Newton steps (full double precision, a third of the divide's latency)
    double inv = __builtin_amdgcn_rcp(m[0]);
    inv = fma(fma(-m[0], inv, 1.0), inv, inv);
    inv = fma(fma(-m[0], inv, 1.0), inv, inv);
    const double rowk = col[0] * inv;
#pragma unroll
    for (int r = 1; r < H; r++) col[r - 1] = fma(-m[r], rowk, col[r]);
    col[H - 1] = rowk;
    g = col_absmax_hi(col, g);
  }
  // (g >= 0: its bits order like its value.  The last pivot row: not-a-number if any pivot was zero or not a number; idle lanes hold zeros)
  // The growth watch reads the doubles' high words as floats, and v_max3_f32 skips an operand that reads as a float NaN - which is what
  // the high word of a double beyond 2^1016, of an infinity or of a NaN looks like.  One INTEGER look at the end closes that hole
  // (ADVICE r5): a double is finite iff its exponent field is not all ones, and an overflow never turns back into a finite number.
  unsigned int top = 0u;
#pragma unroll
  for (int r = 0; r < H; r++) top = max(top, (unsigned int)__double2hiint(col[r]) & 0x7FFFFFFFu);
  return __all(__float_as_uint(g) <= bound && top < 0x7FF00000u);
}
// pivoted: the elimination went through gj12_pivoting (wave-uniform)
__device__ __forceinline__ bool gj12(double (&col)[H], bool& pivoted) {
  pivoted = false;
  double keep[H];
#pragma unroll
  for (int r = 0; r < H; r++) keep[r] = col[r];
  if (__builtin_expect(gj12_loop(col), 1)) return true;  // (wave-uniform)
#pragma unroll
  for (int r = 0; r < H; r++) col[r] = keep[r];
  pivoted = true;
  return gj12_pivoting(col);
}

__device__ void d_m3_mul(const double* A, const double* B, double* C) {
#pragma clang fp contract(off)  // this unit is built with -ffp-contract=fast; the SO(3) helpers keep the reference's rounding (so3_math.h)
  double t[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) t[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
  for (int e = 0; e < 9; e++) C[e] = t[e];
}
__device__ void d_m3t_mul(const double* A, const double* B, double* C) {
#pragma clang fp contract(off)  // this unit is built with -ffp-contract=fast; the SO(3) helpers keep the reference's rounding (so3_math.h)
  double t[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) t[3 * r + c] = A[r] * B[c] + A[3 + r] * B[3 + c] + A[6 + r] * B[6 + c];
  for (int e = 0; e < 9; e++) C[e] = t[e];
}
// Exp(v1,v2,v3) — so3_math.h:61-79 (identity below 1e-5)
__device__ void d_so3_exp(double v1, double v2, double v3, double* R) {
#pragma clang fp contract(off)  // this unit is built with -ffp-contract=fast; the SO(3) helpers keep the reference's rounding (so3_math.h)
  double n = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
  for (int e = 0; e < 9; e++) R[e] = (e % 4 == 0) ? 1.0 : 0.0;
  if (n > 0.00001) {
    double ax[3] = {v1 / n, v2 / n, v3 / n};
    double K[9] = {0, -ax[2], ax[1], ax[2], 0, -ax[0], -ax[1], ax[0], 0};
    double cK[9], KK[9];
    double s = sin(n), c1 = 1.0 - cos(n);
    for (int e = 0; e < 9; e++) cK[e] = c1 * K[e];  // `(1.0 - cos) * K * K` = ((1 - cos) K) K, so3_math.h:73
    d_m3_mul(cK, K, KK);
    for (int e = 0; e < 9; e++) R[e] = (R[e] + s * K[e]) + KK[e];
  }
}
__device__ void d_so3_log(const double* R, double* out) {
#pragma clang fp contract(off)  // this unit is built with -ffp-contract=fast; the SO(3) helpers keep the reference's rounding (so3_math.h)
  double tr = R[0] + R[4] + R[8];
  double theta = (tr > 3.0 - 1e-6) ? 0.0 : acos(0.5 * (tr - 1));
  double K[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double f = (fabs(theta) < 0.001) ? 0.5 : (0.5 * theta / sin(theta));
  out[0] = f * K[0]; out[1] = f * K[1]; out[2] = f * K[2];
}

// One iteration's solve + state update + schedule.  ne = the 91 reduced normal-equation scalars of this pass.
// ONE workgroup of kSolveThreads = 256 lanes (round 1: a single wavefront, ~11 us of serial algebra per iteration).  Every
// global input is fetched in ONE parallel batch into LDS (the control block and `ne` were just written by other kernels, so
// each dependent global read would cost a full memory round trip); then
//   phase A  G (78 lanes) | boxminus: the two rotation logarithms + the vector blocks (second wavefront) - side by side
//   phase A2 A = I + P11 G (144 lanes)
//   phase B  the 12-step register-resident Gauss-Jordan (first wavefront - inherently serial) | u = H^T z - G vec (second)
//   phase C  solution (24 lanes), schedule, boxplus, and on the stopping iteration K H (288 entries) and the covariance
//            (576 entries, 12 MACs each) over all 256 lanes.
#ifdef LII_SOLVE_TRACE
#define LII_TS(k) do { if (threadIdx.x == 0) s_ts[k] = wall_clock64(); } while (0)
#else
#define LII_TS(k)
#endif
constexpr int kSolveThreads = 256;
// The iteration that ends the loop tells the host so through the mapped result block: every lane's stores to `res` have been
// acknowledged (system-scope fence by every lane, then the barrier) before the flag goes out, so a host that sees
// done == seq sees the result.
// Round 6: the result block lives in HOST memory, which the device does not cache - its stores travel straight out, and "acknowledged"
// is what s_waitcnt vmcnt(0) waits for.  A system-scope release fence does more: it writes back every dirty line of the XCD's L2 -
// whatever the launches before left there - which nobody on the host will read (1 - 2 us of the stopping pass; the device-side stores
// of this kernel become visible to the next launch at the kernel boundary as always).  So: every lane waits for its own stores, the
// barrier collects them, one relaxed store raises the flag behind them (posted writes of one device to host memory stay in order).
// (every store into the result block: system scope, i.e. written through to the host as it is issued)
template <class T>
__device__ __forceinline__ void res_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void publish_done(IekfResult* res, int seq) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&res->done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ne_src: where the 91 sums come from - a functor called by every lane AFTER the other loads have been issued; it leaves the sums
// in s_ne[0 .. 90] (LDS; the barrier below makes them visible) and returns 0, or - when they could not be had (uniform) - the code the
// host finds in IekfResult::singular: 3 = the exchange between the ranks timed out, 4 = the sums of this launch's own summing
// workgroups did not arrive in time.  Both end the update with LII_ERR_COMM; neither traps.
template <class NeSrc>
__device__ __forceinline__ void iekf_solve_body(IekfCtrl* c, IekfResult* res, NeSrc ne_src) {
#ifdef LII_SOLVE_TRACE
  __shared__ long long s_ts[16];
#endif
  LII_TS(0);
  __shared__ double s_cov[N * N];  // prior covariance (row-major, stride 24)
  __shared__ double G[H * LDH];    // H^T R^-1 H
  __shared__ double A[H * LDH];    // I + P11 G
  __shared__ double K1c[N * LDH];  // K_1[:, :12]
  __shared__ double vec[N], sol[N], s_u[H], s_KH[N * H];
  __shared__ double s_x[N * N];    // (I - K H) P before it is symmetrised (stopping iteration only)
  __shared__ double s_ne[96], s_st[36], s_prop[36];
  __shared__ int s_int[12];
  __shared__ int s_ok;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  {
    const int* ci = &c->max_it;  // max_it, imu_en, it, search_next, stop, rematch_num, converged, searches, effect_num, singular, seq, plan_mask
    if (tid < 12) s_int[tid] = ci[tid];
    if (tid >= 160 && tid < 196) { s_st[tid - 160] = c->st[tid - 160]; s_prop[tid - 160] = c->prop[tid - 160]; }
    const double* cov = c->st + 36;
    for (int e = tid; e < N * N; e += kSolveThreads) s_cov[e] = cov[e];
  }
  if (const int why = ne_src(s_ne)) {  // (uniform) the sums could not be had
    __syncthreads();
    if (threadIdx.x == 0) { c->stop = 1; c->singular = why; res_store(&res->singular, why); res_store(&res->it, 0); }
    publish_done(res, s_int[10]);
    return;
  }
  __syncthreads();
  if (s_int[4]) return;  // EKF_stop_flg already set: this pass is not due (uniform)
  LII_TS(1);
  const int max_it = s_int[0], it = s_int[2], search_now = s_int[3], rematch0 = s_int[5], searches0 = s_int[7];
  // upper-triangle index t -> (i, j), packed i * 16 + j
  static const unsigned char kTri[78] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 51, 52, 53, 54, 55, 56, 57, 58, 59, 68, 69, 70, 71, 72, 73, 74, 75, 85, 86, 87, 88, 89, 90, 91, 102, 103, 104, 105, 106, 107, 119, 120, 121, 122, 123, 136, 137, 138, 139, 153, 154, 155, 170, 171, 187};
  // ---- phase A: A = I + P11 G on the first 144 lanes, straight from the 78 sums (G[k][j] = sum number tri(k, j)); meanwhile the
  // last wavefront spreads G out for the later phases and takes the vector blocks of vec = state_propagat (-) state.  (The two
  // rotation logarithms of vec - 1.2 us of serial fp64 on two lanes - run beside the elimination, on the second wavefront.)
  if (tid < H * H) {
    const int i = tid / H, j = tid % H;
    double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < H; k++) {
      const int a = k < j ? k : j, b = k < j ? j : k;
      s += s_cov[i * N + k] * s_ne[a * H - (a * (a - 1)) / 2 + (b - a)];
    }
    A[i * LDH + j] = s;
  } else if (wave == 3) {
    for (int t = lane; t < 78; t += 64) {
      const int i = kTri[t] >> 4, j = kTri[t] & 15;
      G[i * LDH + j] = s_ne[t];
      G[j * LDH + i] = s_ne[t];
    }
    if (lane >= 32 && lane < 50) {
      const int q = lane - 32, blk = q / 3, i = q % 3;
      const int sto = blk == 0 ? 9 : (blk == 1 ? 21 : (blk == 2 ? 24 : (blk == 3 ? 27 : (blk == 4 ? 30 : 33))));
      const int vo = blk == 0 ? 3 : (blk == 1 ? 9 : (blk == 2 ? 12 : (blk == 3 ? 15 : (blk == 4 ? 18 : 21))));
      vec[vo + i] = s_prop[sto + i] - s_st[sto + i];
    }
  }
  if (tid == 0) s_ok = 1;
  __syncthreads();
  LII_TS(2);
  LII_TS(3);
  // ---- phase B: K_1[:, :12]^T = A^-1 P[:12, :]  (A = I + P11 G;  K_1[:, :12] = P[:, :12] (I + G P11)^-1 and G, P symmetric).
  // Gauss-Jordan on [A | P[:12, :]]: lanes 0..11 of the first wavefront hold the columns of A, lanes 12..35 the 24 columns of
  // P[:12, :]; afterwards lane 12 + j holds row j of K_1[:, :12].  One elimination gives the gain directly - no inverse to
  // store, no product - and no subtraction of nearly equal terms: the algebraically equal form [M P11 ; P21 - P21 G M P11]
  // cancels catastrophically once the pose block of P has collapsed (1e-8) next to velocity / bias blocks of order 1, the
  // regime of the LIO phase.  Meanwhile the second wavefront forms u = H^T R^-1 z - G vec[:12] (then solution = K_1[:, :12] u
  // + vec, the reference's K z + vec - K H vec[:12] regrouped).
  if (wave == 0) {
    double col[H];
#pragma unroll
    for (int r = 0; r < H; r++) col[r] = lane < H ? A[r * LDH + lane] : (lane < H + N ? s_cov[r * N + (lane - H)] : 0.0);
    bool pivoted;
    const bool ok = gj12(col, pivoted);
    if (lane == 0) s_ok = ok ? (pivoted ? 2 : 1) : 0;
    if (lane >= H && lane < H + N) {
#pragma unroll
      for (int r = 0; r < H; r++) K1c[(lane - H) * LDH + r] = col[r];
    }
  } else if (wave == 1) {
    if (lane < 2) {
      const int o = lane * 12, so = lane * 6;  // rot_end / offset_R_L_I
      double R[9];
      d_m3t_mul(s_st + o, s_prop + o, R);
      d_so3_log(R, vec + so);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (one wavefront: its LDS accesses stay in order)
    if (lane < H) {
      double s2 = s_ne[78 + lane];
#pragma unroll
      for (int k = 0; k < H; k++) s2 -= G[lane * LDH + k] * vec[k];
      s_u[lane] = s2;
    }
  }
  __syncthreads();
  if (!s_ok) {  // uniform
    if (tid == 0) { c->stop = 1; c->singular = 1; res_store(&res->singular, 1); }
    publish_done(res, s_int[10]);
    return;
  }
  LII_TS(4);
  // ---- phase C
  if (tid < N) {
    double s2 = vec[tid];
#pragma unroll
    for (int k = 0; k < H; k++) s2 += K1c[tid * LDH + k] * s_u[k];
    sol[tid] = s2;
  }
  __syncthreads();
  LII_TS(8);
  // schedule (uniform: every lane evaluates it from LDS)
  const double rn = sqrt(sol[0] * sol[0] + sol[1] * sol[1] + sol[2] * sol[2]);
  const double tn = sqrt(sol[3] * sol[3] + sol[4] * sol[4] + sol[5] * sol[5]);
  const int converged = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);
  int rematch = rematch0, search = 0;
  if (converged || ((rematch == 0) && (it == (max_it - 2)))) { search = 1; rematch++; }
  const int do_cov = (rematch >= 2 || (it == max_it - 1));
  // the next pass searches but the host did not enqueue a k-NN launch for it: park the loop and say so (IekfCtrl::plan_mask)
  // ... or the host did not enqueue the next pass at all (bit 16 + k: the launches of pass k are there)
  const unsigned int pm = (unsigned int)s_int[11];
  const bool pass_there = it + 1 >= 16 || ((pm >> (16 + it + 1)) & 1u), knn_there = it + 1 >= 16 || ((pm >> (it + 1)) & 1u);
  const int parked = !do_cov && (!pass_there || (search && !knn_there));
  // state += solution : the two rotations on two lanes of the first wavefront, the vector blocks on 18 more; on the stopping
  // iteration the other three wavefronts start on K H = K_1[:, :12] G (needed for the covariance only) right away
  if (wave == 0) {
    if (lane < 2) {
      double E[9], Rn[9];
      const int o = lane == 0 ? 0 : 12;  // rot_end / offset_R_L_I
      const int so = lane == 0 ? 0 : 6;
      d_so3_exp(sol[so], sol[so + 1], sol[so + 2], E);
      d_m3_mul(s_st + o, E, Rn);
      for (int e = 0; e < 9; e++) c->st[o + e] = Rn[e];
      if (do_cov)
        for (int e = 0; e < 9; e++) res_store(&res->st[o + e], Rn[e]);
    } else if (lane >= 8 && lane < 26) {
      const int q = lane - 8;  // 0..17 : six 3-vectors
      const int blk = q / 3, i = q % 3;
      const int sto = blk == 0 ? 9 : (blk == 1 ? 21 : (blk == 2 ? 24 : (blk == 3 ? 27 : (blk == 4 ? 30 : 33))));
      const int soo = blk == 0 ? 3 : (blk == 1 ? 9 : (blk == 2 ? 12 : (blk == 3 ? 15 : (blk == 4 ? 18 : 21))));
      const double v = s_st[sto + i] + sol[soo + i];
      c->st[sto + i] = v;
      if (do_cov) res_store(&res->st[sto + i], v);
    } else if (lane >= 32 && lane < 32 + N) {
      c->solution[lane - 32] = sol[lane - 32];
    }
    if (lane == 63) {
      c->converged = converged;
      c->rematch_num = rematch;
      c->search_next = search;
      c->effect_num = (int)s_ne[90];
      c->it = it + 1;
      c->searches = searches0 + (search_now ? 1 : 0);
      if (it < 16) c->search_log[it] = search_now | (s_ok == 2 ? 2 : 0);
      if (parked) {
        c->stop = 2;
        res_store(&res->parked_it, it + 1);
        res_store(&res->parked_search, search);
      }
      if (do_cov) {
        c->stop = 1;
        res_store(&res->it, it + 1);
        res_store(&res->searches, searches0 + (search_now ? 1 : 0));
        res_store(&res->effect_num, (int)s_ne[90]);
        res_store(&res->converged, converged);
        res_store(&res->singular, 0);
      }
    }
  } else if (do_cov) {
    // Round 6: every one of the three wavefronts takes EIGHT ROWS of K H and then the same eight rows of
    // state.cov = (I - K H) cov = cov - (K H) cov[0:12, :]   (:1111-1114, all operands already in LDS) - a row of the product needs its
    // own row of K H and nothing else, so no workgroup barrier stands between the two and both run in the shadow of the state update on
    // the first wavefront (round 5: K H here, then barrier - product over 256 lanes - barrier behind the state update: 1.4 us of the
    // stopping pass; the sums of every entry are formed in the same order as before).
    const int r0 = 8 * (wave - 1);
    for (int e = lane; e < 8 * H; e += 64) {
      const int r = r0 + e / H, cc = e % H;
      double s2 = 0;
#pragma unroll
      for (int k = 0; k < H; k++) s2 += K1c[r * LDH + k] * G[k * LDH + cc];
      s_KH[r * H + cc] = s2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = lane; e < 8 * N; e += 64) {
      const int r = r0 + e / N, cc = e % N;
      double s2 = s_cov[r * N + cc];
      for (int k = 0; k < H; k++) s2 -= s_KH[r * H + k] * s_cov[k * N + cc];
      s_x[r * N + cc] = s2;
    }
  }
  LII_TS(9);
  if (do_cov) {  // uniform
    __syncthreads();
    // the posterior leaves as its symmetric part: the gain above relies on P = P^T, and whatever asymmetry rounding puts into
    // (I - K H) P must not feed back into the next scan's gain (it compounds otherwise: the pose block of P grew to 0.4 within
    // 200 scans of a LIO run; the literal two-inversion algebra stays at 4e-5)
    double* covw = c->st + 36;
    for (int e = tid; e < N * N; e += kSolveThreads) {
      const int r = e / N, cc = e % N;
      const double v = 0.5 * (s_x[e] + s_x[cc * N + r]);
      covw[e] = v;
      res_store(&res->st[36 + e], v);
    }
    if (tid < 91) res_store(&res->ne[tid], s_ne[tid]);
    if (tid >= 96 && tid < 112) {
      const int q = tid - 96;
      res_store(&res->search_log[q], (q < it) ? c->search_log[q] : (q == it ? (search_now | (s_ok == 2 ? 2 : 0)) : 0));
    }
    publish_done(res, s_int[10]);
  } else if (parked) {  // uniform
    publish_done(res, s_int[10] | kLoopParked);
  }
#ifdef LII_SOLVE_TRACE
  __syncthreads();
  LII_TS(10);
  if (threadIdx.x < 11) { if (it == 0) res->ts0[threadIdx.x] = s_ts[threadIdx.x]; else res->ts[threadIdx.x] = s_ts[threadIdx.x]; }
#endif
}

// Final reduction of the per-workgroup partials (91 workgroups, one output each, fixed summation order - the same order as
// k_reduce91: final_sum_row in lii_device.h) FUSED with the solve (workgroup 91).  Used when no all-reduce sits between the two
// (single GPU, or the node-local mailbox); one launch less per iteration.
// How the 91 sums reach the solver (round 4): every summing workgroup publishes its double as two self-describing 8-byte words
// {tag, half of the bits} (one store each, straight to the memory side), tag = the number of this pass; the solver - which
// has meanwhile fetched the control block and the prior covariance - polls the 182 words with one wavefront until every tag
// is this pass's.  No flag, no fence, no ticket: a word either carries the tag, and then its payload, or it does not
// (MI355X guide, "data-tagged granules": one producer -> consumer hand-off ~1 us).  Round 3: atomic store of the sum, wait
// for its acknowledgement, returning ticket atomic, and the LAST arriver - not known in advance, so nothing could be
// fetched ahead - loaded everything it needed behind the ticket: three more dependent round trips per pass.
// The solver's code is executed exactly once per launch, on one workgroup, behind kernels that have swept the L2: every instruction
// line is a cold miss served by HBM, one after the other (round 4: ~30 KB of straight-line code; round 5: the elimination is a loop
// and its pivoting form sits in a cold branch, ~10 KB are hot).  The workgroup therefore reads its own code as DATA first - 64-byte
// lines ahead of this point, all requests in flight together while it waits for the sums anyway - so that the instruction fetches
// that follow hit in the XCD's L2.  The value is folded into a word nobody reads (the loads must not be dropped; the wait for them
// lands where the caller consumes the return value).
// How far it may read is bounded by the address of solve_code_end(), a function defined behind the last kernel of this unit: a read
// never leaves the code between this point and that symbol, whatever the toolchain does with the order of the functions - a marker
// that ends up in FRONT of this point turns the prefetch off (round 4 read 24 KB ahead on the strength of the
// definition order alone).
__device__ __attribute__((noinline, used)) void solve_code_end();
__device__ __forceinline__ int warm_code() {
  const char* pc = reinterpret_cast<const char*>(__builtin_amdgcn_s_getpc() & ~63ull);
  const char* end = reinterpret_cast<const char*>(&solve_code_end);
  int acc = 0;
  if (threadIdx.x < 192) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const char* a = pc + (size_t)(threadIdx.x + 192 * u) * 64;
      if (a + 64 <= end) acc ^= *reinterpret_cast<const volatile int*>(a);
    }
  }
  return acc;
}
__device__ __forceinline__ unsigned int pass_tag(int seq, int it) { return ((unsigned int)seq << 6) ^ (unsigned int)(it + 1); }
__global__ __launch_bounds__(kSolveThreads) void k_reduce_solve(const double* __restrict__ partials, int n_blocks, int stride,
                                                                 unsigned long long* __restrict__ gran, IekfCtrl* c, IekfResult* res,
                                                                 MailboxView mb, const int* __restrict__ flag_count) {
  __shared__ double s_w[kSolveThreads / 64];
  const int stop = c->stop, seq = c->seq, it = c->it;  // (one request: the three words share a line)
  const int t = blockIdx.x;
  if (t < kNormalEq) {
    // (n_blocks = the workgroups of the fit launch: its points are dealt out in chunks, every workgroup may hold some.  The row is
    // requested together with the flags, not behind the branch on them: one round trip at the head of the launch instead of two)
    const double acc = final_sum_row<kSolveThreads>(partials + (size_t)t * stride, n_blocks, s_w);
    if (stop) return;  // the loop has ended: nothing is published
    if (threadIdx.x == 0 && !(mb.test_drop_sum && t == 5)) {
      const unsigned int tag = pass_tag(seq, it);
      const unsigned long long b = (unsigned long long)__double_as_longlong(acc);
      __hip_atomic_store(gran + 2 * t, ((unsigned long long)tag << 32) | (b & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gran + 2 * t + 1, ((unsigned long long)tag << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (stop) return;  // (the solver; it may set the flag itself, after every workgroup above has read it and published)
  // How many queries the scan's search passes left unfinished (the two most recent search launches' counts) rides to the host beside the
  // result: the next scan's launch plan puts k_complete_listed behind its search launches when it was more than the fit launch's
  // completion workgroups take.  (a lane that has nothing to do with the sums; its store is waited for in publish_done like every other)
  // (every enqueued search launch empties the OTHER slot: behind pass `it` the two slots hold this pass's count and 0, or - a pass
  // without a search launch - what an earlier pass left; the maximum over the scan's passes is kept in gran[240])
  if (threadIdx.x == kSolveThreads - 1 && flag_count) {
    const int now = max(flag_count[0], flag_count[1]);
    const int before = it == 0 ? 0 : (int)gran[240];
    const int um = max(now, before);
    gran[240] = (unsigned long long)um;
    res_store(&res->unfinished, um);
  }
  const unsigned int tag = pass_tag(seq, it);
  const int warm = warm_code();
  iekf_solve_body(c, res, [&](double* s_ne) {
    __shared__ int s_mb_ok;
    if (threadIdx.x < 64) {  // one wavefront collects the sums (lane l: sums l and l + 64) and runs the exchange between the ranks
      const int l = threadIdx.x;
      const bool two = l + 64 < kNormalEq;
      const int a = 2 * l, b = two ? 2 * (l + 64) : 2 * l;
      unsigned long long g0, g1, g2, g3;
      unsigned int spins = 0;
      bool arrived = true;
      const long long t0 = wall_clock64();
      for (;;) {
        g0 = __hip_atomic_load(gran + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g1 = __hip_atomic_load(gran + a + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g2 = __hip_atomic_load(gran + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g3 = __hip_atomic_load(gran + b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = (unsigned int)(g0 >> 32) == tag && (unsigned int)(g1 >> 32) == tag && (unsigned int)(g2 >> 32) == tag &&
                        (unsigned int)(g3 >> 32) == tag;
        if (__all(ok)) break;
        // A summing workgroup that has not published yet may simply not be running: the 92 workgroups of this launch share the
        // device with whatever else is queued on it.  The wait is bounded by TIME (2 s of the 100 MHz clock, looked at every 1024
        // polls), and running out ends the update with an error the caller can handle (LII_ERR_COMM) - rounds 4 - 5 trapped here
        // after 2^24 polls, which took the process down with the launch.
        if ((++spins & 1023u) == 0u && wall_clock64() - t0 > mb.handoff_ticks) { arrived = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      double v0 = __longlong_as_double((long long)((g1 << 32) | (g0 & 0xFFFFFFFFull)));
      double v1 = two ? __longlong_as_double((long long)((g3 << 32) | (g2 & 0xFFFFFFFFull))) : 0.0;
      bool ok = true;
      if (arrived && (mb.slots || mb.peers)) ok = mailbox_allreduce(mb, v0, v1);  // several ranks: the sums meet the others' in the node-local mailbox
      s_ne[l] = v0;
      if (two) s_ne[l + 64] = v1;
      if (l == 0) s_mb_ok = !arrived ? 4 : (ok ? 0 : 3);
    }
    if (warm == 0x5EED5EED) s_ne[95] = 0.0;  // (s_ne[91 .. 95] is padding)
    __syncthreads();
    return s_mb_ok;
  });
#ifdef LII_GAP_TRACE
  __syncthreads();
  if (threadIdx.x == 0 && c->stop == 1) gran[200] = wall_clock64();  // (measurement builds: see gap_trace, lii_scan.hip)
#endif
}

// (defined BEHIND k_reduce_solve on purpose: that kernel reads its own code ahead, see warm_code)
__global__ __launch_bounds__(kSolveThreads) void k_iekf_solve(IekfCtrl* c, const double* __restrict__ ne, IekfResult* res) {
  iekf_solve_body(c, res, [&](double* s_ne) {
    const int tid = threadIdx.x;
    if (tid >= 64 && tid < 64 + 91) s_ne[tid - 64] = ne[tid - 64];
    return 0;
  });
}

// (the end of the code warm_code may read: see there)
__device__ __attribute__((noinline, used)) void solve_code_end() { asm volatile("s_nop 0"); }

// The same exchange for the host-driven single pass (lii_iekf_iterate): in place on the 91 sums; a timeout poisons them.
__global__ __launch_bounds__(64) void k_mailbox_allreduce(double* out, MailboxView mb) {
  const int l = threadIdx.x;
  double v0 = l < kNormalEq ? out[l] : 0.0, v1 = l + 64 < kNormalEq ? out[l + 64] : 0.0;
  const bool ok = mailbox_allreduce(mb, v0, v1);
  if (l < kNormalEq) out[l] = ok ? v0 : __builtin_nan("");
  if (l + 64 < kNormalEq) out[l + 64] = ok ? v1 : __builtin_nan("");
}

void launch_mailbox_allreduce(double* out91, const MailboxView& mb, hipStream_t s) {
  hipLaunchKernelGGL(k_mailbox_allreduce, dim3(1), dim3(64), 0, s, out91, mb);
}
void launch_reduce_solve(const RegistrationBuffers& rb, unsigned long long* gran, IekfCtrl* c, IekfResult* res, const MailboxView& mb,
                         hipStream_t s, int epoch) {
  const int bound = rb.shard_world > 1 ? (rb.n + rb.shard_world - 1) / rb.shard_world + 1 : rb.n;  // as launch_fit_reduce
  int nb = (bound + kBlock - 1) / kBlock;
  if (nb < 1) nb = 1;
  nb += completion_blocks(epoch);  // (the columns of the fit launch's completion workgroups: as many as THAT launch had)
  hipLaunchKernelGGL(k_reduce_solve, dim3(kNormalEq + 1), dim3(kSolveThreads), 0, s, rb.partials, nb, rb.partial_stride, gran, c, res, mb, rb.flag_count);
}
// A parked loop goes on with the launches the host has put behind this one (plan_mask, as IekfCtrl::plan_mask).
__global__ void k_loop_resume(IekfCtrl* c, unsigned int plan_mask) {
  c->stop = 0;
  c->plan_mask = plan_mask;
}
void launch_loop_resume(IekfCtrl* c, unsigned int plan_mask, hipStream_t s) { hipLaunchKernelGGL(k_loop_resume, dim3(1), dim3(1), 0, s, c, plan_mask); }
void launch_iekf_solve(IekfCtrl* c, const double* ne, IekfResult* res, hipStream_t s) {
  hipLaunchKernelGGL(k_iekf_solve, dim3(1), dim3(kSolveThreads), 0, s, c, ne, res);
}

}  // namespace lii
