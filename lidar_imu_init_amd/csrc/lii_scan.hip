// Scan pre-processing of libliinit_hip for gfx950: de-skew (IMU back-propagation / constant-velocity model) and the voxel-grid
// down-sampling, i.e. everything between the arrival of a scan and its registration.  Reference code replaced:
//   k_time_extent / k_undistort_imu / _cv ... src/IMU_Processing.hpp:390-414 and :246-266
//   k_voxel_* / k_vhash_*                    pcl::VoxelGrid::filter call site src/laserMapping.cpp:917-919 (hashed filter; the
//                                            sample-sort form lives in lii_vsort.hip)
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <math.h>
#include <stdint.h>

#include "lii_device.h"
#include "lii_launch.h"

namespace lii {

// ------------------------------------------------------------------------------------------------
// undistortion
// order-preserving float -> uint map (so that integer atomics give float min / max)
__device__ __forceinline__ unsigned int f2ord(float f) {
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// extent[0] = (ord(t_min) << 32) | index of the first point with that time ; extent[1] = ord(t_max)
// grid-stride over a small grid, wave shuffle + LDS reduction, ONE pair of atomics per block
// copy_to != nullptr: the scan is adopted from a caller-owned device buffer on the way (lii_scan_set_device) - one pass
// and one launch instead of a copy followed by the reduction.
__global__ __launch_bounds__(256) void k_time_extent(const float4* __restrict__ pts, int n, unsigned long long* __restrict__ extent,
                                                     unsigned long long* __restrict__ extent_next, float4* __restrict__ copy_to,
                                                     const uint4* __restrict__ ctrl_src, uint4* __restrict__ ctrl_dst, int ctrl_vec) {
  __shared__ unsigned long long smn[4], smx[4];
  // first kernel of a scan: one extra workgroup pulls the update's control block + IMU pose table out of the caller-side
  // pinned buffer (ctrl_vec 16-byte words over PCIe, ~3 us beside the others' work instead of an H2D copy submission of
  // ~10 us before the launch)
  const int n_scan_blocks = gridDim.x - (ctrl_vec > 0 ? 1 : 0);
  if ((int)blockIdx.x == n_scan_blocks) {
    for (int i0 = threadIdx.x; i0 < ctrl_vec; i0 += 256 * 4) {  // four PCIe reads in flight per lane
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = i0 + 256 * u < ctrl_vec ? ctrl_src[i0 + 256 * u] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (i0 + 256 * u < ctrl_vec) ctrl_dst[i0 + 256 * u] = v[u];
    }
    return;
  }
  // the accumulators ping-pong between two buffers: this launch re-arms the one the NEXT scan will reduce into (nobody reads
  // it any more: its consumers belonged to the previous scan), which saves a separate initialisation launch per scan
  if (blockIdx.x == 0 && threadIdx.x == 0) { extent_next[0] = ~0ull; extent_next[1] = 0ull; }
  unsigned long long mn = ~0ull, mx = 0;
  // (two points per lane and trip, both loads in flight together: the launch is sized for two points per lane)
  const int stride = n_scan_blocks * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 2 * stride) {
    const int i1 = i + stride;
    const float4 p0 = pts[i];
    const float4 p1 = pts[i1 < n ? i1 : i];
    if (copy_to) {
      copy_to[i] = p0;
      if (i1 < n) copy_to[i1] = p1;
    }
    const unsigned int o0 = f2ord(p0.w), o1 = f2ord(p1.w);
    const unsigned long long a0 = ((unsigned long long)o0 << 32) | (unsigned)i;
    const unsigned long long a1 = i1 < n ? (((unsigned long long)o1 << 32) | (unsigned)i1) : ~0ull;
    mn = a0 < mn ? a0 : mn;
    mn = a1 < mn ? a1 : mn;
    mx = (unsigned long long)o0 > mx ? (unsigned long long)o0 : mx;
    mx = (i1 < n && (unsigned long long)o1 > mx) ? (unsigned long long)o1 : mx;
  }
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long a = __shfl_xor(mn, off), b = __shfl_xor(mx, off);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) { mn = smn[w] < mn ? smn[w] : mn; mx = smx[w] > mx ? smx[w] : mx; }
    atomicMin(&extent[0], mn);
    atomicMax(&extent[1], mx);
  }
}
__device__ __forceinline__ float ord2f(unsigned int o) {
  unsigned int u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __uint_as_float(u);
}

// Exp(ang_vel, dt) — include/so3_math.h:37-59 — applied to a vector: R v with Rodrigues, R built explicitly
__device__ __forceinline__ void exp_so3(const double w[3], double dt, double R[9]) {
  double n = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (n > 0.0000001) {
    double ax = w[0] / n, ay = w[1] / n, az = w[2] / n;
    double ang = n * dt;
    double s = sin(ang), c1 = 1.0 - cos(ang);
    // K = skew(axis).  The reference writes `(1.0 - cos) * K * K`, which C++ evaluates as ((1 - cos) K) K: the scalar is
    // rounded into K before the product (so3_math.h:48-53; pinned by tests/test_oracle_math_pinned.py)
    double K[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
    double cK[9], KK[9];
#pragma unroll
    for (int e = 0; e < 9; e++) cK[e] = c1 * K[e];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) KK[3 * r + c] = cK[3 * r] * K[c] + cK[3 * r + 1] * K[3 + c] + cK[3 * r + 2] * K[6 + c];
#pragma unroll
    for (int e = 0; e < 9; e++) R[e] = ((e % 4 == 0) ? 1.0 : 0.0) + s * K[e] + KK[e];
  } else {
#pragma unroll
    for (int e = 0; e < 9; e++) R[e] = (e % 4 == 0) ? 1.0 : 0.0;
  }
}
struct UndistArg {
  double endR[9], endp[3], RLI[9], TLI[3];
};

// One back-propagation step of point p with pose-table head `h` (src/IMU_Processing.hpp:398-411)
__device__ __forceinline__ void backprop_once(const double* __restrict__ head /*22 doubles*/, double t, const UndistArg& u,
                                              double p[3]) {
  double dt = t - head[0];
  const double* acc = head + 1;
  const double* gyr = head + 4;
  const double* vel = head + 7;
  const double* pos = head + 10;
  const double* rot = head + 13;
  double E[9], Ri[9];
  exp_so3(gyr, dt, E);
  mat3_mul(rot, E, Ri);
  double Pi[3];
#pragma unroll
  for (int a = 0; a < 3; a++) Pi[a] = pos[a] + vel[a] * dt + 0.5 * acc[a] * dt * dt;
  double q[3], w[3], e[3], o[3];
  mat3_vec(u.RLI, p, q);
#pragma unroll
  for (int a = 0; a < 3; a++) q[a] += u.TLI[a];
  mat3_vec(Ri, q, w);
#pragma unroll
  for (int a = 0; a < 3; a++) w[a] = w[a] + Pi[a] - u.endp[a];
  mat3t_vec(u.endR, w, e);
#pragma unroll
  for (int a = 0; a < 3; a++) e[a] -= u.TLI[a];
  mat3t_vec(u.RLI, e, o);
  p[0] = o[0]; p[1] = o[1]; p[2] = o[2];
}

// Bounding box of the points a 256-lane workgroup holds (non-finite points excluded; order-preserving uints): wave shuffle +
// LDS reduction; afterwards lanes 0..2 hold the min / max of axis threadIdx.x in lo[0] / hi[0].  Every lane must call it.
__device__ __forceinline__ void block_bbox_reduce(unsigned int (&lo)[3], unsigned int (&hi)[3]) {
  __shared__ unsigned int s_lo[4][3], s_hi[4][3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    for (int off = 32; off > 0; off >>= 1) {
      unsigned int x = __shfl_xor(lo[a], off), y = __shfl_xor(hi[a], off);
      lo[a] = min(lo[a], x);
      hi[a] = max(hi[a], y);
    }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) { s_lo[wave][a] = lo[a]; s_hi[wave][a] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    lo[0] = min(min(s_lo[0][a], s_lo[1][a]), min(s_lo[2][a], s_lo[3][a]));
    hi[0] = max(max(s_hi[0][a], s_hi[1][a]), max(s_hi[2][a], s_hi[3][a]));
  }
}
__device__ __forceinline__ void bbox_point(float x, float y, float z, unsigned int (&lo)[3], unsigned int (&hi)[3]) {
  if (isfinite(x) && isfinite(y) && isfinite(z)) {
    const unsigned int ox = f2ord(x), oy = f2ord(y), oz = f2ord(z);
    lo[0] = min(lo[0], ox); hi[0] = max(hi[0], ox);
    lo[1] = min(lo[1], oy); hi[1] = max(hi[1], oy);
    lo[2] = min(lo[2], oz); hi[2] = max(hi[2], oz);
  }
}
// The de-skew kernels leave the bounding box of their output behind for the voxel filter that follows (a pass of its own
// over the scan costs a launch): one row of 8 uints per workgroup (min xyz, -, max xyz, -), reduced by k_voxel_keys.
// (Folding the rows with atomics instead costs the kernel ~5 us: 400 workgroups x 6 atomics on one cache line.)
__device__ __forceinline__ void deskew_bbox(float4 q, bool in_range, unsigned int* __restrict__ rows, int row) {
  if (!rows) return;  // uniform
  unsigned int lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0, 0, 0};
  if (in_range) bbox_point(q.x, q.y, q.z, lo, hi);
  block_bbox_reduce(lo, hi);
  if (threadIdx.x < 3) {
    rows[row * 8 + threadIdx.x] = lo[0];
    rows[row * 8 + 4 + threadIdx.x] = hi[0];
  }
}
__device__ __forceinline__ void deskew_bbox(float4 q, bool in_range, unsigned int* __restrict__ rows) { deskew_bbox(q, in_range, rows, (int)blockIdx.x); }

// The table of the hashed voxel filter (its kernels further down): open addressing, ONE 64-byte line per slot - the insert's
// CAS, atomicMin, atomicAdd and member store and the emit's reads all touch that one line (round 3 kept five parallel arrays:
// five lines per voxel).
constexpr unsigned int kVhEmpty = 0xFFFFFFFFu;
constexpr int kVhMembers = 10;  // the first arrivals of a voxel (its creator included) sit in the slot itself
struct __attribute__((aligned(64))) VhSlot {
  // The first 16 bytes decide who OWNS the voxel (min(first, creator)); the emit launch reads them with one 16-byte load and the
  // owner hands the slot back with one 16-byte store, so that a lane which looks at the slot while its owner is already freeing
  // it sees either the state before or the state after - never `first` of one and `creator` of the other (round 4 kept the creator
  // in the second 16 bytes: a lane that saw the first half old and the second half freed took itself for the owner of a
  // two-point voxel without members.  Rare - the two loads of a lane are cycles apart - until four processes time-shared one device).
  unsigned long long key;  // all ones = free.  Fused form: packed absolute voxel coordinates; separate form: the PCL voxel index
  unsigned int first;      // smallest point index among the LATER arrivals of the voxel (atomicMin); kVhEmpty: none
  unsigned int creator;    // the point that created the slot (member 0); kVhEmpty in a free slot
  unsigned int count;      // later arrivals so far (atomicAdd): the voxel holds count + 1 points
  unsigned int head;       // arrivals beyond kVhMembers: a list through next[]; kVhEmpty = none
  unsigned int members[kVhMembers - 1];  // [k - 1]: the k-th later arrival
  unsigned int pad;
};
static_assert(sizeof(VhSlot) == 64, "one cache line per slot");
struct VhTable {
  VhSlot* slots;
  unsigned int mask;        // slots - 1
  unsigned int* slot_of;    // per input point: its voxel's slot (kVhEmpty: a non-finite point, dropped as PCL drops it)
  unsigned int* next;       // per input point: list link of a crowded voxel
  unsigned int* crowded;    // one word: the longest list behind a slot so far
  // A job whose ranks share the filter by VOXEL (lii_comm_set_partition(h, 2), fused form only): part_world > 1, and a voxel whose
  // key does not hash to part_rank is none of this rank's business - its points are not inserted (voxel_rank below).
  unsigned int part_world, part_rank;
  int part_bound;           // ... and the down-sampled cloud of this rank may hold this many points (the launches behind are sized for it)
  int* part_overflow;       // ... or this word (mapped host memory) is set: the update reports LII_ERR_CAPACITY
};
__device__ __forceinline__ unsigned long long vh_mix(unsigned long long hk) {
  hk ^= hk >> 33; hk *= 0xFF51AFD7ED558CCDull; hk ^= hk >> 33;
  return hk;
}
// Which rank of a voxel-partitioned job owns a voxel: the upper half of the key's hash scaled to [0, world) - the lower half picks
// the slot.  A hash, not a range of keys: whatever the scene, every rank gets 1 / world of the voxels (+- a percent), and the
// launches behind the filter take as long as their largest share.
__device__ __forceinline__ unsigned int voxel_rank(unsigned long long key, unsigned int world) {
  return (unsigned int)(((vh_mix(key) >> 32) * (unsigned long long)world) >> 32);
}
// A point joins its voxel: the slot is found - or created - by a CAS on the key.  The point that CREATES the slot (no key there
// before) becomes entry 0 of the member list with a plain store and is done: one atomic for the ~95 % of the points of a leaf
// matched to the sensor that stay alone in their voxel - the atomics of the insert, performed at the memory side, are what the
// de-skew launch waits for (100 k of them ~ 5 us).  Only a LATER arrival pays for the bookkeeping: its arrival number (atomicAdd),
// its member entry, and the smallest index among the later arrivals (atomicMin).  The voxel's owner - its first point in input
// order - is min(creator, `first`), which the emit launch evaluates when every point has arrived; it also puts the members in
// input order.  (Round 3: CAS + atomicMin per point here and a launch of its own, k_vhash_link, to collect the members.)
__device__ __forceinline__ void vh_insert(const VhTable& tb, unsigned long long key, int i) {
  unsigned int slot = (unsigned int)vh_mix(key) & tb.mask;
  bool created;
  for (;;) {  // the table has four times as many slots as there are points: a free slot always turns up
    const unsigned long long prev = atomicCAS(&tb.slots[slot].key, ~0ull, key);
    created = prev == ~0ull;
    if (created || prev == key) break;
    slot = (slot + 1u) & tb.mask;
  }
  VhSlot* s = tb.slots + slot;
  if (created) {
    s->creator = (unsigned)i;
  } else {
    atomicMin(&s->first, (unsigned)i);
    const unsigned int k = atomicAdd(&s->count, 1u) + 1u;
    if (k < (unsigned)kVhMembers) s->members[k - 1u] = (unsigned)i;
    else {
      tb.next[i] = atomicExch(&s->head, (unsigned)i);
      // (the watch that changes the handle over to the sort must see the same number on every rank: a rank that holds a share of
      // the voxels does not feed it)
      if (tb.part_world <= 1u) atomicMax(tb.crowded, k + 1u - (unsigned)kVhMembers);
    }
  }
  tb.slot_of[i] = slot;
}
// The insert of the fused form.  The grid PCL lays over the cloud starts at the cloud's bounding box, which is only known when
// every point has been de-skewed - but which points share a voxel is not: floor(x / leaf) decides it (PCL's index is
// floor(x * inv_leaf) - min_b, the same classes).  So the table is keyed by the absolute voxel coordinates (three 21-bit fields
// around a bias of 2^20: +- 52 km at a 5 cm leaf), and the PCL index of a voxel - the key the output is ordered by on the host -
// is computed by k_vhash_emit<true>, which knows the box.  A point outside the 21-bit range is a voxel of its own.
__device__ __forceinline__ void vhash_insert_abs(const float4 P, int i, float leaf, const VhTable& tb) {
  if (isfinite(P.x) && isfinite(P.y) && isfinite(P.z)) {  // (non-finite points are dropped, as PCL drops them)
    const float inv_leaf = 1.0f / leaf;
    const float fx = floorf(P.x * inv_leaf), fy = floorf(P.y * inv_leaf), fz = floorf(P.z * inv_leaf);
    unsigned long long key;
    if (fabsf(fx) < 1048000.f && fabsf(fy) < 1048000.f && fabsf(fz) < 1048000.f) {
      key = ((unsigned long long)(unsigned)((int)fz + (1 << 20)) << 42) | ((unsigned long long)(unsigned)((int)fy + (1 << 20)) << 21) |
            (unsigned long long)(unsigned)((int)fx + (1 << 20));
    } else {
      key = (1ull << 63) | (unsigned long long)(unsigned)i;
    }
    if (tb.part_world > 1u && voxel_rank(key, tb.part_world) != tb.part_rank) {  // another rank's voxel
      tb.slot_of[i] = kVhEmpty;
      return;
    }
    vh_insert(tb, key, i);
    return;
  }
  tb.slot_of[i] = kVhEmpty;
}

// What the two de-skew kernels share.  `in` is the scan as it arrived - the caller's device buffer (lii_scan_job::scan_dev:
// the de-skew reads it in place, round 3 copied it first) or the library's own (`out`).  sorted != 0: the points come in ascending
// time order (lii_scan_job::scan_sorted - the order the reference's preprocess hands every scan over in): the time-earliest point
// is the first, the sweep ends with the last, and the reduction that used to find them (k_time_extent, a launch of its own)
// is not needed; otherwise `extent` holds its result.  One EXTRA workgroup (ctrl_vec > 0) pulls the update's control block out
// of the caller-side pinned buffer over PCIe beside the others' work - nothing in this launch waits for it.
struct DeskewIo {
  const float4* in;
  float4* out;
  int n;
  int sorted;
  const unsigned long long* extent;
  unsigned int* bbox_rows;
  float leaf;       // FUSE: the voxel filter that follows
  VhTable tb;
  const uint4* ctrl_src;
  uint4* ctrl_dst;
  int ctrl_vec;
#ifdef LII_GAP_TRACE
  unsigned long long* gap;
#endif
};
// Measurement builds (-DLII_GAP_TRACE): how long the device sat idle between the end of the pass that stopped the previous scan's
// loop (k_reduce_solve leaves a wall_clock64 stamp, 100 MHz) and the first instruction of this scan's first kernel - the host's
// share of the scan period, measured where it is paid.  Sum and count are read back by lii_destroy.
__device__ __forceinline__ void gap_trace(const DeskewIo& io) {
#ifdef LII_GAP_TRACE
  if (blockIdx.x == 0 && threadIdx.x == 0 && io.gap) {
    const unsigned long long now = wall_clock64(), prev = io.gap[200];
    // (gaps of 50 us and more - the host did something else between two scans: the bench's bookkeeping between its regions, a
    // first-time allocation - are counted apart: the mean is over the back-to-back scans)
    if (prev != 0ull && now > prev) {
      if (now - prev < 5000ull) { io.gap[201] += now - prev; io.gap[202] += 1ull; }
      else io.gap[203] += 1ull;
    }
    io.gap[200] = 0ull;
  }
#endif
}
__device__ __forceinline__ void pull_ctrl(const DeskewIo& io) {
  for (int i0 = threadIdx.x; i0 < io.ctrl_vec; i0 += 256 * 4) {  // four PCIe reads in flight per lane
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = i0 + 256 * u < io.ctrl_vec ? io.ctrl_src[i0 + 256 * u] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (i0 + 256 * u < io.ctrl_vec) io.ctrl_dst[i0 + 256 * u] = v[u];
  }
}
// The IMU pose table of a scan (K x 22 doubles) BY VALUE in the kernel arguments: the host writes the argument block straight
// into device memory when it enqueues the launch, so the table is there when the first wavefront starts - round 3 staged it in
// pinned memory and had the scan's first kernel pull it over PCIe, which made that kernel a launch the de-skew had to wait for.
// KP = 0: the table is in device memory (poses_g; tables beyond the largest argument block, stand-alone lii_undistort_imu).
template <int KP>
struct PoseTab { double v[KP > 0 ? KP * 22 : 1]; };

// IMU-mode de-skew.  The reference walks the time-sorted cloud backwards over the pose table; per point this
// is: head = the LAST pose index h <= K-2 with offset_time[h] < t (strict) — points with no such head stay
// untouched.  Quirk A3: the time-earliest point (first of the sorted cloud) is re-tested against every earlier
// head after being compensated, so it is compensated once per qualifying head, in descending order.
// FUSE: the de-skewed point goes straight into the table of the hashed voxel filter (vhash_insert_abs) - the filter's own
// insert launch is saved (lii_scan_register, hashed filter).
template <bool FUSE>
__device__ __forceinline__ void deskew_imu_point(const DeskewIo& io, const UndistArg& u, int K, const double* __restrict__ poses, int row, int i, bool in_range,
                                                 float4 P);
template <bool FUSE, int KP>
__global__ __launch_bounds__(256) void k_deskew_imu(DeskewIo io, UndistArg u, int K, const double* __restrict__ poses_g, PoseTab<KP> tab) {
  const int n_scan_blocks = gridDim.x - (io.ctrl_vec > 0 ? 1 : 0);
  gap_trace(io);
  if ((int)blockIdx.x == n_scan_blocks) { pull_ctrl(io); return; }
  const double* __restrict__ poses = KP > 0 ? tab.v : poses_g;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = i < io.n;
  float4 P = in_range ? io.in[i] : make_float4(0, 0, 0, 0);
  deskew_imu_point<FUSE>(io, u, K, poses, (int)blockIdx.x, i, in_range, P);
}
template <bool FUSE>
__device__ __forceinline__ void deskew_imu_point(const DeskewIo& io, const UndistArg& u, int K, const double* __restrict__ poses, int row, int i, bool in_range,
                                                 float4 P) {
  // the head search walks the table's time column backwards: staged in LDS once per workgroup (tables of up to 256 poses;
  // longer ones are walked in global memory), a walk of up to K dependent global loads per point otherwise
  __shared__ double s_time[256];
  const bool staged = K <= 256;
  if (staged) {
    if ((int)threadIdx.x < K) s_time[threadIdx.x] = poses[22 * threadIdx.x];
    __syncthreads();
  }
  bool moved = false;
  if (in_range) {
    double t = P.w / double(1000);
    int h = -1;
    if (staged) {
      for (int k = K - 2; k >= 0; k--)
        if (t > s_time[k]) { h = k; break; }
    } else {
      for (int k = K - 2; k >= 0; k--)
        if (t > poses[22 * k]) { h = k; break; }
    }
    if (h >= 0) {
      double p[3] = {P.x, P.y, P.z};
      backprop_once(poses + 22 * h, t, u, p);
      const bool is_begin = io.sorted ? i == 0 : ((unsigned)(io.extent[0] & 0xFFFFFFFFull) == (unsigned)i);
      if (is_begin) {
        for (int k = h - 1; k >= 0; k--) {
          if (t > poses[22 * k]) {
            // the reference reads the already-overwritten float coordinates back
            p[0] = (double)(float)p[0]; p[1] = (double)(float)p[1]; p[2] = (double)(float)p[2];
            backprop_once(poses + 22 * k, t, u, p);
          }
        }
      }
      P = make_float4((float)p[0], (float)p[1], (float)p[2], P.w);
      moved = true;
    }
    if (moved || io.in != io.out) io.out[i] = P;
  }
  deskew_bbox(P, in_range, io.bbox_rows, row);
  if (FUSE && in_range) vhash_insert_abs(P, i, io.leaf, io.tb);
}

// The pre-armed form (lii_launch.h: DeskewGate).  Every workgroup requests its point, then one lane polls the tag of the record's first
// line past the caches (the host writes it last); the record goes to LDS - through the caches, every line checked against its tag -
// and the de-skew runs from there.  Workgroup 0 is the GATE - first in the grid, so that it runs whatever the launch's size: it
// watches the host's state word for CANCEL, bounds the wait (EXPIRED, by compare-and-swap), tells the others through dev_flag when
// the launch is to end without running, and pulls the update's control block once the record is there (what the extra workgroup
// of k_deskew_imu does).  A launch that does not run has written nothing.
__device__ __forceinline__ unsigned long long gate_load_u64(const void* p) {  // past the caches
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <bool FUSE>
__global__ __launch_bounds__(256) void k_deskew_imu_gated(DeskewIo io, DeskewGate gate) {
  __shared__ unsigned long long s_verdict;
  __shared__ int s_bad, s_k;
  __shared__ double s_rec[7 * kGateLines];
  const unsigned long long armed = (gate.seq << 2) | kGateArmed, go = (gate.seq << 2) | kGateGo, cancel = (gate.seq << 2) | kGateCancel;
  const bool gate_wg = blockIdx.x == 0;
  const int row = (int)blockIdx.x - 1;
  const int i = row * (int)blockDim.x + threadIdx.x;
  const bool in_range = !gate_wg && i < io.n;
  // (requested before the wait where the scan has been in device memory all along; a scan announced while its transfer was under way -
  // late_load - is read behind the wait: the host only writes the record once the transfer is complete)
  float4 P = (in_range && !gate.late_load) ? io.in[i] : make_float4(0, 0, 0, 0);
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    unsigned long long verdict = go;
    unsigned int spins = 0;
    for (;;) {
      // line 0's tag: the record is complete.  (That one tag also carries K in its top byte - the host writes it last, so the workgroup
      // knows how many lines to fetch without another load past the caches.)
      const unsigned long long t0w = gate_load_u64(gate.rec + 7);
      if ((t0w & 0x00FFFFFFFFFFFFFFull) == go) { s_k = (int)(t0w >> 56); break; }
      ++spins;
      if (gate_wg) {
        const unsigned long long w = gate_load_u64(&gate.host->word);
        if (w == cancel) { verdict = cancel; break; }
        if (w == armed && (spins & 31u) == 0u && wall_clock64() - t0 > gate.timeout_ticks) {
          // nobody came: EXPIRED - unless the host moves the word in this very moment, then its verdict stands (GO: the record follows)
          unsigned long long expect = armed;
          if (__hip_atomic_compare_exchange_strong(&gate.host->word, &expect, (gate.seq << 2) | kGateExpired, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE,
                                                   __HIP_MEMORY_SCOPE_SYSTEM) || expect == cancel) { verdict = cancel; break; }
        }
      } else if ((spins & 3u) == 0u) {
        if (gate_load_u64(gate.dev_flag) == cancel) { verdict = cancel; break; }
        // (the gate workgroup bounds the wait; this bound - four times as long - only ends a launch whose gate never ran)
        if ((spins & 255u) == 0u && wall_clock64() - t0 > 4 * gate.timeout_ticks) { verdict = cancel; break; }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (gate_wg && verdict == cancel) __hip_atomic_store(gate.dev_flag, cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    s_verdict = verdict;
    s_bad = 0;
  }
  __syncthreads();
  if (s_verdict != go) return;
  if (gate_wg) {
#ifdef LII_GAP_TRACE
    gap_trace(io);
#endif
    if (io.ctrl_vec > 0) pull_ctrl(io);
    return;
  }
  if (in_range && gate.late_load) P = io.in[i];
  // the record -> LDS
  const int K = s_k;
  const int n_lines = (25 + 22 * (K < 2 ? 2 : (K > kGateMaxPoses ? kGateMaxPoses : K)) + 6) / 7;
  for (int e = threadIdx.x; e < 8 * n_lines; e += 256) {
    const double v = gate.rec[e];
    if ((e & 7) == 7) { if (((unsigned long long)__double_as_longlong(v) & 0x00FFFFFFFFFFFFFFull) != go) s_bad = 1; }
    else s_rec[(e >> 3) * 7 + (e & 7)] = v;
  }
  __syncthreads();
  if (s_bad) {  // (uniform; rare) a line came out of a cache as an older record left it: everything again, past the caches
    for (int e = threadIdx.x; e < 8 * n_lines; e += 256)
      if ((e & 7) != 7) s_rec[(e >> 3) * 7 + (e & 7)] = __longlong_as_double((long long)gate_load_u64(gate.rec + e));
    __syncthreads();
  }
  UndistArg u;
#pragma unroll
  for (int e = 0; e < 9; e++) { u.endR[e] = s_rec[1 + e]; u.RLI[e] = s_rec[13 + e]; }
#pragma unroll
  for (int e = 0; e < 3; e++) { u.endp[e] = s_rec[10 + e]; u.TLI[e] = s_rec[22 + e]; }
  deskew_imu_point<FUSE>(io, u, K, s_rec + 25, row, i, i < io.n, P);
}

struct CvArg {
  double omega[3], vel[3], endR[9];
};
// CV-mode de-skew (src/IMU_Processing.hpp:246-266).  The time-earliest point is skipped (quirk A3).
template <bool FUSE>
__global__ __launch_bounds__(256) void k_deskew_cv(DeskewIo io, CvArg a) {
  const int n_scan_blocks = gridDim.x - (io.ctrl_vec > 0 ? 1 : 0);
  if ((int)blockIdx.x == n_scan_blocks) { pull_ctrl(io); return; }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = i < io.n;
  float4 P = in_range ? io.in[i] : make_float4(0, 0, 0, 0);
  if (in_range) {
    const bool is_begin = io.sorted ? i == 0 : ((unsigned)(io.extent[0] & 0xFFFFFFFFull) == (unsigned)i);
    if (!is_begin) {
      // the sweep's end: the largest time stamp (sorted: the last point's)
      const float t_end = io.sorted ? io.in[io.n - 1].w : ord2f((unsigned)io.extent[1]);
      double end_off = t_end / double(1000);
      double dt_j = end_off - P.w / double(1000);
      double R[9];
      exp_so3(a.omega, -dt_j, R);
      double rv[3];
      mat3t_vec(a.endR, a.vel, rv);
      double p[3] = {P.x, P.y, P.z}, o[3];
      mat3_vec(R, p, o);
#pragma unroll
      for (int c = 0; c < 3; c++) o[c] = o[c] + (-rv[c]) * dt_j;
      P = make_float4((float)o[0], (float)o[1], (float)o[2], P.w);
      io.out[i] = P;
    } else if (io.in != io.out) {
      io.out[i] = P;
    }
  }
  deskew_bbox(P, in_range, io.bbox_rows);
  if (FUSE && in_range) vhash_insert_abs(P, i, io.leaf, io.tb);
}

// ------------------------------------------------------------------------------------------------
// voxel-grid down-sampling (PCL VoxelGrid restatement)
// mm[0..2] = ord(min xyz), mm[3..5] = ord(max xyz)
__global__ __launch_bounds__(256) void k_voxel_minmax(const float4* __restrict__ pts, int n, unsigned int* __restrict__ mm,
                                                      unsigned int* __restrict__ mm_next) {
  // grid-stride over a small grid; re-arms the ping-pong partner buffer for the next scan (see k_time_extent)
  if (blockIdx.x == 0 && threadIdx.x < 6) mm_next[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
  unsigned int lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0, 0, 0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = pts[i];
    bbox_point(p.x, p.y, p.z, lo, hi);
  }
  block_bbox_reduce(lo, hi);
  if (threadIdx.x < 3) {  // ONE set of atomics per workgroup (<= 256 of them)
    atomicMin(&mm[threadIdx.x], lo[0]);
    atomicMax(&mm[3 + threadIdx.x], hi[0]);
  }
}

struct VoxelArg {
  float inv_leaf;
  int min_b[3];
  int mul[3];
  int identity;  // PCL's int32 index-overflow guard tripped: output = input
};
// Derives the voxel-grid parameters from the min/max reduction ON THE DEVICE (no host round trip), in every thread of the
// key kernel (six loads + a few flops — cheaper than a launch of its own):
// PCL VoxelGrid::applyFilter — bounding box, (dx*dy*dz) > INT32_MAX -> "Leaf size is too small" -> identity copy.
__device__ __forceinline__ VoxelArg voxel_prepare(const unsigned int* __restrict__ mm, float leaf) {
  VoxelArg v;
  v.inv_leaf = 1.0f / leaf;
  v.identity = 0;
  for (int a = 0; a < 3; a++) { v.min_b[a] = 0; v.mul[a] = 0; }
  if (mm[0] != 0xFFFFFFFFu) {  // at least one finite point
    float mn[3], mx[3];
    for (int a = 0; a < 3; a++) {
      unsigned int lo = mm[a], hi = mm[3 + a];
      unsigned int ul = (lo & 0x80000000u) ? (lo & 0x7FFFFFFFu) : ~lo, uh = (hi & 0x80000000u) ? (hi & 0x7FFFFFFFu) : ~hi;
      mn[a] = __uint_as_float(ul);
      mx[a] = __uint_as_float(uh);
    }
    long long dx = (long long)((mx[0] - mn[0]) * v.inv_leaf) + 1, dy = (long long)((mx[1] - mn[1]) * v.inv_leaf) + 1,
              dz = (long long)((mx[2] - mn[2]) * v.inv_leaf) + 1;
    if (dx * dy * dz > 2147483647LL) {
      v.identity = 1;
    } else {
      int div_b[3];
      for (int a = 0; a < 3; a++) {
        v.min_b[a] = (int)floorf(mn[a] * v.inv_leaf);
        div_b[a] = (int)floorf(mx[a] * v.inv_leaf) - v.min_b[a] + 1;
      }
      v.mul[0] = 1; v.mul[1] = div_b[0]; v.mul[2] = div_b[0] * div_b[1];
    }
  }
  return v;
}
__global__ __launch_bounds__(256) void k_voxel_keys(const float4* __restrict__ pts, int n, const unsigned int* __restrict__ mm,
                                                    const unsigned int* __restrict__ bbox_rows, int n_rows, float leaf,
                                                    unsigned long long* __restrict__ keys, unsigned int* __restrict__ pcl_keys,
                                                    int* __restrict__ filtered,
                                                    unsigned long long* __restrict__ samples, int sample_width) {
  __shared__ unsigned int s_mm[8];
  if (n_rows > 0) {  // the box arrives as one row per de-skew workgroup: every workgroup folds them for itself (a few KB from L2)
    unsigned int lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0, 0, 0};
    for (int r = threadIdx.x; r < n_rows; r += 256) {
      const uint4 a = *reinterpret_cast<const uint4*>(bbox_rows + r * 8), b = *reinterpret_cast<const uint4*>(bbox_rows + r * 8 + 4);
      lo[0] = min(lo[0], a.x); lo[1] = min(lo[1], a.y); lo[2] = min(lo[2], a.z);
      hi[0] = max(hi[0], b.x); hi[1] = max(hi[1], b.y); hi[2] = max(hi[2], b.z);
    }
    block_bbox_reduce(lo, hi);
    if (threadIdx.x < 3) { s_mm[threadIdx.x] = lo[0]; s_mm[3 + threadIdx.x] = hi[0]; }
    __syncthreads();
    mm = s_mm;
  }
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const VoxelArg v = voxel_prepare(mm, leaf);
  if (i == 0) *filtered = v.identity ? 0 : 1;
  float4 p = pts[i];
  unsigned long long key = kVoxDropKey;  // non-finite points sort last and are dropped
  unsigned int pcl = 0x7FFFFFFFu;
  if (v.identity) {
    key = pcl = (unsigned)i;  // every point is its own voxel, in input order
  } else if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
    int i0 = (int)(floorf(p.x * v.inv_leaf) - (float)v.min_b[0]);
    int i1 = (int)(floorf(p.y * v.inv_leaf) - (float)v.min_b[1]);
    int i2 = (int)(floorf(p.z * v.inv_leaf) - (float)v.min_b[2]);
    pcl = (unsigned)(i0 * v.mul[0] + i1 * v.mul[1] + i2 * v.mul[2]);
    key = pcl;
  }
  keys[i] = key;
  pcl_keys[i] = pcl;
  if (samples) {  // the sort's splitter samples (lii_vsort.hip): one jittered position per stratum of `sample_width` points
    const unsigned int j = (unsigned int)i / (unsigned int)sample_width, lo = j * (unsigned int)sample_width;
    unsigned int hsh = j * 2654435761u;
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    const unsigned int w = min((unsigned int)sample_width, (unsigned int)n - lo);
    if (lo + hsh % w == (unsigned int)i) samples[j] = (key << kVoxIdxBits) | (unsigned long long)(unsigned int)i;
  }
}
// ------------------------------------------------------------------------------------------------
// The voxel grid without a sort (the default path; the sample sort of lii_vsort.hip stays for crowded voxels and
// LII_VOXEL_FILTER=sort).
// PCL's filter sorts the points by voxel index only to bring the points of a voxel together and to emit the voxels in index
// order; the centroids themselves depend on the ORDER OF THE POINTS INSIDE a voxel (float sums, input order), not on the
// order of the voxels.  So the points of a voxel are brought together by a hash table instead, and the voxels leave in the
// order of their first points (the PCL index of every output voxel is kept: lii_scan_download / lii_neighbors_download put
// the reference's order back on the host).  Two launches (round 3: three), the first of which rides in the de-skew launch
// inside lii_scan_register:
//   insert          vh_insert: the voxel's slot in an open-addressing table (CAS on the key); the creator of a slot is done with
//                   that, a later arrival adds its arrival number, member entry and index minimum - k_vhash_insert keyed by
//                   the PCL voxel index, or vhash_insert_abs from the de-skew kernels keyed by absolute voxel coordinates
//   k_vhash_emit    a point is the voxel's owner when it is its first point; output position = number of owners before it -
//                   the owners of a workgroup are counted, the count is published, and the counts of the workgroups below
//                   are collected INSIDE the launch, after the owner has formed its centroid (by then they have long been
//                   published: the exchange hides behind the dependent loads of the centroid; round 3 counted in a launch of
//                   its own).  The owner puts the members in input order (ascending index) by repeated selection, adds the
//                   points in that order - the float additions of PCL's centroid - writes the centroid and frees its slot.
// Deterministic: the slot a voxel lands in and the order in which members arrive vary from run to run, neither reaches the
// output.
__global__ __launch_bounds__(256) void k_vh_clear(VhSlot* slots, unsigned int n_slots) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  VhSlot s;
  s.key = ~0ull; s.first = kVhEmpty; s.creator = kVhEmpty; s.count = 0u; s.head = kVhEmpty; s.pad = 0u;
#pragma unroll
  for (int k = 0; k < kVhMembers - 1; k++) s.members[k] = kVhEmpty;
  slots[i] = s;
}
// Folds the bounding-box rows the de-skew workgroups left behind (every workgroup for itself: a few KB from L2).
__device__ __forceinline__ void fold_bbox_rows(const unsigned int* __restrict__ bbox_rows, int n_rows, unsigned int* s_mm /*[8] LDS*/) {
  unsigned int lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0, 0, 0};
  for (int r = threadIdx.x; r < n_rows; r += 256) {
    const uint4 a = *reinterpret_cast<const uint4*>(bbox_rows + r * 8), b = *reinterpret_cast<const uint4*>(bbox_rows + r * 8 + 4);
    lo[0] = min(lo[0], a.x); lo[1] = min(lo[1], a.y); lo[2] = min(lo[2], a.z);
    hi[0] = max(hi[0], b.x); hi[1] = max(hi[1], b.y); hi[2] = max(hi[2], b.z);
  }
  block_bbox_reduce(lo, hi);
  if (threadIdx.x < 3) { s_mm[threadIdx.x] = lo[0]; s_mm[3 + threadIdx.x] = hi[0]; }
  __syncthreads();
}
__global__ __launch_bounds__(256) void k_vhash_insert(const float4* __restrict__ pts, int n, const unsigned int* __restrict__ mm,
                                                      const unsigned int* __restrict__ bbox_rows, int n_rows, float leaf,
                                                      VhTable tb, int* __restrict__ filtered) {
  __shared__ unsigned int s_mm[8];
  if (n_rows > 0) {
    fold_bbox_rows(bbox_rows, n_rows, s_mm);
    mm = s_mm;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const VoxelArg v = voxel_prepare(mm, leaf);
  if (i == 0) *filtered = v.identity ? 0 : 1;
  const float4 p = pts[i];
  unsigned long long key = ~0ull;  // non-finite points are dropped
  if (v.identity) {
    key = (unsigned)i;  // every point is its own voxel
  } else if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
    const int i0 = (int)(floorf(p.x * v.inv_leaf) - (float)v.min_b[0]);
    const int i1 = (int)(floorf(p.y * v.inv_leaf) - (float)v.min_b[1]);
    const int i2 = (int)(floorf(p.z * v.inv_leaf) - (float)v.min_b[2]);
    key = (unsigned)(i0 * v.mul[0] + i1 * v.mul[1] + i2 * v.mul[2]);  // < 2^31 (the overflow guard)
  }
  if (key != ~0ull) vh_insert(tb, key, i);
  else tb.slot_of[i] = kVhEmpty;
}
// exclusive scan over the 256 lanes of a workgroup of 0 / 1 flags; *total = flags set in the workgroup
__device__ __forceinline__ unsigned int block_rank_of_flag(bool f, unsigned int* s_w /*[4] LDS*/, unsigned int* total) {
  const unsigned long long b = __ballot(f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_w[wave] = (unsigned)__popcll(b);
  __syncthreads();
  unsigned int before = 0;
  for (int w = 0; w < wave; w++) before += s_w[w];
  *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  return before + (unsigned)__popcll(b & ((1ull << lane) - 1ull));
}
// ABS: the table was filled by the fused form (absolute voxel coordinates): the box arrives here (one row per de-skew
// workgroup, folded by every workgroup for itself), the PCL index of a voxel is computed from its first point, and PCL's
// overflow guard (the grid would have more than 2^31 voxels: "leaf size too small", the cloud passes unfiltered) is applied
// here - every point then leaves as it is, in input order.
// *crowded (written by vh_insert when a point goes to a list): the host reads it behind the filter and takes the sort path from
// then on when voxels hold dozens of points (a large leaf) - the owner orders the members by repeated selection, quadratic in
// their number.
template <bool ABS>
__global__ __launch_bounds__(256) void k_vhash_emit(const float4* __restrict__ pts, int n, VhTable tb, unsigned long long* __restrict__ counts,
                                                    unsigned int epoch, float4* __restrict__ out, int* __restrict__ n_out,
                                                    unsigned int* __restrict__ pcl_out, const unsigned int* __restrict__ bbox_rows,
                                                    int n_rows, float leaf, int* __restrict__ filtered, int test_late) {
  __shared__ unsigned int s_w[4], s_sum[12];
  const int tid = threadIdx.x, i = blockIdx.x * blockDim.x + tid;
  // everything a lane needs of its own point is requested at once: the point, its slot, and (dependent) the slot's line
  const bool in_range = i < n;
  const float4 p0 = in_range ? pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned int slot = in_range ? tb.slot_of[i] : kVhEmpty;
  VoxelArg v;
  v.identity = 0;
  if (ABS) {
    __shared__ unsigned int s_mm[8];
    fold_bbox_rows(bbox_rows, n_rows, s_mm);
    v = voxel_prepare(s_mm, leaf);
    if (i == 0) *filtered = v.identity ? 0 : 1;
  }
  VhSlot* const sl = tb.slots + (slot != kVhEmpty ? slot : 0u);
  uint4 hd = make_uint4(0u, 0u, kVhEmpty, 0u), m0 = make_uint4(kVhEmpty, kVhEmpty, kVhEmpty, kVhEmpty), m1 = m0;
  uint2 m2 = make_uint2(kVhEmpty, kVhEmpty);
  unsigned long long key = 0;
  if (slot != kVhEmpty) {  // the slot's line in four requests
    const uint4* line = reinterpret_cast<const uint4*>(sl);
    const uint4 a = line[0];  // key, first, creator: ONE request decides the ownership (see VhSlot)
    key = ((unsigned long long)a.y << 32) | a.x;
    hd.x = a.z;
    m0.x = a.w;
    const uint4 b = line[1];  // count, head, members 1..2
    hd.y = b.x; hd.z = b.y;
    m0.y = b.z; m0.z = b.w;
    const uint4 c = line[2], d = line[3];  // members 3..6, 7..9 (+ padding)
    m0.w = c.x; m1 = make_uint4(c.y, c.z, c.w, d.x); m2 = make_uint2(d.y, d.z);
  }
  // (a voxel-partitioned job whose cloud passes unfiltered - PCL's overflow guard - still shares it by voxel: every point of a
  // voxel of this rank leaves on its own)
  const bool ident_part = ABS && v.identity && tb.part_world > 1u;
  const bool first = slot != kVhEmpty && (ident_part || min(hd.x, m0.x) == (unsigned)i);  // the owner: the first point of the voxel in input order
  unsigned int total;
  const unsigned int rank = block_rank_of_flag(first, s_w, &total);
  // The workgroup's count goes out BEFORE it looks at anybody else's (prefix_below, lii_device.h: the exchange of the counts inside
  // the launch, placement-independent).  LII_TEST=emit_late: every seventh workgroup holds its word back until it is done, so that
  // the workgroups above it have to count its block themselves - the path a launch takes whose workgroups are not all resident.
  const bool hold = test_late != 0 && (blockIdx.x % 7u) == 3u && !(ABS && v.identity && !ident_part);
  if (tid == 0 && !hold) __hip_atomic_store(counts + blockIdx.x, ((unsigned long long)epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (ABS && v.identity && !ident_part) {  // (uniform) the cloud passes unfiltered; the voxels' owners still hand their slots back
    if (in_range) { out[i] = p0; pcl_out[i] = (unsigned)i; }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *n_out = n;
    if (first) {
      uint4* line = reinterpret_cast<uint4*>(sl);
      line[0] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, kVhEmpty, kVhEmpty);
      line[1] = make_uint4(0u, kVhEmpty, kVhEmpty, kVhEmpty);
    }
    return;
  }
  float4 cen = make_float4(0.f, 0.f, 0.f, 0.f);
  if (first) {
    const unsigned int cnt = ident_part ? 1u : hd.y + 1u;  // points of the voxel, this one included
    float sx = __fadd_rn(0.f, p0.x), sy = __fadd_rn(0.f, p0.y), sz = __fadd_rn(0.f, p0.z), st = __fadd_rn(0.f, p0.w);
    unsigned int m[kVhMembers] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y};
#pragma unroll
    for (int k = 0; k < kVhMembers; k++)
      if ((unsigned)k >= cnt) m[k] = kVhEmpty;  // (entries beyond the count are stale: the slot is not wiped member by member)
    if (cnt > 512u) {
      // Safety valve for a scan on which voxels turn crowded in the middle of a run (the first scan of a leaf is probed, the
      // following ones watched: lii_downsample): ordering hundreds of members by repeated selection would take tens of
      // milliseconds.  They are added in list order - the centroid is then right to rounding, not bit for bit.
#pragma unroll
      for (int k = 0; k < kVhMembers; k++) {
        if (m[k] == (unsigned)i || m[k] == kVhEmpty) continue;
        const float4 p = pts[m[k]];
        sx = __fadd_rn(sx, p.x); sy = __fadd_rn(sy, p.y); sz = __fadd_rn(sz, p.z); st = __fadd_rn(st, p.w);
      }
      for (unsigned int j = hd.z; j != kVhEmpty; j = tb.next[j]) {
        if (j == (unsigned)i) continue;
        const float4 p = pts[j];
        sx = __fadd_rn(sx, p.x); sy = __fadd_rn(sy, p.y); sz = __fadd_rn(sz, p.z); st = __fadd_rn(st, p.w);
      }
    } else if (cnt > 1u) {
      // the members in input order: every round takes the smallest index above the last one taken (the owner's own index is
      // the smallest of all, so it never comes up again) - from the slot's own members (registers) and, for a crowded voxel,
      // from the list behind them (walked again every round: slow and rare)
      const unsigned int head = cnt > (unsigned)kVhMembers ? hd.z : kVhEmpty;
      unsigned int last = (unsigned)i;
      for (unsigned int r = 1; r < cnt; r++) {
        unsigned int best = kVhEmpty;
#pragma unroll
        for (int k = 0; k < kVhMembers; k++) best = (m[k] > last && m[k] < best) ? m[k] : best;
        for (unsigned int j = head; j != kVhEmpty; j = tb.next[j]) best = (j > last && j < best) ? j : best;
        const float4 p = pts[best];
        sx = __fadd_rn(sx, p.x); sy = __fadd_rn(sy, p.y); sz = __fadd_rn(sz, p.z); st = __fadd_rn(st, p.w);
        last = best;
      }
    }
    const float c = (float)cnt;
    // a single-point voxel reproduces the point exactly (x / 1.0f == x), which is also what the identity path needs
    cen = make_float4(sx / c, sy / c, sz / c, st / c);
  }
  // owners in the workgroups below this one; a block whose word is overdue is counted here: which of its points own their voxel
  // (one 16-byte request per point decides it, as above; a slot its owner - a point of another block - has freed already reads
  // empty: not this point's voxel any more, and it never was its owner)
#ifdef LII_ABL_EMIT_NOPREFIX
  // ABLATION (timing only, tools/ab_build.sh ... -DLII_ABL_EMIT_NOPREFIX): no exchange of counts inside the launch - every workgroup
  // writes its owners to its own stretch of 256 slots and the cloud is taken to have n entries (the holes keep whatever they held).
  // Prices what an UNCOMPACTED down-sampled cloud would save in this kernel; the results of such a build are wrong.
  const unsigned int base_abl = blockIdx.x * 256u;
#endif
  const unsigned int base =
#ifdef LII_ABL_EMIT_NOPREFIX
      true ? base_abl :
#endif
      prefix_below(counts, epoch, (int)blockIdx.x, s_sum, test_late != 0, [&](int q) -> unsigned int {
    const int j = q * 256 + tid;
    bool f = false;
    if (j < n) {
      const unsigned int sq = tb.slot_of[j];
      if (sq != kVhEmpty) {
        const uint4 a = *reinterpret_cast<const uint4*>(tb.slots + sq);
        f = ident_part || min(a.z, a.w) == (unsigned)j;
      }
    }
    unsigned int tot;
    (void)block_rank_of_flag(f, s_w, &tot);
    return tot;
  }).y;
  if (hold) {  // (uniform; tests only) the word must be out before ANY lane of this workgroup frees a slot: a helper that recounts this
               // block from freed slots trusts its own count only while the word is missing (prefix_below)
    if (tid == 0) {
      __hip_atomic_store(counts + blockIdx.x, ((unsigned long long)epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
    __syncthreads();
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {  // size of the down-sampled cloud
    int n_down = (int)(base + total);
#ifdef LII_ABL_EMIT_NOPREFIX
    n_down = n;
#endif
    if (ABS && tb.part_world > 1u && n_down > tb.part_bound) {  // this rank's share outgrew what the launches behind are sized for
      n_down = tb.part_bound;
      __hip_atomic_store(tb.part_overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    *n_out = n_down;
  }
  if (!first) return;
  const unsigned int pos = base + rank;
  out[pos] = cen;
  if (ident_part) {
    pcl_out[pos] = (unsigned)i;
  } else if (ABS) {
    // PCL's index of this voxel, from any of its points (the first): as k_vhash_insert computes it
    const int i0 = (int)(floorf(p0.x * v.inv_leaf) - (float)v.min_b[0]);
    const int i1 = (int)(floorf(p0.y * v.inv_leaf) - (float)v.min_b[1]);
    const int i2 = (int)(floorf(p0.z * v.inv_leaf) - (float)v.min_b[2]);
    pcl_out[pos] = (unsigned)(i0 * v.mul[0] + i1 * v.mul[1] + i2 * v.mul[2]);
  } else {
    pcl_out[pos] = (unsigned int)key;
  }
  // the slot is free again for the next scan
  uint4* line = reinterpret_cast<uint4*>(sl);
  line[0] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, kVhEmpty, kVhEmpty);
  line[1] = make_uint4(0u, kVhEmpty, kVhEmpty, kVhEmpty);
}

// ------------------------------------------------------------------------------------------------
// launchers
static inline int nblk(int n, int b) { return (n + b - 1) / b; }
void launch_time_extent(const float4* pts, int n, unsigned long long* extent, unsigned long long* extent_next, float4* copy_to,
                        const void* ctrl_src, void* ctrl_dst, size_t ctrl_bytes, hipStream_t s) {
  int nb = nblk(n, 256 * 2);  // every workgroup ends with a pair of atomics on one cache line: keep them few
  if (nb > 256) nb = 256;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_time_extent, dim3(nb + (ctrl_bytes ? 1 : 0)), dim3(256), 0, s, pts, n, extent, extent_next, copy_to,
                     static_cast<const uint4*>(ctrl_src), static_cast<uint4*>(ctrl_dst), (int)(ctrl_bytes / 16));
}
void launch_voxel_minmax(const float4* pts, int n, unsigned int* mm, unsigned int* mm_next, hipStream_t s) {
  int nb = nblk(n, 256 * 4);
  if (nb > 256) nb = 256;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_voxel_minmax, dim3(nb), dim3(256), 0, s, pts, n, mm, mm_next);
}
void launch_voxel_keys(const float4* pts, int n, const unsigned int* mm, const unsigned int* bbox_rows, int n_rows, float leaf,
                       unsigned long long* keys, unsigned int* pcl_keys, int* filtered_dev,
                       unsigned long long* samples, int sample_width, hipStream_t s) {
  if (n > 0)
    hipLaunchKernelGGL(k_voxel_keys, dim3(nblk(n, 256)), dim3(256), 0, s, pts, n, mm, bbox_rows, n_rows, leaf, keys, pcl_keys,
                       filtered_dev, samples, sample_width);
}
// fused: the table is keyed by absolute voxel coordinates (the only form a job can share by voxel)
static VhTable vh_table(const VoxelHashBuffers* vh, int n, bool fused) {
  VhTable tb;
  memset(&tb, 0, sizeof(tb));
  if (!vh) return tb;
  tb.slots = static_cast<VhSlot*>(vh->slots);
  tb.mask = (unsigned int)voxel_hash_slots(n) - 1u;
  tb.slot_of = vh->slot_of; tb.next = vh->next; tb.crowded = vh->crowded;
  if (fused && vh->part_world > 1) {
    tb.part_world = (unsigned int)vh->part_world; tb.part_rank = (unsigned int)vh->part_rank;
    tb.part_bound = voxel_partition_bound(n, vh->part_world); tb.part_overflow = vh->part_overflow;
  }
  return tb;
}
int voxel_partition_bound(int n, int world) {
  if (world <= 1) return n;
  const long long b = (long long)n / world + (long long)n / (4 * world) + 2048;  // the mean share + 25 % + 2048: tens of standard deviations of a hashed split
  return (int)(b < n ? b : n);
}
static DeskewIo deskew_io(const DeskewPlan& p) {
  DeskewIo io;
  io.in = p.in; io.out = p.out; io.n = p.n; io.sorted = p.sorted; io.extent = p.extent; io.bbox_rows = p.bbox_rows;
  io.leaf = p.leaf;
  io.tb = vh_table(p.vh, p.n, true);
  io.ctrl_src = static_cast<const uint4*>(p.ctrl_src); io.ctrl_dst = static_cast<uint4*>(p.ctrl_dst); io.ctrl_vec = (int)(p.ctrl_bytes / 16);
#ifdef LII_GAP_TRACE
  io.gap = p.gap;
#endif
  return io;
}
template <bool FUSE, int KP>
static void launch_deskew_imu_t(const DeskewIo& io, const UndistArg& u, int K, const double* poses_host, const double* poses_dev, int nb, hipStream_t s) {
  PoseTab<KP> tab;
  if (KP > 0) memcpy(tab.v, poses_host, sizeof(double) * 22 * (size_t)K);
  hipLaunchKernelGGL((k_deskew_imu<FUSE, KP>), dim3(nb), dim3(256), 0, s, io, u, K, poses_dev, tab);
}
// poses_host != nullptr and K <= 64: the table travels in the kernel arguments; otherwise it is read from poses_dev.
// p.vh != nullptr (and p.leaf > 0): the insert of the hashed voxel filter rides along (launch_voxel_hash(..., stages = 4) goes on).
void launch_deskew_imu(const DeskewPlan& p, const double* poses_host, const double* poses_dev, int K, const UndistArgH& uh, hipStream_t s) {
  if (p.n <= 0) return;
  UndistArg u;
  static_assert(sizeof(UndistArg) == sizeof(UndistArgH), "layout");
  memcpy(&u, &uh, sizeof(u));
  const DeskewIo io = deskew_io(p);
  const int nb = nblk(p.n, 256) + (io.ctrl_vec > 0 ? 1 : 0);
  const bool fuse = p.vh != nullptr;
  const int kp = !poses_host ? 0 : (K <= 16 ? 16 : (K <= 32 ? 32 : (K <= 64 ? 64 : 0)));
  if (fuse) {
    if (kp == 16) launch_deskew_imu_t<true, 16>(io, u, K, poses_host, poses_dev, nb, s);
    else if (kp == 32) launch_deskew_imu_t<true, 32>(io, u, K, poses_host, poses_dev, nb, s);
    else if (kp == 64) launch_deskew_imu_t<true, 64>(io, u, K, poses_host, poses_dev, nb, s);
    else launch_deskew_imu_t<true, 0>(io, u, K, poses_host, poses_dev, nb, s);
  } else {
    if (kp == 16) launch_deskew_imu_t<false, 16>(io, u, K, poses_host, poses_dev, nb, s);
    else if (kp == 32) launch_deskew_imu_t<false, 32>(io, u, K, poses_host, poses_dev, nb, s);
    else if (kp == 64) launch_deskew_imu_t<false, 64>(io, u, K, poses_host, poses_dev, nb, s);
    else launch_deskew_imu_t<false, 0>(io, u, K, poses_host, poses_dev, nb, s);
  }
}
void launch_deskew_imu_gated(const DeskewPlan& p, const DeskewGate& gate, hipStream_t s) {
  if (p.n <= 0) return;
  const DeskewIo io = deskew_io(p);
  const int nb = nblk(p.n, 256) + 1;  // (+ the gate workgroup, which also pulls the control block)
  if (p.vh) hipLaunchKernelGGL(k_deskew_imu_gated<true>, dim3(nb), dim3(256), 0, s, io, gate);
  else hipLaunchKernelGGL(k_deskew_imu_gated<false>, dim3(nb), dim3(256), 0, s, io, gate);
}
void launch_deskew_cv(const DeskewPlan& p, const CvArgH& ah, hipStream_t s) {
  if (p.n <= 0) return;
  CvArg a;
  static_assert(sizeof(CvArg) == sizeof(CvArgH), "layout");
  memcpy(&a, &ah, sizeof(a));
  const DeskewIo io = deskew_io(p);
  const int nb = nblk(p.n, 256) + (io.ctrl_vec > 0 ? 1 : 0);
  if (p.vh) hipLaunchKernelGGL(k_deskew_cv<true>, dim3(nb), dim3(256), 0, s, io, a);
  else hipLaunchKernelGGL(k_deskew_cv<false>, dim3(nb), dim3(256), 0, s, io, a);
}
void launch_voxel_hash_clear(const VoxelHashBuffers& vh, size_t slots, hipStream_t s) {
  hipLaunchKernelGGL(k_vh_clear, dim3((unsigned int)((slots + 255) / 256)), dim3(256), 0, s, static_cast<VhSlot*>(vh.slots), (unsigned int)slots);
}
// stages: 1 = insert keyed by the PCL voxel index (stand-alone filter), 2 = its emit, 4 = the emit behind a fused de-skew
// (the table is keyed by absolute voxel coordinates).  epoch: the number of this filter run (never 0; VoxelHashBuffers::counts).
void launch_voxel_hash(const VoxelHashBuffers& vh, const float4* pts, int n, const unsigned int* mm, const unsigned int* bbox_rows,
                       int n_rows, float leaf, float4* out, int* n_out, int* filtered, unsigned int* pcl_out, int stages, unsigned int epoch,
                       hipStream_t s, int test_late) {
  if (n <= 0) return;
  const VhTable tb = vh_table(&vh, n, (stages & 4) != 0);
  const int nb = nblk(n, 256);
  if (stages & 1) hipLaunchKernelGGL(k_vhash_insert, dim3(nb), dim3(256), 0, s, pts, n, mm, bbox_rows, n_rows, leaf, tb, filtered);
  if (stages & 2) hipLaunchKernelGGL(k_vhash_emit<false>, dim3(nb), dim3(256), 0, s, pts, n, tb, vh.counts, epoch, out, n_out, pcl_out, nullptr, 0, leaf, filtered, test_late);
  if (stages & 4) hipLaunchKernelGGL(k_vhash_emit<true>, dim3(nb), dim3(256), 0, s, pts, n, tb, vh.counts, epoch, out, n_out, pcl_out, bbox_rows, n_rows, leaf, filtered, test_late);
}
size_t voxel_hash_slots(int max_n) {
  size_t slots = 1024;
  while (slots < 4u * (size_t)max_n) slots <<= 1;
  return slots;
}
float ord_to_float(unsigned int o) {
  unsigned int u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

}  // namespace lii
