// HIP kernels of the LI-Init hot path for gfx950 (CDNA4, wave64).  Hand-written.  The path is gather + per-point small algebra + a
// low-rank reduction; its one contraction - the 13 x 13 Gram matrix of a wavefront's rows - runs on the fp64 matrix
// cores (block_reduce_rows: v_mfma_f64_16x16x4_f64).
//
// Kernels and the reference code they replace (paths relative to the reference root):
//   k_map_keys / k_map_gather / k_block_flags / k_cells_fill / k_win_bbox / k_win_fill
//                      device mirror of the ikd-Tree point set as a cell-sorted array + block-hierarchical grid, and the dense cell
//                      window over the map's box (include/ikd-Tree/ikd_Tree.cpp:336-347 Build)
//   k_knn_ck           KD_TREE::Nearest_Search (ikd_Tree.cpp:349-379, Search :825-968) for every point of the scan, after
//                      pointBodyToWorld (src/laserMapping.cpp:209-220, call :973-985): four lanes per query, chunked scan, packed keys
//   k_fit_reduce       its completion workgroups finish the searches the 3 x 3 x 3 pass could not prove exact (knn_fallback_wave:
//                      ikd_Tree.cpp:827-842), then esti_plane + residual / selection (src/laserMapping.cpp:987-1011), Jacobian rows
//                      (:1035-1071) and the H^T R^-1 H / H^T R^-1 z sums (:1073-1080)
//   k_knn_complete     the completion alone (lii_map_incremental of a sharded job)
//   k_complete_listed  the completion of a search pass that listed more unfinished queries than the fit launch's completion workgroups
//                      take (a sensor looking into unmapped space): one wavefront per listed query, in a launch of its own
//   k_reduce91         deterministic final sum of the per-block partials (lii_iekf.hip fuses it with the solve)
//   (de-skew and voxel grid: lii_scan.hip)
//   k_calib_eval       include/LI_init/LI_init.h:91-205 residuals + analytic Jacobians
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <string.h>
#include <cstring>
#include <math.h>
#include <cstdlib>
#include <stdint.h>
#include <utility>

#include "lii_device.h"

namespace lii {

// ------------------------------------------------------------------------------------------------
// helpers
constexpr int kBlockCells = 512;  // 8 x 8 x 8 cells per block

// sort key of a map point: (Bz, By, Bx) block key in the high bits, local cell (lz, ly, lx) in the low 9
__device__ __forceinline__ unsigned long long pack_block(int bx, int by, int bz) {  // biased block coordinates
  return ((unsigned long long)(unsigned)bz << 36) | ((unsigned long long)(unsigned)by << 18) | (unsigned long long)(unsigned)bx;
}
__device__ __forceinline__ unsigned long long point_key(int cx, int cy, int cz) {
  const unsigned ux = (unsigned)(cx + kCellBias), uy = (unsigned)(cy + kCellBias), uz = (unsigned)(cz + kCellBias);
  const unsigned long long bk = pack_block((int)(ux >> kCoarseShift), (int)(uy >> kCoarseShift), (int)(uz >> kCoarseShift));
  const unsigned local = ((uz & 7u) << 6) | ((uy & 7u) << 3) | (ux & 7u);
  return (bk << 9) | local;
}
__device__ __forceinline__ unsigned int hash_key(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (unsigned int)k;
}
__device__ __forceinline__ unsigned int hash_block(int bx, int by, int bz) {
  // block coordinates are < 2^18 after biasing: 24-bit multiplies are full-rate VALU ops
  return (__umul24((unsigned)bx, 7919u * 1021u) ^ __umul24((unsigned)by, 104729u * 13u) ^ __umul24((unsigned)bz, 1299709u)) * 2654435761u;
}

__device__ __forceinline__ int cell_of(float v, float inv_cs) { return (int)floorf(v * inv_cs); }

// Squared distance with the reference's float32 evaluation order and NO fused multiply-add
// (KD_TREE::calc_dist, include/ikd-Tree/ikd_Tree.cpp:1273-1277, compiled without FMA contraction).
__device__ __forceinline__ float dist2_ref(float qx, float qy, float qz, float px, float py, float pz) {
  float dx = __fsub_rn(qx, px), dy = __fsub_rn(qy, py), dz = __fsub_rn(qz, pz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ------------------------------------------------------------------------------------------------
// map index construction
__global__ void k_map_keys(const float4* __restrict__ pts, int n, float inv_cs, unsigned long long* __restrict__ keys,
                           unsigned int* __restrict__ idx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 p = pts[i];
  keys[i] = point_key(cell_of(p.x, inv_cs), cell_of(p.y, inv_cs), cell_of(p.z, inv_cs));
  idx[i] = (unsigned)i;
}

__global__ void k_map_gather(const float4* __restrict__ src, const unsigned int* __restrict__ idx, int n,
                             float4* __restrict__ dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dst[i] = src[idx[i]];
}

// flags[i] = 1 where a new 8x8x8 block starts in the sorted key array (inclusive scan of it = block id + 1)
__global__ void k_block_flags(const unsigned long long* __restrict__ keys, int n, unsigned int* __restrict__ flags) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flags[i] = (i == 0 || (keys[i] >> 9) != (keys[i - 1] >> 9)) ? 1u : 0u;
}

__global__ void k_table_clear(BlockEntry* blocks, unsigned int cap) {
  unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) {
    BlockEntry e;
    e.key = kEmptyKey;
    e.id = 0;
    e.pad = 0;
    blocks[i] = e;
  }
}

// cells must be zero-filled for the n_blocks * 512 entries in use
__global__ void k_cells_fill(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ ranks, int n,
                             BlockEntry* blocks, unsigned int block_mask, uint2* __restrict__ cells, unsigned long long* __restrict__ key_of_id) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = keys[i];
  const bool is_start = (i == 0) || (keys[i - 1] != key);
  const bool is_end = (i == n - 1) || (keys[i + 1] != key);
  if (!is_start && !is_end) return;
  const unsigned int id = ranks[i] - 1;
  const unsigned int local = (unsigned int)(key & 511u);
  unsigned int* cell = reinterpret_cast<unsigned int*>(&cells[(size_t)id * kBlockCells + local]);
  if (is_start) cell[0] = (unsigned)i;
  if (is_end) cell[1] = (unsigned)(i + 1);
  if (is_start && ((i == 0) || ((keys[i - 1] >> 9) != (key >> 9)))) {
    const unsigned long long bk = key >> 9;
    if (key_of_id) key_of_id[id] = bk;  // (WinKeep: the block's coordinates by its id)
    unsigned int slot = hash_block((int)(bk & 0x3FFFF), (int)((bk >> 18) & 0x3FFFF), (int)((bk >> 36) & 0x3FFFF)) & block_mask;
    while (true) {
      unsigned long long prev = atomicCAS(&blocks[slot].key, kEmptyKey, bk);
      if (prev == kEmptyKey) { blocks[slot].id = id; blocks[slot].pad = 1u; break; }  // pad = "id is valid" (k_ins_cells waits on it)
      slot = (slot + 1) & block_mask;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// exact 5-NN with d2 <= max_d2 (KD_TREE::Nearest_Search semantics, quirk A5)
struct Knn5 {
  float d0, d1, d2, d3, d4;
  int i0, i1, i2, i3, i4;
};

__device__ __forceinline__ void knn_insert(Knn5& k, float d, int j) {
  // precondition d < k.d4.  Strict '<' keeps the earlier-visited candidate on exact ties (ikd_Tree.cpp:842).  Branch-free:
  // in a wavefront some lane almost always takes the longest path of a compare-and-return chain anyway.
  const bool c3 = d < k.d3, c2 = d < k.d2, c1 = d < k.d1, c0 = d < k.d0;
  k.d4 = c3 ? k.d3 : d;                 k.i4 = c3 ? k.i3 : j;
  k.d3 = c2 ? k.d2 : (c3 ? d : k.d3);   k.i3 = c2 ? k.i2 : (c3 ? j : k.i3);
  k.d2 = c1 ? k.d1 : (c2 ? d : k.d2);   k.i2 = c1 ? k.i1 : (c2 ? j : k.i2);
  k.d1 = c0 ? k.d0 : (c1 ? d : k.d1);   k.i1 = c0 ? k.i0 : (c1 ? j : k.i1);
  k.d0 = c0 ? d : k.d0;                 k.i0 = c0 ? j : k.i0;
}

__device__ __forceinline__ float axis_gap(float q, int c, float cs, float eps) {
  float lo = (float)c * cs - eps, hi = (float)(c + 1) * cs + eps;
  return fmaxf(fmaxf(lo - q, q - hi), 0.f);
}

// block id of the 8x8x8 block with (unbiased) block coordinates (X, Y, Z), or -1
__device__ __forceinline__ int find_block(const GridView& g, int X, int Y, int Z) {
  const int bb = kCellBias >> kCoarseShift;
  const unsigned long long bk = pack_block(X + bb, Y + bb, Z + bb);
  unsigned int slot = hash_block(X + bb, Y + bb, Z + bb) & g.block_mask;
  while (true) {
    BlockEntry e = g.blocks[slot];
    if (e.key == bk) return (int)e.id;
    if (e.key == kEmptyKey) return -1;
    slot = (slot + 1) & g.block_mask;
  }
}
// [start, end) of cell (ix, iy, iz) in the sorted point array (empty -> start == end)
__device__ __forceinline__ uint2 cell_range(const GridView& g, int ix, int iy, int iz) {
  const int id = find_block(g, ix >> kCoarseShift, iy >> kCoarseShift, iz >> kCoarseShift);
  if (id < 0) return make_uint2(0u, 0u);
  const unsigned local = (((unsigned)iz & 7u) << 6) | (((unsigned)iy & 7u) << 3) | ((unsigned)ix & 7u);
  return g.cells[(size_t)id * kBlockCells + local];
}

__device__ __forceinline__ void scan_range(const float4* __restrict__ pts, const float max_d2, unsigned int start, unsigned int end,
                                           float qx, float qy, float qz, Knn5& k) {
  // four candidates per trip: the loads do not depend on the running top-5, so they are issued together
  for (unsigned int j = start; j < end; j += 4) {
    const unsigned int last = end - 1;
    float4 p0 = pts[j];
    float4 p1 = pts[min(j + 1, last)];
    float4 p2 = pts[min(j + 2, last)];
    float4 p3 = pts[min(j + 3, last)];
    float d0 = dist2_ref(qx, qy, qz, p0.x, p0.y, p0.z);
    float d1 = dist2_ref(qx, qy, qz, p1.x, p1.y, p1.z);
    float d2 = dist2_ref(qx, qy, qz, p2.x, p2.y, p2.z);
    float d3 = dist2_ref(qx, qy, qz, p3.x, p3.y, p3.z);
    if (d0 <= max_d2 && d0 < k.d4) knn_insert(k, d0, (int)j);
    if (j + 1 < end && d1 <= max_d2 && d1 < k.d4) knn_insert(k, d1, (int)(j + 1));
    if (j + 2 < end && d2 <= max_d2 && d2 < k.d4) knn_insert(k, d2, (int)(j + 2));
    if (j + 3 < end && d3 <= max_d2 && d3 < k.d4) knn_insert(k, d3, (int)(j + 3));
  }
}

// ------------------------------------------------------------------------------------------------
// esti_plane<double> (include/common_lib.h:236-269): 5x3 least squares A n = -1 by column-pivoted
// Householder QR — Eigen's ColPivHouseholderQR algorithm (Eigen >= 3.3.4, third-party, restated from
// its published structure: max-norm column pivoting with LAPACK-WN-176 norm down-dating, reflectors,
// solve over nonzeroPivots).  Double precision, contraction off (see Makefile) to track the CPU path.
__device__ __forceinline__ void qr_solve_5x3(double (&a)[5][3], double (&x)[3]) {
  const double eps = 2.220446049250313e-16;
  double hc[3];
  int tr[3];
  double nu[3], nd[3];
  double maxn = 0;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    double s = 0;
#pragma unroll
    for (int r = 0; r < 5; r++) s += a[r][c] * a[r][c];
    nu[c] = nd[c] = sqrt(s);
    maxn = fmax(maxn, nu[c]);
  }
  const double th = maxn * eps;  // Eigen 3.3: threshold_helper = abs2(maxCoeff * epsilon) / rows
  const double thr_helper = th * th / 5.0;
  const double downdate_thr = 1.4901161193847656e-08;  // sqrt(eps)
  int nzp = 3;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int big = k;
    double bigv = nu[k];
#pragma unroll
    for (int j = k + 1; j < 3; j++)
      if (nu[j] > bigv) { bigv = nu[j]; big = j; }
    if (nzp == 3 && bigv * bigv < thr_helper * (double)(5 - k)) nzp = k;
    tr[k] = big;
    if (big != k) {
#pragma unroll
      for (int j = k + 1; j < 3; j++)
        if (j == big) {
#pragma unroll
          for (int r = 0; r < 5; r++) { double t = a[r][k]; a[r][k] = a[r][j]; a[r][j] = t; }
          double t = nu[k]; nu[k] = nu[j]; nu[j] = t;
          t = nd[k]; nd[k] = nd[j]; nd[j] = t;
        }
    }
    double tail = 0;
#pragma unroll
    for (int r = k + 1; r < 5; r++) tail += a[r][k] * a[r][k];
    double c0 = a[k][k], beta, tau;
    if (tail <= 2.2250738585072014e-308) {
      tau = 0;
      beta = c0;
#pragma unroll
      for (int r = k + 1; r < 5; r++) a[r][k] = 0;
    } else {
      beta = sqrt(c0 * c0 + tail);
      if (c0 >= 0) beta = -beta;
      double den = c0 - beta;
#pragma unroll
      for (int r = k + 1; r < 5; r++) a[r][k] /= den;
      tau = (beta - c0) / beta;
    }
    hc[k] = tau;
    a[k][k] = beta;
    if (tau != 0) {
#pragma unroll
      for (int j = k + 1; j < 3; j++) {
        double tmp = a[k][j];
#pragma unroll
        for (int r = k + 1; r < 5; r++) tmp += a[r][k] * a[r][j];
        a[k][j] -= tau * tmp;
#pragma unroll
        for (int r = k + 1; r < 5; r++) a[r][j] -= tau * a[r][k] * tmp;
      }
    }
#pragma unroll
    for (int j = k + 1; j < 3; j++) {
      if (nu[j] != 0) {
        double t = fabs(a[k][j]) / nu[j];
        t = (1.0 + t) * (1.0 - t);
        t = t < 0 ? 0 : t;
        double ratio = nu[j] / nd[j];
        double t2 = t * ratio * ratio;
        if (t2 <= downdate_thr) {
          double s = 0;
#pragma unroll
          for (int r = k + 1; r < 5; r++) s += a[r][j] * a[r][j];
          nd[j] = sqrt(s);
          nu[j] = nd[j];
        } else {
          nu[j] *= sqrt(t);
        }
      }
    }
  }
  double c[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (k < nzp && hc[k] != 0) {
      double tmp = c[k];
#pragma unroll
      for (int r = k + 1; r < 5; r++) tmp += a[r][k] * c[r];
      c[k] -= hc[k] * tmp;
#pragma unroll
      for (int r = k + 1; r < 5; r++) c[r] -= hc[k] * a[r][k] * tmp;
    }
  }
  double y[3] = {0, 0, 0};
#pragma unroll
  for (int i = 2; i >= 0; i--) {
    if (i < nzp) {
      double s = c[i];
#pragma unroll
      for (int j = i + 1; j < 3; j++)
        if (j < nzp) s -= a[i][j] * y[j];
      y[i] = s / a[i][i];
    }
  }
  int perm[3] = {0, 1, 2};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    // swap(perm[k], perm[tr[k]]) with tr[k] >= k, written without dynamic register indexing
#pragma unroll
    for (int j = k + 1; j < 3; j++)
      if (tr[k] == j) { int t = perm[k]; perm[k] = perm[j]; perm[j] = t; }
  }
  x[0] = x[1] = x[2] = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    if (i < nzp) {
      double v = y[i];
      if (perm[i] == 0) x[0] = v;
      else if (perm[i] == 1) x[1] = v;
      else x[2] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// per-block reduction of the 12-column Jacobian rows into 78 + 12 sums (+ count)
// Round 4: the sums of a wavefront are ONE small matrix product - S = V^T V with V the 64 x 13 matrix of its rows (12 Jacobian
// columns + z) - and run on the matrix cores: v_mfma_f64_16x16x4_f64 takes A (16 x 4) and B (4 x 16) with lane l supplying
// A[l % 16][l / 16] and B[l / 16][l % 16], i.e. for S = V^T V over four points the SAME value in both operands - component l % 16
// of point 4 kb + l / 16 - so one LDS read per lane and instruction, sixteen instructions per wavefront (round 3: 64 steps of two
// multiply-adds per lane on the vector ALU, 256 LDS reads per lane: ~1.3 us of every fit launch).  The only contraction of the path;
// fp64 in, fp64 accumulate, fixed order (deterministic).  R^-1 scales the sums, not the factors.
constexpr int kRowStride = 17;    // doubles per point in LDS (16 components + 1 pad: conflict-free writes, near conflict-free reads)
constexpr int kFitWaves = kBlock / 64;
struct ReduceShared {
  double row[kFitWaves][64 * kRowStride];
  double part[kFitWaves][96];
  int cnt[kFitWaves];
};
typedef double v4f64 __attribute__((ext_vector_type(4)));

// partial_out[t * stride] (t < 91) receives this block's sums
__device__ __forceinline__ void block_reduce_rows(ReduceShared& sh, const double (&h)[12], double z, bool sel, double rinv,
                                                  double* __restrict__ partial_out, int stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* row = sh.row[wave];
#pragma unroll
  for (int c = 0; c < 12; c++) row[lane * kRowStride + c] = sel ? h[c] : 0.0;
  row[lane * kRowStride + 12] = sel ? z : 0.0;
  row[lane * kRowStride + 13] = 0.0; row[lane * kRowStride + 14] = 0.0; row[lane * kRowStride + 15] = 0.0;
  unsigned long long m = __ballot(sel);
  if (lane == 0) sh.cnt[wave] = __popcll(m);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (the wavefront reads back only what it wrote itself)
  const int c = lane & 15, q = lane >> 4;
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kb = 0; kb < 16; kb++) {
    const double v = row[(4 * kb + q) * kRowStride + c];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
  }
  // acc[r] = S[q + 4 r][c] (the f64 form's own C / D map: row = (lane >> 4) + 4 reg, col = lane & 15 - not the f32 forms'): the upper
  // triangle of the 12 x 12 block (78), the z column (12)
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = q + 4 * r, j = c;
    if (i < 12 && j <= 12 && i <= j) {
      const int t = j == 12 ? 78 + i : i * 12 - (i * (i - 1)) / 2 + (j - i);
      sh.part[wave][t] = acc[r] * rinv;
    }
  }
  __syncthreads();
  // (kBlock = 64: one wavefront, lanes 0..63 write pairs t and t + 64)
  for (int t = threadIdx.x; t < 91; t += kBlock) {
    if (t < 90) {
      double s = sh.part[0][t];
#pragma unroll
      for (int w = 1; w < kFitWaves; w++) s += sh.part[w][t];
      partial_out[(size_t)t * stride] = s;
    } else {
      int cn = sh.cnt[0];
#pragma unroll
      for (int w = 1; w < kFitWaves; w++) cn += sh.cnt[w];
      partial_out[(size_t)90 * stride] = (double)cn;
    }
  }
}

// blocks are remapped so that each XCD (block b runs on XCD b % 8) works on a CONTIGUOUS eighth of the point
// stream: neighbouring scan points touch the same map cells, which then stay in that XCD's private 4 MiB L2
__device__ __forceinline__ int xcd_remap(int b, int nb_real) {
  const int per = (nb_real + 7) >> 3;  // the grid is launched with 8 * per blocks
  return (b & 7) * per + (b >> 3);
}

// residual gate + Jacobian row of one point (src/laserMapping.cpp:999-1010, :1035-1071)
struct RowOut {
  double h[12];
  double z;
  bool sel;
};
__device__ __forceinline__ void residual_row(const PoseArg& ps, int imu_en, double bx, double by, double bz, double ix,
                                             double iy, double iz, float wx, float wy, float wz, double pa, double pbn,
                                             double pc, double pd, RowOut& o) {
  float pd2 = (float)(pa * wx + pbn * wy + pc * wz + pd);
  double pbnorm = sqrt(bx * bx + by * by + bz * bz);
  float s = (float)(1 - 0.9 * fabsf(pd2) / sqrt(pbnorm));
  if (s > 0.9) {
    o.sel = true;
    // normvec stores n̂ as float (:1004-1006); the Jacobian reads those floats back (:1046-1047)
    double nx = (double)(float)pa, ny = (double)(float)pbn, nz = (double)(float)pc;
    double tx = ps.R[0] * nx + ps.R[3] * ny + ps.R[6] * nz;  // R_end^T n̂
    double ty = ps.R[1] * nx + ps.R[4] * ny + ps.R[7] * nz;
    double tz = ps.R[2] * nx + ps.R[5] * ny + ps.R[8] * nz;
    o.h[0] = -iz * ty + iy * tz;  // [p_I]x R_end^T n̂
    o.h[1] = iz * tx - ix * tz;
    o.h[2] = -iy * tx + ix * ty;
    o.h[3] = nx; o.h[4] = ny; o.h[5] = nz;
    if (imu_en) {
      double ux = ps.RLI[0] * tx + ps.RLI[3] * ty + ps.RLI[6] * tz;  // R_LI^T R_end^T n̂
      double uy = ps.RLI[1] * tx + ps.RLI[4] * ty + ps.RLI[7] * tz;
      double uz = ps.RLI[2] * tx + ps.RLI[5] * ty + ps.RLI[8] * tz;
      o.h[6] = -bz * uy + by * uz;  // [p_L]x ...
      o.h[7] = bz * ux - bx * uz;
      o.h[8] = -by * ux + bx * uy;
      o.h[9] = tx; o.h[10] = ty; o.h[11] = tz;
    }
    o.z = -(double)pd2;
  }
}
// esti_plane on 5 neighbours; returns validity and (n̂, d)
__device__ __forceinline__ bool fit_plane(const float4 (&nb)[5], double plane_thr, double& pa, double& pbn, double& pc,
                                          double& pd) {
  double a[5][3] = {{nb[0].x, nb[0].y, nb[0].z}, {nb[1].x, nb[1].y, nb[1].z}, {nb[2].x, nb[2].y, nb[2].z},
                    {nb[3].x, nb[3].y, nb[3].z}, {nb[4].x, nb[4].y, nb[4].z}};
  double nv[3];
  qr_solve_5x3(a, nv);
  double nn = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
  pa = nv[0] / nn; pbn = nv[1] / nn; pc = nv[2] / nn; pd = 1.0 / nn;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 5; j++)
    if (fabs(pa * nb[j].x + pbn * nb[j].y + pc * nb[j].z + pd) > plane_thr) ok = false;
  return ok;
}

__device__ __forceinline__ bool canon_ties(float4 (&nb)[5]);

// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// The search pass: LPQ (4 by default, 8 optional) lanes per query with box-distance pruning in two rounds.
// Round 1: the 2x2x2 block of cells nearest to the query (own cell + the neighbour on the nearer side of every axis) —
// exactly one cell per lane — which covers the ball of radius g0 = min_axis max(f, cs - f) >= cs / 2 around the query.
// If the merged 5th distance is within g0 the search is complete (the usual case for a converged map).
// Round 2: the other 19 cells of the 3x3x3 block, each tested against the current 5th distance first (the tree's
// calc_box_dist rule), so most of them cost neither a table lookup nor a candidate.

// NC cell lookups with their loads issued as two batches (first probes of all block-table slots, then all cell entries)
// instead of NC dependent probe -> entry chains; a probe that hits a foreign key walks on alone (load factor <= 1/8: rare).
template <int NC>
__device__ __forceinline__ void lookup_cells_batched(const GridView& g, const uint4* __restrict__ tab, const int (&ix)[NC],
                                                     const int (&iy)[NC], const int (&iz)[NC], const bool (&want)[NC],
                                                     uint2 (&out)[NC]) {
  const int bb = kCellBias >> kCoarseShift;
  if (g.win) {  // (uniform) the dense window: one load per cell; the rare cell outside the box takes the tables' way alone
    unsigned int wi[NC];
    bool in[NC];
#pragma unroll
    for (int t = 0; t < NC; t++) {
      const unsigned int ux = (unsigned)(ix[t] - g.wx0), uy = (unsigned)(iy[t] - g.wy0), uz = (unsigned)(iz[t] - g.wz0);
      in[t] = ux < (unsigned)g.wnx && uy < (unsigned)g.wny && uz < (unsigned)g.wnz;
      wi[t] = in[t] ? (uz * (unsigned)g.wny + uy) * (unsigned)g.wnx + ux : 0u;
    }
#pragma unroll
    for (int t = 0; t < NC; t++) out[t] = (want[t] && in[t]) ? g.win[wi[t]] : make_uint2(0u, 0u);
    // (a cell outside the box is EMPTY: the window covers every block the map has - and a block of margin - and is only handed to a
    // launch while the map is as the window found it.  The first form walked the block table for such a cell: the queries of a scan
    // that looks past the map's edge paid a dependent probe each for a block that cannot exist.)
    return;
  }
  unsigned long long bk[NC];
  unsigned int sl[NC];
  uint4 e[NC];
#pragma unroll
  for (int t = 0; t < NC; t++) {
    const int bx = (ix[t] >> kCoarseShift) + bb, by = (iy[t] >> kCoarseShift) + bb, bz = (iz[t] >> kCoarseShift) + bb;
    bk[t] = pack_block(bx, by, bz);
    sl[t] = hash_block(bx, by, bz) & g.block_mask;
  }
#pragma unroll
  for (int t = 0; t < NC; t++) e[t] = want[t] ? tab[sl[t]] : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
  unsigned int ci[NC];
  bool hit[NC];
#pragma unroll
  for (int t = 0; t < NC; t++) {
    unsigned long long ek = ((unsigned long long)e[t].y << 32) | e[t].x;
    while (want[t] && ek != bk[t] && ek != kEmptyKey) {
      sl[t] = (sl[t] + 1) & g.block_mask;
      e[t] = tab[sl[t]];
      ek = ((unsigned long long)e[t].y << 32) | e[t].x;
    }
    hit[t] = want[t] && ek == bk[t];
    ci[t] = e[t].z * (unsigned)kBlockCells + ((((unsigned)iz[t] & 7u) << 6) | (((unsigned)iy[t] & 7u) << 3) | ((unsigned)ix[t] & 7u));
  }
#pragma unroll
  for (int t = 0; t < NC; t++) out[t] = hit[t] ? g.cells[ci[t]] : make_uint2(0u, 0u);
}

// ------------------------------------------------------------------------------------------------
// The search pass.  Four lanes per query (16 queries per wavefront).  Geometry of both kernels below:
// Round 1: the 2x2x2 block of cells nearest to the query (own cell + the neighbour on the nearer side of every axis), two
// cells per lane, which covers the ball of radius g0 = min_axis max(f, cs - f) >= cs / 2 around the query.  If the 5th
// distance is within g0 the search is complete.
// Round 2 (only the queries that need it): the other 19 cells of the 3x3x3 block, each tested against the current 5th
// distance first (the tree's calc_box_dist rule, ikd_Tree.cpp:1279-1289).
// A query whose 3x3x3 block cannot prove its list complete is flagged (kNeedy) and finished by k_fit_reduce / k_knn_complete.


// element t of a PoseArg seen as 24 doubles, without dynamic indexing (which would push the struct into scratch memory)
__device__ __forceinline__ double pose_element(const PoseArg& ps, int t) {
  double v = 0;
#pragma unroll
  for (int e = 0; e < 9; e++) { v = t == e ? ps.R[e] : v; v = t == 12 + e ? ps.RLI[e] : v; }
#pragma unroll
  for (int e = 0; e < 3; e++) { v = t == 9 + e ? ps.p[e] : v; v = t == 21 + e ? ps.TLI[e] : v; }
  return v;
}
// pointBodyToWorld (src/laserMapping.cpp:209-220): fp64 arithmetic, float result
__device__ __forceinline__ void body_to_world(const PoseArg& ps, const float4 pb, float& wx, float& wy, float& wz) {
  double bx = pb.x, by = pb.y, bz = pb.z;
  double ix = ps.RLI[0] * bx + ps.RLI[1] * by + ps.RLI[2] * bz + ps.TLI[0];
  double iy = ps.RLI[3] * bx + ps.RLI[4] * by + ps.RLI[5] * bz + ps.TLI[1];
  double iz = ps.RLI[6] * bx + ps.RLI[7] * by + ps.RLI[8] * bz + ps.TLI[2];
  wx = (float)(ps.R[0] * ix + ps.R[1] * iy + ps.R[2] * iz + ps.p[0]);
  wy = (float)(ps.R[3] * ix + ps.R[4] * iy + ps.R[5] * iz + ps.p[1]);
  wz = (float)(ps.R[6] * ix + ps.R[7] * iy + ps.R[8] * iz + ps.p[2]);
}

// Where a query sits in the grid: its cell, the nearer-side neighbour on every axis, the radius round 1 covers (g0) and the
// radius the whole 3x3x3 block covers (guard).
struct QueryCell {
  int cx, cy, cz, ox, oy, oz;
  float eps, g0, guard;
};
__device__ __forceinline__ QueryCell query_cell(const GridView& g, float wx, float wy, float wz) {
  QueryCell q;
  const float cs = g.cs;
  q.eps = 1e-6f * (fabsf(wx) + fabsf(wy) + fabsf(wz) + 8.f);
  q.cx = cell_of(wx, g.inv_cs); q.cy = cell_of(wy, g.inv_cs); q.cz = cell_of(wz, g.inv_cs);
  const float fx = fminf(fmaxf(wx - (float)q.cx * cs, 0.f), cs), fy = fminf(fmaxf(wy - (float)q.cy * cs, 0.f), cs),
              fz = fminf(fmaxf(wz - (float)q.cz * cs, 0.f), cs);
  q.ox = fx < 0.5f * cs ? -1 : 1; q.oy = fy < 0.5f * cs ? -1 : 1; q.oz = fz < 0.5f * cs ? -1 : 1;
  q.g0 = fminf(fminf(fmaxf(fx, cs - fx), fmaxf(fy, cs - fy)), fmaxf(fz, cs - fz)) - 2.f * q.eps;
  const float mfrac = fmaxf(fminf(fminf(fminf(fx, cs - fx), fminf(fy, cs - fy)), fminf(fz, cs - fz)), 0.f);
  q.guard = cs + mfrac - 2.f * q.eps;
  return q;
}


// ---- packed keys ---------------------------------------------------------------------------------
// The search pass ranks its candidates as 32-bit keys: the float bits of d2 with the low position bits (CkGeom::kPosBits: 8 with
// four lanes per query) replaced by the candidate's position in the group's candidate list - its chunk in the group's table and the
// lane that measured it.  d2 >= 0, so the keys order like the distances (to 15 mantissa bits) and are unique; a sorted list of the
// SEVEN smallest keys is maintained with one v_min_u32 and six v_med3_u32 per candidate - no compares, no selects, no index
// registers.  The lists of a group's lanes are joined by bitonic merges over DPP quad permutes.  At the end the seven winners
// are re-measured exactly and ranked exactly (distance, then position - the visiting order): the five nearest are the exact
// answer unless the exact 5th distance reaches the truncated distance of the 7th key - every candidate that was dropped is at
// least that far - in which case the query is flagged for the completion pass (5th, 6th and 7th distances equal in their kept bits:
// never observed on the bench streams).
constexpr unsigned int kPkInf = 0xFFFFFFFFu;

struct Pk7 {
  unsigned int k0, k1, k2, k3, k4, k5, k6;
};
__device__ __forceinline__ unsigned int umed3(unsigned int a, unsigned int b, unsigned int c) {
  unsigned int r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ void pk_insert(Pk7& L, unsigned int x) {
  L.k6 = umed3(L.k5, L.k6, x);
  L.k5 = umed3(L.k4, L.k5, x);
  L.k4 = umed3(L.k3, L.k4, x);
  L.k3 = umed3(L.k2, L.k3, x);
  L.k2 = umed3(L.k1, L.k2, x);
  L.k1 = umed3(L.k0, L.k1, x);
  L.k0 = min(L.k0, x);
}
template <int CTRL>
__device__ __forceinline__ unsigned int quad_perm(unsigned int v) {
  return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float quad_perm_f(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
#define LII_CE(a, b) { const unsigned int lo_ = min(a, b), hi_ = max(a, b); a = lo_; b = hi_; }
// L <- the seven smallest keys of L and of the partner lane's list (quad permute CTRL), sorted.  C_i = min(L_i, M_{7-i}) with an
// eighth "infinite" key on both sides is a bitonic sequence holding the eight smallest of the sixteen; three half-cleaner stages
// sort it.  Both partners compute the same list.
template <int CTRL>
__device__ __forceinline__ void pk_merge(Pk7& L) {
  const unsigned int m0 = quad_perm<CTRL>(L.k0), m1 = quad_perm<CTRL>(L.k1), m2 = quad_perm<CTRL>(L.k2), m3 = quad_perm<CTRL>(L.k3),
                     m4 = quad_perm<CTRL>(L.k4), m5 = quad_perm<CTRL>(L.k5), m6 = quad_perm<CTRL>(L.k6);
  unsigned int c0 = L.k0, c1 = min(L.k1, m6), c2 = min(L.k2, m5), c3 = min(L.k3, m4), c4 = min(L.k4, m3), c5 = min(L.k5, m2),
               c6 = min(L.k6, m1), c7 = m0;
  LII_CE(c0, c4) LII_CE(c1, c5) LII_CE(c2, c6) LII_CE(c3, c7)
  LII_CE(c0, c2) LII_CE(c1, c3) LII_CE(c4, c6) LII_CE(c5, c7)
  LII_CE(c0, c1) LII_CE(c2, c3) LII_CE(c4, c5)
  c6 = min(c6, c7);
  L.k0 = c0; L.k1 = c1; L.k2 = c2; L.k3 = c3; L.k4 = c4; L.k5 = c5; L.k6 = c6;
}
#undef LII_CE

struct F3 {
  float x, y, z;
};
// xyz of map slot `idx`: 12 of the 16 bytes (w is the insertion id).  The byte offset is formed in 32 bits (the point array holds
// fewer than 2^28 slots), so the load takes the uniform base from scalar registers and ONE address register.
__device__ __forceinline__ F3 load_xyz(const float4* __restrict__ pts, unsigned int idx) {
  typedef float f3v __attribute__((ext_vector_type(3)));
  const f3v v = *reinterpret_cast<const f3v*>(reinterpret_cast<const char*>(pts) + (size_t)(idx << 4));
  F3 r;
  r.x = v.x; r.y = v.y; r.z = v.z;
  return r;
}

template <int LPQ>
__device__ __forceinline__ void pk_group_merge(Pk7& L) {
  if (LPQ >= 2) pk_merge<0xB1>(L);  // lanes 0<->1, 2<->3
  if (LPQ == 4) pk_merge<0x4E>(L);  // lanes 0<->2, 1<->3: every lane of the group now holds the group's seven smallest keys
}
// the value lane J of the group holds
template <int LPQ, int J>
__device__ __forceinline__ float group_bcast_f(float v) {
  if (LPQ == 4) return quad_perm_f<J * 0x55>(v);
  if (LPQ == 2) return quad_perm_f<J == 0 ? 0xA0 : 0xF5>(v);
  return v;
}
// Loops whose index must be a constant expression (register arrays must never be indexed dynamically: they would move to
// scratch memory): f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>).
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
template <class T>
__device__ __forceinline__ T by_value(T x) { return x; }
// arr[BASE + sub] for sub < LPQ as a chain of selects over constant indices (entries behind the array: the last one)
template <int LPQ, int BASE, int N, class T>
__device__ __forceinline__ T pick_by_lane(const T (&arr)[N], int sub) {
  T v = by_value(arr[BASE < N ? BASE : N - 1]);
  static_for<LPQ - 1>([&](auto jc) {
    constexpr int j = decltype(jc)::value + 1;
    v = sub == j ? by_value(arr[BASE + j < N ? BASE + j : N - 1]) : v;
  });
  return v;
}

// ---- the search pass with the GROUP scanning every cell together ("chunked", round 5) ------------------------------------------
// Rounds 3 - 4 (k_knn_pk) gave every lane of a query its own cells: a candidate load of a wavefront then touches up to 64 different cache lines -
// measured 40 per instruction (TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD), 6.6 M (100 k-point scan) and 21 M (500 k) line lookups
// per launch: at one lookup per cycle and compute unit 10.7 and 34.3 us of the vector-memory front end, in launches of 17 - 21 and
// 52 - 64 us (profiles/r05_knn_l1.md).  Here the LPQ lanes of a query read LPQ CONSECUTIVE points of one cell - 64 contiguous bytes
// with four lanes, one or two lines - and neighbouring queries that scan the same cell read the same lines.
// The cells' ranges are cut into chunks of LPQ points; the group keeps a table of its chunks in LDS - word = first map index << 3 |
// points in the chunk (1 .. LPQ; 0: padding behind the last chunk, so that a batch of NB loads needs no bounds test) - in the order
// the cells were looked up: round 1's eight cells, then the outer cells round 2 adds.  A candidate is numbered (chunk << log2 LPQ) |
// lane: the position in its key.  Whatever needs a candidate's map index afterwards - the winners' re-measurement - reads it from
// the table; the per-lane range arithmetic of rounds 3 - 4 (which range does position p belong to?) is gone.
template <int LPQ>
struct CkGeom {
  static constexpr int kLaneShift = LPQ == 4 ? 2 : (LPQ == 2 ? 1 : 0);
  static constexpr int kChunkBits = 6;
  static constexpr int MAXCH = 1 << kChunkBits;                 // chunks a group can number: 64 x LPQ candidates
  static constexpr int kPosBits = kChunkBits + kLaneShift;      // 8 of the 23 mantissa bits with four lanes (rounds 3 - 4: 12)
  static constexpr unsigned int kPosMask = (1u << kPosBits) - 1u;
  static constexpr int NR = 8 / LPQ;                            // cells per lane in round 1
  static constexpr int NW = (7 + LPQ - 1) / LPQ;                // winners a lane re-measures
  static constexpr int MAXPASS = (19 + 2 * LPQ - 1) / (2 * LPQ);  // round 2: two outer cells per lane and pass
};
// sum of `v` over the lanes of the group below this one (exclusive prefix) and over all of them
template <int LPQ>
__device__ __forceinline__ void group_prefix(unsigned int v, int sub, unsigned int& before, unsigned int& total) {
  if (LPQ == 4) {
    const unsigned int v0 = quad_perm<0x00>(v), v1 = quad_perm<0x55>(v), v2 = quad_perm<0xAA>(v), v3 = quad_perm<0xFF>(v);
    before = sub == 0 ? 0u : (sub == 1 ? v0 : (sub == 2 ? v0 + v1 : v0 + v1 + v2));
    total = v0 + v1 + v2 + v3;
  } else if (LPQ == 2) {
    const unsigned int v0 = quad_perm<0xA0>(v), v1 = quad_perm<0xF5>(v);
    before = sub == 0 ? 0u : v0;
    total = v0 + v1;
  } else {
    before = 0u;
    total = v;
  }
}
// The lane's NC cell ranges become chunks at tab[at ...] (the group's table; `at` = where this lane's chunks start); the caller
// has checked that they fit.
template <int LPQ, int NC>
__device__ __forceinline__ void ck_write_chunks(unsigned int* __restrict__ tab, unsigned int at, const uint2 (&r)[NC]) {
#pragma unroll
  for (int t = 0; t < NC; t++) {
    for (unsigned int j = r[t].x; j < r[t].y; j += LPQ) tab[at++] = (j << 3) | min((unsigned)LPQ, r[t].y - j);
  }
}
// chunks [first, end) of the group's table (end is followed by >= NB - 1 padding words), NB loads in flight per lane
template <int LPQ, int NB>
__device__ __forceinline__ void ck_scan(const float4* __restrict__ pts, const unsigned int* __restrict__ tab, unsigned int first, unsigned int end,
                                        int sub, float wx, float wy, float wz, Pk7& L) {
  using G = CkGeom<LPQ>;
  for (unsigned int base = first; base < end; base += NB) {
    unsigned int w[NB];
#pragma unroll
    for (int u = 0; u < NB; u++) w[u] = tab[base + u];
    F3 P[NB];
#pragma unroll
    for (int u = 0; u < NB; u++) P[u] = load_xyz(pts, (w[u] >> 3) + min((unsigned)sub, (w[u] & 7u) - 1u));  // (padding: count 0 -> slot `sub` of the array, discarded)
#pragma unroll
    for (int u = 0; u < NB; u++) {
      const float d = dist2_ref(wx, wy, wz, P[u].x, P[u].y, P[u].z);
      const unsigned int key = (__float_as_uint(d) & ~G::kPosMask) | (((base + u) << G::kLaneShift) | (unsigned)sub);
      pk_insert(L, (unsigned)sub < (w[u] & 7u) ? key : kPkInf);  // (the acceptance test d2 <= max_d2 waits for the re-measurement of the winners)
    }
  }
}
// `forced` > 0: always runs (a host-driven pass: the host has put the pose into `pose`, device memory).
// forced < 0: device-driven loop — pose from `pose` (the control block), runs only when the control block says the next pass
// searches and the loop has not stopped (src/laserMapping.cpp:978, :1102-1106).  An executed pass leaves its pose in
// `search_pose_out` (may be null).
// LPQ = lanes per query (4; 2 lanes issue fewer instructions in total but lose to latency and to the vector-memory front end at every
// size measured: profiles/r05_knn_lpq.md); NB = candidate loads a lane keeps in flight (one batch).
template <int LPQ, int BS, int NB, int WPE>
__global__ __launch_bounds__(BS, WPE) void k_knn_ck(GridView g, RegistrationBuffers rb, const PoseArg* __restrict__ pose,
                                               const IekfCtrl* __restrict__ ctrl, int forced, int nb_real,
                                               double* __restrict__ search_pose_out, int epoch) {
  using G = CkGeom<LPQ>;
  constexpr int QPB = BS / LPQ;
  __shared__ unsigned int s_tab[QPB * (G::MAXCH + NB)];
  // Everything the head of the kernel needs from memory is requested AT ONCE, before anything is waited for: the query point itself
  // (an unsharded cloud: its index depends on nothing that has to be loaded; index clamped, a lane beyond the cloud discards it) and -
  // load_head_scalars - the pose, the loop flags and the size of the cloud.  Round 4 had the flags behind the size behind the pose
  // (24 dependent scalar loads) and the point behind all of them.
  const int blk = xcd_remap(blockIdx.x, nb_real);
  const int sub = threadIdx.x & (LPQ - 1);
  const int ql = blk * QPB + (int)(threadIdx.x / LPQ);
  const bool early = rb.shard_world <= 1;
  float4 pb_early = make_float4(0.f, 0.f, 0.f, 0.f);
  if (early) pb_early = rb.body[min(max(ql, 0), rb.cap - 1)];
  const HeadScalars hs = load_head_scalars(pose, &ctrl->search_next, rb.n_dev ? rb.n_dev : &ctrl->max_it, &ctrl->max_it);
  const PoseArg& ps = hs.ps;
  const int c_search = hs.search_next, c_stop = hs.stop, n_mem = hs.n_mem;
  int lo, n_live;
  shard_range_n(rb, n_mem, lo, n_live);
  // (the list of unfinished queries: launch number e appends to slot e & 1; EVERY enqueued launch - whether its pass is due or not -
  // empties the other slot for launch e + 1: its readers, the fit launch behind launch e - 1, are done)
  if (epoch > 0 && blockIdx.x == 0 && threadIdx.x == 0) rb.flag_count[(epoch + 1) & 1] = 0;
  if (forced < 0 && (c_stop || !c_search)) return;
  if (search_pose_out && blockIdx.x == 0 && threadIdx.x < 24) search_pose_out[threadIdx.x] = pose_element(ps, threadIdx.x);
  if (blk >= nb_real) return;
  if (blk * QPB + (int)((threadIdx.x & ~63u) / LPQ) >= n_live) return;
  const int qi = lo + ql;
  const bool live = ql < n_live;
  float wx = 0, wy = 0, wz = 0;
  if (live && sub == 0) body_to_world(ps, early ? pb_early : rb.body[qi], wx, wy, wz);
  wx = group_bcast_f<LPQ, 0>(wx); wy = group_bcast_f<LPQ, 0>(wy); wz = group_bcast_f<LPQ, 0>(wz);
  const bool active = live && g.n_pts > 0;
  const float INF = __builtin_inff();
  const uint4* __restrict__ tab_blocks = reinterpret_cast<const uint4*>(g.blocks);
  const float4* __restrict__ pts = g.pts;
  unsigned int* const tab = s_tab + (threadIdx.x / LPQ) * (G::MAXCH + NB);

  // Round 1: the 2x2x2 block of cells nearest to the query, NR cells per lane (looked up as a batch), their chunks into the table
  const QueryCell q = query_cell(g, wx, wy, wz);
  const float g0sq = q.g0 * q.g0, guardsq = q.guard * q.guard;
  unsigned int n_chunks;  // chunks in the group's table (group-uniform)
  bool ovf;
  {
    uint2 r[G::NR];
    int jx[G::NR], jy[G::NR], jz[G::NR];
    bool want[G::NR];
#pragma unroll
    for (int t = 0; t < G::NR; t++) {
      const int c = sub * G::NR + t;
      jx[t] = q.cx + ((c & 1) ? q.ox : 0); jy[t] = q.cy + ((c & 2) ? q.oy : 0); jz[t] = q.cz + ((c & 4) ? q.oz : 0);
      want[t] = true;
    }
    lookup_cells_batched<G::NR>(g, tab_blocks, jx, jy, jz, want, r);
    unsigned int mine = 0u;
#pragma unroll
    for (int t = 0; t < G::NR; t++) {
      if (!active) r[t] = make_uint2(0u, 0u);
      mine += (r[t].y - r[t].x + (unsigned)LPQ - 1u) / (unsigned)LPQ;
    }
    unsigned int before;
    group_prefix<LPQ>(mine, sub, before, n_chunks);
    // a group whose cells hold more than MAXCH x LPQ points is left to the completion pass (cells of hundreds of points)
    ovf = n_chunks > (unsigned)G::MAXCH;
    if (ovf) n_chunks = 0u;
    else ck_write_chunks<LPQ, G::NR>(tab, before, r);
    if (sub == 0) {
#pragma unroll
      for (int u = 0; u < NB - 1; u++) tab[n_chunks + u] = 0u;  // padding: a batch of loads runs past the last chunk
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const bool fast = active && !ovf;
  Pk7 L;
  L.k0 = L.k1 = L.k2 = L.k3 = L.k4 = L.k5 = L.k6 = kPkInf;
  ck_scan<LPQ, NB>(pts, tab, 0u, n_chunks, sub, wx, wy, wz, L);
  pk_group_merge<LPQ>(L);

  // Round 2 (the tree's calc_box_dist rule, ikd_Tree.cpp:1279-1289): is the 5th distance - here its upper bound, the 5th key with the
  // position bits set - within the radius round 1 covers?  If not, the outer cells of the 3x3x3 block that can
  // still hold a closer point, two per lane and pass; their chunks continue the table.
  {
    const float ub5 = L.k4 != kPkInf ? __uint_as_float(L.k4 | G::kPosMask) : INF;
    const float bound = fminf(ub5, g.max_d2);
    const bool need2 = fast && !(bound <= g0sq);
    if (__any(need2)) {
      float GX[3], GY[3], GZ[3];
      {
        const float nx = axis_gap(wx, q.cx + q.ox, g.cs, q.eps), fx = axis_gap(wx, q.cx - q.ox, g.cs, q.eps);
        const float ny = axis_gap(wy, q.cy + q.oy, g.cs, q.eps), fy = axis_gap(wy, q.cy - q.oy, g.cs, q.eps);
        const float nz = axis_gap(wz, q.cz + q.oz, g.cs, q.eps), fz = axis_gap(wz, q.cz - q.oz, g.cs, q.eps);
        GX[0] = 0.f; GX[1] = nx * nx; GX[2] = fx * fx;
        GY[0] = 0.f; GY[1] = ny * ny; GY[2] = fy * fy;
        GZ[0] = 0.f; GZ[1] = nz * nz; GZ[2] = fz * fz;
      }
      unsigned int m = 0u;
      static_for<27>([&](auto cc) {
        constexpr int c = decltype(cc)::value, sx = c % 3, sy = (c / 3) % 3, sz = c / 9;
        if constexpr (sx == 2 || sy == 2 || sz == 2) {
          const float d = GX[sx] + GY[sy] + GZ[sz];
          m |= d > bound ? 0u : (1u << c);
        }
      });
      m = need2 ? m : 0u;
      // the group's list continues on lane 0 alone (copies would come back as duplicates), the other lanes start empty
      if (sub != 0) L.k0 = L.k1 = L.k2 = L.k3 = L.k4 = L.k5 = L.k6 = kPkInf;
      unsigned int t = m;
#pragma unroll
      for (int j = 0; j < LPQ - 1; j++) t = j < sub ? (t & (t - 1u)) : t;  // the lane's first survivor: number `sub` of the set bits
      for (int pass = 0; pass < G::MAXPASS; pass++) {
        if (!__any(t != 0u)) break;
        const int c1 = t ? __ffs((int)t) - 1 : -1;
#pragma unroll
        for (int j = 0; j < LPQ; j++) t = t & (t - 1u);
        const int c2 = t ? __ffs((int)t) - 1 : -1;
#pragma unroll
        for (int j = 0; j < LPQ; j++) t = t & (t - 1u);
        uint2 r[2];
        {
          int jx[2], jy[2], jz[2];
          const bool want[2] = {true, true};
          const int ca = c1 < 0 ? 0 : c1, cb = c2 < 0 ? 0 : c2;  // (0 = the query's own cell: looked up for nothing, not branched around)
          auto step = [](int sdig, int o) { return o * ((sdig & 1) - (sdig >> 1)); };
          jx[0] = q.cx + step(ca % 3, q.ox); jy[0] = q.cy + step((ca / 3) % 3, q.oy); jz[0] = q.cz + step(ca / 9, q.oz);
          jx[1] = q.cx + step(cb % 3, q.ox); jy[1] = q.cy + step((cb / 3) % 3, q.oy); jz[1] = q.cz + step(cb / 9, q.oz);
          lookup_cells_batched<2>(g, tab_blocks, jx, jy, jz, want, r);
        }
        if (c1 < 0) r[0] = make_uint2(0u, 0u);
        if (c2 < 0) r[1] = make_uint2(0u, 0u);
        const unsigned int mine = (r[0].y - r[0].x + (unsigned)LPQ - 1u) / (unsigned)LPQ + (r[1].y - r[1].x + (unsigned)LPQ - 1u) / (unsigned)LPQ;
        unsigned int before, add;
        group_prefix<LPQ>(mine, sub, before, add);
        const unsigned int from = n_chunks;
        if (from + add > (unsigned)G::MAXCH) {  // out of table: the group stops here and is left to the completion pass
          ovf = true;
          t = 0u;
          add = 0u;
        } else {
          ck_write_chunks<LPQ, 2>(tab, from + before, r);
        }
        n_chunks = from + add;
        if (sub == 0) {
#pragma unroll
          for (int u = 0; u < NB - 1; u++) tab[n_chunks + u] = 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        ck_scan<LPQ, NB>(pts, tab, from, n_chunks, sub, wx, wy, wz, L);
      }
      pk_group_merge<LPQ>(L);
    }
  }

  // Exact re-measurement of the seven winners: a key's position names its chunk and lane, the table gives the map index - lane `sub`
  // loads and measures winners sub, sub + LPQ, ..., and the seven exact distances are shared by broadcasts inside the group.
  float e[7];
  F3 W[G::NW];
  float d7t;
  bool tie = false;
  {
    const unsigned int K[7] = {L.k0, L.k1, L.k2, L.k3, L.k4, L.k5, L.k6};
    float el[G::NW];
    static_for<G::NW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const unsigned int key = pick_by_lane<LPQ, LPQ * i>(K, sub);
      const unsigned int pos = key == kPkInf ? 0u : (key & G::kPosMask);  // (an empty slot reads chunk 0 - or padding - and is discarded)
      W[i] = load_xyz(pts, (tab[pos >> G::kLaneShift] >> 3) + (pos & (unsigned)(LPQ - 1)));
    });
#pragma unroll
    for (int w = 0; w < 6; w++) tie = tie || ((K[w] ^ K[w + 1]) <= G::kPosMask && K[w + 1] != kPkInf);
    d7t = K[6] != kPkInf ? __uint_as_float(K[6] & ~G::kPosMask) : INF;
#pragma unroll
    for (int i = 0; i < G::NW; i++) el[i] = dist2_ref(wx, wy, wz, W[i].x, W[i].y, W[i].z);
    e[0] = group_bcast_f<LPQ, 0 % LPQ>(el[0 / LPQ]); e[1] = group_bcast_f<LPQ, 1 % LPQ>(el[1 / LPQ]);
    e[2] = group_bcast_f<LPQ, 2 % LPQ>(el[2 / LPQ]); e[3] = group_bcast_f<LPQ, 3 % LPQ>(el[3 / LPQ]);
    e[4] = group_bcast_f<LPQ, 4 % LPQ>(el[4 / LPQ]); e[5] = group_bcast_f<LPQ, 5 % LPQ>(el[5 / LPQ]);
    e[6] = group_bcast_f<LPQ, 6 % LPQ>(el[6 / LPQ]);
#pragma unroll
    for (int w = 0; w < 7; w++) e[w] = (K[w] != kPkInf && e[w] <= g.max_d2) ? e[w] : INF;  // acceptance: d2 <= max_d2 (quirk A5)
  }
  // Exact ranks of this lane's winners and the exact 5th distance.  The keys are in ascending order, so the exact order can differ
  // from the key order only where the distance bits of neighbouring keys agree (a wavefront without such a pair skips the ranking),
  // and there an equal exact distance keeps the key order (position = visiting order).
  int rk[G::NW];
#pragma unroll
  for (int i = 0; i < G::NW; i++) rk[i] = sub + LPQ * i;
  float d5 = e[4];
  if (__any(tie)) {
    int rank[7];
#pragma unroll
    for (int w = 0; w < 7; w++) rank[w] = 0;
#pragma unroll
    for (int v = 0; v < 7; v++)
#pragma unroll
      for (int w = v + 1; w < 7; w++) {
        const bool swapped = e[v] > e[w];
        rank[w] += swapped ? 0 : 1;
        rank[v] += swapped ? 1 : 0;
      }
    d5 = INF;
#pragma unroll
    for (int w = 0; w < 7; w++) d5 = rank[w] == 4 ? e[w] : d5;
    static_for<G::NW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      rk[i] = pick_by_lane<LPQ, LPQ * i>(rank, sub);
    });
  }
  const int found = e[4] < INF ? 5 : (e[3] < INF ? 4 : (e[2] < INF ? 3 : (e[1] < INF ? 2 : (e[0] < INF ? 1 : 0))));
  const bool amb = d7t < INF && !(d5 < d7t);
  const bool need = active && (ovf || amb || !(fminf(d5, g.max_d2) <= guardsq));
  if (live) {
    static_for<G::NW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const float ei = pick_by_lane<LPQ, LPQ * i>(e, sub);
      if (sub + LPQ * i < 7 && ei < INF && rk[i] < 5) rb.nbr[(size_t)rk[i] * rb.cap + qi] = make_float4(W[i].x, W[i].y, W[i].z, ei);
    });
    if (found < 5) {  // the missing neighbours read (0, 0, 0, inf)
#pragma unroll
      for (int r = 0; r < 5; r += LPQ)
        if (sub + r < 5 && sub + r >= found) rb.nbr[(size_t)(sub + r) * rb.cap + qi] = make_float4(0.f, 0.f, 0.f, INF);
    }
    if (sub == (LPQ > 1 ? 1 : 0)) {
      const int cflags = found | (need ? (kNeedy | ((ovf || amb) ? 0 : kCovered)) : 0);
      rb.nbr_count[qi] = cflags;
      if (need && epoch > 0) {  // listed for the completion workgroups of the fit launch behind this one (~80 of 95 k queries)
        const int at = atomicAdd(&rb.flag_count[epoch & 1], 1);
        if (at < kListCap) {
          float4* e = rb.flag_list + 2 * ((epoch & 1) * kListCap + at);
          e[0] = make_float4(wx, wy, wz, __int_as_float(qi));
          e[1] = make_float4(__int_as_float(cflags), 0.f, 0.f, 0.f);
        }
      }
    }
    if (sub == (LPQ == 4 ? 2 : 0)) rb.world[qi] = make_float4(wx, wy, wz, 0.f);
  }
}



// Second stage of the search for a flagged query, run by ONE WAVEFRONT (four flagged queries of a workgroup proceed
// concurrently): every cell that intersects the ball of radius sqrt(min(d5 of stage 1, max_d2)) is visited — looked up through
// the 3x3x3 neighbourhood of 8x8x8-cell blocks (27 block ids held one per lane and fetched by shuffle), so empty space costs
// nothing.  Two passes: the 3x3x3 cells around the query first; their exact top-5 gives the bound that prunes the (up to
// ~1300) outer cells.  The per-lane sorted lists are merged by five rounds of "wave-wide minimum of the list heads, owner
// pops" — ~25 instructions per round instead of a 6-step butterfly of 5-element insertions.  The result is wave-uniform.
// (reductions over the wavefront on DPP row operations: ~60 cycles where six rounds of __shfl_xor - LDS permutes - take ~700; the
// completion of ONE query runs ten of them, on the critical path of the fit launch)
// Measurement builds (-DLII_FALLBACK_TRACE, tools/ab_build.sh): where a completion's time goes - 100 MHz stamps around the phases of
// complete_one / knn_fallback_wave, summed per kind of query (0: a ball of more than 256 cells - the queries that look past the
// map's edge; 1: the others) and read back by lii_destroy.  [kind * 8 + 0] queries, [+1] head -> inner list, [+2] cell entries -> list,
// [+3] candidates, [+4] selection, [+5] winners stored; [16 ..] per completion workgroup with work: count, head, completions, fit + sums.
#ifdef LII_FALLBACK_TRACE
__device__ unsigned long long g_fb_trace[32];
#define LII_FB_TS(v) const long long v = wall_clock64()
__device__ __forceinline__ void fb_add(int at, long long v) { if ((threadIdx.x & 63) == 0) atomicAdd(&g_fb_trace[at], (unsigned long long)v); }
#else
#define LII_FB_TS(v)
#endif
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned int dpp_u32(unsigned int old, unsigned int v) {
  return (unsigned int)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ float wave_min_f32(float v) {  // v >= 0 or +inf: the bit patterns order like the values
  unsigned int x = __float_as_uint(v);
  x = min(x, dpp_u32<0xB1, 0xF>(x, x));    // quad_perm [1, 0, 3, 2]
  x = min(x, dpp_u32<0x4E, 0xF>(x, x));    // quad_perm [2, 3, 0, 1]
  x = min(x, dpp_u32<0x141, 0xF>(x, x));   // row_half_mirror
  x = min(x, dpp_u32<0x140, 0xF>(x, x));   // row_mirror: every lane of a row of 16 holds the row's minimum
  x = min(x, dpp_u32<0x142, 0xA>(x, x));   // row_bcast15 into rows 1 and 3
  x = min(x, dpp_u32<0x143, 0xC>(x, x));   // row_bcast31 into rows 2 and 3: lane 63 holds the minimum of all
  return __uint_as_float((unsigned int)__builtin_amdgcn_readlane((int)x, 63));
}
// pops the 5 smallest (d, idx) over the lists of all lanes into uniform arrays; lists must be sorted ascending (Knn5 is)
__device__ __forceinline__ void wave_select5(Knn5 k, float (&od)[5], int (&oi)[5]) {
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const float m = wave_min_f32(k.d0);
    const unsigned long long owners = __ballot(k.d0 == m && k.i0 != -1);  // (-1: empty; <= -2: a seeded entry, see knn_fallback_wave)
    if (owners == 0ull) { od[r] = __builtin_inff(); oi[r] = -1; continue; }  // fewer than r + 1 candidates in total
    const int owner = __ffsll((long long)owners) - 1;
    od[r] = m;
    oi[r] = __builtin_amdgcn_readlane(k.i0, owner);  // (owner is uniform)
    if ((int)(threadIdx.x & 63) == owner) {
      k.d0 = k.d1; k.d1 = k.d2; k.d2 = k.d3; k.d3 = k.d4; k.d4 = __builtin_inff();
      k.i0 = k.i1; k.i1 = k.i2; k.i2 = k.i3; k.i3 = k.i4; k.i4 = -1;
    }
  }
}
// The candidates of NB cell ranges against a lane's list (the far pass of knn_fallback_wave): four loads in flight per range.
template <int NB>
__device__ __forceinline__ void far_candidates(const GridView& g, const uint2 (&rr)[NB], float wx, float wy, float wz, float bound1, Knn5& k) {
#pragma unroll
  for (int b = 0; b < NB; b++) {
    // candidates must beat the inner 5th distance as well as the lane's own list
    for (unsigned int j = rr[b].x; j < rr[b].y; j += 4) {
      const unsigned int last = rr[b].y - 1;
      const float4 p0 = g.pts[j], p1 = g.pts[min(j + 1, last)], p2 = g.pts[min(j + 2, last)], p3 = g.pts[min(j + 3, last)];
      const float d0 = dist2_ref(wx, wy, wz, p0.x, p0.y, p0.z), d1 = dist2_ref(wx, wy, wz, p1.x, p1.y, p1.z);
      const float d2 = dist2_ref(wx, wy, wz, p2.x, p2.y, p2.z), d3 = dist2_ref(wx, wy, wz, p3.x, p3.y, p3.z);
      if (d0 <= g.max_d2 && d0 < fminf(k.d4, bound1)) knn_insert(k, d0, (int)j);
      if (j + 1 <= last && d1 <= g.max_d2 && d1 < fminf(k.d4, bound1)) knn_insert(k, d1, (int)(j + 1));
      if (j + 2 <= last && d2 <= g.max_d2 && d2 < fminf(k.d4, bound1)) knn_insert(k, d2, (int)(j + 2));
      if (j + 3 <= last && d3 <= g.max_d2 && d3 < fminf(k.d4, bound1)) knn_insert(k, d3, (int)(j + 3));
    }
  }
}
// The far pass gathers its candidates through a LIST (round 5).  Round 4 had every lane scan the cells of its own columns one
// after the other - one dependent round trip per non-empty cell and four candidates: a query that looks past the edge of the map
// (a 2.2 m ball: 11 x 11 x 11 cells, a hundred of them occupied, most lanes holding none and a few holding several) cost its
// workgroup 18 us of candidate loads.  Now the lanes only LIST what they find - every surviving cell range cut into chunks of up to
// four points, word = first map index << 3 | points, appended to a per-wavefront list in LDS behind a prefix sum over the lanes - and
// the wavefront scans the list together, one chunk (four loads in flight) per lane and round: a hundred occupied cells are 225
// chunks, four rounds.
constexpr int kFarCap = 2048;  // words of the list (it lives in the LDS the launch's final reduction uses later: ReduceShared::row)
constexpr int kFarUse = kFarCap - 1;  // chunks it holds: the last word takes the writes of far_list_write that have nothing to say
__device__ __forceinline__ unsigned int wave_excl_prefix_u32(unsigned int v, unsigned int* total) {
  // inclusive scan inside the rows of 16 (row_shr 1, 2, 4, 8: lanes shifted in from outside a row read 0), then the totals of the rows
  // in front (row_bcast15, row_bcast31)
  unsigned int x = v;
  x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
  x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
  x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
  x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
  x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
  x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
  *total = (unsigned int)__builtin_amdgcn_readlane((int)x, 63);
  return x - v;
}
// the chunks of NB cell ranges to list[at ...]
// (round 6: the first two chunks of a cell - eight points, most cells hold no more - are written without a loop, the rest behind ONE
// wave-wide test.  The first form ran a loop per cell range: NB loops of divergent trip counts per lane and trip, ~800 instructions
// of a wavefront that runs alone on its SIMD - half of the 5.4 us "cell entries -> list" of a query at the map's edge.)
template <int NB>
__device__ __forceinline__ void far_list_write(unsigned int* __restrict__ list, unsigned int at, const uint2 (&rr)[NB]) {
  const unsigned int at0 = at;
  bool more = false;
#pragma unroll
  for (int b = 0; b < NB; b++) {  // (no branch: a chunk that does not exist goes to the list's last word, which is never used - kFarUse)
    const unsigned int n = rr[b].y - rr[b].x;
    list[n > 0u ? at : (unsigned)kFarUse] = (rr[b].x << 3) | min(4u, n);
    list[n > 4u ? at + 1u : (unsigned)kFarUse] = ((rr[b].x + 4u) << 3) | min(4u, n - 4u);
    more = more || n > 8u;
    at += (n + 3u) >> 2;
  }
  if (__any(more)) {  // (uniform) cells of more than eight points: their other chunks
    at = at0;
#pragma unroll
    for (int b = 0; b < NB; b++) {
      unsigned int a2 = at + 2u;
      for (unsigned int j = rr[b].x + 8u; j < rr[b].y; j += 4u) list[a2++] = (j << 3) | min(4u, rr[b].y - j);
      at += (rr[b].y - rr[b].x + 3u) >> 2;
    }
  }
}
// every candidate of list[0 .. n) against the lanes' lists (uniform call)
__device__ __forceinline__ void far_list_scan(const GridView& g, const unsigned int* __restrict__ list, unsigned int n, float wx, float wy, float wz,
                                              float bound1, Knn5& k) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // (one chunk - four loads in flight - per lane and round: the launch this runs in, k_fit_reduce, must keep its three wavefronts per
  // SIMD; two chunks per lane cost 24 more vector registers and the 500 k-point scan 2 us per fit launch)
  for (unsigned int i0 = 0; i0 < n; i0 += 64u) {
    const unsigned int w = i0 + lane < n ? list[i0 + lane] : 0u;
    F3 P[4];
#pragma unroll
    for (int u = 0; u < 4; u++) P[u] = load_xyz(g.pts, (w >> 3) + min((unsigned)u, (w & 7u) - 1u));  // (an empty word reads slots 0 .. 3 and discards them)
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float d = dist2_ref(wx, wy, wz, P[u].x, P[u].y, P[u].z);
      if ((unsigned)u < (w & 7u) && d <= g.max_d2 && d < fminf(k.d4, bound1)) knn_insert(k, d, (int)((w >> 3) + (unsigned)u));
    }
  }
  __builtin_amdgcn_wave_barrier();  // (the list is rewritten behind this call)
}
// One trip of the far pass: the lanes' NB surviving cell ranges join the list (scanned first if it is full; a trip that would
// overflow the list on its own - cells of hundreds of points - is scanned lane by lane as in round 4).
template <int NB>
__device__ __forceinline__ void far_list_trip(const GridView& g, unsigned int* __restrict__ list, unsigned int& n_list, const uint2 (&rr)[NB], float wx, float wy,
                                              float wz, float bound1, Knn5& k) {
  unsigned int mine = 0u;
#pragma unroll
  for (int b = 0; b < NB; b++) mine += (rr[b].y - rr[b].x + 3u) >> 2;
  if (!__any(mine != 0u)) return;  // (uniform) nothing in this trip: most trips of a ball that reaches past the map
  unsigned int total;
  const unsigned int before = wave_excl_prefix_u32(mine, &total);
  if (total > (unsigned)kFarUse) {
    far_candidates<NB>(g, rr, wx, wy, wz, bound1, k);
    return;
  }
  if (n_list + total > (unsigned)kFarUse) {
    far_list_scan(g, list, n_list, wx, wy, wz, bound1, k);
    n_list = 0u;
  }
  far_list_write<NB>(list, n_list + before, rr);
  n_list += total;
}
// seeded: the search pass has already measured every point of the 3 x 3 x 3 cells around the query (kCovered) - its list
// (seed_d: the distance of entry `lane` on lanes 0..4, inf where the list ends) stands in for pass 1; a seeded entry that
// survives comes back as index -(2 + its place in the list).
template <int nparts = 1>
__device__ __forceinline__ void knn_fallback_wave(const GridView& g, float wx, float wy, float wz, bool has5, float list_d, bool seeded,
                                                  float seed_d, float (&od)[5], int (&oi)[5], unsigned int* __restrict__ far_list /* [kFarCap] LDS, this wavefront's */
#ifdef LII_FALLBACK_TRACE
                                                  , long long fb_t0, int* fb_kind
#endif
                                                  , int part = 0) {
  // part / nparts (uniform): FOUR wavefronts finish one query together (complete_one_coop) - every one of them runs this routine, the head
  // and the inner pass alike (so they agree on the bound), and takes every fourth pair of cell entries of the far pass' rows; part 0 also
  // carries the inner result.  The four results are joined by the caller.
  const int lane = threadIdx.x & 63;
  // (the block probes below need the query only: they are issued before the search pass's list - list_d: entry `lane`'s distance
  // on lanes 0..4 - is looked at)
  const float cs = g.cs;
  const float eps = 1e-6f * (fabsf(wx) + fabsf(wy) + fabsf(wz) + 8.f);
  const int cx = cell_of(wx, g.inv_cs), cy = cell_of(wy, g.inv_cs), cz = cell_of(wz, g.inv_cs);
  const int X0 = cx >> kCoarseShift, Y0 = cy >> kCoarseShift, Z0 = cz >> kCoarseShift;
  // (with the dense window - GridView::win - the far pass reads whole ROWS of cell entries, see below; the 27 block probes stay: which
  // blocks exist around the query clips the cube far tighter than the window's box - a query on the floor of a hall has no block
  // above it, the box reaches to the ceiling: 588 against 1204 cells per edge query on the bench stream)
  const bool windowed = g.win != nullptr;  // uniform
  const int my_block = lane < 27 ? find_block(g, X0 + (lane % 3) - 1, Y0 + ((lane / 3) % 3) - 1, Z0 + (lane / 9) - 1) : -1;
  auto win_entry = [&](int ixx, int iyy, int izz) -> size_t {  // index of a cell INSIDE the window
    return ((size_t)(izz - g.wz0) * (size_t)g.wny + (size_t)(iyy - g.wy0)) * (size_t)g.wnx + (size_t)(ixx - g.wx0);
  };
  const float d5 = has5 ? __shfl(list_d, 4) : __builtin_inff();
  const float bound0 = fminf(d5, g.max_d2);
  const float r0 = sqrtf(bound0) + 2.f * eps;
  Knn5 k;
  k.d0 = k.d1 = k.d2 = k.d3 = k.d4 = __builtin_inff();
  k.i0 = k.i1 = k.i2 = k.i3 = k.i4 = -1;
  // pass 1: the 3 x 3 x 3 cells around the query, one per lane
  if (seeded) {  // (uniform)
#pragma unroll
    for (int j = 0; j < 5; j++) {
      od[j] = __shfl(seed_d, j);
      oi[j] = od[j] < __builtin_inff() ? -(2 + j) : -1;
    }
  } else {
    const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = (lane / 9) % 3 - 1;
    const int ixx = cx + dx, iyy = cy + dy, izz = cz + dz;
    const int X = (ixx >> kCoarseShift) - X0 + 1, Y = (iyy >> kCoarseShift) - Y0 + 1, Z = (izz >> kCoarseShift) - Z0 + 1;
    const int id = __shfl(my_block, Z * 9 + Y * 3 + X);  // X, Y, Z in 0..2: the neighbour cells stay inside the 3x3x3 blocks
    if (lane < 27 && id >= 0) {
      const float gx = axis_gap(wx, ixx, cs, eps), gy = axis_gap(wy, iyy, cs, eps), gz = axis_gap(wz, izz, cs, eps);
      if (!(gx * gx + gy * gy + gz * gz > bound0)) {
        const unsigned local = (((unsigned)izz & 7u) << 6) | (((unsigned)iyy & 7u) << 3) | ((unsigned)ixx & 7u);
        const uint2 rr = g.cells[(size_t)id * kBlockCells + local];
        scan_range(g.pts, g.max_d2, rr.x, rr.y, wx, wy, wz, k);
      }
    }
    wave_select5(k, od, oi);
  }
  LII_FB_TS(fb_t1);
  const float bound1 = fminf(od[4], bound0);  // exact 5th distance over the inner cells (inf if they hold fewer than 5)
  // pass 2: the rest of the cube around the ball of radius sqrt(bound1) (nothing farther can enter the list), pruned with bound1
  const float r1 = fminf(r0, sqrtf(bound1) + 2.f * eps);
  int ix0 = cell_of(wx - r1, g.inv_cs), ix1 = cell_of(wx + r1, g.inv_cs);
  int iy0 = cell_of(wy - r1, g.inv_cs), iy1 = cell_of(wy + r1, g.inv_cs);
  int iz0 = cell_of(wz - r1, g.inv_cs), iz1 = cell_of(wz + r1, g.inv_cs);
  {
    // The cube is clipped to the blocks that EXIST among the 27 around the query (round 5): a query that looks past the edge of the map
    // walks a ball of 11 x 11 x 11 cells of which only the slab on the map's side can hold a point - the cells of a missing block were
    // visited for nothing (four trips of cell-entry loads per query, 5 us of the ~8 its completion took).  Which of the three block
    // layers per axis hold an existing block comes out of one ballot (lane l < 27 holds block (l % 3, l / 3 % 3, l / 9)).
    const unsigned int ex = (unsigned int)__ballot(lane < 27 && my_block >= 0);
    constexpr unsigned int MX = 0x1249249u, MY = 0x1C0E07u, MZ = 0x1FFu;  // layer 0 of x (lanes 0, 3, 6, ...), of y (0-2, 9-11, 18-20), of z (0-8)
    auto layer_range = [](unsigned int e0, unsigned int e1, unsigned int e2, int& lo, int& hi) {  // first and last layer with a block; lo > hi: none
      lo = e0 ? 0 : (e1 ? 1 : (e2 ? 2 : 3));
      hi = e2 ? 2 : (e1 ? 1 : (e0 ? 0 : -1));
    };
    int lx, hx, ly, hy, lz, hz;
    layer_range(ex & MX, ex & (MX << 1), ex & (MX << 2), lx, hx);
    layer_range(ex & MY, ex & (MY << 3), ex & (MY << 6), ly, hy);
    layer_range(ex & MZ, ex & (MZ << 9), ex & (MZ << 18), lz, hz);
    // block layer L of an axis covers the cells [(B0 + L - 1) * 8, (B0 + L - 1) * 8 + 7]
    ix0 = max(ix0, (X0 + lx - 1) << kCoarseShift); ix1 = min(ix1, ((X0 + hx - 1) << kCoarseShift) + 7);
    iy0 = max(iy0, (Y0 + ly - 1) << kCoarseShift); iy1 = min(iy1, ((Y0 + hy - 1) << kCoarseShift) + 7);
    iz0 = max(iz0, (Z0 + lz - 1) << kCoarseShift); iz1 = min(iz1, ((Z0 + hz - 1) << kCoarseShift) + 7);
  }
  const int nx = max(ix1 - ix0 + 1, 0), ny = max(iy1 - iy0 + 1, 0), nz = max(iz1 - iz0 + 1, 0);
  const int total = nx * ny * nz;
#ifdef LII_FALLBACK_TRACE
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the block probes are in)
  long long fb_t1a = wall_clock64(), fb_ld = 0, fb_tr = 0;
#endif
  k.d0 = k.d1 = k.d2 = k.d3 = k.d4 = __builtin_inff();
  k.i0 = k.i1 = k.i2 = k.i3 = k.i4 = -1;
  {  // the inner result re-enters as five one-element lists (lanes 0..4)
    const float dd = lane == 0 ? od[0] : (lane == 1 ? od[1] : (lane == 2 ? od[2] : (lane == 3 ? od[3] : od[4])));
    const int ii = lane == 0 ? oi[0] : (lane == 1 ? oi[1] : (lane == 2 ? oi[2] : (lane == 3 ? oi[3] : oi[4])));
    if (part == 0 && lane < 5 && ii != -1) { k.d0 = dd; k.i0 = ii; }
  }
  // Four rounds of 64 cells at a time: which cells survive (inside the ball, outside the inner cube, closer than bound1) does
  // not depend on the candidates found on the way, so the four cell entries of a lane are fetched together and their
  // candidates four per trip - the loop is a chain of dependent loads, not of arithmetic.
  const int nxy = nx * ny;
  unsigned int n_far = 0u;  // chunks on the list (uniform)
  // the clipped cube lies inside the window?  (it does whenever the window is the map's: the cube is clipped to blocks that exist, the
  // window covers them all and a block of margin - checked all the same, the row loads below rely on it)
  const bool rows_ok = windowed && total > 0 && ix0 - 1 >= g.wx0 && ix1 + 1 < g.wx0 + g.wnx && iy0 >= g.wy0 && iy1 < g.wy0 + g.wny && iz0 >= g.wz0 &&
                       iz1 < g.wz0 + g.wnz;
  if (rows_ok && nparts == 4) {
    // ROW TRIPS, a quarter of every row: the pairs of entries t = part, part + 4 of the seven a row trip reads (see below) - a lane's
    // instruction count, which is what this pass costs (profiles/r06_edge.md), falls to a third, and every wavefront scans a list of its own.
    constexpr int NV = 7, NVP = 2;
    const int nrows = ny * nz;
    for (int r0 = 0; r0 < nrows; r0 += 64) {                 // uniform trip counts
      const int r = r0 + lane;
      const bool in_row = r < nrows;
      const int rz = (in_row ? r : 0) / ny, ry = (in_row ? r : 0) - rz * ny;
      const int iyy = iy0 + ry, izz = iz0 + rz;
      const float gy = axis_gap(wy, iyy, cs, eps), gz = axis_gap(wz, izz, cs, eps);
      const float gyz = gy * gy + gz * gz;
      const bool row_ok = in_row && !(gyz > bound1);
      const bool yz_inner = abs(iyy - cy) <= 1 && abs(izz - cz) <= 1;
      const int xs0 = ix0 - ((ix0 - g.wx0) & 1);
      for (int xs = xs0; xs <= ix1; xs += 2 * NV) {          // uniform
        const size_t base = win_entry(xs, iyy, izz);
        bool act[2 * NVP];
        uint4 v[NVP];
#pragma unroll
        for (int tt = 0; tt < NVP; tt++) {
          const int t = tt * 4 + part;                       // (uniform)
#pragma unroll
          for (int hh = 0; hh < 2; hh++) {
            const int ixx = xs + 2 * t + hh;
            const bool x_in = t < NV && ixx >= ix0 && ixx <= ix1, x_inner = abs(ixx - cx) <= 1;
            const float gx = axis_gap(wx, ixx, cs, eps);
            act[2 * tt + hh] = row_ok && x_in && !(yz_inner && x_inner) && !(gyz + gx * gx > bound1);  // (the same sum as the whole-row form)
          }
          v[tt] = *reinterpret_cast<const uint4*>(g.win + ((act[2 * tt] || act[2 * tt + 1]) ? base + 2 * (size_t)t : (size_t)0));
        }
        uint2 rr[2 * NVP];
#pragma unroll
        for (int tt = 0; tt < NVP; tt++) {
          rr[2 * tt] = act[2 * tt] ? make_uint2(v[tt].x, v[tt].y) : make_uint2(0u, 0u);
          rr[2 * tt + 1] = act[2 * tt + 1] ? make_uint2(v[tt].z, v[tt].w) : make_uint2(0u, 0u);
        }
        far_list_trip<2 * NVP>(g, far_list, n_far, rr, wx, wy, wz, bound1, k);
      }
    }
  } else if (nparts > 1 && part != 0) {
    // (no window: the far pass over the hashed tables is not split - part 0 walks it alone, the others have nothing to add)
  } else if (rows_ok) {
    // ROW TRIPS (round 6).  In the window the cells of a row (x running) lie next to each other: a lane takes one (y, z) row of the
    // cube and reads it with 16-byte loads, two cell entries each - the 11 x 11 x 5 cube of a query at the map's edge is 55 rows, ONE
    // trip of seven loads per lane, where the column walk over the hashed tables took three trips of eight (phase stamps,
    // profiles/r06_edge.md: 6.6 of the 11.9 us of such a completion).  Cells of blocks that do not exist read as empty entries.
    constexpr int NV = 7;  // 14 cells of a row per trip (a ball of sqrt(max_d2) spans 11 - 12 at the default cell edge)
    const int nrows = ny * nz;
    for (int r0 = 0; r0 < nrows; r0 += 64) {                 // uniform trip counts
      const int r = r0 + lane;
      const bool in_row = r < nrows;
      const int rz = (in_row ? r : 0) / ny, ry = (in_row ? r : 0) - rz * ny;
      const int iyy = iy0 + ry, izz = iz0 + rz;
      const float gy = axis_gap(wy, iyy, cs, eps), gz = axis_gap(wz, izz, cs, eps);
      const float gyz = gy * gy + gz * gz;
      const bool row_ok = in_row && !(gyz > bound1);
      const bool yz_inner = abs(iyy - cy) <= 1 && abs(izz - cz) <= 1;
      const int xs0 = ix0 - ((ix0 - g.wx0) & 1);              // the entry pairs are 16-byte aligned: rows are a multiple of eight cells long
      for (int xs = xs0; xs <= ix1; xs += 2 * NV) {          // uniform
        // (what depends on x alone is the same for every row: lane i works it out for cell xs + i, everyone reads it back - v_readlane -
        // and a cell costs an addition and a comparison instead of the ten instructions of its gap.  The completion's wavefront shares
        // its SIMD with a wavefront of the plane fit: instructions, not loads, are what it waits for - phase stamps, profiles/r06_edge.md)
        float gx2[2 * NV];
        {
          const float gxl = axis_gap(wx, xs + (lane < 2 * NV ? lane : 0), cs, eps);
          const float gx2l = gxl * gxl;
#pragma unroll
          for (int c = 0; c < 2 * NV; c++) gx2[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gx2l), c));
        }
        // which cells of the row can hold a candidate is decided FIRST: a pair of entries is only fetched when one of its cells can
        // (the corners of the cube lie outside the ball - half of its cells; a launch with a thousand unfinished queries is short of
        // memory requests, not of instructions)
        bool act[2 * NV];
#pragma unroll
        for (int c = 0; c < 2 * NV; c++) {
          const int ixx = xs + c;
          const bool x_in = ixx >= ix0 && ixx <= ix1, x_inner = abs(ixx - cx) <= 1;  // (uniform)
          act[c] = row_ok && x_in && !(yz_inner && x_inner) && !(gyz + gx2[c] > bound1);  // (the inner cube: done in pass 1)
        }
        const size_t base = win_entry(xs, iyy, izz);
        uint4 v[NV];
#pragma unroll
        for (int t = 0; t < NV; t++)  // (requested from entry 0 when not wanted, and dropped afterwards: no load behind a branch)
          v[t] = *reinterpret_cast<const uint4*>(g.win + ((act[2 * t] || act[2 * t + 1]) ? base + 2 * (size_t)t : (size_t)0));
#ifdef LII_FALLBACK_TRACE
        const long long fb_a = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the row loads have arrived)
        const long long fb_b = wall_clock64();
        fb_ld += fb_b - fb_a;
#endif
        uint2 rr[2 * NV];
#pragma unroll
        for (int t = 0; t < NV; t++) {
          rr[2 * t] = act[2 * t] ? make_uint2(v[t].x, v[t].y) : make_uint2(0u, 0u);
          rr[2 * t + 1] = act[2 * t + 1] ? make_uint2(v[t].z, v[t].w) : make_uint2(0u, 0u);
        }
#ifdef LII_FALLBACK_TRACE
        const long long fb_c = wall_clock64();
#endif
        far_list_trip<2 * NV>(g, far_list, n_far, rr, wx, wy, wz, bound1, k);
#ifdef LII_FALLBACK_TRACE
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        fb_tr += wall_clock64() - fb_c;
#endif
      }
    }
  } else if (total <= 256) {
  constexpr int NB = 4;
  for (int c0 = 0; c0 < total; c0 += 64 * NB) {  // uniform trip count: the shuffle below needs every lane
    uint2 rr[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const int c = c0 + b * 64 + lane;
      const bool in = c < total;
      const int cc = in ? c : 0;
      const int q2 = cc / nxy, r2 = cc - q2 * nxy, q1 = r2 / nx;
      const int ixx = ix0 + (r2 - q1 * nx), iyy = iy0 + q1, izz = iz0 + q2;
      const int X = (ixx >> kCoarseShift) - X0 + 1, Y = (iyy >> kCoarseShift) - Y0 + 1, Z = (izz >> kCoarseShift) - Z0 + 1;
      const bool in_blocks = (unsigned)X <= 2u && (unsigned)Y <= 2u && (unsigned)Z <= 2u;  // else farther than 8 cells >= sqrt(max_d2)
      const int id = __shfl(my_block, in_blocks ? Z * 9 + Y * 3 + X : 0);
      const bool inner = abs(ixx - cx) <= 1 && abs(iyy - cy) <= 1 && abs(izz - cz) <= 1;  // done in pass 1
      bool act = in && in_blocks && !inner && id >= 0;
      if (act) {
        const float gx = axis_gap(wx, ixx, cs, eps), gy = axis_gap(wy, iyy, cs, eps), gz = axis_gap(wz, izz, cs, eps);
        act = !(gx * gx + gy * gy + gz * gz > bound1);
      }
      rr[b] = make_uint2(0u, 0u);
      if (act) {
        const unsigned local = (((unsigned)izz & 7u) << 6) | (((unsigned)iyy & 7u) << 3) | ((unsigned)ixx & 7u);
        rr[b] = g.cells[(size_t)id * kBlockCells + local];
      }
    }
    far_list_trip<NB>(g, far_list, n_far, rr, wx, wy, wz, bound1, k);
  }
  } else {
  // Which cells survive (inside the ball, outside the inner cube, closer than bound1) does not depend on the candidates found on
  // the way.  A lane takes one (x, y) COLUMN of the cube at a time and walks it in z, NB cells per trip: one integer division per
  // column (the flat cell index of round 3 cost two per cell - a query that looks past the edge of the map walks the whole ball of
  // sqrt(max_d2) = 2.2 m, 11 x 11 x 11 cells at the default cell edge: ~ 2 500 instructions of index arithmetic per lane, 15 us
  // per search pass on the one scan of the bench stream that has such queries).  The NB cell entries of a trip are requested back
  // to back from a valid address whether the cell is wanted or not (entry 0 for the others; the answer is dropped afterwards): a
  // load behind a branch is waited for on its own.
  constexpr int NB = 8;  // (cells of a column per trip: 12 would take the 11-cell ball in one, at 12 more vector registers than k_fit_reduce can spare)
  for (int col0 = 0; col0 < nxy; col0 += 64) {  // uniform trip counts: the shuffles below need every lane
    const int col = col0 + lane;
    const bool in_col = col < nxy;
    const int q1 = (in_col ? col : 0) / nx;
    const int ixx = ix0 + ((in_col ? col : 0) - q1 * nx), iyy = iy0 + q1;
    const int X = (ixx >> kCoarseShift) - X0 + 1, Y = (iyy >> kCoarseShift) - Y0 + 1;
    const bool xy_blocks = in_col && (unsigned)X <= 2u && (unsigned)Y <= 2u;  // else farther than 8 cells >= sqrt(max_d2)
    const bool xy_inner = abs(ixx - cx) <= 1 && abs(iyy - cy) <= 1;
    const float gx = axis_gap(wx, ixx, cs, eps), gy = axis_gap(wy, iyy, cs, eps);
    const float gxy = gx * gx + gy * gy;
    for (int z0 = 0; z0 < nz; z0 += NB) {
      bool act[NB];
      size_t ci[NB];
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const int izz = iz0 + z0 + b;
        const int Z = (izz >> kCoarseShift) - Z0 + 1;
        const bool in_blocks = xy_blocks && z0 + b < nz && (unsigned)Z <= 2u;
        const int id = __shfl(my_block, in_blocks ? Z * 9 + Y * 3 + X : 0);
        const bool inner = xy_inner && abs(izz - cz) <= 1;  // done in pass 1
        const float gz = axis_gap(wz, izz, cs, eps);
        const bool a = in_blocks && !inner && id >= 0 && !(gxy + gz * gz > bound1);
        const unsigned local = (((unsigned)izz & 7u) << 6) | (((unsigned)iyy & 7u) << 3) | ((unsigned)ixx & 7u);
        act[b] = a;
        ci[b] = a ? (size_t)id * kBlockCells + local : (size_t)0;
      }
      uint2 rr[NB];
#pragma unroll
      for (int b = 0; b < NB; b++) rr[b] = g.cells[ci[b]];
#pragma unroll
      for (int b = 0; b < NB; b++) rr[b] = act[b] ? rr[b] : make_uint2(0u, 0u);
      far_list_trip<NB>(g, far_list, n_far, rr, wx, wy, wz, bound1, k);
    }
  }
  }
  LII_FB_TS(fb_t2);
  far_list_scan(g, far_list, n_far, wx, wy, wz, bound1, k);
  LII_FB_TS(fb_t3);
  wave_select5(k, od, oi);
#ifdef LII_FALLBACK_TRACE
  {
    const long long fb_t4 = wall_clock64();
    const int kind = total > 256 ? 0 : 8;
    fb_add(kind + 0, 1); fb_add(kind + 1, fb_t1 - fb_t0); fb_add(kind + 2, fb_t2 - fb_t1); fb_add(kind + 3, fb_t3 - fb_t2); fb_add(kind + 4, fb_t4 - fb_t3);
    fb_add(kind + 6, (long long)n_far); fb_add(kind + 7, (long long)total);
    if (kind == 0) { fb_add(24, fb_t1a - fb_t1); fb_add(25, fb_ld); fb_add(26, fb_tr); }
    *fb_kind = kind;
  }
#endif
}

// Equal squared distances inside the kept list are ordered by x, ascending — what the reference's heap comparator
// (PointType_CMP, include/ikd-Tree/ikd_Tree.h:50-61: ties within 1e-10 fall back to point.x) produces after
// Nearest_Search pops it.  (A tie across the 5th/6th place is resolved by visiting order in the tree and by
// cell order here; that case cannot be reproduced and is documented in DESIGN.md.)
__device__ __forceinline__ bool canon_ties(float4 (&nb)[5]) {
  if (!(nb[0].w == nb[1].w || nb[1].w == nb[2].w || nb[2].w == nb[3].w || nb[3].w == nb[4].w)) return false;  // the usual case
  bool changed = false;
#pragma unroll
  for (int pass = 0; pass < 4; pass++)
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (nb[j].w == nb[j + 1].w && nb[j].x > nb[j + 1].x) {
        float4 t = nb[j]; nb[j] = nb[j + 1]; nb[j + 1] = t;
        changed = true;
      }
  return changed;
}

// Plane fit + residual + Jacobian + block reduction, one lane per point.  FIT = right after a search pass (finishes
// the flagged searches of this workgroup's points, reads the 5 neighbours, caches the plane); !FIT for the
// non-search iterations.  `forced` as in k_knn_ck.
// Completion of the flagged searches among the kBlock points of one workgroup (`my_point`: the calling lane's): one wavefront per
// flagged query, four at a time (they are rare, ~0.07 % of the queries, but clustered at the map frontier).  Every lane of
// the workgroup must call it; on return the completed lists are visible to the whole workgroup.
// One flagged search finished by a whole wavefront: the list goes to rb.nbr / rb.nbr_count.
// (the owner lane has loaded the query's world point and count together with everything else it needs: the completion starts
// with the block probes at once - one dependent round trip less than fetching them here)
// lds_nb / lds_found (or null): the finished list and its count also go to the workgroup's LDS, for the lane that fits this query.
__device__ __forceinline__ void complete_one(const GridView& g, const RegistrationBuffers& rb, int qi, int c00, float wx, float wy, float wz,
                                             unsigned int* __restrict__ far_list, float4* __restrict__ lds_nb = nullptr,
                                             int* __restrict__ lds_found = nullptr, int mark = 0) {
  LII_FB_TS(fb_t0);
  const int lane = threadIdx.x & 63;
  const int c0 = c00 & 0xFF;
  const bool seeded = (c00 & kCovered) != 0;  // uniform
  // the list of the search pass: its 5th distance bounds the far search, and (kCovered) its entries are exact over the inner cells
  const float4 sv = lane < 5 ? rb.nbr[(size_t)lane * rb.cap + qi] : make_float4(0.f, 0.f, 0.f, __builtin_inff());
  float od[5];
  int oi[5];
#ifdef LII_FALLBACK_TRACE
  int fb_kind = 0;
  knn_fallback_wave(g, wx, wy, wz, c0 == kMatch, sv.w, seeded, lane < c0 ? sv.w : __builtin_inff(), od, oi, far_list, fb_t0, &fb_kind);
#else
  knn_fallback_wave(g, wx, wy, wz, c0 == kMatch, sv.w, seeded, lane < c0 ? sv.w : __builtin_inff(), od, oi, far_list);
#endif
  LII_FB_TS(fb_t5);
  const int idx = lane == 0 ? oi[0] : (lane == 1 ? oi[1] : (lane == 2 ? oi[2] : (lane == 3 ? oi[3] : oi[4])));
  const float dd = lane == 0 ? od[0] : (lane == 1 ? od[1] : (lane == 2 ? od[2] : (lane == 3 ? od[3] : od[4])));
  // a seeded entry that stayed in the list: its point is in the lane that loaded it
  const int from = idx <= -2 ? -(idx + 2) : 0;
  const float kx = __shfl(sv.x, from), ky = __shfl(sv.y, from), kz = __shfl(sv.z, from);
  if (lane < 5) {
    float4 v = make_float4(0, 0, 0, 0);
    if (idx >= 0) v = g.pts[idx];
    else if (idx <= -2) v = make_float4(kx, ky, kz, 0.f);
    v.w = dd;
    rb.nbr[(size_t)lane * rb.cap + qi] = v;
    if (lds_nb) lds_nb[lane] = v;
  } else if (lane == 5) {
    const int found = (oi[0] != -1) + (oi[1] != -1) + (oi[2] != -1) + (oi[3] != -1) + (oi[4] != -1);
    rb.nbr_count[qi] = found | mark;  // (mark: kDone from a completion workgroup - see lii_device.h)
    if (lds_found) *lds_found = found;
  }
#ifdef LII_FALLBACK_TRACE
  fb_add(fb_kind + 5, wall_clock64() - fb_t5);
#endif
}
// ONE query finished by the four wavefronts of a workgroup together (round 6; a completion workgroup that holds a single query - up to
// kCompletionBlocksPre listed queries are dealt one per workgroup).  Every wavefront runs knn_fallback_wave on its quarter of the far pass;
// the four results (five candidates each, part 0's with the inner list) meet in LDS and the first wavefront selects the five nearest and
// stores the list as complete_one does.  Every lane of the workgroup must call it.
struct CoopMerge { float d[20]; int i[20]; };
__device__ __forceinline__ void complete_one_coop(const GridView& g, const RegistrationBuffers& rb, int qi, int c00, float wx, float wy, float wz,
                                                  unsigned int* __restrict__ far_lists, CoopMerge& mg, float4* __restrict__ lds_nb,
                                                  int* __restrict__ lds_found, int mark) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = c00 & 0xFF;
  const bool seeded = (c00 & kCovered) != 0;  // uniform
  const float4 sv = lane < 5 ? rb.nbr[(size_t)lane * rb.cap + qi] : make_float4(0.f, 0.f, 0.f, __builtin_inff());
  float od[5];
  int oi[5];
#ifdef LII_FALLBACK_TRACE
  int fb_kind = 0;
  knn_fallback_wave<4>(g, wx, wy, wz, c0 == kMatch, sv.w, seeded, lane < c0 ? sv.w : __builtin_inff(), od, oi, far_lists + wave * kFarCap, wall_clock64(), &fb_kind, wave);
#else
  knn_fallback_wave<4>(g, wx, wy, wz, c0 == kMatch, sv.w, seeded, lane < c0 ? sv.w : __builtin_inff(), od, oi, far_lists + wave * kFarCap, wave);
#endif
  if (lane < 5) {
    mg.d[wave * 5 + lane] = lane == 0 ? od[0] : (lane == 1 ? od[1] : (lane == 2 ? od[2] : (lane == 3 ? od[3] : od[4])));
    mg.i[wave * 5 + lane] = lane == 0 ? oi[0] : (lane == 1 ? oi[1] : (lane == 2 ? oi[2] : (lane == 3 ? oi[3] : oi[4])));
  }
  __syncthreads();
  if (wave != 0) return;
  Knn5 k;
  k.d0 = k.d1 = k.d2 = k.d3 = k.d4 = __builtin_inff();
  k.i0 = k.i1 = k.i2 = k.i3 = k.i4 = -1;
  if (lane < 20) { k.d0 = mg.d[lane]; k.i0 = mg.i[lane]; }  // twenty one-element lists: part 0's entries in the lowest lanes (they win ties, as in one wavefront)
  wave_select5(k, od, oi);
  const int idx = lane == 0 ? oi[0] : (lane == 1 ? oi[1] : (lane == 2 ? oi[2] : (lane == 3 ? oi[3] : oi[4])));
  const float dd = lane == 0 ? od[0] : (lane == 1 ? od[1] : (lane == 2 ? od[2] : (lane == 3 ? od[3] : od[4])));
  const int from = idx <= -2 ? -(idx + 2) : 0;  // a seeded entry that stayed in the list: its point is in the lane that loaded it
  const float kx = __shfl(sv.x, from), ky = __shfl(sv.y, from), kz = __shfl(sv.z, from);
  if (lane < 5) {
    float4 v = make_float4(0, 0, 0, 0);
    if (idx >= 0) v = g.pts[idx];
    else if (idx <= -2) v = make_float4(kx, ky, kz, 0.f);
    v.w = dd;
    rb.nbr[(size_t)lane * rb.cap + qi] = v;
    if (lds_nb) lds_nb[lane] = v;
  } else if (lane == 5) {
    const int found = (oi[0] != -1) + (oi[1] != -1) + (oi[2] != -1) + (oi[3] != -1) + (oi[4] != -1);
    rb.nbr_count[qi] = found | mark;
    if (lds_found) *lds_found = found;
  }
}
constexpr int kHandOver = 64;  // queries of a completion workgroup whose finished lists reach the fitting lane through LDS (usually all: 2 ... 8 per workgroup)
struct NeedyShared {
  int point[kBlock], count[kBlock];
  float w[kBlock][3];
  int aux[kBlock], key[kBlock];  // (completion workgroups: the listed queries in ascending order; as listed)
  float4 nb[kHandOver][5], body[kHandOver];
  int found[kHandOver];
  int n;
  CoopMerge merge;
};
// `count`, `w`: nbr_count and world point of the calling lane's query (loaded by the caller, together).
// far_lists: kBlock / 64 lists of kFarCap words (LDS), one per wavefront
__device__ __forceinline__ void complete_flagged(const GridView& g, const RegistrationBuffers& rb, int my_point, bool live, int count, float4 w,
                                                 NeedyShared& sh, unsigned int* __restrict__ far_lists) {
  if (threadIdx.x == 0) sh.n = 0;
  __syncthreads();
  if (live && (count & kNeedy)) {
    const int at = atomicAdd(&sh.n, 1);
    sh.point[at] = my_point; sh.count[at] = count;
    sh.w[at][0] = w.x; sh.w[at][1] = w.y; sh.w[at][2] = w.z;
  }
  __syncthreads();
  const int nn = sh.n;
  const int wave = threadIdx.x >> 6;
  for (int e = wave; e < nn; e += kBlock / 64) complete_one(g, rb, sh.point[e], sh.count[e], sh.w[e][0], sh.w[e][1], sh.w[e][2], far_lists + wave * kFarCap);
  if (nn) __syncthreads();  // the completed lists are visible to their owners (workgroup-scope release/acquire)
}

// Which point a lane of the fit pass takes: workgroup b of nb takes the chunks b, b + nb, b + 2 nb, ... of 4 consecutive points
// (4 lanes read 64 contiguous bytes).  Flagged searches cluster - a stretch of the scan that looks past the edge of the map -
// and a workgroup completes its own, four at a time: with 256 consecutive points per workgroup one of them can sit on dozens
// (the fit pass of a scan in sweep order took 50 us instead of 19); dealt out in chunks a cluster of 500 points is shared by 125
// workgroups.  (Which workgroup sums which points is fixed either way: the sums stay deterministic.)
constexpr int kFitChunkShift = 2;
__device__ __forceinline__ int fit_point_of(int blk, int nb) {
  return ((((int)threadIdx.x >> kFitChunkShift) * nb + blk) << kFitChunkShift) + ((int)threadIdx.x & ((1 << kFitChunkShift) - 1));
}

// The completion alone, over the whole cloud (lii_map_incremental of a sharded job: the blocks of the other ranks were searched
// by a stand-alone k-NN pass, not by a fit pass).
__global__ __launch_bounds__(kBlock) void k_knn_complete(GridView g, RegistrationBuffers rb) {
  __shared__ NeedyShared sh;
  __shared__ unsigned int s_far[(kBlock / 64) * kFarCap];
  int lo, n_live;
  shard_range(rb, lo, n_live);
  const int q = fit_point_of(blockIdx.x, (int)gridDim.x);
  const bool live = q < n_live;
  const int count = live ? rb.nbr_count[lo + q] : 0;
  const float4 w = live ? rb.world[lo + q] : make_float4(0.f, 0.f, 0.f, 0.f);
  complete_flagged(g, rb, lo + q, live, count, w, sh, s_far);
}

// A search pass that left MORE unfinished queries than the completion workgroups of the fit launch take (kFlagCap; a sensor that looks into
// unmapped space: hundreds to thousands per pass) - round 6: the listed queries (up to kListCap, in the order the search pass listed them)
// are finished HERE, one wavefront each, in a launch of its own between the search and the fit launch.  The fit launch then finds no
// flagged point among its own (beyond kListCap: the few that are left) and runs as on any other scan - every point is fitted and summed
// by its own lane, in the cloud's order: the same bits as with every workgroup finishing its own (31 - 33 us per fit launch there).
// The host enqueues this launch only behind the search launches of a scan whose predecessor listed more than kFlagCap (IekfResult::unfinished);
// it returns at once when the pass did not search or listed no more than kFlagCap (those are the completion workgroups').
__global__ __launch_bounds__(kBlock) void k_complete_listed(GridView g, RegistrationBuffers rb, const IekfCtrl* __restrict__ ctrl, int forced, int epoch) {
  __shared__ unsigned int s_far[(kBlock / 64) * kFarCap];
  const int wave = threadIdx.x >> 6;
  const int e = (int)blockIdx.x * (kBlock / 64) + wave;
  const float4* ent = rb.flag_list + 2 * ((epoch & 1) * kListCap + e);
  const float4 l0 = ent[0];  // (requested beside the flags; dropped when the entry is not there)
  const int cflags = __float_as_int(ent[1].x);
  const int n_listed = rb.flag_count[epoch & 1];
  if (forced >= 0) {
    if (!forced) return;
  } else {
    if (ctrl->stop || !ctrl->search_next) return;  // (as k_fit_reduce decides whether it stands behind an executed search pass)
  }
  if (n_listed <= kFlagCap || e >= min(n_listed, kListCap)) return;  // (uniform per wavefront)
  complete_one(g, rb, __float_as_int(l0.w), cflags, l0.x, l0.y, l0.z, s_far + wave * kFarCap);
}

// POSE_V: the pose lives in vector registers (204 VGPRs: two wavefronts per SIMD - no matter while the launch has no more than two per
// SIMD to offer, i.e. up to ~130 k points) instead of scalar ones (160 VGPRs: three per SIMD, but 56 of the pose's scalar registers
// are spilled into vector-register lanes and fetched back one v_readlane at a time).  Measured: 100 k-point scan 5.9 against 6.5 us
// per cached-plane launch, 500 k-point scan 18.4 against 17.6 (profiles/r05_head_loads.md): launch_fit_reduce picks by size.
// PRE: the launch stands behind a search launch (the host knows when it enqueues it): a lane requests its five neighbours at the head,
// together with everything else, instead of its cached plane and selection flag - the fit of a search pass is then one round trip
// (head) instead of two (head, lists) in front of the QR.  Should the device not search after all, the plane is fetched late.
template <bool POSE_V, bool PRE>
__global__ __launch_bounds__(kBlock, POSE_V ? 2 : 3) void k_fit_reduce(GridView g, RegistrationBuffers rb,
                                                        const PoseArg* __restrict__ pose,
                                                        const IekfCtrl* __restrict__ ctrl, int forced, int imu_en,
                                                        double plane_thr, double rinv, int nb_real, int epoch) {
  __shared__ ReduceShared sh;
  __shared__ NeedyShared sh_needy;
  LII_FB_TS(fb_ta);
  // The last kCompletionBlocks workgroups of the launch are COMPLETION workgroups (round 5): behind a search pass they finish the
  // queries that pass listed as unfinished - and fit, gate and sum them like any other point, into a column of partial sums of their
  // own - while the workgroups of the cloud leave those points out.  Round 4 had every workgroup finish its own flagged queries in
  // front of its fit: four or five dependent round trips that ONE workgroup in five went through and the whole launch waited for
  // (13.5 us per search-pass fit launch against 5.8 on cached planes).  Which workgroup takes which query, and in which order it adds
  // them up, depends on the queries' indices alone: the sums stay deterministic.
  // (they come FIRST in the grid - a launch with more workgroups than the chip holds at once must not start them last - and their
  // number is a multiple of eight: the workgroups of the cloud keep their XCDs)
  static_assert(kCompletionBlocks % 8 == 0 && kCompletionBlocksPre % 8 == 0, "XCD mapping of the cloud's workgroups");
  const int n_comp = completion_blocks(epoch);  // (uniform: as the host sized the grid)
  const bool completion_wg = (int)blockIdx.x < n_comp;
  const int cloud_block = (int)blockIdx.x - n_comp;
  // What a lane reads of its point whatever the pass turns out to be - body point, cached plane, selection flag, world point,
  // neighbour count - is requested BEFORE the flags, the pose and the cloud size have arrived (an unsharded cloud: the point's
  // index does not depend on them; the index is clamped, a lane beyond the cloud discards what it read): one dependent round
  // trip instead of two at the head of every fit launch.
  // (the flags, the size of the cloud and the pose - scalar requests - go out first, the point's data right behind them; all of it is
  // waited for once: k_knn_ck)
  const int q_early = fit_point_of(xcd_remap(completion_wg ? 0 : cloud_block, nb_real), nb_real);
  const bool early = rb.shard_world <= 1 && !completion_wg;
  const int ie = min(max(q_early, 0), rb.cap - 1);
  float4 e_body = make_float4(0.f, 0.f, 0.f, 0.f), e_world = e_body;
  double e_pl[4] = {0, 0, 0, 0};
  int e_count = 0;
  unsigned int e_sel = 0;
  float4 e_nb[5];
#pragma unroll
  for (int j = 0; j < 5; j++) e_nb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (early) {
    e_body = rb.body[ie];
    e_world = rb.world[ie];
    e_count = rb.nbr_count[ie];
    if (PRE) {
#pragma unroll
      for (int j = 0; j < 5; j++) e_nb[j] = rb.nbr[(size_t)j * rb.cap + ie];
    } else {
      const double* pl = rb.plane + 4 * (size_t)ie;
      e_pl[0] = pl[0]; e_pl[1] = pl[1]; e_pl[2] = pl[2]; e_pl[3] = pl[3];
      e_sel = rb.selected[ie];
    }
  }
  // (a completion workgroup: entry `lane` of the search pass's list - its length arrives with the scalars below, what lies behind the
  // end is read and dropped)
  const float4* flag_list = rb.flag_list + 2 * (epoch & 1) * kListCap;
  // (unconditional - the other workgroups read the list's first line and drop it: behind a branch the compiler joins the two paths
  // with a register copy, and the wait for the entry would stand in front of the scalar requests)
  static_assert(kFlagCap <= kBlock, "one lane per listed query");
  const float4* my_entry = flag_list + (completion_wg ? 2 * threadIdx.x : 0u);
  const float4 l0 = my_entry[0];
  const int l1x = __float_as_int(my_entry[1].x);
  const HeadScalars hs = load_head_scalars(pose, &ctrl->search_next, rb.n_dev ? rb.n_dev : &ctrl->max_it, epoch > 0 ? rb.flag_count + (epoch & 1) : &ctrl->max_it);
  asm volatile("" : "+v"(e_sel));  // (the selection flag is not looked at before this point: the compiler tests it where it is loaded, and the wait for it would stand in front of the scalar requests)
  PoseArg ps = hs.ps;
  if (POSE_V) ps = pose_to_vgprs(hs.ps);
  const int c_search = hs.search_next, c_stop = hs.stop, n_mem = hs.n_mem;
  int lo, n_live;
  shard_range_n(rb, n_mem, lo, n_live);
  bool FIT;
  if (forced >= 0) {
    FIT = forced != 0;
  } else {
    if (c_stop) return;
    FIT = c_search != 0;
  }
  // the unfinished queries of the search pass go to the completion workgroups when the pass could list them all
  const int n_flagged = epoch > 0 ? hs.extra : kFlagCap + 1;
  const bool defer = FIT && n_flagged <= kFlagCap;  // (uniform over the launch)
  int blk, i;
  bool live;
  float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
  bool skip_row = false;  // a flagged point in a workgroup of the cloud: the completion workgroups own it
  bool handed = false;    // a completion workgroup's lane whose query came through LDS (list, count, body point)
  if (!completion_wg) {
    blk = xcd_remap(cloud_block, nb_real);
    if (blk >= nb_real) return;  // uniform per block
    const int q = fit_point_of(blk, nb_real);
    i = lo + q;
    live = q < n_live;
    if (FIT) {  // uniform per workgroup
      const int count0 = live ? (early ? e_count : rb.nbr_count[i]) : 0;
      if (live) w4 = early ? e_world : rb.world[i];  // written by the search pass with the same arithmetic
      if (defer) {
        // (kNeedy: not completed yet; kDone: a completion workgroup of THIS launch has stored the finished list already - the
        // point is theirs either way, and its list may be half rewritten: never read here)
        skip_row = live && (count0 & (kNeedy | kDone)) != 0;
      } else {
        // (the far lists live where the launch's final reduction stages its rows later: one wavefront's row area holds kFarCap words)
        static_assert(sizeof(sh.row[0]) >= sizeof(unsigned int) * kFarCap, "far list");
        complete_flagged(g, rb, i, live, count0, w4, sh_needy, reinterpret_cast<unsigned int*>(&sh.row[0][0]));
      }
    }
  } else {
    // Completion workgroup j: the listed queries with (index / 4) % kCompletionBlocks == j (clusters of flagged queries - a stretch of
    // the sweep that looks past the edge of the map - are dealt out over the workgroups), in ascending order of their index: lane L
    // takes the L-th.  Not behind a search pass, or with the queries left to their own workgroups: nothing to do but the zero column.
    blk = nb_real + (int)blockIdx.x;
    i = 0;
    live = false;
    if (defer) {
      const int j = (int)blockIdx.x;
      LII_FB_TS(fb_tb);
      if (threadIdx.x == 0) sh_needy.n = 0;
      __syncthreads();
      // The entries of this workgroup, ranked by query index: query (aux), neighbour count with the flags (count), world point (w) and
      // body point - everything complete_one and the fit start from - go to LDS from the lane that read the entry at the head of the
      // launch: one dependent round trip (list + scalars) in front of the block probes, where the first form of this had three (the
      // list's length, the queries' indices, the entries).
      // Round 6: the listed queries are dealt out by their RANK in ascending order of the query index - rank r goes to workgroup
      // r % kCompletionBlocks, place r / kCompletionBlocks there - so that no workgroup holds more than ceil(n / 32) of them: 73 queries
      // are at most three per workgroup, one per wavefront, ONE round of completions.  (Round 5 dealt by (index / 4) % 32: clusters of
      // flagged queries left workgroups with six or eight - two rounds of ~12 us - beside idle ones; the order of the sums still
      // depends on the indices alone.)
      const int qx = __float_as_int(l0.w);
      const bool listed = (int)threadIdx.x < n_flagged;
      if (listed) sh_needy.key[threadIdx.x] = qx;
      __syncthreads();
      int rank_all = 0;
      for (int u = 0; u < n_flagged; u++) rank_all += sh_needy.key[u] < qx ? 1 : 0;  // (uniform trip count, broadcast reads)
      const bool mine = listed && (rank_all % n_comp) == j;
      const int m = n_flagged > j ? (n_flagged - j + n_comp - 1) / n_comp : 0;
      float4 my_body = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mine) my_body = rb.body[qx];
      if (threadIdx.x == 0) sh_needy.n = m;
      if (mine) {
        const int rank = rank_all / n_comp;
        sh_needy.aux[rank] = qx;
        sh_needy.count[rank] = l1x;
        sh_needy.w[rank][0] = l0.x; sh_needy.w[rank][1] = l0.y; sh_needy.w[rank][2] = l0.z;
        if (rank < kHandOver) sh_needy.body[rank] = my_body;
      }
      __syncthreads();
      const int wave = threadIdx.x >> 6;
      // (the four-wavefront form only in the launch that has the registers and runs behind a search: the other three instances of this
      // kernel keep their code and their register count - the 3-wavefronts-per-SIMD ones would spill 20 - 40 registers)
      if (POSE_V && PRE && m == 1) {  // (uniform) the workgroup's one query: its four wavefronts together
        complete_one_coop(g, rb, sh_needy.aux[0], sh_needy.count[0], sh_needy.w[0][0], sh_needy.w[0][1], sh_needy.w[0][2],
                          reinterpret_cast<unsigned int*>(&sh.row[0][0]), sh_needy.merge, &sh_needy.nb[0][0], &sh_needy.found[0], kDone);
      } else {
        for (int e = wave; e < m; e += kBlock / 64)  // one wavefront per query
          complete_one(g, rb, sh_needy.aux[e], sh_needy.count[e], sh_needy.w[e][0], sh_needy.w[e][1], sh_needy.w[e][2],
                       reinterpret_cast<unsigned int*>(&sh.row[0][0]) + wave * kFarCap, e < kHandOver ? &sh_needy.nb[e][0] : nullptr,
                       e < kHandOver ? &sh_needy.found[e] : nullptr, kDone);
      }
      __syncthreads();  // the completed lists are visible to the lanes that fit them (workgroup-scope release / acquire)
#ifdef LII_FALLBACK_TRACE
      if (threadIdx.x == 0 && m > 0) {
        const long long fb_tc = wall_clock64();
        atomicAdd(&g_fb_trace[16], 1ull); atomicAdd(&g_fb_trace[17], (unsigned long long)(fb_tb - fb_ta)); atomicAdd(&g_fb_trace[18], (unsigned long long)(fb_tc - fb_tb));
        atomicAdd(&g_fb_trace[20], (unsigned long long)m);
        sh_needy.key[kBlock - 1] = (int)(fb_tc & 0x7FFFFFFF);  // (low bits: the end stamp below subtracts them)
      }
#endif
      if ((int)threadIdx.x < m) {
        i = sh_needy.aux[threadIdx.x];
        live = true;
        handed = (int)threadIdx.x < kHandOver;
        w4 = make_float4(sh_needy.w[threadIdx.x][0], sh_needy.w[threadIdx.x][1], sh_needy.w[threadIdx.x][2], 0.f);
      }
    }
  }
  RowOut o;
#pragma unroll
  for (int c = 0; c < 12; c++) o.h[c] = 0;
  o.z = 0;
  o.sel = false;
  if (live && !skip_row) {
    float4 pb = e_body;
    if (!early) {  // (by value on every path: a choice between the three ADDRESSES sends e_body to scratch and waits for it at the head)
      if (handed) pb = sh_needy.body[threadIdx.x & (kHandOver - 1)];
      else pb = rb.body[i];
    }
    double bx = pb.x, by = pb.y, bz = pb.z;
    double ix = ps.RLI[0] * bx + ps.RLI[1] * by + ps.RLI[2] * bz + ps.TLI[0];
    double iy = ps.RLI[3] * bx + ps.RLI[4] * by + ps.RLI[5] * bz + ps.TLI[1];
    double iz = ps.RLI[6] * bx + ps.RLI[7] * by + ps.RLI[8] * bz + ps.TLI[2];
    float wx, wy, wz;
    double pa = 0, pbn = 0, pc = 0, pd = 0;
    bool candidate;
    if (FIT) {
      wx = w4.x; wy = w4.y; wz = w4.z;
      int found;
      float4 nb[5];
      if (handed) {
        found = sh_needy.found[threadIdx.x & (kHandOver - 1)];
#pragma unroll
        for (int j = 0; j < 5; j++) nb[j] = sh_needy.nb[threadIdx.x & (kHandOver - 1)][j];
      } else if (PRE && early && !(e_count & (kNeedy | kDone))) {  // (an unflagged query: nobody has touched its list since the search pass wrote it)
        found = e_count;
#pragma unroll
        for (int j = 0; j < 5; j++) nb[j] = e_nb[j];
      } else {
        found = rb.nbr_count[i] & kCountMask;
#pragma unroll
        for (int j = 0; j < 5; j++) nb[j] = rb.nbr[(size_t)j * rb.cap + i];
      }
      if (found == kMatch && canon_ties(nb)) {
#pragma unroll
        for (int j = 0; j < 5; j++) rb.nbr[(size_t)j * rb.cap + i] = nb[j];
      }
      candidate = (found == kMatch) && !(nb[4].w > 5.0f);
      if (candidate) candidate = fit_plane(nb, plane_thr, pa, pbn, pc, pd);
      double* pl = rb.plane + 4 * (size_t)i;
      pl[0] = pa; pl[1] = pbn; pl[2] = pc; pl[3] = pd;
    } else {
      wx = (float)(ps.R[0] * ix + ps.R[1] * iy + ps.R[2] * iz + ps.p[0]);
      wy = (float)(ps.R[3] * ix + ps.R[4] * iy + ps.R[5] * iz + ps.p[1]);
      wz = (float)(ps.R[6] * ix + ps.R[7] * iy + ps.R[8] * iz + ps.p[2]);
      rb.world[i] = make_float4(wx, wy, wz, 0.f);
      if (early && !PRE) {
        candidate = e_sel != 0;
        pa = e_pl[0]; pbn = e_pl[1]; pc = e_pl[2]; pd = e_pl[3];
      } else {
        candidate = rb.selected[i] != 0;
        const double* pl = rb.plane + 4 * (size_t)i;
        pa = pl[0]; pbn = pl[1]; pc = pl[2]; pd = pl[3];
      }
    }
    if (candidate) residual_row(ps, imu_en, bx, by, bz, ix, iy, iz, wx, wy, wz, pa, pbn, pc, pd, o);
    rb.selected[i] = o.sel ? 1 : 0;
  }
  block_reduce_rows(sh, o.h, o.z, o.sel, rinv, rb.partials + blk, rb.partial_stride);
#ifdef LII_FALLBACK_TRACE
  if (completion_wg && defer && threadIdx.x == 0 && sh_needy.n > 0)
    atomicAdd(&g_fb_trace[19], (unsigned long long)(((int)(wall_clock64() & 0x7FFFFFFF) - sh_needy.key[kBlock - 1]) & 0x7FFFFFFF));
#endif
}

// Deterministic final reduction of the transposed partials: out[t] = sum_b partials[t * stride + b].
// One 64-lane workgroup per output: coalesced loads, per-lane sums over b = lane + 64 k in a fixed order, then a
// fixed shuffle tree.  (91 independent workgroups: the partials were written by other XCDs, so every load is an
// L2 miss; one latency instead of a dependent chain of them.)
__global__ __launch_bounds__(256) void k_reduce91(const double* __restrict__ partials, int n_blocks, int stride,
                                                   double* __restrict__ out, const IekfCtrl* __restrict__ ctrl, int forced,
                                                   RegistrationBuffers rb) {
  __shared__ double s_w[4];
  if (forced < 0 && ctrl->stop) return;
  // (n_blocks = the workgroups of the fit launch: its points are dealt out in chunks, every workgroup may hold some)
  const int t = blockIdx.x;
  const double acc = final_sum_row<256>(partials + (size_t)t * stride, n_blocks, s_w);  // same order as k_reduce_solve
  if (threadIdx.x == 0) out[t] = acc;
}

// ------------------------------------------------------------------------------------------------
// LI-Init residual / Jacobian evaluators (include/LI_init/LI_init.h:91-205).  Records are 22 doubles:
// rot_end[9], ang_vel[3], linear_vel[3], ang_acc[3], linear_acc[3], timestamp.
// Tangent convention R <- Exp(delta) R  (Appendix B of SURVEY.md):
//   stage 1/2: r = R w_L - w_I [- (dT + t_d) a_I + b_g];  dr/ddelta = -[R w_L]x, dr/db_g = I, dr/dt_d = -alpha_I
//   stage 3:   r = R_LL0 R_LI^T a_I - R_LL0 b_a + R_GL0 g - a_L - R_LL0 ([w]x^2 + [alpha]x) T_IL
//              dr/ddelta_G = -[R_GL0 g]x, dr/db_a = -R_LL0, dr/dT_IL = -R_LL0 ([w]x^2 + [alpha]x)
// Output per block: dof*dof + dof + 1 doubles (J^T J row-major, J^T r, 0.5 sum r^2), reduced on one block.

template <int STAGE>
__global__ __launch_bounds__(256) void k_calib_eval(const double* __restrict__ imu, const double* __restrict__ lidar, int n,
                                                    const double* __restrict__ params, double* __restrict__ out) {
  constexpr int DOF = STAGE == 1 ? 3 : (STAGE == 2 ? 7 : 9);
  constexpr int NOUT = DOF * DOF + DOF + 1;
  __shared__ double sh[256];
  double acc[NOUT];
#pragma unroll
  for (int k = 0; k < NOUT; k++) acc[k] = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double* I = imu + 22 * (size_t)i;
    const double* L = lidar + 22 * (size_t)i;
    double r[3];
    double J[3][DOF];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < DOF; b++) J[a][b] = 0;
    if (STAGE == 1 || STAGE == 2) {
      const double* R = params;
      double Rw[3];
      mat3_vec(R, L + 9, Rw);
#pragma unroll
      for (int a = 0; a < 3; a++) r[a] = Rw[a] - I[9 + a];
      // -[Rw]x
      J[0][1] = Rw[2]; J[0][2] = -Rw[1];
      J[1][0] = -Rw[2]; J[1][2] = Rw[0];
      J[2][0] = Rw[1]; J[2][1] = -Rw[0];
      if (STAGE == 2) {
        const double* bg = params + 9;
        double td = params[12];
        double dT = L[21] - I[21];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          r[a] = r[a] - (dT + td) * I[15 + a] + bg[a];
          J[a][(3 + a) % DOF] = 1.0;
          J[a][6 % DOF] = -I[15 + a];
        }
      }
    } else {
      const double* RG = params;        // R_GL0
      const double* ba = params + 9;    // bias_aL
      const double* Til = params + 12;  // T_IL
      const double* RLI = params + 15;  // R_LI (fixed)
      const double* RLL0 = L;           // rot_end
      const double g[3] = {0, 0, -9.81};  // STD_GRAV (LI_init.h:27)
      double aI_L[3], t1[3], Rg[3];
      mat3t_vec(RLI, I + 18, aI_L);  // R_LI^T a_I
      mat3_vec(RLL0, aI_L, t1);      // R_LL0 R_LI^T a_I
      mat3_vec(RG, g, Rg);
      const double* w = L + 9;
      const double* al = L + 15;
      double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
      double A[9] = {0, -al[2], al[1], al[2], 0, -al[0], -al[1], al[0], 0};
      double M[9], RM[9];
      mat3_mul(W, W, M);
#pragma unroll
      for (int e = 0; e < 9; e++) M[e] += A[e];
      mat3_mul(RLL0, M, RM);
      double Rb[3], RMt[3];
      mat3_vec(RLL0, ba, Rb);
      mat3_vec(RM, Til, RMt);
#pragma unroll
      for (int a = 0; a < 3; a++) r[a] = t1[a] - Rb[a] + Rg[a] - L[18 + a] - RMt[a];
      J[0][1] = Rg[2]; J[0][2] = -Rg[1];
      J[1][0] = -Rg[2]; J[1][2] = Rg[0];
      J[2][0] = Rg[1]; J[2][1] = -Rg[0];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
          J[a][(3 + b) % DOF] = -RLL0[3 * a + b];
          J[a][(6 + b) % DOF] = -RM[3 * a + b];
        }
    }
#pragma unroll
    for (int a = 0; a < DOF; a++) {
#pragma unroll
      for (int b = 0; b < DOF; b++) acc[a * DOF + b] += J[0][a] * J[0][b] + J[1][a] * J[1][b] + J[2][a] * J[2][b];
      acc[DOF * DOF + a] += J[0][a] * r[0] + J[1][a] * r[1] + J[2][a] * r[2];
    }
    acc[DOF * DOF + DOF] += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  }
  // fixed-order block reduction (tree over 256 threads), one output at a time
#pragma unroll
  for (int k = 0; k < NOUT; k++) {
    sh[threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = sh[0];
    __syncthreads();
  }
}
template __global__ void k_calib_eval<1>(const double*, const double*, int, const double*, double*);
template __global__ void k_calib_eval<2>(const double*, const double*, int, const double*, double*);
template __global__ void k_calib_eval<3>(const double*, const double*, int, const double*, double*);

}  // namespace lii

// ------------------------------------------------------------------------------------------------
// launchers
#include "lii_launch.h"
namespace lii {
static inline int nblk(int n, int b) { return (n + b - 1) / b; }

void launch_map_keys(const float4* pts, int n, float inv_cs, unsigned long long* keys, unsigned int* idx, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_map_keys, dim3(nblk(n, 256)), dim3(256), 0, s, pts, n, inv_cs, keys, idx);
}
void launch_map_gather(const float4* src, const unsigned int* idx, int n, float4* dst, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_map_gather, dim3(nblk(n, 256)), dim3(256), 0, s, src, idx, n, dst);
}
void launch_block_flags(const unsigned long long* keys, int n, unsigned int* flags, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_block_flags, dim3(nblk(n, 256)), dim3(256), 0, s, keys, n, flags);
}
// ---- dense cell window (GridView::win) ---------------------------------------------------------------------------------------------
// box[0..2] = min, box[3..5] = max of the BIASED block coordinates over the occupied slots of the block table
__global__ void k_win_bbox(const BlockEntry* __restrict__ blocks, unsigned int cap, unsigned int* __restrict__ box) {
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  const unsigned long long k = blocks[i].key;
  if (k == kEmptyKey) return;
  const unsigned int bx = (unsigned)(k & 0x3FFFF), by = (unsigned)((k >> 18) & 0x3FFFF), bz = (unsigned)((k >> 36) & 0x3FFFF);
  atomicMin(&box[0], bx); atomicMin(&box[1], by); atomicMin(&box[2], bz);
  atomicMax(&box[3], bx); atomicMax(&box[4], by); atomicMax(&box[5], bz);
}
// one workgroup of 512 lanes per slot of the block table: the 512 cell entries of an occupied block go to their places in the window
__global__ __launch_bounds__(512) void k_win_fill(const BlockEntry* __restrict__ blocks, const uint2* __restrict__ cells, uint2* __restrict__ win,
                                                  int wx0, int wy0, int wz0, int wnx, int wny, int wnz) {
  const BlockEntry e = blocks[blockIdx.x];
  if (e.key == kEmptyKey) return;
  const int bb = kCellBias >> kCoarseShift;
  const int bx = (int)(e.key & 0x3FFFF) - bb, by = (int)((e.key >> 18) & 0x3FFFF) - bb, bz = (int)((e.key >> 36) & 0x3FFFF) - bb;
  const int l = threadIdx.x;
  const unsigned int ux = (unsigned)(bx * 8 + (l & 7) - wx0), uy = (unsigned)(by * 8 + ((l >> 3) & 7) - wy0), uz = (unsigned)(bz * 8 + (l >> 6) - wz0);
  if (ux < (unsigned)wnx && uy < (unsigned)wny && uz < (unsigned)wnz) win[((size_t)uz * wny + uy) * wnx + ux] = cells[(size_t)e.id * kBlockCells + l];
}
void launch_win_bbox(const BlockEntry* blocks, unsigned int cap, unsigned int* box, hipStream_t s) {
  hipLaunchKernelGGL(k_win_bbox, dim3((cap + 255) / 256), dim3(256), 0, s, blocks, cap, box);
}
void launch_win_fill(const BlockEntry* blocks, unsigned int cap, const uint2* cells, uint2* win, const int org[3], const int dim[3], hipStream_t s) {
  hipLaunchKernelGGL(k_win_fill, dim3(cap), dim3(512), 0, s, blocks, cells, win, org[0], org[1], org[2], dim[0], dim[1], dim[2]);
}
void launch_table_clear(BlockEntry* blocks, unsigned int cap, hipStream_t s) {
  hipLaunchKernelGGL(k_table_clear, dim3(nblk((int)cap, 256)), dim3(256), 0, s, blocks, cap);
}
void launch_cells_fill(const unsigned long long* keys, const unsigned int* ranks, int n, BlockEntry* blocks,
                       unsigned int block_mask, uint2* cells, unsigned long long* key_of_id, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_cells_fill, dim3(nblk(n, 256)), dim3(256), 0, s, keys, ranks, n, blocks, block_mask, cells, key_of_id);
}
int register_blocks(int n) { return nblk(n, kBlock); }
#ifdef LII_FALLBACK_TRACE
void fb_trace_read(unsigned long long out[32]) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fb_trace), sizeof(unsigned long long) * 32); }
#endif
// upper bound of the points one rank registers (the exact split is taken on the device from the exact cloud size)
static inline int shard_bound(const RegistrationBuffers& rb) {
  return rb.shard_world > 1 ? (rb.n + rb.shard_world - 1) / rb.shard_world + 1 : rb.n;
}
// The search launch: k_knn_ck, four lanes per query.
// 128 lanes per workgroup, 6 loads in flight per lane, 7 wavefronts per SIMD (69 VGPRs): measured on the per-lane form of rounds 3 - 4
// against 64 / 256 lanes, 4 .. 12 loads, 6 / 8 wavefronts per SIMD (within 1 - 2 %: profiles/r03_knn_ab.md), and 2 / 1 lanes per query
// (slower at every cloud size: profiles/r05_knn_lpq.md).
#ifndef LII_KNN_BS  // (experiment builds: tools/ab_build.sh ... -DLII_KNN_BS=256 -DLII_KNN_NB=4 -DLII_KNN_WPE=8)
#define LII_KNN_BS 128
#endif
#ifndef LII_KNN_NB
#define LII_KNN_NB 6
#endif
#ifndef LII_KNN_WPE
#define LII_KNN_WPE 7
#endif
// epoch: the number of this search launch (> 0; the fit launch behind it gets the same) - or 0: no list of unfinished queries, every
// workgroup of the fit launch finishes its own (hipGraph replays, whose arguments are frozen)
void launch_knn(const GridView& g, const RegistrationBuffers& rb, const PoseArg* pose,
                const IekfCtrl* ctrl, int forced, double* search_pose_out, hipStream_t s, int epoch, hipEvent_t ev_start, hipEvent_t ev_stop) {
  int nq = nblk(shard_bound(rb), LII_KNN_BS / 4);
  if (nq < 1) nq = 1;
  const int nq_pad = ((nq + 7) / 8) * 8;
  if (ev_start && ev_stop) {  // (measurement: the dispatch's own time stamps, no barrier packets around it)
    hipExtLaunchKernelGGL((k_knn_ck<4, LII_KNN_BS, LII_KNN_NB, LII_KNN_WPE>), dim3(nq_pad), dim3(LII_KNN_BS), 0, s, ev_start, ev_stop, 0u, g, rb, pose, ctrl, forced, nq,
                          search_pose_out, epoch);
    return;
  }
  hipLaunchKernelGGL((k_knn_ck<4, LII_KNN_BS, LII_KNN_NB, LII_KNN_WPE>), dim3(nq_pad), dim3(LII_KNN_BS), 0, s, g, rb, pose, ctrl, forced, nq, search_pose_out, epoch);
}
void launch_knn_complete(const GridView& g, const RegistrationBuffers& rb, hipStream_t s) {
  int nb = nblk(shard_bound(rb), kBlock);
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_knn_complete, dim3(nb), dim3(kBlock), 0, s, g, rb);
}
void launch_complete_listed(const GridView& g, const RegistrationBuffers& rb, const IekfCtrl* ctrl, int forced, hipStream_t s, int epoch) {
  if (epoch <= 0) return;
  hipLaunchKernelGGL(k_complete_listed, dim3(kListCap / (kBlock / 64)), dim3(kBlock), 0, s, g, rb, ctrl, forced, epoch);
}
void launch_fit_reduce(const GridView& g, const RegistrationBuffers& rb, const PoseArg* pose,
                       const IekfCtrl* ctrl, int forced, int imu_en, double plane_thr, double rinv, hipStream_t s, int epoch) {
  int nb = nblk(shard_bound(rb), kBlock);
  if (nb < 1) nb = 1;
  const int nb_pad = ((nb + 7) / 8) * 8;
  // (four wavefronts per workgroup on 1024 SIMDs: up to 512 workgroups are two wavefronts per SIMD at most)
  // (+ the completion workgroups behind the workgroups of the cloud)
  const dim3 grid(nb_pad + completion_blocks(epoch)), block(kBlock);
  const bool pre = epoch > 0;  // behind a search launch that lists its unfinished queries: the lanes request their neighbour lists at the head
  if (nb <= 512) {
    if (pre) hipLaunchKernelGGL((k_fit_reduce<true, true>), grid, block, 0, s, g, rb, pose, ctrl, forced, imu_en, plane_thr, rinv, nb, epoch);
    else hipLaunchKernelGGL((k_fit_reduce<true, false>), grid, block, 0, s, g, rb, pose, ctrl, forced, imu_en, plane_thr, rinv, nb, epoch);
  } else {
    if (pre) hipLaunchKernelGGL((k_fit_reduce<false, true>), grid, block, 0, s, g, rb, pose, ctrl, forced, imu_en, plane_thr, rinv, nb, epoch);
    else hipLaunchKernelGGL((k_fit_reduce<false, false>), grid, block, 0, s, g, rb, pose, ctrl, forced, imu_en, plane_thr, rinv, nb, epoch);
  }
}
void launch_reduce91(const RegistrationBuffers& rb, double* out91, const IekfCtrl* ctrl, int forced, hipStream_t s, int epoch) {
  int nb = nblk(shard_bound(rb), kBlock);
  if (nb < 1) nb = 1;
  nb += completion_blocks(epoch);  // (the columns of the fit launch's completion workgroups: as many as THAT launch had)
  hipLaunchKernelGGL(k_reduce91, dim3(kNormalEq), dim3(256), 0, s, rb.partials, nb, rb.partial_stride, out91, ctrl, forced, rb);
}
void launch_calib_eval(int stage, const double* imu, const double* lidar, int n, const double* params, double* out,
                       hipStream_t s) {
  if (stage == 1) hipLaunchKernelGGL(k_calib_eval<1>, dim3(1), dim3(256), 0, s, imu, lidar, n, params, out);
  else if (stage == 2) hipLaunchKernelGGL(k_calib_eval<2>, dim3(1), dim3(256), 0, s, imu, lidar, n, params, out);
  else hipLaunchKernelGGL(k_calib_eval<3>, dim3(1), dim3(256), 0, s, imu, lidar, n, params, out);
}
}  // namespace lii
