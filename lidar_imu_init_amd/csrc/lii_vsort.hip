// Back half of the voxel-grid filter: sort the n (voxel key, point index) pairs ascending by key then by index — what a
// stable sort of the keys gives (PCL sorts cloud_point_index_idx by voxel index, filters/voxel_grid.hpp; the oracle restates
// it stably) — then one centroid per run of equal keys, written in key order.
//
// The library path it replaces (merge sort + flag kernel + scan + centroid kernel) is 14 dependent launches at 100 k pairs,
// each paying a launch and a search / look-back latency: ~80 us for 0.8 MB of data.  Here the dependent chain is five short
// kernels:
//   1. splitters  S = 2B samples (one per stratum of the input, jittered; drawn by k_voxel_keys while it writes the keys)
//                 are ranked all-pairs across the whole chip; every 2nd becomes the lower bound of a bucket.  The sort key
//                 is the 64-bit composite (key << 32 | index): composites are distinct, so equal voxel keys cannot overload
//                 a bucket and no step needs to be stable.
//   2. classify   a workgroup bins its tile of 1024 pairs into the B ~ n/64 buckets (branch-free binary search over the
//                 splitters in LDS) and adds its counts to the bucket totals.
//   3. scatter    every workgroup scans the totals into bucket offsets for itself, reserves its share of each bucket with
//                 one global atomic and hands the slots out with LDS atomics (the order inside a bucket is arbitrary).
//   4. local      a workgroup per group of 4 consecutive buckets: every composite is ranked all-pairs inside ITS bucket
//                 (~50 pairs, one lane per composite); the rank is the output slot.  The sorted group is still in LDS, so
//                 the voxel starts of the group are counted here as well.
//   5. centroids  a workgroup per group again: the number of voxel starts in the groups before it (a sum over <= 512
//                 counts) is its output offset; one lane per voxel start accumulates the run in index order.
// Barrier-synchronised LDS sorts were measured for steps 1 and 4 (~300 ns per bitonic stage, 13-23 us per kernel) and
// dropped; so was a ticketed "last workgroup scans" tail in step 2 (three more dependent round trips).
// Groups larger than the LDS tile (adversarial inputs only) are ranked out of global memory — slow and correct.
#include <hip/hip_runtime.h>

#include "lii_device.h"
#include "lii_launch.h"

namespace lii {

namespace {
using u64 = unsigned long long;
constexpr int kTile = 1024;        // pairs per classify / scatter workgroup (256 threads x 4)
constexpr int kMaxBuckets = 2048;
constexpr int kPerLane = kMaxBuckets / 256;
constexpr int kGroup = 4;          // consecutive buckets per local / centroid workgroup (1 once the buckets are large)
constexpr int kLocalCap = 2048;    // composites a group keeps in LDS
// Sort keys are kVoxKeyBits wide (room for an order key of up to 34 bits: a Morton order over the voxels was measured in round 2), the point index takes the low
// kVoxIdxBits of the 64-bit composite.  kVoxDropKey = non-finite points: sorted last, start no voxel.
constexpr u64 kDropKey = kVoxDropKey;

__device__ __forceinline__ u64 composite(u64 key, unsigned int i) { return (key << kVoxIdxBits) | (u64)i; }
__device__ __forceinline__ u64 comp_key(u64 c) { return c >> kVoxIdxBits; }
__device__ __forceinline__ unsigned int comp_idx(u64 c) { return (unsigned int)(c & ((1ull << kVoxIdxBits) - 1ull)); }

// #{k in [k0, k1) : e[k] < x}; k0, k1 multiples of 8 (arrays are padded with ~0, which is never smaller).  Eight broadcast
// reads are issued before the compares so the LDS latency is paid once per eight composites, not per composite.
__device__ __forceinline__ int count_less(const u64* e, int k0, int k1, u64 x) {
  int cnt = 0;
  for (int k = k0; k < k1; k += 8) {
    const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(e + k);
    const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(e + k + 2);
    const ulonglong2 c = *reinterpret_cast<const ulonglong2*>(e + k + 4);
    const ulonglong2 d = *reinterpret_cast<const ulonglong2*>(e + k + 6);
    cnt += (a.x < x) + (a.y < x) + (b.x < x) + (b.y < x) + (c.x < x) + (c.y < x) + (d.x < x) + (d.y < x);
  }
  return cnt;
}

// Exclusive scan over the 256 lanes of a workgroup (wave scan + four wave totals through LDS).  Returns the exclusive
// prefix of `v`; *total receives the sum over the workgroup.
__device__ __forceinline__ unsigned int block_exclusive_scan(unsigned int v, unsigned int* wtot /*[4] LDS*/, unsigned int* total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned int inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  __syncthreads();  // wtot may still be read from a previous call
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  unsigned int before = 0;
  for (int w = 0; w < wave; w++) before += wtot[w];
  *total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  return before + inc - v;
}

// ---------------------------------------------------------------------------------------------------- 1. splitters
// A workgroup ranks 8 samples, each against 32 segments of S / 32 composites (one lane per (sample, segment)).  Samples
// arrive compact (samples[j] for stratum j < n_strata; the rest count as +inf).  The sample of rank r > 0, r % per == 0, is
// the lower bound of bucket r / per.
__global__ __launch_bounds__(256) void k_vsort_splitters(const u64* __restrict__ samples, int n_strata, int S, int B,
                                                          u64* __restrict__ splitters) {
  extern __shared__ __align__(16) u64 smp[];
  __shared__ int part[32];
  const int tid = threadIdx.x;
  {  // S <= 4096: eight 16-byte loads per lane, all in flight before the first LDS write
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(samples);
    ulonglong2 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int p2 = tid + k * 256;
      v[k] = p2 * 2 < S ? src[p2] : make_ulonglong2(~0ull, ~0ull);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int p2 = tid + k * 256;
      if (p2 * 2 < S) {
        if (p2 * 2 >= n_strata) v[k].x = ~0ull;
        if (p2 * 2 + 1 >= n_strata) v[k].y = ~0ull;
        *reinterpret_cast<ulonglong2*>(smp + p2 * 2) = v[k];
      }
    }
  }
  __syncthreads();
  const int sj = blockIdx.x * 8 + (tid & 7), seg = tid >> 3;  // S >= 256: 32 segments of S / 32 >= 8 composites
  const int len = S >> 5;
  const u64 e = smp[sj];
  int cnt = count_less(smp, seg * len, (seg + 1) * len, e);
  cnt += __shfl_xor(cnt, 8);  // the 8 segments of a wavefront that share the sample
  cnt += __shfl_xor(cnt, 16);
  cnt += __shfl_xor(cnt, 32);
  if ((tid & 63) < 8) part[(tid >> 6) * 8 + (tid & 7)] = cnt;
  __syncthreads();
  if (tid < 8 && sj < n_strata) {
    const int rank = part[tid] + part[8 + tid] + part[16 + tid] + part[24 + tid];
    const int per = S / B;
    if (rank > 0 && rank % per == 0) splitters[rank / per - 1] = e;
  }
}
// (S < 256 only for n < ~8 k: one workgroup, every lane ranks one sample against all)
__global__ __launch_bounds__(256) void k_vsort_splitters_small(const u64* __restrict__ samples, int n_strata, int S, int B,
                                                                u64* __restrict__ splitters) {
  __shared__ __align__(16) u64 smp[256];
  const int tid = threadIdx.x;
  smp[tid] = tid < n_strata && tid < S ? samples[tid] : ~0ull;
  __syncthreads();
  if (tid >= n_strata || tid >= S) return;
  const u64 e = smp[tid];
  const int rank = count_less(smp, 0, (S + 7) & ~7, e), per = S / B;
  if (rank > 0 && rank % per == 0) splitters[rank / per - 1] = e;
}

// ---------------------------------------------------------------------------------------------------- 2. classify
// tot[b] is zero on entry (k_vsort_local of the previous sort clears it) and holds the bucket sizes on exit.
__global__ __launch_bounds__(256) void k_vsort_classify(const u64* __restrict__ keys, int n, int B,
                                                         const u64* __restrict__ splitters, int n_splitters,
                                                         unsigned short* __restrict__ bucket_of, unsigned int* tot) {
  __shared__ u64 spl[kMaxBuckets];
  __shared__ unsigned int cnt[kMaxBuckets];
  const int tid = threadIdx.x;
  constexpr int E = kTile / 256;
  const int base = blockIdx.x * kTile;
  u64 c[E];
  int pos[E];
#pragma unroll
  for (int k = 0; k < E; k++) {  // the key loads go out together with the splitter loads below
    const int i = base + k * 256 + tid;
    c[k] = composite(keys[i < n ? i : n - 1], (unsigned int)i);
    pos[k] = 0;
  }
  {
    u64 v[kPerLane];
#pragma unroll
    for (int k = 0; k < kPerLane; k++) {
      const int b = tid + k * 256;
      v[k] = b < n_splitters ? splitters[b] : ~0ull;  // buckets past the last real sample stay empty
    }
#pragma unroll
    for (int k = 0; k < kPerLane; k++) {
      const int b = tid + k * 256;
      if (b < B) {
        spl[b] = v[k];
        cnt[b] = 0;
      }
    }
  }
  __syncthreads();
  for (int step = B >> 1; step > 0; step >>= 1) {  // the four searches side by side: one LDS latency per step
#pragma unroll
    for (int k = 0; k < E; k++) pos[k] += c[k] >= spl[pos[k] + step - 1] ? step : 0;
  }
#pragma unroll
  for (int k = 0; k < E; k++) {
    const int i = base + k * 256 + tid;
    if (i < n) {
      bucket_of[i] = (unsigned short)pos[k];
      atomicAdd(&cnt[pos[k]], 1u);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPerLane; k++) {
    const int b = tid + k * 256;
    if (b < B && cnt[b]) atomicAdd(&tot[b], cnt[b]);
  }
}

// ---------------------------------------------------------------------------------------------------- 3. scatter
// Every workgroup turns the B totals into offsets itself (8 KB of loads and a block scan: cheaper than a kernel or a ticketed
// tail doing it once); workgroup 0 leaves them in bucket_start for the kernels that follow.  cursor[b] is zero on entry.
__global__ __launch_bounds__(256) void k_vsort_scatter(const u64* __restrict__ keys, int n, int B,
                                                        const unsigned short* __restrict__ bucket_of,
                                                        const unsigned int* __restrict__ tot, unsigned int* cursor,
                                                        unsigned int* __restrict__ bucket_start, u64* __restrict__ comp) {
  __shared__ unsigned int start[kMaxBuckets], off[kMaxBuckets], wtot[4];
  const int tid = threadIdx.x;
  constexpr int E = kTile / 256;
  const int base = blockIdx.x * kTile;
  unsigned short mine[E];
  u64 key[E];
#pragma unroll
  for (int k = 0; k < E; k++) {  // everything this workgroup reads from global memory goes out in one round
    const int i = base + k * 256 + tid;
    mine[k] = i < n ? bucket_of[i] : (unsigned short)0xFFFF;
    key[k] = i < n ? keys[i] : 0ull;
  }
  const int per = B >= 256 ? B >> 8 : 1;  // consecutive buckets per lane in the scan
  unsigned int v[kPerLane], loc = 0;
#pragma unroll
  for (int k = 0; k < kPerLane; k++) {
    const int b = tid * per + k;
    v[k] = (k < per && b < B) ? tot[b] : 0u;
    loc += v[k];
  }
#pragma unroll
  for (int k = 0; k < kPerLane; k++) {
    const int b = tid + k * 256;
    if (b < B) off[b] = 0;
  }
  unsigned int total;
  unsigned int run = block_exclusive_scan(loc, wtot, &total);
#pragma unroll
  for (int k = 0; k < kPerLane; k++) {
    const int b = tid * per + k;
    if (k < per && b < B) {
      start[b] = run;
      if (blockIdx.x == 0) bucket_start[b] = run;
      run += v[k];
    }
  }
  if (blockIdx.x == 0 && tid == 0) bucket_start[B] = total;  // == n
  __syncthreads();
#pragma unroll
  for (int k = 0; k < E; k++)
    if (mine[k] != 0xFFFF) atomicAdd(&off[mine[k]], 1u);
  __syncthreads();
  {
    unsigned int c[kPerLane], r[kPerLane];
#pragma unroll
    for (int k = 0; k < kPerLane; k++) {  // the returning atomics of a lane overlap instead of queueing up
      const int b = tid + k * 256;
      c[k] = b < B ? off[b] : 0u;
      r[k] = c[k] ? atomicAdd(&cursor[b], c[k]) : 0u;
    }
#pragma unroll
    for (int k = 0; k < kPerLane; k++) {
      const int b = tid + k * 256;
      if (b < B) off[b] = start[b] + r[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < E; k++) {
    const int i = base + k * 256 + tid;
    if (i < n) comp[atomicAdd(&off[mine[k]], 1u)] = composite(key[k], (unsigned int)i);
  }
}

// ---------------------------------------------------------------------------------------------------- 4. local
// The group's range of `comp` goes to LDS; a composite is ranked against its own bucket only — all-pairs inside a window,
// no barrier in the loop.  The window is widened to multiples of 8 for the batched LDS reads: what it picks up below the
// bucket belongs to earlier buckets (smaller: counted, then subtracted), above it to later ones or the padding (larger).
// The ranked composites are parked in a second LDS array: coalesced stores, and the voxel starts of the group can be
// counted on the spot (group_count).  The first pair of a group needs the key before it: the maximum over the nearest
// non-empty bucket below.  Clears tot / cursor of its buckets for the next sort.
template <int kGroup>
__global__ __launch_bounds__(256) void k_vsort_local(const u64* __restrict__ comp, const unsigned int* __restrict__ bucket_start,
                                                      int B, unsigned int* tot, unsigned int* cursor,
                                                      u64* __restrict__ keys_out, unsigned int* __restrict__ idx_out,
                                                      unsigned int* __restrict__ group_count) {
  __shared__ __align__(16) u64 e[kLocalCap + 8];
  __shared__ __align__(16) u64 srt[kLocalCap];
  __shared__ unsigned int bs[kGroup + 2];  // [0] start of the bucket below the group, [1..kGroup+1] the group's boundaries
  __shared__ u64 s_prev;
  __shared__ unsigned int s_count;
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * kGroup;
  if (tid <= kGroup + 1) {
    const int b = b0 - 1 + tid;
    bs[tid] = bucket_start[b < 0 ? 0 : (b > B ? B : b)];
  }
  if (tid < kGroup && b0 + tid < B) {
    tot[b0 + tid] = 0u;
    cursor[b0 + tid] = 0u;
  }
  if (tid == 0) { s_prev = 0ull; s_count = 0u; }
  __syncthreads();
  const unsigned int s0 = bs[1];
  const int M = (int)(bs[kGroup + 1] - s0);
  if (M <= 0) {
    if (tid == 0) group_count[blockIdx.x] = 0u;
    return;
  }
  const bool have_prev = s0 > 0;  // the key of the pair just below the group: max over the nearest non-empty bucket below
  if (have_prev) {
    unsigned int p0 = bs[0];
    if (p0 == s0) {  // the bucket right below is empty (rare): walk down, one lane, dependent loads
      if (tid == 0) {
        int b = b0 - 1;
        unsigned int st = s0;
        while (b > 0 && st == s0) st = bucket_start[--b];
        bs[0] = st;
      }
      __syncthreads();
      p0 = bs[0];
    }
    u64 mx = 0;
    for (unsigned int k = p0 + tid; k < s0; k += 256) mx = max(mx, comp_key(comp[k]));
    if (mx) atomicMax(&s_prev, mx);
  }
  const bool in_lds = M <= kLocalCap;
  if (in_lds) {
    const int M8 = (M + 7) & ~7;
    for (int k = tid; k < M8; k += 256) e[k] = k < M ? comp[s0 + k] : ~0ull;
  }
  __syncthreads();
  for (int t = tid; t < M; t += 256) {
    int g = 0;
#pragma unroll
    for (int q = 2; q <= kGroup; q++) g += (unsigned int)t + s0 >= bs[q] ? 1 : 0;
    const int lo = (int)(bs[g + 1] - s0), hi = (int)(bs[g + 2] - s0);
    if (in_lds) {
      const u64 x = e[t];
      const int lo8 = lo & ~7;
      srt[lo + count_less(e, lo8, (hi + 7) & ~7, x) - (lo - lo8)] = x;
    } else {  // oversized group (adversarial input): the same ranking out of global memory, straight to the output
      const u64 x = comp[s0 + t];
      int cnt = 0;
      for (int k = lo; k < hi; k++) cnt += comp[s0 + k] < x ? 1 : 0;
      keys_out[s0 + lo + cnt] = comp_key(x);
      idx_out[s0 + lo + cnt] = comp_idx(x);
    }
  }
  __threadfence_block();
  __syncthreads();
  unsigned int starts = 0;
  for (int k = tid; k < M; k += 256) {
    u64 key, before;
    if (in_lds) {
      const u64 x = srt[k];
      key = comp_key(x);
      keys_out[s0 + k] = key;
      idx_out[s0 + k] = comp_idx(x);
      before = k > 0 ? comp_key(srt[k - 1]) : s_prev;
    } else {
      key = keys_out[s0 + k];
      before = k > 0 ? keys_out[s0 + k - 1] : s_prev;
    }
    starts += (key != kDropKey && ((k == 0 && !have_prev) || key != before)) ? 1u : 0u;
  }
  for (int off = 32; off > 0; off >>= 1) starts += __shfl_down(starts, off);
  if ((tid & 63) == 0 && starts) atomicAdd(&s_count, starts);
  __syncthreads();
  if (tid == 0) group_count[blockIdx.x] = s_count;
}

// n <= 512: one workgroup ranks the pairs straight from the keys and plays a single group
__global__ __launch_bounds__(512) void k_vsort_small(const u64* __restrict__ keys, int n, u64* __restrict__ keys_out,
                                                      unsigned int* __restrict__ idx_out, unsigned int* __restrict__ bucket_start,
                                                      unsigned int* __restrict__ group_count) {
  __shared__ __align__(16) u64 e[512];
  __shared__ u64 srt[512];
  __shared__ unsigned int s_count;
  const int tid = threadIdx.x;
  e[tid] = tid < n ? composite(keys[tid], (unsigned int)tid) : ~0ull;
  if (tid == 0) s_count = 0u;
  if (tid <= kGroup) bucket_start[tid] = tid == 0 ? 0u : (unsigned int)n;
  __syncthreads();
  if (tid < n) srt[count_less(e, 0, (n + 7) & ~7, e[tid])] = e[tid];
  __syncthreads();
  unsigned int st = 0;
  if (tid < n) {
    const u64 key = comp_key(srt[tid]);
    keys_out[tid] = key;
    idx_out[tid] = comp_idx(srt[tid]);
    st = (key != kDropKey && (tid == 0 || key != comp_key(srt[tid - 1]))) ? 1u : 0u;
  }
  for (int off = 32; off > 0; off >>= 1) st += __shfl_down(st, off);
  if ((tid & 63) == 0 && st) atomicAdd(&s_count, st);
  __syncthreads();
  if (tid == 0) group_count[0] = s_count;
}

// ---------------------------------------------------------------------------------------------------- 5. centroids
// One lane per voxel start accumulates its run sequentially in index order, float32, then divides by the count — the
// order the oracle uses.  A run may continue past the end of the group (a voxel split by a splitter): the lane just keeps
// reading the sorted arrays.
template <int kGroup>
__global__ __launch_bounds__(256) void k_voxel_centroids(const float4* __restrict__ pts, const u64* __restrict__ keys,
                                                          const unsigned int* __restrict__ idx,
                                                          const unsigned int* __restrict__ bucket_start, int B, int n,
                                                          const unsigned int* __restrict__ group_count, float4* __restrict__ out,
                                                          int* __restrict__ n_out, const unsigned int* __restrict__ pcl_in,
                                                          unsigned int* __restrict__ pcl_out, unsigned int* __restrict__ max_run) {
  __shared__ unsigned int wtot[4], s_sum[4];
  const int tid = threadIdx.x, g = blockIdx.x;
  const int b0 = g * kGroup, b1 = min(b0 + kGroup, B);
  const unsigned int s0 = bucket_start[b0], s1 = bucket_start[b1];
  unsigned int before = 0;  // voxel starts in the groups below this one
  for (int q = tid; q < g; q += 256) before += group_count[q];
  for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
  if ((tid & 63) == 0) s_sum[tid >> 6] = before;
  __syncthreads();
  unsigned int slot0 = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
  if (g == (int)gridDim.x - 1 && tid == 0) *n_out = (int)(slot0 + group_count[g]);  // size of the down-sampled cloud
  for (unsigned int base = s0; base < s1; base += 256) {
    const unsigned int i = base + tid;
    u64 k = 0, k_next = kDropKey;
    unsigned int flag = 0, id0 = 0;
    if (i < s1) {  // everything the common case (a voxel of one point) needs, in one round of loads
      k = keys[i];
      const u64 k_prev = i > 0 ? keys[i - 1] : kDropKey;
      if (i + 1 < (unsigned int)n) k_next = keys[i + 1];
      id0 = idx[i];
      flag = (k != kDropKey && (i == 0 || k_prev != k)) ? 1u : 0u;
    }
    unsigned int total;
    const unsigned int slot = slot0 + block_exclusive_scan(flag, wtot, &total);
    slot0 += total;
    if (flag) {
      const float4 p0 = pts[id0];
      float sx = __fadd_rn(0.f, p0.x), sy = __fadd_rn(0.f, p0.y), sz = __fadd_rn(0.f, p0.z), st = __fadd_rn(0.f, p0.w);
      unsigned int j = i + 1;
      if (k_next == k) {  // the run goes on (a second point of the voxel follows): the general walk
        for (; j < (unsigned int)n && keys[j] == k; j++) {
          const float4 p = pts[idx[j]];
          sx = __fadd_rn(sx, p.x); sy = __fadd_rn(sy, p.y); sz = __fadd_rn(sz, p.z); st = __fadd_rn(st, p.w);
        }
      }
      if (max_run && j - i > 8u) atomicMax(max_run, j - i);  // (the host picks the hashed filter for sparse voxels: lii_capi.cpp)
      const float c = (float)(j - i);
      // a single-point voxel reproduces the point exactly (x / 1.0f == x), which is also what the identity path needs
      out[slot] = make_float4(sx / c, sy / c, sz / c, st / c);
      pcl_out[slot] = pcl_in[id0];
    }
  }
}

int buckets_for(int n) {
  int want = (n + 63) / 64, b = 8;
  while (b < want && b < kMaxBuckets) b <<= 1;
  return b;
}
}  // namespace

// tot | cursor | bucket_start (B + 1) | group_count (<= B)
size_t voxel_sort_hist_elems(int) { return 4 * (size_t)kMaxBuckets + 2; }

// Sampling plan for n pairs: S = 2B strata of `width` consecutive positions, one jittered sample each (k_voxel_keys draws
// them while it writes the keys — a regular stride aliases with the scan pattern of a spinning LiDAR and doubles the largest
// bucket).
VoxelSortPlan voxel_sort_plan(int n) {
  VoxelSortPlan p;
  p.buckets = n > 512 ? buckets_for(n) : 0;
  p.samples = 2 * p.buckets;
  p.width = p.samples ? (n + p.samples - 1) / p.samples : 1;
  p.strata = p.samples ? (n + p.width - 1) / p.width : 0;
  return p;
}

void launch_voxel_sort_centroids(const VoxelSortBuffers& vb, const float4* pts, int n, float4* out, int* n_out, hipStream_t s) {
  if (n <= 0) return;
  unsigned int *tot = vb.hist, *cursor = vb.hist + kMaxBuckets, *bucket_start = vb.hist + 2 * kMaxBuckets,
               *group_count = vb.hist + 3 * kMaxBuckets + 2;
  if (n <= 512) {
    hipLaunchKernelGGL(k_vsort_small, dim3(1), dim3(512), 0, s, vb.keys_in, n, vb.keys_out, vb.idx_out, bucket_start, group_count);
    hipLaunchKernelGGL(k_voxel_centroids<kGroup>, dim3(1), dim3(256), 0, s, pts, vb.keys_out, vb.idx_out, bucket_start, kGroup, n,
                       group_count, out, n_out, vb.pcl_in, vb.pcl_out, vb.max_run);
    return;
  }
  const VoxelSortPlan p = voxel_sort_plan(n);
  const int B = p.buckets, S = p.samples, G = (n + kTile - 1) / kTile, groups = (B + kGroup - 1) / kGroup;
  const int n_splitters = min(B - 1, (p.strata - 1) / (S / B));  // ranks per, 2 per, ... below the number of real samples
  if (S >= 256)
    hipLaunchKernelGGL(k_vsort_splitters, dim3(S / 8), dim3(256), (size_t)S * sizeof(u64), s, vb.samples, p.strata, S, B, vb.splitters);
  else
    hipLaunchKernelGGL(k_vsort_splitters_small, dim3(1), dim3(256), 0, s, vb.samples, p.strata, S, B, vb.splitters);
  hipLaunchKernelGGL(k_vsort_classify, dim3(G), dim3(256), 0, s, vb.keys_in, n, B, vb.splitters, n_splitters, vb.bucket_of, tot);
  hipLaunchKernelGGL(k_vsort_scatter, dim3(G), dim3(256), 0, s, vb.keys_in, n, B, vb.bucket_of, tot, cursor, bucket_start, vb.comp);
  if (n <= B * 64) {  // buckets of ~<= 64 pairs: four to a workgroup
    hipLaunchKernelGGL(k_vsort_local<kGroup>, dim3(groups), dim3(256), 0, s, vb.comp, bucket_start, B, tot, cursor, vb.keys_out,
                       vb.idx_out, group_count);
    hipLaunchKernelGGL(k_voxel_centroids<kGroup>, dim3(groups), dim3(256), 0, s, pts, vb.keys_out, vb.idx_out, bucket_start, B, n,
                       group_count, out, n_out, vb.pcl_in, vb.pcl_out, vb.max_run);
  } else {  // the bucket count is capped (n > 131 k): larger buckets, one to a workgroup
    hipLaunchKernelGGL(k_vsort_local<1>, dim3(B), dim3(256), 0, s, vb.comp, bucket_start, B, tot, cursor, vb.keys_out, vb.idx_out,
                       group_count);
    hipLaunchKernelGGL(k_voxel_centroids<1>, dim3(B), dim3(256), 0, s, pts, vb.keys_out, vb.idx_out, bucket_start, B, n, group_count,
                       out, n_out, vb.pcl_in, vb.pcl_out, vb.max_run);
  }
}

}  // namespace lii
