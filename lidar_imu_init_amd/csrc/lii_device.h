// Internal device-side declarations of libliinit_hip (gfx950 only).  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lii {

constexpr int kMatch = 5;          // NUM_MATCH_POINTS — reference include/common_lib.h:28
constexpr int kNeedy = 0x100;      // nbr_count flag: the 3x3x3 search pass could not prove this list exact yet
constexpr int kDone = 0x400;       // ... finished by a COMPLETION workgroup of the fit launch behind the search pass: the workgroups of the cloud leave the
                                   // point out whether they read its count before (kNeedy) or after (kDone) the completion stored it (ownership must
                                   // not depend on when a workgroup of the same launch looks).  Readers of the count mask with kCountMask.
constexpr int kCountMask = 0xFF;
constexpr int kCovered = 0x200;    // ... but measured every point of those cells: the list is exact over them (the completion skips them)
constexpr int kBlock = 256;        // 4 wavefronts of 64
constexpr int kNormalEq = 91;      // 78 + 12 + 1
constexpr int kCoarseShift = 3;    // coarse occupancy cell = 8 x 8 x 8 fine cells
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kCellBias = 1 << 20;
// voxel-filter sort: 64-bit composites = (sort key << kVoxIdxBits) | point index.  An order key over the voxels of a grid PCL accepts
// (dx dy dz < 2^31) needs at most 31 + 3 bits (every axis width rounded up to a power of two); 28 bits index 268 M points.
constexpr int kVoxIdxBits = 28;
constexpr int kVoxKeyBits = 36;
constexpr unsigned long long kVoxDropKey = (1ull << kVoxKeyBits) - 1ull;

// device-resident counters of the in-place map (lii_map.hip): slots used at the tail of the point array, live points, occupied
// 8x8x8 blocks, entries of the work list of the update in flight, "a capacity was exceeded" flag, events of the last fold
constexpr int kMapCtrUsed = 0, kMapCtrValid = 1, kMapCtrBlocks = 2, kMapCtrWork = 3, kMapCtrOverflow = 4, kMapCtrEvents = 5, kMapCtrDropped = 6, kMapCtrSlots = 7, kMapCtrWinStale = 8 /* an in-place update touched a cell outside the dense window: the window no longer mirrors the cell tables */, kMapCtrTicket = 15 /* k_ins_write: workgroups that are done */, kMapCtrWords = 16;

// 3 x 3 row-major helpers (no FMA contraction in the units that use them: the reference's rounding)
__device__ __forceinline__ void mat3_mul(const double A[9], const double B[9], double C[9]) {
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
__device__ __forceinline__ void mat3_vec(const double A[9], const double v[3], double o[3]) {
#pragma unroll
  for (int r = 0; r < 3; r++) o[r] = A[3 * r] * v[0] + A[3 * r + 1] * v[1] + A[3 * r + 2] * v[2];
}
__device__ __forceinline__ void mat3t_vec(const double A[9], const double v[3], double o[3]) {
#pragma unroll
  for (int r = 0; r < 3; r++) o[r] = A[r] * v[0] + A[3 + r] * v[1] + A[6 + r] * v[2];
}

// Pose the per-point kernels need: state.rot_end, pos_end, offset_R_L_I, offset_T_L_I (row-major).
struct PoseArg {
  double R[9];
  double p[3];
  double RLI[9];
  double TLI[3];
};

// The pose a per-point kernel works with lives in DEVICE MEMORY on every path (the control block of the device-driven loop, or the
// handle's pose slot, which a host-driven pass uploads first) and is read in one batch of loads: one round
// trip.  Rounds 1 - 4 also carried a pose by value in the kernel arguments and chose with `cond ? *pose : by_value`: the compiler
// selected the 24 doubles ONE BY ONE - 24 conditional scalar loads, sixteen of them behind a wait for the one before (three of those
// cache misses: the block was written by the previous launch, on another XCD), at the head of every search and fit launch - and the
// second copy of the pose took the scalar registers the first requests of the kernel needed to go out together.
// What the head of a search or fit launch needs from memory besides its point: the pose (24 doubles = three 64-byte lines), the loop
// flags (IekfCtrl::search_next, ::stop: neighbours in the block) and the device-resident size of the cloud.  Five SCALAR loads,
// issued back to back and waited for once - written out as instructions, because the compiler cannot be made to: it sinks scalar
// loads to their first use (behind the waits for everything else), splits a batch it has no registers for into dependent ones, and
// turned round 4's `cond ? *pose : by_value` into 24 conditional loads with a wait between them.  (Round 5 first read the pose with
// vector loads - one batch as well, and 0.7 us off every fit launch of the 100 k-point stream - but thirteen more vector-memory
// instructions per wavefront cost the 500 k-point scan 6 us per search launch: profiles/r05_head_loads.md.)
// The caller issues its own vector loads (the point's data) BEFORE this call: the "memory" clobber keeps them in front, and they are
// in flight while the wavefront waits for the scalars.  `n_ptr` must be a valid address (a kernel whose cloud size is not on the
// device passes any int it may read and ignores the value).
typedef unsigned int lii_u16v __attribute__((ext_vector_type(16)));
typedef unsigned int lii_u2v __attribute__((ext_vector_type(2)));
struct HeadScalars {
  PoseArg ps;
  int search_next, stop, n_mem, extra;
};
// extra_ptr: one more int the kernel wants with the batch (a valid address; k_fit_reduce: the number of listed queries)
__device__ __forceinline__ HeadScalars load_head_scalars(const PoseArg* __restrict__ pose, const int* __restrict__ search_next_and_stop,
                                                         const int* __restrict__ n_ptr, const int* __restrict__ extra_ptr) {
  lii_u16v l0, l1, l2;
  lii_u2v fl;
  unsigned int nm, ex;
  asm volatile(
      "s_load_dwordx16 %0, %6, 0x0\n\t"
      "s_load_dwordx16 %1, %6, 0x40\n\t"
      "s_load_dwordx16 %2, %6, 0x80\n\t"
      "s_load_dwordx2 %3, %7, 0x0\n\t"
      "s_load_dword %4, %8, 0x0\n\t"
      "s_load_dword %5, %9, 0x0\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(l0), "=&s"(l1), "=&s"(l2), "=&s"(fl), "=&s"(nm), "=&s"(ex)
      : "s"(pose), "s"(search_next_and_stop), "s"(n_ptr), "s"(extra_ptr)
      : "memory");
  auto dbl = [](unsigned int lo, unsigned int hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); };
  HeadScalars h;
  PoseArg& m = h.ps;
  m.R[0] = dbl(l0[0], l0[1]); m.R[1] = dbl(l0[2], l0[3]); m.R[2] = dbl(l0[4], l0[5]); m.R[3] = dbl(l0[6], l0[7]);
  m.R[4] = dbl(l0[8], l0[9]); m.R[5] = dbl(l0[10], l0[11]); m.R[6] = dbl(l0[12], l0[13]); m.R[7] = dbl(l0[14], l0[15]);
  m.R[8] = dbl(l1[0], l1[1]);
  m.p[0] = dbl(l1[2], l1[3]); m.p[1] = dbl(l1[4], l1[5]); m.p[2] = dbl(l1[6], l1[7]);
  m.RLI[0] = dbl(l1[8], l1[9]); m.RLI[1] = dbl(l1[10], l1[11]); m.RLI[2] = dbl(l1[12], l1[13]); m.RLI[3] = dbl(l1[14], l1[15]);
  m.RLI[4] = dbl(l2[0], l2[1]); m.RLI[5] = dbl(l2[2], l2[3]); m.RLI[6] = dbl(l2[4], l2[5]); m.RLI[7] = dbl(l2[6], l2[7]);
  m.RLI[8] = dbl(l2[8], l2[9]);
  m.TLI[0] = dbl(l2[10], l2[11]); m.TLI[1] = dbl(l2[12], l2[13]); m.TLI[2] = dbl(l2[14], l2[15]);
  h.search_next = (int)fl[0]; h.stop = (int)fl[1]; h.n_mem = (int)nm; h.extra = (int)ex;
  return h;
}
// The same pose in VECTOR registers (24 moves): a kernel that keeps the pose for its whole life has no scalar registers for it -
// k_fit_reduce spilled 62 of them into vector-register lanes and fetched them back one v_readlane at a time.
__device__ __forceinline__ PoseArg pose_to_vgprs(const PoseArg& s) {
  PoseArg v;
  auto mv = [](double x) { double r; asm volatile("v_mov_b64 %0, %1" : "=v"(r) : "s"(x)); return r; };
#pragma unroll
  for (int k = 0; k < 9; k++) { v.R[k] = mv(s.R[k]); v.RLI[k] = mv(s.RLI[k]); }
#pragma unroll
  for (int k = 0; k < 3; k++) { v.p[k] = mv(s.p[k]); v.TLI[k] = mv(s.TLI[k]); }
  return v;
}
__device__ __forceinline__ PoseArg load_pose(const PoseArg* __restrict__ pose) { return *pose; }  // (no hurry: k_map_decide)
// (a kernel that still takes a pose by value beside the pointer - k_map_decide: the caller's final state, or the control block's)
__device__ __forceinline__ PoseArg load_pose(bool from_memory, const PoseArg* __restrict__ pose, const PoseArg& by_value) {
  PoseArg ps = by_value;
  if (from_memory) ps = load_pose(pose);  // (uniform)
  return ps;
}

// One occupied 8x8x8 block of grid cells: open-addressing table entry  block key -> dense block id.
struct __attribute__((aligned(16))) BlockEntry {
  unsigned long long key;  // packed (Bz, By, Bx) = cell coordinate >> 3; kEmptyKey = free slot
  unsigned int id;         // index of the block's 512-entry cell table
  unsigned int pad;
};

// Device view of the local-map k-NN index.  Points are sorted by (block key, local cell index x-fastest), so
// the cells of one block are contiguous and neighbouring cells sit next to each other in `cells`:
//   cells[id * 512 + (lz*64 + ly*8 + lx)] = (first, one-past-last) index into pts of that cell (0,0 if empty).
// The block table is tiny (a few thousand entries for a 1 M-point map) and stays cache resident; the cell
// tables are spatially coherent, unlike a per-cell hash.
struct GridView {
  const float4* pts;  // xyz + w = bit-cast insertion id
  const BlockEntry* blocks;
  const uint2* cells;
  unsigned int block_mask;
  int n_pts;
  float cs;
  float inv_cs;
  float max_d2;
  // Dense CELL WINDOW (round 6): the cell entries of the map's bounding box, direct-indexed by cell coordinate -
  // win[((cz - wz0) * wny + (cy - wy0)) * wnx + (cx - wx0)] = the (first, end) entry the block table + cell table would give - so
  // that a lookup inside the box is ONE load instead of the dependent pair block probe -> cell entry.  nullptr: no window (the box
  // exceeds the budget, or the map has changed since the window was filled); a cell outside the box goes through the hash.
  const uint2* win;
  int wx0, wy0, wz0, wnx, wny, wnz;
};

// The dense cell window kept current through an in-place map update (round 6): whatever rewrites a cell entry (k_cell_apply) or counts an
// insert into it (k_ins_write) does the same to the window's copy.  key_of_id[id] = packed biased block coordinates of block `id` (filled by
// k_cells_fill and by the lane of k_ins_cells that creates a block); a cell outside the box raises ctr[kMapCtrWinStale] and the host drops
// the window with the update's counters.  win == nullptr: no window to keep.
struct WinKeep {
  uint2* win;
  const unsigned long long* key_of_id;
  int x0, y0, z0, nx, ny, nz;
};
__device__ __forceinline__ long long win_index_of_entry(const WinKeep& w, unsigned int e) {  // -1: outside the window
  const unsigned long long k = w.key_of_id[e >> 9];
  const int bb = kCellBias >> kCoarseShift;
  const int bx = (int)(k & 0x3FFFF) - bb, by = (int)((k >> 18) & 0x3FFFF) - bb, bz = (int)((k >> 36) & 0x3FFFF) - bb;
  const unsigned int l = e & 511u;
  const unsigned int ux = (unsigned)(bx * 8 + (int)(l & 7u) - w.x0), uy = (unsigned)(by * 8 + (int)((l >> 3) & 7u) - w.y0), uz = (unsigned)(bz * 8 + (int)(l >> 6) - w.z0);
  if (ux < (unsigned)w.nx && uy < (unsigned)w.ny && uz < (unsigned)w.nz) return ((long long)uz * w.ny + uy) * w.nx + ux;
  return -1;
}

struct RegistrationBuffers {
  const float4* body;   // down-sampled LiDAR-frame points (x,y,z,t)
  float4* world;        // world coordinates of the last pass (x,y,z,-)
  float4* nbr;          // SoA: nbr[k * cap + i], k < 5 (xyz of neighbour k, w = d2)
  int* nbr_count;       // neighbours found (0..5)
  double* plane;        // 4 doubles per point: n̂, d  (pabcd)
  unsigned char* selected;
  double* partials;     // transposed per-block partial sums: partials[t * partial_stride + block], t < 91
  int partial_stride;
  int n;              // number of points, or an upper bound of it when n_dev != nullptr
  int cap;
  const int* n_dev;   // device-resident point count (set by the sync-free voxel filter), or nullptr
  int shard_rank;     // points of one scan sharded across ranks: this rank registers the contiguous
  int shard_world;    // block [n * rank / world, n * (rank + 1) / world) of the down-sampled cloud; world <= 1: all of it
  // The queries a search pass could not finish (kNeedy), listed by that pass for the completion workgroups of the fit launch behind
  // it (k_fit_reduce): flag_count[e & 1] entries in flag_list[(e & 1) * kListCap ...], e = the search launch's number (`epoch`); the first
  // kListCap are stored.
  // An entry is everything the completion needs to start with: (world point, query index) and (neighbour count with its flags, -, -, -).
  int* flag_count;
  float4* flag_list;  // 2 x kListCap entries of two float4
};
constexpr int kFlagCap = 256;          // listed queries a fit launch hands to its completion workgroups (more: every workgroup finishes its own, as in round 4 ...
constexpr int kListCap = 4096;         // ... unless a launch of its own in front of the fit launch has finished the listed ones, one wavefront per query:
                                       // k_complete_listed, enqueued when the scan before listed more than kFlagCap - the launch plan's way)
#ifndef LII_COMPLETION_BLOCKS
#define LII_COMPLETION_BLOCKS 32
#endif
#ifndef LII_COMPLETION_BLOCKS_PRE
#define LII_COMPLETION_BLOCKS_PRE 96
#endif
// (a fit launch BEHIND A SEARCH LAUNCH that lists its unfinished queries - epoch > 0 - gets kCompletionBlocksPre of them: up to that many listed
// queries are one per workgroup, finished by its four wavefronts together; the launches on cached planes keep kCompletionBlocks, which only write a zero column)
constexpr int kCompletionBlocksPre = LII_COMPLETION_BLOCKS_PRE;
__host__ __device__ inline int completion_blocks(int epoch) { return epoch > 0 ? kCompletionBlocksPre : LII_COMPLETION_BLOCKS; }
constexpr int kCompletionBlocks = LII_COMPLETION_BLOCKS;  // ... of which there are this many, behind the workgroups of the cloud; each writes one more column of partial sums

// The block of the down-sampled cloud this rank registers: first index and size.  Every rank holds the WHOLE cloud (the
// de-skew and the voxel filter run replicated, so the cloud is bit-identical everywhere) and the split needs no exchange.
__device__ __forceinline__ void shard_range_n(const RegistrationBuffers& rb, int n_mem /* the cloud's size as loaded from rb.n_dev */, int& lo, int& n_live) {
  const int n_all = rb.n_dev ? n_mem : rb.n;
  lo = 0;
  n_live = n_all;
  if (rb.shard_world > 1) {
    lo = (int)(((long long)n_all * rb.shard_rank) / rb.shard_world);
    n_live = (int)(((long long)n_all * (rb.shard_rank + 1)) / rb.shard_world) - lo;
  }
}
__device__ __forceinline__ void shard_range(const RegistrationBuffers& rb, int& lo, int& n_live) {
  const int n_all = rb.n_dev ? *rb.n_dev : rb.n;
  lo = 0;
  n_live = n_all;
  if (rb.shard_world > 1) {
    lo = (int)(((long long)n_all * rb.shard_rank) / rb.shard_world);
    n_live = (int)(((long long)n_all * (rb.shard_rank + 1)) / rb.shard_world) - lo;
  }
}

// Device-resident control block of one scan registration (lii_iekf_update): the state, the propagated state and
// the loop flags of src/laserMapping.cpp:957-1134 live in HBM so that the whole iterated update is enqueued once
// and synchronised once.  The first 24 doubles of `st` are exactly a PoseArg (rot_end, pos_end, offset_R_L_I,
// offset_T_L_I of lii_state), which the per-point kernels read through a pointer.
constexpr int kStateDoubles = 36 + 24 * 24;
struct IekfCtrl {
  double st[kStateDoubles];    // lii_state: current estimate (in/out); its first 24 doubles are the PoseArg the kernels read
  double prop[36];             // state_propagat without its covariance (only used through boxminus)
  double solution[24];
  int max_it;
  int imu_en;
  int it;            // iterations executed so far
  int search_next;   // nearest_search_en of the next pass
  int stop;          // EKF_stop_flg
  int rematch_num;
  int converged;     // flg_EKF_converged of the last iteration
  int searches;      // k-NN passes executed
  int effect_num;    // effect_feat_num of the last iteration
  int singular;      // a matrix inversion failed
  int seq;           // number of this update (host); echoed into IekfResult::done by the iteration that ends the loop
  unsigned int plan_mask;  // bit k (k < 16): the host enqueued a k-NN launch ahead of iteration k; bit 16 + k: it enqueued pass k
                           // at all (iterations >= 16: always both).  The host predicts the search pattern and the number of
                           // passes from the previous scans and leaves out the launches that would only read the flags and
                           // return (~4.5 us each on the device); a solve whose next pass is not there, or asks for a search
                           // the plan does not hold, parks the loop (stop = 2) and tells the host, which enqueues the rest.
  int search_log[16];  // search_log[it]: bit 0 = iteration `it` ran the k-NN pass (for profiling and the next launch plan), bit 1 = its
                       // elimination left the pivot-free loop for the pivoting routine (lii_last_solve_info)
  double search_pose[24];  // the PoseArg of the last executed k-NN pass (written by that pass): a sharded job re-runs the search
                           // for the blocks of the other ranks at exactly this pose before map_incremental (lii_capi.cpp)
};

// "Last workgroup finishes the job" kernels publish their partial results with device-scope ATOMIC stores / adds (performed
// at the coherent level, read back with device-scope atomic loads) and then draw a ticket.  All the ticket needs is that
// those atomics have completed: s_waitcnt vmcnt(0).  __threadfence() would also write the L2 back (buffer_wbl2) and
// invalidate it - measured at ~5 us per launch when every workgroup does it.
__device__ __forceinline__ void wait_published_atomics() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // compiler ordering (+ LDS) ...
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // ... and vmcnt(0): the global atomics have been acknowledged
}

// Deterministic final sum of one row of the transposed partials (one value per fit workgroup) by a workgroup of BS lanes:
// lane l adds the values l, l + BS, l + 2 BS, ... in that order (four loads in flight: the partials were written by other
// XCDs, every load is an L2 miss), a fixed shuffle tree joins the lanes of a wavefront, the wavefront sums are added in
// order.  k_reduce91 and k_reduce_solve share it so that the fused and the three-launch (RCCL) forms of the loop produce
// bit-identical sums.  Returns the sum in every lane of the calling workgroup's thread 0 (other lanes: unspecified).
template <int BS>
__device__ __forceinline__ double final_sum_row(const double* __restrict__ row, int n_blocks, double* s_w /* [BS / 64] LDS */) {
  const int tid = threadIdx.x;
  double acc = 0;
  for (int b0 = tid; b0 < n_blocks; b0 += BS * 4) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = b0 + BS * u < n_blocks ? row[b0 + BS * u] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (b0 + BS * u < n_blocks) acc += v[u];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((tid & 63) == 0) s_w[tid >> 6] = acc;
  __syncthreads();
  double total = s_w[0];
#pragma unroll
  for (int w = 1; w < BS / 64; w++) total += s_w[w];
  return total;
}

// Prefix of per-workgroup counts INSIDE one launch, placement-independent (MI355X guide, Contract [G]: "HIP promises nothing about
// dispatch order ... deadlock or stale data if an order or a co-location is assumed; placement-independent protocols only").
// Every workgroup publishes its word - (epoch << 32 | a << 16 | b), a, b <= 256: counts over its 256 points; epoch: the number of
// this run, so that a word left by an earlier run is never mistaken - BEFORE it looks at anybody else's, then adds up the words of
// the workgroups below it.  A word that has not arrived after a bounded wait is NOT waited for any longer: its workgroup may not
// have been dispatched yet - the hardware queues of several processes share the compute units, an XCD can fall behind the others -
// and may be unable to start before THIS workgroup has left.  The waiting workgroup then counts that block itself (`recount(q)`:
// all 256 lanes, uniform; what the block's own workgroup will find), publishes the word on its behalf - the value is the same
// whoever writes it - and goes on: no workgroup ever depends on one that is not running.  Rounds 3 - 4 spun until the word came and
// trapped after ~1 s ("workgroups are dispatched in index order: the lowest unfinished one waits for nobody" - true of ONE XCD's
// queue, not across the eight: eight processes on one device locked each other out XCD by XCD).
// The fast path - every word there at the first or second look - is the old one: no ticket, no extra round trip.
// Returns (sum of a, sum of b) over the blocks below `my_block` in every lane.  `late_test`: tests only - treat every word that is
// not there at the first look as overdue.
constexpr unsigned int kPrefixSpinLimit = 256;  // looks at one word before giving up on it (~1 us each: far beyond a healthy launch's skew)
template <class Recount>
__device__ __forceinline__ uint2 prefix_below(unsigned long long* __restrict__ words, unsigned int epoch, int my_block, unsigned int* s_sum /*[12] LDS*/,
                                              bool late_test, Recount&& recount) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned int pa = 0, pb = 0;
  bool late = false;
  for (int q = threadIdx.x; q < my_block; q += 256) {
    unsigned long long v = __hip_atomic_load(words + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int spins = 0;
    while ((unsigned int)(v >> 32) != epoch) {
      if (late_test || ++spins > kPrefixSpinLimit) { late = true; break; }
      __builtin_amdgcn_s_sleep(2);
      v = __hip_atomic_load(words + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (late) break;
    pa += ((unsigned int)v >> 16) & 0xFFFFu;
    pb += (unsigned int)v & 0xFFFFu;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { pa += __shfl_down(pa, off); pb += __shfl_down(pb, off); }
  const bool wave_late = __any(late);
  if (lane == 0) { s_sum[w] = pa; s_sum[4 + w] = pb; s_sum[8 + w] = wave_late ? 1u : 0u; }
  __syncthreads();
  if (!(s_sum[8] | s_sum[9] | s_sum[10] | s_sum[11]))  // (uniform) the usual case
    return make_uint2(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3], s_sum[4] + s_sum[5] + s_sum[6] + s_sum[7]);
  // Some word is overdue: the blocks below one after the other, the whole workgroup together (uniform control flow: lane 0 looks,
  // everybody takes what it saw).  A word that is there is taken; a block whose word is not is counted here and now.  The recount
  // is only valid while the block's own workgroup has not begun to change what it is counted from, which it does AFTER its word is
  // out: the word is looked at again behind the recount, and if it has arrived in the meantime it wins.
  __shared__ unsigned long long s_word;
  unsigned int sa = 0, sb = 0;
  for (int q = 0; q < my_block; q++) {
    __syncthreads();
    if (threadIdx.x == 0) s_word = __hip_atomic_load(words + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    unsigned long long v = s_word;
    if ((unsigned int)(v >> 32) != epoch) {  // uniform
      const unsigned int mine = recount(q);  // (a << 16 | b; contains barriers)
      __syncthreads();
      if (threadIdx.x == 0) s_word = __hip_atomic_load(words + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      v = s_word;
      if ((unsigned int)(v >> 32) != epoch) {
        v = ((unsigned long long)epoch << 32) | mine;
        if (threadIdx.x == 0) __hip_atomic_store(words + q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    sa += ((unsigned int)v >> 16) & 0xFFFFu;
    sb += (unsigned int)v & 0xFFFFu;
  }
  return make_uint2(sa, sb);
}

// Node-local exchange of the 91 normal-equation scalars between the ranks of one job, inside the
// reduce+solve launch.  Rank r owns two slots per SOURCE rank (parity of the exchange number) of 96 doubles + a sequence flag.
// Two homes for the slots:
//   * in HBM, peer-mapped (the default on one node): every rank allocates fine-grained device memory for the slots it READS
//     and exports it through a HIP IPC handle; an exchange PUSHES the rank's 91 sums and then the flag into every peer's
//     memory - remote stores over xGMI, one fabric hop - and polls / sums its own memory only (`peers` != nullptr);
//   * in a POSIX shared-memory segment that every rank has registered with its own device (host memory over PCIe: the fallback
//     when the IPC handles cannot be opened): write own slot, publish the flag, poll and read the peers' slots.
// Either way the slots are summed in rank order (every rank forms the bit-identical sum), and two parities suffice: a rank
// can be at most one exchange ahead of the slowest reader.
constexpr int kMailboxSlotDoubles = 128;               // 1 KiB: [0..90] data, [96] sequence flag (as u64)
constexpr int kMailboxFlagAt = 96;
constexpr int kMailboxMaxRanks = 64;                   // one polling lane per rank
struct MailboxView {
  double* slots;            // host-memory form: device address of the registered segment's slot area; nullptr = no exchange
  double* const* peers;     // HBM form: peers[r] = rank r's slot area as mapped into this process (peers[rank] = the own one)
  unsigned long long* seq;  // device memory: exchanges completed so far (identical on all ranks)
  int n_ranks, rank;
  long long timeout_ticks;  // of the 100 MHz wall clock: how long to wait for a peer that may never arrive
  long long handoff_ticks;  // ... and for the sums of the launch's own summing workgroups (k_reduce_solve; 2 s)
  int test_drop_sum;        // LII_TEST=sum_lost: summing workgroup 5 of every launch keeps its sum to itself (tests: the solver's bounded wait)
};

// The second exchange of a sharded job, once per map update (lii_map_incremental; kernels in lii_exchange.hip): every rank decides
// for ITS points of the down-sampled cloud which ones enter the map, and the two lists - PointToAdd, PointNoNeedDownsample: a few
// thousand float4 - are pushed into every rank's gather area (peer-mapped fine-grained HBM behind the mailbox slots, remote stores
// over xGMI) and put together in rank order there, so that all replicas of the map receive the identical batch.  One block per
// (parity, source rank): a 64-byte header {u64 sequence flag, i32 n_add, i32 n_nodown} and the payload (the add list, then the
// no-down-sample list).  Two parities for the same reason as the slots.
constexpr int kGatherHeaderBytes = 64;
struct GatherView {
  unsigned char* const* peers;  // peers[r] = rank r's gather area as mapped into this process (peers[rank]: the own one); nullptr = no exchange
  size_t block_bytes;           // header + payload capacity
  int cap_points;               // payload capacity in float4 (the handle's max_scan_points: a rank's lists never hold more)
  int n_ranks, rank;
  long long timeout_ticks;
};

// Called by ONE wavefront (64 lanes).  Lane l holds this rank's sums l (v0) and l + 64 (v1; lanes 0 .. 26) and receives the
// sums over the ranks in their place.  Returns false when a peer did not publish within the time limit (the communicator is
// unusable afterwards).
__device__ inline bool mailbox_allreduce(const MailboxView& mb, double& v0, double& v1) {
  const int lane = threadIdx.x & 63;
  const unsigned long long q = *mb.seq + 1ull;
  const int par = (int)(q & 1ull);
  const double* rd;   // where this rank reads the slots from: slot of source rank r at rd + (r * 2 + par) * kMailboxSlotDoubles
  if (mb.peers) {
    for (int r = 0; r < mb.n_ranks; r++) {  // push into every rank's memory (the own one included)
      double* dst = mb.peers[r] + (size_t)(mb.rank * 2 + par) * kMailboxSlotDoubles;
      if (lane < kNormalEq) __hip_atomic_store(dst + lane, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (lane + 64 < kNormalEq) __hip_atomic_store(dst + lane + 64, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();  // the wavefront's data stores are complete before the flags leave
    if (lane < mb.n_ranks)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(mb.peers[lane] + (size_t)(mb.rank * 2 + par) * kMailboxSlotDoubles + kMailboxFlagAt), q,
                         __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    rd = mb.peers[mb.rank];
  } else {
    double* mine = mb.slots + (size_t)(mb.rank * 2 + par) * kMailboxSlotDoubles;
    if (lane < kNormalEq) __hip_atomic_store(mine + lane, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lane + 64 < kNormalEq) __hip_atomic_store(mine + lane + 64, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (lane == 0)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine + kMailboxFlagAt), q, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    rd = mb.slots;
  }
  const long long t0 = wall_clock64();
  bool ok = true;
  for (;;) {
    unsigned long long v = q;
    if (lane < mb.n_ranks)
      v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(rd + (size_t)(lane * 2 + par) * kMailboxSlotDoubles + kMailboxFlagAt),
                            __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (__all(v >= q)) break;
    if (wall_clock64() - t0 > mb.timeout_ticks) { ok = false; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (ok) {
    for (int i = lane; i < kNormalEq; i += 64) {
      double acc = 0;
#pragma unroll 8
      for (int r = 0; r < mb.n_ranks; r++)
        acc += __hip_atomic_load(rd + (size_t)(r * 2 + par) * kMailboxSlotDoubles + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (i < 64) v0 = acc; else v1 = acc;
    }
  }
  __threadfence();
  if (lane == 0) *mb.seq = q;
  return ok;
}

// What the host needs back from one iterated update.  Lives in pinned, device-mapped HOST memory: the solve kernel of the
// stopping iteration writes it there directly, so the update ends with one stream synchronisation and no D2H copy.
constexpr int kLoopParked = 0x40000000;
struct IekfResult {
  double st[kStateDoubles];
  double ne[96];  // the 91 normal-equation scalars of the last executed pass
  int it, searches, effect_num, converged, singular;
  int done;  // == IekfCtrl::seq once everything above has landed (written last, after a system-scope fence): the host polls it
             // (== seq | kLoopParked: the loop is parked ahead of iteration `parked_it`, see IekfCtrl::plan_mask)
  int parked_it;
  int part_overflow;  // set by the voxel filter of a voxel-partitioned job (k_vhash_emit): this rank's share outgrew its bound
  int search_log[16];
  int parked_search;  // ... and that iteration searches (1) or not (0): the host puts a k-NN launch in front of it only then
  int unfinished;     // queries the scan's search passes left unfinished (max over its two most recent search launches; k_reduce_solve)
  long long ts[16];  // LII_SOLVE_TRACE builds: wall_clock64 stamps of the solve phases (stopping iteration)
  long long ts0[16]; // ... of iteration 0
};

}  // namespace lii
