// Device-side maintenance of the local map with the reference's incremental k-d tree SET semantics, IN PLACE (DESIGN.md
// section 2): the point array keeps slack behind every cell, an update touches the cells of its batch only.
//   k_map_decide                    map_incremental's per-point decision and its two lists         src/laserMapping.cpp:516-553
//   k_add_keys / k_add_fold8        KD_TREE::Add_Points(points, downsample_on = true)  include/ikd-Tree/ikd_Tree.cpp:381-426:
//                         per down-sample voxel "keep the point closest to the voxel centre": the points of one batch that
//                         fall into the same voxel interact sequentially, different voxels are independent - eight lanes
//                         fold one voxel group in the batch order (stable sort by voxel), querying the current map grid with
//                         the reference's exact float box predicate (Search_by_range / Delete_by_range, :616-629, :970-985).
//                         Round 6, lii_map_incremental on one rank: the hash insert of the hash-grouped fold rides in k_map_decide, the
//                         inserts' cells (k_ins_cells) in the fold launch - k_add_fold8<true, true> - : four launches per update.
//   k_ins_cells -> k_cell_apply -> k_ins_write   apply the tombstones and the inserts cell by cell: find / create the cell of
//                         every insert, squeeze the touched cells (moving one to the tail of the array when its slack is used
//                         up), claim a slot per insert.  Inserts that find no room wait in a list for the host's rebuild.
//   k_box_tomb_cells      KD_TREE::Delete_Point_Boxes                                  include/ikd-Tree/ikd_Tree.cpp:500-516
//   k_cell_caps / k_spread / k_cell_counts / k_gather_live   (re)build: compact cell-sorted array <-> slack layout
// Set equivalence with the tree is tested against the oracle's restated tree and against the unmodified reference tree
// (tests/test_gpu_map.py).  Ties between equal squared distances to the voxel centre are broken by map order here and by
// tree traversal order in the reference (measure zero).
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <math.h>

#include "lii_device.h"
#include "lii_launch.h"

namespace lii {

namespace {
constexpr int kBias = 1 << 20;
constexpr int kCells = 512;
constexpr unsigned long long kInvalidKey = ~0ull;

__device__ __forceinline__ unsigned int d_hash_block(int bx, int by, int bz) {
  return (__umul24((unsigned)bx, 7919u * 1021u) ^ __umul24((unsigned)by, 104729u * 13u) ^ __umul24((unsigned)bz, 1299709u)) * 2654435761u;
}
__device__ __forceinline__ unsigned long long d_pack_block(int bx, int by, int bz) {
  return ((unsigned long long)(unsigned)bz << 36) | ((unsigned long long)(unsigned)by << 18) | (unsigned long long)(unsigned)bx;
}
// FRESH: blocks may be CREATED beside this lookup, in the same launch (k_add_fold8<true, true>: the inserts' cells ride in the fold).  A block
// whose key is there and whose id is not yet (pad == 0; k_ins_cells' creator stores key -> id -> pad, the pad with release order) is a block
// of this very launch: it holds no point yet - empty, like a block that is not in the table.  id and pad are ONE aligned 8-byte word: a copy
// of the entry that shows pad = 1 shows the id that was stored before it, however old the copy of the key beside it is.
template <bool FRESH = false>
__device__ __forceinline__ uint2 d_cell_range(const GridView& g, int ix, int iy, int iz) {
  const int bb = kBias >> kCoarseShift;
  const int bx = (ix >> kCoarseShift) + bb, by = (iy >> kCoarseShift) + bb, bz = (iz >> kCoarseShift) + bb;
  const unsigned long long bk = d_pack_block(bx, by, bz);
  unsigned int sl = d_hash_block(bx, by, bz) & g.block_mask;
  while (true) {
    BlockEntry e = g.blocks[sl];
    if (e.key == bk) {
      if (FRESH && e.pad == 0u) return make_uint2(0u, 0u);
      const unsigned local = (((unsigned)iz & 7u) << 6) | (((unsigned)iy & 7u) << 3) | ((unsigned)ix & 7u);
      return g.cells[(size_t)e.id * kCells + local];
    }
    if (e.key == kEmptyKey) return make_uint2(0u, 0u);
    sl = (sl + 1) & g.block_mask;
  }
}
// the same lookup, also returning the cell's entry index (block id * 512 + local cell; -1: the block is not in the table)
template <bool FRESH = false>
__device__ __forceinline__ uint2 d_cell_range_e(const GridView& g, int ix, int iy, int iz, long long& entry) {
  const int bb = kBias >> kCoarseShift;
  const int bx = (ix >> kCoarseShift) + bb, by = (iy >> kCoarseShift) + bb, bz = (iz >> kCoarseShift) + bb;
  const unsigned long long bk = d_pack_block(bx, by, bz);
  unsigned int sl = d_hash_block(bx, by, bz) & g.block_mask;
  while (true) {
    BlockEntry e = g.blocks[sl];
    if (e.key == bk) {
      if (FRESH && e.pad == 0u) { entry = -1; return make_uint2(0u, 0u); }
      const unsigned local = (((unsigned)iz & 7u) << 6) | (((unsigned)iy & 7u) << 3) | ((unsigned)ix & 7u);
      entry = (long long)e.id * kCells + local;
      return g.cells[(size_t)entry];
    }
    if (e.key == kEmptyKey) { entry = -1; return make_uint2(0u, 0u); }
    sl = (sl + 1) & g.block_mask;
  }
}
// A cell that loses or gains points goes on the work list of the in-place update (see "in-place map update" below): top bit of
// tp[e] = listed, low bits = inserts pending for it.
__device__ __forceinline__ void touch_cell(unsigned int* __restrict__ tp, unsigned int* __restrict__ work, int* __restrict__ ctr, unsigned int work_cap,
                                           unsigned int e) {
  const unsigned int old = atomicOr(&tp[e], 0x80000000u);
  if (!(old & 0x80000000u)) {
    // one atomic on the list counter per wavefront, not per lane (thousands of cells are listed per update)
    const unsigned long long m = __ballot(1);  // the lanes that are here together
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned int base = 0;
    if (lane == leader) base = (unsigned int)atomicAdd(&ctr[kMapCtrWork], __popcll(m));
    base = __shfl(base, leader);
    const unsigned int at = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
    if (at < work_cap) work[at] = e; else ctr[kMapCtrOverflow] = 1;
  }
}
// calc_dist — float32, the reference's evaluation order, no FMA (ikd_Tree.cpp:1273-1277, laserMapping.cpp:152-155)
__device__ __forceinline__ float d_dist2(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
// Thousands of lanes adding to the same counter serialise at ~2.4 ns each (12 us per update for the live-point counter alone):
// the lanes of a wavefront add up first and issue ONE atomic.
// wave_atomic_add_all: every lane of the wavefront is active at the call (butterfly sum).
__device__ __forceinline__ void wave_atomic_add_all(int* ctr, int v) {
  int sum = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if ((threadIdx.x & 63) == 0 && sum != 0) atomicAdd(ctr, sum);
}
// wave_atomic_add_few: any set of active lanes, few contributors (`has`): the active lanes walk the contributors' bits together.
__device__ __forceinline__ void wave_atomic_add_few(unsigned int* ctr, bool has, unsigned int v) {
  const unsigned long long act = __ballot(1);
  unsigned long long m = __ballot(has);
  unsigned int sum = 0;
  while (m) {
    const int l = __ffsll((long long)m) - 1;
    sum += (unsigned int)__shfl((int)v, l);
    m &= m - 1ull;
  }
  if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1 && sum != 0u) atomicAdd(ctr, sum);
}

__device__ __forceinline__ bool d_same_point(float ax, float ay, float az, float bx, float by, float bz) {
  // same_point (ikd_Tree.cpp:1269-1271): fabs(float - float) < EPSS (1e-6, double)
  return (double)fabsf(ax - bx) < 1e-6 && (double)fabsf(ay - by) < 1e-6 && (double)fabsf(az - bz) < 1e-6;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ map_incremental
// cls[i]: 0 = not added, 1 = PointToAdd (with down-sampling), 2 = PointNoNeedDownsample.  world[i] = the world point.
// cls[i] = fa | fn << 1; blk_counts[block] = (#fa, #fn) of the block's 256 points, for the single-pass compaction below.
// guard != nullptr: the launch was enqueued BEHIND the passes of iterated update number `seq`, before the host knew how that
// update ends (lii_scan_job::map_update): the pose is the update's final state (the control block's first 24 doubles), and the
// launch does nothing - empty lists - unless that update has ended regularly (a loop that parked itself is continued by the host,
// which then makes the map update again).
// The two order-preserving compactions happen in the same launch (round 4; a launch of its own before): a workgroup counts its
// two kinds of points, publishes the counts as ONE word (run number << 32 | adds << 16 | no-down-samples) before it waits for
// anything, adds up the words of the workgroups below it - prefix_below, lii_device.h: a word that is overdue is not waited for, its
// block is decided again by the workgroup that needs it - ranks its own points with wavefront ballots and writes both lists; the
// last workgroup leaves the list sizes in counts[0..1].
// The decision for point i (src/laserMapping.cpp:516-553): fa = PointToAdd, fn = PointNoNeedDownsample, wp = its world point.
struct MapDecision {
  unsigned int fa, fn;
  float4 wp;
};
__device__ __forceinline__ MapDecision map_decide_point(const RegistrationBuffers& rb, const PoseArg& ps, double fsd, int have_search, int i, int n, int lo,
                                                        int n_live) {
  MapDecision d;
  d.fa = 0; d.fn = 0;
  d.wp = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < rb.n && i >= lo && i < lo + n_live && i < n) {  // rb.n is the launch bound
    float4 pb = rb.body[i];
    double bx = pb.x, by = pb.y, bz = pb.z;
    double ix = ps.RLI[0] * bx + ps.RLI[1] * by + ps.RLI[2] * bz + ps.TLI[0];
    double iy = ps.RLI[3] * bx + ps.RLI[4] * by + ps.RLI[5] * bz + ps.TLI[1];
    double iz = ps.RLI[6] * bx + ps.RLI[7] * by + ps.RLI[8] * bz + ps.TLI[2];
    const float wx = (float)(ps.R[0] * ix + ps.R[1] * iy + ps.R[2] * iz + ps.p[0]);
    const float wy = (float)(ps.R[3] * ix + ps.R[4] * iy + ps.R[5] * iz + ps.p[1]);
    const float wz = (float)(ps.R[6] * ix + ps.R[7] * iy + ps.R[8] * iz + ps.p[2]);
    d.wp = make_float4(wx, wy, wz, 0.f);
    const int cnt = have_search ? (rb.nbr_count[i] & kCountMask) : 0;
    if (cnt > 0) {
      // mid_point = floor(p / filter_size_map) * filter_size_map + 0.5 * filter_size_map, double math stored to float (:529-534)
      const float mx = (float)(floor(wx / fsd) * fsd + 0.5 * fsd);
      const float my = (float)(floor(wy / fsd) * fsd + 0.5 * fsd);
      const float mz = (float)(floor(wz / fsd) * fsd + 0.5 * fsd);
      const float dist = d_dist2(wx, wy, wz, mx, my, mz);
      const float4 n0 = rb.nbr[i];
      if ((double)fabsf(n0.x - mx) > 0.5 * fsd && (double)fabsf(n0.y - my) > 0.5 * fsd && (double)fabsf(n0.z - mz) > 0.5 * fsd) {
        d.fn = 1;  // PointNoNeedDownsample (:536-541)
      } else {
        bool need_add = true;
        if (cnt >= kMatch) {
#pragma unroll
          for (int k = 0; k < kMatch; k++) {
            const float4 q = rb.nbr[(size_t)k * rb.cap + i];
            if (need_add && d_dist2(q.x, q.y, q.z, mx, my, mz) < dist) need_add = false;
          }
        }
        d.fa = need_add ? 1u : 0u;
      }
    } else {
      d.fa = 1;  // no neighbour list: always added (:551-553)
    }
  }
  return d;
}
struct AddHash {
  unsigned long long* key;   // voxel key, kInvalidKey = free
  unsigned long long* best;  // (float bits of d2 to the voxel centre << 32) | ~batch index; ~0 = none
  unsigned int* slot_of;     // per batch point
  unsigned int mask;         // slots - 1
};
__device__ __forceinline__ unsigned int ah_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xFF51AFD7ED558CCDull; k ^= k >> 33; k *= 0xC4CEB9FE1A85EC53ull; k ^= k >> 33;
  return (unsigned int)k;
}
__device__ __forceinline__ void add_box(const float4 p, float ds, float (&bmin)[3], float (&bmax)[3], float (&mid)[3]) {
  const float cc[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    bmin[a] = floorf(cc[a] / ds) * ds;
    bmax[a] = bmin[a] + ds;
    mid[a] = (float)((double)bmin[a] + (double)(bmax[a] - bmin[a]) / 2.0);
  }
}
// one batch point into the table: its voxel's slot (found or created) and its bid for the voxel's minimum; 0xFFFFFFFF: a non-finite point
__device__ __forceinline__ unsigned int addh_insert_one(const float4 p, unsigned int index, float ds, const AddHash& tb) {
  const int vx = (int)floorf(p.x / ds), vy = (int)floorf(p.y / ds), vz = (int)floorf(p.z / ds);  // (:390-395, float arithmetic)
  const unsigned long long key = ((unsigned long long)(unsigned)(vz + kBias) << 42) | ((unsigned long long)(unsigned)(vy + kBias) << 21) |
                                 (unsigned long long)(unsigned)(vx + kBias);
  float bmin[3], bmax[3], mid[3];
  add_box(p, ds, bmin, bmax, mid);
  const float d = d_dist2(p.x, p.y, p.z, mid[0], mid[1], mid[2]);
  if (!(d == d)) return 0xFFFFFFFFu;  // (a non-finite point takes no part)
  unsigned int slot = ah_hash(key) & tb.mask;
  while (true) {
    const unsigned long long prev = atomicCAS(tb.key + slot, kInvalidKey, key);
    if (prev == kInvalidKey || prev == key) break;
    slot = (slot + 1) & tb.mask;
  }
  atomicMin(tb.best + slot, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(~index));
  return slot;
}
// (adds << 16 | no-down-samples) over the 256 points of a workgroup; every lane must call it
__device__ __forceinline__ unsigned int map_decide_count(unsigned int fa, unsigned int fn, unsigned int* s_a /*[4]*/, unsigned int* s_n /*[4]*/,
                                                         unsigned long long* ma_out, unsigned long long* mn_out) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long ma = __ballot(fa != 0), mn = __ballot(fn != 0);
  if (lane == 0) { s_a[w] = (unsigned int)__popcll(ma); s_n[w] = (unsigned int)__popcll(mn); }
  __syncthreads();
  *ma_out = ma; *mn_out = mn;
  return ((s_a[0] + s_a[1] + s_a[2] + s_a[3]) << 16) | (s_n[0] + s_n[1] + s_n[2] + s_n[3]);
}
__global__ __launch_bounds__(256) void k_map_decide(RegistrationBuffers rb, PoseArg ps_val, double fsd, int have_search,
                                                    unsigned long long* __restrict__ blk_counts, unsigned int epoch, float4* __restrict__ world_out,
                                                    float4* __restrict__ dst_add, float4* __restrict__ dst_nodown, int* __restrict__ counts, int bound_a,
                                                    int bound_n, const IekfCtrl* __restrict__ guard, int seq, int test_late, AddHash tb, float ds,
                                                    unsigned int* __restrict__ ins_flag, int* __restrict__ events) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const PoseArg ps = load_pose(guard != nullptr, reinterpret_cast<const PoseArg*>(guard), ps_val);  // (IekfCtrl::st leads the block)
  const bool go = !guard || (guard->stop == 1 && guard->singular == 0 && guard->seq == seq);  // uniform
  const int n = go ? (rb.n_dev ? *rb.n_dev : rb.n) : 0;
  int lo = 0, n_live = n;  // a rank of a job split by index decides for its block (the lists are exchanged afterwards)
  if (rb.shard_world > 1) shard_range(rb, lo, n_live);
  const MapDecision d = map_decide_point(rb, ps, fsd, have_search, i, n, lo, n_live);
  const unsigned int fa = d.fa, fn = d.fn;
  const float4 wp = d.wp;
  if (i < rb.n && i >= lo && i < lo + n_live && i < n) world_out[i] = wp;
  __shared__ unsigned int s_a[4], s_n[4], s_sum[12], s_ra[4], s_rn[4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long ma, mn;
  const unsigned int word = map_decide_count(fa, fn, s_a, s_n, &ma, &mn);
  const unsigned int tot_a = word >> 16, tot_n = word & 0xFFFFu;
  // the workgroup's word goes out before it looks at anybody else's; a block whose word is overdue is decided here again
  // (prefix_below, lii_device.h: placement-independent - nothing the decision reads changes during this launch)
  const bool hold = test_late != 0 && (blockIdx.x % 7u) == 3u;  // LII_TEST=emit_late, as k_vhash_emit
  if (threadIdx.x == 0 && !hold)
    __hip_atomic_store(blk_counts + blockIdx.x, ((unsigned long long)epoch << 32) | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint2 below = prefix_below(blk_counts, epoch, (int)blockIdx.x, s_sum, test_late != 0, [&](int q) -> unsigned int {
    const MapDecision dq = map_decide_point(rb, ps, fsd, have_search, q * 256 + (int)threadIdx.x, n, lo, n_live);
    unsigned long long xa, xn;
    return map_decide_count(dq.fa, dq.fn, s_ra, s_rn, &xa, &xn);
  });
  if (threadIdx.x == 0 && hold)
    __hip_atomic_store(blk_counts + blockIdx.x, ((unsigned long long)epoch << 32) | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned int base_a = below.x, base_n = below.y;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    const int ca = (int)(base_a + tot_a), cn = (int)(base_n + tot_n);
    counts[0] = ca;
    counts[1] = cn;
    // the update behind this launch may have been enqueued for PREDICTED list sizes (lii_map_incremental): a list that outgrew
    // its bound empties both for that update (counts[3..4] = what it works on) and raises counts[2]; the host repeats it with
    // the exact sizes before the next search
    const int over = (ca > bound_a || cn > bound_n) ? 1 : 0;
    counts[2] = over;
    counts[3] = over ? 0 : ca;
    counts[4] = over ? 0 : cn;
  }
  for (int k = 0; k < w; k++) { base_a += s_a[k]; base_n += s_n[k]; }
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  const unsigned int at_a = base_a + (unsigned int)__popcll(ma & lanes_below);
  if (fa) dst_add[at_a] = wp;
  if (fn) dst_nodown[base_n + (unsigned int)__popcll(mn & lanes_below)] = wp;
  // lii_scan_job::map_update (round 6): the fold's hash insert rides here - every point of the add list knows its place in the list, which
  // is all k_addh_insert took from a launch of its own.  Only places below the bound the fold is enqueued for are inserted (its table is
  // sized by the bound); a list that outgrows the bound leaves a table the repeated update clears first (map_join).
  if (tb.key) {  // (uniform)
    if (i == 0) *events = 0;
    if (i < bound_a) ins_flag[i] = 0u;
    if (fa && at_a < (unsigned int)bound_a) tb.slot_of[at_a] = addh_insert_one(wp, at_a, ds, tb);
  }
}


// ------------------------------------------------------------------------------------------------ Add_Points (down-sampling)
// n may be an upper bound: entries i >= *n_dev get the invalid key and sort to the end.
__global__ void k_add_keys(const float4* __restrict__ pts, int n, const int* __restrict__ n_dev, float ds,
                           unsigned long long* __restrict__ keys, unsigned int* __restrict__ idx, int* __restrict__ events) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *events = 0;  // the fold of this batch counts from zero (it runs behind this launch)
  if (i >= n) return;
  unsigned long long key = kInvalidKey;
  if (!n_dev || i < *n_dev) {
    const float4 p = pts[i];
    // floor(PointToAdd[i].x / downsample_size): float arithmetic (:390-395)
    const int vx = (int)floorf(p.x / ds), vy = (int)floorf(p.y / ds), vz = (int)floorf(p.z / ds);
    key = ((unsigned long long)(unsigned)(vz + kBias) << 42) | ((unsigned long long)(unsigned)(vy + kBias) << 21) |
          (unsigned long long)(unsigned)(vx + kBias);
  }
  keys[i] = key;
  idx[i] = (unsigned)i;
}


// Hash-grouped form of the same fold (lii_map_incremental; round 3).  What Add_Points leaves in a down-sample box does not depend
// on the order of the batch except for ties: with E the existing in-box point closest to the centre (strictly closer than any
// other, first in walk order) and P* the batch point of the voxel closest to the centre (ties: the LAST in batch order - every
// later point replaces the running winner at equal distance, ikd_Tree.cpp:407-413), the box ends up with P* if d(P*) <= d(E) and
// with E otherwise (and with nothing else).  Only the event counter Add_Points returns depends on the order - and
// lii_map_incremental does not report it.  So the voxel groups need not be sorted, not even brought together: every batch point
// finds its voxel's slot in an open-addressing table (CAS on the 63-bit voxel key) and takes part in ONE 64-bit atomicMin per
// slot on (distance bits << 32 | ~index); the point that holds the minimum afterwards is P* and leads the voxel through
// k_add_fold8<true>.  Two launches instead of the key kernel, the batch sort (5 - 7 launches) and the fold.
__global__ void k_addh_insert(const float4* __restrict__ pts, int n, const int* __restrict__ n_dev, float ds, AddHash tb,
                              unsigned int* __restrict__ ins_flag, int* __restrict__ events) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *events = 0;  // (this form does not count Add_Points' events)
  if (i >= n) return;
  ins_flag[i] = 0;
  unsigned int slot = 0xFFFFFFFFu;
  if (!n_dev || i < *n_dev) slot = addh_insert_one(pts[i], (unsigned int)i, ds, tb);
  tb.slot_of[i] = slot;
}

namespace {
__device__ __forceinline__ unsigned int m_hash_block(int bx, int by, int bz) { return d_hash_block(bx, by, bz); }
}  // namespace

// An insert that found no room (block tables full, no slack left and no tail to move the cell to) is kept for the host: the
// next read of the counters rebuilds the index with more room and inserts these points again.
__device__ __forceinline__ void drop_point(float4* __restrict__ dropped, unsigned int drop_cap, int* __restrict__ ctr, const float4 p) {
  const unsigned int at = (unsigned int)atomicAdd(&ctr[kMapCtrDropped], 1);
  if (at < drop_cap) dropped[at] = make_float4(p.x, p.y, p.z, 0.f);
}

// One insert finds its cell entry - creating the 8x8x8 block when the map has never seen it -, bumps the cell's pending count and puts the
// cell on the work list; *ins_e_out = the entry (0xFFFFFFFF: parked for the host's rebuild).
struct InsCells {
  BlockEntry* blocks;
  unsigned int mask;
  float inv_cs;
  unsigned int tables_cap;
  unsigned int* tp;
  unsigned int* work;
  int* ctr;
  unsigned int work_cap;
  float4* dropped;
  unsigned int drop_cap;
  unsigned long long* key_of_id;
};
__device__ __forceinline__ void ins_cell_one(const float4 p, unsigned int* __restrict__ ins_e_out, const InsCells& a) {
  BlockEntry* blocks = a.blocks;
  const unsigned int mask = a.mask, tables_cap = a.tables_cap;
  int* ctr = a.ctr;
  const int ix = (int)floorf(p.x * a.inv_cs), iy = (int)floorf(p.y * a.inv_cs), iz = (int)floorf(p.z * a.inv_cs);
  const int bb = kBias >> kCoarseShift;
  const int bx = (ix >> kCoarseShift) + bb, by = (iy >> kCoarseShift) + bb, bz = (iz >> kCoarseShift) + bb;
  const unsigned long long bk = d_pack_block(bx, by, bz);
  unsigned int sl = m_hash_block(bx, by, bz) & mask;
  long long id = -1;
  // One loop, no waiting inside it: a lane that finds its block's key but not yet its id (another lane - possibly of this very
  // wavefront - is creating the block) goes round the loop again on the SAME slot.  The lanes of a wavefront execute the loop
  // body together, so the creator's stores are issued in the round in which it wins the slot and the waiter sees them in the
  // next one; an inner spin loop would depend on the order in which the compiler lays out the two branches (wavefronts have
  // no independent thread scheduling).
  for (unsigned int probes = 0; probes <= mask;) {
    unsigned long long k = __hip_atomic_load(&blocks[sl].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == kEmptyKey) {
      // A slot of the block table is reserved before it is claimed: the table must keep a free slot, or probes for blocks that
      // are not there (every search does them) would never end.
      const int s_used = atomicAdd(&ctr[kMapCtrSlots], 1);
      if ((unsigned int)s_used >= mask) {
        atomicSub(&ctr[kMapCtrSlots], 1);
        ctr[kMapCtrOverflow] = 1;
        break;  // (id < 0: the point is parked for the host's rebuild)
      }
      const unsigned long long prev = atomicCAS(&blocks[sl].key, kEmptyKey, bk);
      if (prev == kEmptyKey) {  // this lane creates the block: a cell table from the (zeroed) pool
        const int nid = atomicAdd(&ctr[kMapCtrBlocks], 1);
        // The LAST table of the pool is never handed out: it stays all-empty, and a block that finds the pool exhausted points
        // at it - searches see an empty block, lanes waiting for this block's id are released and park their points like this
        // one does (leaving `pad` at 0 would have them wait forever; the host rebuilds with more room).
        const bool got_table = (unsigned int)nid + 1u < tables_cap;
        if (!got_table) ctr[kMapCtrOverflow] = 1;
        if (got_table && a.key_of_id) a.key_of_id[nid] = bk;  // (WinKeep; read by the launches behind this one)
        __hip_atomic_store(&blocks[sl].id, got_table ? (unsigned int)nid : tables_cap - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&blocks[sl].pad, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        id = got_table ? nid : -1;
        break;
      }
      atomicSub(&ctr[kMapCtrSlots], 1);  // somebody else took the slot
      k = prev;
    }
    if (k == bk) {
      if (__hip_atomic_load(&blocks[sl].pad, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        const unsigned int got = __hip_atomic_load(&blocks[sl].id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        id = got + 1u == tables_cap ? -1 : (long long)got;  // (the shared empty table: its creator found the pool exhausted)
        break;
      }
      __builtin_amdgcn_s_sleep(1);  // the id is on its way: look at this slot again
      continue;
    }
    sl = (sl + 1) & mask;
    probes++;
  }
  if (id < 0) {  // no table left for a new block: the point waits in the dropped list for the host's rebuild (nothing is lost)
    *ins_e_out = 0xFFFFFFFFu;
    ctr[kMapCtrOverflow] = 1;
    drop_point(a.dropped, a.drop_cap, ctr, p);
    return;
  }
  const unsigned int e = (unsigned int)id * kCells + ((((unsigned)iz & 7u) << 6) | (((unsigned)iy & 7u) << 3) | ((unsigned)ix & 7u));
  *ins_e_out = e;
  touch_cell(a.tp, a.work, ctr, a.work_cap, e);
  atomicAdd(&a.tp[e], 1u);
}

// Add_Points with down-sampling, one voxel group (the batch points of one down-sample box, adjacent after the sort) per EIGHT
// lanes.  tomb[j] = 1 marks existing map point j as deleted; ins_flag[i] = 1 / ins_pts[i] = the point to insert, for the group
// starting at sorted position i; *events accumulates the reference's tmp_counter (ikd_Tree.cpp:381-426).
// The down-sample box (edge ds) spans at most 2 x 2 x 2 grid cells (cell edge >= ds: the usual 3 ds), one per lane, so the
// existing in-box points are found with ONE dependent lookup chain per group instead of eight in sequence; lane 0 then replays the
// batch points of the voxel in batch order - the reference's sequential "keep the point closest to the centre" rule: existing
// count, strict '<', same_point 1e-6 - and the lanes mark the tombstones of their own cells (putting the cell on the work list
// of the in-place update).  Groups whose box spans more cells (a caller-chosen cell edge below ds) are walked by lane 0 alone,
// cells in z-outer / x-inner order.  The lanes are numbered in that order too, ties of the squared distance go to the lower
// lane / lower index: both forms visit the existing points in the same order.
// HASHED: the groups come out of the table of k_addh_insert instead of the sorted keys - the batch point that holds its voxel's
// minimum leads (ins_flag / ins_pts at its own batch index), the replay is the comparison of that point with the existing one.
// CELLS (round 6; HASHED only): what k_ins_cells did in a launch of its own behind the fold rides here - a leader whose batch point stays
// finds / creates that point's cell at once (fc.ins_e[i]; every other position of the list gets 0xFFFFFFFF), and the workgroups behind the
// fold's own (blockIdx >= fc.fold_blocks) do the same for the second insert list, which does not pass through the fold.  Cell lookups of
// such a launch run beside block creations: d_cell_range<FRESH>.
struct FoldCells {
  InsCells a;
  unsigned int* ins_e;     // per position of the folded list
  const int* n_dev;        // the folded list holds *n_dev points (n is the launch bound); may be nullptr
  const float4* list2;     // the second list: n2 points (or *n2_dev), all of them inserts, entries to ins_e2
  int n2;
  const int* n2_dev;
  unsigned int* ins_e2;
  unsigned int fold_blocks;
};
template <bool HASHED, bool CELLS>
__global__ __launch_bounds__(256) void k_add_fold8(const float4* __restrict__ add_pts, const unsigned long long* __restrict__ keys,
                                                   const unsigned int* __restrict__ idx, AddHash tb, int n, float ds, GridView g,
                                                   unsigned char* __restrict__ tomb, float4* __restrict__ ins_pts,
                                                   unsigned int* __restrict__ ins_flag, unsigned int* __restrict__ events,
                                                   unsigned int* __restrict__ tp, unsigned int* __restrict__ work, int* __restrict__ ctr,
                                                   unsigned int work_cap, FoldCells fc) {
  if (CELLS && blockIdx.x >= fc.fold_blocks) {  // (uniform per workgroup) the second insert list
    const int j = (int)((blockIdx.x - fc.fold_blocks) * blockDim.x + threadIdx.x);
    if (j < fc.n2 && !(fc.n2_dev && j >= *fc.n2_dev)) ins_cell_one(fc.list2[j], &fc.ins_e2[j], fc.a);
    return;
  }
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int c = threadIdx.x & 7;
  const bool in_bound = i < n;  // (n is a launch bound when the list's size is on the device: fc.n_dev)
  const bool in_range = in_bound && !(HASHED && fc.n_dev && i >= *fc.n_dev);
  unsigned long long key = kInvalidKey;
  unsigned int slot = 0xFFFFFFFFu;
  bool leader_pos;  // uniform over the 8 lanes
  if (HASHED) {
    slot = in_range ? tb.slot_of[i] : 0xFFFFFFFFu;
    // (a slot its leader has already cleared reads all ones: not a minimum - distances are finite)
    const unsigned long long bs = slot != 0xFFFFFFFFu ? tb.best[slot] : ~0ull;
    leader_pos = bs != ~0ull && (unsigned int)bs == ~(unsigned int)i;
  } else {
    if (in_range && c == 0) ins_flag[i] = 0;
    key = in_range ? keys[i] : kInvalidKey;
    leader_pos = in_range && key != kInvalidKey && !(i > 0 && keys[i - 1] == key);
  }
  if (!leader_pos) {
    if (CELLS && in_bound && c == 0) fc.ins_e[i] = 0xFFFFFFFFu;  // (k_ins_write looks at every position below the bound)
    return;
  }
  const float4 p0 = add_pts[HASHED ? (unsigned int)i : idx[i]];
  float bmin[3], bmax[3], mid[3];
  add_box(p0, ds, bmin, bmax, mid);
  int c0[3], c1[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float slack = 1e-6f * (fabsf(bmin[a]) + fabsf(bmax[a]) + 8.f);
    c0[a] = (int)floorf((bmin[a] - slack) * g.inv_cs);
    c1[a] = (int)floorf((bmax[a] + slack) * g.inv_cs);
  }
  const bool wide = (c1[0] - c0[0] > 1) || (c1[1] - c0[1] > 1) || (c1[2] - c0[2] > 1);  // uniform over the 8 lanes
  const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
  const bool mine = !wide && c0[0] + dx <= c1[0] && c0[1] + dy <= c1[1] && c0[2] + dz <= c1[2];
  // the first eight batch points of the group, one per lane (the replay below reads them by shuffle instead of walking
  // keys[t] -> idx[t] -> add_pts[...] one dependent pair of loads per point); the loads overlap the grid lookups
  const int tt = i + c;
  bool mv = false;
  float4 mp = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!HASHED) {
    mv = tt < n && keys[tt] == key;
    if (mv) mp = add_pts[idx[tt]];
  }
  int n0 = 0, best = -1;
  float bestd = __builtin_inff();
  uint2 r = make_uint2(0u, 0u);
  long long my_entry = -1;
  if (g.n_pts > 0) {
    if (mine) {
      r = d_cell_range_e<CELLS>(g, c0[0] + dx, c0[1] + dy, c0[2] + dz, my_entry);
      for (unsigned int j0 = r.x; j0 < r.y; j0 += 4u) {  // four candidates per trip, loaded together; visited in index order
        float4 q4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) q4[u] = g.pts[min(j0 + (unsigned)u, r.y - 1u)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float4 q = q4[u];
          if (j0 + (unsigned)u < r.y && bmin[0] <= q.x && bmax[0] > q.x && bmin[1] <= q.y && bmax[1] > q.y && bmin[2] <= q.z && bmax[2] > q.z) {
            n0++;
            const float d = d_dist2(q.x, q.y, q.z, mid[0], mid[1], mid[2]);
            if (d < bestd) { bestd = d; best = (int)(j0 + (unsigned)u); }
          }
        }
      }
    } else if (wide && c == 0) {  // the sequential walk (box wider than 2 x 2 x 2 cells)
      for (int cz = c0[2]; cz <= c1[2]; cz++)
        for (int cy = c0[1]; cy <= c1[1]; cy++)
          for (int cx = c0[0]; cx <= c1[0]; cx++) {
            const uint2 rr = d_cell_range<CELLS>(g, cx, cy, cz);
            for (unsigned int j = rr.x; j < rr.y; j++) {
              const float4 q = g.pts[j];
              if (bmin[0] <= q.x && bmax[0] > q.x && bmin[1] <= q.y && bmax[1] > q.y && bmin[2] <= q.z && bmax[2] > q.z) {
                n0++;
                const float d = d_dist2(q.x, q.y, q.z, mid[0], mid[1], mid[2]);
                if (d < bestd) { bestd = d; best = (int)j; }
              }
            }
          }
    }
  }
  // the group's count and its first minimum in walk order (lower lane wins a tie: the lanes are the walk's cell order)
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    const int on = __shfl_xor(n0, off), ob = __shfl_xor(best, off);
    const float od = __shfl_xor(bestd, off);
    const int other_lane = c ^ off;
    n0 += on;
    const bool take = ob >= 0 && (best < 0 || od < bestd || (od == bestd && other_lane < c));
    // (after a step both partners hold the same winner; "other_lane < c" orders the ORIGINAL owners only in the first step,
    // later steps compare sub-group winners whose owner order is the order of the sub-groups: the lower sub-group holds lower lanes)
    if (take) { best = ob; bestd = od; }
  }
  // Sequential fold over the points of this voxel, in batch order - run by all eight lanes alike (same inputs, same result in
  // every lane): the first eight points come out of the lanes' registers by shuffle, a longer group goes on from memory.
  bool ev = false, cur_new = false;
  int cur_old = -1;
  if (HASHED) {
    // the leader IS the batch point that stays if a batch point does; E stays if it is strictly closer (and then every other
    // existing in-box point goes); a lone E that stays leaves the voxel untouched
    const float dp = d_dist2(p0.x, p0.y, p0.z, mid[0], mid[1], mid[2]);
    const bool old_wins = (n0 > 0) && (bestd < dp);
    ev = !old_wins || n0 > 1;
    cur_new = !old_wins;
    cur_old = old_wins ? best : -1;
    if (c == 0) {
      if (cur_new) { ins_pts[i] = make_float4(p0.x, p0.y, p0.z, 0.f); ins_flag[i] = 1; }
      tb.key[slot] = kInvalidKey;  // the slot is free again for the next batch
      tb.best[slot] = ~0ull;
    }
  } else {
    const int lead = (threadIdx.x & 63) & ~7;
    const unsigned int gm = (unsigned int)((__ballot(mv) >> lead) & 0xFFull);  // the group is contiguous: a prefix of the 8 lanes
    const int m8 = __popc(gm);
    float4 qb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 > 0 && best >= 0) qb = g.pts[best];  // the existing point a batch point may lose against
    float cx_ = 0, cy_ = 0, cz_ = 0, cd = 0;
    unsigned int n_events = 0;
    auto replay = [&](const float4 p) {
      const float dp = d_dist2(p.x, p.y, p.z, mid[0], mid[1], mid[2]);
      if (!ev) {
        const bool old_wins = (n0 > 0) && (bestd < dp);
        float rx = p.x, ry = p.y, rz = p.z;
        if (old_wins) { rx = qb.x; ry = qb.y; rz = qb.z; }
        if (n0 > 1 || d_same_point(p.x, p.y, p.z, rx, ry, rz)) {
          ev = true;
          n_events++;
          cur_new = !old_wins;
          cur_old = old_wins ? best : -1;
          cx_ = rx; cy_ = ry; cz_ = rz;
          cd = old_wins ? bestd : dp;
        }
      } else {
        const bool cur_wins = cd < dp;
        const float rx = cur_wins ? cx_ : p.x, ry = cur_wins ? cy_ : p.y, rz = cur_wins ? cz_ : p.z;
        if (d_same_point(p.x, p.y, p.z, rx, ry, rz)) {
          n_events++;
          if (!cur_wins) { cur_new = true; cur_old = -1; cx_ = p.x; cy_ = p.y; cz_ = p.z; cd = dp; }
        }
      }
    };
    for (int t = 0; t < m8; t++) {
      float4 p;
      p.x = __shfl(mp.x, lead + t); p.y = __shfl(mp.y, lead + t); p.z = __shfl(mp.z, lead + t); p.w = 0.f;
      replay(p);
    }
    if (m8 == 8)
      for (int t = i + 8; t < n && keys[t] == key; t++) replay(add_pts[idx[t]]);
    wave_atomic_add_few(events, c == 0 && ev, n_events);  // (the groups of this wavefront that are still here: all leaders)
    if (c == 0 && ev) {
      if (cur_new) {
        ins_pts[i] = make_float4(cx_, cy_, cz_, 0.f);
        ins_flag[i] = 1;
      }
    }
  }
  if (ev) {
  // delete every existing in-box point except a surviving one
  // (the lane whose cell holds a tombstoned point also puts that cell on the work list of the in-place update)
  if (n0 == 1) {
    if (!wide) {
      if (mine && best != cur_old && (unsigned)best >= r.x && (unsigned)best < r.y) {
        tomb[best] = 1;
        touch_cell(tp, work, ctr, work_cap, (unsigned int)my_entry);
      }
    } else if (c == 0 && best != cur_old) {
      tomb[best] = 1;
      const float4 q = g.pts[best];
      long long e;
      (void)d_cell_range_e<CELLS>(g, (int)floorf(q.x * g.inv_cs), (int)floorf(q.y * g.inv_cs), (int)floorf(q.z * g.inv_cs), e);
      if (e >= 0) touch_cell(tp, work, ctr, work_cap, (unsigned int)e);
    }
  } else if (n0 > 1) {
    if (mine) {
      bool any = false;
      for (unsigned int j = r.x; j < r.y; j++) {
        const float4 q = g.pts[j];
        if ((int)j != cur_old && bmin[0] <= q.x && bmax[0] > q.x && bmin[1] <= q.y && bmax[1] > q.y && bmin[2] <= q.z && bmax[2] > q.z) { tomb[j] = 1; any = true; }
      }
      if (any) touch_cell(tp, work, ctr, work_cap, (unsigned int)my_entry);
    } else if (wide && c == 0) {
      for (int cz = c0[2]; cz <= c1[2]; cz++)
        for (int cy = c0[1]; cy <= c1[1]; cy++)
          for (int cx = c0[0]; cx <= c1[0]; cx++) {
            long long e;
            const uint2 rr2 = d_cell_range_e<CELLS>(g, cx, cy, cz, e);
            bool any = false;
            for (unsigned int j = rr2.x; j < rr2.y; j++) {
              const float4 q = g.pts[j];
              if ((int)j != cur_old && bmin[0] <= q.x && bmax[0] > q.x && bmin[1] <= q.y && bmax[1] > q.y && bmin[2] <= q.z && bmax[2] > q.z) {
                tomb[j] = 1;
                any = true;
              }
            }
            if (any && e >= 0) touch_cell(tp, work, ctr, work_cap, (unsigned int)e);
          }
    }
  }
  }
  // (behind the tombstones: the leader's atomics are not in the way of the other seven lanes' marks)
  if (CELLS) {
    // The cell of the batch point that stays is one of the (up to) eight cells the group's lanes have just looked up: its entry comes out
    // of that lane's register - no probe of the block table, the chain k_ins_cells spent its launch on.  Only a point whose block does
    // not exist yet (or is being created beside this lookup), or a box wider than 2 x 2 x 2 cells, goes through the find-or-create loop.
    const int ix = (int)floorf(p0.x * g.inv_cs) - c0[0], iy = (int)floorf(p0.y * g.inv_cs) - c0[1], iz = (int)floorf(p0.z * g.inv_cs) - c0[2];
    const bool among = !wide && (unsigned)ix < 2u && (unsigned)iy < 2u && (unsigned)iz < 2u;
    const int e_lane = __shfl((int)my_entry, (among ? (ix | (iy << 1) | (iz << 2)) : 0), 8);  // (all eight lanes of a leader's group are here)
    if (c == 0) {
      if (!cur_new) {
        fc.ins_e[i] = 0xFFFFFFFFu;
      } else if (among && e_lane >= 0) {
        fc.ins_e[i] = (unsigned int)e_lane;
        touch_cell(tp, work, ctr, work_cap, (unsigned int)e_lane);
        atomicAdd(&tp[e_lane], 1u);
      } else {
        ins_cell_one(make_float4(p0.x, p0.y, p0.z, 0.f), &fc.ins_e[i], fc.a);
      }
    }
  }
}



// ================================================================================================ in-place map update
// Round 2: the map is no longer re-sorted / re-indexed on every update.  The point array keeps SLACK behind every cell
// (capacity end in cell_cap[]); an update touches only the cells it changes:
//   k_touch_tombs   the cells that own a tombstoned slot (set by the fold or by a box deletion) go on the work list
//   k_ins_cells     every insert finds its cell entry - creating the 8x8x8 block (hash insert + a zeroed cell table from the
//                   pool) when the map has never seen it - bumps the cell's pending count and puts the cell on the work list
//   k_cell_apply    one lane per listed cell: tombstoned points are squeezed out in place, and when the survivors + the pending
//                   inserts exceed the capacity the cell moves to the tail of the array with fresh slack
//   k_ins_write     every insert claims a slot behind its cell's end
// O(batch) per update instead of six passes over the whole map.  The order of the points inside a cell (and of relocated cells
// in the array) depends on the order the atomics resolve: the map is the same SET on every run and every rank, its array
// order is not.  A rebuild (gather -> sort -> index -> spread, lii_capi.cpp) restores the cell-sorted order when the tail
// fills up or the block table gets crowded.


// flags == nullptr: every point of the list is an insert.  n_dev != nullptr: the list holds *n_dev points (n is the launch bound).
// A second list (list2, n2 points, all of them inserts, entries to ins_e2) rides in the same launch: lanes [n, n + n2).
__global__ void k_ins_cells(const float4* __restrict__ list, const unsigned int* __restrict__ flags, int n, const int* __restrict__ n_dev,
                            const float4* __restrict__ list2, int n2, const int* __restrict__ n2_dev, unsigned int* __restrict__ ins_e2,
                            unsigned int* __restrict__ ins_e, InsCells a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) {
    i -= n;
    if (i >= n2 || (n2_dev && i >= *n2_dev)) return;
    list = list2; flags = nullptr; ins_e = ins_e2;
  } else if (n_dev && i >= *n_dev) {
    return;
  }
  if (flags && !flags[i]) { ins_e[i] = 0xFFFFFFFFu; return; }
  ins_cell_one(list[i], &ins_e[i], a);
}

__global__ void k_cell_apply(const unsigned int* __restrict__ work, uint2* __restrict__ cells, unsigned int* __restrict__ cell_cap,
                             float4* __restrict__ pts, unsigned char* __restrict__ tomb, unsigned int* __restrict__ tp, int* __restrict__ ctr,
                             unsigned int pts_cap, int launch_bound, WinKeep wk) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_work = ctr[kMapCtrWork];
  const bool valid = w < n_work && w < launch_bound;
  if (!__any(valid)) return;  // (uniform per wavefront)
  int gone = 0;
  if (valid) {
  const unsigned int e = work[w];
  const uint2 c = cells[e];
  const unsigned int tpv = tp[e], capv = cell_cap[e];  // (loaded beside the cell entry, not behind the squeeze)
  unsigned int first = c.x, end = c.y, wpos = c.x;
  // squeeze the tombstones out, eight slots at a time: the flags and the points of a chunk are loaded together (the loop was a
  // chain of ~20 dependent loads per cell: 15 us per update)
  for (unsigned int j0 = first; j0 < end; j0 += 8u) {
    unsigned char t[8];
    float4 p[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const unsigned int jj = min(j0 + (unsigned)u, end - 1u);
      t[u] = tomb[jj];
      p[u] = pts[jj];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const unsigned int j = j0 + (unsigned)u;
      if (j < end) {
        if (!t[u]) {
          if (wpos != j) pts[wpos] = p[u];  // wpos <= j: a slot of this chunk already read, or of an earlier one
          wpos++;
        } else {
          tomb[j] = 0;
        }
      }
    }
  }
  const unsigned int deleted = end - wpos, alive = wpos - first;
  const unsigned int pending = tpv & 0x7FFFFFFFu;
  tp[e] = 0u;
  if (first + alive + pending > capv) {  // (an empty, never used cell has first = end = cap = 0)
    const unsigned int need = alive + pending, newcap = need + max(2u, need >> 2);
    const unsigned int nf = (unsigned int)atomicAdd(&ctr[kMapCtrUsed], (int)newcap);
    if (nf + newcap > pts_cap) {
      ctr[kMapCtrOverflow] = 1;  // the inserts of this cell are parked by k_ins_write (cap stays): the host rebuilds and re-inserts
    } else {
      for (unsigned int j0 = 0; j0 < alive; j0 += 8u) {
        float4 p[8];
#pragma unroll
        for (int u = 0; u < 8; u++) p[u] = pts[first + min(j0 + (unsigned)u, alive - 1u)];
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (j0 + (unsigned)u < alive) pts[nf + j0 + (unsigned)u] = p[u];
      }
      first = nf;
      cell_cap[e] = nf + newcap;
    }
  }
  cells[e] = make_uint2(first, first + alive);
  if (wk.win) {  // (uniform) the window's copy of the entry
    const long long wi = win_index_of_entry(wk, e);
    if (wi >= 0) wk.win[wi] = make_uint2(first, first + alive);
    else __hip_atomic_store(&ctr[kMapCtrWinStale], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  gone = (int)deleted;
  }
  wave_atomic_add_all(&ctr[kMapCtrValid], -gone);
}

__global__ void k_ins_write(const float4* __restrict__ list, const unsigned int* __restrict__ ins_e, int n, const int* __restrict__ n_dev,
                            const float4* __restrict__ list2, const unsigned int* __restrict__ ins_e2, int n2, const int* __restrict__ n2_dev,
                            uint2* __restrict__ cells, const unsigned int* __restrict__ cell_cap, float4* __restrict__ pts, int* __restrict__ ctr,
                            float4* __restrict__ dropped, unsigned int drop_cap, int* __restrict__ host, int n_words, int seq_at, int seq, WinKeep wk) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  // (the counters this launch writes are read again by its LAST workgroup, on whichever XCD that runs: atomic stores, not plain ones)
  if (i == 0) __hip_atomic_store(&ctr[kMapCtrWork], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the work list has been consumed (k_cell_apply ran before this launch)
  bool valid = true;
  if (i >= n) {
    i -= n;
    valid = i < n2 && !(n2_dev && i >= *n2_dev);
    list = list2; ins_e = ins_e2;
  } else if (n_dev && i >= *n_dev) {
    valid = false;
  }
  int added = 0;
  if (valid) {
    const unsigned int e = ins_e[i];
    if (e != 0xFFFFFFFFu) {
      const unsigned int slot = atomicAdd(reinterpret_cast<unsigned int*>(&cells[e]) + 1, 1u);
      const float4 p = list[i];
      if (slot < cell_cap[e]) {
        pts[slot] = make_float4(p.x, p.y, p.z, 0.f);
        added = 1;
        if (wk.win) {  // (uniform) the window counts the insert as the cell table did (k_cell_apply left the two entries equal)
          const long long wi = win_index_of_entry(wk, e);
          if (wi >= 0) atomicAdd(reinterpret_cast<unsigned int*>(&wk.win[wi]) + 1, 1u);
        }
      } else {
        atomicSub(reinterpret_cast<unsigned int*>(&cells[e]) + 1, 1u);
        __hip_atomic_store(&ctr[kMapCtrOverflow], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        drop_point(dropped, drop_cap, ctr, p);
      }
    }
  }
  wave_atomic_add_all(&ctr[kMapCtrValid], added);  // every lane of the wavefront arrives here
  if (!host) return;  // (uniform)
  // The update ends here: the workgroup that finishes last (a ticket) hands the map's counters - and the list sizes behind them - to
  // the host's pinned, mapped buffer and puts the number of the update behind them; the host polls that word (an event behind a
  // copy woke it ~ 15 us late, and the copy was a launch of its own: the next scan's search is waiting for exactly this).
  wait_published_atomics();
  __syncthreads();
  __shared__ int s_last;
  if (threadIdx.x == 0) s_last = atomicAdd(reinterpret_cast<unsigned int*>(&ctr[kMapCtrTicket]), 1u) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  if ((int)threadIdx.x < n_words) {
    int v = __hip_atomic_load(&ctr[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == kMapCtrTicket) { v = 0; __hip_atomic_store(&ctr[kMapCtrTicket], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __hip_atomic_store(host + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // (the words above are system-scope stores into host memory: every lane waits for its own, the barrier collects them, a relaxed store
  // raises the number behind them - no system-scope release fence, which would write the whole L2 back first: lii_iekf.hip, publish_done)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host + seq_at, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- (re)build: slack layout from the compact, cell-sorted array
__global__ void k_cell_caps(const uint2* __restrict__ cells, int n_entries, unsigned int* __restrict__ caps) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  const unsigned int cnt = cells[e].y - cells[e].x;
  caps[e] = cnt ? cnt + max(2u, cnt >> 2) : 0u;
}
// capsum = inclusive scan of caps.  Points are copied cell by cell (one lane per cell: ~9 points), then the tables are rewritten.
__global__ void k_spread(const float4* __restrict__ src, uint2* __restrict__ cells, unsigned int* __restrict__ cell_cap,
                         const unsigned int* __restrict__ caps, const unsigned int* __restrict__ capsum, int n_entries, float4* __restrict__ dst,
                         int* __restrict__ ctr, int n_valid, int n_blocks) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  if (e == n_entries - 1) {
    ctr[kMapCtrUsed] = (int)capsum[e];
    ctr[kMapCtrValid] = n_valid;
    ctr[kMapCtrBlocks] = n_blocks;
    ctr[kMapCtrSlots] = n_blocks;  // occupied slots of the block table
    ctr[kMapCtrWork] = 0;
    ctr[kMapCtrOverflow] = 0;
  }
  const uint2 c = cells[e];
  const unsigned int cnt = c.y - c.x, nf = capsum[e] - caps[e];
  for (unsigned int j = 0; j < cnt; j++) dst[nf + j] = src[c.x + j];
  cells[e] = cnt ? make_uint2(nf, nf + cnt) : make_uint2(0u, 0u);
  cell_cap[e] = cnt ? nf + caps[e] : 0u;
}
// ---- gather the live points (download, rebuild): cnt -> inclusive scan -> copy
__global__ void k_cell_counts(const uint2* __restrict__ cells, int n_entries, unsigned int* __restrict__ cnt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_entries) cnt[e] = cells[e].y - cells[e].x;
}
__global__ void k_gather_live(const float4* __restrict__ pts, const uint2* __restrict__ cells, const unsigned int* __restrict__ cntsum, int n_entries,
                              float4* __restrict__ dst, int dst_cap) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  const uint2 c = cells[e];
  const unsigned int cnt = c.y - c.x, at = cntsum[e] - cnt;
  for (unsigned int j = 0; j < cnt; j++)
    if ((int)(at + j) < dst_cap) dst[at + j] = pts[c.x + j];
}
// KD_TREE::Delete_Point_Boxes on the slack layout: one lane per cell entry walks its live points
__global__ void k_box_tomb_cells(const float4* __restrict__ pts, const uint2* __restrict__ cells, int n_entries, const float* __restrict__ boxes,
                                 int n_boxes, unsigned char* __restrict__ tomb, unsigned int* __restrict__ tp, unsigned int* __restrict__ work,
                                 int* __restrict__ ctr, unsigned int work_cap) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  const uint2 c = cells[e];
  bool any = false;
  for (unsigned int j = c.x; j < c.y; j++) {
    const float4 p = pts[j];
    bool dead = false;
    for (int b = 0; b < n_boxes; b++) {
      const float* q = boxes + 6 * b;
      dead = dead || (q[0] <= p.x && q[3] > p.x && q[1] <= p.y && q[4] > p.y && q[2] <= p.z && q[5] > p.z);
    }
    if (dead) { tomb[j] = 1; any = true; }
  }
  if (any) touch_cell(tp, work, ctr, work_cap, (unsigned int)e);
}

static inline int nblk(int n, int b) { return (n + b - 1) / b; }
void launch_ins_cells(const float4* list, const unsigned int* flags, int n, const int* n_dev, const float4* list2, int n2, const int* n2_dev, unsigned int* ins_e2,
                      BlockEntry* blocks, unsigned int mask, float inv_cs, unsigned int tables_cap, unsigned int* ins_e, unsigned int* tp,
                      unsigned int* work, int* ctr, unsigned int work_cap, float4* dropped, unsigned int drop_cap, unsigned long long* key_of_id, hipStream_t s) {
  if (n < 0) n = 0;
  if (n2 < 0) n2 = 0;
  InsCells a;
  a.blocks = blocks; a.mask = mask; a.inv_cs = inv_cs; a.tables_cap = tables_cap; a.tp = tp; a.work = work; a.ctr = ctr; a.work_cap = work_cap;
  a.dropped = dropped; a.drop_cap = drop_cap; a.key_of_id = key_of_id;
  if (n + n2 > 0)
    hipLaunchKernelGGL(k_ins_cells, dim3(nblk(n + n2, 256)), dim3(256), 0, s, list, flags, n, n_dev, list2, n2, n2_dev, ins_e2, ins_e, a);
}
void launch_cell_apply(const unsigned int* work, uint2* cells, unsigned int* cell_cap, float4* pts, unsigned char* tomb, unsigned int* tp, int* ctr,
                       unsigned int pts_cap, int launch_bound, const WinKeep& wk, hipStream_t s) {
  if (launch_bound > 0) hipLaunchKernelGGL(k_cell_apply, dim3(nblk(launch_bound, 128)), dim3(128), 0, s, work, cells, cell_cap, pts, tomb, tp, ctr, pts_cap, launch_bound, wk);
}
void launch_ins_write(const float4* list, const unsigned int* ins_e, int n, const int* n_dev, const float4* list2, const unsigned int* ins_e2, int n2, const int* n2_dev,
                      uint2* cells, const unsigned int* cell_cap, float4* pts, int* ctr, float4* dropped, unsigned int drop_cap, hipStream_t s,
                      int* host, int n_words, int seq_at, int seq, const WinKeep& wk) {
  if (n < 0) n = 0;
  if (n2 < 0) n2 = 0;
  hipLaunchKernelGGL(k_ins_write, dim3(nblk(n + n2 > 0 ? n + n2 : 1, 256)), dim3(256), 0, s, list, ins_e, n, n_dev, list2, ins_e2, n2, n2_dev, cells, cell_cap, pts,
                     ctr, dropped, drop_cap, host, n_words, seq_at, seq, wk);
}
void launch_cell_caps(const uint2* cells, int n_entries, unsigned int* caps, hipStream_t s) {
  if (n_entries > 0) hipLaunchKernelGGL(k_cell_caps, dim3(nblk(n_entries, 256)), dim3(256), 0, s, cells, n_entries, caps);
}
void launch_spread(const float4* src, uint2* cells, unsigned int* cell_cap, const unsigned int* caps, const unsigned int* capsum, int n_entries,
                   float4* dst, int* ctr, int n_valid, int n_blocks, hipStream_t s) {
  if (n_entries > 0) hipLaunchKernelGGL(k_spread, dim3(nblk(n_entries, 256)), dim3(256), 0, s, src, cells, cell_cap, caps, capsum, n_entries, dst, ctr, n_valid, n_blocks);
}
void launch_cell_counts(const uint2* cells, int n_entries, unsigned int* cnt, hipStream_t s) {
  if (n_entries > 0) hipLaunchKernelGGL(k_cell_counts, dim3(nblk(n_entries, 256)), dim3(256), 0, s, cells, n_entries, cnt);
}
void launch_gather_live(const float4* pts, const uint2* cells, const unsigned int* cntsum, int n_entries, float4* dst, int dst_cap, hipStream_t s) {
  if (n_entries > 0) hipLaunchKernelGGL(k_gather_live, dim3(nblk(n_entries, 256)), dim3(256), 0, s, pts, cells, cntsum, n_entries, dst, dst_cap);
}
void launch_box_tomb_cells(const float4* pts, const uint2* cells, int n_entries, const float* boxes, int n_boxes, unsigned char* tomb, unsigned int* tp,
                           unsigned int* work, int* ctr, unsigned int work_cap, hipStream_t s) {
  if (n_entries > 0) hipLaunchKernelGGL(k_box_tomb_cells, dim3(nblk(n_entries, 256)), dim3(256), 0, s, pts, cells, n_entries, boxes, n_boxes, tomb, tp, work, ctr, work_cap);
}

// The last launch of an in-place update: the map's counters (and the list sizes behind them) go to the caller's pinned, mapped buffer
// with plain system-scope stores, the number of the update behind them - the host polls that word instead of waiting for an event
// behind a copy (an event wait wakes the host ~ 15 us after the fact; the search of the next scan is waiting for exactly this).
__global__ __launch_bounds__(64) void k_map_publish(const int* __restrict__ ctr, int n_words, int* __restrict__ host, int seq_at, int seq) {
  const int l = threadIdx.x;
  if (l < n_words) __hip_atomic_store(host + l, ctr[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (one wavefront: its system-scope stores are out before the number goes: see k_ins_write)
  if (l == 0) __hip_atomic_store(host + seq_at, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_map_publish(const int* ctr, int n_words, int* host, int seq_at, int seq, hipStream_t s) {
  hipLaunchKernelGGL(k_map_publish, dim3(1), dim3(64), 0, s, ctr, n_words, host, seq_at, seq);
}
size_t add_hash_slots(int max_n) {
  size_t slots = 1024;
  while (slots < 4u * (size_t)max_n) slots <<= 1;
  return slots;
}
void launch_map_decide_compact(const RegistrationBuffers& rb, const PoseArg& ps, double fsd, int have_search, unsigned long long* blk_counts, unsigned int epoch,
                               float4* world, float4* dst_add, float4* dst_nodown, int* counts, int bound_add, int bound_nodown, hipStream_t s,
                               const IekfCtrl* guard, int seq, int test_late, unsigned long long* hkey, unsigned long long* hbest, unsigned int* slot_of,
                               float ds, unsigned int* ins_flag, int* events) {
  if (rb.n <= 0) return;
  AddHash tb;  // hkey != nullptr: the hash insert of the fold behind this launch rides in it (table sized by bound_add, as launch_add_fold_hashed sizes it)
  tb.key = hkey; tb.best = hbest; tb.slot_of = slot_of;
  tb.mask = hkey ? (unsigned int)(add_hash_slots(bound_add) - 1) : 0u;
  hipLaunchKernelGGL(k_map_decide, dim3(nblk(rb.n, 256)), dim3(256), 0, s, rb, ps, fsd, have_search, blk_counts, epoch, world, dst_add, dst_nodown, counts,
                     bound_add, bound_nodown, guard, seq, test_late, tb, ds, ins_flag, events);
}
void launch_add_keys(const float4* pts, int n, const int* n_dev, float ds, unsigned long long* keys, unsigned int* idx, int* events, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_add_keys, dim3(nblk(n, 256)), dim3(256), 0, s, pts, n, n_dev, ds, keys, idx, events);
}
void launch_add_fold(const float4* add_pts, const unsigned long long* keys, const unsigned int* idx, int n, float ds, const GridView& g,
                     unsigned char* tomb, float4* ins_pts, unsigned int* ins_flag, unsigned int* events, unsigned int* tp, unsigned int* work,
                     int* ctr, unsigned int work_cap, hipStream_t s) {
  AddHash none;
  none.key = nullptr; none.best = nullptr; none.slot_of = nullptr; none.mask = 0;
  FoldCells fc;
  std::memset(&fc, 0, sizeof(fc));
  if (n > 0) hipLaunchKernelGGL((k_add_fold8<false, false>), dim3(nblk(n * 8, 256)), dim3(256), 0, s, add_pts, keys, idx, none, n, ds, g, tomb, ins_pts, ins_flag,
                            events, tp, work, ctr, work_cap, fc);
}
// The hash-grouped form: n is a launch bound when n_dev != nullptr.  key / best: add_hash_slots(max batch) words each, all ones
// between calls (the fold leaves them so); slot_of: one word per batch point.
void launch_add_fold_hashed(const float4* add_pts, int n, const int* n_dev, float ds, const GridView& g, unsigned long long* hkey,
                            unsigned long long* hbest, unsigned int* slot_of, unsigned char* tomb, float4* ins_pts, unsigned int* ins_flag,
                            unsigned int* events, unsigned int* tp, unsigned int* work, int* ctr, unsigned int work_cap, hipStream_t s,
                            bool inserted, const FoldCellsH* cells) {
  if (n <= 0) return;
  AddHash tb;
  tb.key = hkey; tb.best = hbest; tb.slot_of = slot_of;
  tb.mask = (unsigned int)(add_hash_slots(n) - 1);
  // inserted: k_map_decide has filled the table for exactly this list and bound (launch_map_decide_compact with hkey)
  if (!inserted) hipLaunchKernelGGL(k_addh_insert, dim3(nblk(n, 256)), dim3(256), 0, s, add_pts, n, n_dev, ds, tb, ins_flag, reinterpret_cast<int*>(events));
  FoldCells fc;
  std::memset(&fc, 0, sizeof(fc));
  fc.n_dev = n_dev;
  if (!cells) {
    hipLaunchKernelGGL((k_add_fold8<true, false>), dim3(nblk(n * 8, 256)), dim3(256), 0, s, add_pts, nullptr, nullptr, tb, n, ds, g, tomb, ins_pts, ins_flag,
                       events, tp, work, ctr, work_cap, fc);
    return;
  }
  // cells: the inserts' cells ride in the fold (and the second list's in workgroups behind it)
  fc.a.blocks = cells->blocks; fc.a.mask = cells->mask; fc.a.inv_cs = g.inv_cs; fc.a.tables_cap = cells->tables_cap;
  fc.a.tp = tp; fc.a.work = work; fc.a.ctr = ctr; fc.a.work_cap = work_cap;
  fc.a.dropped = cells->dropped; fc.a.drop_cap = cells->drop_cap; fc.a.key_of_id = cells->key_of_id;
  fc.ins_e = cells->ins_e;
  fc.list2 = cells->list2; fc.n2 = cells->n2 > 0 ? cells->n2 : 0; fc.n2_dev = cells->n2_dev; fc.ins_e2 = cells->ins_e2;
  fc.fold_blocks = (unsigned int)nblk(n * 8, 256);
  hipLaunchKernelGGL((k_add_fold8<true, true>), dim3(fc.fold_blocks + (unsigned int)nblk(fc.n2, 256)), dim3(256), 0, s, add_pts, nullptr, nullptr, tb, n, ds, g,
                     tomb, ins_pts, ins_flag, events, tp, work, ctr, work_cap, fc);
}

}  // namespace lii
