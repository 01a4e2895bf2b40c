// libliinit_hip — the device-resident local map (host side): index (re)build, in-place updates, the lii_map_* entry points and
// lii_map_incremental.  Kernels: lii_map.hip, lii_kernels.hip (index), lii_sort.hip.  Reference: include/ikd-Tree/ikd_Tree.cpp
// (Build :336-347, Add_Points :381-456, Delete_Point_Boxes :500-516), src/laserMapping.cpp:516-559 (map_incremental).
#include "lii_context.h"

using namespace lii_impl;

namespace lii_impl {


// Builds the device map from n float4 points in d_map_unsorted:
// key (block | local cell) -> radix sort -> gather (d_map: cell-sorted, compact) -> block ids by scan -> per-block cell tables +
// block table -> capacities with slack -> scan -> spread into d_pts (the live array) + cell_cap.  Room for `extra_blocks` more
// 8x8x8 blocks is provisioned in the tables (in-place updates create blocks without a rebuild).
int build_index(lii_handle h, int n, int extra_blocks) {
  hipStream_t s = h->stream;
  h->n_map = n;
  h->n_blocks = 0;
  h->n_used = 0;
  h->map_dirty = false;
  unsigned int n_blocks = 0;
  unsigned int* ranks = reinterpret_cast<unsigned int*>(h->d_keys_a);  // free after the sort
  if (n > 0) {
    const float inv_cs = 1.0f / h->cell_size;
    launch_map_keys(h->d_map_unsorted, n, inv_cs, h->d_keys_a, h->d_idx_a, s);
    sort_pairs_u64(h->d_sort_temp, h->sort_temp_bytes, h->d_keys_a, h->d_keys_b, h->d_idx_a, h->d_idx_b, n, s);
    launch_map_gather(h->d_map_unsorted, h->d_idx_b, n, h->d_map, s);
    unsigned int* flags = h->d_idx_a;  // free after the sort
    launch_block_flags(h->d_keys_b, n, flags, s);
    inclusive_scan_u32(h->d_sort_temp, h->sort_temp_bytes, flags, ranks, n, s);
    HIPCHK(h, hipMemcpyAsync(h->h_small, ranks + (n - 1), sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    std::memcpy(&n_blocks, h->h_small, sizeof(unsigned int));
  }
  const size_t want_blocks = size_t(n_blocks) + size_t(std::max(extra_blocks, 0));
  if (want_blocks + 1 > h->cells_cap_blocks || !h->d_cell_cap) {
    for (void* q : {static_cast<void*>(h->d_cells), static_cast<void*>(h->d_cell_cap), static_cast<void*>(h->d_tp), static_cast<void*>(h->d_cs_a),
                    static_cast<void*>(h->d_cs_b), static_cast<void*>(h->d_block_key)})
      if (q) HIPCHK(h, hipFree(q));
    h->d_cells = nullptr; h->d_cell_cap = nullptr; h->d_tp = nullptr; h->d_cs_a = nullptr; h->d_cs_b = nullptr; h->d_block_key = nullptr;
    // (+ 1: the last table of the pool is the shared all-empty one, k_ins_cells)
    const size_t want = h->map_tight ? want_blocks + 1 : std::max<size_t>(std::max<size_t>(want_blocks * 2, h->cells_cap_blocks), 4096);
    HIPCHK(h, dmalloc(&h->d_cells, want * 512));
    HIPCHK(h, dmalloc(&h->d_cell_cap, want * 512));
    HIPCHK(h, dmalloc(&h->d_tp, want * 512));
    HIPCHK(h, dmalloc(&h->d_cs_a, want * 512));
    HIPCHK(h, dmalloc(&h->d_cs_b, want * 512));
    HIPCHK(h, dmalloc(&h->d_block_key, want));
    h->cells_cap_blocks = want;
    if (want * 512 * sizeof(unsigned int) + 4096 > h->sort_temp_bytes) {  // the scans over the cell entries need their temporary storage
      if (h->d_sort_temp) HIPCHK(h, hipFree(h->d_sort_temp));
      h->d_sort_temp = nullptr;
      h->sort_temp_bytes = std::max(h->sort_temp_bytes, sort_temp_bytes(int(std::min<size_t>(want * 512, 0x7FFFFFFF))));
      HIPCHK(h, hipMalloc(&h->d_sort_temp, h->sort_temp_bytes));
    }
  }
  unsigned int bcap = next_pow2(std::max(1024u, 8u * (unsigned int)want_blocks));  // load factor <= 1/8 now, <= 1/2 before the next rebuild
  if (bcap > h->blocks_cap) {
    if (h->d_blocks) HIPCHK(h, hipFree(h->d_blocks));
    h->d_blocks = nullptr;
    HIPCHK(h, dmalloc(&h->d_blocks, size_t(bcap)));
    h->blocks_cap = bcap;
  }
  bcap = h->blocks_cap;
  h->block_mask = bcap - 1;
  h->n_blocks = int(n_blocks);
  // every table entry of the pool starts out zero: blocks created later by k_ins_cells find an empty cell table
  const size_t entries = h->cells_cap_blocks * 512;
  HIPCHK(h, hipMemsetAsync(h->d_cells, 0, sizeof(uint2) * entries, s));
  HIPCHK(h, hipMemsetAsync(h->d_cell_cap, 0, sizeof(unsigned int) * entries, s));
  HIPCHK(h, hipMemsetAsync(h->d_tp, 0, sizeof(unsigned int) * entries, s));
  HIPCHK(h, hipMemsetAsync(h->d_tomb, 0, size_t(h->pts_cap), s));
  HIPCHK(h, hipMemsetAsync(h->d_mapctr, 0, sizeof(int) * kMapCtrWords, s));
  launch_table_clear(h->d_blocks, bcap, s);
  if (n > 0) {
    launch_cells_fill(h->d_keys_b, ranks, n, h->d_blocks, h->block_mask, h->d_cells, h->d_block_key, s);
    const int ne = int(n_blocks) * 512;
    unsigned int* caps = h->d_cs_a;
    unsigned int* capsum = h->d_cs_b;
    launch_cell_caps(h->d_cells, ne, caps, s);
    inclusive_scan_u32(h->d_sort_temp, h->sort_temp_bytes, caps, capsum, ne, s);
    launch_spread(h->d_map, h->d_cells, h->d_cell_cap, caps, capsum, ne, h->d_pts, h->d_mapctr, n, int(n_blocks), s);
  }
  HIPCHK(h, hipGetLastError());
  h->win_valid = false;
  if (h->use_window && n > 0) {
    // the dense cell window: the box of the occupied blocks, one block of margin on every side (queries at the map's edge look one
    // cell out), if it fits 64 MiB of entries
    unsigned int* box = reinterpret_cast<unsigned int*>(h->d_cs_a);  // (free again)
    const unsigned int init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    std::memcpy(h->h_small + 2100, init, sizeof(init));
    HIPCHK(h, hipMemcpyAsync(box, h->h_small + 2100, sizeof(init), hipMemcpyHostToDevice, s));
    launch_win_bbox(h->d_blocks, bcap, box, s);
    HIPCHK(h, hipMemcpyAsync(h->h_small + 2100, box, sizeof(init), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    unsigned int bxx[6];
    std::memcpy(bxx, h->h_small + 2100, sizeof(bxx));
    const int bb = kCellBias >> kCoarseShift;
    size_t entries_w = 1;
    for (int a = 0; a < 3; a++) {
      h->win_org[a] = (int(bxx[a]) - bb - 1) * 8;
      h->win_dim[a] = (int(bxx[3 + a]) - int(bxx[a]) + 3) * 8;
      entries_w *= size_t(h->win_dim[a]);
    }
    if (bxx[0] <= bxx[3] && entries_w * sizeof(uint2) <= (64u << 20)) {
      if (entries_w > h->win_cap) {
        if (h->d_win) HIPCHK(h, hipFree(h->d_win));
        h->d_win = nullptr;
        HIPCHK(h, dmalloc(&h->d_win, entries_w));
        h->win_cap = entries_w;
      }
      HIPCHK(h, hipMemsetAsync(h->d_win, 0, sizeof(uint2) * entries_w, s));
      launch_win_fill(h->d_blocks, bcap, h->d_cells, h->d_win, h->win_org, h->win_dim, s);
      HIPCHK(h, hipGetLastError());
      h->win_valid = true;
    }
  }
  int rc = LII_OK;
  {  // the slots in use (sum of the capacities) - and a first capacity check
    HIPCHK(h, hipMemcpyAsync(h->h_small + 3072, h->d_mapctr, sizeof(int) * kMapCtrWords, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    int c[kMapCtrWords];
    std::memcpy(c, h->h_small + 3072, sizeof(c));
    h->n_used = c[kMapCtrUsed];
    if ((unsigned int)h->n_used > h->pts_cap) rc = fail(h, LII_ERR_CAPACITY, "local map with its per-cell slack exceeds the point array");
    h->pts_cap_eff = h->map_tight ? std::min<unsigned int>(h->pts_cap, (unsigned int)h->n_used + 256u) : h->pts_cap;
  }
  return rc;
}

// The device map is always current - unless the last in-place update ran out of provisioned room and parked some of its inserts
// (kMapCtrOverflow): a search must not run against that map (the parked points are missing from it, and which ones they are
// depends on the order of the update's atomics - the replicated maps of a sharded job would drift apart).  Every update sends
// its overflow flag to pinned memory behind its kernels (no synchronisation there); whoever searches next looks at it - by then
// it has long arrived - and only an update that did overflow pays for settling (rebuild + re-insertion of the parked points).
// lii_map_incremental leaves its in-place update running on a stream of its own: the next scan's arrival, de-skew and voxel
// filter (which touch neither the map nor the update's scratch) overlap it.  Whatever reads or writes the map, its counters or
// that scratch joins the update first - the search passes through commit_map, every lii_map_* entry point directly.
// The list sizes the next lii_map_incremental is enqueued for: the largest of the last eight calls + 25 % + 1024 (consecutive
// scans of a stream resemble each other; launching for some padding costs little, a list that outgrows its bound a repeat).
void note_list_sizes(lii_handle h, int n_add, int n_nodown) {
  h->list_hist[h->list_hist_n & 7][0] = n_add;
  h->list_hist[h->list_hist_n & 7][1] = n_nodown;
  h->list_hist_n++;
  int ma = 0, mn = 0;
  for (int k = 0; k < std::min(h->list_hist_n, 8); k++) { ma = std::max(ma, h->list_hist[k][0]); mn = std::max(mn, h->list_hist[k][1]); }
  h->pred_add = ma + ma / 4 + 1024;
  h->pred_nodown = mn + mn / 4 + 1024;
}
// Waits until the last in-place update has published its counters (k_map_publish): the word behind them is polled - the update's
// last packet is microseconds away when somebody asks - with a look at the event now and then (a stream in error must not hang us).
static int wait_mapflag(lii_handle h) {
  volatile int* seqw = &h->h_mapflag[kMapFlagSeqAt];
  unsigned int spins = 0;
  while (*seqw != h->map_seq) {
    if ((++spins & 0x3FFF) == 0) {
      const hipError_t q = hipEventQuery(h->ev_mapflag);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) return fail(h, LII_ERR_HIP, std::string("map update: ") + hipGetErrorString(q));
    }
    __builtin_ia32_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  if (*seqw != h->map_seq) HIPCHK(h, hipEventSynchronize(h->ev_mapflag));
  return LII_OK;
}
int map_join(lii_handle h) {
  // (an update enqueued for predicted sizes is settled here whichever stream it ran on: when map_apply had to rebuild the index
  // first, the update went onto the handle's own stream - map_async false - and its list sizes need the same check)
  if (!h->map_async && !h->lists_predicted) return LII_OK;
  h->map_async = false;
  { const int rcw = wait_mapflag(h); if (rcw != LII_OK) return rcw; }
  if (h->lists_predicted) {
    // lii_map_incremental enqueued this update for predicted list sizes.  The exact ones came along behind it: they feed the
    // next prediction, and an update whose lists outgrew their bounds did nothing (k_map_decide emptied them) - it is
    // repeated now, with the exact sizes (the lists themselves are untouched until the next lii_map_incremental).
    h->lists_predicted = false;
    const int ca = h->h_mapflag[kMapCtrWords], cn = h->h_mapflag[kMapCtrWords + 1];
    note_list_sizes(h, ca, cn);
    if (h->h_mapflag[kMapCtrWords + 2]) {
      h->map_repeats++;
      if (h->diag && h->map_repeats <= 8)
        std::fprintf(stderr, "[libliinit_hip] map update repeated: lists of %d / %d points, enqueued for %d / %d\n", ca, cn, h->bound_add, h->bound_nodown);
      h->map_flag_pending = false;  // (of the update that did nothing)
      if (h->map_fuse) h->ah_filled = true;  // (k_map_decide filled the fold's table up to the bound and no fold consumed it: map_apply clears it first)
      return map_apply(h, h->d_list_add, ca, true, h->d_list_nodown, cn, false, nullptr, nullptr, false);
    }
  }
  return LII_OK;
}
int commit_map(lii_handle h) {
  {
    const int rc = map_join(h);
    if (rc != LII_OK) return rc;
  }
  if (!h->map_dirty || !h->map_flag_pending) return LII_OK;
  { const int rcw = wait_mapflag(h); if (rcw != LII_OK) return rcw; }
  h->map_flag_pending = false;
  if (h->h_mapflag[kMapCtrOverflow] != 0) {  // ran out of provisioned room: rebuild + re-insertion of the parked points
    const int rc = map_counters(h, false);
    if (rc != LII_OK) return rc;
  } else {  // the counters came along: the host's copies are current again without a read of their own
    if (h->h_mapflag[kMapCtrWinStale] && h->win_valid) { h->win_valid = false; h->win_dropped++; }  // (the update touched a cell outside the window's box)
    h->n_used = h->h_mapflag[kMapCtrUsed];
    h->n_map = h->h_mapflag[kMapCtrValid];
    h->n_blocks = int(std::min<size_t>(size_t(std::max(h->h_mapflag[kMapCtrBlocks], 0)), h->cells_cap_blocks));
    h->map_dirty = false;
  }
  return LII_OK;
}

// Host copies of the device counters (one small synchronising read).  An update in flight that ran out of room (block tables,
// slack + tail of the point array) has parked the inserts it could not place in d_dropped: the index is rebuilt with more room
// and those points are inserted again - nothing is lost, the caller sees no error.  Only a dropped list that itself overflowed
// (cannot happen: it holds a whole batch) or a work-list overflow turns into LII_ERR_CAPACITY.
int map_counters(lii_handle h, bool already_synced) {
  {
    const int rc = map_join(h);
    if (rc != LII_OK) return rc;
  }
  h->map_flag_pending = false;
  if (!h->map_dirty) return LII_OK;
  if (!already_synced) {
    HIPCHK(h, hipMemcpyAsync(h->h_small + 3072, h->d_mapctr, sizeof(int) * kMapCtrWords, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  int c[kMapCtrWords];
  std::memcpy(c, h->h_small + 3072, sizeof(c));
  if (c[kMapCtrWinStale] && h->win_valid) { h->win_valid = false; h->win_dropped++; }
  h->n_used = c[kMapCtrUsed];
  h->n_map = c[kMapCtrValid];
  h->n_blocks = int(std::min<size_t>(size_t(std::max(c[kMapCtrBlocks], 0)), h->cells_cap_blocks));  // (the counter runs past the pool when it is exhausted)
  h->map_dirty = false;
  if (c[kMapCtrOverflow]) {
    const int n_drop = c[kMapCtrDropped];
    if (n_drop < 0 || (unsigned int)n_drop > h->drop_cap || (size_t)n_drop > size_t(h->cfg.max_map_points)) {
      (void)map_rebuild(h, 0);
      return fail(h, LII_ERR_CAPACITY, "local map update ran out of room and could not keep the inserts; the map was rebuilt from the points it holds");
    }
    h->map_recoveries++;
    if (n_drop > 0)  // (the rebuild leaves d_dropped alone; the batch buffer is free: its Add_Points call has returned)
      HIPCHK(h, hipMemcpyAsync(h->d_batch, h->d_dropped, sizeof(float4) * size_t(n_drop), hipMemcpyDeviceToDevice, h->stream));
    int rc = map_rebuild(h, std::max(4096, 2 * n_drop));
    if (rc != LII_OK) return rc;
    if (n_drop > 0) {
      const bool tight = h->map_tight;
      h->map_tight = false;  // the second attempt gets its room
      rc = map_apply(h, h->d_batch, n_drop, false, nullptr, 0);
      h->map_tight = tight;
      if (rc != LII_OK) return rc;
      // settle it now: the caller of map_counters goes on with counters that include the re-inserted points
      HIPCHK(h, hipMemcpyAsync(h->h_small + 3072, h->d_mapctr, sizeof(int) * kMapCtrWords, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      std::memcpy(c, h->h_small + 3072, sizeof(c));
      h->n_used = c[kMapCtrUsed];
      h->n_map = c[kMapCtrValid];
      h->n_blocks = int(std::min<size_t>(size_t(std::max(c[kMapCtrBlocks], 0)), h->cells_cap_blocks));
      h->map_dirty = false;
      if (c[kMapCtrOverflow]) return fail(h, LII_ERR_CAPACITY, "local map: the re-insertion after a rebuild ran out of room again");
    }
  }
  return LII_OK;
}
// Gathers the live points into d_map_unsorted (entry order) and returns their number.
int map_gather(lii_handle h, int* n_out) {
  hipStream_t s = h->stream;
  const int ne = h->n_blocks * 512;
  *n_out = 0;
  if (ne <= 0) return LII_OK;
  launch_cell_counts(h->d_cells, ne, h->d_cs_a, s);
  inclusive_scan_u32(h->d_sort_temp, h->sort_temp_bytes, h->d_cs_a, h->d_cs_b, ne, s);
  launch_gather_live(h->d_pts, h->d_cells, h->d_cs_b, ne, h->d_map_unsorted, h->cfg.max_map_points, s);
  HIPCHK(h, hipMemcpyAsync(h->h_small + 3000, h->d_cs_b + (ne - 1), sizeof(unsigned int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  unsigned int n = 0;
  std::memcpy(&n, h->h_small + 3000, sizeof(n));
  *n_out = int(std::min<unsigned int>(n, (unsigned int)h->cfg.max_map_points));
  return LII_OK;
}
// Garbage collection: the live points, re-sorted and laid out with fresh slack (restores the cell-sorted order of the array).
int map_rebuild(lii_handle h, int extra_blocks) {
  int n = 0;
  int rc = map_gather(h, &n);
  if (rc != LII_OK) return rc;
  return build_index(h, n, extra_blocks);
}

// Applies one Add_Points batch IN PLACE (lii_map.hip): `list` holds n_list points (device float4) added with or without the
// per-voxel down-sampling; `extra` (n_extra points) is added without it afterwards (map_incremental's PointNoNeedDownsample).
// Nothing is synchronised: the counters move on the device (map_counters reads them when somebody asks).  The capacity check is
// made BEFORE anything is touched and is conservative: n_valid + n_list + n_extra <= max_map_points (a down-sampled batch may
// replace points instead of adding them; the check still counts every point of it) - a refused batch leaves the map untouched.
// n_list_dev / n_extra_dev != nullptr: the lists hold *n_list_dev / *n_extra_dev points (device-resident), n_list / n_extra are
// bounds of them (lii_map_incremental's predicted sizes); the launches are made for the bounds.
// count_events = false (lii_map_incremental): Add_Points' event counter is not needed - the down-sampled list is folded through
// the hash table instead of the batch sort (lii_map.hip: AddHash).
lii::WinKeep win_keep_view(lii_handle h) {
  lii::WinKeep wk = {};
  if (h->win_valid && h->win_keep && h->d_win && h->d_block_key) {
    wk.win = h->d_win; wk.key_of_id = h->d_block_key;
    wk.x0 = h->win_org[0]; wk.y0 = h->win_org[1]; wk.z0 = h->win_org[2];
    wk.nx = h->win_dim[0]; wk.ny = h->win_dim[1]; wk.nz = h->win_dim[2];
  }
  return wk;
}
int map_apply(lii_handle h, const float4* list, int n_list, bool downsample, const float4* extra, int n_extra, bool beside,
              const int* n_list_dev, const int* n_extra_dev, bool count_events, bool prefilled) {
  hipStream_t s = h->stream;
  int rc = map_counters(h);
  if (rc != LII_OK) return rc;
  const int n_ins = n_list + n_extra;
  if (n_ins <= 0) return LII_OK;
  if ((long long)h->n_map + n_ins > (long long)h->cfg.max_map_points) return fail(h, LII_ERR_CAPACITY, "local map exceeds max_map_points");
  // Room is provisioned for the usual batch, not for the worst one: a handful of new 8x8x8-cell blocks, and a tail slot budget
  // of 8 per insert (an insert that does not fit its cell's slack moves the cell - ~9 points + fresh slack - to the tail; most
  // fit).  A batch that needs more parks the inserts it cannot place in d_dropped and the next map_counters() rebuilds and
  // re-inserts them (lossless, slow: a full rebuild).
  const long long tail_need = h->map_tight ? 0 : 8ll * n_ins + 4096;
  const size_t spare_blocks = h->map_tight ? 0 : std::min<size_t>(size_t(n_ins), 1024);
  const unsigned int work_need = 9u * (unsigned int)n_list + (unsigned int)n_ins + 64u;
  if (work_need > h->work_cap) return fail(h, LII_ERR_CAPACITY, "Add_Points batch larger than the work list of the in-place update");
  if (h->test_force_rebuild || (long long)h->pts_cap_eff - h->n_used < tail_need || size_t(h->n_blocks) + spare_blocks + 1 > h->cells_cap_blocks ||
      2ull * (size_t(h->n_blocks) + spare_blocks) > size_t(h->blocks_cap)) {
    rc = map_rebuild(h, h->map_tight ? 0 : std::max(4096, h->n_blocks / 2));
    if (rc != LII_OK) return rc;
    if ((long long)h->pts_cap_eff - h->n_used < tail_need) return fail(h, LII_ERR_CAPACITY, "local map: no room left behind the cells for an in-place update");
    beside = false;  // (the rebuild ran on the handle's stream and has not been waited for)
  }
  // `beside`: the update runs on the map stream from here (see map_join), behind what the handle's stream holds now
  if (beside) {
    if (!h->map_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->map_stream, hipStreamNonBlocking));  // (a handle that left a job)
    HIPCHK(h, hipEventRecord(h->ev_lists, h->stream));
    s = h->map_stream;
    HIPCHK(h, hipStreamWaitEvent(s, h->ev_lists, 0));
  }
  const GridView g = grid_view(h);
  h->map_dirty = true;
  const unsigned int tables_cap = (unsigned int)h->cells_cap_blocks;
  if (!(downsample && n_list > 0)) HIPCHK(h, hipMemsetAsync(h->d_mapctr + kMapCtrEvents, 0, sizeof(int), s));  // (else: k_add_keys / k_addh_insert)
  // one launch each for the cells of both insert lists and for writing both (the second list rides behind the first)
  const float4* list_a = list;
  const unsigned int* flags_a = nullptr;
  bool cells_done = false;
  const bool hashed = downsample && n_list > 0 && !count_events && !h->fold_sorted && n_list <= h->cfg.max_scan_points;
  // The table of the hash-grouped fold is all ones between updates.  k_map_decide may have filled it for THIS list and bound (`prefilled`:
  // map_update_early); a fill that no fold consumed - the list outgrew its bound, or the update before this one never got as far as its
  // fold - is cleared before anything else uses the table.
  if (h->ah_filled && !(hashed && prefilled)) {
    const size_t slots = lii::add_hash_slots(h->cfg.max_scan_points);
    HIPCHK(h, hipMemsetAsync(h->d_ah_key, 0xFF, 8 * slots, s));
    HIPCHK(h, hipMemsetAsync(h->d_ah_best, 0xFF, 8 * slots, s));
    h->ah_cleared++;
    prefilled = false;
  }
  h->ah_filled = false;
  if (hashed) {
    // round 6: the cells of the inserts are found / created inside the fold launch, the second list's in workgroups behind the fold's
    // (LII_MAP_FUSE=0: launch_ins_cells behind the fold, the form of rounds 2 - 5)
    lii::FoldCellsH fcells;
    fcells.blocks = h->d_blocks; fcells.mask = h->block_mask; fcells.tables_cap = tables_cap; fcells.ins_e = h->d_ins_e; fcells.dropped = h->d_dropped;
    fcells.drop_cap = h->drop_cap; fcells.key_of_id = h->d_block_key; fcells.list2 = extra; fcells.n2 = n_extra; fcells.n2_dev = n_extra_dev; fcells.ins_e2 = h->d_ins_e2;
    cells_done = h->map_fuse;
    launch_add_fold_hashed(list, n_list, n_list_dev, h->ds, g, h->d_ah_key, h->d_ah_best, h->d_ah_slot, h->d_tomb, h->d_ins, h->d_u32_a,
                           reinterpret_cast<unsigned int*>(h->d_mapctr + kMapCtrEvents), h->d_tp, h->d_work, h->d_mapctr, h->work_cap, s,
                           prefilled, cells_done ? &fcells : nullptr);
    list_a = h->d_ins;
    flags_a = h->d_u32_a;
  } else if (downsample && n_list > 0) {
    launch_add_keys(list, n_list, n_list_dev, h->ds, h->d_keys_a, h->d_idx_a, h->d_mapctr + kMapCtrEvents, s);
    sort_pairs_u64(h->d_sort_temp, h->sort_temp_bytes, h->d_keys_a, h->d_keys_b, h->d_idx_a, h->d_idx_b, n_list, s);
    launch_add_fold(list, h->d_keys_b, h->d_idx_b, n_list, h->ds, g, h->d_tomb, h->d_ins, h->d_u32_a,
                    reinterpret_cast<unsigned int*>(h->d_mapctr + kMapCtrEvents), h->d_tp, h->d_work, h->d_mapctr, h->work_cap, s);
    list_a = h->d_ins;
    flags_a = h->d_u32_a;
  }
  if (!cells_done)
    launch_ins_cells(list_a, flags_a, n_list, flags_a ? nullptr : n_list_dev, extra, n_extra, n_extra_dev, h->d_ins_e2, h->d_blocks, h->block_mask, g.inv_cs, tables_cap, h->d_ins_e, h->d_tp,
                     h->d_work, h->d_mapctr, h->work_cap, h->d_dropped, h->drop_cap, h->d_block_key, s);
  // (cell entries change: the dense window is kept current by the two launches that change them - WinKeep - or dropped)
  const lii::WinKeep wk = win_keep_view(h);
  if (wk.win) h->win_kept++;
  else if (h->win_valid) { h->win_valid = false; h->win_dropped++; }
  launch_cell_apply(h->d_work, h->d_cells, h->d_cell_cap, h->d_pts, h->d_tomb, h->d_tp, h->d_mapctr, h->pts_cap_eff, (int)work_need, wk, s);
  h->map_seq = h->map_seq == 0x7FFFFFFF ? 1 : h->map_seq + 1;
  launch_ins_write(list_a, h->d_ins_e, n_list, flags_a ? nullptr : n_list_dev, extra, h->d_ins_e2, n_extra, n_extra_dev, h->d_cells, h->d_cell_cap, h->d_pts, h->d_mapctr, h->d_dropped,
                   h->drop_cap, s, h->h_mapflag, kMapCtrWords + 8, kMapFlagSeqAt, h->map_seq, wk);
  HIPCHK(h, hipGetLastError());
  // (the update's counters, its overflow flag and the list sizes of lii_map_incremental travel to the host with the last workgroup
  // of k_ins_write: see wait_mapflag / commit_map; the event behind it is for a stream in error only)
  h->lists_predicted = n_list_dev != nullptr;
  HIPCHK(h, hipEventRecord(h->ev_mapflag, s));
  h->map_flag_pending = true;
  h->map_async = beside;
  return LII_OK;
}

// lii_scan_job::map_update: map_incremental enqueued BEHIND the passes of the iterated update that is still running (called by
// update_on_device between its last launch and its wait for the result): the host's ~ 45 us of launches for the map update pass
// while the device registers the scan, instead of after the result has come back with the device idle.  The decision kernel
// takes the update's final state from the control block and does nothing unless the update has ended regularly (k_map_decide).
// Only the form that waits for nothing can be enqueued ahead: predicted list sizes, a map with room for them, one rank.
int map_update_early(lii_handle h) {
  const int nb = h->n_body;
  if (nb <= 0 || nb > h->cfg.max_map_points || h->net.n_ranks > 1 || h->pred_add < 0 || h->map_dirty || h->map_async || h->lists_predicted) return 0;
  if ((long long)h->n_map + std::min(nb, h->pred_add) + std::min(nb, h->pred_nodown) > (long long)h->cfg.max_map_points) return 0;
  RegistrationBuffers rb = reg_buffers(h);
  int ba = std::min(nb, h->pred_add), bn = std::min(nb, h->pred_nodown);
  if (h->test_pred_small) { ba = std::min(ba, 16); bn = std::min(bn, 16); }
  h->bound_add = ba; h->bound_nodown = bn;
  PoseArg unused;
  std::memset(&unused, 0, sizeof(unused));
  if (++h->decide_epoch == 0u) h->decide_epoch = 1u;
  // round 6: the hash insert of the fold rides in the decision launch when the fold that follows will take the hash-grouped form
  const bool fill = h->map_fuse && ba > 0 && !h->fold_sorted && ba <= h->cfg.max_scan_points && !h->ah_filled;
  launch_map_decide_compact(rb, unused, double(h->cfg.map_downsample_size), 1, reinterpret_cast<unsigned long long*>(h->d_u32_b), h->decide_epoch, h->d_world,
                            h->d_list_add, h->d_list_nodown, h->d_counts, ba, bn, h->stream, h->d_ctrl, h->update_seq, h->test_emit_late ? 1 : 0,
                            fill ? h->d_ah_key : nullptr, h->d_ah_best, h->d_ah_slot, h->ds, h->d_u32_a, h->d_mapctr + kMapCtrEvents);
  if (fill) h->ah_filled = true;
  // (on the handle's own stream: the update sits right behind the passes anyway, and handing it to the map stream costs more - an
  // event between two hardware queues - than the next scan's de-skew and voxel filter beside it bring back: 5 007 against 4 826
  // scans/s with an update every scan, gpurun_out/r4y)
  const int rc = map_apply(h, h->d_list_add, ba, true, h->d_list_nodown, bn, false, h->d_counts + 3, h->d_counts + 4, false, fill);
  return rc == LII_OK ? 1 : rc;
}

}  // namespace lii_impl

extern "C" {

// ------------------------------------------------------------------------------------------------ map
int lii_map_reset(lii_handle h) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h) return LII_ERR_INVALID;
  {
    const int rcj = map_join(h);
    if (rcj != LII_OK) return rcj;
  }
  h->have_search = false;
  return build_index(h, 0);
}
namespace {
// host xyz (stride in bytes) -> pinned float4 staging -> device buffer
int upload_xyz(lii_handle h, const void* xyz, int n, int stride_bytes, float4* dst) {
  const char* src = static_cast<const char*>(xyz);
  HIPCHK(h, hipEventSynchronize(h->ev_stage));  // an asynchronous scan upload may still be reading the staging buffer
  for (int i = 0; i < n; i++) {
    const float* f = reinterpret_cast<const float*>(src + size_t(i) * stride_bytes);
    h->h_stage[i] = make_float4(f[0], f[1], f[2], 0.f);
  }
  if (n > 0) {
    HIPCHK(h, hipMemcpyAsync(dst, h->h_stage, sizeof(float4) * size_t(n), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // h_stage is reused
  }
  return LII_OK;
}
}  // namespace
int lii_map_build(lii_handle h, const void* xyz, int32_t n, int32_t stride_bytes) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || (!xyz && n > 0) || n < 0 || stride_bytes < 12 || stride_bytes % 4) return fail(h, LII_ERR_INVALID, "lii_map_build: bad arguments");
  if (n > h->cfg.max_map_points) return fail(h, LII_ERR_CAPACITY, "lii_map_build: n > max_map_points");
  {
    const int rcj = map_join(h);
    if (rcj != LII_OK) return rcj;
  }
  h->have_search = false;
  int rc = upload_xyz(h, xyz, n, stride_bytes, h->d_map_unsorted);
  if (rc != LII_OK) return rc;
  rc = build_index(h, n);
  if (rc != LII_OK) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return LII_OK;
}
int lii_map_add_points(lii_handle h, const void* xyz, int32_t n, int32_t stride_bytes, int32_t downsample_on, int32_t* n_added) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || (!xyz && n > 0) || n < 0 || stride_bytes < 12 || stride_bytes % 4) return fail(h, LII_ERR_INVALID, "lii_map_add_points: bad arguments");
  if (n > h->cfg.max_map_points) return fail(h, LII_ERR_CAPACITY, "lii_map_add_points: batch larger than max_map_points");
  if (n_added) *n_added = 0;
  if (n == 0) return LII_OK;
  // settle an earlier update first: should it have to be completed by a rebuild, the re-insertion uses the batch buffer
  int rc = map_counters(h);
  if (rc != LII_OK) return rc;
  rc = upload_xyz(h, xyz, n, stride_bytes, h->d_batch);
  if (rc != LII_OK) return rc;
  h->have_search = false;
  rc = map_apply(h, h->d_batch, n, downsample_on != 0, nullptr, 0);
  if (rc != LII_OK) return rc;
  int ev = 0;  // this entry point reports Add_Points' counter: one synchronising read (lii_map_incremental does not)
  HIPCHK(h, hipMemcpyAsync(h->h_small + 3090, h->d_mapctr + kMapCtrEvents, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  rc = map_counters(h);  // synchronises; reports a capacity problem of this very update
  if (rc != LII_OK) return rc;
  std::memcpy(&ev, h->h_small + 3090, sizeof(int));
  if (n_added) *n_added = downsample_on ? ev : 0;
  return LII_OK;
}
int lii_map_delete_boxes(lii_handle h, const float* boxes, int32_t n_boxes, int32_t* n_deleted) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || (!boxes && n_boxes > 0) || n_boxes < 0 || n_boxes > 4096) return fail(h, LII_ERR_INVALID, "lii_map_delete_boxes: bad arguments (<= 4096 boxes)");
  if (n_deleted) *n_deleted = 0;
  int rc = map_counters(h);
  if (rc != LII_OK) return rc;
  const int n_old = h->n_map;
  if (n_boxes == 0 || n_old == 0) return LII_OK;
  hipStream_t s = h->stream;
  float* d_boxes = reinterpret_cast<float*>(h->d_keys_b);  // scratch of the index build (max_map_points * 8 bytes), free between calls
  if (size_t(n_boxes) * 24 > size_t(h->cfg.max_map_points) * 8)
    return fail(h, LII_ERR_INVALID, "lii_map_delete_boxes: more boxes than the handle's scratch holds (max_map_points / 3)");
  std::memcpy(h->h_small + 4096, boxes, sizeof(float) * 6 * size_t(n_boxes));
  HIPCHK(h, hipMemcpyAsync(d_boxes, h->h_small + 4096, sizeof(float) * 6 * size_t(n_boxes), hipMemcpyHostToDevice, s));
  // in place: every cell walks its live points, the cells that lose points squeeze them out (k_cell_apply)
  const int ne = h->n_blocks * 512;
  if ((unsigned int)ne > h->work_cap) {  // more cells than the work list holds: rebuild-free fallback is not worth it - grow the list
    if (h->d_work) HIPCHK(h, hipFree(h->d_work));
    h->d_work = nullptr;
    h->work_cap = (unsigned int)ne + 4096u;
    HIPCHK(h, dmalloc(&h->d_work, size_t(h->work_cap)));
  }
  h->map_dirty = true;
  launch_box_tomb_cells(h->d_pts, h->d_cells, ne, d_boxes, n_boxes, h->d_tomb, h->d_tp, h->d_work, h->d_mapctr, h->work_cap, s);
  const lii::WinKeep wk = win_keep_view(h);
  if (!wk.win) h->win_valid = false;
  launch_cell_apply(h->d_work, h->d_cells, h->d_cell_cap, h->d_pts, h->d_tomb, h->d_tp, h->d_mapctr, h->pts_cap_eff, ne, wk, s);
  launch_ins_write(h->d_pts, h->d_ins_e, 0, nullptr, nullptr, nullptr, 0, nullptr, h->d_cells, h->d_cell_cap, h->d_pts, h->d_mapctr, h->d_dropped, h->drop_cap,
                   s, nullptr, 0, 0, 0, wk);  // (re-arms the work list)
  rc = map_counters(h);
  if (rc != LII_OK) return rc;
  if (n_deleted) *n_deleted = n_old - h->n_map;
  if (h->n_map != n_old) h->have_search = false;
  return LII_OK;
}
int lii_map_size(lii_handle h, int32_t* n_valid) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !n_valid) return LII_ERR_INVALID;
  int rc = map_counters(h);  // (a pending in-place update: one small synchronising read)
  *n_valid = h->n_map;
  return rc;
}
int lii_map_download(lii_handle h, float* xyz_out, int32_t capacity, int32_t* n) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !n) return LII_ERR_INVALID;
  int rc = map_counters(h);
  if (rc != LII_OK) return rc;
  int cnt = h->n_map;
  *n = cnt;
  if (!xyz_out) return LII_OK;
  if (capacity < cnt) return fail(h, LII_ERR_CAPACITY, "lii_map_download: capacity too small");
  if (cnt == 0) return LII_OK;
  rc = map_gather(h, &cnt);  // the live points, cell by cell (the array itself has slack between the cells)
  if (rc != LII_OK) return rc;
  *n = cnt;
  HIPCHK(h, hipMemcpyAsync(h->h_stage, h->d_map_unsorted, sizeof(float4) * size_t(cnt), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < cnt; i++) {
    xyz_out[3 * size_t(i)] = h->h_stage[i].x;
    xyz_out[3 * size_t(i) + 1] = h->h_stage[i].y;
    xyz_out[3 * size_t(i) + 2] = h->h_stage[i].z;
  }
  return LII_OK;
}
int lii_map_commit(lii_handle h) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h) return LII_ERR_INVALID;
  {
    const int rcj = map_join(h);
    if (rcj != LII_OK) return rcj;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return LII_OK;
}


int lii_map_incremental(lii_handle h, const lii_state* state, int32_t* n_add, int32_t* n_no_downsample) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !state) return fail(h, LII_ERR_INVALID, "lii_map_incremental: bad arguments");
  if (n_add) *n_add = 0;
  if (n_no_downsample) *n_no_downsample = 0;
  const int nb = h->n_body;  // upper bound while the exact count is still on the device
  if (nb <= 0) return LII_OK;
  if (nb > h->cfg.max_map_points) return fail(h, LII_ERR_CAPACITY, "lii_map_incremental: scan larger than max_map_points");
  {
    const int rcj = map_join(h);
    if (rcj != LII_OK) return rcj;
  }
  hipStream_t s = h->stream;
  RegistrationBuffers rb = reg_buffers(h);
  const lii::GatherView gv = gather_view(h);
  // (the lists travel through the peers' gather areas - the node-local mailbox - or through ncclAllGather; a one-rank RCCL job
  // exchanges with itself: the only form of that path a single-device box can execute)
  const bool exchange_rccl = h->net.comm != nullptr && h->net.library_partition;
  const bool exchange = (h->net.n_ranks > 1 && h->net.library_partition && gv.peers != nullptr) || exchange_rccl;
  if (rb.shard_world > 1 && !exchange) {
    // A job split by index on a transport without the list exchange (host-memory mailbox, RCCL): this rank holds neighbour lists
    // for its own block only, but every rank must take the SAME decisions for the whole cloud or the replicated maps drift apart.
    // The search is repeated here for the whole cloud at the pose of the last executed search pass (IekfCtrl::search_pose,
    // identical on every rank) - one more k-NN pass - and the lists come out bit-identical on every rank.
    rb.shard_world = 1;
    if (h->have_search) {
      const GridView g = grid_view(h);
      lii::launch_knn(g, rb, reinterpret_cast<const PoseArg*>(h->d_ctrl->search_pose), h->d_ctrl, 2, nullptr, s, 0);  // (k_knn_complete behind it works from the flags in nbr_count, not from a list)
      launch_knn_complete(g, rb, s);
    }
  }
  // decision per point on the device (world point, neighbour list of the last search) and both order-preserving compactions
  const bool sharded = h->net.n_ranks > 1 || exchange_rccl;
  // (near max_map_points the padded bounds could fail the capacity test a batch of the exact sizes passes: the waiting form then)
  const bool room_for_bounds = h->pred_add >= 0 && !h->map_dirty &&
                               (long long)h->n_map + std::min(nb, h->pred_add) + std::min(nb, h->pred_nodown) <= (long long)h->cfg.max_map_points;
  if (!n_add && !n_no_downsample && !sharded && h->pred_add >= 0 && room_for_bounds) {
    // Nobody asks for the list sizes: the update is enqueued for PREDICTED sizes (note_list_sizes) right
    // behind the compaction, on the map stream - no host round trip, and the next scan's arrival / de-skew / voxel filter overlap
    // it.  The exact sizes stay on the device (d_counts); commit_map reads them behind the update and repeats an update whose
    // lists outgrew the prediction.  The host's copies of the map counters: see commit_map (the previous update has been joined
    // by the search of this scan, so they are current here).
    int ba = std::min(nb, h->pred_add), bn = std::min(nb, h->pred_nodown);
    if (h->test_pred_small) { ba = std::min(ba, 16); bn = std::min(bn, 16); }
    h->bound_add = ba; h->bound_nodown = bn;  // (LII_TEST=pred_small: every update outgrows its bounds)
    if (++h->decide_epoch == 0u) h->decide_epoch = 1u;
    launch_map_decide_compact(rb, pose_of(*state), double(h->cfg.map_downsample_size), h->have_search ? 1 : 0,
                              reinterpret_cast<unsigned long long*>(h->d_u32_b), h->decide_epoch, h->d_world, h->d_list_add, h->d_list_nodown, h->d_counts, ba, bn, s, nullptr, 0, h->test_emit_late ? 1 : 0);
    return map_apply(h, h->d_list_add, ba, true, h->d_list_nodown, bn, true, h->d_counts + 3, h->d_counts + 4, false);
  }
  if (++h->decide_epoch == 0u) h->decide_epoch = 1u;
  launch_map_decide_compact(rb, pose_of(*state), double(h->cfg.map_downsample_size), h->have_search ? 1 : 0, reinterpret_cast<unsigned long long*>(h->d_u32_b),
                            h->decide_epoch, h->d_world, h->d_list_add, h->d_list_nodown, h->d_counts, nb, nb, s, nullptr, 0, h->test_emit_late ? 1 : 0);
  if (exchange) {
    // This rank has decided for ITS points (its block of the cloud, or its voxels): the lists of all ranks, in rank order, are the
    // batch every replica of the map receives (lii_exchange.hip: remote stores into the peers' gather areas; in place here).
    if (exchange_rccl) {
      const int rcx = lists_exchange_rccl(h, s);
      if (rcx != LII_OK) return rcx;
    } else {
      launch_lists_exchange(gv, h->d_list_add, h->d_list_nodown, h->d_counts, 5, h->net.d_gather_ticket, ++h->net.gather_seq, h->d_list_add,
                            h->d_list_nodown, s);
    }
  }
  // The sizes of the two lists, now: a converged map takes a few thousand of the ~100 k points, and everything downstream
  // (voxel keys, the batch sort, the per-voxel fold, the insert compaction) is launched for the exact count instead of the
  // scan-sized bound - one small host round trip (~15 us) against ~80 us of kernels working on padding.
  // (the same round trip brings the map's counters up to date: the previous update ran without a synchronisation)
  HIPCHK(h, hipMemcpyAsync(h->h_small + 3090, h->d_counts, 6 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipMemcpyAsync(h->h_small + 3072, h->d_mapctr, sizeof(int) * kMapCtrWords, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  int n_lists[6];
  std::memcpy(n_lists, h->h_small + 3090, sizeof(n_lists));
  if (exchange && n_lists[5])
    return fail(h, LII_ERR_COMM, "list exchange timed out (a rank of the job did not reach this map update); re-create the communicator");
  {
    const int rc0 = map_counters(h, true);
    if (rc0 != LII_OK) return rc0;
  }
  note_list_sizes(h, n_lists[0], n_lists[1]);
  // Add_Points(PointToAdd, true) then Add_Points(PointNoNeedDownsample, false)  (:556-557)
  // (the stream has just been synchronised: the update may run beside whatever the caller enqueues next - a sharded job keeps
  // one stream: its search of the whole cloud above reads the control block the next scan's arrival rewrites)
  int rc = map_apply(h, h->d_list_add, n_lists[0], true, h->d_list_nodown, n_lists[1], !sharded, nullptr, nullptr, false);
  if (rc != LII_OK) return rc;
  if (n_add) *n_add = n_lists[0];
  if (n_no_downsample) *n_no_downsample = n_lists[1];
  return LII_OK;
}

}  // extern "C"
