// libliinit_hip — C-ABI implementation (host side).  Declarations and the reference code each entry point
// replaces: include/liinit_hip.h.  Device work: lii_kernels.hip / lii_sort.hip on ONE stream per handle.
// There is deliberately no CPU implementation of any stage here: without a usable gfx950 device
// lii_create fails with LII_ERR_NO_DEVICE.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/liinit_hip.h"
#include "lii_hostmath.h"
#include "lii_launch.h"

using namespace lii;

namespace {
constexpr size_t kCtrlBytes = (sizeof(IekfCtrl) + 255) / 256 * 256;
}

struct lii_context {
  lii_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // ---- local map (device resident).  d_pts is the live point array: cell by cell with slack behind every cell (in-place
  // updates, lii_map.hip); d_map_unsorted / d_map are staging for (re)builds (input, then cell-sorted and compact).
  float ds = 0.2f;              // ikd-Tree downsample box (set_downsample_param)
  unsigned char* d_tomb = nullptr;
  float4* d_batch = nullptr;    // a host-provided Add_Points batch (M)
  float4* d_dropped = nullptr;  // inserts an in-place update found no room for (kMapCtrDropped of them): re-inserted after a rebuild
  unsigned int drop_cap = 0;
  bool map_tight = false;       // LII_TEST=map_tight: no spare room is provisioned (tests: forces the recovery path)
  long long map_recoveries = 0;
  float4 *d_ins = nullptr, *d_ins_c = nullptr;         // fold output / compacted inserts or host batches (M each)
  unsigned int *d_u32_a = nullptr, *d_u32_b = nullptr, *d_u32_c = nullptr;  // flags / ranks (max(N, M) each)
  float4 *d_list_add = nullptr, *d_list_nodown = nullptr;  // map_incremental lists (N each)
  int* d_counts = nullptr;      // [0] add list, [1] no-downsample list, [2] alive, [3] inserted, [4] total, [5] events
  float4* d_map_unsorted = nullptr;
  float4* d_map = nullptr;
  float4* d_pts = nullptr;            // pts_cap slots
  unsigned int pts_cap = 0;
  unsigned int pts_cap_eff = 0;  // = pts_cap (LII_TEST=map_tight: a few slots behind the cells, so that updates run out of room)
  unsigned int* d_cell_cap = nullptr; // capacity end of every cell entry (same indexing as d_cells)
  unsigned int* d_tp = nullptr;       // per cell entry: on-work-list bit | pending inserts
  unsigned int *d_cs_a = nullptr, *d_cs_b = nullptr;  // per cell entry scratch (capacities / counts and their scans)
  unsigned int* d_work = nullptr;     // work list of the update in flight (cell entries)
  unsigned int work_cap = 0;
  unsigned int *d_ins_e = nullptr, *d_ins_e2 = nullptr;  // cell entry of every insert (fold output / plain list)
  unsigned long long* d_ah_key = nullptr;   // hash-grouped fold of lii_map_incremental (lii_map.hip: AddHash): voxel keys,
  unsigned long long* d_ah_best = nullptr;  // per-slot minima (both all ones between updates),
  unsigned int* d_ah_slot = nullptr;        // the slot of every batch point
  bool fold_sorted = false;                 // LII_TEST=fold_sort: lii_map_incremental folds through the batch sort as lii_map_add_points does
  int* d_mapctr = nullptr;            // kMapCtr* counters
  int n_used = 0;                     // host copy of kMapCtrUsed as of the last map_counters()
  bool map_dirty = false;             // an update has been enqueued since the last map_counters(): n_map / n_used / n_blocks are stale
  unsigned long long *d_keys_a = nullptr, *d_keys_b = nullptr, *d_keys_c = nullptr;
  unsigned int *d_idx_a = nullptr, *d_idx_b = nullptr;
  BlockEntry* d_blocks = nullptr;   // capacity-managed (grows on demand)
  unsigned int blocks_cap = 0;      // allocated entries
  unsigned int block_mask = 0;      // entries in use - 1
  uint2* d_cells = nullptr;         // capacity-managed: 512 entries per occupied block
  size_t cells_cap_blocks = 0;
  int n_blocks = 0;
  unsigned int* d_counter = nullptr;
  int partial_stride = 0;
  int n_map = 0;
  int* n_map_pinned = nullptr;  // small pinned scratch for H2D of counters
  float cell_size = 0.3f;
  void* d_sort_temp = nullptr;
  size_t sort_temp_bytes = 0;

  // ---- scan
  float4* d_scan = nullptr;   // raw / undistorted (x,y,z,t_ms)
  // lii_scan_upload_next / lii_scan_advance: the next scan travels on a copy stream into a second buffer
  float4* d_scan_next = nullptr;
  float4* h_stage_next = nullptr;   // pinned staging for sources that are not (pinned, stride 16)
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_next = nullptr;      // the transfer of the next scan
  hipEvent_t ev_scan_free = nullptr; // the compute stream has finished with the buffer the next transfer writes to
  int n_scan_next = -1;              // >= 0: a scan is waiting in d_scan_next
  float4* d_body = nullptr;   // down-sampled body points
  float4* d_world = nullptr;
  float4* d_nbr = nullptr;    // 5 x cap
  int* d_nbr_count = nullptr;
  double* d_plane = nullptr;
  unsigned char* d_selected = nullptr;
  IekfCtrl* d_ctrl = nullptr;   // device-resident loop state of lii_iekf_update
  PoseArg* d_pose = nullptr;    // pose slot of the host-driven lii_iekf_iterate
  IekfCtrl* h_ctrl = nullptr;   // pinned upload image
  IekfResult* h_res = nullptr;  // pinned, device-mapped: written by the solve kernel of the stopping iteration
  lii_pose6d* h_poses = nullptr;  // pinned staging of the IMU pose table (lives behind h_ctrl: one upload can carry both)
  int update_seq = 0;           // IekfCtrl::seq of the last update (never 0)
  bool poll_result = true;      // LII_TEST=sync_result: end an update with hipStreamSynchronize instead of polling IekfResult::done
  bool poses_preloaded = false, ctrl_preloaded = false;  // lii_scan_register uploaded them already
  hipEvent_t ev_poses = nullptr;  // the last pose-table upload
  hipEvent_t ev_stage = nullptr;  // the last scan upload through h_stage
  bool host_solve = false;      // LII_TEST=host_solve: drive the loop from the host (A/B, reference arrangement)
  double* d_partials = nullptr;
  double* d_out91 = nullptr;
  unsigned long long* d_gran = nullptr;  // k_reduce_solve: the 91 sums of a pass on their way to the solver, 2 x 91 tagged words
  unsigned long long* d_extent = nullptr;  // 2 x {min (time|index), max time}: ping-pong accumulators
  unsigned int* d_mm = nullptr;           // 2 x {min xyz, max xyz} (order-preserving uints)
  int extent_sel = 0, mm_sel = 0;
  bool knn_plan = true;        // LII_KNN_PLAN=0: every k-NN launch is enqueued (IekfCtrl::plan_mask)
  bool test_pred_small = false;
  bool test_force_rebuild = false;  // LII_TEST=force_rebuild: every in-place map update rebuilds the index first (the branch a map low on room takes)
  bool no_fast_prologue = false;  // LII_TEST=no_fast: a time-sorted scan takes the general path as well (k_time_extent in front of the de-skew)
  bool no_fuse = false;        // LII_TEST=no_fuse: lii_scan_register keeps the de-skew and the voxel filter's insert in separate launches
  bool use_graph = false;      // LII_TEST=graph: the passes of an update are captured once per (cloud bound, plan, map view) and replayed
  std::map<std::string, hipGraphExec_t> graphs;
  int plan_passes_prev = 32;   // passes the update before the last one ran (the plan enqueues the larger of the last two)
  int knn_plan_force = -1;     // LII_TEST=plan_force=<mask>: use this plan for every update (tests: forces the parked path)
  unsigned int plan_next = 0xFFFFFFFFu, plan_cur = 0xFFFFFFFFu;
  long long map_repeats = 0;   // map updates repeated because a list outgrew its predicted size
  long long plan_parked = 0;   // updates that had to be continued by the host
  bool staging_busy = false;  // h_ctrl / h_poses were handed to the device by lii_scan_register and no wait has covered the read yet
  size_t ctrl_pending = 0;    // bytes of h_ctrl (+ poses) the next k_time_extent launch carries to d_ctrl; 0 = nothing pending
  bool extent_valid = false;  // d_extent[extent_sel] holds the time extent of d_scan (lii_scan_set_device computed it on the way)
  unsigned int* d_bbox_rows = nullptr;  // one row per de-skew workgroup: bounding box of its output points
  int bbox_rows = 0;                    // rows valid for the current d_scan (0: the voxel filter makes its own pass)
  unsigned long long *d_vkeys_a = nullptr, *d_vkeys_b = nullptr;  // sort keys of the voxel filter (kVoxKeyBits wide)
  unsigned int* d_vidx_b = nullptr;
  unsigned long long *d_vcomp = nullptr, *d_vsplit = nullptr;  // sample sort of the voxel filter (lii_vsort.hip)
  unsigned int* d_vhist = nullptr;
  unsigned short* d_vbucket = nullptr;
  unsigned int *d_vpcl_in = nullptr, *d_vpcl_out = nullptr;  // PCL voxel index per input point / per output voxel
  VoxelHashBuffers vh = {};      // the voxel grid by hashing (the default; LII_VOXEL_FILTER=sort: the sample sort)
  unsigned int vh_epoch = 0;     // number of the last hashed filter run (VoxelHashBuffers::counts)
  bool voxel_sort = false;       // LII_VOXEL_FILTER=sort
  bool vh_pinned = false;        // LII_VOXEL_FILTER=hash: no probing
  float fuse_leaf = 0.f;         // lii_scan_register -> lii_undistort_imu: the voxel filter that follows runs at this leaf (0: none)
  float vh_inserted_leaf = 0.f;
  bool vh_inserted = false;      // ... and the de-skew has filled the hashed filter's table on the way (lii_downsample goes on from there)
  int vh_mode = 1;               // 1: sparse voxels (hashed filter), 0: crowded voxels (sample sort)
  float vh_leaf = -1.f;          // the leaf size the choice was probed for
  unsigned int vh_watch = 0;
  unsigned long long vh_calls = 0, vh_due = 0;  // filter runs so far; the run at which the pending `crowded` read-back is applied
  bool voxel_path_hash = false;  // the path the last filter took
  unsigned int* h_vh_crowded = nullptr;  // pinned: VoxelHashBuffers::crowded of the last hashed filter (read lazily)
  hipEvent_t ev_vh = nullptr;
  bool vh_flag_pending = false;
  bool body_reordered = false;   // d_body is in the order of the voxels' first points: the download entry points restore the PCL order (pcl_perm)
  std::vector<int> pcl_perm;     // pcl_perm[r] = position in d_body of the r-th point in PCL order (valid while pcl_perm_valid)
  bool pcl_perm_valid = false;
  double* d_poses = nullptr;
  int n_scan = 0, n_body = 0;   // n_body is an upper bound while n_body_pending (the exact count lives in d_nbody)
  bool n_body_pending = false;
  int last_filtered = 1;
  int* d_nbody = nullptr;       // [0] size of the down-sampled cloud, [1] `filtered` flag of the last voxel filter
  bool body_is_scan = false;
  bool have_search = false;
  int knn_variant = 0;   // search pass: 0 = packed keys (k_knn_pk); 5 = exact lists throughout (k_knn_exact, its reference form) -
                         // LII_KNN_VARIANT selects
  hipStream_t map_stream = nullptr;  // the in-place update of lii_map_incremental runs here, beside the next scan's pre-processing
  bool map_async = false;            // ... and may still be running (map_join waits for it: ev_mapflag is its last packet)
  int bound_add = 0, bound_nodown = 0;  // ... the sizes the update in flight was enqueued for
  int list_hist[8][2] = {};             // ... from the sizes of the last eight calls (note_list_sizes)
  int list_hist_n = 0;
  int pred_add = -1, pred_nodown = -1;  // lii_map_incremental: list sizes the next update is enqueued for (< 0: none yet)
  bool lists_predicted = false;         // the update in flight ran on predicted sizes: commit_map checks it against the exact ones
  hipEvent_t ev_lists = nullptr;        // the two lists are complete (compute stream -> map stream)
  int* h_mapflag = nullptr;       // pinned, behind the last in-place update: [0..15] the map counters, [16..20] the list counts of
                                  // lii_map_incremental (k_compact_lists) - read by commit_map / map_join
  hipEvent_t ev_mapflag = nullptr;
  bool map_flag_pending = false;
  bool diag = false;     // LII_DIAG=1: counters of the rare paths on stderr when the handle is destroyed

  // ---- pinned staging
  float4* h_stage = nullptr;     // max(max_scan, max_map) float4
  size_t h_stage_elems = 0;
  double* h_small = nullptr;     // 4096 doubles

  // ---- calibration
  double *d_cal_imu = nullptr, *d_cal_lidar = nullptr, *d_cal_params = nullptr, *d_cal_out = nullptr;
  int n_cal = 0;

  void* ingest = nullptr;  // lii_ingest.hip state (frames of the last driver message)
  bool li_init_device = false;  // lii_li_init_set_device: zero-phase filter + cross-correlation of lii_li_init_run on the device

  // ---- comm
  ncclComm_t comm = nullptr;   // RCCL transport (ranks on several nodes, or forced)
  MailboxHost mailbox;         // node-local transport: the exchange happens inside k_reduce_solve
  unsigned long long* d_mb_seq = nullptr;
  long long mailbox_timeout_ticks = 3000000000ll;  // 30 s (LII_MAILBOX_TIMEOUT_S): ranks may start a scan seconds apart
  int n_ranks = 1, rank = 0;
  std::string comm_why;           // which transport this rank ended up with and why (lii_comm_describe)
  bool library_partition = true;  // lii_comm_set_partition: the library splits the down-sampled cloud over the ranks (every rank
                                  // hands over the whole scan); false: the caller hands every rank its own points

  // ---- profiling
  bool kp_active = false;            // inside a lii_scan_register that is being profiled launch by launch
  int prof_mode = 0;                 // the last lii_set_profiling value; 3: an event in front of every launch of lii_scan_register
  std::vector<hipEvent_t> kp_ev;     // ... the events (created on demand, reused),
  std::vector<int> kp_kind;          // ... kind * 64 + iteration of the launch behind each (kind LII_KP_KINDS: end mark)
  int kp_n = 0;
  lii_kernel_profile kprof{};
  bool profiling = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_it[32] = {};  // per-iteration brackets of the k-NN kernel in the device-driven loop
  double timings[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double host_us[6] = {0, 0, 0, 0, 0, 0};  // LII_DIAG: per lii_scan_register - entry -> first launch, -> pre-processing enqueued, -> loop enqueued, -> result; calls; gap between calls
  std::chrono::steady_clock::time_point host_last_return;
  double host_loop_enq_us = 0;
};

namespace {

thread_local std::string g_err;

int fail(lii_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  g_err = msg;
  return code;
}
// lii_set_profiling(h, 3): an event in front of the launch(es) that follow; `it` = the iteration of a loop launch
int kp_mark(lii_handle h, int kind, int it = 0) {
  if (h->prof_mode != 3) return LII_OK;
  if (h->kp_n >= (int)h->kp_ev.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return fail(h, LII_ERR_HIP, "hipEventCreate (kernel profile)");
    h->kp_ev.push_back(e);
    h->kp_kind.push_back(0);
  }
  if (hipEventRecord(h->kp_ev[size_t(h->kp_n)], h->stream) != hipSuccess) return fail(h, LII_ERR_HIP, "hipEventRecord (kernel profile)");
  h->kp_kind[size_t(h->kp_n)] = kind * 64 + std::min(it, 63);
  h->kp_n++;
  return LII_OK;
}
#define HIPCHK(h, call)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      return fail(h, LII_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                    \
  } while (0)

template <class T>
hipError_t dmalloc(T** p, size_t n) {
  return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T));
}
unsigned int next_pow2(unsigned int v) {
  unsigned int p = 1;
  while (p < v) p <<= 1;
  return p;
}

GridView grid_view(const lii_context* c) {
  GridView g;
  g.pts = c->d_pts;
  g.blocks = c->d_blocks;
  g.cells = c->d_cells;
  g.block_mask = c->block_mask;
  g.n_pts = c->map_dirty ? std::max(c->n_map, 1) : c->n_map;  // (an update in flight may have put the first points in)
  g.cs = c->cell_size;
  g.inv_cs = 1.0f / c->cell_size;
  g.max_d2 = c->cfg.max_match_dist2;
  return g;
}
RegistrationBuffers reg_buffers(const lii_context* c) {
  RegistrationBuffers rb;
  rb.body = c->d_body;
  rb.world = c->d_world;
  rb.nbr = c->d_nbr;
  rb.nbr_count = c->d_nbr_count;
  rb.plane = c->d_plane;
  rb.selected = c->d_selected;
  rb.partials = c->d_partials;
  rb.partial_stride = c->partial_stride;
  rb.n = c->n_body;
  rb.n_dev = c->n_body_pending ? c->d_nbody : nullptr;
  rb.cap = c->cfg.max_scan_points;
  rb.shard_rank = c->rank;
  rb.shard_world = (c->n_ranks > 1 && c->library_partition) ? c->n_ranks : 1;
  return rb;
}
PoseArg pose_of(const lii_state& s) {
  PoseArg p;
  std::memcpy(p.R, s.rot_end, 72);
  std::memcpy(p.p, s.pos_end, 24);
  std::memcpy(p.RLI, s.offset_R_L_I, 72);
  std::memcpy(p.TLI, s.offset_T_L_I, 24);
  return p;
}

// Builds the device map from n float4 points in d_map_unsorted:
// key (block | local cell) -> radix sort -> gather (d_map: cell-sorted, compact) -> block ids by scan -> per-block cell tables +
// block table -> capacities with slack -> scan -> spread into d_pts (the live array) + cell_cap.  Room for `extra_blocks` more
// 8x8x8 blocks is provisioned in the tables (in-place updates create blocks without a rebuild).
int build_index(lii_handle h, int n, int extra_blocks = 0) {
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
                    static_cast<void*>(h->d_cs_b)})
      if (q) HIPCHK(h, hipFree(q));
    h->d_cells = nullptr; h->d_cell_cap = nullptr; h->d_tp = nullptr; h->d_cs_a = nullptr; h->d_cs_b = nullptr;
    // (+ 1: the last table of the pool is the shared all-empty one, k_ins_cells)
    const size_t want = h->map_tight ? want_blocks + 1 : std::max<size_t>(std::max<size_t>(want_blocks * 2, h->cells_cap_blocks), 4096);
    HIPCHK(h, dmalloc(&h->d_cells, want * 512));
    HIPCHK(h, dmalloc(&h->d_cell_cap, want * 512));
    HIPCHK(h, dmalloc(&h->d_tp, want * 512));
    HIPCHK(h, dmalloc(&h->d_cs_a, want * 512));
    HIPCHK(h, dmalloc(&h->d_cs_b, want * 512));
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
    launch_cells_fill(h->d_keys_b, ranks, n, h->d_blocks, h->block_mask, h->d_cells, s);
    const int ne = int(n_blocks) * 512;
    unsigned int* caps = h->d_cs_a;
    unsigned int* capsum = h->d_cs_b;
    launch_cell_caps(h->d_cells, ne, caps, s);
    inclusive_scan_u32(h->d_sort_temp, h->sort_temp_bytes, caps, capsum, ne, s);
    launch_spread(h->d_map, h->d_cells, h->d_cell_cap, caps, capsum, ne, h->d_pts, h->d_mapctr, n, int(n_blocks), s);
  }
  HIPCHK(h, hipGetLastError());
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
int map_counters(lii_handle h, bool already_synced);
int map_rebuild(lii_handle h, int extra_blocks);
int map_apply(lii_handle h, const float4* list, int n_list, bool downsample, const float4* extra, int n_extra, bool beside = false,
              const int* n_list_dev = nullptr, const int* n_extra_dev = nullptr, bool count_events = true);
int map_counters(lii_handle h, bool already_synced = false);
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
int map_join(lii_handle h) {
  // (an update enqueued for predicted sizes is settled here whichever stream it ran on: when map_apply had to rebuild the index
  // first, the update went onto the handle's own stream - map_async false - and its list sizes need the same check)
  if (!h->map_async && !h->lists_predicted) return LII_OK;
  h->map_async = false;
  HIPCHK(h, hipEventSynchronize(h->ev_mapflag));
  if (h->lists_predicted) {
    // lii_map_incremental enqueued this update for predicted list sizes.  The exact ones came along behind it: they feed the
    // next prediction, and an update whose lists outgrew their bounds did nothing (k_compact_lists emptied them) - it is
    // repeated now, with the exact sizes (the lists themselves are untouched until the next lii_map_incremental).
    h->lists_predicted = false;
    const int ca = h->h_mapflag[kMapCtrWords], cn = h->h_mapflag[kMapCtrWords + 1];
    note_list_sizes(h, ca, cn);
    if (h->h_mapflag[kMapCtrWords + 2]) {
      h->map_repeats++;
      if (h->diag && h->map_repeats <= 8)
        std::fprintf(stderr, "[libliinit_hip] map update repeated: lists of %d / %d points, enqueued for %d / %d\n", ca, cn, h->bound_add, h->bound_nodown);
      h->map_flag_pending = false;  // (of the update that did nothing)
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
  HIPCHK(h, hipEventSynchronize(h->ev_mapflag));
  h->map_flag_pending = false;
  if (h->h_mapflag[kMapCtrOverflow] != 0) {  // ran out of provisioned room: rebuild + re-insertion of the parked points
    const int rc = map_counters(h, false);
    if (rc != LII_OK) return rc;
  } else {  // the counters came along: the host's copies are current again without a read of their own
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
int map_apply(lii_handle h, const float4* list, int n_list, bool downsample, const float4* extra, int n_extra, bool beside,
              const int* n_list_dev, const int* n_extra_dev, bool count_events) {
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
  if (downsample && n_list > 0 && !count_events && !h->fold_sorted && n_list <= h->cfg.max_scan_points) {
    launch_add_fold_hashed(list, n_list, n_list_dev, h->ds, g, h->d_ah_key, h->d_ah_best, h->d_ah_slot, h->d_tomb, h->d_ins, h->d_u32_a,
                           reinterpret_cast<unsigned int*>(h->d_mapctr + kMapCtrEvents), h->d_tp, h->d_work, h->d_mapctr, h->work_cap, s);
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
  launch_ins_cells(list_a, flags_a, n_list, flags_a ? nullptr : n_list_dev, extra, n_extra, n_extra_dev, h->d_ins_e2, h->d_blocks, h->block_mask, g.inv_cs, tables_cap, h->d_ins_e, h->d_tp,
                   h->d_work, h->d_mapctr, h->work_cap, h->d_dropped, h->drop_cap, s);
  launch_cell_apply(h->d_work, h->d_cells, h->d_cell_cap, h->d_pts, h->d_tomb, h->d_tp, h->d_mapctr, h->pts_cap_eff, (int)work_need, s);
  launch_ins_write(list_a, h->d_ins_e, n_list, flags_a ? nullptr : n_list_dev, extra, h->d_ins_e2, n_extra, n_extra_dev, h->d_cells, h->d_cell_cap, h->d_pts, h->d_mapctr, h->d_dropped,
                   h->drop_cap, s);
  HIPCHK(h, hipGetLastError());
  // the update's overflow flag travels to the host behind its kernels (see commit_map)
  // (ONE copy: the map counters and the list counts of lii_map_incremental sit behind each other - every small copy is a blit
  // kernel of ~5 us on this stream)
  HIPCHK(h, hipMemcpyAsync(h->h_mapflag, h->d_mapctr, sizeof(int) * (kMapCtrWords + 8), hipMemcpyDeviceToHost, s));
  h->lists_predicted = n_list_dev != nullptr;
  HIPCHK(h, hipEventRecord(h->ev_mapflag, s));
  h->map_flag_pending = true;
  h->map_async = beside;
  return LII_OK;
}

void launch_knn(lii_handle h, const GridView& g, const RegistrationBuffers& rb, const PoseArg& ps, const PoseArg* pose, int forced) {
  lii::launch_knn(h->knn_variant, g, rb, ps, pose, h->d_ctrl, forced, h->d_ctrl->search_pose, h->stream);
}

// Fetches the exact size of the down-sampled cloud from the device (one small synchronising copy).
int resolve_n_body(lii_handle h) {
  if (!h->n_body_pending) return LII_OK;
  HIPCHK(h, hipMemcpyAsync(h->h_small + 2048, h->d_nbody, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int v[2];
  std::memcpy(v, h->h_small + 2048, sizeof(v));
  h->n_body = v[0];
  h->last_filtered = v[1];
  h->n_body_pending = false;
  return LII_OK;
}

// The down-sampled cloud lives on the device in the order of the voxels' first points (k_vhash_emit); the reference's filter emits it
// ascending in the PCL voxel index.  Every entry point that hands per-point data of the down-sampled cloud to the host
// (lii_scan_download 1 / 2, lii_neighbors_download) restores that order: perm[r] = device position of the r-th point in PCL
// order, from the PCL index kept per output voxel (distinct per voxel).  Host work, off the per-scan path.
int pcl_order(lii_handle h, const int** perm) {
  *perm = nullptr;
  if (!h->body_reordered) return LII_OK;
  int rc = resolve_n_body(h);
  if (rc != LII_OK) return rc;
  if (!h->pcl_perm_valid) {
    const int n = h->n_body;
    std::vector<unsigned int> keys(size_t(std::max(n, 1)));
    if (n > 0) {
      HIPCHK(h, hipMemcpyAsync(h->h_stage, h->d_vpcl_out, sizeof(unsigned int) * size_t(n), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      std::memcpy(keys.data(), h->h_stage, sizeof(unsigned int) * size_t(n));
    }
    h->pcl_perm.resize(size_t(n));
    for (int i = 0; i < n; i++) h->pcl_perm[size_t(i)] = i;
    std::sort(h->pcl_perm.begin(), h->pcl_perm.end(), [&](int a, int b) { return keys[size_t(a)] < keys[size_t(b)]; });
    h->pcl_perm_valid = true;
  }
  *perm = h->pcl_perm.data();
  return LII_OK;
}

// The time extent of a scan ping-pongs between two accumulators (whoever fills one re-arms the other for the next scan).
// Returns the accumulator that holds it, launching the reduction unless the scan's arrival already produced it.
void extent_discard(lii_handle h) {
  if (h->extent_valid) h->extent_sel ^= 1;
  h->extent_valid = false;
}
unsigned long long* extent_of_scan(lii_handle h) {
  unsigned long long* ext = h->d_extent + 2 * h->extent_sel;
  if (!h->extent_valid) {
    launch_time_extent(h->d_scan, h->n_scan, ext, h->d_extent + 2 * (h->extent_sel ^ 1), nullptr, h->h_ctrl, h->d_ctrl, h->ctrl_pending,
                       h->stream);
    h->ctrl_pending = 0;
  }
  h->extent_sel ^= 1;  // consumed: the partner (re-armed by whoever filled `ext`) serves the next scan
  h->extent_valid = false;
  return ext;
}
MailboxView mailbox_view(lii_handle h) {
  MailboxView v;
  v.slots = h->mailbox.dev_slots;
  v.peers = h->mailbox.d_peers;
  v.seq = h->d_mb_seq;
  v.n_ranks = h->n_ranks;
  v.rank = h->rank;
  v.timeout_ticks = h->mailbox_timeout_ticks;
  return v;
}

int iterate(lii_handle h, const lii_state* st, bool search, bool imu_en, double* out91) {
  if (h->n_body <= 0) return fail(h, LII_ERR_STATE, "no down-sampled scan (call lii_downsample / lii_downsample_skip)");
  int rc = commit_map(h);
  if (rc != LII_OK) return rc;
  if (!search && !h->have_search) return fail(h, LII_ERR_STATE, "non-search iteration before any search");
  GridView g = grid_view(h);
  RegistrationBuffers rb = reg_buffers(h);
  const bool prof = h->profiling;
  if (prof) HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  const PoseArg ps = pose_of(*st);
  if (search) launch_knn(h, g, rb, ps, h->d_pose, 1);
  if (prof) HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
  launch_fit_reduce(g, rb, ps, h->d_pose, h->d_ctrl, search ? 1 : 0, imu_en ? 1 : 0, h->cfg.plane_threshold,
                    h->cfg.laser_point_cov_inv, h->stream);
  if (prof) HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
  launch_reduce91(rb, h->d_out91, h->d_ctrl, 1, h->stream);
  if (prof) HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
  if (search) h->have_search = true;
  if (h->comm) {
    ncclResult_t r = ncclAllReduce(h->d_out91, h->d_out91, kNormalEq, ncclDouble, ncclSum, h->comm, h->stream);
    if (r != ncclSuccess) return fail(h, LII_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
  } else if (h->mailbox.dev_slots) {
    launch_mailbox_allreduce(h->d_out91, mailbox_view(h), h->stream);
  }
  HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_out91, sizeof(double) * kNormalEq, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::memcpy(out91, h->h_small, sizeof(double) * kNormalEq);
  if (h->mailbox.dev_slots && out91[kNormalEq - 1] != out91[kNormalEq - 1])
    return fail(h, LII_ERR_COMM, "mailbox exchange timed out (a rank of the job did not reach this pass)");
  if (prof) {
    // timings: [0] sum ms search-pass kernel, [1] sum ms residual-pass kernel, [2] sum ms reduce kernel,
    //          [3] host solve ms (last update), [4] total ms (last update), [5]/[6] launch counts of [0]/[1]
    float a = 0, b = 0;
    HIPCHK(h, hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
    HIPCHK(h, hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
    if (search) {
      float k = 0;
      HIPCHK(h, hipEventElapsedTime(&k, h->ev[0], h->ev[3]));
      h->timings[0] += a; h->timings[5] += 1; h->timings[7] += k;  // [7]: the k-NN kernel alone
    } else { h->timings[1] += a; h->timings[6] += 1; }
    h->timings[2] += b;
  }
  return LII_OK;
}

// The whole iterated update enqueued once: prologue (P^-1), then max_iterations x {k-NN, fallback, fit+reduce,
// final reduce, 24-state solve}; every kernel consults the device-resident control block and returns at once when
// its pass is not due (no search scheduled / loop already stopped).  One synchronisation at the end.
void fill_ctrl(lii_handle h, const lii_state* state, const lii_state* state_prop, const lii_iekf_opts* opts) {
  IekfCtrl* hc = h->h_ctrl;
  std::memcpy(hc->st, state, sizeof(lii_state));
  std::memcpy(hc->prop, state_prop, sizeof(hc->prop));
  hc->max_it = opts->max_iterations;
  hc->imu_en = opts->imu_en;
  hc->it = 0; hc->search_next = 1; hc->stop = 0; hc->rematch_num = 0; hc->converged = 0; hc->searches = 0;
  hc->effect_num = 0; hc->singular = 0;
  h->update_seq = h->update_seq == 0x7FFFFFFF ? 1 : h->update_seq + 1;
  hc->seq = h->update_seq;
  // which k-NN launches ride along (IekfCtrl::plan_mask): the first pass always; the others as the previous update needed them
  unsigned int plan = 0xFFFFFFFFu;
  if (h->knn_plan && !h->comm) {
    plan = (h->knn_plan_force >= 0 ? ((unsigned int)h->knn_plan_force | 0xFFFF0000u) : h->plan_next) | 0x00010001u;
  }
  hc->plan_mask = plan;
  h->plan_cur = plan;
}

int update_on_device(lii_handle h, lii_state* state, const lii_state* state_prop, const lii_iekf_opts* opts,
                     lii_iekf_report* report) {
  static_assert(sizeof(lii_state) == sizeof(double) * kStateDoubles, "lii_state layout");
  if (h->n_body <= 0) return fail(h, LII_ERR_STATE, "no down-sampled scan (call lii_downsample / lii_downsample_skip)");
  int rc = commit_map(h);
  if (rc != LII_OK) return rc;
  hipStream_t s = h->stream;
  // The stream lii_map_incremental leaves its update on is created with the first update of a handle that is NOT a rank of a
  // sharded job (those never update beside a scan).  Not in lii_create: a second compute queue per process - even one whose
  // stream has been destroyed again: the runtime keeps the hardware queue - makes several processes on one device oversubscribe
  // the hardware queues, and a kernel that waits for a peer's kernel (the mailbox) then waits for a time slice: the one-device
  // rehearsal of a 2-rank job fell from 4 000 to 1 350 scans/s.
  if (!h->map_stream && h->n_ranks <= 1) HIPCHK(h, hipStreamCreateWithFlags(&h->map_stream, hipStreamNonBlocking));
  if (!h->ctrl_preloaded) {
    if (h->staging_busy) HIPCHK(h, hipStreamSynchronize(s));  // a lii_scan_register that failed half way left the buffer in use
    fill_ctrl(h, state, state_prop, opts);
    HIPCHK(h, hipMemcpyAsync(h->d_ctrl, h->h_ctrl, sizeof(IekfCtrl), hipMemcpyHostToDevice, s));
  }
  h->ctrl_preloaded = false;
  h->h_res->singular = 0;
  h->h_res->it = -1;  // overwritten by the stopping iteration
  GridView g = grid_view(h);
  RegistrationBuffers rb = reg_buffers(h);
  const PoseArg* pose = reinterpret_cast<const PoseArg*>(h->d_ctrl);  // first 24 doubles of IekfCtrl::st
  const PoseArg ps0 = pose_of(*state);  // unused by the device-driven kernels (they read `pose`)
  const bool prof = h->profiling && h->prof_mode != 3;  // (mode 3 brackets every launch itself: kp_mark)
  const double* ne = h->comm ? h->d_out91 + 128 : h->d_out91;
  unsigned int plan = h->plan_cur;  // fill_ctrl chose it (the control block on the device carries the same mask)
  const unsigned int plan0 = plan;
  // (profiling = HIP events around the k-NN launches only - the dominant kernel, lii_last_timings [5] / [7]; every event is a
  // barrier packet on the stream, so the rest of the loop is left alone: launch plan and result polling work as always)
  auto enqueue_pass = [&](int it) -> int {
    const bool knn = it >= 16 || ((plan >> it) & 1u);
    if (knn) {
      if (prof && it < 16) HIPCHK(h, hipEventRecord(h->ev_it[2 * it], s));
      if (h->kp_active) { const int r = kp_mark(h, LII_KP_KNN, it); if (r != LII_OK) return r; }
      launch_knn(h, g, rb, ps0, pose, -1);
      if (prof && it < 16) HIPCHK(h, hipEventRecord(h->ev_it[2 * it + 1], s));
    }
    if (h->kp_active) { const int r = kp_mark(h, LII_KP_FIT, it); if (r != LII_OK) return r; }
    launch_fit_reduce(g, rb, ps0, pose, h->d_ctrl, -1, opts->imu_en ? 1 : 0, h->cfg.plane_threshold, h->cfg.laser_point_cov_inv, s);
    if (h->kp_active) { const int r = kp_mark(h, LII_KP_SOLVE, it); if (r != LII_OK) return r; }
    if (!h->comm) {  // single GPU or node-local mailbox: final sum (+ exchange) and solve in one launch
      launch_reduce_solve(rb, h->d_gran, h->d_ctrl, h->h_res, mailbox_view(h), s);
      return LII_OK;
    }
    launch_reduce91(rb, h->d_out91, h->d_ctrl, -1, s);
    // every rank enqueues the same number of all-reduces; a pass that is skipped on the device re-sums the
    // unchanged local buffer on all ranks alike, so the ranks stay in lock-step without a host decision
    ncclResult_t r = ncclAllReduce(h->d_out91, h->d_out91 + 128, kNormalEq, ncclDouble, ncclSum, h->comm, s);
    if (r != ncclSuccess) return fail(h, LII_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    launch_iekf_solve(h->d_ctrl, ne, h->h_res, s);
    return LII_OK;
  };
  const auto t_loop0 = std::chrono::steady_clock::now();
  auto enqueue_planned = [&]() -> int {
    for (int it = 0; it < opts->max_iterations; it++) {
      if (it < 16 && !((plan >> (16 + it)) & 1u)) break;  // the plan ends here
      const int r = enqueue_pass(it);
      if (r != LII_OK) return r;
    }
    return LII_OK;
  };
  if (h->use_graph && !h->comm && !h->profiling) {
    // The same launches, captured once and replayed (hipGraphLaunch): every kernel argument of the loop is a device pointer or
    // a constant of the configuration, except the bound of the cloud size (rounded up here: the kernels take the exact size
    // from the device), the plan and the view of the map - the key of the cache.  Measured against the plain launches in
    // profiles/r03_hipgraph_ab.md.
    if (rb.n_dev) rb.n = std::min(rb.cap, (rb.n + 4095) & ~4095);
    struct { const void* p[4]; unsigned int mask; int n_pts, n, plan, max_it, imu_en, variant, shard; float cs; } kv;
    std::memset(&kv, 0, sizeof(kv));
    kv.p[0] = g.pts; kv.p[1] = g.blocks; kv.p[2] = g.cells; kv.p[3] = rb.n_dev;
    kv.mask = g.block_mask; kv.n_pts = g.n_pts; kv.n = rb.n; kv.plan = (int)plan; kv.max_it = opts->max_iterations;
    kv.imu_en = opts->imu_en ? 1 : 0; kv.variant = h->knn_variant; kv.shard = rb.shard_world * 4096 + rb.shard_rank; kv.cs = g.cs;
    const std::string key(reinterpret_cast<const char*>(&kv), sizeof(kv));
    auto f = h->graphs.find(key);
    if (f == h->graphs.end()) {
      if (h->graphs.size() >= 64) {
        for (auto& e : h->graphs) (void)hipGraphExecDestroy(e.second);
        h->graphs.clear();
      }
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      HIPCHK(h, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      rc = enqueue_planned();
      const hipError_t e_end = hipStreamEndCapture(s, &graph);
      if (rc != LII_OK) return rc;
      HIPCHK(h, e_end);
      HIPCHK(h, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      HIPCHK(h, hipGraphDestroy(graph));
      f = h->graphs.emplace(key, exec).first;
    }
    HIPCHK(h, hipGraphLaunch(f->second, s));
  } else {
    rc = enqueue_planned();
    if (rc != LII_OK) return rc;
  }
  // The iteration that stops the loop writes the result block (mapped host memory) and then its sequence number.  Polling
  // that word instead of synchronising the stream returns as soon as the result exists: the passes enqueued behind the
  // stopping one (they only read `stop` and return) drain while the caller already prepares the next scan.
  // The plan also ends the enqueued loop after as many passes as the last updates ran: the launches behind the stopping pass
  // only drain (3 x 4.5 us on stream100k, about what the host needs to come back with the next scan: + 0 .. 3 % scans/s,
  // gpurun_out/r3x4).  A loop that parked itself (the next pass is not there, or needs a search the plan did not hold - the
  // pattern changed against the previous scans) is continued from here with every launch: one host round trip, on those scans.
  auto wait_result = [&](bool first) -> int {
    const int parked_word = first ? (h->update_seq | kLoopParked) : h->update_seq;
    if (h->poll_result && !h->comm) {
      volatile int* done = &h->h_res->done;
      unsigned int spins = 0;
      while (*done != h->update_seq && *done != parked_word) {
        if ((++spins & 0x3FFF) == 0) {  // every ~50 us: is the stream still alive?
          const hipError_t q = hipStreamQuery(s);
          if (q == hipSuccess) break;  // everything ran; `done` is final (a loop that never stopped is reported below)
          if (q != hipErrorNotReady) return fail(h, LII_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        }
        __builtin_ia32_pause();
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (*done != h->update_seq && *done != parked_word) HIPCHK(h, hipStreamSynchronize(s));
    } else {
      HIPCHK(h, hipStreamSynchronize(s));  // the stopping iteration's solve has written h_res (mapped host memory)
    }
    return LII_OK;
  };
  if (h->diag) h->host_loop_enq_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_loop0).count();
  if (h->kp_active) { rc = kp_mark(h, LII_KP_KINDS); if (rc != LII_OK) return rc; }  // (end mark of the planned passes)
  rc = wait_result(true);
  if (rc != LII_OK) return rc;
  if (h->h_res->done == (h->update_seq | kLoopParked)) {
    const int from = h->h_res->parked_it;
    h->plan_parked++;
    plan = 0xFFFFFFFFu;
    launch_loop_resume(h->d_ctrl, s);
    for (int it = from; it < opts->max_iterations; it++) {
      rc = enqueue_pass(it);
      if (rc != LII_OK) return rc;
    }
    if (h->kp_active) { rc = kp_mark(h, LII_KP_KINDS); if (rc != LII_OK) return rc; }
    rc = wait_result(false);
    if (rc != LII_OK) return rc;
  }
  h->staging_busy = false;  // the wait above covers everything enqueued before the stopping pass
  const IekfResult* hr = h->h_res;
  h->have_search = true;
#ifdef LII_SOLVE_TRACE
  {
    static int cnt = 0;
    if (++cnt % 100 == 0) {
      auto row = [&](const long long* t) {
        fprintf(stderr, " loads+sums %lld | A %lld | elimination %lld | solution %lld | state %lld | cov %lld ;", t[1] - t[0], t[2] - t[1], t[4] - t[3], t[8] - t[4],
                t[9] - t[8], t[10] - t[9]);
      };
      fprintf(stderr, "[solve trace, 10 ns ticks] stopping pass:");
      row(hr->ts);
      fprintf(stderr, "  pass 0:");
      row(hr->ts0);
      fprintf(stderr, "\n");
    }
  }
#endif
  if (h->kp_active && h->kp_n > 1 && hr->it > 0) {
    // per-launch brackets (lii_set_profiling(h, 3)): the time from the event in front of a launch to the next event, for the
    // launches that executed (a pass the loop did not reach, or a k-NN launch whose pass did not search, only read a flag)
    HIPCHK(h, hipEventSynchronize(h->kp_ev[size_t(h->kp_n - 1)]));
    for (int i = 0; i + 1 < h->kp_n; i++) {
      int kind = h->kp_kind[size_t(i)] / 64;
      const int it = h->kp_kind[size_t(i)] % 64;
      if (kind >= LII_KP_KINDS) continue;
      const bool loop_kind = kind == LII_KP_KNN || kind == LII_KP_FIT || kind == LII_KP_SOLVE;
      if (loop_kind && (it >= hr->it || it >= 16)) continue;
      if (kind == LII_KP_KNN && !hr->search_log[it]) continue;
      if (kind == LII_KP_FIT && hr->search_log[it]) kind = LII_KP_FIT_SEARCH;
      float ms = 0;
      if (hipEventElapsedTime(&ms, h->kp_ev[size_t(i)], h->kp_ev[size_t(i + 1)]) != hipSuccess) continue;
      h->kprof.ms[kind] += ms;
      h->kprof.launches[kind] += 1;
    }
    h->kprof.scans += 1;
  }
  if (hr->singular == 3)
    return fail(h, LII_ERR_COMM, "mailbox exchange timed out (a rank of the job did not reach this pass); re-create the communicator");
  if (hr->singular) return fail(h, LII_ERR_INVALID, "singular covariance / normal matrix in the device solve");
  if (hr->it < 0) return fail(h, LII_ERR_HIP, "device loop ended without a result");
  std::memcpy(state, hr->st, sizeof(lii_state));
  {  // the next update's plan: this one's pattern; passes it did not reach keep their launch
    unsigned int next = 0xFFFFFFFFu;
    for (int q = 0; q < 16 && q < hr->it; q++)
      if (!hr->search_log[q]) next &= ~(1u << q);
    // ... and as many passes as the longer of the last two updates ran (a scan that needs more parks and is continued)
    for (int q = std::max(hr->it, h->plan_passes_prev); q < 16; q++) next &= ~(1u << (16 + q));
    h->plan_passes_prev = hr->it;
    h->plan_next = next;
  }
  if (report) {
    report->iterations = hr->it;
    report->searches = hr->searches;
    report->effect_num = hr->effect_num;
    report->converged = hr->converged;
    std::memcpy(report->normal_eq, hr->ne, sizeof(double) * kNormalEq);
  }
  if (prof) {
    // the k-NN kernel alone, over EVERY pass that actually searched (the device logs which iterations did); the events were
    // recorded ahead of the stopping pass, whose result has arrived: they have completed
    for (int it = 0; it < opts->max_iterations && it < 16; it++) {
      if (it >= hr->it || !hr->search_log[it] || !((plan0 >> it) & 1u)) continue;
      float kk = 0;
      if (hipEventElapsedTime(&kk, h->ev_it[2 * it], h->ev_it[2 * it + 1]) != hipSuccess) continue;
      h->timings[7] += kk;
      h->timings[5] += 1;
    }
  }
  return LII_OK;
}

}  // namespace

int lii_internal_fail(lii_context* h, int code, const std::string& msg) { return fail(h, code, msg); }
int lii_internal_li_init_on_device(lii_context* h) { return h && h->li_init_device ? 1 : 0; }
hipStream_t lii_internal_stream(lii_context* h) { return h->stream; }
void** lii_internal_ingest_slot(lii_context* h) { return &h->ingest; }

extern "C" {

int lii_abi_version(void) { return LII_ABI_VERSION; }

const char* lii_strerror(int status) {
  switch (status) {
    case LII_OK: return "ok";
    case LII_ERR_INVALID: return "invalid argument";
    case LII_ERR_NO_DEVICE: return "no usable HIP device (libliinit_hip has no CPU fallback)";
    case LII_ERR_HIP: return "HIP runtime error";
    case LII_ERR_CAPACITY: return "capacity exceeded";
    case LII_ERR_STATE: return "call order violated";
    case LII_ERR_COMM: return "RCCL error";
    default: return "unknown status";
  }
}
const char* lii_last_error(lii_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

int lii_device_count(int* count) {
  if (!count) return LII_ERR_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(nullptr, LII_ERR_NO_DEVICE, hipGetErrorString(e)); }
  *count = n;
  return LII_OK;
}

int lii_create(const lii_config* cfg, lii_handle* out) {
  if (!cfg || !out || cfg->struct_size != (int32_t)sizeof(lii_config)) return fail(nullptr, LII_ERR_INVALID, "bad lii_config");
  if (cfg->max_scan_points <= 0 || cfg->max_map_points <= 0) return fail(nullptr, LII_ERR_INVALID, "capacities must be > 0");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, LII_ERR_NO_DEVICE, "no HIP device visible; libliinit_hip has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, LII_ERR_INVALID, "device ordinal out of range");
  lii_handle h = new lii_context;
  h->cfg = *cfg;
  if (h->cfg.max_match_dist2 <= 0) h->cfg.max_match_dist2 = 5.0f;
  if (h->cfg.plane_threshold <= 0) h->cfg.plane_threshold = 0.1;
  if (h->cfg.laser_point_cov_inv <= 0) h->cfg.laser_point_cov_inv = 1000.0;
  if (h->cfg.map_downsample_size <= 0) h->cfg.map_downsample_size = 0.2f;
  h->cell_size = cfg->map_cell_size > 0 ? cfg->map_cell_size : 3.0f * h->cfg.map_downsample_size;
  // the 3x3x3 neighbourhood of 8x8x8-cell blocks must cover the acceptance radius sqrt(max_match_dist2)
  h->cell_size = std::max(h->cell_size, std::sqrt(h->cfg.max_match_dist2) / 8.0f * 1.001f);
  if (const char* v = std::getenv("LII_KNN_VARIANT")) h->knn_variant = std::atoi(v);  // A/B knob for profiling
  if (const char* v = std::getenv("LII_DIAG")) h->diag = std::atoi(v) != 0;
  if (const char* v = std::getenv("LII_VOXEL_FILTER")) {
    h->voxel_sort = std::string(v) == "sort";
    if (std::string(v) == "hash") { h->vh_pinned = true; h->vh_mode = 1; }
  }
  if (const char* v = std::getenv("LII_KNN_PLAN")) h->knn_plan = std::atoi(v) != 0;
  if (const char* v = std::getenv("LII_TEST")) {
    // arrangements the test-suite and the A/B measurements ask for, comma-separated: "map_tight" (an in-place map update without
    // spare room), "plan_force=<mask>" (a launch plan that is wrong on purpose), "host_solve" (the iteration loop driven from the
    // host around lii_iekf_iterate with the literal two-inversion algebra), "sync_result" (every update ends with
    // hipStreamSynchronize instead of polling the result word), "graph" (the enqueued passes of an update replayed from a
    // captured hipGraph), "pred_small" (lii_map_incremental predicts list sizes that are always too small), "fold_sort"
    // (lii_map_incremental folds its list through the batch sort, as lii_map_add_points does, instead of the hash table), "no_fuse"
    // (lii_scan_register keeps the de-skew and the insert of the hashed voxel filter in separate launches), "no_fast" (a
    // time-sorted scan takes the general path of lii_scan_register too: k_time_extent in front of the de-skew), "force_rebuild"
    // (every in-place map update takes the branch that rebuilds the index first)
    const std::string t(v);
    h->map_tight = t.find("map_tight") != std::string::npos;
    const size_t q = t.find("plan_force=");
    if (q != std::string::npos) h->knn_plan_force = int(std::strtol(t.c_str() + q + 11, nullptr, 0) & 0x7FFFFFFF);
    h->host_solve = t.find("host_solve") != std::string::npos;
    h->poll_result = t.find("sync_result") == std::string::npos;
    h->use_graph = t.find("graph") != std::string::npos;
    h->test_pred_small = t.find("pred_small") != std::string::npos;
    h->fold_sorted = t.find("fold_sort") != std::string::npos;
    h->no_fuse = t.find("no_fuse") != std::string::npos;
    h->no_fast_prologue = t.find("no_fast") != std::string::npos;
    h->test_force_rebuild = t.find("force_rebuild") != std::string::npos;
  }
  h->ds = h->cfg.map_downsample_size;
  h->device = cfg->device;
#define CK(call)                                                                  \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      int rc_ = fail(nullptr, LII_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
      lii_destroy(h);                                                             \
      return rc_;                                                                 \
    }                                                                             \
  } while (0)
  CK(hipSetDevice(h->device));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, h->device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
    int rc = fail(nullptr, LII_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    lii_destroy(h);
    return rc;
  }
  CK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t N = size_t(cfg->max_scan_points), M = size_t(cfg->max_map_points);
  const size_t NM = std::max(N, M);
  CK(dmalloc(&h->d_map_unsorted, M));
  CK(dmalloc(&h->d_map, M));
  h->pts_cap = (unsigned int)std::min<size_t>(3 * M + 65536, 0x7FFFFFF0u);
  CK(dmalloc(&h->d_pts, size_t(h->pts_cap)));
  h->work_cap = (unsigned int)std::min<size_t>(10 * NM + 4096, 0x7FFFFFF0u);
  CK(dmalloc(&h->d_work, size_t(h->work_cap)));
  CK(dmalloc(&h->d_ins_e, NM));
  CK(dmalloc(&h->d_ins_e2, NM));
  CK(dmalloc(&h->d_mapctr, kMapCtrWords + 8));  // (+ the list counts of lii_map_incremental: one copy brings both to the host)
  CK(hipMemset(h->d_mapctr, 0, sizeof(int) * (kMapCtrWords + 8)));
  h->d_counts = h->d_mapctr + kMapCtrWords;
  CK(dmalloc(&h->d_keys_a, M));
  CK(dmalloc(&h->d_keys_b, M));
  CK(dmalloc(&h->d_keys_c, M));
  CK(dmalloc(&h->d_idx_a, M));
  CK(dmalloc(&h->d_idx_b, M));
  h->blocks_cap = 4096;
  h->block_mask = h->blocks_cap - 1;
  CK(dmalloc(&h->d_blocks, size_t(h->blocks_cap)));
  h->cells_cap_blocks = std::max<size_t>(4096, M / 64);
  CK(dmalloc(&h->d_cells, h->cells_cap_blocks * 512));
  CK(dmalloc(&h->d_cell_cap, h->cells_cap_blocks * 512));
  CK(dmalloc(&h->d_tp, h->cells_cap_blocks * 512));
  CK(dmalloc(&h->d_cs_a, h->cells_cap_blocks * 512));
  CK(dmalloc(&h->d_cs_b, h->cells_cap_blocks * 512));
  CK(hipMemset(h->d_cells, 0, sizeof(uint2) * h->cells_cap_blocks * 512));
  CK(hipMemset(h->d_cell_cap, 0, sizeof(unsigned int) * h->cells_cap_blocks * 512));
  CK(hipMemset(h->d_tp, 0, sizeof(unsigned int) * h->cells_cap_blocks * 512));
  CK(dmalloc(&h->d_counter, 4));
  CK(dmalloc(&h->d_tomb, size_t(h->pts_cap)));
  CK(hipMemset(h->d_tomb, 0, size_t(h->pts_cap)));
  CK(dmalloc(&h->d_ins, M));
  CK(dmalloc(&h->d_batch, M));
  h->drop_cap = (unsigned int)NM;
  CK(dmalloc(&h->d_dropped, NM));
  CK(dmalloc(&h->d_ins_c, M));
  CK(dmalloc(&h->d_u32_a, NM));
  CK(dmalloc(&h->d_u32_b, NM));
  CK(dmalloc(&h->d_u32_c, NM));
  {
    const size_t slots = add_hash_slots(int(N));
    CK(dmalloc(&h->d_ah_key, slots)); CK(dmalloc(&h->d_ah_best, slots)); CK(dmalloc(&h->d_ah_slot, N));
    CK(hipMemset(h->d_ah_key, 0xFF, 8 * slots)); CK(hipMemset(h->d_ah_best, 0xFF, 8 * slots));
  }
  CK(dmalloc(&h->d_list_add, N));
  CK(dmalloc(&h->d_list_nodown, N));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->n_map_pinned), 64, hipHostMallocDefault));
  h->n_map_pinned[0] = 0;
  h->sort_temp_bytes = sort_temp_bytes(int(std::max<size_t>(NM, h->cells_cap_blocks * 512)));
  CK(hipMalloc(&h->d_sort_temp, h->sort_temp_bytes));
  CK(dmalloc(&h->d_scan, N));
  CK(dmalloc(&h->d_body, N));
  CK(dmalloc(&h->d_world, N));
  CK(dmalloc(&h->d_nbr, N * kMatch));
  CK(dmalloc(&h->d_nbr_count, N));
  CK(dmalloc(&h->d_plane, N * 4));
  CK(dmalloc(&h->d_selected, N));
  CK(dmalloc(&h->d_nbody, 4));
  {
    // control block and pose table share one allocation so that lii_scan_register uploads both with one copy
    void* p = nullptr;
    CK(hipMalloc(&p, kCtrlBytes + sizeof(lii_pose6d) * 1024 + 1024));
    h->d_ctrl = static_cast<IekfCtrl*>(p);
    h->d_poses = reinterpret_cast<double*>(static_cast<char*>(p) + kCtrlBytes);
  }
  CK(dmalloc(&h->d_pose, 1));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_ctrl), kCtrlBytes + sizeof(lii_pose6d) * 1024 + 1024, hipHostMallocDefault));
  h->h_poses = reinterpret_cast<lii_pose6d*>(reinterpret_cast<char*>(h->h_ctrl) + kCtrlBytes);
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_res), sizeof(IekfResult), hipHostMallocMapped));
  std::memset(h->h_res, 0, sizeof(IekfResult));
  CK(hipEventCreateWithFlags(&h->ev_poses, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&h->ev_mapflag, hipEventDisableTiming));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_mapflag), 256, hipHostMallocDefault));
  std::memset(h->h_mapflag, 0, 256);
  CK(hipEventCreateWithFlags(&h->ev_lists, hipEventDisableTiming));
  CK(hipMemset(h->d_counter, 0, 16));
  h->partial_stride = register_blocks(int(N)) + 8;
  CK(dmalloc(&h->d_partials, size_t(h->partial_stride) * kNormalEq));
  CK(dmalloc(&h->d_out91, 256));  // [0,91): local sums, [128,219): all-reduced sums (sharded scans)
  CK(dmalloc(&h->d_gran, 256));
  CK(hipMemset(h->d_gran, 0, 8 * 256));
  CK(dmalloc(&h->d_extent, 4));
  CK(dmalloc(&h->d_mm, 16));
  CK(dmalloc(&h->d_bbox_rows, (N / 256 + 2) * 8));
  {
    const unsigned long long e0[4] = {~0ull, 0ull, ~0ull, 0ull};
    const unsigned int m0[16] = {~0u, ~0u, ~0u, 0, 0, 0, 0, 0, ~0u, ~0u, ~0u, 0, 0, 0, 0, 0};
    CK(hipMemcpy(h->d_extent, e0, sizeof(e0), hipMemcpyHostToDevice));
    CK(hipMemcpy(h->d_mm, m0, sizeof(m0), hipMemcpyHostToDevice));
  }
  CK(dmalloc(&h->d_vkeys_a, N));
  CK(dmalloc(&h->d_vkeys_b, N));
  CK(dmalloc(&h->d_vidx_b, N));
  CK(dmalloc(&h->d_vcomp, N));
  CK(dmalloc(&h->d_vsplit, 2048 + 4096));  // splitters | samples
  CK(dmalloc(&h->d_vhist, voxel_sort_hist_elems((int)N)));
  CK(hipMemset(h->d_vhist, 0, sizeof(unsigned int) * voxel_sort_hist_elems((int)N)));
  CK(dmalloc(&h->d_vbucket, N));
  CK(dmalloc(&h->d_vpcl_in, N));
  CK(dmalloc(&h->d_vpcl_out, N));
  {
    const size_t slots = voxel_hash_slots((int)N);
    CK(hipMalloc(&h->vh.slots, 64 * slots));
    launch_voxel_hash_clear(h->vh, slots, h->stream);
    CK(dmalloc(&h->vh.slot_of, N)); CK(dmalloc(&h->vh.next, N));
    CK(dmalloc(&h->vh.counts, N / 256 + 8)); CK(hipMemset(h->vh.counts, 0, 8 * (N / 256 + 8)));
    CK(dmalloc(&h->vh.crowded, 4));
    CK(hipMemset(h->vh.crowded, 0, 16));
    CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_vh_crowded), 64, hipHostMallocDefault));
    *h->h_vh_crowded = 0;
    CK(hipEventCreateWithFlags(&h->ev_vh, hipEventDisableTiming));
  }
  CK(dmalloc(&h->d_cal_params, 64));
  CK(dmalloc(&h->d_cal_out, 128));
  h->h_stage_elems = NM * kMatch;  // large enough for the neighbour download too
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_stage), sizeof(float4) * h->h_stage_elems, hipHostMallocDefault));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_small), sizeof(double) * 32768, hipHostMallocDefault));
  for (int i = 0; i < 4; i++) CK(hipEventCreate(&h->ev[i]));
  for (int i = 0; i < 32; i++) CK(hipEventCreate(&h->ev_it[i]));
  launch_table_clear(h->d_blocks, h->blocks_cap, h->stream);
  CK(hipStreamSynchronize(h->stream));
#undef CK
  *out = h;
  return LII_OK;
}

int lii_destroy(lii_handle h) {
  if (!h) return LII_OK;
  (void)hipSetDevice(h->device);
  if (h->comm) ncclCommDestroy(h->comm);
  if (h->map_stream) (void)hipStreamSynchronize(h->map_stream);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->diag) std::fprintf(stderr, "[libliinit_hip] map updates completed by a rebuild + re-insertion: %lld\n", h->map_recoveries);
  if (h->diag) std::fprintf(stderr, "[libliinit_hip] updates continued by the host after a parked loop: %lld\n", h->plan_parked);
  if (h->diag) std::fprintf(stderr, "[libliinit_hip] map updates repeated with exact list sizes: %lld\n", h->map_repeats);
  if (h->diag && h->host_us[4] > 0)
    std::fprintf(stderr, "[libliinit_hip] host side of lii_scan_register, us per call over %.0f calls: first launch submitted %.1f, pre-processing enqueued %.1f, "
                 "loop enqueue %.1f, call %.1f, between calls %.1f\n", h->host_us[4], h->host_us[0] / h->host_us[4], h->host_us[1] / h->host_us[4],
                 h->host_us[2] / h->host_us[4], h->host_us[3] / h->host_us[4], h->host_us[5] / std::max(1.0, h->host_us[4] - 1));
  for (auto& e : h->graphs) (void)hipGraphExecDestroy(e.second);
  h->graphs.clear();
  mailbox_close(&h->mailbox);
  if (h->d_mb_seq) (void)hipFree(h->d_mb_seq);
  if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
  for (hipEvent_t e : h->kp_ev) (void)hipEventDestroy(e);
  if (h->ev_next) (void)hipEventDestroy(h->ev_next);
  if (h->ev_scan_free) (void)hipEventDestroy(h->ev_scan_free);
  if (h->h_stage_next) (void)hipHostFree(h->h_stage_next);
  if (h->d_scan_next) (void)hipFree(h->d_scan_next);
  void* dev[] = {h->d_dropped, h->d_pts, h->d_cell_cap, h->d_tp, h->d_cs_a, h->d_cs_b, h->d_work, h->d_ins_e, h->d_ins_e2, h->d_mapctr, h->d_map_unsorted, h->d_map, h->d_keys_a, h->d_keys_b, h->d_keys_c, h->d_idx_a, h->d_idx_b, h->d_blocks, h->d_cells,
                 h->d_counter, h->d_tomb, h->d_batch, h->d_ins, h->d_ins_c, h->d_u32_a, h->d_u32_b, h->d_u32_c, h->d_list_add, h->d_list_nodown, h->d_ah_key, h->d_ah_best, h->d_ah_slot, h->d_sort_temp, h->d_scan, h->d_body, h->d_world, h->d_nbr, h->d_nbr_count, h->d_plane,
                 h->d_selected, h->d_nbody, h->d_ctrl, h->d_pose, h->d_partials, h->d_out91, h->d_gran, h->d_extent, h->d_mm, h->d_bbox_rows, h->d_vkeys_a, h->d_vkeys_b,
                 h->d_vidx_b, h->d_vcomp, h->d_vsplit, h->d_vhist, h->d_vbucket, h->d_vpcl_in, h->d_vpcl_out, h->vh.slots, h->vh.slot_of, h->vh.next, h->vh.counts, h->vh.crowded, h->d_cal_imu, h->d_cal_lidar, h->d_cal_params,
                 h->d_cal_out};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (h->ingest) ingest_destroy(h->ingest);
  if (h->h_stage) (void)hipHostFree(h->h_stage);
  if (h->h_small) (void)hipHostFree(h->h_small);
  if (h->h_ctrl) (void)hipHostFree(h->h_ctrl);
  if (h->h_res) (void)hipHostFree(h->h_res);
  if (h->ev_poses) (void)hipEventDestroy(h->ev_poses);
  if (h->ev_stage) (void)hipEventDestroy(h->ev_stage);
  if (h->ev_mapflag) (void)hipEventDestroy(h->ev_mapflag);
  if (h->ev_vh) (void)hipEventDestroy(h->ev_vh);
  if (h->h_vh_crowded) (void)hipHostFree(h->h_vh_crowded);
  if (h->h_mapflag) (void)hipHostFree(h->h_mapflag);
  if (h->ev_lists) (void)hipEventDestroy(h->ev_lists);
  if (h->n_map_pinned) (void)hipHostFree(h->n_map_pinned);
  for (int i = 0; i < 4; i++)
    if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  for (int i = 0; i < 32; i++)
    if (h->ev_it[i]) (void)hipEventDestroy(h->ev_it[i]);
  if (h->map_stream) (void)hipStreamDestroy(h->map_stream);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return LII_OK;
}

int lii_synchronize(lii_handle h) {
  if (!h) return LII_ERR_INVALID;
  {
    const int rc = map_join(h);
    if (rc != LII_OK) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return LII_OK;
}

// ------------------------------------------------------------------------------------------------ map
int lii_map_reset(lii_handle h) {
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
  launch_cell_apply(h->d_work, h->d_cells, h->d_cell_cap, h->d_pts, h->d_tomb, h->d_tp, h->d_mapctr, h->pts_cap_eff, ne, s);
  launch_ins_write(h->d_pts, h->d_ins_e, 0, nullptr, nullptr, nullptr, 0, nullptr, h->d_cells, h->d_cell_cap, h->d_pts, h->d_mapctr, h->d_dropped, h->drop_cap,
                   s);  // (re-arms the work list)
  rc = map_counters(h);
  if (rc != LII_OK) return rc;
  if (n_deleted) *n_deleted = n_old - h->n_map;
  if (h->n_map != n_old) h->have_search = false;
  return LII_OK;
}
int lii_map_size(lii_handle h, int32_t* n_valid) {
  if (!h || !n_valid) return LII_ERR_INVALID;
  int rc = map_counters(h);  // (a pending in-place update: one small synchronising read)
  *n_valid = h->n_map;
  return rc;
}
int lii_map_download(lii_handle h, float* xyz_out, int32_t capacity, int32_t* n) {
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
  if (!h) return LII_ERR_INVALID;
  {
    const int rcj = map_join(h);
    if (rcj != LII_OK) return rcj;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return LII_OK;
}

// ------------------------------------------------------------------------------------------------ scan
int lii_scan_upload(lii_handle h, const void* points, int32_t n, int32_t stride_bytes, int32_t time_offset_bytes) {
  if (!h || (!points && n > 0) || n < 0 || stride_bytes < 16 || time_offset_bytes < 12 || time_offset_bytes + 4 > stride_bytes)
    return fail(h, LII_ERR_INVALID, "lii_scan_upload: bad arguments");
  if (n > h->cfg.max_scan_points) return fail(h, LII_ERR_CAPACITY, "lii_scan_upload: n > max_scan_points");
  const char* src = static_cast<const char*>(points);
  HIPCHK(h, hipEventSynchronize(h->ev_stage));  // the previous upload has left the staging buffer
  if (stride_bytes == 16 && time_offset_bytes == 12) {
    if (n > 0) std::memcpy(h->h_stage, src, sizeof(float4) * size_t(n));  // already (x, y, z, t) records
  } else {
    for (int i = 0; i < n; i++) {
      const float* f = reinterpret_cast<const float*>(src + size_t(i) * stride_bytes);
      float t;
      std::memcpy(&t, src + size_t(i) * stride_bytes + time_offset_bytes, 4);
      h->h_stage[i] = make_float4(f[0], f[1], f[2], t);
    }
  }
  if (n > 0) HIPCHK(h, hipMemcpyAsync(h->d_scan, h->h_stage, sizeof(float4) * size_t(n), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_stage, h->stream));
  h->n_scan = n;
  extent_discard(h);
  h->bbox_rows = 0;
  h->n_body = 0;
  h->n_body_pending = false;
  h->have_search = false;
  return LII_OK;
}
int lii_scan_upload_next(lii_handle h, const void* points, int32_t n, int32_t stride_bytes, int32_t time_offset_bytes) {
  if (!h || (!points && n > 0) || n < 0 || stride_bytes < 16 || time_offset_bytes < 12 || time_offset_bytes + 4 > stride_bytes)
    return fail(h, LII_ERR_INVALID, "lii_scan_upload_next: bad arguments");
  if (n > h->cfg.max_scan_points) return fail(h, LII_ERR_CAPACITY, "lii_scan_upload_next: n > max_scan_points");
  if (!h->copy_stream) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_next, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_scan_free, hipEventDisableTiming));
    HIPCHK(h, dmalloc(&h->d_scan_next, size_t(h->cfg.max_scan_points)));
  }
  if (h->n_scan_next >= 0) HIPCHK(h, hipEventSynchronize(h->ev_next));  // a scan that was never advanced to is replaced
  h->n_scan_next = -1;
  const void* src = points;
  bool direct = false;
  if (n > 0 && stride_bytes == 16 && time_offset_bytes == 12) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, points) == hipSuccess) direct = attr.type == hipMemoryTypeHost;
    else (void)hipGetLastError();  // pageable memory: not an error
  }
  if (n > 0 && !direct) {
    if (!h->h_stage_next)
      HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_stage_next), sizeof(float4) * size_t(h->cfg.max_scan_points), hipHostMallocDefault));
    const char* p = static_cast<const char*>(points);
    if (stride_bytes == 16 && time_offset_bytes == 12) {
      std::memcpy(h->h_stage_next, p, sizeof(float4) * size_t(n));
    } else {
      for (int i = 0; i < n; i++) {
        const float* f = reinterpret_cast<const float*>(p + size_t(i) * stride_bytes);
        float t;
        std::memcpy(&t, p + size_t(i) * stride_bytes + time_offset_bytes, 4);
        h->h_stage_next[i] = make_float4(f[0], f[1], f[2], t);
      }
    }
    src = h->h_stage_next;
  }
  // the buffer being written was the current scan of an earlier call: whatever the compute stream still has to do with it
  // (kernels enqueued up to now) comes first
  HIPCHK(h, hipEventRecord(h->ev_scan_free, h->stream));
  HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_scan_free, 0));
  if (n > 0) HIPCHK(h, hipMemcpyAsync(h->d_scan_next, src, sizeof(float4) * size_t(n), hipMemcpyHostToDevice, h->copy_stream));
  HIPCHK(h, hipEventRecord(h->ev_next, h->copy_stream));
  h->n_scan_next = n;
  return LII_OK;
}
int lii_scan_advance(lii_handle h) {
  if (!h) return LII_ERR_INVALID;
  if (h->n_scan_next < 0) return fail(h, LII_ERR_STATE, "lii_scan_advance: no scan under way (call lii_scan_upload_next)");
  HIPCHK(h, hipEventSynchronize(h->ev_next));  // long done when the transfer overlapped a registration; frees the caller's buffer
  std::swap(h->d_scan, h->d_scan_next);
  h->n_scan = h->n_scan_next;
  h->n_scan_next = -1;
  extent_discard(h);
  h->bbox_rows = 0;
  h->n_body = 0;
  h->n_body_pending = false;
  h->have_search = false;
  return LII_OK;
}
int lii_scan_set_device(lii_handle h, const void* dev_float4, int32_t n) {
  if (!h || (!dev_float4 && n > 0) || n < 0) return fail(h, LII_ERR_INVALID, "lii_scan_set_device: bad arguments");
  if (n > h->cfg.max_scan_points) return fail(h, LII_ERR_CAPACITY, "lii_scan_set_device: n > max_scan_points");
  extent_discard(h);
  h->bbox_rows = 0;
  if (n > 0) {  // copy + time extent of the scan in one pass (the de-skew that usually follows needs the extent)
    launch_time_extent(static_cast<const float4*>(dev_float4), n, h->d_extent + 2 * h->extent_sel,
                       h->d_extent + 2 * (h->extent_sel ^ 1), h->d_scan, h->h_ctrl, h->d_ctrl, h->ctrl_pending, h->stream);
    h->ctrl_pending = 0;
    h->extent_valid = true;
  }
  h->n_scan = n;
  h->n_body = 0;
  h->n_body_pending = false;
  h->have_search = false;
  return LII_OK;
}
int lii_undistort_imu(lii_handle h, const lii_pose6d* poses, int32_t n_poses, const double end_R[9], const double end_p[3],
                      const double R_LI[9], const double T_LI[3]) {
  if (!h || !poses || n_poses < 1 || n_poses > 1024 || !end_R || !end_p || !R_LI || !T_LI)
    return fail(h, LII_ERR_INVALID, "lii_undistort_imu: bad arguments");
  static_assert(sizeof(lii_pose6d) == 22 * sizeof(double), "lii_pose6d layout");
  if (h->n_scan <= 0 || n_poses < 2) return LII_OK;  // nothing to compensate (IMUpose needs a head and a tail)
  if (!h->poses_preloaded) {
    if (h->staging_busy) { HIPCHK(h, hipStreamSynchronize(h->stream)); h->staging_busy = false; }
    HIPCHK(h, hipEventSynchronize(h->ev_poses));  // the previous table has left the staging buffer (normally long ago)
    std::memcpy(h->h_poses, poses, sizeof(lii_pose6d) * size_t(n_poses));
    HIPCHK(h, hipMemcpyAsync(h->d_poses, h->h_poses, sizeof(lii_pose6d) * size_t(n_poses), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipEventRecord(h->ev_poses, h->stream));
  }
  h->poses_preloaded = false;
  UndistArgH u;
  std::memcpy(u.endR, end_R, 72);
  std::memcpy(u.endp, end_p, 24);
  std::memcpy(u.RLI, R_LI, 72);
  std::memcpy(u.TLI, T_LI, 24);
  unsigned long long* ext = extent_of_scan(h);
  // lii_scan_register told us the hashed voxel filter follows at a leaf that has been probed: its insert rides in the de-skew
  // (one launch less per scan; lii_downsample goes on with the emit)
  h->vh_inserted = h->fuse_leaf > 0.f && !h->voxel_sort && h->vh_mode == 1 && (h->vh_pinned || h->fuse_leaf == h->vh_leaf) && !h->no_fuse;
  if (h->vh_inserted) h->vh_inserted_leaf = h->fuse_leaf;
  DeskewPlan dp = {};
  dp.in = dp.out = h->d_scan; dp.n = h->n_scan; dp.sorted = 0; dp.extent = ext; dp.bbox_rows = h->d_bbox_rows;
  dp.leaf = h->fuse_leaf; dp.vh = h->vh_inserted ? &h->vh : nullptr;
  launch_deskew_imu(dp, nullptr, h->d_poses, n_poses, u, h->stream);
  h->bbox_rows = (h->n_scan + 255) / 256;
  HIPCHK(h, hipGetLastError());
  return LII_OK;
}
int lii_undistort_cv(lii_handle h, const double omega[3], const double vel[3], const double end_R[9]) {
  if (!h || !omega || !vel || !end_R) return fail(h, LII_ERR_INVALID, "lii_undistort_cv: bad arguments");
  if (h->n_scan <= 0) return LII_OK;
  CvArgH a;
  std::memcpy(a.omega, omega, 24);
  std::memcpy(a.vel, vel, 24);
  std::memcpy(a.endR, end_R, 72);
  unsigned long long* ext = extent_of_scan(h);
  h->vh_inserted = h->fuse_leaf > 0.f && !h->voxel_sort && h->vh_mode == 1 && (h->vh_pinned || h->fuse_leaf == h->vh_leaf) && !h->no_fuse;
  if (h->vh_inserted) h->vh_inserted_leaf = h->fuse_leaf;
  DeskewPlan dp = {};
  dp.in = dp.out = h->d_scan; dp.n = h->n_scan; dp.sorted = 0; dp.extent = ext; dp.bbox_rows = h->d_bbox_rows;
  dp.leaf = h->fuse_leaf; dp.vh = h->vh_inserted ? &h->vh : nullptr;
  launch_deskew_cv(dp, a, h->stream);
  h->bbox_rows = (h->n_scan + 255) / 256;
  HIPCHK(h, hipGetLastError());
  return LII_OK;
}
int lii_downsample_skip(lii_handle h, int32_t* n_down) {
  if (!h) return LII_ERR_INVALID;
  if (h->kp_active) { const int r = kp_mark(h, LII_KP_VOXEL); if (r != LII_OK) return r; }
  if (h->n_scan > 0)
    HIPCHK(h, hipMemcpyAsync(h->d_body, h->d_scan, sizeof(float4) * size_t(h->n_scan), hipMemcpyDeviceToDevice, h->stream));
  h->n_body = h->n_scan;
  h->n_body_pending = false;
  h->have_search = false;
  h->body_reordered = false;
  if (n_down) *n_down = h->n_body;
  return LII_OK;
}
int lii_downsample(lii_handle h, float leaf, int32_t* n_down, int32_t* filtered) {
  if (!h || !(leaf > 0)) return fail(h, LII_ERR_INVALID, "lii_downsample: bad arguments");
  h->have_search = false;
  h->n_body_pending = false;
  const int n = h->n_scan;
  if (n <= 0) {
    h->n_body = 0;
    if (n_down) *n_down = 0;
    if (filtered) *filtered = 1;
    return LII_OK;
  }
  // Entirely on the stream: bounding box -> (device) overflow guard + grid parameters -> keys (+ splitter samples) ->
  // sample sort -> centroids + count (lii_vsort.hip).  The size of the result stays in HBM (d_nbody); the registration
  // kernels read it there, so the host learns it only if the caller asks (n_down / filtered != NULL, or a download).
  hipStream_t s = h->stream;
  if (h->kp_active) { const int r = kp_mark(h, LII_KP_VOXEL); if (r != LII_OK) return r; }
  unsigned int* mm = h->d_mm + 8 * h->mm_sel;  // the box: rows a de-skew kernel left behind, or a pass of its own over the scan
  if (h->bbox_rows == 0) {
    h->mm_sel ^= 1;
    launch_voxel_minmax(h->d_scan, n, mm, h->d_mm + 8 * h->mm_sel, s);
  }
  // Which filter: the hashed one orders the members of a voxel by repeated selection - right for the few points per voxel of a
  // leaf matched to the sensor (the shipped configurations: leaf 0.05), quadratic for a voxel of hundreds.  The first scan of
  // a leaf size is probed: its table is built, the longest member list (VoxelHashBuffers::crowded) read back - the one wait of
  // this function, once per leaf size - and the scan goes on through the hashed emit or through the sample sort.  After that
  // the counter is read behind the filter now and then (the sort reports its largest voxel the same way) and the handle
  // changes over when the voxels fill up or thin out in the middle of a run.  LII_VOXEL_FILTER=sort | hash pins the choice.
  // (the counter is applied a fixed number of filter runs after it was requested - eight scans later the copy has long
  // arrived, the wait is free - so that the change-over does not depend on timing: the two filters emit the cloud in different
  // orders, and the ranks of a sharded job, which split it by index, must change over at the same scan)
  h->vh_calls++;
  if (h->vh_flag_pending && h->vh_calls >= h->vh_due) {
    HIPCHK(h, hipEventSynchronize(h->ev_vh));
    h->vh_flag_pending = false;
    if (h->vh_mode == 0 && *h->h_vh_crowded <= 32u) h->vh_mode = 1;         // sparse again -> hash
    else if (h->vh_mode == 1 && *h->h_vh_crowded > 64u) h->vh_mode = 0;     // crowded -> the sort
  }
  int hash_stages = 3;
  // the de-skew of this scan has already filled the table (lii_scan_register: k_deskew_*<true>): the emit follows, whatever
  // the watch above has decided for the scans to come
  const bool inserted = h->vh_inserted;
  h->vh_inserted = false;
  if (inserted && leaf != h->vh_inserted_leaf) return fail(h, LII_ERR_STATE, "lii_downsample: the de-skew prepared the voxel filter for another leaf size");
  if (inserted) hash_stages = 4;
  if (!inserted && !h->vh_pinned && !h->voxel_sort && leaf != h->vh_leaf) {
    h->vh_leaf = leaf;
    h->vh_flag_pending = false;
    HIPCHK(h, hipMemsetAsync(h->vh.crowded, 0, sizeof(unsigned int), s));
    launch_voxel_hash(h->vh, h->d_scan, n, mm, h->d_bbox_rows, h->bbox_rows, leaf, h->d_body, h->d_nbody, h->d_nbody + 1, h->d_vpcl_out, 1, 0u, s);
    HIPCHK(h, hipMemcpyAsync(h->h_vh_crowded, h->vh.crowded, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemsetAsync(h->vh.crowded, 0, sizeof(unsigned int), s));
    HIPCHK(h, hipStreamSynchronize(s));
    h->vh_mode = *h->h_vh_crowded > 32u ? 0 : 1;
    hash_stages = 2;
    if (h->vh_mode == 0)  // the emit would have left the table clean for the next scan; the sort does not know of it
      launch_voxel_hash_clear(h->vh, voxel_hash_slots(n), s);  // (the slots this scan could have touched)
  }
  const bool use_hash = inserted || (h->vh_mode == 1 && !h->voxel_sort);
  h->voxel_path_hash = use_hash;
  if (use_hash) {
    if (++h->vh_epoch == 0u) h->vh_epoch = 1u;
    launch_voxel_hash(h->vh, h->d_scan, n, mm, h->d_bbox_rows, h->bbox_rows, leaf, h->d_body, h->d_nbody, h->d_nbody + 1, h->d_vpcl_out, hash_stages, h->vh_epoch, s);
    if (!h->vh_pinned && !h->vh_flag_pending && (++h->vh_watch & 15) == 0) {  // (every 16th scan: the copy costs a packet on the stream)
      HIPCHK(h, hipMemcpyAsync(h->h_vh_crowded, h->vh.crowded, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
      HIPCHK(h, hipMemsetAsync(h->vh.crowded, 0, sizeof(unsigned int), s));
      HIPCHK(h, hipEventRecord(h->ev_vh, s));
      h->vh_flag_pending = true;
      h->vh_due = h->vh_calls + 8;
    }
  } else {
    const VoxelSortPlan plan = voxel_sort_plan(n);
    launch_voxel_keys(h->d_scan, n, mm, h->d_bbox_rows, h->bbox_rows, leaf, h->d_vkeys_a, h->d_vpcl_in,
                      h->d_nbody + 1, plan.samples ? h->d_vsplit + 2048 : nullptr, plan.width, s);
    VoxelSortBuffers vb;
    vb.pcl_in = h->d_vpcl_in; vb.pcl_out = h->d_vpcl_out;
    vb.samples = h->d_vsplit + 2048;
    vb.keys_in = h->d_vkeys_a; vb.keys_out = h->d_vkeys_b; vb.idx_out = h->d_vidx_b;
    vb.comp = h->d_vcomp; vb.splitters = h->d_vsplit; vb.hist = h->d_vhist; vb.bucket_of = h->d_vbucket;
    vb.max_run = h->voxel_sort ? nullptr : h->vh.crowded;
    if (!h->voxel_sort) HIPCHK(h, hipMemsetAsync(h->vh.crowded, 0, sizeof(unsigned int), s));
    launch_voxel_sort_centroids(vb, h->d_scan, n, h->d_body, h->d_nbody, s);
    if (!h->voxel_sort && !h->vh_flag_pending && (++h->vh_watch & 15) == 0) {  // how crowded are the voxels now?
      HIPCHK(h, hipMemcpyAsync(h->h_vh_crowded, h->vh.crowded, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
      HIPCHK(h, hipMemsetAsync(h->vh.crowded, 0, sizeof(unsigned int), s));
      HIPCHK(h, hipEventRecord(h->ev_vh, s));
      h->vh_flag_pending = true;
      h->vh_due = h->vh_calls + 8;
    }
  }
  HIPCHK(h, hipGetLastError());
  h->n_body = n;  // upper bound until resolved
  h->n_body_pending = true;
  h->body_reordered = use_hash;  // (the hashed filter emits the voxels in the order of their first points)
  h->pcl_perm_valid = false;
  if (n_down || filtered) {
    int rc = resolve_n_body(h);
    if (rc != LII_OK) return rc;
    if (n_down) *n_down = h->n_body;
    if (filtered) *filtered = h->last_filtered;
  }
  return LII_OK;
}
int lii_scan_download(lii_handle h, int32_t which, float* out_float4, int32_t capacity, int32_t* n) {
  if (!h || !n) return LII_ERR_INVALID;
  const float4* src = which == 0 ? h->d_scan : (which == 1 ? h->d_body : h->d_world);
  if (which != 0) { int rc0 = resolve_n_body(h); if (rc0 != LII_OK) return rc0; }
  int cnt = which == 0 ? h->n_scan : h->n_body;
  *n = cnt;
  if (!out_float4) return LII_OK;
  if (capacity < cnt) return fail(h, LII_ERR_CAPACITY, "lii_scan_download: capacity too small");
  if (cnt > 0) {
    const int* perm = nullptr;
    if (which != 0) { int rc1 = pcl_order(h, &perm); if (rc1 != LII_OK) return rc1; }
    HIPCHK(h, hipMemcpyAsync(h->h_stage, src, sizeof(float4) * size_t(cnt), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!perm) std::memcpy(out_float4, h->h_stage, sizeof(float4) * size_t(cnt));
    else
      for (int r = 0; r < cnt; r++) std::memcpy(out_float4 + 4 * size_t(r), &h->h_stage[perm[r]], sizeof(float4));
  }
  return LII_OK;
}

// ------------------------------------------------------------------------------------------------ registration
int lii_iekf_iterate(lii_handle h, const lii_state* state, int32_t search, int32_t imu_en, double out91[91]) {
  if (!h || !state || !out91) return fail(h, LII_ERR_INVALID, "lii_iekf_iterate: bad arguments");
  return iterate(h, state, search != 0, imu_en != 0, out91);
}

int lii_iekf_update(lii_handle h, lii_state* state, const lii_state* state_prop, const lii_iekf_opts* opts,
                    lii_iekf_report* report) {
  if (!h || !state || !state_prop || !opts || opts->max_iterations < 1) return fail(h, LII_ERR_INVALID, "lii_iekf_update: bad arguments");
  const int max_it = opts->max_iterations;
  auto t_begin = std::chrono::steady_clock::now();
  if (!h->host_solve) {
    int rc = update_on_device(h, state, state_prop, opts, report);
    h->timings[4] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return rc;
  }
  double host_ms = 0;
  // cov is constant inside the loop (it is only rewritten on exit, :1112-1114), so invert it once
  std::vector<double> Pinv(kDim * kDim), A(kDim * kDim), K1(kDim * kDim), KH(kDim * 12), G(kDim * kDim);
  if (!mat_inverse(state->cov, kDim, Pinv.data())) return fail(h, LII_ERR_INVALID, "state covariance is singular");
  int rematch_num = 0;
  bool search = true, stop = false, converged = false;
  int it = 0, searches = 0;
  double ne[kNormalEq];
  for (it = 0; it < max_it; it++) {
    int rc = iterate(h, state, search, opts->imu_en != 0, ne);
    if (rc != LII_OK) return rc;
    if (search) searches++;
    auto t0 = std::chrono::steady_clock::now();
    // H_T_H (+) P^-1  (:1080-1081)
    A = Pinv;
    double HTH[12][12];
    int t = 0;
    for (int i = 0; i < 12; i++)
      for (int j = i; j < 12; j++) { HTH[i][j] = ne[t]; HTH[j][i] = ne[t]; t++; }
    for (int i = 0; i < 12; i++)
      for (int j = 0; j < 12; j++) A[size_t(i) * kDim + j] += HTH[i][j];
    if (!mat_inverse(A.data(), kDim, K1.data())) return fail(h, LII_ERR_INVALID, "normal matrix is singular");
    double vec[kDim], sol[kDim];
    state_minus(*state_prop, *state, vec);
    for (int r = 0; r < kDim; r++) {
      double kz = 0;
      for (int c = 0; c < 12; c++) kz += K1[size_t(r) * kDim + c] * ne[78 + c];
      double khv = 0;
      for (int c = 0; c < 12; c++) {
        double s = 0;
        for (int k = 0; k < 12; k++) s += K1[size_t(r) * kDim + k] * HTH[k][c];
        KH[size_t(r) * 12 + c] = s;
        khv += s * vec[c];
      }
      sol[r] = kz + vec[r] - khv;
    }
    state_plus(*state, sol);
    double rn = std::sqrt(sol[0] * sol[0] + sol[1] * sol[1] + sol[2] * sol[2]);
    double tn = std::sqrt(sol[3] * sol[3] + sol[4] * sol[4] + sol[5] * sol[5]);
    converged = (rn * 57.3 < 0.01) && (tn * 100 < 0.015);
    search = false;
    if (converged || ((rematch_num == 0) && (it == (max_it - 2)))) {
      search = true;
      rematch_num++;
    }
    if (!stop && (rematch_num >= 2 || (it == max_it - 1))) {
      // state.cov = (I - G) cov, G[:, :12] = K H   (:1111-1114)
      std::vector<double> newcov(kDim * kDim);
      for (int r = 0; r < kDim; r++)
        for (int c = 0; c < kDim; c++) {
          double s = state->cov[size_t(r) * kDim + c];
          for (int k = 0; k < 12; k++) s -= KH[size_t(r) * 12 + k] * state->cov[size_t(k) * kDim + c];
          newcov[size_t(r) * kDim + c] = s;
        }
      std::memcpy(state->cov, newcov.data(), sizeof(double) * kDim * kDim);
      stop = true;
    }
    host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stop) { it++; break; }
  }
  if (report) {
    report->iterations = it;
    report->searches = searches;
    report->effect_num = int(ne[90]);
    report->converged = converged ? 1 : 0;
    std::memcpy(report->normal_eq, ne, sizeof(ne));
  }
  h->timings[3] = host_ms;
  h->timings[4] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return LII_OK;
}

int lii_scan_register(lii_handle h, const lii_scan_job* job, lii_state* state, const lii_state* state_prop,
                      lii_iekf_report* report) {
  // (struct_size 48: a job of ABI 5, without scan_sorted)
  if (!h || !job || (job->struct_size != sizeof(lii_scan_job) && job->struct_size != 48u) || !state || !state_prop || job->opts.max_iterations < 1)
    return fail(h, LII_ERR_INVALID, "lii_scan_register: bad arguments");
  const bool sorted = job->struct_size >= sizeof(lii_scan_job) && job->scan_sorted == 1;
  int rc = LII_OK;
  const auto t_entry = std::chrono::steady_clock::now();
  if (h->diag && h->host_us[4] > 0) h->host_us[5] += std::chrono::duration<double, std::micro>(t_entry - h->host_last_return).count();
  const bool adopt = job->scan_dev != nullptr && job->n_scan_dev > 0;
  if (adopt && job->n_scan_dev > h->cfg.max_scan_points) return fail(h, LII_ERR_CAPACITY, "lii_scan_register: n_scan_dev > max_scan_points");
  const int n_next = adopt ? job->n_scan_dev : h->n_scan;
  // A scan in ascending time order: ONE launch takes it from wherever it arrived (the caller's device buffer is read in place)
  // to the de-skewed scan with the voxel filter's table filled, and one extra workgroup of it pulls the update's control block
  // over PCIe; the IMU pose table (<= 64 poses) travels in the kernel arguments.  Round 3 needed k_time_extent in front (copy +
  // time extent + pull: 8.9 us per scan).
  h->kp_active = h->prof_mode == 3 && !h->host_solve;
  h->kp_n = 0;
  if (h->kp_active) { rc = kp_mark(h, LII_KP_DESKEW); if (rc != LII_OK) { h->kp_active = false; return rc; } }
  const bool fast = sorted && !h->host_solve && n_next > 0 && !h->no_fast_prologue &&
                    ((job->undistort == 1 && job->imu_poses && job->n_imu_poses >= 2 && job->n_imu_poses <= 64) || job->undistort == 2);
  if (job->undistort != 0 && job->undistort != 1 && job->undistort != 2) return fail(h, LII_ERR_INVALID, "lii_scan_register: undistort must be 0, 1 or 2");
  const auto t_first = std::chrono::steady_clock::now();
  if (fast) {
    if (h->staging_busy) HIPCHK(h, hipStreamSynchronize(h->stream));  // (a call that failed half way left the buffer in use)
    h->staging_busy = true;
    fill_ctrl(h, state, state_prop, &job->opts);
    h->ctrl_preloaded = true;
    extent_discard(h);
    h->n_scan = n_next;
    h->n_body = 0;
    h->n_body_pending = false;
    h->have_search = false;
    const float fuse_leaf = job->leaf > 0 ? job->leaf : 0.f;
    h->vh_inserted = fuse_leaf > 0.f && !h->voxel_sort && h->vh_mode == 1 && (h->vh_pinned || fuse_leaf == h->vh_leaf) && !h->no_fuse;
    if (h->vh_inserted) h->vh_inserted_leaf = fuse_leaf;
    DeskewPlan dp = {};
    dp.in = adopt ? static_cast<const float4*>(job->scan_dev) : h->d_scan;
    dp.out = h->d_scan; dp.n = n_next; dp.sorted = 1; dp.extent = nullptr; dp.bbox_rows = h->d_bbox_rows;
    dp.leaf = fuse_leaf; dp.vh = h->vh_inserted ? &h->vh : nullptr;
    dp.ctrl_src = h->h_ctrl; dp.ctrl_dst = h->d_ctrl; dp.ctrl_bytes = (sizeof(IekfCtrl) + 15) / 16 * 16;
    if (job->undistort == 1) {
      UndistArgH u;
      std::memcpy(u.endR, state->rot_end, 72);
      std::memcpy(u.endp, state->pos_end, 24);
      std::memcpy(u.RLI, state->offset_R_L_I, 72);
      std::memcpy(u.TLI, state->offset_T_L_I, 24);
      launch_deskew_imu(dp, reinterpret_cast<const double*>(job->imu_poses), nullptr, job->n_imu_poses, u, h->stream);
    } else {
      CvArgH a;  // CV model: bias_g = omega, vel_end = v
      std::memcpy(a.omega, state->bias_g, 24);
      std::memcpy(a.vel, state->vel_end, 24);
      std::memcpy(a.endR, state->rot_end, 72);
      launch_deskew_cv(dp, a, h->stream);
    }
    h->bbox_rows = (n_next + 255) / 256;
    const hipError_t e_launch = hipGetLastError();
    if (e_launch != hipSuccess) { h->vh_inserted = false; h->ctrl_preloaded = false; rc = fail(h, LII_ERR_HIP, std::string("de-skew launch: ") + hipGetErrorString(e_launch)); }
  } else {
  if (job->undistort == 1 && !h->host_solve && job->imu_poses && job->n_imu_poses >= 2 && job->n_imu_poses <= 1024 && n_next > 0) {
    // the control block of the update AND the pose table of the de-skew (they sit behind each other) reach the device once.
    // The staging buffer is free again: the previous call returned after its stopping pass, which runs behind the kernel
    // that read the buffer - unless that call failed half way (then wait).  No event: recording one between the de-skew and
    // the voxel filter cost a ~5 us bubble on the device per scan.
    if (h->staging_busy) HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev_poses));  // a pose table uploaded by a stand-alone lii_undistort_imu (long done)
    h->staging_busy = true;
    fill_ctrl(h, state, state_prop, &job->opts);
    std::memcpy(h->h_poses, job->imu_poses, sizeof(lii_pose6d) * size_t(job->n_imu_poses));
    const size_t bytes = kCtrlBytes + sizeof(lii_pose6d) * size_t(job->n_imu_poses);
    if (adopt || !h->extent_valid) {
      h->ctrl_pending = (bytes + 15) / 16 * 16;  // rides in the scan's first kernel (k_time_extent), which is launched below
    } else {
      // (size rounded to 1 KiB: the runtime splits an H2D copy with an unaligned tail into two blit kernels)
      HIPCHK(h, hipMemcpyAsync(h->d_ctrl, h->h_ctrl, (bytes + 1023) / 1024 * 1024, hipMemcpyHostToDevice, h->stream));
    }
    h->poses_preloaded = h->ctrl_preloaded = true;
  }
  if (adopt) {
    rc = lii_scan_set_device(h, job->scan_dev, job->n_scan_dev);
    if (rc != LII_OK) { h->ctrl_pending = 0; h->poses_preloaded = h->ctrl_preloaded = false; return rc; }
  }
  if (job->undistort == 1) {
    h->fuse_leaf = job->leaf > 0 ? job->leaf : 0.f;  // (the voxel filter follows in this call: its insert may ride in the de-skew)
    rc = lii_undistort_imu(h, job->imu_poses, job->n_imu_poses, state->rot_end, state->pos_end, state->offset_R_L_I,
                           state->offset_T_L_I);
    h->fuse_leaf = 0.f;
    if (rc != LII_OK) h->vh_inserted = false;
  } else if (job->undistort == 2) {
    h->fuse_leaf = job->leaf > 0 ? job->leaf : 0.f;
    rc = lii_undistort_cv(h, state->bias_g, state->vel_end, state->rot_end);  // CV model: bias_g = omega, vel_end = v
    h->fuse_leaf = 0.f;
    if (rc != LII_OK) h->vh_inserted = false;
  }
  }
  if (h->ctrl_preloaded && h->ctrl_pending) {  // no kernel picked the block up (cannot happen with the conditions above; a guard)
    HIPCHK(h, hipMemcpyAsync(h->d_ctrl, h->h_ctrl, h->ctrl_pending, hipMemcpyHostToDevice, h->stream));
    h->ctrl_pending = 0;
  }
  if (rc == LII_OK) rc = job->leaf > 0 ? lii_downsample(h, job->leaf, nullptr, nullptr) : lii_downsample_skip(h, nullptr);
  const auto t_pre = std::chrono::steady_clock::now();
  if (rc == LII_OK) rc = lii_iekf_update(h, state, state_prop, &job->opts, report);
  h->poses_preloaded = h->ctrl_preloaded = false;  // also on the error paths
  h->kp_active = false;
  if (h->diag) {
    const auto t_end = std::chrono::steady_clock::now();
    h->host_us[0] += std::chrono::duration<double, std::micro>(t_first - t_entry).count();
    h->host_us[1] += std::chrono::duration<double, std::micro>(t_pre - t_entry).count();
    h->host_us[2] += h->host_loop_enq_us;
    h->host_us[3] += std::chrono::duration<double, std::micro>(t_end - t_entry).count();
    h->host_us[4] += 1;
    h->host_last_return = t_end;
  }
  return rc;
}

int lii_neighbors_download(lii_handle h, float* pts, int32_t* counts, uint8_t* selected, int32_t capacity) {
  if (!h) return LII_ERR_INVALID;
  { int rc0 = resolve_n_body(h); if (rc0 != LII_OK) return rc0; }
  const int n = h->n_body;
  if (capacity < n) return fail(h, LII_ERR_CAPACITY, "lii_neighbors_download: capacity too small");
  if (n == 0) return LII_OK;
  const size_t cap = size_t(h->cfg.max_scan_points);
  hipStream_t s = h->stream;
  const int* perm = nullptr;  // rows come out in the order lii_scan_download(1) uses (the reference's feats_down_body order)
  { int rc1 = pcl_order(h, &perm); if (rc1 != LII_OK) return rc1; }
  if (pts) {
    for (int k = 0; k < kMatch; k++)
      HIPCHK(h, hipMemcpyAsync(h->h_stage + size_t(k) * n, h->d_nbr + size_t(k) * cap, sizeof(float4) * size_t(n), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    for (int i = 0; i < n; i++)
      for (int k = 0; k < kMatch; k++) {
        const float4 v = h->h_stage[size_t(k) * n + (perm ? perm[i] : i)];
        float* o = pts + (size_t(i) * kMatch + k) * 3;
        o[0] = v.x; o[1] = v.y; o[2] = v.z;
      }
  }
  if (counts) {
    HIPCHK(h, hipMemcpyAsync(h->h_stage, h->d_nbr_count, sizeof(int) * size_t(n), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    const int* src = reinterpret_cast<const int*>(h->h_stage);
    for (int i = 0; i < n; i++) counts[i] = src[perm ? perm[i] : i];
  }
  if (selected) {
    HIPCHK(h, hipMemcpyAsync(h->h_stage, h->d_selected, size_t(n), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    const uint8_t* src = reinterpret_cast<const uint8_t*>(h->h_stage);
    for (int i = 0; i < n; i++) selected[i] = src[perm ? perm[i] : i];
  }
  return LII_OK;
}

int lii_map_incremental(lii_handle h, const lii_state* state, int32_t* n_add, int32_t* n_no_downsample) {
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
  if (rb.shard_world > 1) {
    // A sharded job: this rank holds neighbour lists for its own block only, but every rank must take the SAME decisions for
    // the whole cloud or the replicated maps drift apart.  No exchange: the search is repeated here for the whole cloud at the
    // pose of the last executed search pass (IekfCtrl::search_pose, identical on every rank) - one more k-NN pass, a few
    // percent of what the map update itself costs - and the lists come out bit-identical on every rank.
    rb.shard_world = 1;
    if (h->have_search) {
      const GridView g = grid_view(h);
      lii::launch_knn(h->knn_variant, g, rb, pose_of(*state), reinterpret_cast<const PoseArg*>(h->d_ctrl->search_pose), h->d_ctrl, 2, nullptr, s);
      launch_knn_complete(g, rb, s);
    }
  }
  // decision per point on the device (world point, neighbour list of the last search) and both order-preserving compactions
  const bool sharded = h->n_ranks > 1;
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
    launch_map_decide_compact(rb, pose_of(*state), double(h->cfg.map_downsample_size), h->have_search ? 1 : 0, h->d_u32_a,
                              reinterpret_cast<uint2*>(h->d_u32_b), h->d_world, h->d_list_add, h->d_list_nodown, h->d_counts, ba, bn, s);
    return map_apply(h, h->d_list_add, ba, true, h->d_list_nodown, bn, true, h->d_counts + 3, h->d_counts + 4, false);
  }
  launch_map_decide_compact(rb, pose_of(*state), double(h->cfg.map_downsample_size), h->have_search ? 1 : 0, h->d_u32_a,
                            reinterpret_cast<uint2*>(h->d_u32_b), h->d_world, h->d_list_add, h->d_list_nodown, h->d_counts, nb, nb, s);
  // The sizes of the two lists, now: a converged map takes a few thousand of the ~100 k points, and everything downstream
  // (voxel keys, the batch sort, the per-voxel fold, the insert compaction) is launched for the exact count instead of the
  // scan-sized bound - one small host round trip (~15 us) against ~80 us of kernels working on padding.
  // (the same round trip brings the map's counters up to date: the previous update ran without a synchronisation)
  HIPCHK(h, hipMemcpyAsync(h->h_small + 3090, h->d_counts, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipMemcpyAsync(h->h_small + 3072, h->d_mapctr, sizeof(int) * kMapCtrWords, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  int n_lists[2];
  std::memcpy(n_lists, h->h_small + 3090, sizeof(n_lists));
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

// ------------------------------------------------------------------------------------------------ calibration
int lii_calib_set_buffers(lii_handle h, const lii_calib_state* imu, const lii_calib_state* lidar, int32_t n) {
  if (!h || !imu || !lidar || n <= 0) return fail(h, LII_ERR_INVALID, "lii_calib_set_buffers: bad arguments");
  static_assert(sizeof(lii_calib_state) == 22 * sizeof(double), "lii_calib_state layout");
  if (n > h->n_cal || !h->d_cal_imu) {
    if (h->d_cal_imu) (void)hipFree(h->d_cal_imu);
    if (h->d_cal_lidar) (void)hipFree(h->d_cal_lidar);
    h->d_cal_imu = h->d_cal_lidar = nullptr;
    HIPCHK(h, dmalloc(&h->d_cal_imu, size_t(n) * 22));
    HIPCHK(h, dmalloc(&h->d_cal_lidar, size_t(n) * 22));
  }
  HIPCHK(h, hipMemcpyAsync(h->d_cal_imu, imu, sizeof(lii_calib_state) * size_t(n), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_cal_lidar, lidar, sizeof(lii_calib_state) * size_t(n), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->n_cal = n;
  return LII_OK;
}

int lii_calib_eval(lii_handle h, int32_t stage, const double* params, double* JtJ, double* Jtr, double* cost) {
  if (!h || !params || stage < 1 || stage > 3) return fail(h, LII_ERR_INVALID, "lii_calib_eval: bad arguments");
  if (h->n_cal <= 0) return fail(h, LII_ERR_STATE, "lii_calib_eval: no buffers uploaded");
  const int np = stage == 1 ? 9 : (stage == 2 ? 13 : 24);
  const int dof = stage == 1 ? 3 : (stage == 2 ? 7 : 9);
  std::memcpy(h->h_small, params, sizeof(double) * np);
  HIPCHK(h, hipMemcpyAsync(h->d_cal_params, h->h_small, sizeof(double) * np, hipMemcpyHostToDevice, h->stream));
  launch_calib_eval(stage, h->d_cal_imu, h->d_cal_lidar, h->n_cal, h->d_cal_params, h->d_cal_out, h->stream);
  const int n_out = dof * dof + dof + 1;
  HIPCHK(h, hipMemcpyAsync(h->h_small + 64, h->d_cal_out, sizeof(double) * n_out, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const double* o = h->h_small + 64;
  if (JtJ) std::memcpy(JtJ, o, sizeof(double) * dof * dof);
  if (Jtr) std::memcpy(Jtr, o + dof * dof, sizeof(double) * dof);
  if (cost) *cost = o[dof * dof + dof];
  return LII_OK;
}

int lii_li_init_set_device(lii_handle h, int32_t on_device) {
  if (!h) return LII_ERR_INVALID;
  h->li_init_device = on_device != 0;
  return LII_OK;
}

// ------------------------------------------------------------------------------------------------ multi-GPU
int lii_comm_unique_id(uint8_t id_out[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  if (!id_out) return LII_ERR_INVALID;
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return fail(nullptr, LII_ERR_COMM, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
  std::memcpy(id_out, &id, 128);
  return LII_OK;
}
namespace {
void comm_drop(lii_handle h) {
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->comm) { ncclCommDestroy(h->comm); h->comm = nullptr; }
  mailbox_close(&h->mailbox);
  h->n_ranks = 1;
  h->rank = 0;
}
}  // namespace
int lii_comm_init_ex(lii_handle h, int32_t n_ranks, int32_t rank, const uint8_t id_in[128], int32_t transport) {
  if (!h || !id_in || n_ranks < 1 || rank < 0 || rank >= n_ranks || transport < LII_COMM_AUTO || transport > LII_COMM_MAILBOX_HOST)
    return fail(h, LII_ERR_INVALID, "lii_comm_init: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  comm_drop(h);
  h->n_ranks = n_ranks;
  h->rank = rank;
  // a single rank needs no exchange; asked for by name, the RCCL transport is still set up (a one-rank communicator), so that
  // the three-launch form of the loop - final sum, ncclAllReduce, solve - can be exercised on one device
  if (n_ranks == 1 && transport != LII_COMM_RCCL) return LII_OK;
  if (transport != LII_COMM_RCCL) {
    if (!h->d_mb_seq) HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->d_mb_seq), sizeof(unsigned long long)));
    HIPCHK(h, hipMemset(h->d_mb_seq, 0, sizeof(unsigned long long)));
    // LII_MAILBOX_TIMEOUT_S=<exchange>[,<set-up>]: how long a reduce+solve kernel waits for a peer's sums (30 s), how long this
    // call waits for all ranks in the node-local segment (20 s)
    double wait_s = 20.0;
    if (const char* t = std::getenv("LII_MAILBOX_TIMEOUT_S")) {
      h->mailbox_timeout_ticks = (long long)(std::atof(t) * 1e8);
      if (const char* c = std::strchr(t, ',')) wait_s = std::atof(c + 1);
    }
    std::string why;
    if (mailbox_open(id_in, n_ranks, rank, wait_s, transport != LII_COMM_MAILBOX_HOST, &h->mailbox, &why) == 0) {
      if (transport == LII_COMM_MAILBOX && !h->mailbox.d_peers) {  // asked for by name: no silent change of the transport
        mailbox_close(&h->mailbox);
        h->n_ranks = 1; h->rank = 0;
        return fail(h, LII_ERR_COMM, "peer-mapped HBM mailbox unavailable: " + why);
      }
      h->comm_why = h->mailbox.d_peers ? "mailbox in peer-mapped HBM (HIP IPC; every rank's device reaches every other's)"
                                       : (transport == LII_COMM_MAILBOX_HOST ? std::string("mailbox in registered host memory (asked for)")
                                                                             : "mailbox in registered host memory - the HBM form was not possible: " + why);
      if (h->diag) std::fprintf(stderr, "[libliinit_hip] rank %d of %d: %s\n", rank, n_ranks, h->comm_why.c_str());
      return LII_OK;
    }
    if (transport == LII_COMM_MAILBOX || transport == LII_COMM_MAILBOX_HOST) {
      h->n_ranks = 1; h->rank = 0;
      return fail(h, LII_ERR_COMM, "node-local mailbox unavailable: " + why);
    }
    h->comm_why = "RCCL - the node-local mailbox was not possible: " + why;
  } else {
    h->comm_why = "RCCL (asked for)";
  }
  ncclUniqueId id;
  std::memcpy(&id, id_in, 128);
  ncclResult_t r = ncclCommInitRank(&h->comm, n_ranks, id, rank);
  if (r != ncclSuccess) {
    h->comm = nullptr; h->n_ranks = 1; h->rank = 0;
    return fail(h, LII_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  if (h->diag) std::fprintf(stderr, "[libliinit_hip] rank %d of %d: %s\n", rank, n_ranks, h->comm_why.c_str());
  return LII_OK;
}
int lii_comm_describe(lii_handle h, char* out, int32_t capacity) {
  if (!h || !out || capacity < 1) return LII_ERR_INVALID;
  const std::string s = (h->comm || h->mailbox.dev_slots || h->mailbox.d_peers) ? h->comm_why : std::string("no communicator");
  std::snprintf(out, size_t(capacity), "%s", s.c_str());
  return LII_OK;
}
int lii_comm_set_partition(lii_handle h, int32_t library_partition) {
  if (!h) return LII_ERR_INVALID;
  h->library_partition = library_partition != 0;
  return LII_OK;
}
int lii_comm_init(lii_handle h, int32_t n_ranks, int32_t rank, const uint8_t id_in[128]) {
  return lii_comm_init_ex(h, n_ranks, rank, id_in, LII_COMM_AUTO);
}
int lii_comm_transport(lii_handle h, int32_t* transport) {
  if (!h || !transport) return LII_ERR_INVALID;
  *transport = h->comm ? LII_COMM_RCCL : (h->mailbox.d_peers ? LII_COMM_MAILBOX : (h->mailbox.dev_slots ? LII_COMM_MAILBOX_HOST : LII_COMM_AUTO));
  return LII_OK;
}
int lii_comm_rccl_ranks(lii_handle h, int32_t* n_ranks) {
  if (!h || !n_ranks) return LII_ERR_INVALID;
  *n_ranks = 0;
  if (h->comm) {
    int n = 0;
    const ncclResult_t r = ncclCommCount(h->comm, &n);
    if (r != ncclSuccess) return fail(h, LII_ERR_COMM, std::string("ncclCommCount: ") + ncclGetErrorString(r));
    *n_ranks = n;
  }
  return LII_OK;
}
int lii_comm_destroy(lii_handle h) {
  if (!h) return LII_ERR_INVALID;
  (void)hipSetDevice(h->device);
  comm_drop(h);
  return LII_OK;
}

// ------------------------------------------------------------------------------------------------ utilities
int lii_dev_alloc(lii_handle h, size_t bytes, void** dev_ptr) {
  if (!h || !dev_ptr) return LII_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMalloc(dev_ptr, bytes));
  return LII_OK;
}
int lii_dev_free(lii_handle h, void* dev_ptr) {
  if (!h) return LII_ERR_INVALID;
  HIPCHK(h, hipFree(dev_ptr));
  return LII_OK;
}
int lii_dev_upload(lii_handle h, void* dev_dst, const void* host_src, size_t bytes) {
  if (!h || !dev_dst || !host_src) return LII_ERR_INVALID;
  HIPCHK(h, hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
  return LII_OK;
}
int lii_set_profiling(lii_handle h, int32_t enabled) {
  if (!h) return LII_ERR_INVALID;
  h->profiling = enabled != 0;
  h->prof_mode = enabled;
  if (enabled == 1) {
    for (double& t : h->timings) t = 0;  // 1: (re)start the accumulation; 2: resume; 0: pause (accumulators kept)
    h->kprof = lii_kernel_profile{};
  }
  return LII_OK;
}
int lii_last_kernel_profile(lii_handle h, lii_kernel_profile* out) {
  if (!h || !out || out->struct_size != sizeof(lii_kernel_profile)) return fail(h, LII_ERR_INVALID, "lii_last_kernel_profile: bad arguments");
  *out = h->kprof;
  out->struct_size = sizeof(lii_kernel_profile);
  return LII_OK;
}
int lii_last_timings(lii_handle h, double out_ms[8]) {
  if (!h || !out_ms) return LII_ERR_INVALID;
  std::memcpy(out_ms, h->timings, sizeof(h->timings));
  return LII_OK;
}

}  // extern "C"
