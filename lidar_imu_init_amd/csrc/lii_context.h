// The handle of libliinit_hip and the host-side internals its translation units share (not part of the C-ABI):
//   lii_capi.cpp           life cycle, scan in / de-skew / voxel grid, downloads, profiling
//   lii_capi_map.cpp       the device-resident local map: (re)build, in-place updates, lii_map_*, lii_map_incremental
//   lii_capi_register.cpp  the registration loop: lii_iekf_*, lii_scan_register, neighbour download
//   lii_capi_comm.cpp      the communicator of a sharded job (node-local mailbox / RCCL)
//   lii_capi_calib.cpp     the LI_init evaluators' entry points
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/liinit_hip.h"
#include "lii_launch.h"

using lii::BlockEntry; using lii::IekfCtrl; using lii::IekfResult; using lii::PoseArg; using lii::VoxelHashBuffers; using lii::MailboxHost;
constexpr size_t kCtrlBytes = (sizeof(lii::IekfCtrl) + 255) / 256 * 256;

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
  // dense cell window over the map's box (GridView::win; LII_WINDOW=0 turns it off): filled by build_index, dropped by whatever changes a cell entry
  bool use_window = true;
  uint2* d_win = nullptr;
  size_t win_cap = 0;            // entries allocated
  int win_org[3] = {0, 0, 0}, win_dim[3] = {0, 0, 0};
  bool win_valid = false;
  long long win_kept = 0, win_dropped = 0;  // LII_DIAG: in-place updates the window was kept current through / times it had to be dropped
  bool win_keep = true;          // LII_WINDOW_KEEP=0: the first in-place update drops the window (round 6's first form) instead of keeping it current
  unsigned long long* d_block_key = nullptr;  // packed block coordinates by block id (WinKeep::key_of_id), cells_cap_blocks entries
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
  bool wide_listed = false;                 // the last scan's search passes listed more than kFlagCap unfinished queries: the next one's search launches are followed by k_complete_listed
  bool wide_prev = false;                   // ... the scan before it did (two in a row switch wide_listed)
  bool unfinished_known = false;            // IekfResult::unfinished belongs to the last search the handle ran (lii_last_unfinished_queries)
  bool wide_enabled = true;                 // LII_WIDE_COMPLETION=0: never (every workgroup of the fit launch finishes its own, rounds 4 - 6)
  long long wide_scans = 0, wide_switches = 0;
  void (*wait_hook)(void*) = nullptr;       // lii_scan_job::while_waiting of the call under way (update_on_device calls it once, before its wait)
  void* wait_hook_arg = nullptr;
  bool in_wait_hook = false;               // ... while it runs: entry points that change which frames are current refuse (lii_ingest_end, lii_frame_select)
  bool map_fuse = true;                     // round 6 (LII_MAP_FUSE=0 turns both off): the fold's hash insert rides in k_map_decide, the inserts' cells in the fold launch
  bool ah_filled = false;                   // k_map_decide has filled the fold's table and no fold has consumed it yet
  long long ah_cleared = 0;                 // ... times such a fill had to be cleared (a list that outgrew its bound)
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
  // lii_frame_select: the frame stays where the ingest left it until something reads the scan - lii_scan_register takes it from
  // there like a caller's device buffer (lii_scan_job::scan_dev), every other reader copies it into d_scan first (scan_materialize)
  const float4* scan_pending = nullptr;
  int scan_pending_n = 0;
  // lii_scan_upload_next / lii_scan_advance: the next scan travels on a copy stream into a second buffer
  float4* d_scan_next = nullptr;
  float4* h_stage_next = nullptr;   // pinned staging for sources that are not (pinned, stride 16)
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_next = nullptr;      // the transfer of the next scan
  hipEvent_t ev_scan_free = nullptr; // the compute stream has finished with the buffer the next transfer writes to
  int n_scan_next = -1;              // >= 0: a scan is waiting in d_scan_next
  const void* pin_cache_ptr[8] = {};  // lii_scan_upload_next: the last source buffers and whether the copy engine can read them directly
  bool pin_cache_direct[8] = {};
  int pin_cache_at = 0;
  bool scan_buf_idle = false;        // everything ever enqueued on the CURRENT scan buffer is known to have completed (an update's result came back behind it,
                                     // nothing touched the buffer since): lii_scan_upload_next may write the other buffer - the one that was current before the
                                     // last lii_scan_advance - without an event between the two streams
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
  unsigned long long* d_gran = nullptr;  // k_reduce_solve: the 91 sums of a pass on their way to the solver, 2 x 91 tagged words; [200 ..]: the gap trace's stamps; [240]: the scan's largest count of unfinished queries so far
  unsigned long long* d_extent = nullptr;  // 2 x {min (time|index), max time}: ping-pong accumulators
  unsigned int* d_mm = nullptr;           // 2 x {min xyz, max xyz} (order-preserving uints)
  int extent_sel = 0, mm_sel = 0;
  bool knn_plan = true;        // LII_KNN_PLAN=0: every k-NN launch is enqueued (IekfCtrl::plan_mask)
  bool test_pred_small = false;
  bool test_force_rebuild = false;  // LII_TEST=force_rebuild: every in-place map update rebuilds the index first (the branch a map low on room takes)
  int solo_share = 0;             // LII_TEST=solo_share=<N>: kernel-timing rehearsal of one rank's share (N > 1: by voxel, N < -1: by index)
  bool no_gather = false;         // LII_TEST=no_gather: no gather areas behind the mailbox slots (the map update of a sharded job repeats the search)
  bool no_fast_prologue = false;  // LII_TEST=no_fast: a time-sorted scan takes the general path as well (k_time_extent in front of the de-skew)
  bool no_fuse = false;        // LII_TEST=no_fuse: lii_scan_register keeps the de-skew and the voxel filter's insert in separate launches
  bool test_sum_lost = false;  // LII_TEST=sum_lost: one summing workgroup of k_reduce_solve never publishes - the solver's wait must end in LII_ERR_COMM
  bool test_emit_late = false; // LII_TEST=emit_late: every seventh workgroup of k_vhash_emit / k_map_decide publishes its count late: the others count its block themselves (prefix_below)
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
  bool body_partitioned = false; // the down-sampled cloud on this rank holds ITS voxels only (a voxel-partitioned job, fused filter): no split by index
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
  int* d_flags = nullptr;       // the lists of unfinished queries (RegistrationBuffers::flag_count / flag_list): 2 counters + 2 x kFlagCap entries of two float4
  int knn_epoch = 0;            // number of the last enqueued search launch (never 0 again once used)
  int last_pivoted_passes = 0;  // lii_last_solve_info: passes of the last device update whose elimination needed the pivoting routine
  hipStream_t map_stream = nullptr;  // the in-place update of lii_map_incremental runs here, beside the next scan's pre-processing
  bool map_async = false;            // ... and may still be running (map_join waits for it: ev_mapflag is its last packet)
  int bound_add = 0, bound_nodown = 0;  // ... the sizes the update in flight was enqueued for
  int list_hist[8][2] = {};             // ... from the sizes of the last eight calls (note_list_sizes)
  int list_hist_n = 0;
  int pred_add = -1, pred_nodown = -1;  // lii_map_incremental: list sizes the next update is enqueued for (< 0: none yet)
  bool map_after_update = false;        // lii_scan_job::map_update: the iterated update in progress enqueues the map update behind its passes
  bool map_enqueued_early = false;      // ... and did (update_on_device -> map_update_early)
  bool lists_predicted = false;         // the update in flight ran on predicted sizes: commit_map checks it against the exact ones
  hipEvent_t ev_lists = nullptr;        // the two lists are complete (compute stream -> map stream)
  int* h_mapflag = nullptr;       // pinned, behind the last in-place update: [0..15] the map counters, [16..20] the list counts of
                                  // lii_map_incremental (k_map_decide) - read by commit_map / map_join
  hipEvent_t ev_mapflag = nullptr;
  unsigned int decide_epoch = 0;     // runs of k_map_decide so far (its in-launch exchange of block counts tells its words from older ones by it)
  int map_seq = 0;                   // number of the last in-place update: k_map_publish leaves it in h_mapflag[kMapFlagSeqAt] behind the counters
  bool map_flag_pending = false;
  bool diag = false;     // LII_DIAG=1: counters of the rare paths on stderr when the handle is destroyed

  // ---- the pre-armed prologue (lii_launch.h: DeskewGate; lii_scan_job::next_scan_dev)
  struct Prearm {
    lii::GateState* state = nullptr;         // pinned, device-mapped: the state word
    double* d_ring = nullptr;                // device memory the HOST writes (large BAR): kGateRing records of kGateLines x 8 doubles
    unsigned long long* d_flag = nullptr;
    unsigned long long seq = 0;
    bool enabled = true;                     // LII_PREARM=0: a job's next_scan_dev is ignored
    bool armed = false;                      // a gated de-skew launch sits on the stream, waiting for its record
    const void* scan_dev = nullptr;          // ... enqueued for this scan,
    int n = 0;
    float leaf = 0.f;                        // ... this leaf,
    bool fuse = false;                       // ... with the hashed filter's insert riding along or not
    bool late = false;                       // ... which was still being uploaded then (lii_scan_upload_next): read in place, behind the wait
    bool want_late = false;
    const void* want_dev = nullptr;          // lii_scan_register -> update_on_device: the job in progress names this next scan
    int want_n = 0;
    float want_leaf = 0.f;
    long long timeout_ticks = 200000000ll;   // 2 s of the 100 MHz clock (LII_PREARM_TIMEOUT_MS)
    long long n_used = 0, n_cancelled = 0, n_expired = 0;  // LII_DIAG
  } pre;

  // ---- pinned staging
  float4* h_stage = nullptr;     // max(max_scan, max_map) float4
  size_t h_stage_elems = 0;
  double* h_small = nullptr;     // 4096 doubles

  // ---- calibration
  struct CalibState {  // lii_capi_calib.cpp: the LI_init evaluators' buffers
    double *d_cal_imu = nullptr, *d_cal_lidar = nullptr, *d_cal_params = nullptr, *d_cal_out = nullptr;
    int n_cal = 0;

    bool li_init_device = false;  // lii_li_init_set_device: zero-phase filter + cross-correlation of lii_li_init_run on the device
  } cal;
  void* ingest = nullptr;  // lii_ingest.hip state (frames of the last driver message)

  // ---- comm
  struct CommState {  // lii_capi_comm.cpp: the communicator of a sharded job
    ncclComm_t comm = nullptr;   // RCCL transport (ranks on several nodes, or forced)
    MailboxHost mailbox;         // node-local transport: the exchange happens inside k_reduce_solve
    unsigned long long* d_mb_seq = nullptr;
    unsigned int* d_gather_ticket = nullptr;  // the list exchange of lii_map_incremental (lii_exchange.hip): its ticket word,
    unsigned char* d_gx = nullptr;            // RCCL form of the list exchange: send block | N gathered blocks | N headers | pointer tables
    size_t gx_block = 0; int gx_ranks = 0;    // ... laid out for this block size (64 + 16 max_scan_points) and this many ranks
    unsigned long long gather_seq = 0;        // ... and the exchanges enqueued so far (the ranks call in lock-step: the same on all)
    long long mailbox_timeout_ticks = 3000000000ll;  // 30 s (LII_MAILBOX_TIMEOUT_S): ranks may start a scan seconds apart
    int n_ranks = 1, rank = 0;
    std::string comm_why;           // which transport this rank ended up with and why (lii_comm_describe)
    bool library_partition = true;  // lii_comm_set_partition: the library splits the down-sampled cloud over the ranks (every rank
                                    // hands over the whole scan); false: the caller hands every rank its own points
    bool voxel_partition = false;   // lii_comm_set_partition(h, 2): ... by VOXEL where the filter is fused into the de-skew (this rank
                                    // filters and registers the voxels whose key hashes to it), by index otherwise
  } net;

  // ---- profiling
  struct ProfState {  // lii_set_profiling / lii_last_timings / lii_last_kernel_profile, LII_DIAG
    bool kp_active = false;            // inside a lii_scan_register that is being profiled launch by launch
    int prof_mode = 0;                 // the last lii_set_profiling value; 3: an event in front of every launch of lii_scan_register
    std::vector<hipEvent_t> kp_ev;     // ... the events (created on demand, reused),
    std::vector<int> kp_kind;          // ... kind * 64 + iteration of the launch behind each (kind LII_KP_KINDS: end mark)
    int kp_n = 0;
    lii_kernel_profile kprof{};
    bool profiling = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool bracket_events = false;       // LII_PROF_BRACKET=1: events recorded around the k-NN launches instead of inside their dispatch
    hipEvent_t ev_it[32] = {};
    unsigned int ev_it_due = 0u;       // iterations whose pair of ev_it holds a k-NN launch that has not been read yet (harvest_knn_events)  // per-iteration brackets of the k-NN kernel in the device-driven loop
    double timings[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double host_map_us[2] = {0, 0};  // LII_DIAG: per update - waiting for the map update in flight (commit_map), enqueueing the map update behind the passes
    double host_us[6] = {0, 0, 0, 0, 0, 0};  // LII_DIAG: per lii_scan_register - entry -> first launch, -> pre-processing enqueued, -> loop enqueued, -> result; calls; gap between calls
    std::chrono::steady_clock::time_point host_last_return;
    double host_loop_enq_us = 0;
  } prof;
};


namespace lii_impl {
using namespace lii;

int fail(lii_handle h, int code, const std::string& msg);
#define HIPCHK(h, call)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      return lii_impl::fail(h, LII_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));          \
  } while (0)
template <class T>
hipError_t dmalloc(T** p, size_t n) {
  return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T));
}
inline unsigned int next_pow2(unsigned int v) {
  unsigned int p = 1;
  while (p < v) p <<= 1;
  return p;
}
// lii_capi.cpp
int kp_mark(lii_handle h, int kind, int it = 0);
void harvest_knn_events(lii_handle h);
GridView grid_view(const lii_context* c);
RegistrationBuffers reg_buffers(const lii_context* c);
PoseArg pose_of(const lii_state& s);
int resolve_n_body(lii_handle h);
bool fuse_filter(lii_handle h, float leaf);  // does the de-skew of this scan fill the hashed voxel filter's table on the way?
int pcl_order(lii_handle h, const int** perm);
void extent_discard(lii_handle h);
int scan_materialize(lii_handle h);
bool gate_move(lii::GateState* st, unsigned long long seq, unsigned long long to);  // the state word: armed -> `to`, if still armed
void prearm_cancel(lii_handle h);  // a gated de-skew launch that waits on the stream is told to end (every entry point that uses the stream calls this first)  // a frame selected by lii_frame_select and not read yet -> d_scan (lii_scan_set_device)
unsigned long long* extent_of_scan(lii_handle h);
MailboxView mailbox_view(lii_handle h);
lii::GatherView gather_view(lii_handle h);  // .peers == nullptr: this job has no list exchange (single rank, host-memory mailbox, RCCL)
// lii_capi_map.cpp
int build_index(lii_handle h, int n, int extra_blocks = 0);
void note_list_sizes(lii_handle h, int n_add, int n_nodown);
int map_join(lii_handle h);
constexpr int kMapFlagSeqAt = 40;  // (h_mapflag: 64 ints; [0, kMapCtrWords + 8): the counters and list sizes)
int map_update_early(lii_handle h);  // 1: enqueued behind the passes of the update in progress, 0: not possible this time, < 0: error
int commit_map(lii_handle h);
int map_counters(lii_handle h, bool already_synced = false);
lii::WinKeep win_keep_view(lii_handle h);  // the window as the update's launches keep it current (win == nullptr: nothing to keep)
int map_gather(lii_handle h, int* n_out);
int map_rebuild(lii_handle h, int extra_blocks);
int map_apply(lii_handle h, const float4* list, int n_list, bool downsample, const float4* extra, int n_extra, bool beside = false,
              const int* n_list_dev = nullptr, const int* n_extra_dev = nullptr, bool count_events = true, bool prefilled = false);
// lii_capi_comm.cpp
void comm_drop(lii_handle h);
void partition_refresh(lii_handle h);
int lists_exchange_rccl(lii_handle h, hipStream_t s);  // (lii_capi_comm.cpp) the list exchange of lii_map_incremental over ncclAllGather; synchronises the stream once  // (lii_capi_comm.cpp) the voxel filter's view of the job after the communicator or its partition changed
}  // namespace lii_impl
