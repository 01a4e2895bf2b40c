// Host-callable launchers of the kernels in lii_kernels.hip / lii_sort.hip (internal; not the C-ABI).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>

#include <string>

#include "lii_device.h"

struct lii_context;
// internal hooks of the handle for the translation units that live beside lii_capi.cpp (not exported in the C-ABI header)
int lii_internal_fail(lii_context* h, int code, const std::string& msg);
int lii_internal_scan_defer(lii_context* h, const void* dev_float4, int32_t n);  // (lii_capi.cpp) lii_frame_select's hand-over
int lii_internal_scan_materialize(lii_context* h);
int lii_internal_in_wait_hook(lii_context* h);       // (lii_capi.cpp) 1: inside lii_scan_job::while_waiting of a registration under way
int lii_internal_scan_is_deferred(lii_context* h);  // (lii_capi.cpp) 1: a selected frame that nobody has read yet
void lii_internal_prearm_cancel(lii_context* h);  // (lii_capi.cpp) see lii_impl::prearm_cancel  // (lii_capi.cpp) a selected frame nobody has read yet -> the handle's own scan buffer
hipStream_t lii_internal_stream(lii_context* h);
void** lii_internal_ingest_slot(lii_context* h);

namespace lii {
void ingest_destroy(void* slot);  // lii_ingest.hip

struct UndistArgH { double endR[9], endp[3], RLI[9], TLI[3]; };
struct CvArgH { double omega[3], vel[3], endR[9]; };

// map index
void launch_map_keys(const float4* pts, int n, float inv_cs, unsigned long long* keys, unsigned int* idx, hipStream_t s);
void launch_map_gather(const float4* src, const unsigned int* idx, int n, float4* dst, hipStream_t s);
void launch_block_flags(const unsigned long long* keys, int n, unsigned int* flags, hipStream_t s);
void launch_table_clear(BlockEntry* blocks, unsigned int cap, hipStream_t s);
void launch_win_bbox(const BlockEntry* blocks, unsigned int cap, unsigned int* box, hipStream_t s);
void launch_win_fill(const BlockEntry* blocks, unsigned int cap, const uint2* cells, uint2* win, const int org[3], const int dim[3], hipStream_t s);
void launch_cells_fill(const unsigned long long* keys, const unsigned int* ranks, int n, BlockEntry* blocks,
                       unsigned int block_mask, uint2* cells, unsigned long long* key_of_id, hipStream_t s);
// registration
// (`pose`: device memory on every path - a host-driven pass uploads it first)
// epoch: the number of this search launch (> 0, RegistrationBuffers::flag_*) and of the fit launch behind it; 0: no list of unfinished queries
// ev_start / ev_stop (both or none): the launch's own dispatch carries the two events (hipExtLaunchKernelGGL: the time stamps of the
// dispatch packet's completion signal) - the kernel's duration without a barrier packet in front of and behind it on the stream
void launch_knn(const GridView& g, const RegistrationBuffers& rb, const PoseArg* pose,
                const IekfCtrl* ctrl, int forced, double* search_pose_out, hipStream_t s, int epoch,
                hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
#ifdef LII_FALLBACK_TRACE
void fb_trace_read(unsigned long long out[32]);  // (measurement builds: the completion's phase sums, lii_kernels.hip)
#endif
void launch_knn_complete(const GridView& g, const RegistrationBuffers& rb, hipStream_t s);
void launch_complete_listed(const GridView& g, const RegistrationBuffers& rb, const IekfCtrl* ctrl, int forced, hipStream_t s, int epoch);  // (k_complete_listed: behind search launch `epoch`)
void launch_fit_reduce(const GridView& g, const RegistrationBuffers& rb, const PoseArg* pose,
                       const IekfCtrl* ctrl, int forced, int imu_en, double plane_thr, double rinv, hipStream_t s, int epoch);
void launch_reduce91(const RegistrationBuffers& rb, double* out91, const IekfCtrl* ctrl, int forced, hipStream_t s, int epoch);  // epoch: of the fit launch whose columns it sums
void launch_reduce_solve(const RegistrationBuffers& rb, unsigned long long* gran, IekfCtrl* c, IekfResult* res, const MailboxView& mb,
                         hipStream_t s, int epoch);
void launch_mailbox_allreduce(double* out91, const MailboxView& mb, hipStream_t s);
// node-local mailbox (lii_mailbox.cpp)
struct MailboxHost {
  void* map = nullptr;        // the shared segment in this process (rendezvous; in the host-memory form also the slots)
  size_t bytes = 0;
  double* dev_slots = nullptr;  // host-memory form: device address of the segment's slot area
  bool registered = false;
  char name[64] = {0};        // non-empty while the segment still has a name to unlink
  // HBM form: the own slot area (fine-grained device memory, exported through a HIP IPC handle), the peers' areas as opened
  // here, and the device-resident table of all of them (MailboxView::peers)
  double* own = nullptr;
  double* peer_ptr[64] = {nullptr};
  double** d_peers = nullptr;
  int n_peers = 0;
  // ... and behind the slots of every area the gather area of the list exchange (GatherView; gather_points > 0 at set-up)
  unsigned char** d_gather_peers = nullptr;
  size_t gather_block = 0;
  int gather_cap = 0;
};
size_t mailbox_segment_bytes(int n_ranks);
// want_hbm: try the peer-mapped HBM form first (falls back to host memory inside the same rendezvous when a rank cannot
// export / open the handles).  Returns 0 on success (m->d_peers != nullptr: HBM form; else m->dev_slots: host-memory form).
// gather_points: capacity (float4) of one rank's lists in the exchange of lii_map_incremental (HBM form only; 0: no gather areas).
int mailbox_open(const uint8_t id[128], int n_ranks, int rank, double wait_s, bool want_hbm, int gather_points, MailboxHost* m, std::string* why);
// the list exchange (lii_exchange.hip): push this rank's lists (sizes in counts[0..1], device) into every rank's gather area, wait for
// all ranks' lists of exchange number `seq` (1, 2, ... - the same on every rank) and leave them concatenated in rank order in
// dst_add / dst_nodown (may be the sources), the totals in counts[0..1]; counts[err_at] = 1 when a rank did not deliver in time.
void launch_lists_push(const GatherView& gv, const float4* src_add, const float4* src_nodown, const int* counts, unsigned int* ticket,
                       unsigned long long seq, hipStream_t s);
void launch_lists_collect(const GatherView& gv, unsigned long long seq, float4* dst_add, float4* dst_nodown, int* counts, int err_at, hipStream_t s);
void launch_lists_exchange(const GatherView& gv, const float4* src_add, const float4* src_nodown, int* counts, int err_at, unsigned int* ticket,
                           unsigned long long seq, float4* dst_add, float4* dst_nodown, hipStream_t s);
void mailbox_close(MailboxHost* m);
void launch_loop_resume(IekfCtrl* c, unsigned int plan_mask, hipStream_t s);
void launch_iekf_solve(IekfCtrl* c, const double* ne, IekfResult* res, hipStream_t s);
int register_blocks(int n);
// undistortion
void launch_time_extent(const float4* pts, int n, unsigned long long* extent, unsigned long long* extent_next, float4* copy_to,
                        const void* ctrl_src, void* ctrl_dst, size_t ctrl_bytes, hipStream_t s);
// voxel grid
void launch_voxel_minmax(const float4* pts, int n, unsigned int* mm, unsigned int* mm_next, hipStream_t s);
void launch_voxel_keys(const float4* pts, int n, const unsigned int* mm, const unsigned int* bbox_rows, int n_rows, float leaf,
                       unsigned long long* keys, unsigned int* pcl_keys, int* filtered_dev,
                       unsigned long long* samples, int sample_width, hipStream_t s);
// the voxel grid by hashing (lii_scan.hip: k_vhash_*): `slots` holds voxel_hash_slots(max_n) 64-byte slots, all free between
// scans (launch_voxel_hash_clear once; every emit leaves the slots it used free again)
struct VoxelHashBuffers {
  void* slots;
  unsigned int *slot_of, *next;   // per input point
  unsigned long long* counts;     // per workgroup of 256 points: (run number << 32) | voxels owned by the workgroup
  unsigned int* crowded;          // one word: the longest member list behind a slot so far
  // part_world > 1: a job that shares the FUSED filter (insert inside the de-skew) by voxel - rank part_rank inserts the voxels
  // whose key hashes to it and emits at most voxel_partition_bound(n, part_world) of them (else *part_overflow = 1, mapped host memory)
  int part_world, part_rank;
  int* part_overflow;
};
size_t voxel_hash_slots(int max_n);
int voxel_partition_bound(int n, int world);
void launch_voxel_hash_clear(const VoxelHashBuffers& vh, size_t slots, hipStream_t s);
void launch_voxel_hash(const VoxelHashBuffers& vh, const float4* pts, int n, const unsigned int* mm, const unsigned int* bbox_rows,
                       int n_rows, float leaf, float4* out, int* n_out, int* filtered, unsigned int* pcl_out, int stages, unsigned int epoch,
                       hipStream_t s, int test_late = 0 /* LII_TEST=emit_late: see k_vhash_emit */);
// de-skew (k_deskew_imu / k_deskew_cv): where the scan comes from and goes to, what is known about it, what rides along
struct DeskewPlan {
  const float4* in;     // the scan as it arrived (a caller's device buffer, or == out)
  float4* out;          // the de-skewed scan
  int n;
  int sorted;           // ascending time order: no time extent needed
  const unsigned long long* extent;  // !sorted: the result of launch_time_extent
  unsigned int* bbox_rows;           // one bounding-box row per workgroup of 256 points is left here
  float leaf;                        // vh != nullptr: the leaf of the hashed voxel filter whose insert rides along
  const VoxelHashBuffers* vh;
  const void* ctrl_src;              // ctrl_bytes > 0: one extra workgroup copies this (pinned host memory) to ctrl_dst
  void* ctrl_dst;
  size_t ctrl_bytes;
#ifdef LII_GAP_TRACE
  unsigned long long* gap;  // measurement builds only (tools/ab_build.sh ... -DLII_GAP_TRACE): words 200 .. 202 of the granule buffer
#endif
};
void launch_deskew_imu(const DeskewPlan& p, const double* poses_host, const double* poses_dev, int K, const UndistArgH& u, hipStream_t s);
// THE PRE-ARMED PROLOGUE (round 6).  The de-skew launch of the NEXT scan is enqueued while the current scan's passes still run, and waits
// on the device for a record the host writes once it knows what that launch needs - the propagated state and the IMU pose table, which
// depend on the current scan's result.  What the host then pays between two scans is a few stores, not a launch: the dependent chain
// result -> host -> first kernel of the next scan loses the runtime's launch path and the command processor's fetch + dispatch
// (profiles/r06_chain.md).
// The record: {K, UndistArgH, K x 22 doubles} in DEVICE memory, which the host writes directly (large BAR: posted writes, no round
// trip; a first form kept it in mapped host memory and had one workgroup fetch and re-publish it - three PCIe round trips and two
// cache-wide fences on the critical path, 10 us slower than the launch it replaced).  It is laid out in 64-byte lines of seven
// payload doubles and a TAG (seq << 2 | kGateGo); the host writes the payload, then the tags, line 0's tag last, a store fence between
// the three - so a workgroup that sees line 0's tag finds every line complete, and a line it finds with another tag is a stale cached
// copy (the record lives in a ring of kGateRing slots), fetched again past the caches.
// The state word (GateState::word, mapped HOST memory) is seq << 2 | state; the host arms it before it enqueues the launch, and exactly
// one side moves it on, by compare-and-swap: the host to GO (then it writes the record) or CANCEL (this scan is not the one that was
// announced, or another entry point needs the stream), the launch itself to EXPIRED when nobody came for `timeout_ticks` (100 MHz) -
// it then ends having touched nothing, so a forgotten launch can delay the stream by that long, never hang it.
constexpr unsigned long long kGateArmed = 0ull, kGateGo = 1ull, kGateCancel = 2ull, kGateExpired = 3ull;
constexpr int kGateMaxPoses = 29;                                   // pose tables the pre-armed form takes (longer ones: the plain launch)
constexpr int kGateLines = (25 + 22 * kGateMaxPoses + 6) / 7;       // 64-byte lines of a record (95)
constexpr int kGateRing = 64;                                       // record slots (seq % kGateRing)
struct GateState { unsigned long long word; };                      // pinned, device-mapped host memory: seq << 2 | kGate*
struct DeskewGate {
  GateState* host;                  // the state word (polled by the gate workgroup alone; one compare-and-swap over PCIe if it gives up)
  const double* rec;                // this launch's record slot in device memory (kGateLines x 8 doubles)
  unsigned long long* dev_flag;     // seq << 2 | kGateCancel when the launch is to end without running (published by the gate workgroup)
  unsigned long long seq;
  long long timeout_ticks;
  int late_load;                    // 1: the scan itself may still be on its way when the launch starts (lii_scan_upload_next): no point is read before the record is there
};
void launch_deskew_imu_gated(const DeskewPlan& p, const DeskewGate& gate, hipStream_t s);
void launch_deskew_cv(const DeskewPlan& p, const CvArgH& a, hipStream_t s);
// calibration
void launch_calib_eval(int stage, const double* imu, const double* lidar, int n, const double* params, double* out,
                       hipStream_t s);

// device-side map maintenance (lii_map.hip)
void launch_map_decide_compact(const RegistrationBuffers& rb, const PoseArg& ps, double fsd, int have_search, unsigned long long* blk_counts, unsigned int epoch,
                               float4* world, float4* dst_add, float4* dst_nodown, int* counts, int bound_add, int bound_nodown, hipStream_t s,
                               const IekfCtrl* guard = nullptr, int seq = 0, int test_late = 0,
                               // (hkey != nullptr: the hash insert of the fold that follows rides in this launch - table sized by bound_add)
                               unsigned long long* hkey = nullptr, unsigned long long* hbest = nullptr, unsigned int* slot_of = nullptr, float ds = 0.f,
                               unsigned int* ins_flag = nullptr, int* events = nullptr);  // blk_counts: one word per 256 points; epoch: the number of this run (never 0); guard: see k_map_decide
void launch_map_publish(const int* ctr, int n_words, int* host, int seq_at, int seq, hipStream_t s);
void launch_add_keys(const float4* pts, int n, const int* n_dev, float ds, unsigned long long* keys, unsigned int* idx, int* events, hipStream_t s);
void launch_add_fold(const float4* add_pts, const unsigned long long* keys, const unsigned int* idx, int n, float ds, const GridView& g,
                     unsigned char* tomb, float4* ins_pts, unsigned int* ins_flag, unsigned int* events, unsigned int* tp, unsigned int* work,
                     int* ctr, unsigned int work_cap, hipStream_t s);
size_t add_hash_slots(int max_n);
// the cells of the inserts found / created inside the fold launch (k_add_fold8<true, true>) instead of by launch_ins_cells behind it
struct FoldCellsH {
  BlockEntry* blocks; unsigned int mask; unsigned int tables_cap; unsigned int* ins_e; float4* dropped; unsigned int drop_cap; unsigned long long* key_of_id;
  const float4* list2; int n2; const int* n2_dev; unsigned int* ins_e2;
};
void launch_add_fold_hashed(const float4* add_pts, int n, const int* n_dev, float ds, const GridView& g, unsigned long long* hkey,
                            unsigned long long* hbest, unsigned int* slot_of, unsigned char* tomb, float4* ins_pts, unsigned int* ins_flag,
                            unsigned int* events, unsigned int* tp, unsigned int* work, int* ctr, unsigned int work_cap, hipStream_t s,
                            bool inserted = false, const FoldCellsH* cells = nullptr);
// in-place map update (lii_map.hip)
void launch_ins_cells(const float4* list, const unsigned int* flags, int n, const int* n_dev, const float4* list2, int n2, const int* n2_dev, unsigned int* ins_e2,
                      BlockEntry* blocks, unsigned int mask, float inv_cs, unsigned int tables_cap, unsigned int* ins_e, unsigned int* tp,
                      unsigned int* work, int* ctr, unsigned int work_cap, float4* dropped, unsigned int drop_cap, unsigned long long* key_of_id, hipStream_t s);
void launch_cell_apply(const unsigned int* work, uint2* cells, unsigned int* cell_cap, float4* pts, unsigned char* tomb, unsigned int* tp, int* ctr,
                       unsigned int pts_cap, int launch_bound, const WinKeep& wk, hipStream_t s);
void launch_ins_write(const float4* list, const unsigned int* ins_e, int n, const int* n_dev, const float4* list2, const unsigned int* ins_e2, int n2, const int* n2_dev,
                      uint2* cells, const unsigned int* cell_cap, float4* pts, int* ctr, float4* dropped, unsigned int drop_cap, hipStream_t s,
                      int* host, int n_words, int seq_at, int seq, const WinKeep& wk)  /* host != nullptr: the last workgroup publishes the counters there */;
void launch_cell_caps(const uint2* cells, int n_entries, unsigned int* caps, hipStream_t s);
void launch_spread(const float4* src, uint2* cells, unsigned int* cell_cap, const unsigned int* caps, const unsigned int* capsum, int n_entries,
                   float4* dst, int* ctr, int n_valid, int n_blocks, hipStream_t s);
void launch_cell_counts(const uint2* cells, int n_entries, unsigned int* cnt, hipStream_t s);
void launch_gather_live(const float4* pts, const uint2* cells, const unsigned int* cntsum, int n_entries, float4* dst, int dst_cap, hipStream_t s);
void launch_box_tomb_cells(const float4* pts, const uint2* cells, int n_entries, const float* boxes, int n_boxes, unsigned char* tomb, unsigned int* tp,
                           unsigned int* work, int* ctr, unsigned int work_cap, hipStream_t s);

// rocPRIM wrappers (lii_sort.hip)
size_t sort_temp_bytes(int max_n);
void sort_pairs_u64(void* temp, size_t temp_bytes, const unsigned long long* kin, unsigned long long* kout,
                    const unsigned int* vin, unsigned int* vout, int n, hipStream_t s);
// back half of the voxel-grid filter (lii_vsort.hip): sample sort of the (key, index) pairs + centroids
struct VoxelSortBuffers {
  const unsigned long long* keys_in;   // kVoxKeyBits-wide sort keys (lii_device.h)
  unsigned long long* keys_out;
  unsigned int* idx_out;
  unsigned long long* comp;       // n composites, bucket order
  unsigned long long* splitters;  // 2048
  const unsigned long long* samples;  // 4096, written by k_voxel_keys (voxel_sort_plan)
  unsigned int* hist;             // voxel_sort_hist_elems(max n), zero-initialised once
  unsigned short* bucket_of;      // n
  const unsigned int* pcl_in;     // PCL voxel index of every input point ...
  unsigned int* pcl_out;          // ... and of every output voxel (the order the reference's filter would emit them in)
  unsigned int* max_run;          // optional: receives (atomicMax) the largest number of points of one voxel when it exceeds 8
};
size_t voxel_sort_hist_elems(int max_n);
struct VoxelSortPlan { int buckets, samples, width, strata; };
VoxelSortPlan voxel_sort_plan(int n);
// sort + one centroid per voxel: out[0 .. *n_out) in key order, *n_out (device) = number of occupied voxels
void launch_voxel_sort_centroids(const VoxelSortBuffers& vb, const float4* pts, int n, float4* out, int* n_out, hipStream_t s);
void sort_pairs_u32(void* temp, size_t temp_bytes, const unsigned int* kin, unsigned int* kout, const unsigned int* vin,
                    unsigned int* vout, int n, hipStream_t s);
void inclusive_scan_u32(void* temp, size_t temp_bytes, const unsigned int* in, unsigned int* out, int n, hipStream_t s);

float ord_to_float(unsigned int o);

}  // namespace lii
