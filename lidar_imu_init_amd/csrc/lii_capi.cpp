// libliinit_hip — C-ABI implementation (host side): life cycle of the handle, scan in / de-skew / voxel grid / downloads, profiling.
// Declarations and the reference code each entry point replaces: include/liinit_hip.h.  Device work: lii_scan.hip, lii_kernels.hip,
// lii_vsort.hip, lii_sort.hip on ONE stream per handle.  The other entry points: lii_capi_map.cpp, lii_capi_register.cpp,
// lii_capi_comm.cpp, lii_capi_calib.cpp (lii_context.h).
// There is deliberately no CPU implementation of any stage here: without a usable gfx950 device
// lii_create fails with LII_ERR_NO_DEVICE.
#include "lii_context.h"

using namespace lii_impl;

namespace lii_impl {

thread_local std::string g_err;

// The profiling events only bracket kernels of ONE stream for hipEventElapsedTime: nothing those kernels wrote has to become visible to the
// host or to another device when an event is recorded, so the system-scope release a default event performs there - an L2 write-back
// between the search launch and the fit launch that reads its lists, on the timed path of every profiled step - is left out.
#ifndef LII_PROF_EVENT_FLAGS
#define LII_PROF_EVENT_FLAGS hipEventDisableSystemFence
#endif
static constexpr unsigned kProfEventFlags = LII_PROF_EVENT_FLAGS;

// The k-NN launches of the last profiled update: their events are read here, not behind the update (see update_on_device).
void harvest_knn_events(lii_handle h) {
  const unsigned int due = h->prof.ev_it_due;
  h->prof.ev_it_due = 0u;
  for (int it = 0; it < 16; it++) {
    if (!((due >> it) & 1u)) continue;
    float kk = 0;
    if (hipEventElapsedTime(&kk, h->prof.ev_it[2 * it], h->prof.ev_it[2 * it + 1]) != hipSuccess) continue;
    h->prof.timings[7] += kk;
    h->prof.timings[5] += 1;
  }
}

int fail(lii_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  g_err = msg;
  return code;
}
// lii_set_profiling(h, 3): an event in front of the launch(es) that follow; `it` = the iteration of a loop launch
int kp_mark(lii_handle h, int kind, int it) {
  if (h->prof.prof_mode != 3) return LII_OK;
  if (h->prof.kp_n >= (int)h->prof.kp_ev.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, kProfEventFlags) != hipSuccess) return fail(h, LII_ERR_HIP, "hipEventCreate (kernel profile)");
    h->prof.kp_ev.push_back(e);
    h->prof.kp_kind.push_back(0);
  }
  if (hipEventRecord(h->prof.kp_ev[size_t(h->prof.kp_n)], h->stream) != hipSuccess) return fail(h, LII_ERR_HIP, "hipEventRecord (kernel profile)");
  h->prof.kp_kind[size_t(h->prof.kp_n)] = kind * 64 + std::min(it, 63);
  h->prof.kp_n++;
  return LII_OK;
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
  g.win = (c->win_valid && !c->map_dirty) ? c->d_win : nullptr;
  g.wx0 = c->win_org[0]; g.wy0 = c->win_org[1]; g.wz0 = c->win_org[2];
  g.wnx = c->win_dim[0]; g.wny = c->win_dim[1]; g.wnz = c->win_dim[2];
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
  rb.flag_count = c->d_flags;
  rb.flag_list = reinterpret_cast<float4*>(c->d_flags + 4);
  rb.shard_rank = c->net.rank;
  rb.shard_world = (c->net.n_ranks > 1 && c->net.library_partition && !c->body_partitioned) ? c->net.n_ranks : 1;
  if (c->solo_share < -1 && c->net.n_ranks <= 1) { rb.shard_world = -c->solo_share; rb.shard_rank = 0; }
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
// The insert of the hashed voxel filter rides in the de-skew kernel when the hashed filter is the one in use for this leaf.  A job
// that shares the filter by voxel (lii_comm_set_partition(h, 2)) always uses it - every rank must take the same path, and the
// watch that changes over to the sort sees a different share of the voxels on every rank - unless the sort has been pinned.
bool fuse_filter(lii_handle h, float leaf) {
  if (!(leaf > 0.f) || h->voxel_sort || h->no_fuse) return false;
  if (h->vh.part_world > 1) return true;
  return h->vh_mode == 1 && (h->vh_pinned || leaf == h->vh_leaf);
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
  h->scan_pending = nullptr;  // (whatever replaces the scan replaces a selected frame that nobody has read as well)
  h->scan_pending_n = 0;
}
// The state word of the gate record is moved on by exactly one side (compare-and-swap on both): true = this call moved it.
bool gate_move(lii::GateState* st, unsigned long long seq, unsigned long long to) {
  unsigned long long expect = (seq << 2) | lii::kGateArmed;
  return __atomic_compare_exchange_n(&st->word, &expect, (seq << 2) | to, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
}
void prearm_cancel(lii_handle h) {
  if (!h || !h->pre.armed) return;
  h->pre.armed = false;
  if (gate_move(h->pre.state, h->pre.seq, lii::kGateCancel)) h->pre.n_cancelled++;
  else h->pre.n_expired++;  // (the launch had given up already)
}
int scan_materialize(lii_handle h) {
  if (!h->scan_pending) return LII_OK;
  const float4* src = h->scan_pending;
  const int n = h->scan_pending_n;
  return lii_scan_set_device(h, src, n);  // (clears the pending frame through extent_discard, then copies from `src`)
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
lii::GatherView gather_view(lii_handle h) {
  lii::GatherView g;
  g.peers = h->net.mailbox.d_gather_peers;
  g.block_bytes = h->net.mailbox.gather_block;
  g.cap_points = h->net.mailbox.gather_cap;
  g.n_ranks = h->net.n_ranks;
  g.rank = h->net.rank;
  g.timeout_ticks = h->net.mailbox_timeout_ticks;
  return g;
}
MailboxView mailbox_view(lii_handle h) {
  MailboxView v;
  v.slots = h->net.mailbox.dev_slots;
  v.peers = h->net.mailbox.d_peers;
  v.seq = h->net.d_mb_seq;
  v.n_ranks = h->net.n_ranks;
  v.rank = h->net.rank;
  v.timeout_ticks = h->net.mailbox_timeout_ticks;
  v.handoff_ticks = h->test_sum_lost ? 20000000ll : 200000000ll;  // 0.2 s under test, 2 s otherwise
  v.test_drop_sum = h->test_sum_lost ? 1 : 0;
  return v;
}

}  // namespace lii_impl

int lii_internal_fail(lii_context* h, int code, const std::string& msg) { return fail(h, code, msg); }
int lii_internal_scan_materialize(lii_context* h) { return scan_materialize(h); }
int lii_internal_scan_is_deferred(lii_context* h) { return h && h->scan_pending ? 1 : 0; }
int lii_internal_in_wait_hook(lii_context* h) { return h && h->in_wait_hook ? 1 : 0; }
void lii_internal_prearm_cancel(lii_context* h) { prearm_cancel(h); }
// lii_frame_select's hand-over (lii_ingest.hip): the frame becomes the current scan WITHOUT being copied - round 6: the copy + time
// extent launch of lii_scan_set_device was 5 - 8 us per sub-frame in front of a registration that reads the frame in place anyway
// and, told that it is time-sorted, needs no extent.
int lii_internal_scan_defer(lii_handle h, const void* dev_float4, int32_t n) {
  if (!h || (!dev_float4 && n > 0) || n < 0) return fail(h, LII_ERR_INVALID, "lii_frame_select: bad frame");
  if (n > h->cfg.max_scan_points) return fail(h, LII_ERR_CAPACITY, "lii_frame_select: frame larger than max_scan_points");
  extent_discard(h);
  h->scan_buf_idle = false;
  h->bbox_rows = 0;
  h->scan_pending = n > 0 ? static_cast<const float4*>(dev_float4) : nullptr;
  h->scan_pending_n = n > 0 ? n : 0;
  h->n_scan = n;
  h->n_body = 0;
  h->n_body_pending = false;
  h->have_search = false;
  return LII_OK;
}

int lii_internal_li_init_on_device(lii_context* h) { return h && h->cal.li_init_device ? 1 : 0; }
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
  if (const char* v = std::getenv("LII_DIAG")) h->diag = std::atoi(v) != 0;
  if (const char* v = std::getenv("LII_PROF_BRACKET")) h->prof.bracket_events = std::atoi(v) != 0;
  if (const char* v = std::getenv("LII_VOXEL_FILTER")) {
    h->voxel_sort = std::string(v) == "sort";
    if (std::string(v) == "hash") { h->vh_pinned = true; h->vh_mode = 1; }
  }
  if (const char* v = std::getenv("LII_KNN_PLAN")) h->knn_plan = std::atoi(v) != 0;
  if (const char* v = std::getenv("LII_WINDOW")) h->use_window = std::atoi(v) != 0;
  if (const char* v = std::getenv("LII_WINDOW_KEEP")) h->win_keep = std::atoi(v) != 0;
  if (const char* v = std::getenv("LII_TEST")) {
    // arrangements the test-suite and the A/B measurements ask for, comma-separated: "map_tight" (an in-place map update without
    // spare room), "plan_force=<mask>" (a launch plan that is wrong on purpose), "host_solve" (the iteration loop driven from the
    // host around lii_iekf_iterate with the literal two-inversion algebra), "sync_result" (every update ends with
    // hipStreamSynchronize instead of polling the result word), "graph" (the enqueued passes of an update replayed from a
    // captured hipGraph), "pred_small" (lii_map_incremental predicts list sizes that are always too small), "fold_sort"
    // (lii_map_incremental folds its list through the batch sort, as lii_map_add_points does, instead of the hash table), "no_fuse"
    // (lii_scan_register keeps the de-skew and the insert of the hashed voxel filter in separate launches), "no_fast" (a
    // time-sorted scan takes the general path of lii_scan_register too: k_time_extent in front of the de-skew), "force_rebuild"
    // (every in-place map update takes the branch that rebuilds the index first), "no_gather" (a sharded job sets up no gather areas:
    // its map update repeats the search instead of exchanging the lists), "solo_share=<N>" / "solo_share=-<N>" (kernel-timing
    // rehearsal: ONE process works on rank 0's share of an N-rank job split by voxel / by index, nothing is exchanged - the
    // durations of a rank's launches without N devices; the result is that of a part of the cloud), "emit_late" (every seventh
    // workgroup of the voxel filter's emit and of the map update's decision launch publishes its count only when it is done: the
    // workgroups above it take the path of a launch whose workgroups are not all resident and count its block themselves)
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
    h->no_gather = t.find("no_gather") != std::string::npos;
    h->test_emit_late = t.find("emit_late") != std::string::npos;
    h->test_sum_lost = t.find("sum_lost") != std::string::npos;
    const size_t qs = t.find("solo_share=");
    if (qs != std::string::npos) h->solo_share = int(std::strtol(t.c_str() + qs + 11, nullptr, 0));
  }
  if (const char* mf = std::getenv("LII_MAP_FUSE")) h->map_fuse = mf[0] != '0';
  if (const char* wc = std::getenv("LII_WIDE_COMPLETION")) h->wide_enabled = wc[0] != '0';
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
  CK(dmalloc(&h->d_block_key, h->cells_cap_blocks));
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
  CK(hipMemset(h->d_u32_b, 0, sizeof(unsigned int) * NM));  // (k_map_decide's block words: run number 0 is never used)
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
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->pre.state), 64, hipHostMallocMapped));
  std::memset(h->pre.state, 0, 64);
  CK(dmalloc(&h->pre.d_ring, size_t(lii::kGateRing) * lii::kGateLines * 8));
  CK(hipMemset(h->pre.d_ring, 0, sizeof(double) * size_t(lii::kGateRing) * lii::kGateLines * 8));
  CK(dmalloc(&h->pre.d_flag, 8));
  CK(hipMemset(h->pre.d_flag, 0, 64));
  {  // the host writes the record straight into device memory: only where the whole of it is visible to the host (large BAR)
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, h->device) != hipSuccess) large_bar = 0;
    h->pre.enabled = large_bar != 0;
  }
  if (const char* v = std::getenv("LII_PREARM")) h->pre.enabled = h->pre.enabled && std::atoi(v) != 0;
  if (const char* v = std::getenv("LII_PREARM_TIMEOUT_MS")) h->pre.timeout_ticks = std::max(1ll, (long long)(std::atof(v) * 1e5));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_res), sizeof(IekfResult), hipHostMallocMapped));
  std::memset(h->h_res, 0, sizeof(IekfResult));
  partition_refresh(h);
  CK(hipEventCreateWithFlags(&h->ev_poses, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&h->ev_mapflag, hipEventDisableTiming));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_mapflag), 256, hipHostMallocMapped));  // (written by k_map_publish)
  std::memset(h->h_mapflag, 0, 256);
  CK(hipEventCreateWithFlags(&h->ev_lists, hipEventDisableTiming));
  CK(hipMemset(h->d_counter, 0, 16));
  h->partial_stride = register_blocks(int(N)) + std::max(lii::kCompletionBlocks, lii::kCompletionBlocksPre) + 8;  // (+ the columns of the fit launches' completion workgroups)
  CK(dmalloc(&h->d_flags, 4 + 16 * lii::kListCap));  // 2 counters (+ 2 pad), 2 x kListCap entries of two float4
  CK(hipMemset(h->d_flags, 0, sizeof(int) * (4 + 16 * lii::kListCap)));
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
  CK(dmalloc(&h->cal.d_cal_params, 64));
  CK(dmalloc(&h->cal.d_cal_out, 128));
  h->h_stage_elems = NM * kMatch;  // large enough for the neighbour download too
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_stage), sizeof(float4) * h->h_stage_elems, hipHostMallocDefault));
  CK(hipHostMalloc(reinterpret_cast<void**>(&h->h_small), sizeof(double) * 32768, hipHostMallocDefault));
  for (int i = 0; i < 4; i++) CK(hipEventCreateWithFlags(&h->prof.ev[i], kProfEventFlags));
  for (int i = 0; i < 32; i++) CK(hipEventCreateWithFlags(&h->prof.ev_it[i], kProfEventFlags));
  launch_table_clear(h->d_blocks, h->blocks_cap, h->stream);
  CK(hipStreamSynchronize(h->stream));
#undef CK
  *out = h;
  return LII_OK;
}

int lii_destroy(lii_handle h) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h) return LII_OK;
  (void)hipSetDevice(h->device);
  if (h->net.comm) ncclCommDestroy(h->net.comm);
  if (h->map_stream) (void)hipStreamSynchronize(h->map_stream);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
#ifdef LII_FALLBACK_TRACE
  {
    unsigned long long t[32] = {};
    lii::fb_trace_read(t);
    for (int kind = 0; kind < 2; kind++) {
      const unsigned long long* q = t + 8 * kind;
      if (q[0] == 0) continue;
      const double n = double(q[0]);
      std::fprintf(stderr, "[libliinit_hip completion trace] %s: %llu queries, us per query: head -> inner list %.2f, cell entries -> list %.2f, candidates %.2f, selection %.2f, "
                   "winners stored %.2f; chunks listed %.1f, cells of the cube %.0f\n", kind == 0 ? "balls of more than 256 cells" : "balls of up to 256 cells", q[0],
                   0.01 * q[1] / n, 0.01 * q[2] / n, 0.01 * q[3] / n, 0.01 * q[4] / n, 0.01 * q[5] / n, double(q[6]) / n, double(q[7]) / n);
    }
    if (t[0] > 0)
      std::fprintf(stderr, "[libliinit_hip completion trace] of 'cell entries -> list' (window rows only), us per query: until the block probes are in %.2f, waiting for the row loads %.2f, "
                   "list trips %.2f\n", 0.01 * t[24] / double(t[0]), 0.01 * t[25] / double(t[0]), 0.01 * t[26] / double(t[0]));
    if (t[16] > 0)
      std::fprintf(stderr, "[libliinit_hip completion trace] completion workgroups with work: %llu (%.2f queries each), us: head %.2f, completions %.2f, fit + sums %.2f\n", t[16],
                   double(t[20]) / double(t[16]), 0.01 * t[17] / double(t[16]), 0.01 * t[18] / double(t[16]), 0.01 * t[19] / double(t[16]));
  }
#endif
#ifdef LII_GAP_TRACE
  if (h->d_gran) {
    unsigned long long g[4] = {0, 0, 0, 0};
    if (hipMemcpy(g, h->d_gran + 200, sizeof(g), hipMemcpyDeviceToHost) == hipSuccess && g[2] > 0)
      std::fprintf(stderr, "[libliinit_hip gap trace] device idle between the stopping solve of a scan and the first kernel of the next: %.2f us mean over %llu "
                   "back-to-back scans (%llu gaps of 50 us or more left out)\n", 0.01 * double(g[1]) / double(g[2]), g[2], g[3]);
  }
#endif
  if (h->diag) std::fprintf(stderr, "[libliinit_hip] pre-armed prologues: %lld used, %lld cancelled, %lld expired\n", h->pre.n_used, h->pre.n_cancelled, h->pre.n_expired);
  if (h->diag) std::fprintf(stderr, "[libliinit_hip] dense cell window: kept current through %lld in-place updates, dropped %lld times\n", h->win_kept, h->win_dropped);
  if (h->diag) std::fprintf(stderr, "[libliinit_hip] map updates completed by a rebuild + re-insertion: %lld\n", h->map_recoveries);
  if (h->diag) std::fprintf(stderr, "[libliinit_hip] updates continued by the host after a parked loop: %lld\n", h->plan_parked);
  if (h->diag) std::fprintf(stderr, "[libliinit_hip] map updates repeated with exact list sizes: %lld\n", h->map_repeats);
  if (h->diag) std::fprintf(stderr, "[libliinit_hip] scans whose search passes left more than %d unfinished queries: %lld (the completion launch was switched on / off %lld times); fold tables cleared: %lld\n", lii::kFlagCap, h->wide_scans, h->wide_switches, h->ah_cleared);
  if (h->diag && h->prof.host_us[4] > 0)
    std::fprintf(stderr, "[libliinit_hip] host side of lii_scan_register, us per call over %.0f calls: first launch submitted %.1f, pre-processing enqueued %.1f, "
                 "loop enqueue %.1f, call %.1f, between calls %.1f\n", h->prof.host_us[4], h->prof.host_us[0] / h->prof.host_us[4], h->prof.host_us[1] / h->prof.host_us[4],
                 h->prof.host_us[2] / h->prof.host_us[4], h->prof.host_us[3] / h->prof.host_us[4], h->prof.host_us[5] / std::max(1.0, h->prof.host_us[4] - 1));
  if (h->diag && h->prof.host_us[4] > 0 && (h->prof.host_map_us[0] > 0 || h->prof.host_map_us[1] > 0))
    std::fprintf(stderr, "[libliinit_hip] map update, host us per scan: waited for the update in flight %.1f, enqueued behind the passes %.1f\n",
                 h->prof.host_map_us[0] / h->prof.host_us[4], h->prof.host_map_us[1] / h->prof.host_us[4]);
  for (auto& e : h->graphs) (void)hipGraphExecDestroy(e.second);
  h->graphs.clear();
  mailbox_close(&h->net.mailbox);
  if (h->net.d_mb_seq) (void)hipFree(h->net.d_mb_seq);
  if (h->net.d_gather_ticket) (void)hipFree(h->net.d_gather_ticket);
  if (h->net.d_gx) (void)hipFree(h->net.d_gx);
  if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
  for (hipEvent_t e : h->prof.kp_ev) (void)hipEventDestroy(e);
  if (h->ev_next) (void)hipEventDestroy(h->ev_next);
  if (h->ev_scan_free) (void)hipEventDestroy(h->ev_scan_free);
  if (h->h_stage_next) (void)hipHostFree(h->h_stage_next);
  if (h->d_scan_next) (void)hipFree(h->d_scan_next);
  void* dev[] = {h->d_dropped, h->d_pts, h->d_cell_cap, h->d_tp, h->d_cs_a, h->d_cs_b, h->d_work, h->d_ins_e, h->d_ins_e2, h->d_mapctr, h->d_map_unsorted, h->d_map, h->d_keys_a, h->d_keys_b, h->d_keys_c, h->d_idx_a, h->d_idx_b, h->d_blocks, h->d_cells, h->d_win, h->d_block_key,
                 h->d_counter, h->d_tomb, h->d_batch, h->d_ins, h->d_ins_c, h->d_u32_a, h->d_u32_b, h->d_u32_c, h->d_list_add, h->d_list_nodown, h->d_ah_key, h->d_ah_best, h->d_ah_slot, h->d_sort_temp, h->d_scan, h->d_body, h->d_world, h->d_nbr, h->d_nbr_count, h->d_plane,
                 h->d_selected, h->d_nbody, h->d_ctrl, h->d_pose, h->d_partials, h->d_out91, h->d_gran, h->d_extent, h->d_mm, h->d_bbox_rows, h->d_vkeys_a, h->d_vkeys_b,
                 h->d_vidx_b, h->d_vcomp, h->d_vsplit, h->d_vhist, h->d_vbucket, h->d_vpcl_in, h->d_vpcl_out, h->vh.slots, h->vh.slot_of, h->vh.next, h->vh.counts, h->vh.crowded, h->cal.d_cal_imu, h->cal.d_cal_lidar, h->cal.d_cal_params,
                 h->cal.d_cal_out, h->d_flags};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (h->ingest) ingest_destroy(h->ingest);
  if (h->h_stage) (void)hipHostFree(h->h_stage);
  if (h->h_small) (void)hipHostFree(h->h_small);
  if (h->h_ctrl) (void)hipHostFree(h->h_ctrl);
  if (h->h_res) (void)hipHostFree(h->h_res);
  if (h->pre.state) (void)hipHostFree(h->pre.state);
  if (h->pre.d_ring) (void)hipFree(h->pre.d_ring);
  if (h->pre.d_flag) (void)hipFree(h->pre.d_flag);
  if (h->ev_poses) (void)hipEventDestroy(h->ev_poses);
  if (h->ev_stage) (void)hipEventDestroy(h->ev_stage);
  if (h->ev_mapflag) (void)hipEventDestroy(h->ev_mapflag);
  if (h->ev_vh) (void)hipEventDestroy(h->ev_vh);
  if (h->h_vh_crowded) (void)hipHostFree(h->h_vh_crowded);
  if (h->h_mapflag) (void)hipHostFree(h->h_mapflag);
  if (h->ev_lists) (void)hipEventDestroy(h->ev_lists);
  if (h->n_map_pinned) (void)hipHostFree(h->n_map_pinned);
  for (int i = 0; i < 4; i++)
    if (h->prof.ev[i]) (void)hipEventDestroy(h->prof.ev[i]);
  for (int i = 0; i < 32; i++)
    if (h->prof.ev_it[i]) (void)hipEventDestroy(h->prof.ev_it[i]);
  if (h->map_stream) (void)hipStreamDestroy(h->map_stream);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return LII_OK;
}

int lii_synchronize(lii_handle h) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h) return LII_ERR_INVALID;
  {
    const int rc = map_join(h);
    if (rc != LII_OK) return rc;
  }
  // What is left on the stream when a caller of the per-scan loop synchronises are the launches behind the stopping pass (they read a
  // flag and return): microseconds.  hipStreamSynchronize spins briefly and then sleeps on an interrupt - ~60 us until the caller
  // runs again (bench.py's debug stamps, round 6) - so the stream is polled first, for up to ~0.2 ms, and the blocking wait only
  // takes over for work that really lasts.
  {
    const auto t0 = std::chrono::steady_clock::now();
    for (int polls = 1;; polls++) {
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) {
        if (h->diag) std::fprintf(stderr, "[libliinit_hip] lii_synchronize: stream empty after %d polls, %.1f us\n", polls,
                                  std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        return LII_OK;
      }
      if (q != hipErrorNotReady) return fail(h, LII_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
      if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
      __builtin_ia32_pause();
    }
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return LII_OK;
}

// ------------------------------------------------------------------------------------------------ scan
int lii_scan_upload(lii_handle h, const void* points, int32_t n, int32_t stride_bytes, int32_t time_offset_bytes) {
  if (h) h->scan_buf_idle = false;  // (work on the current scan buffer goes out)
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
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
  // (a pre-armed launch that waits for the scan in the OTHER buffer - the one lii_scan_advance made current - stays: this call only
  // puts an event and a wait between the two streams; one that waits for the buffer this call overwrites is told to end)
  if (h && h->pre.armed && !(h->pre.late && h->pre.scan_dev == h->d_scan)) lii_internal_prearm_cancel(h);
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
    // (asked once per source buffer: a loop that cycles through a few pinned buffers pays the runtime's look-up - microseconds - once each;
    // a buffer wrongly remembered as pinned is still copied correctly, the runtime stages it)
    bool known = false;
    for (int q = 0; q < 8; q++)
      if (h->pin_cache_ptr[q] == points) { direct = h->pin_cache_direct[q]; known = true; break; }
    if (!known) {
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, points) == hipSuccess) direct = attr.type == hipMemoryTypeHost;
      else (void)hipGetLastError();  // pageable memory: not an error
      h->pin_cache_ptr[h->pin_cache_at] = points; h->pin_cache_direct[h->pin_cache_at] = direct;
      h->pin_cache_at = (h->pin_cache_at + 1) & 7;
    }
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
  // (... unless it is known to have nothing left to do with it: the last update's result came back behind every launch that read the
  // buffer, and nothing has touched a scan buffer since - the per-scan loop's case, two runtime calls less between two registrations)
  if (!h->scan_buf_idle) {
    HIPCHK(h, hipEventRecord(h->ev_scan_free, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->copy_stream, h->ev_scan_free, 0));
  }
  if (n > 0) HIPCHK(h, hipMemcpyAsync(h->d_scan_next, src, sizeof(float4) * size_t(n), hipMemcpyHostToDevice, h->copy_stream));
  HIPCHK(h, hipEventRecord(h->ev_next, h->copy_stream));
  h->n_scan_next = n;
  return LII_OK;
}
int lii_scan_advance(lii_handle h) {
  // (a pre-armed launch that waits for exactly the scan this call makes current stays)
  if (h && h->pre.armed && !(h->pre.late && h->pre.scan_dev == h->d_scan_next && h->pre.n == h->n_scan_next)) lii_internal_prearm_cancel(h);
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
  if (h) h->scan_buf_idle = false;  // (work on the current scan buffer goes out)
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
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
  if (h) h->scan_buf_idle = false;  // (work on the current scan buffer goes out)
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !poses || n_poses < 1 || n_poses > 1024 || !end_R || !end_p || !R_LI || !T_LI)
    return fail(h, LII_ERR_INVALID, "lii_undistort_imu: bad arguments");
  { const int rcm = scan_materialize(h); if (rcm != LII_OK) return rcm; }
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
  h->vh_inserted = fuse_filter(h, h->fuse_leaf);
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
  if (h) h->scan_buf_idle = false;  // (work on the current scan buffer goes out)
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !omega || !vel || !end_R) return fail(h, LII_ERR_INVALID, "lii_undistort_cv: bad arguments");
  { const int rcm = scan_materialize(h); if (rcm != LII_OK) return rcm; }
  if (h->n_scan <= 0) return LII_OK;
  CvArgH a;
  std::memcpy(a.omega, omega, 24);
  std::memcpy(a.vel, vel, 24);
  std::memcpy(a.endR, end_R, 72);
  unsigned long long* ext = extent_of_scan(h);
  h->vh_inserted = fuse_filter(h, h->fuse_leaf);
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
  if (h) h->scan_buf_idle = false;  // (work on the current scan buffer goes out)
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h) return LII_ERR_INVALID;
  { const int rcm = scan_materialize(h); if (rcm != LII_OK) return rcm; }
  if (h->prof.kp_active) { const int r = kp_mark(h, LII_KP_VOXEL); if (r != LII_OK) return r; }
  if (h->n_scan > 0)
    HIPCHK(h, hipMemcpyAsync(h->d_body, h->d_scan, sizeof(float4) * size_t(h->n_scan), hipMemcpyDeviceToDevice, h->stream));
  h->n_body = h->n_scan;
  h->n_body_pending = false;
  h->have_search = false;
  h->body_reordered = false;
  h->body_partitioned = false;
  if (n_down) *n_down = h->n_body;
  return LII_OK;
}
int lii_downsample(lii_handle h, float leaf, int32_t* n_down, int32_t* filtered) {
  if (h) h->scan_buf_idle = false;  // (work on the current scan buffer goes out)
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !(leaf > 0)) return fail(h, LII_ERR_INVALID, "lii_downsample: bad arguments");
  { const int rcm = scan_materialize(h); if (rcm != LII_OK) return rcm; }
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
  if (h->prof.kp_active) { const int r = kp_mark(h, LII_KP_VOXEL); if (r != LII_OK) return r; }
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
  const bool by_voxel = inserted && h->vh.part_world > 1;  // this rank emits ITS voxels only
  if (use_hash) {
    if (++h->vh_epoch == 0u) h->vh_epoch = 1u;
    launch_voxel_hash(h->vh, h->d_scan, n, mm, h->d_bbox_rows, h->bbox_rows, leaf, h->d_body, h->d_nbody, h->d_nbody + 1, h->d_vpcl_out, hash_stages, h->vh_epoch, s, h->test_emit_late ? 1 : 0);
    if (!h->vh_pinned && !by_voxel && !h->vh_flag_pending && (++h->vh_watch & 15) == 0) {  // (every 16th scan: the copy costs a packet on the stream)
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
  h->n_body = by_voxel ? voxel_partition_bound(n, h->vh.part_world) : n;  // upper bound until resolved
  h->n_body_pending = true;
  h->body_partitioned = by_voxel;
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
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !n) return LII_ERR_INVALID;
  if (which == 0) { const int rcm = scan_materialize(h); if (rcm != LII_OK) return rcm; }
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

// ------------------------------------------------------------------------------------------------ utilities
int lii_dev_alloc(lii_handle h, size_t bytes, void** dev_ptr) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !dev_ptr) return LII_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMalloc(dev_ptr, bytes));
  return LII_OK;
}
int lii_dev_free(lii_handle h, void* dev_ptr) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h) return LII_ERR_INVALID;
  HIPCHK(h, hipFree(dev_ptr));
  return LII_OK;
}
int lii_dev_upload(lii_handle h, void* dev_dst, const void* host_src, size_t bytes) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !dev_dst || !host_src) return LII_ERR_INVALID;
  HIPCHK(h, hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
  return LII_OK;
}
int lii_set_profiling(lii_handle h, int32_t enabled) {
  if (!h) return LII_ERR_INVALID;
  h->prof.profiling = enabled != 0;
  h->prof.prof_mode = enabled;
  if (enabled == 1) {
    h->prof.ev_it_due = 0u;  // (a new accumulation: what has not been read belongs to the old one)
    for (double& t : h->prof.timings) t = 0;  // 1: (re)start the accumulation; 2: resume; 0: pause (accumulators kept)
    h->prof.kprof = lii_kernel_profile{};
  }
  return LII_OK;
}
int lii_last_kernel_profile(lii_handle h, lii_kernel_profile* out) {
  if (!h || !out || out->struct_size != sizeof(lii_kernel_profile)) return fail(h, LII_ERR_INVALID, "lii_last_kernel_profile: bad arguments");
  *out = h->prof.kprof;
  out->struct_size = sizeof(lii_kernel_profile);
  return LII_OK;
}
int lii_last_timings(lii_handle h, double out_ms[8]) {
  if (!h || !out_ms) return LII_ERR_INVALID;
  harvest_knn_events(h);
  std::memcpy(out_ms, h->prof.timings, sizeof(h->prof.timings));
  return LII_OK;
}

}  // extern "C"
