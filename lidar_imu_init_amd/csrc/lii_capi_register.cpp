// libliinit_hip — the registration loop (host side): lii_iekf_iterate / lii_iekf_update / lii_scan_register and the neighbour
// download.  Kernels: lii_kernels.hip (k-NN, fit + reduce), lii_iekf.hip (final sum + 24-state solve), lii_scan.hip (prologue of
// lii_scan_register).  Reference: src/laserMapping.cpp:909-1134.
#include "lii_context.h"
#include "lii_hostmath.h"

using namespace lii_impl;

namespace {


// (every enqueued search launch has a number, which the fit launch behind it shares: the list of unfinished queries the one leaves
// and the other consumes - RegistrationBuffers::flag_*; 0 = no list: launches captured into a hipGraph, whose arguments are frozen)
int next_knn_epoch(lii_handle h, bool listed) {
  if (!listed) return 0;
  h->knn_epoch = h->knn_epoch >= 0x3FFFFFFE ? 1 : h->knn_epoch + 1;  // (consecutive numbers alternate between the two slots, across the wrap as well)
  return h->knn_epoch;
}
void launch_knn(lii_handle h, const GridView& g, const RegistrationBuffers& rb, const PoseArg* pose, int forced, int epoch,
                hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr) {
  lii::launch_knn(g, rb, pose, h->d_ctrl, forced, h->d_ctrl->search_pose, h->stream, epoch, ev_start, ev_stop);
}


int iterate(lii_handle h, const lii_state* st, bool search, bool imu_en, double* out91) {
  if (h->n_body <= 0) return fail(h, LII_ERR_STATE, "no down-sampled scan (call lii_downsample / lii_downsample_skip)");
  int rc = commit_map(h);
  if (rc != LII_OK) return rc;
  if (!search && !h->have_search) return fail(h, LII_ERR_STATE, "non-search iteration before any search");
  GridView g = grid_view(h);
  RegistrationBuffers rb = reg_buffers(h);
  const bool prof = h->prof.profiling;
  if (prof) HIPCHK(h, hipEventRecord(h->prof.ev[0], h->stream));
  // the pose of a host-driven pass travels to the handle's pose slot first: the kernels read it from device memory on every path
  {
    PoseArg* stage = reinterpret_cast<PoseArg*>(h->h_small + 20000);  // (pinned, a stretch nothing else uses; this call ends with a synchronisation)
    *stage = pose_of(*st);
    HIPCHK(h, hipMemcpyAsync(h->d_pose, stage, sizeof(PoseArg), hipMemcpyHostToDevice, h->stream));
  }
  if (search) h->unfinished_known = false;  // (a search pass of this host-driven call: lii_last_unfinished_queries reads the device again)
  const int epoch = search ? next_knn_epoch(h, true) : 0;
  if (search) launch_knn(h, g, rb, h->d_pose, 1, epoch);
  if (prof) HIPCHK(h, hipEventRecord(h->prof.ev[3], h->stream));
  launch_fit_reduce(g, rb, h->d_pose, h->d_ctrl, search ? 1 : 0, imu_en ? 1 : 0, h->cfg.plane_threshold,
                    h->cfg.laser_point_cov_inv, h->stream, epoch);
  if (prof) HIPCHK(h, hipEventRecord(h->prof.ev[1], h->stream));
  launch_reduce91(rb, h->d_out91, h->d_ctrl, 1, h->stream, epoch);
  if (prof) HIPCHK(h, hipEventRecord(h->prof.ev[2], h->stream));
  if (search) h->have_search = true;
  if (h->net.comm) {
    ncclResult_t r = ncclAllReduce(h->d_out91, h->d_out91, kNormalEq, ncclDouble, ncclSum, h->net.comm, h->stream);
    if (r != ncclSuccess) return fail(h, LII_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
  } else if (h->net.mailbox.dev_slots) {
    launch_mailbox_allreduce(h->d_out91, mailbox_view(h), h->stream);
  }
  HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_out91, sizeof(double) * kNormalEq, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::memcpy(out91, h->h_small, sizeof(double) * kNormalEq);
  if (h->net.mailbox.dev_slots && out91[kNormalEq - 1] != out91[kNormalEq - 1])
    return fail(h, LII_ERR_COMM, "mailbox exchange timed out (a rank of the job did not reach this pass)");
  if (prof) {
    // timings: [0] sum ms search-pass kernel, [1] sum ms residual-pass kernel, [2] sum ms reduce kernel,
    //          [3] host solve ms (last update), [4] total ms (last update), [5]/[6] launch counts of [0]/[1]
    float a = 0, b = 0;
    HIPCHK(h, hipEventElapsedTime(&a, h->prof.ev[0], h->prof.ev[1]));
    HIPCHK(h, hipEventElapsedTime(&b, h->prof.ev[1], h->prof.ev[2]));
    if (search) {
      float k = 0;
      HIPCHK(h, hipEventElapsedTime(&k, h->prof.ev[0], h->prof.ev[3]));
      h->prof.timings[0] += a; h->prof.timings[5] += 1; h->prof.timings[7] += k;  // [7]: the k-NN kernel alone
    } else { h->prof.timings[1] += a; h->prof.timings[6] += 1; }
    h->prof.timings[2] += b;
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
  // (the mask holds 16 passes: a longer loop is enqueued whole - the device takes every pass from 16 on for enqueued, which a
  // plan that ends earlier would not honour - ADVICE r4)
  unsigned int plan = 0xFFFFFFFFu;
  if (h->knn_plan && !h->net.comm && opts->max_iterations <= 16) {
    plan = (h->knn_plan_force >= 0 ? ((unsigned int)h->knn_plan_force | 0xFFFF0000u) : h->plan_next) | 0x00010001u;
  }
  hc->plan_mask = plan;
  h->plan_cur = plan;
}

int update_on_device(lii_handle h, lii_state* state, const lii_state* state_prop, const lii_iekf_opts* opts,
                     lii_iekf_report* report) {
  static_assert(sizeof(lii_state) == sizeof(double) * kStateDoubles, "lii_state layout");
  if (h->n_body <= 0) return fail(h, LII_ERR_STATE, "no down-sampled scan (call lii_downsample / lii_downsample_skip)");
  const auto t_map0 = std::chrono::steady_clock::now();
  int rc = commit_map(h);
  if (rc != LII_OK) return rc;
  if (h->diag) h->prof.host_map_us[0] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_map0).count();
  hipStream_t s = h->stream;
  // The stream lii_map_incremental leaves its update on is created with the first update of a handle that is NOT a rank of a
  // sharded job (those never update beside a scan).  Not in lii_create: a second compute queue per process - even one whose
  // stream has been destroyed again: the runtime keeps the hardware queue - makes several processes on one device oversubscribe
  // the hardware queues, and a kernel that waits for a peer's kernel (the mailbox) then waits for a time slice: the one-device
  // rehearsal of a 2-rank job fell from 4 000 to 1 350 scans/s.
  if (!h->map_stream && h->net.n_ranks <= 1) HIPCHK(h, hipStreamCreateWithFlags(&h->map_stream, hipStreamNonBlocking));
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
  const bool prof = h->prof.profiling && h->prof.prof_mode != 3;  // (mode 3 brackets every launch itself: kp_mark)
  if (prof) harvest_knn_events(h);  // (the events are about to be used again)
  const double* ne = h->net.comm ? h->d_out91 + 128 : h->d_out91;
  unsigned int plan = h->plan_cur;  // fill_ctrl chose it (the control block on the device carries the same mask)
  const unsigned int plan0 = plan;
  // (profiling = HIP events around the k-NN launches only - the dominant kernel, lii_last_timings [5] / [7]; every event is a
  // barrier packet on the stream, so the rest of the loop is left alone: launch plan and result polling work as always)
  const bool graph_mode = h->use_graph && !h->net.comm && !h->prof.profiling;
  auto enqueue_pass = [&](int it) -> int {
    const bool knn = it >= 16 || ((plan >> it) & 1u);
    // (a fit launch that is not behind a search launch never runs as a search pass - the plan parks the loop instead - and needs no number)
    const int epoch = knn ? next_knn_epoch(h, !graph_mode) : 0;
    if (knn) {
      // (the k-NN launches of a profiled update carry their two events IN the dispatch: the kernel's own start and end stamps.  Rounds
      // 1 - 5 recorded an event in front of and behind the launch - two barrier packets each, + 22 us on a profiled scan, and the
      // bracket held the dispatch as well as the kernel; LII_PROF_BRACKET=1 keeps that form for comparison)
      const bool in_dispatch = prof && it < 16 && !h->prof.bracket_events;
      if (prof && it < 16 && !in_dispatch) HIPCHK(h, hipEventRecord(h->prof.ev_it[2 * it], s));
      if (h->prof.kp_active) { const int r = kp_mark(h, LII_KP_KNN, it); if (r != LII_OK) return r; }
      if (in_dispatch) launch_knn(h, g, rb, pose, -1, epoch, h->prof.ev_it[2 * it], h->prof.ev_it[2 * it + 1]);
      else launch_knn(h, g, rb, pose, -1, epoch);
      if (prof && it < 16 && !in_dispatch) HIPCHK(h, hipEventRecord(h->prof.ev_it[2 * it + 1], s));
    }
    if (h->prof.kp_active) { const int r = kp_mark(h, LII_KP_FIT, it); if (r != LII_OK) return r; }
    // the scan before left more unfinished queries per search pass than the fit launch's completion workgroups take: this one's
    // are finished by a launch of their own, one wavefront per listed query (k_complete_listed; nothing to do -> it returns at once)
    if (knn && h->wide_listed && !graph_mode && !h->net.comm && h->net.n_ranks <= 1) launch_complete_listed(g, rb, h->d_ctrl, -1, s, epoch);
    launch_fit_reduce(g, rb, pose, h->d_ctrl, -1, opts->imu_en ? 1 : 0, h->cfg.plane_threshold, h->cfg.laser_point_cov_inv, s, epoch);
    if (h->prof.kp_active) { const int r = kp_mark(h, LII_KP_SOLVE, it); if (r != LII_OK) return r; }
    if (!h->net.comm) {  // single GPU or node-local mailbox: final sum (+ exchange) and solve in one launch
      launch_reduce_solve(rb, h->d_gran, h->d_ctrl, h->h_res, mailbox_view(h), s, epoch);
      return LII_OK;
    }
    launch_reduce91(rb, h->d_out91, h->d_ctrl, -1, s, epoch);
    // every rank enqueues the same number of all-reduces; a pass that is skipped on the device re-sums the
    // unchanged local buffer on all ranks alike, so the ranks stay in lock-step without a host decision
    ncclResult_t r = ncclAllReduce(h->d_out91, h->d_out91 + 128, kNormalEq, ncclDouble, ncclSum, h->net.comm, s);
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
  if (graph_mode) {
    // The same launches, captured once and replayed (hipGraphLaunch): every kernel argument of the loop is a device pointer or
    // a constant of the configuration, except the bound of the cloud size (rounded up here: the kernels take the exact size
    // from the device), the plan and the view of the map - the key of the cache.  Measured against the plain launches in
    // profiles/r03_hipgraph_ab.md.
    if (rb.n_dev) rb.n = std::min(rb.cap, (rb.n + 4095) & ~4095);
    struct { const void* p[4]; unsigned int mask; int n_pts, n, plan, max_it, imu_en, shard; float cs; } kv;
    std::memset(&kv, 0, sizeof(kv));
    kv.p[0] = g.pts; kv.p[1] = g.blocks; kv.p[2] = g.cells; kv.p[3] = rb.n_dev;
    kv.mask = g.block_mask; kv.n_pts = g.n_pts; kv.n = rb.n; kv.plan = (int)plan; kv.max_it = opts->max_iterations;
    kv.imu_en = opts->imu_en ? 1 : 0;
    kv.shard = rb.shard_world * 4096 + rb.shard_rank; kv.cs = g.cs;
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
  // pattern changed against the previous scans) is continued from here (below): one host round trip, on those scans.
  auto wait_result = [&](bool first) -> int {
    const int parked_word = first ? (h->update_seq | kLoopParked) : h->update_seq;
    if (h->poll_result && !h->net.comm) {
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
  if (h->diag) h->prof.host_loop_enq_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_loop0).count();
  if (h->prof.kp_active) { rc = kp_mark(h, LII_KP_KINDS); if (rc != LII_OK) return rc; }  // (end mark of the planned passes)
  h->map_enqueued_early = false;
  if (h->map_after_update && !h->net.comm && !h->prof.profiling) {
    // lii_scan_job::map_update: the map update goes out now, behind the passes, while the device still works on them
    const auto t_me0 = std::chrono::steady_clock::now();
    // (a batch that cannot be enqueued now - no room for the predicted sizes - is not this update's failure: the map update is made
    // when the update has ended, by the call that reports such things)
    h->map_enqueued_early = map_update_early(h) == 1;
    // (map_apply rebuilds the index when the map is short of room - build_index may free and reallocate the block table, the cell
    // tables and their side arrays and change the mask: a loop that parks is continued below with launches that must see the map
    // as it is NOW, not the view taken before the passes went out - ADVICE r4)
    g = grid_view(h);
    if (h->diag) h->prof.host_map_us[1] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_me0).count();
  }
  // THE NEXT SCAN'S PROLOGUE, pre-armed (lii_launch.h: DeskewGate): the job announced the scan the next call will bring - its de-skew +
  // filter-insert launch goes out now, behind this update's passes (and its map update), and waits on the device for the record
  // the next lii_scan_register writes.
  if (h->pre.want_dev && !h->net.comm && h->net.n_ranks <= 1 && h->prof.prof_mode != 3 && !graph_mode && !h->map_async && h->poll_result) {  // (poll_result: a caller that ends its updates with a stream synchronise - LII_TEST=sync_result - would wait for the waiting launch)  // (mode 3 puts an event in front of every launch: not in front of one that waits)
    lii::GateState* gst = h->pre.state;
    h->pre.seq = (h->pre.seq + 1) & 0x003FFFFFFFFFFFFFull;  // (seq << 2 stays below the tag's top byte)
    __atomic_store_n(&gst->word, (h->pre.seq << 2) | lii::kGateArmed, __ATOMIC_RELEASE);
    h->pre.scan_dev = h->pre.want_dev; h->pre.n = h->pre.want_n; h->pre.leaf = h->pre.want_leaf; h->pre.late = h->pre.want_late;
    h->pre.fuse = fuse_filter(h, h->pre.leaf);
    DeskewPlan dp = {};
    dp.in = static_cast<const float4*>(h->pre.scan_dev);
    dp.out = h->pre.late ? const_cast<float4*>(dp.in) : h->d_scan;  // (late: the buffer becomes the handle's scan buffer with lii_scan_advance)
    dp.n = h->pre.n; dp.sorted = 1; dp.extent = nullptr; dp.bbox_rows = h->d_bbox_rows;
    dp.leaf = h->pre.leaf; dp.vh = h->pre.fuse ? &h->vh : nullptr;
    dp.ctrl_src = h->h_ctrl; dp.ctrl_dst = h->d_ctrl; dp.ctrl_bytes = (sizeof(IekfCtrl) + 15) / 16 * 16;
#ifdef LII_GAP_TRACE
    dp.gap = h->d_gran;
#endif
    lii::DeskewGate gate = {gst, h->pre.d_ring + size_t(h->pre.seq % lii::kGateRing) * lii::kGateLines * 8, h->pre.d_flag, h->pre.seq, h->pre.timeout_ticks, h->pre.late ? 1 : 0};
    launch_deskew_imu_gated(dp, gate, s);
    h->pre.armed = hipGetLastError() == hipSuccess;
    if (!h->pre.armed) __atomic_store_n(&gst->word, (h->pre.seq << 2) | lii::kGateCancel, __ATOMIC_RELEASE);
  }
  h->pre.want_dev = nullptr;
  // lii_scan_job::while_waiting (ABI 9): everything is enqueued, the thread would only wait now - the caller's hook runs first
  if (h->wait_hook) {
    void (*hook)(void*) = h->wait_hook;
    h->wait_hook = nullptr;
    h->in_wait_hook = true;
    hook(h->wait_hook_arg);
    h->in_wait_hook = false;
  }
  rc = wait_result(true);
  if (rc != LII_OK) return rc;
  // A parked loop is continued with what it is known to need: the pass it parked in front of - behind a k-NN launch if that pass
  // searches - and as many passes behind it as the longer of the last two updates ran; should that not be enough (the next search
  // comes at another pass as well, or the update runs longer) it parks again and is continued again.  (Round 3 enqueued every
  // remaining pass with every k-NN launch: up to nine launches that only read a flag, ~ 20 us on a scan that parks.)
  for (int round = 0; h->h_res->done == (h->update_seq | kLoopParked); round++) {
    if (round > 2 * opts->max_iterations) return fail(h, LII_ERR_HIP, "device loop parked again and again");
    const int from = h->h_res->parked_it;
    const bool search = h->h_res->parked_search != 0;
    h->plan_parked++;
    prearm_cancel(h);  // (the launches that continue the loop would queue up behind a launch that waits for this call to return)
    h->map_enqueued_early = false;  // (that launch saw a parked loop and did nothing: the caller makes the map update when the loop has ended)
    g = grid_view(h);  // (see above: never a view older than the last thing that may have rebuilt the index)
    const int last = std::min(opts->max_iterations, std::max(h->plan_passes_prev, from + 1));  // (exclusive; from < max_iterations: the last pass never parks)
    unsigned int rplan = 0u;
    for (int q = from; q < last && q < 16; q++) rplan |= 1u << (16 + q);
    if (search && from < 16) rplan |= 1u << from;
    plan = rplan;
    h->h_res->done = 0;  // (the parked word of this round must not be taken for the next one's)
    std::atomic_thread_fence(std::memory_order_release);
    launch_loop_resume(h->d_ctrl, rplan, s);
    for (int it = from; it < last; it++) {
      rc = enqueue_pass(it);
      if (rc != LII_OK) return rc;
    }
    if (h->prof.kp_active) { rc = kp_mark(h, LII_KP_KINDS); if (rc != LII_OK) return rc; }
    rc = wait_result(true);
    if (rc != LII_OK) return rc;
  }
  h->unfinished_known = !h->net.comm;
  if (!h->net.comm && h->wide_enabled) {  // (the launch plan of the completions, see enqueue_pass)
    // (two scans in a row decide: a stream on which one scan in eight crosses the capacity - bench --edge - would pay for a launch that
    // finds nothing behind every such scan and never have it where it is needed)
    const bool wide = h->h_res->unfinished > lii::kFlagCap;
    if (wide == h->wide_prev && wide != h->wide_listed) { h->wide_listed = wide; h->wide_switches++; }
    h->wide_prev = wide;
    if (wide) h->wide_scans++;
  }
  h->staging_busy = false;  // the wait above covers everything enqueued before the stopping pass
  h->scan_buf_idle = true;  // ... every launch that read or wrote the current scan buffer among it (what was enqueued behind - drained passes, the map update, a pre-armed launch on the OTHER buffer - does not touch it)
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
  if (h->prof.kp_active && h->prof.kp_n > 1 && hr->it > 0) {
    // per-launch brackets (lii_set_profiling(h, 3)): the time from the event in front of a launch to the next event, for the
    // launches that executed (a pass the loop did not reach, or a k-NN launch whose pass did not search, only read a flag)
    HIPCHK(h, hipEventSynchronize(h->prof.kp_ev[size_t(h->prof.kp_n - 1)]));
    for (int i = 0; i + 1 < h->prof.kp_n; i++) {
      int kind = h->prof.kp_kind[size_t(i)] / 64;
      const int it = h->prof.kp_kind[size_t(i)] % 64;
      if (kind >= LII_KP_KINDS) continue;
      const bool loop_kind = kind == LII_KP_KNN || kind == LII_KP_FIT || kind == LII_KP_SOLVE;
      if (loop_kind && (it >= hr->it || it >= 16)) continue;
      if (kind == LII_KP_KNN && !(hr->search_log[it] & 1)) continue;
      if (kind == LII_KP_FIT && (hr->search_log[it] & 1)) kind = LII_KP_FIT_SEARCH;
      float ms = 0;
      if (hipEventElapsedTime(&ms, h->prof.kp_ev[size_t(i)], h->prof.kp_ev[size_t(i + 1)]) != hipSuccess) continue;
      h->prof.kprof.ms[kind] += ms;
      h->prof.kprof.launches[kind] += 1;
    }
    h->prof.kprof.scans += 1;
  }
  if (hr->part_overflow) {
    h->h_res->part_overflow = 0;
    return fail(h, LII_ERR_CAPACITY, "voxel-partitioned job: this rank's share of the down-sampled cloud exceeds its bound (mean share + 25 % + 2048 points); use lii_comm_set_partition(h, 1)");
  }
  if (hr->singular == 3)
    return fail(h, LII_ERR_COMM, "mailbox exchange timed out (a rank of the job did not reach this pass); re-create the communicator");
  if (hr->singular == 4)
    return fail(h, LII_ERR_COMM, "the sums of a pass did not reach the solver within 2 s (a summing workgroup of the launch was not scheduled: is the device shared?)");
  if (hr->singular) return fail(h, LII_ERR_INVALID, "singular covariance / normal matrix in the device solve");
  if (hr->it < 0) return fail(h, LII_ERR_HIP, "device loop ended without a result");
  std::memcpy(state, hr->st, sizeof(lii_state));
  h->last_pivoted_passes = 0;
  for (int q = 0; q < 16 && q < hr->it; q++) h->last_pivoted_passes += (hr->search_log[q] >> 1) & 1;
  {  // the next update's plan: this one's pattern; passes it did not reach keep their launch
    unsigned int next = 0xFFFFFFFFu;
    for (int q = 0; q < 16 && q < hr->it; q++)
      if (!(hr->search_log[q] & 1)) next &= ~(1u << q);
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
    // the k-NN kernel alone, over EVERY pass that actually searched (the device logs which iterations did).  The events are READ
    // LATER (harvest_knn_events: in front of the next profiled update, or when the timings are asked for): asking the runtime for
    // an event's time right behind the launch it belongs to makes it wait for its completion handler - ~10 us per pair on the
    // host's path to the next scan (round 6: a profiled step cost + 22 us against its neighbours).
    unsigned int due = 0u;
    for (int it = 0; it < opts->max_iterations && it < 16; it++) {
      if (it >= hr->it || !(hr->search_log[it] & 1) || !((plan0 >> it) & 1u)) continue;
      due |= 1u << it;
    }
    h->prof.ev_it_due = due;
  }
  return LII_OK;
}


}  // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------ registration
int lii_last_solve_info(lii_handle h, int32_t* pivoted_passes) {
  if (!h || !pivoted_passes) return LII_ERR_INVALID;
  *pivoted_passes = h->last_pivoted_passes;
  return LII_OK;
}

int lii_last_unfinished_queries(lii_handle h, int32_t* n_last) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !n_last) return LII_ERR_INVALID;
  if (h->unfinished_known) { *n_last = h->h_res->unfinished; return LII_OK; }  // (came with the last update's result: the largest count among its search passes)
  int c[2] = {0, 0};
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(c, h->d_flags, sizeof(c), hipMemcpyDeviceToHost));  // RegistrationBuffers::flag_count, one word per slot
  *n_last = c[h->knn_epoch & 1];  // (consecutive launch numbers alternate between the two slots; a launch clears the other one)
  return LII_OK;
}

int lii_iekf_iterate(lii_handle h, const lii_state* state, int32_t search, int32_t imu_en, double out91[91]) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !state || !out91) return fail(h, LII_ERR_INVALID, "lii_iekf_iterate: bad arguments");
  return iterate(h, state, search != 0, imu_en != 0, out91);
}

int lii_iekf_update(lii_handle h, lii_state* state, const lii_state* state_prop, const lii_iekf_opts* opts,
                    lii_iekf_report* report) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !state || !state_prop || !opts || opts->max_iterations < 1) return fail(h, LII_ERR_INVALID, "lii_iekf_update: bad arguments");
  const int max_it = opts->max_iterations;
  auto t_begin = std::chrono::steady_clock::now();
  if (!h->host_solve) {
    int rc = update_on_device(h, state, state_prop, opts, report);
    h->prof.timings[4] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
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
  h->prof.timings[3] = host_ms;
  h->prof.timings[4] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return LII_OK;
}

int lii_scan_register(lii_handle h, const lii_scan_job* job, lii_state* state, const lii_state* state_prop,
                      lii_iekf_report* report) {
  // (struct_size 48: a job of ABI 5, without scan_sorted)
  // (struct_size 88: ABI 9; 72: ABI 8, without while_waiting; 56: ABI 6 - 7, without next_scan_dev; 48: ABI 5, without scan_sorted)
  static_assert(sizeof(lii_scan_job) == 88, "lii_scan_job: the sizes of the earlier ABIs are accepted by number");
  if (!h || !job || (job->struct_size != sizeof(lii_scan_job) && job->struct_size != 72u && job->struct_size != 56u && job->struct_size != 48u) || !state || !state_prop || job->opts.max_iterations < 1)
    return fail(h, LII_ERR_INVALID, "lii_scan_register: bad arguments");
  const bool sorted = job->struct_size >= 56u && job->scan_sorted == 1;
  h->scan_buf_idle = false;
  int rc = LII_OK;
  const auto t_entry = std::chrono::steady_clock::now();
  if (h->diag && h->prof.host_us[4] > 0) h->prof.host_us[5] += std::chrono::duration<double, std::micro>(t_entry - h->prof.host_last_return).count();
  // the scan to adopt: the caller's device buffer (lii_scan_job::scan_dev), or the frame lii_frame_select left where the ingest put it
  const bool from_job = job->scan_dev != nullptr && job->n_scan_dev > 0;
  const void* const src_dev = from_job ? job->scan_dev : static_cast<const void*>(h->scan_pending);
  const int src_n = from_job ? job->n_scan_dev : h->scan_pending_n;
  const bool adopt = src_dev != nullptr && src_n > 0;
  if (adopt && src_n > h->cfg.max_scan_points) return fail(h, LII_ERR_CAPACITY, "lii_scan_register: n_scan_dev > max_scan_points");
  const int n_next = adopt ? src_n : h->n_scan;
  // A gated de-skew launch waits on the stream (the previous call enqueued it for the scan its job announced): it is used when THIS
  // call asks for exactly that, and told to end otherwise - before anything here could wait for the stream.
  const float leaf_now = job->leaf > 0 ? job->leaf : 0.f;
  // (the announced scan: the job's own device buffer, or - announced while lii_scan_upload_next was bringing it - the handle's current
  // scan after lii_scan_advance, de-skewed in place)
  const void* const cur_dev = from_job ? job->scan_dev : (adopt ? nullptr : static_cast<const void*>(h->d_scan));
  const int cur_n = from_job ? job->n_scan_dev : h->n_scan;
  bool use_pre = h->pre.armed && sorted && job->undistort == 1 && job->imu_poses && job->n_imu_poses >= 2 && job->n_imu_poses <= lii::kGateMaxPoses &&
                 cur_dev != nullptr && cur_dev == h->pre.scan_dev && cur_n == h->pre.n && h->pre.late == !from_job && leaf_now == h->pre.leaf && !h->host_solve && !h->no_fast_prologue &&
                 h->prof.prof_mode != 3 && !h->staging_busy && fuse_filter(h, leaf_now) == h->pre.fuse;
  if (!use_pre) prearm_cancel(h);
  // ... and what this job announces for the next call (update_on_device arms it behind the passes)
  h->pre.want_dev = nullptr;
  if (job->struct_size >= 72u && job->next_scan_dev && job->next_n_scan > 0 && job->next_n_scan <= h->cfg.max_scan_points && h->pre.enabled &&
      sorted && job->undistort == 1) {
    h->pre.want_dev = job->next_scan_dev; h->pre.want_n = job->next_n_scan; h->pre.want_leaf = leaf_now; h->pre.want_late = false;
  } else if (h->n_scan_next > 0 && h->d_scan_next && h->pre.enabled && sorted && job->undistort == 1 && !from_job) {
    // a scan is on its way through lii_scan_upload_next: it is the next call's (after lii_scan_advance), de-skewed where it lands
    h->pre.want_dev = h->d_scan_next; h->pre.want_n = h->n_scan_next; h->pre.want_leaf = leaf_now; h->pre.want_late = true;
  }
  // A scan in ascending time order: ONE launch takes it from wherever it arrived (the caller's device buffer is read in place)
  // to the de-skewed scan with the voxel filter's table filled, and one extra workgroup of it pulls the update's control block
  // over PCIe; the IMU pose table (<= 64 poses) travels in the kernel arguments.  Round 3 needed k_time_extent in front (copy +
  // time extent + pull: 8.9 us per scan).
  h->prof.kp_active = h->prof.prof_mode == 3 && !h->host_solve;
  h->prof.kp_n = 0;
  if (h->prof.kp_active) { rc = kp_mark(h, LII_KP_DESKEW); if (rc != LII_OK) { h->prof.kp_active = false; return rc; } }
  const bool fast = sorted && !h->host_solve && n_next > 0 && !h->no_fast_prologue &&
                    ((job->undistort == 1 && job->imu_poses && job->n_imu_poses >= 2 && job->n_imu_poses <= 64) || job->undistort == 2);
  if (use_pre && !fast) { prearm_cancel(h); use_pre = false; }  // (cannot happen with the conditions above; a waiting launch must never be left behind a call that will not feed it)
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
    h->vh_inserted = fuse_filter(h, fuse_leaf);
    if (h->vh_inserted) h->vh_inserted_leaf = fuse_leaf;
    if (use_pre) {
      // GO - unless the launch gave up in this very moment (then the scan is launched below like any other) - and the record: straight
      // into device memory (large BAR), payload, then the tags, line 0's last, a store fence between the three
      h->pre.armed = false;
      if (gate_move(h->pre.state, h->pre.seq, lii::kGateGo)) {
        h->pre.n_used++;
        const int K = job->n_imu_poses;
        const int n_pay = 25 + 22 * K, n_lines = (n_pay + 6) / 7;
        double pay[7 * lii::kGateLines];
        pay[0] = double(K);
        std::memcpy(pay + 1, state->rot_end, 72);
        std::memcpy(pay + 10, state->pos_end, 24);
        std::memcpy(pay + 13, state->offset_R_L_I, 72);
        std::memcpy(pay + 22, state->offset_T_L_I, 24);
        std::memcpy(pay + 25, job->imu_poses, sizeof(lii_pose6d) * size_t(K));
        for (int e = n_pay; e < 7 * n_lines; e++) pay[e] = 0.0;
        volatile double* rec = h->pre.d_ring + size_t(h->pre.seq % lii::kGateRing) * lii::kGateLines * 8;
        double tag;
        const unsigned long long tagw = (h->pre.seq << 2) | lii::kGateGo;
        std::memcpy(&tag, &tagw, 8);
        for (int l = 0; l < n_lines; l++)
          for (int c = 0; c < 7; c++) rec[8 * l + c] = pay[7 * l + c];
        __builtin_ia32_sfence();
        for (int l = 1; l < n_lines; l++) rec[8 * l + 7] = tag;
        __builtin_ia32_sfence();
        {  // (line 0's tag carries K in its top byte: lii_scan.hip, k_deskew_imu_gated)
          const unsigned long long tag0 = tagw | ((unsigned long long)K << 56);
          double t0d;
          std::memcpy(&t0d, &tag0, 8);
          rec[7] = t0d;
        }
        __builtin_ia32_sfence();
      } else {
        use_pre = false;
        h->pre.n_expired++;
      }
    }
    DeskewPlan dp = {};
    dp.in = adopt ? static_cast<const float4*>(src_dev) : h->d_scan;
    dp.out = h->d_scan; dp.n = n_next; dp.sorted = 1; dp.extent = nullptr; dp.bbox_rows = h->d_bbox_rows;
    dp.leaf = fuse_leaf; dp.vh = h->vh_inserted ? &h->vh : nullptr;
    dp.ctrl_src = h->h_ctrl; dp.ctrl_dst = h->d_ctrl; dp.ctrl_bytes = (sizeof(IekfCtrl) + 15) / 16 * 16;
#ifdef LII_GAP_TRACE
    dp.gap = h->d_gran;
#endif
    if (use_pre) {
      // (the launch is on the stream already and has its go)
    } else if (job->undistort == 1) {
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
    rc = lii_scan_set_device(h, src_dev, src_n);
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
  const bool map_update = job->struct_size >= 56u && job->map_update == 1;
  h->map_after_update = map_update && !h->host_solve;
  h->map_enqueued_early = false;
  // lii_scan_job::while_waiting (ABI 9): update_on_device calls it when its launches are out, before it waits for the result; an
  // arrangement without that wait (LII_TEST=host_solve) calls it behind the update - once per call that got this far, never otherwise
  h->wait_hook = (rc == LII_OK && job->struct_size >= 88u) ? job->while_waiting : nullptr;
  h->wait_hook_arg = job->struct_size >= 88u ? job->while_waiting_arg : nullptr;
  if (rc == LII_OK) rc = lii_iekf_update(h, state, state_prop, &job->opts, report);
  if (h->wait_hook) {
    void (*hook)(void*) = h->wait_hook;
    h->wait_hook = nullptr;
    hook(h->wait_hook_arg);
  }
  h->map_after_update = false;
  // map_incremental behind the update: already enqueued behind its passes (update_on_device), or made now
  if (rc == LII_OK && map_update && !h->map_enqueued_early) rc = lii_map_incremental(h, state, nullptr, nullptr);
  h->map_enqueued_early = false;
  h->poses_preloaded = h->ctrl_preloaded = false;  // also on the error paths
  h->prof.kp_active = false;
  if (h->diag) {
    const auto t_end = std::chrono::steady_clock::now();
    h->prof.host_us[0] += std::chrono::duration<double, std::micro>(t_first - t_entry).count();
    h->prof.host_us[1] += std::chrono::duration<double, std::micro>(t_pre - t_entry).count();
    h->prof.host_us[2] += h->prof.host_loop_enq_us;
    h->prof.host_us[3] += std::chrono::duration<double, std::micro>(t_end - t_entry).count();
    h->prof.host_us[4] += 1;
    h->prof.host_last_return = t_end;
  }
  return rc;
}

int lii_neighbors_download(lii_handle h, float* pts, int32_t* counts, uint8_t* selected, int32_t capacity) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
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
    for (int i = 0; i < n; i++) counts[i] = src[perm ? perm[i] : i] & kCountMask;  // (without the completion flags)
  }
  if (selected) {
    HIPCHK(h, hipMemcpyAsync(h->h_stage, h->d_selected, size_t(n), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    const uint8_t* src = reinterpret_cast<const uint8_t*>(h->h_stage);
    for (int i = 0; i < n; i++) selected[i] = src[perm ? perm[i] : i];
  }
  return LII_OK;
}
}  // extern "C"
