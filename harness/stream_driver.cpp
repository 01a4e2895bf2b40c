// A laserMapping-shaped host loop over the C-ABI, in the reference's host language: per scan the propagated state and the
// IMU pose table go in, lii_scan_register (de-skew -> voxel grid -> iterated update) runs, the map takes the scan
// (map_incremental) when asked.  bench.py times THIS loop: a ROS node calling the library is C++, and a ctypes round trip
// per scan costs ~25 us of interpreter time that no deployment would pay.  Harness code, not product: it only calls
// include/liinit_hip.h.   (the call sites it stands for: src/laserMapping.cpp:905-919 + :960-1120 + :1130 map_incremental)
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "liinit_hip.h"

extern "C" {

// The streams this loop is given are time-sorted, as the reference's preprocess hands every scan over (src/preprocess.cpp:296-302):
// the jobs say so (lii_scan_job::scan_sorted).  lii_stream_set_sorted(0) withdraws the claim (A/B, unsorted test streams).
static int32_t g_scan_sorted = 1;
void lii_stream_set_sorted(int32_t sorted) { g_scan_sorted = sorted ? 1 : 0; }
// The map update of a step rides in the registration job (lii_scan_job::map_update: its launches are enqueued behind the update's
// passes); lii_stream_set_map_in_job(0): lii_map_incremental as a call of its own behind lii_scan_register, the form of round 3 (A/B).
static int32_t g_map_in_job = 1;
void lii_stream_set_map_in_job(int32_t in_job) { g_map_in_job = in_job ? 1 : 0; }
// lii_stream_run_wire: 1 = the overlapped ingest (lii_ingest_pcl2_begin / lii_ingest_end), 0 = one lii_ingest_pcl2 call per message
static int32_t g_wire_overlap = 0;
void lii_stream_set_wire_overlap(int32_t overlap) { g_wire_overlap = overlap; }  // (1: begun from the registration's while_waiting hook; 4: behind the registrations; 3: before them; 2: one message under way, behind)

// The scans of these streams are resident in device memory: every job announces its successor (lii_scan_job::next_scan_dev) and the
// library pre-arms that scan's first launch.  lii_stream_set_announce(0): no announcement, the form of ABI <= 7 (A/B).
static int32_t g_announce_next = 1;
void lii_stream_set_announce(int32_t on) { g_announce_next = on ? 1 : 0; }

typedef struct lii_stream_scan {
  const void* scan_dev;      // device-resident float4 (x, y, z, t_ms), caller-owned
  int32_t n_points;
  int32_t n_poses;
  const lii_pose6d* poses;   // IMUpose table of this scan
  const lii_state* state0;   // the state IMU propagation hands to the update (state_propagat == start of the iteration)
} lii_stream_scan;

// Runs steps [first, first + steps) of the cyclic stream.  profile_every > 0: HIP-event kernel timing on every Nth step;
// profile_every < 0: every launch of every step bracketed (lii_set_profiling(h, 3): lii_last_kernel_profile).
// Returns the first non-zero library status; totals[0] += iterations, totals[1] += k-NN passes.
// The slowest step of the last lii_stream_run (host clock, call to return) and where it sat: a one-off stall of the runtime (a pool
// that grows, a page that is read in) shows here instead of hiding in the mean.  out: {slowest [us], its step index, second slowest [us]}
static double g_slowest[3];
void lii_stream_last_slowest(double out[3]) { std::memcpy(out, g_slowest, sizeof(g_slowest)); }

int lii_stream_run(lii_handle h, const lii_stream_scan* scans, int32_t n_scans, int32_t first, int32_t steps, float leaf,
                   int32_t max_iterations, int32_t imu_en, int32_t map_update, int32_t profile_every, int64_t totals[2],
                   lii_state* last_state) {
  if (!h || !scans || n_scans < 1 || steps < 0 || !totals) return LII_ERR_INVALID;
  lii_state st;
  lii_iekf_report rep;
  const bool trace = std::getenv("LII_STREAM_TRACE") != nullptr;  // per-step wall time on stderr (diagnostic)
  auto t_prev = std::chrono::steady_clock::now();
  auto t_step = t_prev;
  g_slowest[0] = g_slowest[1] = g_slowest[2] = 0.0;
  for (int32_t k = first; k < first + steps; k++) {
    if (k > first) {
      const auto t_now = std::chrono::steady_clock::now();
      const double us = std::chrono::duration<double, std::micro>(t_now - t_step).count();
      t_step = t_now;
      if (us > g_slowest[0]) { g_slowest[2] = g_slowest[0]; g_slowest[0] = us; g_slowest[1] = k - 1 - first; }
      else if (us > g_slowest[2]) g_slowest[2] = us;
    }
    if (trace && k > first) {
      const auto t_now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "%.1f ", std::chrono::duration<double, std::micro>(t_now - t_prev).count());
      t_prev = t_now;
    }
    const lii_stream_scan& sc = scans[k % n_scans];
    int rc = lii_set_profiling(h, profile_every < 0 ? 3 : ((profile_every > 0 && k % profile_every == 0) ? 2 : 0));
    if (rc != LII_OK) return rc;
    std::memcpy(&st, sc.state0, sizeof(st));
    lii_scan_job job;
    std::memset(&job, 0, sizeof(job));
    job.struct_size = sizeof(job);
    job.undistort = 1;
    job.imu_poses = sc.poses;
    job.n_imu_poses = sc.n_poses;
    job.leaf = leaf;
    job.opts.max_iterations = max_iterations;
    job.opts.imu_en = imu_en;
    job.scan_dev = sc.scan_dev;
    job.n_scan_dev = sc.n_points;
    job.scan_sorted = g_scan_sorted;
    job.map_update = (map_update && g_map_in_job) ? 1 : 0;
    if (g_announce_next && k + 1 < first + steps) {  // the next scan of the stream is in device memory already: its prologue is pre-armed
      const lii_stream_scan& nx = scans[(k + 1) % n_scans];
      job.next_scan_dev = nx.scan_dev;
      job.next_n_scan = nx.n_points;
    }
    rc = lii_scan_register(h, &job, &st, sc.state0, &rep);
    if (rc != LII_OK) return rc;
    totals[0] += rep.iterations;
    totals[1] += rep.searches;
    if (map_update && !g_map_in_job) {
      rc = lii_map_incremental(h, &st, nullptr, nullptr);
      if (rc != LII_OK) return rc;
    }
  }
  if (trace) std::fprintf(stderr, "%.1f [us per step, %d steps]\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_prev).count(), steps);
  if (last_state && steps > 0) std::memcpy(last_state, &st, sizeof(st));
  return LII_OK;
}

// The COMPLETE per-scan pipeline out of host memory: the next scan travels (lii_scan_upload_next, copy stream) while the current
// one is registered and inserted into the map; lii_scan_advance swaps.  host_scans[k]: (x, y, z, t_ms) float records of scan k
// (pinned memory is read by the copy engine directly, pageable memory is staged by the library).
// (the reference: the driver callback queues the scan - src/laserMapping.cpp:331-366 - while the main loop works on its predecessor)
int lii_stream_run_pipeline(lii_handle h, const lii_stream_scan* scans, const void* const* host_scans, int32_t n_scans, int32_t steps,
                            float leaf, int32_t max_iterations, int32_t imu_en, int32_t overlap, int64_t totals[2]) {
  if (!h || !scans || !host_scans || n_scans < 1 || steps < 0 || !totals) return LII_ERR_INVALID;
  lii_state st;
  lii_iekf_report rep;
  int rc = lii_set_profiling(h, 0);
  if (rc != LII_OK) return rc;
  if (overlap && steps > 0) {
    rc = lii_scan_upload_next(h, host_scans[0], scans[0].n_points, 16, 12);
    if (rc == LII_OK) rc = lii_scan_advance(h);
    if (rc != LII_OK) return rc;
  }
  for (int32_t k = 0; k < steps; k++) {
    const lii_stream_scan& sc = scans[k % n_scans];
    if (overlap) {
      if (k + 1 < steps) rc = lii_scan_upload_next(h, host_scans[(k + 1) % n_scans], scans[(k + 1) % n_scans].n_points, 16, 12);
    } else {
      rc = lii_scan_upload(h, host_scans[k % n_scans], sc.n_points, 16, 12);
    }
    if (rc != LII_OK) return rc;
    std::memcpy(&st, sc.state0, sizeof(st));
    lii_scan_job job;
    std::memset(&job, 0, sizeof(job));
    job.struct_size = sizeof(job);
    job.undistort = 1;
    job.imu_poses = sc.poses;
    job.n_imu_poses = sc.n_poses;
    job.leaf = leaf;
    job.opts.max_iterations = max_iterations;
    job.opts.imu_en = imu_en;
    job.scan_sorted = g_scan_sorted;
    job.map_update = g_map_in_job;
    rc = lii_scan_register(h, &job, &st, sc.state0, &rep);
    if (rc != LII_OK) return rc;
    totals[0] += rep.iterations;
    totals[1] += rep.searches;
    if (!g_map_in_job) rc = lii_map_incremental(h, &st, nullptr, nullptr);
    if (rc != LII_OK) return rc;
    if (overlap && k + 1 < steps) {
      rc = lii_scan_advance(h);
      if (rc != LII_OK) return rc;
    }
  }
  return lii_synchronize(h);
}

// FROM THE WIRE (round 6): a driver message in, poses out.  Per message the PointCloud2 bytes go through the device ingest
// (lii_ingest_pcl2 = process_cut_frame_pcl2: H2D of the raw bytes, decode, filters, time sort, cut into sub-frames), every sub-frame
// becomes the current scan (lii_frame_select) and is registered with the map update in the job - the reference's callback
// (src/laserMapping.cpp:326-379 -> src/preprocess.cpp:115-335) and main loop (:909-1134, map_incremental :516-559) in one host loop.
// scans[j]: state and pose table of sub-frame j of the cyclic stream (message m holds sub-frames m * cut ... m * cut + cut - 1).
// ingest_us[0] += host time inside lii_ingest_pcl2 (its one synchronisation included), [1] += sub-frames registered.
int lii_stream_run_wire(lii_handle h, const lii_stream_scan* scans, int32_t n_scans, const void* const* msgs, const int32_t* msg_points,
                        int32_t n_msgs, int32_t steps, const lii_pc2_fields* fields, const lii_ingest_opts* opts0, float leaf,
                        int32_t max_iterations, int32_t imu_en, int32_t map_update, int64_t totals[2], double ingest_us[2]) {
  if (!h || !scans || !msgs || !msg_points || !fields || !opts0 || n_scans < 1 || n_msgs < 1 || steps < 0 || !totals || !ingest_us) return LII_ERR_INVALID;
  lii_state st;
  lii_iekf_report rep;
  int rc = lii_set_profiling(h, 0);
  if (rc != LII_OK) return rc;
  const int cut = opts0->cut_frame_num > 0 ? opts0->cut_frame_num : 1;
  // g_wire_overlap (lii_stream_set_wire_overlap): the messages queue on the device (lii_ingest_pcl2_begin / lii_ingest_end, ABI 9) -
  // message m + 1 is decoded, and the bytes of m + 2 travel, while the sub-frames of message m are registered
  auto begin = [&](int32_t m) {
    lii_ingest_opts o = *opts0;
    o.stamp_s = opts0->stamp_s + 0.1 * m;
    o.scan_count = opts0->scan_count + m;
    return lii_ingest_pcl2_begin(h, msgs[m % n_msgs], msg_points[m % n_msgs], fields, &o);
  };
  // (how the next message is put under way: 1 = from the registration's while_waiting hook, inside the call, when its launches are out and
  // the thread would only wait; 4 = behind the registrations; 3 = before them; 2 = one message under way instead of two, behind)
  struct Hook { decltype(begin)* fn; int32_t m; int rc; bool due; } hook = {&begin, 0, LII_OK, false};
  auto hook_fn = [](void* a) { Hook* k = static_cast<Hook*>(a); if (k->due) { k->rc = (*k->fn)(k->m); k->due = false; } };
  if (g_wire_overlap) {
    for (int32_t m = 0; m < (g_wire_overlap == 2 ? 1 : 2) && m < steps; m++) {
      rc = begin(m);
      if (rc != LII_OK) return rc;
    }
  }
  for (int32_t m = 0; m < steps; m++) {
    const int32_t jm = m % n_msgs;
    lii_ingest_opts o = *opts0;
    o.stamp_s = opts0->stamp_s + 0.1 * m;
    o.scan_count = opts0->scan_count + m;
    lii_frame_info frames[64];
    int32_t nf = 0;
    const auto t0 = std::chrono::steady_clock::now();
    if (g_wire_overlap) {
      rc = lii_ingest_end(h, frames, 64, &nf);
      if (rc == LII_OK && g_wire_overlap == 3 && m + 2 < steps) rc = begin(m + 2);  // (3: the next message begun BEFORE the registrations, A/B)
    } else {
      rc = lii_ingest_pcl2(h, msgs[jm], msg_points[jm], fields, &o, frames, 64, &nf);
    }
    ingest_us[0] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (rc != LII_OK) return rc;
    for (int32_t f = 0; f < nf; f++) {
      rc = lii_frame_select(h, f);
      if (rc != LII_OK) return rc;
      const lii_stream_scan& sc = scans[(jm * cut + (f < cut ? f : cut - 1)) % n_scans];
      std::memcpy(&st, sc.state0, sizeof(st));
      lii_scan_job job;
      std::memset(&job, 0, sizeof(job));
      job.struct_size = sizeof(job);
      job.undistort = 1;
      job.imu_poses = sc.poses;
      job.n_imu_poses = sc.n_poses;
      job.leaf = leaf;
      job.opts.max_iterations = max_iterations;
      job.opts.imu_en = imu_en;
      job.scan_sorted = 1;  // (the ingest delivers every frame in ascending time order, as the reference's preprocess does)
      job.map_update = map_update ? 1 : 0;
      if (g_wire_overlap == 1 && f == nf - 1 && m + 2 < steps) {  // (with the message's last sub-frame: the context of message m - 1 is free since lii_ingest_end)
        hook.m = m + 2; hook.due = true; hook.rc = LII_OK;
        job.while_waiting = hook_fn; job.while_waiting_arg = &hook;
      }
      rc = lii_scan_register(h, &job, &st, sc.state0, &rep);
      if (rc != LII_OK) return rc;
      if (hook.rc != LII_OK) return hook.rc;
      totals[0] += rep.iterations;
      totals[1] += rep.searches;
      ingest_us[1] += 1.0;
    }
    if (g_wire_overlap == 1 && hook.due) {  // (a message without frames, or a registration that never reached its wait: begun here)
      hook.due = false;
      rc = begin(hook.m);
      if (rc != LII_OK) return rc;
    }
    if (g_wire_overlap == 1 && nf == 0 && m + 2 < steps) {
      rc = begin(m + 2);
      if (rc != LII_OK) return rc;
    }
    // the message after next is put under way HERE: its copy and launches are enqueued while the device still runs the map update of
    // the registration that has just returned (begun before the registrations, the host's ~ 50 us of enqueueing sat in front of them)
    if (g_wire_overlap == 4 || g_wire_overlap == 2) {
      const int32_t ahead = g_wire_overlap == 2 ? 1 : 2;
      const auto t1 = std::chrono::steady_clock::now();
      if (m + ahead < steps) rc = begin(m + ahead);
      ingest_us[0] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
      if (rc != LII_OK) return rc;
    }
  }
  return lii_synchronize(h);
}

}  // extern "C"
