// Ingest on the device: wire-format decode, blind / NaN / ring / decimation filters, per-point time synthesis, time sort and
// sub-frame cutting, straight into device-resident scan frames (gfx950).  Replaces
//   Preprocess::process_cut_frame_livox   reference src/preprocess.cpp:50-113
//   Preprocess::process_cut_frame_pcl2    reference src/preprocess.cpp:115-335
//   Preprocess::process (oust_handler / velodyne_handler / l515_handler / avia_handler, feature extraction disabled)
//                                         reference src/preprocess.cpp:337-713 - lii_ingest_opts::cut_frame_num == 0: the callbacks'
//                                         branch for initialization/cut_frame: false (no time sort, no cut, input order)
// for the point layouts of src/preprocess.h:35-116.  HBM-bound byte work: one pass over the raw records, one scan, one
// stable radix sort of the kept points by time, one pass that writes the frames.  One host synchronisation per message
// (the frame table lands in mapped host memory).  Overlapped forms (ABI 9: lii_ingest_*_begin / lii_ingest_end): a ring of three message
// contexts, the raw bytes on a copy stream, the launches on a stream of their own - message k + 1 (and the bytes of k + 2) travel and are
// decoded while the handle's stream registers the sub-frames of message k (the reference queues driver messages the same way:
// lidar_buffer / time_buffer, src/laserMapping.cpp:326-379, consumed by sync_packages :432-480).
//
// The order of points with equal time stamps is the input order (stable sort); the reference's std::sort leaves it
// implementation-defined (oracle/orc_ingest.hpp).
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>

#include "../../include/liinit_hip.h"
#include "lii_device.h"
#include "lii_launch.h"

namespace lii {

namespace {

constexpr int kMaxLines = 128;   // MAX_LINE_NUM, src/preprocess.cpp:114
constexpr int kMaxFrames = 64;

struct IngestTable {  // written by k_cut_plan into mapped host memory
  int n_frames;
  int n_kept;        // pl_surf.size()
  int n_emitted;     // points inside frames
  int unsorted;      // 1: the kept points did NOT arrive in ascending time order (k_cut_apply)
  int first[kMaxFrames];   // sorted index of the first point of frame c
  int last[kMaxFrames];    // sorted index of the boundary point of frame c
  double begin_ms[kMaxFrames];
  double delta[kMaxFrames];  // stamp_ms - last_frame_end_time while frame c is filled
  double tail_ms[kMaxFrames];  // adjusted curvature of the frame's last (boundary) point
};

struct IngestCtx {
  int cap = 0;  // raw points the buffers hold
  size_t raw_bytes = 0;
  uint8_t* d_raw = nullptr;
  float4* d_pts = nullptr;       // decoded (x, y, z, curvature) per raw point
  double* d_yaw = nullptr;       // azimuth [deg] per raw point (time synthesis)
  uint8_t* d_ring = nullptr;
  unsigned int *d_flag = nullptr, *d_rank = nullptr, *d_aux = nullptr, *d_aux_rank = nullptr;
  unsigned int *d_key_a = nullptr, *d_key_b = nullptr, *d_idx_a = nullptr, *d_idx_b = nullptr;
  float4* d_frames = nullptr;    // the emitted frames, contiguous
  void* d_temp = nullptr;
  size_t temp_bytes = 0;
  IngestTable* h_table = nullptr;  // pinned + mapped
  IngestTable table;               // host copy of the last message
  bool have = false;
  bool cut_msg = false;            // the message in this context went through the time sort + cut (cut_frame_num != 0)
  bool sort_skipped = false;       // the message in this context was enqueued without its time sort (IngestRing::predict_sorted)
  int n_tail = 0, required_tail = 0;  // what ingest_redo_sorted needs of the message: its raw point count, the cut it asked for
  double stamp_ms_tail = 0;
  int n_under_way = 0;              // points of the message an overlapped call put under way in this context
};

template <class T>
__device__ __forceinline__ T rd(const uint8_t* p) {  // unaligned-safe field read
  T v;
  memcpy(&v, p, sizeof(T));
  return v;
}
__device__ __forceinline__ unsigned int ford(float f) {  // order-preserving float -> uint
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Pc2Arg {
  lii_pc2_fields f;
  int lidar_type, n_scans, pfn;
  double blind2, stamp_s;
  int whole;  // Preprocess::process instead of process_cut_frame_pcl2 (the handlers differ in which filters they apply)
};

// given_offset_time (src/preprocess.cpp:132-139, :246-253): the LAST point carries a positive time
__device__ __forceinline__ bool pc2_given_time(const uint8_t* raw, int n, const Pc2Arg& a) {
  if (a.lidar_type != LII_LIDAR_VELO && a.lidar_type != LII_LIDAR_ROBOSENSE) return true;
  const uint8_t* last = raw + (size_t)(n - 1) * a.f.point_step;
  const double tl = a.lidar_type == LII_LIDAR_VELO ? (double)rd<float>(last + a.f.time) : rd<double>(last + a.f.time);
  return tl > 0;
}

// one thread per raw point: decode, blind / NaN test, decimation and ring filter (the per-type loops :141-294)
__global__ void k_pc2_decode(const uint8_t* __restrict__ raw, int n, Pc2Arg a, float4* __restrict__ pts, double* __restrict__ yaw,
                             uint8_t* __restrict__ ring_out, unsigned int* __restrict__ pass0, unsigned int* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* p = raw + (size_t)i * a.f.point_step;
  const float x = rd<float>(p + a.f.x), y = rd<float>(p + a.f.y), z = rd<float>(p + a.f.z);
  float t;
  int ring;
  if (a.lidar_type == LII_LIDAR_VELO) {
    t = (float)((double)rd<float>(p + a.f.time) * 1000.0);
    ring = rd<uint16_t>(p + a.f.ring);
  } else if (a.lidar_type == LII_LIDAR_OUSTER) {
    t = (float)((double)rd<uint32_t>(p + a.f.time) / 1e6);
    ring = rd<uint8_t>(p + a.f.ring);
  } else if (a.lidar_type == LII_LIDAR_PANDAR) {
    const double ts0 = rd<double>(raw + a.f.time);
    t = (float)((rd<double>(p + a.f.time) - ts0) * 1000);
    ring = rd<uint16_t>(p + a.f.ring);
  } else if (a.lidar_type == LII_LIDAR_L515) {  // l515_handler (:444-470): x, y, z only; curvature 0
    t = 0.f;
    ring = 0;
  } else {
    t = (float)((rd<double>(p + a.f.time) - a.stamp_s + 0.1) * 1000.0);
    ring = rd<uint16_t>(p + a.f.ring);
  }
  const float d = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));  // float arithmetic, then widened
  // (l515_handler tests the range alone: a NaN range is not below the blind radius, such a point passes)
  const bool ok = a.lidar_type == LII_LIDAR_L515 ? !((double)d < a.blind2) : !((double)d < a.blind2 || isnan(x) || isnan(y) || isnan(z));
  pts[i] = make_float4(x, y, z, t);
  ring_out[i] = (uint8_t)min(ring, 255);
  yaw[i] = atan2((double)y, (double)x) * 57.2957;
  pass0[i] = ok ? 1u : 0u;
  // (velodyne_handler's feature-less branch, :661-700, has no `ring < N_SCANS` test; oust_handler's, :545-563, has)
  const bool ring_ok = (a.whole && a.lidar_type != LII_LIDAR_OUSTER) || ring < a.n_scans;
  keep[i] = (ok && (i % a.pfn == 0) && ring_ok) ? 1u : 0u;
}

// Time synthesis from the azimuth when the driver gives none (:163-185, :272-294): a sequential recurrence per ring —
// one lane per ring walks the cloud in input order.  The ring's first surviving point only seeds the recurrence and is
// dropped (`continue`).  Returns at once when the message carries per-point time.
__global__ __launch_bounds__(kMaxLines) void k_pc2_ring_times(const uint8_t* __restrict__ raw, int n, Pc2Arg a,
                                                              float4* __restrict__ pts, const double* __restrict__ yaw,
                                                              const uint8_t* __restrict__ ring,
                                                              const unsigned int* __restrict__ pass0,
                                                              unsigned int* __restrict__ keep) {
  if (n <= 0 || pc2_given_time(raw, n, a)) return;
  const int layer = threadIdx.x;
  const double omega_l = 3.61;
  bool first = true;
  double yaw_fp = 0;
  float time_last = 0.f;
  for (int i = 0; i < n; i++) {
    if (ring[i] != layer || !pass0[i]) continue;
    const double yaw_angle = yaw[i];
    if (first) {
      yaw_fp = yaw_angle;
      first = false;
      time_last = 0.f;
      keep[i] = 0u;
      continue;
    }
    float t;
    if (yaw_angle <= yaw_fp) t = (float)((yaw_fp - yaw_angle) / omega_l);
    else t = (float)((yaw_fp - yaw_angle + 360.0) / omega_l);
    if (t < time_last) t = (float)((double)t + 360.0 / omega_l);
    time_last = t;
    pts[i].w = t;
  }
}
// rings >= 128 index out of bounds in the reference's time synthesis; such points are dropped here
__global__ void k_pc2_drop_wide_rings(const uint8_t* __restrict__ raw, int n, Pc2Arg a, const uint8_t* __restrict__ ring,
                                      unsigned int* __restrict__ keep) {
  if (n <= 0 || pc2_given_time(raw, n, a)) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && ring[i] >= kMaxLines) keep[i] = 0u;
}

struct LivoxArg {
  lii_livox_fields f;
  int n_scans, pfn;
  double blind2;
};
// v_i = the tag / line test of point i >= 1 (:60-62)
__global__ void k_livox_valid(const uint8_t* __restrict__ raw, int n, LivoxArg a, unsigned int* __restrict__ valid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* p = raw + (size_t)i * a.f.point_step;
  const int line = rd<uint8_t>(p + a.f.line), tag = rd<uint8_t>(p + a.f.tag);
  valid[i] = (i >= 1 && line < a.n_scans && ((tag & 0x30) == 0x10 || (tag & 0x30) == 0x00)) ? 1u : 0u;
}
// cnt = inclusive scan of v (valid_point_num).  A point is decoded ("populated" in pl_full) when v_i and cnt_i % pfn == 0;
// it is kept when it is outside the blind zone and differs from pl_full[i-1], which is the previous RAW point if that one
// was populated and the zero point otherwise (:64-81).
__global__ void k_livox_decode(const uint8_t* __restrict__ raw, int n, LivoxArg a, const unsigned int* __restrict__ valid,
                               const unsigned int* __restrict__ cnt, float4* __restrict__ pts, unsigned int* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool k = false;
  float4 out = make_float4(0, 0, 0, 0);
  if (valid[i] && (cnt[i] % (unsigned)a.pfn) == 0u) {
    const uint8_t* p = raw + (size_t)i * a.f.point_step;
    const float x = rd<float>(p + a.f.x), y = rd<float>(p + a.f.y), z = rd<float>(p + a.f.z);
    const float t = __fdiv_rn((float)rd<uint32_t>(p + a.f.offset_time), 1000000.f);
    out = make_float4(x, y, z, t);
    const float d = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
    if (!((double)d < a.blind2)) {
      float px = 0.f, py = 0.f, pz = 0.f;
      if (valid[i - 1] && (cnt[i - 1] % (unsigned)a.pfn) == 0u) {
        const uint8_t* q = raw + (size_t)(i - 1) * a.f.point_step;
        px = rd<float>(q + a.f.x); py = rd<float>(q + a.f.y); pz = rd<float>(q + a.f.z);
      }
      k = ((double)fabsf(__fsub_rn(x, px)) > 1e-7) || ((double)fabsf(__fsub_rn(y, py)) > 1e-7) ||
          ((double)fabsf(__fsub_rn(z, pz)) > 1e-7);
    }
  }
  pts[i] = out;
  keep[i] = k ? 1u : 0u;
}

// kept points -> (time key, raw index) in input order
__global__ void k_ingest_compact(const float4* __restrict__ pts, const unsigned int* __restrict__ keep,
                                 const unsigned int* __restrict__ rank, int n, unsigned int* __restrict__ key,
                                 unsigned int* __restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned int kkey = 0xFFFFFFFFu, kidx = 0u;  // padding beyond the kept count sorts last
  if (keep[i]) {
    const unsigned int r = rank[i] - 1u;
    key[r] = ford(pts[i].w);
    idx[r] = (unsigned)i;
  }
  // slots [kept, n) are filled by the tail threads so that the sort can run over the fixed length n
  const unsigned int kept = rank[n - 1];
  if ((unsigned)i >= kept) { key[i] = kkey; idx[i] = kidx; }
}

// The cutting loop (:88-112 == :298-334) as a plan: boundary indices from the unsigned-arithmetic test, the
// last_frame_end_time chain in double from the boundary points' float times.  One thread.
__global__ void k_cut_plan(const float4* __restrict__ pts, const unsigned int* __restrict__ sorted_idx,
                           const unsigned int* __restrict__ rank, int n, double stamp_ms, int required_cut_num,
                           IngestTable* __restrict__ dev, IngestTable* __restrict__ host) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  IngestTable t;
  const unsigned int size = n > 0 ? rank[n - 1] : 0u;
  t.n_kept = (int)size;
  t.n_frames = 0;
  t.n_emitted = 0;
  t.unsorted = 0;
  double lfe = stamp_ms;
  unsigned int cut_num = 0;
  int start = 1;
  while (t.n_frames < kMaxFrames) {
    const int b = (int)((cut_num + 1u) * size / (unsigned)required_cut_num) - 1;
    if (b < start || b > (int)size - 1) break;  // a boundary the running counter can never equal: cutting stops
    const int c = t.n_frames;
    const double delta = stamp_ms - lfe;
    t.first[c] = start;
    t.last[c] = b;
    t.begin_ms[c] = lfe;
    t.delta[c] = delta;
    const float tb = (float)((double)pts[sorted_idx[b]].w + delta);  // the boundary point's adjusted curvature
    t.tail_ms[c] = (double)tb;
    lfe += (double)tb;
    t.n_frames = c + 1;
    t.n_emitted = b;  // points 1..b
    cut_num++;
    start = b + 1;
  }
  *dev = t;
  *host = t;
}
// frame point (sorted position s in [1, n_emitted]) -> frames[s - 1] with curvature += stamp_ms - last_frame_end_time
// ... and the launch looks at the kept points' time keys in INPUT order (key_in; position s against s - 1): a message that arrives in
// ascending time order - drivers that publish in firing order do - needs no sort at all, and the next message of the stream is then
// enqueued without one (IngestRing::predict_sorted).  `unsorted` (device table and its host mirror) is raised by any lane that sees a
// descent; a message that was enqueued without the sort and raises it is done again with the sort (ingest_redo_sorted).
__global__ void k_cut_apply(const float4* __restrict__ pts, const unsigned int* __restrict__ sorted_idx,
                            IngestTable* __restrict__ tab, int n, float4* __restrict__ frames, const unsigned int* __restrict__ key_in,
                            IngestTable* __restrict__ host_tab) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x + 1;
  if (s < tab->n_kept && s < n && key_in[s] < key_in[s - 1]) {
    tab->unsorted = 1;
    host_tab->unsorted = 1;
  }
  if (s > tab->n_emitted || s >= n + 1) return;
  int c = 0;
  while (c + 1 < tab->n_frames && s > tab->last[c]) c++;
  float4 p = pts[sorted_idx[s]];
  p.w = (float)((double)p.w + tab->delta[c]);
  frames[s - 1] = p;
}

// Preprocess::process: the kept points in input order are THE cloud - one frame, stamped with the message's own time (the caller
// pushes header.stamp, laserMapping.cpp:340,377), the first kept point included, curvatures as decoded.
__global__ void k_whole_apply(const float4* __restrict__ pts, const unsigned int* __restrict__ idx, const unsigned int* __restrict__ rank, int n,
                              double stamp_ms, float4* __restrict__ frames, IngestTable* __restrict__ dev, IngestTable* __restrict__ host) {
  const unsigned int kept = n > 0 ? rank[n - 1] : 0u;
  const unsigned int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < kept) frames[r] = pts[idx[r]];
  if (r == 0) {
    IngestTable t;
    t.n_kept = (int)kept;
    t.n_frames = kept > 0u ? 1 : 0;  // (an empty cloud is skipped by the node: laserMapping.cpp:909-914)
    t.n_emitted = (int)kept;
    t.unsorted = 0;
    t.first[0] = 1;
    t.last[0] = (int)kept;
    t.begin_ms[0] = stamp_ms;
    t.delta[0] = 0.0;
    t.tail_ms[0] = kept > 0u ? (double)pts[idx[kept - 1u]].w : 0.0;  // points.back().curvature (sync_packages, laserMapping.cpp:452-456)
    *dev = t;
    *host = t;
  }
}

template <class T>
hipError_t dm(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * (n ? n : 1)); }

void ingest_release(IngestCtx* c) {
  void* ptrs[] = {c->d_raw, c->d_pts, c->d_yaw, c->d_ring, c->d_flag, c->d_rank, c->d_aux, c->d_aux_rank, c->d_key_a, c->d_key_b,
                  c->d_idx_a, c->d_idx_b, c->d_frames, c->d_temp};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (c->h_table) (void)hipHostFree(c->h_table);
}

// The handle's ingest state: a ring of three message contexts.  `front` holds the message whose frames lii_frame_select serves (the
// one-call forms lii_ingest_pcl2 / _livox work in it, on the handle's stream).  The overlapped forms (lii_ingest_*_begin / lii_ingest_end,
// ABI 9) put up to two more messages under way in the other contexts: raw bytes on a copy stream, decode ... cut on a
// kernel stream behind them, while the handle's stream registers the sub-frames of `front`.
struct IngestRing {
  static constexpr int kSlots = 3;
  IngestCtx slot[kSlots];
  int front = 0;
  int pending[kSlots] = {0, 0, 0};  // FIFO of contexts under way, oldest first
  int n_pending = 0;
  hipStream_t s_copy = nullptr, s_kern = nullptr;
  hipEvent_t ev_copied[kSlots] = {nullptr, nullptr, nullptr}, ev_done[kSlots] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_read = nullptr;  // recorded on the handle's stream when a message leaves `front`: whatever that stream still has to do with its frames lies in front of it
  bool guard[kSlots] = {false, false, false};  // ... and the contexts whose next message waits for it
  // The launch plan of the time sort (its eleven launches are two thirds of a message's kernel time): the last cut message's kept points
  // arrived in ascending time order -> the next one is enqueued WITHOUT the sort, the device checks the order, a message that fails the
  // check is done again with the sort before its frames are handed out.  LII_INGEST_SORT=always: never predicted.
  bool predict_sorted = false;
  bool never_predict = false;
  long long n_unsorted_skipped = 0, n_redone = 0;
};

void ingest_free(IngestRing* r) {
  if (!r) return;
  if (getenv("LII_DIAG") && (r->n_unsorted_skipped || r->n_redone))
    fprintf(stderr, "[libliinit_hip] ingest: messages cut without their time sort (they arrived in time order): %lld, done again with it: %lld\n",
            r->n_unsorted_skipped, r->n_redone);
  if (r->s_copy) { (void)hipStreamSynchronize(r->s_copy); (void)hipStreamDestroy(r->s_copy); }
  if (r->s_kern) { (void)hipStreamSynchronize(r->s_kern); (void)hipStreamDestroy(r->s_kern); }
  for (int k = 0; k < IngestRing::kSlots; k++) {
    if (r->ev_copied[k]) (void)hipEventDestroy(r->ev_copied[k]);
    if (r->ev_done[k]) (void)hipEventDestroy(r->ev_done[k]);
    ingest_release(&r->slot[k]);
  }
  if (r->ev_read) (void)hipEventDestroy(r->ev_read);
  delete r;
}

#define ICHK(h, expr)                                                                                         \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) return lii_internal_fail(h, LII_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

int ingest_reserve(lii_handle h, IngestCtx* c, int n, size_t raw_bytes) {
  if (!c->h_table) ICHK(h, hipHostMalloc(reinterpret_cast<void**>(&c->h_table), sizeof(IngestTable), hipHostMallocMapped));
  if (raw_bytes > c->raw_bytes) {
    if (c->d_raw) (void)hipFree(c->d_raw);
    c->d_raw = nullptr;
    ICHK(h, dm(&c->d_raw, raw_bytes + 64));
    c->raw_bytes = raw_bytes;
  }
  if (n > c->cap) {
    void** ptrs[] = {(void**)&c->d_pts, (void**)&c->d_yaw, (void**)&c->d_ring, (void**)&c->d_flag, (void**)&c->d_rank, (void**)&c->d_aux,
                     (void**)&c->d_aux_rank, (void**)&c->d_key_a, (void**)&c->d_key_b, (void**)&c->d_idx_a, (void**)&c->d_idx_b,
                     (void**)&c->d_frames, (void**)&c->d_temp};
    for (void** p : ptrs) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    const int cap = n + n / 4 + 1024;
    ICHK(h, dm(&c->d_pts, cap));
    ICHK(h, dm(&c->d_yaw, cap));
    ICHK(h, dm(&c->d_ring, cap));
    ICHK(h, dm(&c->d_flag, cap));
    ICHK(h, dm(&c->d_rank, cap));
    ICHK(h, dm(&c->d_aux, cap));
    ICHK(h, dm(&c->d_aux_rank, cap));
    ICHK(h, dm(&c->d_key_a, cap));
    ICHK(h, dm(&c->d_key_b, cap));
    ICHK(h, dm(&c->d_idx_a, cap));
    ICHK(h, dm(&c->d_idx_b, cap));
    ICHK(h, dm(&c->d_frames, cap));
    c->temp_bytes = sort_temp_bytes(cap);
    ICHK(h, hipMalloc(&c->d_temp, c->temp_bytes));
    c->cap = cap;
  }
  return LII_OK;
}

IngestRing* ring_of(lii_handle h) {
  void** slot = lii_internal_ingest_slot(h);
  if (!*slot) {
    IngestRing* r = new IngestRing();
    const char* e = getenv("LII_INGEST_SORT");
    r->never_predict = e && e[0] == 'a';
    *slot = r;
  }
  return static_cast<IngestRing*>(*slot);
}

// The overlapped forms' streams, at the DEFAULT priority: a low-priority stream was measured first (a registration launch ahead of a
// message that is not due yet, the idea went) and made the whole loop 2.4 x SLOWER than the serial form - 1 904 against 4 454 scans/s
// at the default priority, 2 541 serial (profiles/r06_ingest_overlap.md): queues of unequal priority are time-sliced on this device.
// LII_INGEST_PRIO=low brings that form back for measurements.
int ring_streams(lii_handle h, IngestRing* r) {
  if (r->s_kern) return LII_OK;
  const char* ep = getenv("LII_INGEST_PRIO");
  if (ep && ep[0] == 'l') {
    int least = 0, greatest = 0;
    ICHK(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
    ICHK(h, hipStreamCreateWithPriority(&r->s_kern, hipStreamNonBlocking, least));
  } else {
    ICHK(h, hipStreamCreateWithFlags(&r->s_kern, hipStreamNonBlocking));
  }
  ICHK(h, hipStreamCreateWithFlags(&r->s_copy, hipStreamNonBlocking));
  for (int k = 0; k < IngestRing::kSlots; k++) {
    ICHK(h, hipEventCreateWithFlags(&r->ev_copied[k], hipEventDisableTiming));
    ICHK(h, hipEventCreateWithFlags(&r->ev_done[k], hipEventDisableTiming));
  }
  ICHK(h, hipEventCreateWithFlags(&r->ev_read, hipEventDisableTiming));
  return LII_OK;
}

// shared tail, enqueued on `s`: scan of the keep flags, compaction, stable sort by time, plan, apply (the frame table lands in mapped host memory)
int ingest_tail(lii_handle h, IngestCtx* c, int n, const lii_ingest_opts* o, int uncut_below, hipStream_t s) {
  const int nb = (n + 255) / 256;
  inclusive_scan_u32(c->d_temp, c->temp_bytes, c->d_flag, c->d_rank, n, s);
  hipLaunchKernelGGL(k_ingest_compact, dim3(nb), dim3(256), 0, s, c->d_pts, c->d_flag, c->d_rank, n, c->d_key_a, c->d_idx_a);
  IngestTable* d_table = reinterpret_cast<IngestTable*>(c->d_aux);  // d_aux is free again after the Livox scan
  c->sort_skipped = false;
  c->cut_msg = o->cut_frame_num != 0;
  if (o->cut_frame_num == 0) {  // Preprocess::process: no sort, no cut
    hipLaunchKernelGGL(k_whole_apply, dim3(nb), dim3(256), 0, s, c->d_pts, c->d_idx_a, c->d_rank, n, o->stamp_s * 1000, c->d_frames, d_table, c->h_table);
  } else {
    IngestRing* r = ring_of(h);
    c->sort_skipped = r->predict_sorted && !r->never_predict;
    if (!c->sort_skipped) sort_pairs_u32(c->d_temp, c->temp_bytes, c->d_key_a, c->d_key_b, c->d_idx_a, c->d_idx_b, n, s);
    const unsigned int* order = c->sort_skipped ? c->d_idx_a : c->d_idx_b;  // (a stable sort of keys in ascending order is the identity)
    int required = o->cut_frame_num;
    if (o->scan_count < uncut_below) required = 1;
    c->n_tail = n; c->required_tail = required; c->stamp_ms_tail = o->stamp_s * 1000;
    hipLaunchKernelGGL(k_cut_plan, dim3(1), dim3(64), 0, s, c->d_pts, order, c->d_rank, n, c->stamp_ms_tail, required, d_table, c->h_table);
    hipLaunchKernelGGL(k_cut_apply, dim3(nb), dim3(256), 0, s, c->d_pts, order, d_table, n, c->d_frames, c->d_key_a, c->h_table);
  }
  ICHK(h, hipGetLastError());
  return LII_OK;
}
// A message that was enqueued without its time sort and turned out not to be in time order: sort, plan and apply again (on `s`, waited for).
int ingest_redo_sorted(lii_handle h, IngestCtx* c, hipStream_t s) {
  const int n = c->n_tail, nb = (n + 255) / 256;
  IngestTable* d_table = reinterpret_cast<IngestTable*>(c->d_aux);
  sort_pairs_u32(c->d_temp, c->temp_bytes, c->d_key_a, c->d_key_b, c->d_idx_a, c->d_idx_b, n, s);
  hipLaunchKernelGGL(k_cut_plan, dim3(1), dim3(64), 0, s, c->d_pts, c->d_idx_b, c->d_rank, n, c->stamp_ms_tail, c->required_tail, d_table, c->h_table);
  hipLaunchKernelGGL(k_cut_apply, dim3(nb), dim3(256), 0, s, c->d_pts, c->d_idx_b, d_table, n, c->d_frames, c->d_key_a, c->h_table);
  ICHK(h, hipGetLastError());
  ICHK(h, hipStreamSynchronize(s));
  c->sort_skipped = false;
  return LII_OK;
}
// ... and, once whatever stream ran it is known to be through: the frame table out
int ingest_collect(lii_handle h, IngestCtx* c, lii_frame_info* frames, int32_t max_frames, int32_t* n_frames, hipStream_t s) {
  if (c->cut_msg) {
    IngestRing* r = ring_of(h);
    const bool unsorted = c->h_table->unsorted != 0;
    if (unsorted && c->sort_skipped) {  // the prediction was wrong for this message: its frames are cut again, behind the sort
      r->n_redone++;
      const int rc = ingest_redo_sorted(h, c, s);
      if (rc != LII_OK) return rc;
    } else if (c->sort_skipped) {
      r->n_unsorted_skipped++;
    }
    r->predict_sorted = !unsorted;
  }
  c->table = *c->h_table;
  c->have = true;
  *n_frames = c->table.n_frames;
  if (c->table.n_frames > max_frames) return lii_internal_fail(h, LII_ERR_CAPACITY, "lii_ingest: more frames than max_frames");
  for (int k = 0; k < c->table.n_frames; k++) {
    frames[k].begin_time_s = c->table.begin_ms[k] / double(1000);  // laserMapping.cpp:334,370
    frames[k].offset = c->table.first[k] - 1;
    frames[k].count = c->table.last[k] - c->table.first[k] + 1;
    frames[k].last_offset_ms = c->table.tail_ms[k];
  }
  return LII_OK;
}

// argument checks of the PointCloud2 forms; *o may be redirected to `whole_opts`
int pcl2_check(lii_handle h, const void* data, int32_t n_points, const lii_pc2_fields* f, const lii_ingest_opts** po, lii_ingest_opts* whole_opts) {
  const lii_ingest_opts* o = *po;
  if (!h || !f || !o || o->struct_size != sizeof(lii_ingest_opts) || n_points < 0 || (!data && n_points > 0) ||
      f->point_step <= 0 || o->point_filter_num < 1 || o->cut_frame_num < 0 || o->cut_frame_num > kMaxFrames)
    return lii_internal_fail(h, LII_ERR_INVALID, "lii_ingest_pcl2: bad arguments");
  // An L515 message is ONE frame whatever cut_frame says: the reference's callback sends only Velodyne, Ouster, Pandar and RoboSense
  // through process_cut_frame_pcl2 and everything else - L515 - through Preprocess::process (src/laserMapping.cpp:363-377).  ADVICE r5:
  // with `cut_frame: true` in the yaml this call used to fail with "Wrong LiDAR Type" where the reference processes the message.
  if (o->lidar_type == LII_LIDAR_L515 && o->cut_frame_num != 0) {
    *whole_opts = *o;
    whole_opts->cut_frame_num = 0;
    o = *po = whole_opts;
  }
  if (o->cut_frame_num == 0) {  // Preprocess::process knows Ouster, Velodyne and L515 (src/preprocess.cpp:337-354)
    if (o->lidar_type != LII_LIDAR_VELO && o->lidar_type != LII_LIDAR_OUSTER && o->lidar_type != LII_LIDAR_L515)
      return lii_internal_fail(h, LII_ERR_INVALID, "lii_ingest_pcl2: Error LiDAR Type (src/preprocess.cpp:350-352: the non-cutting Preprocess::process handles Ouster, Velodyne and L515)");
  } else if (o->lidar_type != LII_LIDAR_VELO && o->lidar_type != LII_LIDAR_OUSTER && o->lidar_type != LII_LIDAR_PANDAR &&
             o->lidar_type != LII_LIDAR_ROBOSENSE)
    return lii_internal_fail(h, LII_ERR_INVALID, "lii_ingest_pcl2: Wrong LiDAR Type (src/preprocess.cpp:290-292)");
  return LII_OK;
}
int livox_check(lii_handle h, const void* points, int32_t n_points, const lii_livox_fields* f, const lii_ingest_opts* o) {
  if (!h || !f || !o || o->struct_size != sizeof(lii_ingest_opts) || n_points < 0 || (!points && n_points > 0) ||
      f->point_step <= 0 || o->point_filter_num < 1 || o->cut_frame_num < 0 || o->cut_frame_num > kMaxFrames)
    return lii_internal_fail(h, LII_ERR_INVALID, "lii_ingest_livox: bad arguments");
  return LII_OK;
}

// H2D of the raw bytes on `s_copy`, the message's launches on `s_kern` behind it (one stream for the one-call forms)
int pcl2_enqueue(lii_handle h, IngestCtx* c, const void* data, int32_t n_points, const lii_pc2_fields* f, const lii_ingest_opts* o,
                 hipStream_t s_copy, hipStream_t s, hipEvent_t ev_copied) {
  const size_t bytes = (size_t)n_points * f->point_step;
  int rc = ingest_reserve(h, c, n_points, bytes);
  if (rc != LII_OK) return rc;
  ICHK(h, hipMemcpyAsync(c->d_raw, data, bytes, hipMemcpyHostToDevice, s_copy));
  if (s_copy != s) {
    ICHK(h, hipEventRecord(ev_copied, s_copy));
    ICHK(h, hipStreamWaitEvent(s, ev_copied, 0));
  }
  Pc2Arg a;
  a.f = *f;
  a.lidar_type = o->lidar_type;
  a.n_scans = o->n_scans;
  a.pfn = o->point_filter_num;
  a.blind2 = o->blind * o->blind;
  a.stamp_s = o->stamp_s;
  a.whole = o->cut_frame_num == 0 ? 1 : 0;
  const int nb = (n_points + 255) / 256;
  hipLaunchKernelGGL(k_pc2_decode, dim3(nb), dim3(256), 0, s, c->d_raw, n_points, a, c->d_pts, c->d_yaw, c->d_ring, c->d_aux, c->d_flag);
  if (o->lidar_type == LII_LIDAR_VELO || o->lidar_type == LII_LIDAR_ROBOSENSE) {
    hipLaunchKernelGGL(k_pc2_ring_times, dim3(1), dim3(kMaxLines), 0, s, c->d_raw, n_points, a, c->d_pts, c->d_yaw, c->d_ring, c->d_aux,
                       c->d_flag);
    hipLaunchKernelGGL(k_pc2_drop_wide_rings, dim3(nb), dim3(256), 0, s, c->d_raw, n_points, a, c->d_ring, c->d_flag);
  }
  return ingest_tail(h, c, n_points, o, 20, s);
}
int livox_enqueue(lii_handle h, IngestCtx* c, const void* points, int32_t n_points, const lii_livox_fields* f, const lii_ingest_opts* o,
                  hipStream_t s_copy, hipStream_t s, hipEvent_t ev_copied) {
  const size_t bytes = (size_t)n_points * f->point_step;
  int rc = ingest_reserve(h, c, n_points, bytes);
  if (rc != LII_OK) return rc;
  ICHK(h, hipMemcpyAsync(c->d_raw, points, bytes, hipMemcpyHostToDevice, s_copy));
  if (s_copy != s) {
    ICHK(h, hipEventRecord(ev_copied, s_copy));
    ICHK(h, hipStreamWaitEvent(s, ev_copied, 0));
  }
  LivoxArg a;
  a.f = *f;
  a.n_scans = o->n_scans;
  a.pfn = o->point_filter_num;
  a.blind2 = o->blind * o->blind;
  const int nb = (n_points + 255) / 256;
  hipLaunchKernelGGL(k_livox_valid, dim3(nb), dim3(256), 0, s, c->d_raw, n_points, a, c->d_aux);
  inclusive_scan_u32(c->d_temp, c->temp_bytes, c->d_aux, c->d_aux_rank, n_points, s);
  hipLaunchKernelGGL(k_livox_decode, dim3(nb), dim3(256), 0, s, c->d_raw, n_points, a, c->d_aux, c->d_aux_rank, c->d_pts, c->d_flag);
  return ingest_tail(h, c, n_points, o, 5, s);
}

// A context for the next overlapped message: not `front`, not under way.  What the handle's stream may still be doing with the frames of
// the message that lay in it comes first (IngestRing::ev_read).
int ring_take(lii_handle h, IngestRing* r, int* out) {
  int rc = ring_streams(h, r);
  if (rc != LII_OK) return rc;
  if (r->n_pending >= IngestRing::kSlots - 1)
    return lii_internal_fail(h, LII_ERR_STATE, "lii_ingest_*_begin: two messages are under way already (call lii_ingest_end)");
  int k = -1;
  for (int q = 0; q < IngestRing::kSlots && k < 0; q++) {
    bool used = q == r->front;
    for (int p = 0; p < r->n_pending; p++) used = used || r->pending[p] == q;
    if (!used) k = q;
  }
  if (r->guard[k]) {  // (the event's latest record lies behind every earlier one on the same stream)
    ICHK(h, hipStreamWaitEvent(r->s_copy, r->ev_read, 0));
    r->guard[k] = false;
  }
  *out = k;
  return LII_OK;
}
void ring_push(IngestRing* r, int k) { r->pending[r->n_pending++] = k; }

}  // namespace

void ingest_destroy(void* slot) { ingest_free(static_cast<IngestRing*>(slot)); }

}  // namespace lii

using namespace lii;

extern "C" {

int lii_ingest_pcl2(lii_handle h, const void* data, int32_t n_points, const lii_pc2_fields* f, const lii_ingest_opts* o,
                    lii_frame_info* frames, int32_t max_frames, int32_t* n_frames) {
  if (lii_internal_in_wait_hook(h)) return lii_internal_fail(h, LII_ERR_STATE, "lii_ingest_pcl2: not from lii_scan_job::while_waiting (lii_ingest_pcl2_begin is)");
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!frames || !n_frames) return lii_internal_fail(h, LII_ERR_INVALID, "lii_ingest_pcl2: bad arguments");
  lii_ingest_opts whole_opts;
  int rc = pcl2_check(h, data, n_points, f, &o, &whole_opts);
  if (rc != LII_OK) return rc;
  { const int rcm = lii_internal_scan_materialize(h); if (rcm != LII_OK) return rcm; }  // (a selected frame of the last message nobody has read: the frames are about to be overwritten)
  *n_frames = 0;
  IngestRing* r = ring_of(h);
  IngestCtx* c = &r->slot[r->front];
  c->have = false;
  if (n_points == 0) return LII_OK;
  hipStream_t s = lii_internal_stream(h);
  rc = pcl2_enqueue(h, c, data, n_points, f, o, s, s, nullptr);
  if (rc != LII_OK) return rc;
  ICHK(h, hipStreamSynchronize(s));
  return ingest_collect(h, c, frames, max_frames, n_frames, s);
}

int lii_ingest_livox(lii_handle h, const void* points, int32_t n_points, const lii_livox_fields* f, const lii_ingest_opts* o,
                     lii_frame_info* frames, int32_t max_frames, int32_t* n_frames) {
  if (lii_internal_in_wait_hook(h)) return lii_internal_fail(h, LII_ERR_STATE, "lii_ingest_livox: not from lii_scan_job::while_waiting (lii_ingest_livox_begin is)");
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!frames || !n_frames) return lii_internal_fail(h, LII_ERR_INVALID, "lii_ingest_livox: bad arguments");
  int rc = livox_check(h, points, n_points, f, o);
  if (rc != LII_OK) return rc;
  { const int rcm = lii_internal_scan_materialize(h); if (rcm != LII_OK) return rcm; }
  *n_frames = 0;
  IngestRing* r = ring_of(h);
  IngestCtx* c = &r->slot[r->front];
  c->have = false;
  if (n_points == 0) return LII_OK;
  hipStream_t s = lii_internal_stream(h);
  rc = livox_enqueue(h, c, points, n_points, f, o, s, s, nullptr);
  if (rc != LII_OK) return rc;
  ICHK(h, hipStreamSynchronize(s));
  return ingest_collect(h, c, frames, max_frames, n_frames, s);
}

// The overlapped forms (ABI 9).  Neither uses the handle's stream: a registration under way - or a pre-armed launch - is not disturbed.
int lii_ingest_pcl2_begin(lii_handle h, const void* data, int32_t n_points, const lii_pc2_fields* f, const lii_ingest_opts* o) {
  lii_ingest_opts whole_opts;
  int rc = pcl2_check(h, data, n_points, f, &o, &whole_opts);
  if (rc != LII_OK) return rc;
  IngestRing* r = ring_of(h);
  int k = -1;
  rc = ring_take(h, r, &k);
  if (rc != LII_OK) return rc;
  IngestCtx* c = &r->slot[k];
  c->have = false;
  c->n_under_way = n_points;
  if (n_points > 0) {
    rc = pcl2_enqueue(h, c, data, n_points, f, o, r->s_copy, r->s_kern, r->ev_copied[k]);
    if (rc != LII_OK) return rc;
    ICHK(h, hipEventRecord(r->ev_done[k], r->s_kern));
  }
  ring_push(r, k);
  return LII_OK;
}
int lii_ingest_livox_begin(lii_handle h, const void* points, int32_t n_points, const lii_livox_fields* f, const lii_ingest_opts* o) {
  int rc = livox_check(h, points, n_points, f, o);
  if (rc != LII_OK) return rc;
  IngestRing* r = ring_of(h);
  int k = -1;
  rc = ring_take(h, r, &k);
  if (rc != LII_OK) return rc;
  IngestCtx* c = &r->slot[k];
  c->have = false;
  c->n_under_way = n_points;
  if (n_points > 0) {
    rc = livox_enqueue(h, c, points, n_points, f, o, r->s_copy, r->s_kern, r->ev_copied[k]);
    if (rc != LII_OK) return rc;
    ICHK(h, hipEventRecord(r->ev_done[k], r->s_kern));
  }
  ring_push(r, k);
  return LII_OK;
}
int lii_ingest_end(lii_handle h, lii_frame_info* frames, int32_t max_frames, int32_t* n_frames) {
  if (!h || !frames || !n_frames) return lii_internal_fail(h, LII_ERR_INVALID, "lii_ingest_end: bad arguments");
  // (inside lii_scan_job::while_waiting the registration under way may still be reading the frames this call would retire)
  if (lii_internal_in_wait_hook(h)) return lii_internal_fail(h, LII_ERR_STATE, "lii_ingest_end: not from lii_scan_job::while_waiting (only the begin functions are)");
  IngestRing* r = ring_of(h);
  if (r->n_pending < 1) return lii_internal_fail(h, LII_ERR_STATE, "lii_ingest_end: no message under way (call lii_ingest_pcl2_begin / lii_ingest_livox_begin)");
  // A selected frame of the outgoing message that nobody has read is copied into the handle's own scan buffer first - by the handle's
  // stream, which the next message that takes this context waits for.
  if (lii_internal_scan_is_deferred(h)) {
    lii_internal_prearm_cancel(h);
    const int rcm = lii_internal_scan_materialize(h);
    if (rcm != LII_OK) return rcm;
  }
  // ... and whatever else that stream has been given to do with the outgoing frames (an asynchronous reader that copied a selected frame
  // on demand) is in front of this event, which the next message to take the context waits for - long complete in the per-scan loop
  ICHK(h, hipEventRecord(r->ev_read, lii_internal_stream(h)));  // (a message is under way: the ring's streams and events exist)
  r->guard[r->front] = true;
  const int k = r->pending[0];
  IngestCtx* c = &r->slot[k];
  *n_frames = 0;
  if (c->n_under_way > 0) {
    // long done when the message overlapped a registration: asked first (a blocking wait's wake-up is tens of microseconds)
    hipError_t e = hipEventQuery(r->ev_done[k]);
    for (int spin = 0; e == hipErrorNotReady && spin < 2000; spin++) e = hipEventQuery(r->ev_done[k]);
    if (e == hipErrorNotReady) { (void)hipGetLastError(); e = hipEventSynchronize(r->ev_done[k]); }
    ICHK(h, e);
  }
  for (int p = 1; p < r->n_pending; p++) r->pending[p - 1] = r->pending[p];
  r->n_pending--;
  r->slot[r->front].have = false;
  r->front = k;
  if (c->n_under_way <= 0) return LII_OK;  // (an empty message: no frame, as in the one-call forms)
  return ingest_collect(h, c, frames, max_frames, n_frames, r->s_kern);
}

int lii_frame_select(lii_handle h, int32_t frame) {
  if (lii_internal_in_wait_hook(h)) return lii_internal_fail(h, LII_ERR_STATE, "lii_frame_select: not from lii_scan_job::while_waiting");
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h) return LII_ERR_INVALID;
  IngestRing* r = ring_of(h);
  IngestCtx* c = &r->slot[r->front];
  if (!c->have || frame < 0 || frame >= c->table.n_frames) return lii_internal_fail(h, LII_ERR_STATE, "lii_frame_select: no such frame");
  const int first = c->table.first[frame], cnt = c->table.last[frame] - first + 1;
  return lii_internal_scan_defer(h, c->d_frames + (first - 1), cnt);  // (read in place by lii_scan_register, copied by any other reader)
}

}  // extern "C"
