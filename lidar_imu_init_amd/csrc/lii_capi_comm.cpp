// libliinit_hip — the communicator of a sharded job: node-local mailbox (lii_mailbox.cpp; the exchange itself runs inside
// k_reduce_solve, lii_iekf.hip) or RCCL.  SURVEY.md section 8(e).
#include "lii_context.h"

using namespace lii_impl;

extern "C" {

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
}  // extern "C"
namespace lii_impl {
void partition_refresh(lii_handle h) {
  // (by voxel, a rank never sees the other ranks' points: the map update needs the list exchange, which the peer-mapped mailbox and
  // RCCL carry - a job on the host-memory mailbox stays with the split by index; lii_comm_describe says which)
  const bool by_voxel = h->net.n_ranks > 1 && h->net.library_partition && h->net.voxel_partition && (h->net.mailbox.d_gather_peers || h->net.comm);
  h->vh.part_world = by_voxel ? h->net.n_ranks : 0;
  h->vh.part_rank = by_voxel ? h->net.rank : 0;
  h->vh.part_overflow = h->h_res ? &h->h_res->part_overflow : nullptr;
  if (h->solo_share > 1 && h->net.n_ranks <= 1) { h->vh.part_world = h->solo_share; h->vh.part_rank = 0; }  // (LII_TEST=solo_share)
}
// The list exchange of a sharded job's map update over RCCL (lii_exchange.hip describes the mailbox form): pack this rank's two lists
// into one block, all-gather the 64-byte headers, learn the longest block of the job from them (one synchronising copy - the map
// update that follows reads the list sizes back anyway), all-gather the blocks trimmed to that length into a local area laid out like
// a gather area, and let k_lists_collect put the lists together in rank order, in place.
// Three steps, the collective a parameter of the last: lii_selftest_list_exchange plays the ranks of a job one after the other on ONE
// device with copies standing in for ncclAllGather - the trimmed layout with N > 1 is then exercised where RCCL cannot be (a
// communicator holds one rank per device).
namespace {
struct GxLayout {
  unsigned char *send, *recv, *hdrs;
  unsigned char** tabs;
  size_t block;
};
int gx_prepare(lii_handle h, int N, GxLayout* L) {
  const size_t block = size_t(kGatherHeaderBytes) + sizeof(float4) * size_t(h->cfg.max_scan_points);
  if (!h->net.d_gx || h->net.gx_block != block || h->net.gx_ranks != N) {
    if (h->net.d_gx) HIPCHK(h, hipFree(h->net.d_gx));
    h->net.d_gx = nullptr;
    const size_t bytes = block * (size_t(N) + 1) + size_t(N) * kGatherHeaderBytes + sizeof(void*) * (size_t(N) + 1);
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->net.d_gx), bytes));
    HIPCHK(h, hipMemset(h->net.d_gx, 0, bytes));
    h->net.gx_block = block; h->net.gx_ranks = N;
    std::vector<unsigned char*> tab(size_t(N) + 1);
    tab[0] = h->net.d_gx;                                        // the send block (a view of one rank)
    for (int r = 0; r < N; r++) tab[size_t(r) + 1] = h->net.d_gx + block;  // the gathered blocks (every entry: this rank's local area)
    HIPCHK(h, hipMemcpy(h->net.d_gx + block * (size_t(N) + 1) + size_t(N) * kGatherHeaderBytes, tab.data(), sizeof(void*) * tab.size(), hipMemcpyHostToDevice));
    if (!h->net.d_gather_ticket) {
      HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->net.d_gather_ticket), sizeof(unsigned int)));
      HIPCHK(h, hipMemset(h->net.d_gather_ticket, 0, sizeof(unsigned int)));
    }
  }
  L->block = block;
  L->send = h->net.d_gx;
  L->recv = h->net.d_gx + block;
  L->hdrs = h->net.d_gx + block * (size_t(N) + 1);
  L->tabs = reinterpret_cast<unsigned char**>(L->hdrs + size_t(N) * kGatherHeaderBytes);
  return LII_OK;
}
// this rank's lists (h->d_list_add / d_list_nodown, sizes in h->d_counts[0..1]) -> the send block
void gx_push(lii_handle h, const GxLayout& L, unsigned long long seq, hipStream_t s) {
  lii::GatherView one;
  one.peers = L.tabs; one.block_bytes = L.block; one.cap_points = h->cfg.max_scan_points; one.n_ranks = 1; one.rank = 0;
  one.timeout_ticks = h->net.mailbox_timeout_ticks;
  launch_lists_push(one, h->d_list_add, h->d_list_nodown, h->d_counts, h->net.d_gather_ticket, seq, s);
}
// gather(send, recv, bytes): every rank's first `bytes` bytes of `send`, in rank order, into `recv` (0: fine; else an error text)
template <class Gather>
int gx_gather_collect(lii_handle h, const GxLayout& L, int N, int rank, unsigned long long seq, hipStream_t s, Gather&& gather) {
  std::string e = gather(L.send, L.hdrs, size_t(kGatherHeaderBytes));
  if (!e.empty()) return fail(h, LII_ERR_COMM, "list headers: " + e);
  unsigned char* host_hdrs = reinterpret_cast<unsigned char*>(h->h_small + 3200);
  HIPCHK(h, hipMemcpyAsync(host_hdrs, L.hdrs, size_t(N) * kGatherHeaderBytes, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  long long longest = 0;
  for (int q = 0; q < N; q++) {
    unsigned long long c = 0;
    std::memcpy(&c, host_hdrs + size_t(q) * kGatherHeaderBytes + 8, 8);
    longest = std::max(longest, (long long)(unsigned int)c + (long long)(unsigned int)(c >> 32));
  }
  if (longest > h->cfg.max_scan_points) return fail(h, LII_ERR_COMM, "list exchange: a rank announced more points than a scan holds");
  const size_t trimmed = size_t(kGatherHeaderBytes) + sizeof(float4) * size_t(longest);
  e = gather(L.send, L.recv, trimmed);
  if (!e.empty()) return fail(h, LII_ERR_COMM, "lists: " + e);
  lii::GatherView all;
  all.peers = L.tabs + 1; all.block_bytes = trimmed; all.cap_points = h->cfg.max_scan_points; all.n_ranks = N; all.rank = rank;
  all.timeout_ticks = h->net.mailbox_timeout_ticks;
  launch_lists_collect(all, seq, h->d_list_add, h->d_list_nodown, h->d_counts, 5, s);
  return LII_OK;
}
}  // namespace
int lists_exchange_rccl(lii_handle h, hipStream_t s) {
  const int N = h->net.n_ranks;
  GxLayout L;
  int rc = gx_prepare(h, N, &L);
  if (rc != LII_OK) return rc;
  const unsigned long long seq = 2ull * ++h->net.gather_seq;  // (even: both views keep to parity 0)
  gx_push(h, L, seq, s);
  return gx_gather_collect(h, L, N, h->net.rank, seq, s, [&](const unsigned char* send, unsigned char* recv, size_t bytes) -> std::string {
    const ncclResult_t r = ncclAllGather(send, recv, bytes, ncclChar, h->net.comm, s);
    return r == ncclSuccess ? std::string() : std::string("ncclAllGather: ") + ncclGetErrorString(r);
  });
}
void comm_drop(lii_handle h) {
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->net.comm) { ncclCommDestroy(h->net.comm); h->net.comm = nullptr; }
  mailbox_close(&h->net.mailbox);
  h->net.n_ranks = 1;
  h->net.rank = 0;
  // The list exchange numbers its rounds, and the ranks of a job must count alike: whatever transport comes next starts from 0 on
  // EVERY rank - a handle that is re-attached beside a fresh one would otherwise carry its old count into the headers, and
  // k_lists_collect's `v >= seq` on the fresh rank's side would never see it arrive.  The RCCL form's local areas
  // hold headers of the old numbering: dropped with it (lists_exchange_rccl lays them out again, zeroed).
  h->net.gather_seq = 0;
  if (h->net.d_gx) { (void)hipFree(h->net.d_gx); h->net.d_gx = nullptr; h->net.gx_block = 0; h->net.gx_ranks = 0; }
  if (h->net.d_gather_ticket) (void)hipMemset(h->net.d_gather_ticket, 0, sizeof(unsigned int));
  partition_refresh(h);
}
}  // namespace lii_impl
namespace {
int comm_init(lii_handle h, int32_t n_ranks, int32_t rank, const uint8_t id_in[128], int32_t transport) {
  HIPCHK(h, hipSetDevice(h->device));
  comm_drop(h);
  h->net.n_ranks = n_ranks;
  h->net.rank = rank;
  // a single rank needs no exchange; asked for by name, the RCCL transport is still set up (a one-rank communicator), so that
  // the three-launch form of the loop - final sum, ncclAllReduce, solve - can be exercised on one device
  if (n_ranks == 1 && transport != LII_COMM_RCCL) return LII_OK;
  if (transport != LII_COMM_RCCL) {
    if (!h->net.d_mb_seq) HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->net.d_mb_seq), sizeof(unsigned long long)));
    HIPCHK(h, hipMemset(h->net.d_mb_seq, 0, sizeof(unsigned long long)));
    if (!h->net.d_gather_ticket) HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->net.d_gather_ticket), sizeof(unsigned int)));
    HIPCHK(h, hipMemset(h->net.d_gather_ticket, 0, sizeof(unsigned int)));
    h->net.gather_seq = 0;
    // LII_MAILBOX_TIMEOUT_S=<exchange>[,<set-up>]: how long a reduce+solve kernel waits for a peer's sums (30 s), how long this
    // call waits for all ranks in the node-local segment (20 s)
    double wait_s = 20.0;
    if (const char* t = std::getenv("LII_MAILBOX_TIMEOUT_S")) {
      h->net.mailbox_timeout_ticks = (long long)(std::atof(t) * 1e8);
      if (const char* c = std::strchr(t, ',')) wait_s = std::atof(c + 1);
    }
    std::string why;
    if (mailbox_open(id_in, n_ranks, rank, wait_s, transport != LII_COMM_MAILBOX_HOST, h->no_gather ? 0 : h->cfg.max_scan_points, &h->net.mailbox, &why) == 0) {
      if (transport == LII_COMM_MAILBOX && !h->net.mailbox.d_peers) {  // asked for by name: no silent change of the transport
        mailbox_close(&h->net.mailbox);
        h->net.n_ranks = 1; h->net.rank = 0;
        return fail(h, LII_ERR_COMM, "peer-mapped HBM mailbox unavailable: " + why);
      }
      h->net.comm_why = h->net.mailbox.d_peers ? "mailbox in peer-mapped HBM (HIP IPC; every rank's device reaches every other's)"
                                       : (transport == LII_COMM_MAILBOX_HOST ? std::string("mailbox in registered host memory (asked for)")
                                                                             : "mailbox in registered host memory - the HBM form was not possible: " + why);
      if (h->diag) std::fprintf(stderr, "[libliinit_hip] rank %d of %d: %s\n", rank, n_ranks, h->net.comm_why.c_str());
      return LII_OK;
    }
    if (transport == LII_COMM_MAILBOX || transport == LII_COMM_MAILBOX_HOST) {
      h->net.n_ranks = 1; h->net.rank = 0;
      return fail(h, LII_ERR_COMM, "node-local mailbox unavailable: " + why);
    }
    h->net.comm_why = "RCCL - the node-local mailbox was not possible: " + why;
  } else {
    h->net.comm_why = "RCCL (asked for)";
  }
  ncclUniqueId id;
  std::memcpy(&id, id_in, 128);
  ncclResult_t r = ncclCommInitRank(&h->net.comm, n_ranks, id, rank);
  if (r != ncclSuccess) {
    h->net.comm = nullptr; h->net.n_ranks = 1; h->net.rank = 0;
    return fail(h, LII_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  if (h->diag) std::fprintf(stderr, "[libliinit_hip] rank %d of %d: %s\n", rank, n_ranks, h->net.comm_why.c_str());
  return LII_OK;
}
}  // namespace
extern "C" {
int lii_comm_init_ex(lii_handle h, int32_t n_ranks, int32_t rank, const uint8_t id_in[128], int32_t transport) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || !id_in || n_ranks < 1 || rank < 0 || rank >= n_ranks || transport < LII_COMM_AUTO || transport > LII_COMM_MAILBOX_HOST)
    return fail(h, LII_ERR_INVALID, "lii_comm_init: bad arguments");
  const int rc = comm_init(h, n_ranks, rank, id_in, transport);
  partition_refresh(h);  // (n_ranks / rank as the set-up left them: 1 / 0 after a failure)
  return rc;
}
int lii_comm_describe(lii_handle h, char* out, int32_t capacity) {
  if (!h || !out || capacity < 1) return LII_ERR_INVALID;
  std::string s = (h->net.comm || h->net.mailbox.dev_slots || h->net.mailbox.d_peers) ? h->net.comm_why : std::string("no communicator");
  if (h->net.n_ranks > 1) {
    s += !h->net.library_partition ? "; the caller splits the cloud"
         : (h->vh.part_world > 1 ? "; cloud split by voxel (fused filter) / by index, map lists exchanged"
                                 : ((h->net.mailbox.d_gather_peers || h->net.comm) ? "; cloud split by index, map lists exchanged" : "; cloud split by index, map update repeats the search"));
    if (h->net.voxel_partition && h->vh.part_world <= 1) s += " (the split by voxel needs the peer-mapped mailbox or RCCL)";
  }
  std::snprintf(out, size_t(capacity), "%s", s.c_str());
  return LII_OK;
}
int lii_comm_set_partition(lii_handle h, int32_t library_partition) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || library_partition < 0 || library_partition > 2) return fail(h, LII_ERR_INVALID, "lii_comm_set_partition: 0 (caller), 1 (by index) or 2 (by voxel)");
  h->net.library_partition = library_partition != 0;
  h->net.voxel_partition = library_partition == 2;
  partition_refresh(h);
  return LII_OK;
}
int lii_comm_init(lii_handle h, int32_t n_ranks, int32_t rank, const uint8_t id_in[128]) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  return lii_comm_init_ex(h, n_ranks, rank, id_in, LII_COMM_AUTO);
}
int lii_comm_transport(lii_handle h, int32_t* transport) {
  if (!h || !transport) return LII_ERR_INVALID;
  *transport = h->net.comm ? LII_COMM_RCCL : (h->net.mailbox.d_peers ? LII_COMM_MAILBOX : (h->net.mailbox.dev_slots ? LII_COMM_MAILBOX_HOST : LII_COMM_AUTO));
  return LII_OK;
}
int lii_comm_rccl_ranks(lii_handle h, int32_t* n_ranks) {
  if (!h || !n_ranks) return LII_ERR_INVALID;
  *n_ranks = 0;
  if (h->net.comm) {
    int n = 0;
    const ncclResult_t r = ncclCommCount(h->net.comm, &n);
    if (r != ncclSuccess) return fail(h, LII_ERR_COMM, std::string("ncclCommCount: ") + ncclGetErrorString(r));
    *n_ranks = n;
  }
  return LII_OK;
}
int lii_comm_destroy(lii_handle h) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h) return LII_ERR_INVALID;
  (void)hipSetDevice(h->device);
  comm_drop(h);
  return LII_OK;
}

// Self-test of the list exchange's two layouts with SEVERAL ranks on one device (liinit_hip.h).  The handle plays rank 0 .. n_ranks - 1
// in turn: form 0 - the gather areas of the mailbox transport (every rank's push stores into every rank's area, every rank collects from
// its own); form 1 - the trimmed all-gather layout of the RCCL transport, through the very functions lists_exchange_rccl is made of,
// device copies standing in for the two ncclAllGather calls.  Every rank must end with the identical pair of lists; rank 0's is returned.
int lii_selftest_list_exchange(lii_handle h, int32_t n_ranks, int32_t form, const float* add_xyzw, const int32_t* n_add, const float* nodown_xyzw,
                               const int32_t* n_nodown, float* out_add, int32_t* out_n_add, float* out_nodown, int32_t* out_n_nodown, int32_t capacity) {
  lii_internal_prearm_cancel(h);  // (a pre-armed de-skew launch waiting on the stream is told to end: this entry point uses the stream)
  if (!h || n_ranks < 1 || n_ranks > kMailboxMaxRanks || (form != 0 && form != 1) || !n_add || !n_nodown || !out_add || !out_n_add || !out_nodown || !out_n_nodown)
    return fail(h, LII_ERR_INVALID, "lii_selftest_list_exchange: bad arguments");
  if (h->net.n_ranks > 1 || h->net.comm) return fail(h, LII_ERR_STATE, "lii_selftest_list_exchange: the handle is a rank of a job");
  // TEST ENTRY POINT: it plays its ranks in the handle's own list buffers (d_list_add / d_list_nodown / d_counts).  A map update whose
  // lists have not been consumed yet would be clobbered without a trace - refused; the sequence number and the all-gather
  // scratch it uses are put back / released on every way out.
  if (h->map_async || h->map_dirty || h->lists_predicted || h->map_enqueued_early)
    return fail(h, LII_ERR_STATE, "lii_selftest_list_exchange: a map update is pending on this handle (call lii_map_commit first)");
  const unsigned long long gather_seq_at_entry = h->net.gather_seq;
  struct Restore {
    lii_handle h; unsigned long long seq;
    ~Restore() {
      h->net.gather_seq = seq;
      if (h->net.d_gx) { (void)hipFree(h->net.d_gx); h->net.d_gx = nullptr; h->net.gx_block = 0; h->net.gx_ranks = 0; }
    }
  } restore{h, gather_seq_at_entry};
  const int N = n_ranks, cap = h->cfg.max_scan_points;
  std::vector<size_t> at_a(size_t(N) + 1, 0), at_n(size_t(N) + 1, 0);
  for (int r = 0; r < N; r++) {
    if (n_add[r] < 0 || n_nodown[r] < 0 || n_add[r] + n_nodown[r] > cap) return fail(h, LII_ERR_CAPACITY, "lii_selftest_list_exchange: a rank's lists exceed max_scan_points");
    at_a[size_t(r) + 1] = at_a[size_t(r)] + size_t(n_add[r]);
    at_n[size_t(r) + 1] = at_n[size_t(r)] + size_t(n_nodown[r]);
  }
  if (at_a[size_t(N)] > size_t(cap) || at_n[size_t(N)] > size_t(cap) || at_a[size_t(N)] > size_t(capacity) || at_n[size_t(N)] > size_t(capacity))
    return fail(h, LII_ERR_CAPACITY, "lii_selftest_list_exchange: the joined lists exceed max_scan_points or the output capacity");
  HIPCHK(h, hipSetDevice(h->device));
  int rc = map_join(h);
  if (rc != LII_OK) return rc;
  hipStream_t s = h->stream;
  HIPCHK(h, hipStreamSynchronize(s));
  const size_t block = size_t(kGatherHeaderBytes) + sizeof(float4) * size_t(cap);
  std::vector<unsigned char*> bufs;  // temporaries of this call
  auto cleanup = [&]() { for (unsigned char* b : bufs) (void)hipFree(b); };
  auto upload_rank = [&](int r) -> int {  // rank r's lists and sizes into the handle's list buffers
    const int c[6] = {n_add[r], n_nodown[r], 0, 0, 0, 0};
    if (n_add[r]) HIPCHK(h, hipMemcpyAsync(h->d_list_add, add_xyzw + 4 * at_a[size_t(r)], sizeof(float4) * size_t(n_add[r]), hipMemcpyHostToDevice, s));
    if (n_nodown[r]) HIPCHK(h, hipMemcpyAsync(h->d_list_nodown, nodown_xyzw + 4 * at_n[size_t(r)], sizeof(float4) * size_t(n_nodown[r]), hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->d_counts, c, sizeof(c), hipMemcpyHostToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return LII_OK;
  };
  std::vector<float> got_a, got_n, ref_a, ref_n;
  int ref_na = -1, ref_nn = -1;
  auto download_and_compare = [&](int q) -> int {
    int c[6];
    HIPCHK(h, hipMemcpyAsync(c, h->d_counts, sizeof(c), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (c[5]) return fail(h, LII_ERR_COMM, "lii_selftest_list_exchange: the collect timed out");
    if (c[0] < 0 || c[1] < 0 || c[0] > cap || c[1] > cap) return fail(h, LII_ERR_HIP, "lii_selftest_list_exchange: sizes out of range");
    got_a.assign(4 * size_t(c[0]), 0.f); got_n.assign(4 * size_t(c[1]), 0.f);
    if (c[0]) HIPCHK(h, hipMemcpy(got_a.data(), h->d_list_add, sizeof(float4) * size_t(c[0]), hipMemcpyDeviceToHost));
    if (c[1]) HIPCHK(h, hipMemcpy(got_n.data(), h->d_list_nodown, sizeof(float4) * size_t(c[1]), hipMemcpyDeviceToHost));
    if (q == 0) { ref_a = got_a; ref_n = got_n; ref_na = c[0]; ref_nn = c[1]; return LII_OK; }
    if (c[0] != ref_na || c[1] != ref_nn || std::memcmp(got_a.data(), ref_a.data(), sizeof(float) * got_a.size()) != 0 ||
        std::memcmp(got_n.data(), ref_n.data(), sizeof(float) * got_n.size()) != 0)
      return fail(h, LII_ERR_COMM, "lii_selftest_list_exchange: rank " + std::to_string(q) + " put together other lists than rank 0");
    return LII_OK;
  };
  if (!h->net.d_gather_ticket) {
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->net.d_gather_ticket), sizeof(unsigned int)));
    HIPCHK(h, hipMemset(h->net.d_gather_ticket, 0, sizeof(unsigned int)));
  }
  const unsigned long long seq = form == 0 ? ++h->net.gather_seq : 2ull * ++h->net.gather_seq;
  if (form == 0) {
    // N gather areas (two parities x N source blocks each) and the table of them, as mailbox_open lays them out
    const size_t area = 2 * size_t(N) * block;
    unsigned char* base = nullptr;
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&base), area * size_t(N) + sizeof(void*) * size_t(N)));
    bufs.push_back(base);
    if (hipMemset(base, 0, area * size_t(N)) != hipSuccess) { cleanup(); return fail(h, LII_ERR_HIP, "hipMemset"); }
    std::vector<unsigned char*> tab(static_cast<size_t>(N));
    for (int r = 0; r < N; r++) tab[size_t(r)] = base + area * size_t(r);
    unsigned char** d_tab = reinterpret_cast<unsigned char**>(base + area * size_t(N));
    if (hipMemcpy(d_tab, tab.data(), sizeof(void*) * size_t(N), hipMemcpyHostToDevice) != hipSuccess) { cleanup(); return fail(h, LII_ERR_HIP, "hipMemcpy"); }
    lii::GatherView gv;
    gv.peers = d_tab; gv.block_bytes = block; gv.cap_points = cap; gv.n_ranks = N; gv.timeout_ticks = 200000000ll;  // 2 s
    for (int r = 0; r < N && rc == LII_OK; r++) {
      rc = upload_rank(r);
      gv.rank = r;
      if (rc == LII_OK) launch_lists_push(gv, h->d_list_add, h->d_list_nodown, h->d_counts, h->net.d_gather_ticket, seq, s);
      if (rc == LII_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(h, LII_ERR_HIP, "lists push");
    }
    for (int q = 0; q < N && rc == LII_OK; q++) {
      gv.rank = q;
      launch_lists_collect(gv, seq, h->d_list_add, h->d_list_nodown, h->d_counts, 5, s);
      rc = download_and_compare(q);
    }
  } else {
    GxLayout L;
    rc = gx_prepare(h, N, &L);
    unsigned char* sends = nullptr;  // every rank's send block, kept while the others are played
    if (rc == LII_OK) {
      if (hipMalloc(reinterpret_cast<void**>(&sends), block * size_t(N)) != hipSuccess) rc = fail(h, LII_ERR_HIP, "hipMalloc");
      else bufs.push_back(sends);
    }
    for (int r = 0; r < N && rc == LII_OK; r++) {
      rc = upload_rank(r);
      if (rc != LII_OK) break;
      gx_push(h, L, seq, s);
      if (hipMemcpyAsync(sends + block * size_t(r), L.send, block, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        rc = fail(h, LII_ERR_HIP, "lists push (send block)");
    }
    for (int q = 0; q < N && rc == LII_OK; q++) {
      // rank q's view: its own send block in place, the all-gathers deliver the first `bytes` bytes of every rank's block in rank order
      if (hipMemcpyAsync(L.send, sends + block * size_t(q), block, hipMemcpyDeviceToDevice, s) != hipSuccess) { rc = fail(h, LII_ERR_HIP, "hipMemcpyAsync"); break; }
      if (hipMemsetAsync(L.recv, 0xEE, block * size_t(N), s) != hipSuccess) { rc = fail(h, LII_ERR_HIP, "hipMemsetAsync"); break; }  // (nothing stale may pass for data)
      rc = gx_gather_collect(h, L, N, q, seq, s, [&](const unsigned char*, unsigned char* recv, size_t bytes) -> std::string {
        for (int r = 0; r < N; r++)
          if (hipMemcpyAsync(recv + bytes * size_t(r), sends + block * size_t(r), bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return "copy standing in for the all-gather failed";
        return std::string();
      });
      if (rc == LII_OK) rc = download_and_compare(q);
    }
  }
  cleanup();
  if (rc != LII_OK) return rc;
  *out_n_add = ref_na; *out_n_nodown = ref_nn;
  if (ref_na) std::memcpy(out_add, ref_a.data(), sizeof(float) * ref_a.size());
  if (ref_nn) std::memcpy(out_nodown, ref_n.data(), sizeof(float) * ref_n.size());
  return LII_OK;
}

}  // extern "C"
