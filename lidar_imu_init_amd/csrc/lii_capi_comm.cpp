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
int lists_exchange_rccl(lii_handle h, hipStream_t s) {
  const int N = h->net.n_ranks;
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
  unsigned char* send = h->net.d_gx;
  unsigned char* recv = h->net.d_gx + block;
  unsigned char* hdrs = h->net.d_gx + block * (size_t(N) + 1);
  unsigned char** tabs = reinterpret_cast<unsigned char**>(hdrs + size_t(N) * kGatherHeaderBytes);
  const unsigned long long seq = 2ull * ++h->net.gather_seq;  // (even: both views keep to parity 0)
  lii::GatherView one;
  one.peers = tabs; one.block_bytes = block; one.cap_points = h->cfg.max_scan_points; one.n_ranks = 1; one.rank = 0;
  one.timeout_ticks = h->net.mailbox_timeout_ticks;
  launch_lists_push(one, h->d_list_add, h->d_list_nodown, h->d_counts, h->net.d_gather_ticket, seq, s);
  ncclResult_t r = ncclAllGather(send, hdrs, kGatherHeaderBytes, ncclChar, h->net.comm, s);
  if (r != ncclSuccess) return fail(h, LII_ERR_COMM, std::string("ncclAllGather (list headers): ") + ncclGetErrorString(r));
  unsigned char* host_hdrs = reinterpret_cast<unsigned char*>(h->h_small + 3200);
  HIPCHK(h, hipMemcpyAsync(host_hdrs, hdrs, size_t(N) * kGatherHeaderBytes, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  long long longest = 0;
  for (int q = 0; q < N; q++) {
    unsigned long long c = 0;
    std::memcpy(&c, host_hdrs + size_t(q) * kGatherHeaderBytes + 8, 8);
    longest = std::max(longest, (long long)(unsigned int)c + (long long)(unsigned int)(c >> 32));
  }
  if (longest > h->cfg.max_scan_points) return fail(h, LII_ERR_COMM, "list exchange: a rank announced more points than a scan holds");
  const size_t trimmed = size_t(kGatherHeaderBytes) + sizeof(float4) * size_t(longest);
  r = ncclAllGather(send, recv, trimmed, ncclChar, h->net.comm, s);
  if (r != ncclSuccess) return fail(h, LII_ERR_COMM, std::string("ncclAllGather (lists): ") + ncclGetErrorString(r));
  lii::GatherView all;
  all.peers = tabs + 1; all.block_bytes = trimmed; all.cap_points = h->cfg.max_scan_points; all.n_ranks = N; all.rank = h->net.rank;
  all.timeout_ticks = h->net.mailbox_timeout_ticks;
  launch_lists_collect(all, seq, h->d_list_add, h->d_list_nodown, h->d_counts, 5, s);
  return LII_OK;
}
void comm_drop(lii_handle h) {
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->net.comm) { ncclCommDestroy(h->net.comm); h->net.comm = nullptr; }
  mailbox_close(&h->net.mailbox);
  h->net.n_ranks = 1;
  h->net.rank = 0;
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
  if (!h || library_partition < 0 || library_partition > 2) return fail(h, LII_ERR_INVALID, "lii_comm_set_partition: 0 (caller), 1 (by index) or 2 (by voxel)");
  h->net.library_partition = library_partition != 0;
  h->net.voxel_partition = library_partition == 2;
  partition_refresh(h);
  return LII_OK;
}
int lii_comm_init(lii_handle h, int32_t n_ranks, int32_t rank, const uint8_t id_in[128]) {
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
  if (!h) return LII_ERR_INVALID;
  (void)hipSetDevice(h->device);
  comm_drop(h);
  return LII_OK;
}

}  // extern "C"
