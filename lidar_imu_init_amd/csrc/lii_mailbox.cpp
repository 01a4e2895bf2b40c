// Host side of the node-local mailbox (lii_device.h: MailboxView / mailbox_allreduce): one POSIX shared-memory segment per
// job, mapped by every rank.  The ranks meet in it (atomics in its header - neither RCCL nor a second channel is needed; ranks
// that sit on different nodes never meet and the caller falls back to RCCL), and then set up the slots a kernel publishes its
// 91 sums in:
//   * HBM form: every rank allocates fine-grained device memory for the slots it reads, exports it as a HIP IPC handle through
//     the segment and opens the others' - a peer-mapped exchange over xGMI (a push and a local poll), also between two
//     processes on one device;
//   * host-memory form (when a rank cannot export or open a handle): the segment itself, registered with every rank's device.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include "lii_device.h"
#include "lii_launch.h"

namespace lii {

namespace {
constexpr size_t kHeaderBytes = 8192;
constexpr size_t kHandlesAt = 1024;  // n_ranks x 64 bytes: the IPC handles of the ranks' slot areas
enum : uint32_t { kPending = 0, kReady = 1, kFailed = 2 };
struct SegmentHeader {
  std::atomic<uint32_t> arrived;  // ranks that mapped + registered the segment
  std::atomic<uint32_t> state;    // kPending -> kReady (all arrived) | kFailed (somebody gave up / could not register)
  uint32_t n_ranks;
  std::atomic<uint32_t> exported, export_failed;  // HBM form: ranks that wrote their handle / that could not
  std::atomic<uint32_t> opened, open_failed;      // ... that opened all the others' / that could not
};
static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
static_assert(kHandlesAt + 64 * 64 <= kHeaderBytes, "header layout");
// waits until `counter` reaches n (false: timed out)
bool wait_count(std::atomic<uint32_t>& counter, uint32_t n, double wait_s) {
  const auto t0 = std::chrono::steady_clock::now();
  while (counter.load() < n) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_s) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
  return true;
}
static_assert(std::atomic<uint32_t>::is_always_lock_free, "segment atomics must be address-free");

unsigned long long fnv1a(const uint8_t* p, size_t n) {
  unsigned long long h = 1469598103934665603ull;
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}
}  // namespace

size_t mailbox_segment_bytes(int n_ranks) {
  size_t b = kHeaderBytes + (size_t)n_ranks * 2 * kMailboxSlotDoubles * sizeof(double);
  return (b + 4095) & ~(size_t)4095;
}

void mailbox_close(MailboxHost* m) {
  if (!m) return;
  for (int r = 0; r < m->n_peers; r++)
    if (m->peer_ptr[r] && m->peer_ptr[r] != m->own) (void)hipIpcCloseMemHandle(m->peer_ptr[r]);
  if (m->d_peers) (void)hipFree(m->d_peers);
  if (m->own) (void)hipFree(m->own);
  if (m->registered) (void)hipHostUnregister(m->map);
  if (m->map) munmap(m->map, m->bytes);
  if (m->name[0]) shm_unlink(m->name);  // normally gone already (the last arriver unlinks); harmless otherwise
  *m = MailboxHost{};
}

// Returns 0 when every rank of the job met in the segment (m is filled), 1 when they did not (different nodes, a rank that
// could not register, timeout): the caller then uses RCCL.  `why` explains a non-zero return.
int mailbox_open(const uint8_t id[128], int n_ranks, int rank, double wait_s, bool want_hbm, MailboxHost* m, std::string* why) {
  *m = MailboxHost{};
  if (n_ranks > kMailboxMaxRanks) { *why = "more ranks than mailbox lanes"; return 1; }
  std::snprintf(m->name, sizeof(m->name), "/lii_mbx_%016llx_%d", fnv1a(id, 128), n_ranks);
  const int fd = shm_open(m->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { *why = std::string("shm_open: ") + std::strerror(errno); m->name[0] = 0; return 1; }
  m->bytes = mailbox_segment_bytes(n_ranks);
  if (ftruncate(fd, (off_t)m->bytes) != 0) {  // same size from every rank; new pages are zero
    *why = std::string("ftruncate: ") + std::strerror(errno);
    close(fd);
    mailbox_close(m);
    return 1;
  }
  m->map = mmap(nullptr, m->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m->map == MAP_FAILED) { m->map = nullptr; *why = std::string("mmap: ") + std::strerror(errno); mailbox_close(m); return 1; }
  auto* hdr = reinterpret_cast<SegmentHeader*>(m->map);
  void* dev = nullptr;
  hipError_t e = hipHostRegister(m->map, m->bytes, hipHostRegisterPortable | hipHostRegisterMapped);
  if (e == hipSuccess) {
    m->registered = true;
    e = hipHostGetDevicePointer(&dev, m->map, 0);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    hdr->state.store(kFailed);  // before `arrived`: whoever sees the full count also sees the failure
    hdr->arrived.fetch_add(1);
    *why = std::string("hipHostRegister of the shared segment: ") + hipGetErrorString(e);
    mailbox_close(m);
    return 1;
  }
  hdr->n_ranks = (uint32_t)n_ranks;
  const uint32_t mine = hdr->arrived.fetch_add(1) + 1;
  if (mine == (uint32_t)n_ranks) {
    uint32_t expect = kPending;
    hdr->state.compare_exchange_strong(expect, kReady);
    shm_unlink(m->name);  // every rank has it mapped; the name can go
    m->name[0] = 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (hdr->state.load() == kPending) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > wait_s) {
      uint32_t expect = kPending;
      hdr->state.compare_exchange_strong(expect, kFailed);  // one verdict for everybody, whoever decides first
      break;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  if (hdr->state.load() != kReady) {
    *why = "not all ranks met in the node-local segment (ranks on several nodes, or a rank could not register it)";
    mailbox_close(m);
    return 1;
  }
  m->dev_slots = reinterpret_cast<double*>(reinterpret_cast<char*>(dev) + kHeaderBytes);
  if (!want_hbm) return 0;
  // ---- the HBM form on top: export the own slot area, open the others'.  Every step is collective: one rank that cannot
  // makes all of them stay with the host-memory form (the verdicts travel through the segment's counters).
  const size_t slot_bytes = (size_t)n_ranks * 2 * kMailboxSlotDoubles * sizeof(double);
  auto* handles = reinterpret_cast<hipIpcMemHandle_t*>(reinterpret_cast<char*>(m->map) + kHandlesAt);
  bool mine_ok = false;
  {
    void* p = nullptr;
    hipError_t a = hipExtMallocWithFlags(&p, slot_bytes, hipDeviceMallocFinegrained);
    if (a == hipSuccess && hipMemset(p, 0, slot_bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
      hipIpcMemHandle_t hnd;
      if (hipIpcGetMemHandle(&hnd, p) == hipSuccess) {
        std::memcpy(&handles[rank], &hnd, sizeof(hnd));
        m->own = static_cast<double*>(p);
        mine_ok = true;
      }
    }
    if (!mine_ok) { (void)hipGetLastError(); if (p) (void)hipFree(p); hdr->export_failed.fetch_add(1); }
  }
  std::atomic_thread_fence(std::memory_order_release);
  hdr->exported.fetch_add(1);
  bool hbm = wait_count(hdr->exported, (uint32_t)n_ranks, wait_s) && hdr->export_failed.load() == 0;
  if (hbm) {
    std::atomic_thread_fence(std::memory_order_acquire);
    m->n_peers = n_ranks;
    bool opened_all = true;
    for (int r = 0; r < n_ranks && opened_all; r++) {
      if (r == rank) { m->peer_ptr[r] = m->own; continue; }
      void* q = nullptr;
      hipIpcMemHandle_t hnd;
      std::memcpy(&hnd, &handles[r], sizeof(hnd));
      if (hipIpcOpenMemHandle(&q, hnd, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); opened_all = false; break; }
      m->peer_ptr[r] = static_cast<double*>(q);
    }
    if (opened_all) {
      if (hipMalloc(reinterpret_cast<void**>(&m->d_peers), sizeof(double*) * (size_t)n_ranks) != hipSuccess ||
          hipMemcpy(m->d_peers, m->peer_ptr, sizeof(double*) * (size_t)n_ranks, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        opened_all = false;
      }
    }
    if (!opened_all) hdr->open_failed.fetch_add(1);
    hdr->opened.fetch_add(1);
    hbm = wait_count(hdr->opened, (uint32_t)n_ranks, wait_s) && hdr->open_failed.load() == 0;
  }
  if (!hbm) {  // stay with the host-memory form (all ranks take this branch together)
    for (int r = 0; r < m->n_peers; r++)
      if (m->peer_ptr[r] && m->peer_ptr[r] != m->own) (void)hipIpcCloseMemHandle(m->peer_ptr[r]);
    for (int r = 0; r < 64; r++) m->peer_ptr[r] = nullptr;
    m->n_peers = 0;
    if (m->d_peers) { (void)hipFree(m->d_peers); m->d_peers = nullptr; }
    if (m->own) { (void)hipFree(m->own); m->own = nullptr; }
  }
  return 0;
}

}  // namespace lii
