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
constexpr size_t kBusIdsAt = 5120;   // n_ranks x 32 bytes: PCI bus id of every rank's device (peer-access check)
enum : uint32_t { kPending = 0, kReady = 1, kFailed = 2 };
enum : uint32_t { kHbmUndecided = 0, kHbmYes = 1, kHbmNo = 2 };
struct SegmentHeader {
  std::atomic<uint32_t> arrived;  // ranks that mapped + registered the segment
  std::atomic<uint32_t> state;    // kPending -> kReady (all arrived) | kFailed (somebody gave up / could not register)
  uint32_t n_ranks;
  std::atomic<uint32_t> exported, export_failed;  // HBM form: ranks that wrote their handle / that could not
  std::atomic<uint32_t> opened, open_failed;      // ... that opened all the others' / that could not
  std::atomic<uint32_t> hbm_verdict;              // ONE decision for the job: kHbmYes | kHbmNo, set once by whoever decides first
};
static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
static_assert(kHandlesAt + 64 * 64 <= kBusIdsAt && kBusIdsAt + 64 * 32 <= kHeaderBytes, "header layout");
// waits until `counter` reaches n (false: timed out, or - give_up != nullptr - somebody has decided against the HBM form meanwhile)
bool wait_count(std::atomic<uint32_t>& counter, uint32_t n, double wait_s, const std::atomic<uint32_t>* give_up = nullptr) {
  const auto t0 = std::chrono::steady_clock::now();
  while (counter.load() < n) {
    if (give_up && give_up->load() == 2u) return false;
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
  if (m->d_gather_peers) (void)hipFree(m->d_gather_peers);
  if (m->own) (void)hipFree(m->own);
  if (m->registered) (void)hipHostUnregister(m->map);
  if (m->map) munmap(m->map, m->bytes);
  if (m->name[0]) shm_unlink(m->name);  // normally gone already (the last arriver unlinks); harmless otherwise
  *m = MailboxHost{};
}

// Returns 0 when every rank of the job met in the segment (m is filled), 1 when they did not (different nodes, a rank that
// could not register, timeout): the caller then uses RCCL.  `why` explains a non-zero return - and, with a zero return, why the
// HBM form was asked for but the host-memory form was taken (empty otherwise).
int mailbox_open(const uint8_t id[128], int n_ranks, int rank, double wait_s, bool want_hbm, int gather_points, MailboxHost* m, std::string* why) {
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
  if (why) why->clear();
  if (!want_hbm) return 0;
  // ---- the HBM form on top: export the own slot area, open the others'.  Every step is collective: one rank that cannot
  // makes all of them stay with the host-memory form (the verdicts travel through the segment's counters).
  // one allocation, one handle per rank: the slots, then (gather_points > 0) the gather areas of the list exchange
  const size_t slot_region = ((size_t)n_ranks * 2 * kMailboxSlotDoubles * sizeof(double) + 4095) & ~(size_t)4095;
  const size_t gather_block = gather_points > 0 ? (size_t)kGatherHeaderBytes + sizeof(float4) * (size_t)gather_points : 0;
  const size_t slot_bytes = slot_region + 2 * (size_t)n_ranks * gather_block;
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
  {  // where this rank's device sits: the peers check (and enable) access to it before they open its handle
    int dev_now = 0;
    char* bus = reinterpret_cast<char*>(m->map) + kBusIdsAt + 32 * (size_t)rank;
    if (hipGetDevice(&dev_now) != hipSuccess || hipDeviceGetPCIBusId(bus, 32, dev_now) != hipSuccess) { (void)hipGetLastError(); bus[0] = 0; }
  }
  std::atomic_thread_fence(std::memory_order_release);
  hdr->exported.fetch_add(1);
  // The verdict HBM / host memory is ONE word of the segment, written once: a rank that gives up - a timeout, a
  // handle it cannot export or open - tries to set it to "no", a rank that sees every rank through tries to set it to "yes", and
  // everybody takes what the word says afterwards.  (Counters alone let a rank that timed out in a wait leave for the host form
  // while the others, seeing the full count a moment later, stayed with the HBM form.)
  auto decide = [&](uint32_t want) {
    uint32_t expect = kHbmUndecided;
    hdr->hbm_verdict.compare_exchange_strong(expect, want);
    return hdr->hbm_verdict.load() == kHbmYes;
  };
  std::string hbm_why;
  bool hbm = wait_count(hdr->exported, (uint32_t)n_ranks, wait_s, &hdr->hbm_verdict) && hdr->export_failed.load() == 0;
  if (!hbm) { hbm_why = hdr->export_failed.load() ? "a rank could not allocate / export its slot area" : "timed out waiting for the ranks' IPC handles"; (void)decide(kHbmNo); }
  if (hbm) {
    std::atomic_thread_fence(std::memory_order_acquire);
    m->n_peers = n_ranks;
    bool opened_all = true;
    int my_dev = 0;
    (void)hipGetDevice(&my_dev);
    for (int r = 0; r < n_ranks && opened_all; r++) {
      if (r == rank) { m->peer_ptr[r] = m->own; continue; }
      // peer access from this rank's device to the device that holds rank r's slots (the same device in a one-device rehearsal)
      const char* bus = reinterpret_cast<const char*>(m->map) + kBusIdsAt + 32 * (size_t)r;
      int peer_dev = -1;
      if (bus[0] && hipDeviceGetByPCIBusId(&peer_dev, bus) == hipSuccess && peer_dev != my_dev) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, my_dev, peer_dev) != hipSuccess || !can) {
          (void)hipGetLastError();
          hbm_why = std::string("device ") + std::to_string(my_dev) + " cannot access its peer " + bus + " (hipDeviceCanAccessPeer)";
          opened_all = false;
          break;
        }
        const hipError_t pe = hipDeviceEnablePeerAccess(peer_dev, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
          hbm_why = std::string("hipDeviceEnablePeerAccess(") + bus + "): " + hipGetErrorString(pe);
          (void)hipGetLastError();
          opened_all = false;
          break;
        }
        (void)hipGetLastError();
      } else {
        (void)hipGetLastError();  // (a bus id this process cannot resolve - another visibility mask -: the open below decides)
      }
      void* q = nullptr;
      hipIpcMemHandle_t hnd;
      std::memcpy(&hnd, &handles[r], sizeof(hnd));
      const hipError_t oe = hipIpcOpenMemHandle(&q, hnd, hipIpcMemLazyEnablePeerAccess);
      if (oe != hipSuccess) {
        hbm_why = std::string("hipIpcOpenMemHandle of rank ") + std::to_string(r) + "'s slots: " + hipGetErrorString(oe);
        (void)hipGetLastError();
        opened_all = false;
        break;
      }
      m->peer_ptr[r] = static_cast<double*>(q);
    }
    if (opened_all) {
      if (hipMalloc(reinterpret_cast<void**>(&m->d_peers), sizeof(double*) * (size_t)n_ranks) != hipSuccess ||
          hipMemcpy(m->d_peers, m->peer_ptr, sizeof(double*) * (size_t)n_ranks, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        hbm_why = "no device memory for the table of peer pointers";
        opened_all = false;
      }
    }
    if (opened_all && gather_block) {
      unsigned char* g[64];
      for (int r = 0; r < n_ranks; r++) g[r] = reinterpret_cast<unsigned char*>(m->peer_ptr[r]) + slot_region;
      if (hipMalloc(reinterpret_cast<void**>(&m->d_gather_peers), sizeof(unsigned char*) * (size_t)n_ranks) != hipSuccess ||
          hipMemcpy(m->d_gather_peers, g, sizeof(unsigned char*) * (size_t)n_ranks, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        hbm_why = "no device memory for the table of gather areas";
        opened_all = false;
      }
      m->gather_block = gather_block;
      m->gather_cap = gather_points;
    }
    if (!opened_all) { hdr->open_failed.fetch_add(1); (void)decide(kHbmNo); }
    hdr->opened.fetch_add(1);
    const bool all_through = wait_count(hdr->opened, (uint32_t)n_ranks, wait_s, &hdr->hbm_verdict) && hdr->open_failed.load() == 0;
    if (!all_through && hbm_why.empty()) hbm_why = hdr->open_failed.load() ? "another rank could not open the IPC handles" : "timed out waiting for the ranks to open the IPC handles";
    hbm = decide(all_through ? kHbmYes : kHbmNo);
    if (!hbm && hbm_why.empty()) hbm_why = "another rank decided for the host-memory form";
  } else {
    hbm = decide(kHbmNo);  // (kHbmNo is already there)
  }
  if (why) *why = hbm ? "" : hbm_why;
  if (!hbm) {  // stay with the host-memory form (all ranks take this branch together)
    for (int r = 0; r < m->n_peers; r++)
      if (m->peer_ptr[r] && m->peer_ptr[r] != m->own) (void)hipIpcCloseMemHandle(m->peer_ptr[r]);
    for (int r = 0; r < 64; r++) m->peer_ptr[r] = nullptr;
    m->n_peers = 0;
    if (m->d_peers) { (void)hipFree(m->d_peers); m->d_peers = nullptr; }
    if (m->d_gather_peers) { (void)hipFree(m->d_gather_peers); m->d_gather_peers = nullptr; }
    m->gather_block = 0; m->gather_cap = 0;
    if (m->own) { (void)hipFree(m->own); m->own = nullptr; }
  }
  return 0;
}

}  // namespace lii
