// Host side of the node-local mailbox (lii_device.h: MailboxView / mailbox_allreduce): one POSIX shared-memory segment per
// job, mapped by every rank and registered with the rank's own device, so a kernel can publish its 91 sums and read the
// peers' without a collective-library launch.  The rendezvous uses only the segment itself (two atomics in its header), so
// it needs neither RCCL nor a second channel; ranks that sit on different nodes never meet in the segment and the caller
// falls back to RCCL.
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
constexpr size_t kHeaderBytes = 4096;
enum : uint32_t { kPending = 0, kReady = 1, kFailed = 2 };
struct SegmentHeader {
  std::atomic<uint32_t> arrived;  // ranks that mapped + registered the segment
  std::atomic<uint32_t> state;    // kPending -> kReady (all arrived) | kFailed (somebody gave up / could not register)
  uint32_t n_ranks;
};
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
  if (m->registered) (void)hipHostUnregister(m->map);
  if (m->map) munmap(m->map, m->bytes);
  if (m->name[0]) shm_unlink(m->name);  // normally gone already (the last arriver unlinks); harmless otherwise
  *m = MailboxHost{};
}

// Returns 0 when every rank of the job met in the segment (m is filled), 1 when they did not (different nodes, a rank that
// could not register, timeout): the caller then uses RCCL.  `why` explains a non-zero return.
int mailbox_open(const uint8_t id[128], int n_ranks, int rank, double wait_s, MailboxHost* m, std::string* why) {
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
  return 0;
}

}  // namespace lii
