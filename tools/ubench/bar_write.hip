// Can the host store straight into device memory (large BAR)?  Allocation forms tried: hipExtMallocWithFlags(hipDeviceMallocFinegrained),
// hipExtMallocWithFlags(hipDeviceMallocUncached), plain hipMalloc.  For each: pointer attributes, a host store (in a child process - a
// fault must not take the probe down), a kernel that reads the word back, and the latency host store -> device sees it (a kernel polls).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/bar_write.hip -o /tmp/bar_write
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <csetjmp>
#include <csignal>
#include <unistd.h>
static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

__global__ void k_read(const unsigned long long* p, unsigned long long* out) { *out = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// polls p[0] until it equals `want`, then stamps the wall clock into mapped host memory
__global__ void k_poll(const unsigned long long* p, unsigned long long want, unsigned long long* host_seen) {
  unsigned int spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
    if (++spins > 200000000u) { *host_seen = 2; return; }
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(host_seen, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static int try_form(const char* name, int form) {
  unsigned long long* d = nullptr;
  hipError_t e = hipSuccess;
  if (form == 0) e = hipExtMallocWithFlags((void**)&d, 4096, hipDeviceMallocFinegrained);
  else if (form == 1) e = hipExtMallocWithFlags((void**)&d, 4096, hipDeviceMallocUncached);
  else e = hipMalloc((void**)&d, 4096);
  if (e != hipSuccess) { printf("%s: allocation failed: %s\n", name, hipGetErrorString(e)); return 1; }
  hipMemset(d, 0, 4096);
  hipDeviceSynchronize();
  unsigned long long *out = nullptr, *seen = nullptr;
  hipHostMalloc((void**)&out, 64, hipHostMallocMapped);
  hipHostMalloc((void**)&seen, 64, hipHostMallocMapped);
  fflush(stdout);
  // (the store is tried in THIS process behind a SIGSEGV handler: a forked child does not inherit the driver's mappings)
  signal(SIGSEGV, on_segv);
  signal(SIGBUS, on_segv);
  if (sigsetjmp(g_jmp, 1) != 0) { printf("%s: a host store faults\n", name); signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL); return 1; }
  *(volatile unsigned long long*)d = 0x1234ull;
  signal(SIGSEGV, SIG_DFL);
  signal(SIGBUS, SIG_DFL);
  *(volatile unsigned long long*)d = 0xABCDull;
  __sync_synchronize();
  *out = 0;
  hipLaunchKernelGGL(k_read, dim3(1), dim3(1), 0, 0, d, out);
  hipDeviceSynchronize();
  printf("%s: host store works, the device reads back %llx\n", name, *out);
  // latency: a polling kernel, the host stores, the kernel answers into mapped host memory
  double best = 1e9, sum = 0;
  const int reps = 50;
  for (int r = 0; r < reps; r++) {
    *seen = 0;
    const unsigned long long want = 0x1000ull + r;
    hipLaunchKernelGGL(k_poll, dim3(1), dim3(1), 0, 0, d, want, seen);
    usleep(200);  // (the kernel is polling by now)
    const auto t0 = std::chrono::steady_clock::now();
    *(volatile unsigned long long*)d = want;
    __sync_synchronize();
    while (*(volatile unsigned long long*)seen == 0) {}
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    hipDeviceSynchronize();
    if (r >= 5) { sum += us; if (us < best) best = us; }
  }
  printf("%s: host store -> polling kernel sees it -> its answer reaches the host: mean %.2f us, best %.2f us\n", name, sum / (reps - 5), best);
  // the same round trip with the word in mapped HOST memory (what a gate polling over PCIe pays)
  unsigned long long* hw = nullptr;
  hipHostMalloc((void**)&hw, 64, hipHostMallocMapped);
  *hw = 0; sum = 0; best = 1e9;
  for (int r = 0; r < reps; r++) {
    *seen = 0;
    const unsigned long long want = 0x2000ull + r;
    hipLaunchKernelGGL(k_poll, dim3(1), dim3(1), 0, 0, hw, want, seen);
    usleep(200);
    const auto t0 = std::chrono::steady_clock::now();
    *(volatile unsigned long long*)hw = want;
    __sync_synchronize();
    while (*(volatile unsigned long long*)seen == 0) {}
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    hipDeviceSynchronize();
    if (r >= 5) { sum += us; if (us < best) best = us; }
  }
  printf("%s: ... with the word in mapped host memory instead: mean %.2f us, best %.2f us\n", name, sum / (reps - 5), best);
  return 0;
}
int main() {
  try_form("fine-grained device memory", 0);
  try_form("uncached device memory", 1);
  try_form("plain hipMalloc", 2);
  return 0;
}
