// libliinit_hip — the list exchange of a sharded job's map update (lii_map_incremental; GatherView in lii_device.h).
//
// the ranks of a job hold replicas of the map, and every replica must receive the identical batch - what
// map_incremental (src/laserMapping.cpp:516-559) decides per point of the down-sampled cloud.  Every rank decides for ITS points
// (the block or the voxels it registered: it holds their neighbour lists) and the two lists travel: a few thousand float4 per
// rank and scan, pushed into every rank's gather area with remote stores (peer-mapped fine-grained HBM, xGMI between devices)
// and put together in rank order on arrival.  Round 3 exchanged nothing and repeated the search for the whole cloud on every rank.
//
//   k_lists_push     every workgroup copies a stretch of {add list, no-down-sample list} into the block (parity, this rank) of
//                    EVERY rank's area; the workgroup that finishes last (a ticket) writes the headers and then the flags
//   k_lists_collect  waits for the flags of all source ranks in the OWN area, adds up the sizes in rank order and copies the
//                    payloads behind each other; a source that does not deliver in time raises counts[err_at]
// A job on the RCCL transport runs the same two kernels around two ncclAllGather calls (lii_capi_comm.cpp: lists_exchange_rccl): the
// push packs this rank's block into a send buffer (a view of one rank), the blocks - trimmed to the longest list of the job, which
// a first all-gather of the 64-byte headers tells - are gathered into a local area laid out like a gather area, the collect reads that.
#include <hip/hip_runtime.h>

#include "lii_device.h"
#include "lii_launch.h"

namespace lii {

namespace {

__device__ __forceinline__ void store_sys(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long load_sys(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_lists_push(GatherView gv, const float4* __restrict__ src_add, const float4* __restrict__ src_nodown,
                                                    const int* __restrict__ counts, unsigned int* __restrict__ ticket, unsigned long long seq) {
  const int na = min(max(counts[0], 0), gv.cap_points), nn = min(max(counts[1], 0), gv.cap_points - na);
  const int total = na + nn;
  const size_t block_at = ((size_t)(seq & 1ull) * gv.n_ranks + gv.rank) * gv.block_bytes;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const float4 v = i < na ? src_add[i] : src_nodown[i - na];
    const unsigned long long lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
    const unsigned long long hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
    for (int r = 0; r < gv.n_ranks; r++) {
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(gv.peers[r] + block_at + kGatherHeaderBytes) + 2 * (size_t)i;
      store_sys(dst, lo);
      store_sys(dst + 1, hi);
    }
  }
  __threadfence_system();  // this workgroup's stores are complete everywhere before its ticket counts
  __syncthreads();
  __shared__ unsigned int s_last;
  if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  // the last workgroup: every payload store of the launch has been fenced; headers, then flags
  if (threadIdx.x == 0) *ticket = 0u;
  if ((int)threadIdx.x < gv.n_ranks) {
    unsigned long long* hd = reinterpret_cast<unsigned long long*>(gv.peers[threadIdx.x] + block_at);
    store_sys(hd + 1, ((unsigned long long)(unsigned int)nn << 32) | (unsigned int)na);
    __threadfence_system();
    __hip_atomic_store(hd, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(256) void k_lists_collect(GatherView gv, unsigned long long seq, float4* __restrict__ dst_add,
                                                       float4* __restrict__ dst_nodown, int* __restrict__ counts, int err_at) {
  __shared__ int s_na[kMailboxMaxRanks], s_nn[kMailboxMaxRanks], s_oa[kMailboxMaxRanks + 1], s_on[kMailboxMaxRanks + 1];
  __shared__ int s_ok;
  const unsigned char* own = gv.peers[gv.rank];
  const size_t par_at = (size_t)(seq & 1ull) * gv.n_ranks * gv.block_bytes;
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    const unsigned long long* hd = reinterpret_cast<const unsigned long long*>(own + par_at + (size_t)(l < gv.n_ranks ? l : 0) * gv.block_bytes);
    const long long t0 = wall_clock64();
    bool ok = true;
    for (;;) {
      unsigned long long v = seq;
      if (l < gv.n_ranks) v = __hip_atomic_load(hd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (__all(v >= seq)) break;
      if (wall_clock64() - t0 > gv.timeout_ticks) { ok = false; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    if (l < gv.n_ranks) {
      const unsigned long long c = ok ? load_sys(hd + 1) : 0ull;
      s_na[l] = (int)(unsigned int)c;
      s_nn[l] = (int)(unsigned int)(c >> 32);
    }
    if (l == 0) s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = 0, n = 0;
    for (int r = 0; r < gv.n_ranks; r++) { s_oa[r] = a; s_on[r] = n; a += s_na[r]; n += s_nn[r]; }
    s_oa[gv.n_ranks] = a; s_on[gv.n_ranks] = n;
    if (blockIdx.x == 0) {
      counts[0] = s_ok ? a : 0;
      counts[1] = s_ok ? n : 0;
      counts[err_at] = s_ok ? 0 : 1;
    }
  }
  __syncthreads();
  if (!s_ok) return;
  // the lists together hold at most the whole down-sampled cloud of the scan (<= the destination's capacity); a header that says
  // otherwise is not copied past it
  for (int r = 0; r < gv.n_ranks; r++) {
    const unsigned long long* pay = reinterpret_cast<const unsigned long long*>(own + par_at + (size_t)r * gv.block_bytes + kGatherHeaderBytes);
    const int na = s_na[r], tot = na + s_nn[r];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += gridDim.x * blockDim.x) {
      const unsigned long long lo = load_sys(pay + 2 * (size_t)i), hi = load_sys(pay + 2 * (size_t)i + 1);
      const float4 v = make_float4(__uint_as_float((unsigned int)lo), __uint_as_float((unsigned int)(lo >> 32)), __uint_as_float((unsigned int)hi),
                                   __uint_as_float((unsigned int)(hi >> 32)));
      if (i < na) { if (s_oa[r] + i < gv.cap_points) dst_add[s_oa[r] + i] = v; }
      else if (s_on[r] + (i - na) < gv.cap_points) dst_nodown[s_on[r] + (i - na)] = v;
    }
  }
}

}  // namespace

// (a few thousand points per list: a handful of workgroups moves them; every one of the push's ends in a device-scope ticket)
constexpr int kListBlocks = 16;
void launch_lists_push(const GatherView& gv, const float4* src_add, const float4* src_nodown, const int* counts, unsigned int* ticket,
                       unsigned long long seq, hipStream_t s) {
  hipLaunchKernelGGL(k_lists_push, dim3(kListBlocks), dim3(256), 0, s, gv, src_add, src_nodown, counts, ticket, seq);
}
void launch_lists_collect(const GatherView& gv, unsigned long long seq, float4* dst_add, float4* dst_nodown, int* counts, int err_at, hipStream_t s) {
  hipLaunchKernelGGL(k_lists_collect, dim3(kListBlocks), dim3(256), 0, s, gv, seq, dst_add, dst_nodown, counts, err_at);
}
void launch_lists_exchange(const GatherView& gv, const float4* src_add, const float4* src_nodown, int* counts, int err_at, unsigned int* ticket,
                           unsigned long long seq, float4* dst_add, float4* dst_nodown, hipStream_t s) {
  launch_lists_push(gv, src_add, src_nodown, counts, ticket, seq, s);
  launch_lists_collect(gv, seq, dst_add, dst_nodown, counts, err_at, s);
}

}  // namespace lii
