// rocPRIM-backed device sort / scan used by the map-index build and the voxel-grid filter.
// (rocPRIM ships header-only under /opt/rocm/include; it is the platform's native primitive library.)
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "lii_device.h"
#include "lii_launch.h"

namespace lii {

size_t sort_temp_bytes(int max_n) {
  size_t a = 0, b = 0, c = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                            (const unsigned int*)nullptr, (unsigned int*)nullptr, (size_t)max_n, 0, 64, 0);
  (void)rocprim::radix_sort_pairs(nullptr, b, (const unsigned int*)nullptr, (unsigned int*)nullptr, (const unsigned int*)nullptr,
                            (unsigned int*)nullptr, (size_t)max_n, 0, 32, 0);
  (void)rocprim::inclusive_scan(nullptr, c, (const unsigned int*)nullptr, (unsigned int*)nullptr, (size_t)max_n,
                          rocprim::plus<unsigned int>(), 0);
  size_t d = 0;
  (void)rocprim::merge(nullptr, d, (const unsigned long long*)nullptr, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                       (const float4*)nullptr, (const float4*)nullptr, (float4*)nullptr, (size_t)max_n, (size_t)max_n,
                       rocprim::less<unsigned long long>(), 0);
  size_t m = a > b ? a : b;
  m = m > c ? m : c;
  m = m > d ? m : d;
  return m + 256;
}
void sort_pairs_u64(void* temp, size_t temp_bytes, const unsigned long long* kin, unsigned long long* kout,
                    const unsigned int* vin, unsigned int* vout, int n, hipStream_t s) {
  if (n <= 0) return;
  (void)rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t)n, 0, 63, s);
}
void sort_pairs_u32(void* temp, size_t temp_bytes, const unsigned int* kin, unsigned int* kout, const unsigned int* vin,
                    unsigned int* vout, int n, hipStream_t s) {
  if (n <= 0) return;
  (void)rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t)n, 0, 32, s);
}
void inclusive_scan_u32(void* temp, size_t temp_bytes, const unsigned int* in, unsigned int* out, int n, hipStream_t s) {
  if (n <= 0) return;
  (void)rocprim::inclusive_scan(temp, temp_bytes, in, out, (size_t)n, rocprim::plus<unsigned int>(), s);
}

}  // namespace lii
