// Micro-benchmarks behind the round-5 work on the 24-state solve (k_reduce_solve, lii_iekf.hip): what do the instructions of one
// elimination step cost a wavefront that has the compute unit to itself?  One workgroup of 64 (or 256) lanes; every block is timed
// with s_memtime (shader clock) and s_memrealtime (100 MHz), so that the clock the lone workgroup runs at comes out as well.
// build: hipcc --offload-arch=gfx950 -O3 -o solve_ubench solve_ubench.hip ; run: ./solve_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}
#define T0() do { __builtin_amdgcn_s_waitcnt(0); asm volatile("s_nop 0" ::: "memory"); c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); } while (0)
#define T1(slot) do { __builtin_amdgcn_s_waitcnt(0); asm volatile("s_nop 0" ::: "memory"); c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime(); if (threadIdx.x == 0) { out[2 * (slot)] = (long long)(c1 - c0); out[2 * (slot) + 1] = (long long)(r1 - r0); } } while (0)

__global__ void k_bench(long long* out, double* sink, const double* src, int reps) {
  __shared__ double lds[64 * 16];
  unsigned long long c0, c1, r0, r1;
  const int lane = threadIdx.x & 63;
  double col[12];
  for (int r = 0; r < 12; r++) col[r] = src[lane * 12 + r];
  double acc = 0;
  // 0: empty (timer overhead)
  T0(); T1(0);
  // 1: reps x 24 v_readlane (one column of 12 doubles from lane k), results folded with 12 fma
  T0();
  for (int it = 0; it < reps; it++) {
    const int k = it % 12;
    double m[12];
#pragma unroll
    for (int r = 0; r < 12; r++) m[r] = readlane_f64(col[r], k);
#pragma unroll
    for (int r = 0; r < 12; r++) col[r] = fma(m[r], 1e-9, col[r]);
  }
  T1(1);
  // 2: the same without the readlanes (12 fma on a loop-carried value)
  T0();
  for (int it = 0; it < reps; it++) {
#pragma unroll
    for (int r = 0; r < 12; r++) col[r] = fma(col[(r + 1) % 12], 1e-9, col[r]);
  }
  T1(2);
  // 3: reciprocal + two Newton steps + multiply, dependent chain
  T0();
  double x = col[0] + 2.0;
  for (int it = 0; it < reps; it++) {
    double inv = __builtin_amdgcn_rcp(x);
    inv = fma(fma(-x, inv, 1.0), inv, inv);
    inv = fma(fma(-x, inv, 1.0), inv, inv);
    x = x * inv + 1.5;
  }
  acc += x;
  T1(3);
  // 4: the broadcast through LDS: lane k writes its 12 doubles (6 x 16 bytes), every lane reads them back (same address)
  T0();
  for (int it = 0; it < reps; it++) {
    const int k = it % 12;
    if (lane == k) {
#pragma unroll
      for (int r = 0; r < 12; r += 2) { d2v w; w.x = col[r]; w.y = col[r + 1]; *reinterpret_cast<volatile d2v*>(&lds[r]) = w; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    double m[12];
#pragma unroll
    for (int r = 0; r < 12; r += 2) { const d2v v = *reinterpret_cast<const volatile d2v*>(&lds[r]); m[r] = v.x; m[r + 1] = v.y; }
#pragma unroll
    for (int r = 0; r < 12; r++) col[r] = fma(m[r], 1e-9, col[r]);
  }
  T1(4);
  // 5: 12 v_max_f64 with |.| (the growth watch)
  T0();
  double g = 0;
  for (int it = 0; it < reps; it++) {
#pragma unroll
    for (int r = 0; r < 12; r++) asm volatile("v_max_f64 %0, |%1|, |%2|" : "=v"(g) : "v"(g), "v"(col[r]));
  }
  acc += g;
  T1(5);
  // 6: one full elimination step as in gj12_loop (readlanes, reciprocal, 11 fma, 12 max)
  T0();
  for (int it = 0; it < reps; it++) {
    const int k = it % 12;
    double m[12];
#pragma unroll
    for (int r = 0; r < 12; r++) m[r] = readlane_f64(col[r], k);
    double inv = __builtin_amdgcn_rcp(m[0] + 3.0);
    inv = fma(fma(-(m[0] + 3.0), inv, 1.0), inv, inv);
    inv = fma(fma(-(m[0] + 3.0), inv, 1.0), inv, inv);
    const double rowk = col[0] * inv;
#pragma unroll
    for (int r = 1; r < 12; r++) col[r - 1] = fma(-m[r] * 1e-9, rowk, col[r]);
    col[11] = rowk;
#pragma unroll
    for (int r = 0; r < 12; r++) asm volatile("v_max_f64 %0, |%1|, |%2|" : "=v"(g) : "v"(g), "v"(col[r]));
  }
  acc += g;
  T1(6);
  // 7: the same step with the LDS broadcast instead of the readlanes
  T0();
  for (int it = 0; it < reps; it++) {
    const int k = it % 12;
    if (lane == k) {
#pragma unroll
      for (int r = 0; r < 12; r += 2) { d2v w; w.x = col[r]; w.y = col[r + 1]; *reinterpret_cast<volatile d2v*>(&lds[r]) = w; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    double m[12];
#pragma unroll
    for (int r = 0; r < 12; r += 2) { const d2v v = *reinterpret_cast<const volatile d2v*>(&lds[r]); m[r] = v.x; m[r + 1] = v.y; }
    double inv = __builtin_amdgcn_rcp(m[0] + 3.0);
    inv = fma(fma(-(m[0] + 3.0), inv, 1.0), inv, inv);
    inv = fma(fma(-(m[0] + 3.0), inv, 1.0), inv, inv);
    const double rowk = col[0] * inv;
#pragma unroll
    for (int r = 1; r < 12; r++) col[r - 1] = fma(-m[r] * 1e-9, rowk, col[r]);
    col[11] = rowk;
#pragma unroll
    for (int r = 0; r < 12; r++) asm volatile("v_max_f64 %0, |%1|, |%2|" : "=v"(g) : "v"(g), "v"(col[r]));
  }
  acc += g;
  T1(7);
  // 8: 48 independent v_fma_f64 per iteration (issue rate of the double-precision pipe)
  T0();
  double f[12];
  for (int r = 0; r < 12; r++) f[r] = col[r];
  for (int it = 0; it < reps; it++) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int r = 0; r < 12; r++) f[r] = fma(f[r], 1.0000001, 1e-12);
  }
  for (int r = 0; r < 12; r++) acc += f[r];
  T1(8);
  for (int r = 0; r < 12; r++) acc += col[r];
  sink[threadIdx.x] = acc;
}

int main() {
  long long* d_out; double *d_sink, *d_src;
  hipMalloc(&d_out, 64 * sizeof(long long)); hipMalloc(&d_sink, 1024 * sizeof(double)); hipMalloc(&d_src, 64 * 12 * 8);
  std::vector<double> src(64 * 12);
  for (size_t i = 0; i < src.size(); i++) src[i] = 1.0 + 0.001 * (double)(i % 97);
  hipMemcpy(d_src, src.data(), src.size() * 8, hipMemcpyHostToDevice);
  const char* names[9] = {"empty", "24 readlane + 12 fma", "12 fma", "rcp + 4 fma + mul + add", "lds broadcast (6 w + 6 r) + 12 fma", "12 max", "gj step (readlane)", "gj step (lds)", "48 fma"};
  for (int threads : {64, 256}) {
    for (int pass = 0; pass < 3; pass++) {
      const int reps = 120;
      hipLaunchKernelGGL(k_bench, dim3(1), dim3(threads), 0, 0, d_out, d_sink, d_src, reps);
      hipDeviceSynchronize();
      long long out[64];
      hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
      if (pass < 2) continue;
      printf("%d lanes, %d repetitions per block:\n", threads, reps);
      for (int s = 0; s < 9; s++)
        printf("  %-36s %8.1f clk / rep  %7.1f ns / rep   (clock %.2f GHz)\n", names[s], (double)(out[2 * s] - out[0]) / reps, (double)(out[2 * s + 1] - out[1]) * 10.0 / reps,
               out[2 * s + 1] > out[1] ? (double)(out[2 * s] - out[0]) / ((double)(out[2 * s + 1] - out[1]) * 10.0) : 0.0);
    }
  }
  return 0;
}
