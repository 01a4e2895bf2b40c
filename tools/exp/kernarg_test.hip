// experiment: how large may a kernel argument block be on gfx950 / ROCm 7.2?  (pose table passed by value)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { double v[N]; };
template <int N> __global__ void k(Big<N> b, double* out) {
  double s = 0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) s += b.v[i];
  atomicAdd(out, s);
}
template <int N> void run(double* d) {
  Big<N> b;
  double want = 0;
  for (int i = 0; i < N; i++) { b.v[i] = i * 0.5 + 1; want += b.v[i]; }
  hipMemset(d, 0, 8);
  hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
  hipError_t e = hipGetLastError();
  hipError_t e2 = hipDeviceSynchronize();
  double got = 0;
  hipMemcpy(&got, d, 8, hipMemcpyDeviceToHost);
  printf("kernarg %5zu B: launch=%s sync=%s got=%g want=%g %s\n", sizeof(b) + 8, hipGetErrorString(e), hipGetErrorString(e2), got, want, got == want ? "OK" : "MISMATCH");
  // launch cost of a big kernarg: 2000 launches
  hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
  hipEventRecord(a, 0);
  for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
  hipEventRecord(c, 0); hipEventSynchronize(c);
  float ms = 0; hipEventElapsedTime(&ms, a, c);
  printf("   %.2f us per launch (back to back)\n", ms * 1000 / 2000);
}
int main() {
  double* d; hipMalloc(&d, 8);
  run<64>(d); run<500>(d); run<511>(d); run<528>(d); run<768>(d); run<1024>(d); run<2048>(d);
  return 0;
}
