// Cycles of one potrf_trtri16 call (16x16 Cholesky + inverse by one wave), registers only.
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../tgp/pytorch_amd/csrc -I../../include potrf_rate.hip -o potrf_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "tgp_dev.hpp"
using namespace tgp;
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* tm, const double* A, int iters) {
  const int lane = threadIdx.x & 63;
  double a0[16];
  for (int c = 0; c < 16; ++c) a0[c] = A[(lane & 15) * 16 + c];
  double acc = 0;
  // one timed region around `iters` DEPENDENT calls (the next input depends on the previous result), so that nothing
  // can be moved out of the region
  unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    double a[16], x[16];
    for (int c = 0; c < 16; ++c) a[c] = a0[c] + 1e-30 * acc;
    const int bad = potrf_trtri16(a, x, lane);
    double s = bad;
    for (int c = 0; c < 16; ++c) s += a[c] + x[c];
    acc = s;
    asm volatile("" : "+v"(acc));
  }
  unsigned long long t1 = clock64();
  const unsigned long long t = t1 - t0;
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) tm[0] = t / iters;
}
int main() {
  double h[256];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + (i > j ? i - j : j - i));
  double* A; hipMalloc(&A, 2048); hipMemcpy(A, h, 2048, hipMemcpyHostToDevice);
  double* out; unsigned long long* tm; hipMalloc(&out, 512); hipMalloc(&tm, 8);
  k<<<1, 64>>>(out, tm, A, 20); hipDeviceSynchronize();
  unsigned long long r; hipMemcpy(&r, tm, 8, hipMemcpyDeviceToHost);
  double o[64]; hipMemcpy(o, out, 512, hipMemcpyDeviceToHost);
  printf("potrf_trtri16: %llu cycles per call (%.2f us at 2.3 GHz), checksum %.12g\n", r, r / 2300.0, o[3]);
  return 0;
}
