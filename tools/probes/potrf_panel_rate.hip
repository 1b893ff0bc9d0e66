// Cycles of the round-3 panel factorisation primitives, registers only, one wave:
//   potrf_trtri16 (rounds 1-2), potrf_panel16<true> (diagonal tile + 64 panel rows), potrf_panel16<false>, trtri16
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../tgp/pytorch_amd/csrc potrf_panel_rate.hip -o potrf_panel_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "tgp_dev.hpp"
using namespace tgp;
template <int WHAT>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* tm, const double* A, int iters) {
  const int lane = threadIdx.x & 63;
  double a0[16], p0[16];
  for (int c = 0; c < 16; ++c) { a0[c] = A[(lane & 15) * 16 + c]; p0[c] = 0.01 * (lane + 1) / (1.0 + c); }
  double acc = 0;
  unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    double a[16], x[16], pr[16];
    for (int c = 0; c < 16; ++c) { a[c] = a0[c] + 1e-30 * acc; pr[c] = p0[c]; x[c] = 0.0; }
    int bad = 0;
    if (WHAT == 0) bad = potrf_trtri16(a, x, lane & 15);
    if (WHAT == 1) bad = potrf_panel16<true>(a, pr);
    if (WHAT == 2) bad = potrf_panel16<false>(a, pr);
    if (WHAT == 3) trtri16(a, x, lane & 15);
    if (WHAT == 4) bad = potrf_panel16<true, true>(a, pr, x, lane & 15);
    double s = bad;
    for (int c = 0; c < 16; ++c) s += a[c] + x[c] + pr[c];
    acc = s;
    asm volatile("" : "+v"(acc));
  }
  unsigned long long t1 = clock64();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) tm[0] = (t1 - t0) / iters;
}
int main() {
  double h[256];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + (i > j ? i - j : j - i));
  double* A; hipMalloc(&A, 2048); hipMemcpy(A, h, 2048, hipMemcpyHostToDevice);
  double* out; unsigned long long* tm; hipMalloc(&out, 512); hipMalloc(&tm, 8);
  const char* names[5] = {"potrf_trtri16 (rounds 1-2)", "potrf_panel16<true> (tile + 64 panel rows)", "potrf_panel16<false> (tile only)", "trtri16 (stand-alone)", "potrf_panel16<true, true> (tile + 64 rows + inverse)"};
  for (int w = 0; w < 5; ++w) {
    for (int rep = 0; rep < 2; ++rep) {
      if (w == 0) k<0><<<1, 64>>>(out, tm, A, 20);
      if (w == 1) k<1><<<1, 64>>>(out, tm, A, 20);
      if (w == 2) k<2><<<1, 64>>>(out, tm, A, 20);
      if (w == 3) k<3><<<1, 64>>>(out, tm, A, 20);
      if (w == 4) k<4><<<1, 64>>>(out, tm, A, 20);
      hipDeviceSynchronize();
    }
    unsigned long long r; hipMemcpy(&r, tm, 8, hipMemcpyDeviceToHost);
    double o[64]; hipMemcpy(o, out, 512, hipMemcpyDeviceToHost);
    printf("%-46s %6llu cycles per call (%.2f us at 2.3 GHz), checksum %.12g\n", names[w], r, r / 2300.0, o[3]);
  }
  return 0;
}
