// Does instruction fetch limit long straight-line code?  A block of NI unrolled, 8-chain-interleaved f64 FMAs is run
// PASSES times by one wave (and by 4 waves x 135 blocks); cycles per instruction for the cold first pass and for the
// warm later ones.  Build: hipcc -O3 --offload-arch=gfx950 icache_rate.hip -o icache_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NI>
__global__ void k(double* out, unsigned long long* tm, int passes) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = 1.0 + threadIdx.x * 1e-3 + i;
  const double y = 0.999999, z = 1e-9;
  for (int ps = 0; ps < passes; ++ps) {
    unsigned long long t0 = clock64();
#pragma unroll
    for (int rep = 0; rep < NI / 8; ++rep)
#pragma unroll
      for (int i = 0; i < 8; ++i) { x[i] = fma(x[i], y, z); asm volatile("" : "+v"(x[i])); }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) tm[ps] = t1 - t0;
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NI> void run(int blocks, int threads) {
  double* out; unsigned long long* tm; hipMalloc(&out, blocks * threads * 8); hipMalloc(&tm, 64);
  k<NI><<<blocks, threads>>>(out, tm, 4); hipDeviceSynchronize();
  unsigned long long h[4]; hipMemcpy(h, tm, 32, hipMemcpyDeviceToHost);
  printf("NI %5d (%3d KB code) %3d blocks x %3d thr: cycles/instr pass0 %.2f pass1 %.2f pass2 %.2f pass3 %.2f\n", NI, NI * 8 / 1024, blocks, threads,
         (double)h[0] / NI, (double)h[1] / NI, (double)h[2] / NI, (double)h[3] / NI);
  hipFree(out); hipFree(tm);
}
int main() {
  run<512>(1, 64); run<2048>(1, 64); run<4096>(1, 64); run<8192>(1, 64); run<16384>(1, 64);
  run<2048>(135, 256); run<8192>(135, 256); run<16384>(135, 256);
  return 0;
}
