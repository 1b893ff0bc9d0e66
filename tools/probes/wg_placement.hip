// Where does the hardware put the workgroups of a launch that fits two per CU?  Records, per workgroup, the XCC / SE / CU it
// ran on and its start time, for a launch shaped like tgp::k_gemm (256 threads, 74 KB of LDS, ~250 VGPRs -> two per CU).
// Question behind it: do workgroups i and i + 256 of a fresh launch share a CU, i.e. could a static tile order
// pair a long k-chain with a short one on the same CU?
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 wg_placement.hip -o wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>

__global__ __launch_bounds__(256, 2) void k_place(unsigned* out, long long* t0, int spin) {
  extern __shared__ double sm[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const long long t = wall_clock64();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; t0[blockIdx.x] = t; }
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 1e-9);
  sm[threadIdx.x] = x;
  if (x == 12345.678) out[0] = 0;
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 632, spin = argc > 2 ? atoi(argv[2]) : 20000;
  unsigned* out; long long* t0;
  hipMalloc(&out, nwg * 8); hipMalloc(&t0, nwg * 8);
  hipFuncSetAttribute((const void*)k_place, hipFuncAttributeMaxDynamicSharedMemorySize, 74 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_place, dim3(nwg), dim3(256), 74 * 1024, 0, out, t0, spin);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * nwg); std::vector<long long> ht(nwg);
    hipMemcpy(h.data(), out, nwg * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t0, nwg * 8, hipMemcpyDeviceToHost);
    long long tmin = ht[0]; for (auto v : ht) tmin = v < tmin ? v : tmin;
    auto cu_of = [&](int i) { const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15; return (int)(xcc << 16 | ((hw >> 13) & 7) << 8 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 15)); };
    int same256 = 0, mod8 = 0; std::map<int, int> per_cu;
    for (int i = 0; i < nwg; ++i) { per_cu[cu_of(i)]++; if ((int)(h[2 * i + 1] & 15) == (i & 7)) ++mod8; }
    for (int i = 0; i + 256 < nwg && i < 256; ++i) if (cu_of(i) == cu_of(i + 256)) ++same256;
    printf("rep %d: %d workgroups on %zu distinct CUs; xcc == id mod 8 for %d; (i, i+256) on the same CU: %d of %d\n", rep, nwg, per_cu.size(), mod8, same256, nwg > 256 ? (nwg - 256 < 256 ? nwg - 256 : 256) : 0);
    if (rep == 2) {
      printf("first 24 workgroups: id xcc se sh cu  start(ticks)\n");
      for (int i = 0; i < 24; ++i) printf("  %3d  %u %u %u %2u  %lld\n", i, h[2 * i + 1] & 15, (h[2 * i] >> 13) & 7, (h[2 * i] >> 12) & 1, (h[2 * i] >> 8) & 15, ht[i] - tmin);
      printf("workgroups 256..279:\n");
      for (int i = 256; i < 280 && i < nwg; ++i) printf("  %3d  %u %u %u %2u  %lld\n", i, h[2 * i + 1] & 15, (h[2 * i] >> 13) & 7, (h[2 * i] >> 12) & 1, (h[2 * i] >> 8) & 15, ht[i] - tmin);
      // which earlier workgroup shares the CU of workgroup i (i >= 256)?
      int hist[8] = {0};
      for (int i = 256; i < 512 && i < nwg; ++i) { int j = -1; for (int k = 0; k < 256; ++k) if (cu_of(k) == cu_of(i)) { j = k; break; } const int d = j < 0 ? 7 : ((i - 256 - j) == 0 ? 0 : (((i - j) & 7) == 0 ? 1 : 2)); hist[d]++; }
      printf("second-slot workgroups 256..511: partner is i-256: %d, another id of the same XCD: %d, other: %d, none: %d\n", hist[0], hist[1], hist[2], hist[7]);
    }
  }
  return 0;
}
