// Cycles of the GEMM 1+2 phase of k_rows (14 triangular tile steps, MT = 7) in isolation, with the kernel's own
// mfma_chain, and with its parts removed one at a time: (0) as in the kernel: commit panel -> barrier -> request the
// panel two ahead -> chain; (1) no global loads (the commit writes stale registers), barrier kept; (2) no commit and no
// barrier either: the chains alone.
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../tgp/pytorch_amd/csrc -I../../include gemmphase_rate.hip -o gemmphase_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "tgp_rows.hpp"
using namespace tgp;
template <int VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const double* __restrict__ JT, const double* __restrict__ Lq,
                                                                                 double* out, unsigned long long* tm) {
  constexpr int MT = 7, MP = 112;
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  double* pan = reinterpret_cast<double*>(raw);   // 2 x (MP x 16) (+ 14 resident panels for VAR >= 1)
  const int tid = threadIdx.x, lane = tid & 63, nl = lane & 15, q = lane >> 4;
  double Kr[4 * MT];
  for (int i = 0; i < 4 * MT; ++i) Kr[i] = 1e-3 * (i + tid);
  double stg[2][MT];
  auto issue = [&](int pp, double (&st)[MT]) {
    const bool lower = pp < MT;
    const int i = lower ? pp : pp - MT;
    const double* __restrict__ Mt = (lower ? JT : Lq) + (size_t)((lower ? 0 : 16 * i) + (tid >> 4)) * MP + 16 * i + (tid & 15);
    const int nb = lower ? i + 1 : MT - i;
#pragma unroll
    for (int u = 0; u < MT; ++u)
      if (u < nb) st[u] = Mt[(size_t)16 * u * MP];
  };
  auto commit = [&](int pp, const double (&st)[MT], double* base) {
    const bool lower = pp < MT;
    const int i = lower ? pp : pp - MT;
    const int nb = lower ? i + 1 : MT - i;
    double* buf = base + ((lower ? 0 : 16 * i) + (tid >> 4)) * 16 + (tid & 15);
#pragma unroll
    for (int u = 0; u < MT; ++u)
      if (u < nb) buf[16 * u * 16] = st[u];
  };
  issue(0, stg[0]); issue(1, stg[1]);
  commit(0, stg[0], pan); commit(1, stg[1], pan + MP * 16);
  __syncthreads();
  d4 Aa[MT], Ba[MT];
  unsigned long long t0 = clock64();
  if (VAR == 0) { issue(0, stg[0]); issue(1, stg[1]); }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const double* buf = pan + (i & 1) * (MP * 16);
    if (VAR <= 1) commit(i, stg[i & 1], pan + (i & 1) * (MP * 16));
    if (VAR <= 1) __syncthreads();
    if (VAR == 0 && i + 2 < 2 * MT) issue(i + 2, stg[i & 1]);
    Aa[i] = mfma_chain<4 * MT>(buf + q * 16 + nl, 0, 4 * (i + 1), [&](int st) { return Kr[st]; });
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int pp = MT + i;
    const double* buf = pan + (pp & 1) * (MP * 16);
    if (VAR <= 1) commit(pp, stg[pp & 1], pan + (pp & 1) * (MP * 16));
    if (VAR <= 1) __syncthreads();
    if (VAR == 0 && pp + 2 < 2 * MT) issue(pp + 2, stg[pp & 1]);
    Ba[i] = mfma_chain<4 * MT>(buf + q * 16 + nl, 4 * i, 4 * (MT - i), [&](int st) { return Aa[i + st / 4][st % 4]; });
  }
  double s = 0;
  for (int i = 0; i < MT; ++i) s += Ba[i][0] + Ba[i][1] + Ba[i][2] + Ba[i][3];
  asm volatile("" : "+v"(s));
  unsigned long long t1 = clock64();
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) tm[0] = t1 - t0;
}
template <int VAR> void run(const char* name, const double* JT, const double* Lq, double* out, unsigned long long* tm, int blocks) {
  const size_t lds = (size_t)2 * 112 * 16 * 8;
  hipFuncSetAttribute((const void*)k<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipMemset(tm, 0, 8);
  for (int w = 0; w < 2; ++w) { k<VAR><<<blocks, 256, lds>>>(JT, Lq, out, tm); if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) printf("launch failed\n"); }
  unsigned long long h; hipMemcpy(&h, tm, 8, hipMemcpyDeviceToHost);
  printf("%-46s %3d blocks: %6llu cycles for 224 MFMAs per wave (%.0f per MFMA; 64 = matrix-pipe bound)\n", name, blocks, h, h / 224.0);
}
int main() {
  double *JT, *Lq, *out; unsigned long long* tm;
  hipMalloc(&JT, 112 * 112 * 8); hipMalloc(&Lq, 112 * 112 * 8); hipMalloc(&out, 256 * 256 * 8); hipMalloc(&tm, 8);
  hipMemset(JT, 0, 112 * 112 * 8); hipMemset(Lq, 0, 112 * 112 * 8);
  for (int blocks : {1, 135}) {
    run<0>("as in k_rows (commit, barrier, prefetch, chain)", JT, Lq, out, tm, blocks);
    run<1>("no global loads, commit + barrier kept", JT, Lq, out, tm, blocks);
    run<2>("chains alone (no commit, no barrier)", JT, Lq, out, tm, blocks);
  }
  return 0;
}
