// Round 6: cycles of the FORWARD SUBSTITUTION phase of k_rows (7 tile steps, MT = 7: 112 MFMAs) in isolation -- the kernel's own
// subst_chain on panels in the kernel's layout, with few other registers live -- against the plain product phase (mfma_chain, 112
// MFMAs): in the kernel the substitution runs at ~150 cycles per MFMA and the plain chains at ~90; here the question is what the
// chain costs when the wave is NOT at 256 VGPRs + 170 AGPRs.  Variants: (0) commit -> barrier -> request two ahead -> chain; (1)
// chains alone on resident panels.
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../tgp/pytorch_amd/csrc -I../../include substphase_rate.hip -o substphase_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "tgp_rows.hpp"
using namespace tgp;
template <int VAR, bool SUBST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const double* __restrict__ LT, double* out, unsigned long long* tm) {
  constexpr int MT = 7, MP = 112;
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  double* pan = reinterpret_cast<double*>(raw);   // VAR 0: 2 x (MP x 16); VAR 1: MT resident panels
  const int tid = threadIdx.x, lane = tid & 63, nl = lane & 15, q = lane >> 4;
  double Kr[4 * MT];
  for (int i = 0; i < 4 * MT; ++i) Kr[i] = 1e-3 * (i + tid);
  double stg[2][MT];
  auto issue = [&](int i, double (&st)[MT]) {     // lower-type panel i: rows [0, 16 (i + 1)) (the last block stands for -Dinv_i^T)
    const double* __restrict__ Mt = LT + (size_t)(tid >> 4) * MP + 16 * i + (tid & 15);
#pragma unroll
    for (int u = 0; u < MT; ++u)
      if (u < i + 1) st[u] = Mt[(size_t)16 * u * MP];
  };
  auto commit = [&](int i, const double (&st)[MT], double* base) {
    double* buf = base + (tid >> 4) * 16 + (tid & 15);
#pragma unroll
    for (int u = 0; u < MT; ++u)
      if (u < i + 1) buf[16 * u * 16] = st[u];
  };
  if (VAR == 1) {
    for (int i = 0; i < MT; ++i) { issue(i, stg[0]); commit(i, stg[0], pan + i * (MP * 16)); }
  }
  issue(0, stg[0]); issue(1, stg[1]);
  __syncthreads();
  d4 Aa[MT];
  unsigned long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const double* buf = VAR == 1 ? pan + i * (MP * 16) : pan + (i & 1) * (MP * 16);
    if (VAR == 0) { commit(i, stg[i & 1], pan + (i & 1) * (MP * 16)); __syncthreads(); if (i + 2 < MT) issue(i + 2, stg[i & 1]); }
    if (SUBST) {
      const d4 c0 = {-Kr[4 * i], -Kr[4 * i + 1], -Kr[4 * i + 2], -Kr[4 * i + 3]};
      Aa[i] = subst_chain<4 * MT>(buf + q * 16 + nl, 4 * i, 4 * i, c0, [&](int st) { return st; }, [&](int st) { return Aa[st / 4][st % 4]; });
    } else {
      Aa[i] = mfma_chain<4 * MT>(buf + q * 16 + nl, 0, 4 * (i + 1), [&](int st) { return Kr[st]; });
    }
  }
  double s = 0;
  for (int i = 0; i < MT; ++i) s += Aa[i][0] + Aa[i][1] + Aa[i][2] + Aa[i][3];
  asm volatile("" : "+v"(s));
  unsigned long long t1 = clock64();
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) tm[0] = t1 - t0;
}
template <int VAR, bool SUBST> void run(const char* name, const double* LT, double* out, unsigned long long* tm, int blocks) {
  const size_t lds = (size_t)(VAR == 1 ? 7 : 2) * 112 * 16 * 8;
  hipFuncSetAttribute((const void*)k<VAR, SUBST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipMemset(tm, 0, 8);
  for (int w = 0; w < 2; ++w) { k<VAR, SUBST><<<blocks, 256, lds>>>(LT, out, tm); if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) printf("launch failed\n"); }
  unsigned long long h; hipMemcpy(&h, tm, 8, hipMemcpyDeviceToHost);
  printf("%-64s %3d blocks: %6llu cycles for 112 MFMAs per wave (%.0f per MFMA; 64 = matrix-pipe bound)\n", name, blocks, h, h / 112.0);
}
int main() {
  double *LT, *out; unsigned long long* tm;
  hipMalloc(&LT, 112 * 112 * 8); hipMalloc(&out, 256 * 256 * 8); hipMalloc(&tm, 8);
  hipMemset(LT, 0, 112 * 112 * 8);
  for (int blocks : {1, 216}) {
    run<0, true>("substitution: commit, barrier, prefetch, subst_chain", LT, out, tm, blocks);
    run<1, true>("substitution: subst_chain alone on resident panels", LT, out, tm, blocks);
    run<0, false>("plain product: commit, barrier, prefetch, mfma_chain", LT, out, tm, blocks);
    run<1, false>("plain product: mfma_chain alone on resident panels", LT, out, tm, blocks);
  }
  return 0;
}
