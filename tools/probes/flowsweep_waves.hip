// The flow quadrature (flow_forward_store + flow_backward_store, tanh 3x2 program, everything in LDS as in the row kernels)
// with 1, 2, 3 waves per SIMD and 1, 2, 4 nodes in flight per lane: cycles per node evaluation PER SIMD LANE, i.e. what the
// phase costs for a fixed amount of quadrature work on a CU, however it is spread over waves and nodes in flight.
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../tgp/pytorch_amd/csrc -I../../include flowsweep_waves.hip -o flowsweep_waves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "tgp_dev.hpp"
using namespace tgp;
template <int NB>
__global__ void k(double* out, unsigned long long* tm, FlowProg prog, const double* tp_g, int P, int trips) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  double* sm = reinterpret_cast<double*>(raw);
  double* tpL = sm; double* tgL = sm + 64; double* tiL = sm + 128;
  int32_t* progL = reinterpret_cast<int32_t*>(sm + 192);
  const int nthr = blockDim.x;
  double* acc = sm + 256;                       // [P][nthr/4]
  double* stack = acc + 32 * (nthr / 4);        // [nslots*NB][nthr]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 15, q = lane >> 4;
  if (tid < P) { tpL[tid] = tp_g[tid]; tgL[tid] = 0.5; tiL[tid] = rcp_fast(tp_g[tid]); }
  for (int i = tid; i < 4 * prog.nblk; i += nthr) progL[i] = prog.blk[i];
  for (int i = tid; i < 32 * (nthr / 4); i += nthr) acc[i] = 0;
  __syncthreads();
  FlowDev F{progL, prog.nblk, tpL, tgL, tiL};
  double cm = 0;
  __syncthreads();
  unsigned long long t0 = clock64();
  for (int it = 0; it < trips; ++it) {
    double f[NB], c[NB];
    for (int u = 0; u < NB; ++u) f[u] = 0.1 * u + 1e-3 * tid + 0.01 * it;
    flow_forward_store<NB>(F, f, nullptr, stack + tid, nthr);
    for (int u = 0; u < NB; ++u) c[u] = 1.0 - f[u];
    flow_backward_store<NB>(F, c, nullptr, stack + tid, nthr, prog.nslots, acc + wave * 16 + nl, nthr / 4, q == 0, acc, nthr);
    for (int u = 0; u < NB; ++u) cm += c[u];
  }
  __syncthreads();
  unsigned long long t1 = clock64();
  out[blockIdx.x * nthr + tid] = cm + acc[tid];
  if (tid == 0 && blockIdx.x == 0) tm[0] = t1 - t0;
}
template <int NB>
void run(int threads, int trips, FlowProg prog, const double* tp, int P, double* out, unsigned long long* tm) {
  const size_t lds = (256 + 32 * (size_t)(threads / 4) + (size_t)prog.nslots * NB * threads) * 8;
  if (lds > 160 * 1024 - 512) { printf("%d waves/SIMD, %d nodes in flight: LDS %zu KB does not fit\n", threads / 256, NB, lds / 1024); return; }
  hipFuncSetAttribute((const void*)k<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<NB><<<1, threads, lds>>>(out, tm, prog, tp, P, trips); hipDeviceSynchronize();
  k<NB><<<1, threads, lds>>>(out, tm, prog, tp, P, trips); hipDeviceSynchronize();
  unsigned long long r; hipMemcpy(&r, tm, 8, hipMemcpyDeviceToHost);
  const double evals = (double)(threads / 256) * NB * trips;   // node evaluations per SIMD lane
  printf("%d waves/SIMD, %d nodes in flight, %2d trips: %8llu cycles = %.0f cycles per node evaluation per SIMD lane (LDS %zu KB)\n",
         threads / 256, NB, trips, r, r / evals, lds / 1024);
}
int main() {
  FlowProg prog{};
  int P = 0, b = 0;
  const int NBLK = getenv("NBLK") ? atoi(getenv("NBLK")) : 3;
  for (int blk = 0; blk < NBLK; ++blk) {
    prog.blk[4 * b] = 2; prog.blk[4 * b + 1] = 2; prog.blk[4 * b + 2] = P; prog.blk[4 * b + 3] = 2; P += 8; ++b;
    prog.blk[4 * b] = 0; prog.blk[4 * b + 1] = 0; prog.blk[4 * b + 2] = P; prog.blk[4 * b + 3] = 0; P += 2; ++b;
  }
  prog.nblk = b; prog.nslots = flow_slots(prog.blk, b);
  double h[64]; for (int i = 0; i < 64; ++i) h[i] = 0.7 + 0.01 * i;
  double* tp; hipMalloc(&tp, 512); hipMemcpy(tp, h, 512, hipMemcpyHostToDevice);
  double* out; unsigned long long* tm; hipMalloc(&out, 1024 * 8); hipMalloc(&tm, 16);
  run<8>(256, 3, prog, tp, P, out, tm);
  run<4>(256, 6, prog, tp, P, out, tm);
  run<2>(256, 12, prog, tp, P, out, tm);
  run<1>(256, 24, prog, tp, P, out, tm);
  run<4>(512, 3, prog, tp, P, out, tm);
  run<2>(512, 6, prog, tp, P, out, tm);
  run<1>(512, 12, prog, tp, P, out, tm);
  run<2>(768, 4, prog, tp, P, out, tm);
  run<1>(768, 8, prog, tp, P, out, tm);
  run<1>(1024, 6, prog, tp, P, out, tm);
  return 0;
}
