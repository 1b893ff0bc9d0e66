// Cycles of flow_forward_store<4> / flow_backward_store<4> (tanh 3x2 program) called exactly as k_rows calls them:
// program, parameters, stack and accumulators in LDS; 1 wave and 4 waves per workgroup.
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../tgp/pytorch_amd/csrc -I../../include flowsweep_rate.hip -o flowsweep_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "tgp_dev.hpp"
using namespace tgp;
__global__ void k(double* out, unsigned long long* tm, FlowProg prog, const double* tp_g, int P, int trips) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  double* sm = reinterpret_cast<double*>(raw);
  double* tpL = sm; double* tgL = sm + 64; double* tiL = sm + 128;
  int32_t* progL = reinterpret_cast<int32_t*>(sm + 192);
  double* acc = sm + 256;            // [P][64]
  double* stack = acc + 64 * 64;     // [nslots*4][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 15, q = lane >> 4;
  if (tid < P) { tpL[tid] = tp_g[tid]; tgL[tid] = 0.5; tiL[tid] = rcp_fast(tp_g[tid]); }
  for (int i = tid; i < 4 * prog.nblk; i += blockDim.x) progL[i] = prog.blk[i];
  for (int i = tid; i < 64 * 64; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  FlowDev F{progL, prog.nblk, tpL, tgL, tiL};
  double cm = 0;
  unsigned long long tf = 0, tb = 0;
  for (int it = 0; it < trips; ++it) {
    double f[4], c[4];
    for (int u = 0; u < 4; ++u) f[u] = 0.1 * u + 1e-3 * tid + 0.01 * it;
    unsigned long long t0 = clock64();
    flow_forward_store<4>(F, f, nullptr, stack + tid, 256);
    unsigned long long t1 = clock64();
    for (int u = 0; u < 4; ++u) c[u] = 1.0 - f[u];
    flow_backward_store<4>(F, c, nullptr, stack + tid, 256, prog.nslots, acc + wave * 16 + nl, 64, q == 0, acc, 256);
    unsigned long long t2 = clock64();
    for (int u = 0; u < 4; ++u) cm += c[u];
    if (it > 0) { tf += t1 - t0; tb += t2 - t1; }
  }
  out[blockIdx.x * blockDim.x + tid] = cm + acc[tid];
  if (tid == 0 && blockIdx.x == 0) { tm[0] = tf / (trips - 1); tm[1] = tb / (trips - 1); }
}
int main() {
  FlowProg prog{};
  int P = 0, b = 0;
  for (int blk = 0; blk < 3; ++blk) {
    prog.blk[4 * b] = 2; prog.blk[4 * b + 1] = 2; prog.blk[4 * b + 2] = P; prog.blk[4 * b + 3] = 2; P += 8; ++b;
    prog.blk[4 * b] = 0; prog.blk[4 * b + 1] = 0; prog.blk[4 * b + 2] = P; prog.blk[4 * b + 3] = 0; P += 2; ++b;
  }
  prog.nblk = b; prog.nslots = flow_slots(prog.blk, b);
  double h[64]; for (int i = 0; i < 64; ++i) h[i] = 0.7 + 0.01 * i;
  double* tp; hipMalloc(&tp, 512); hipMemcpy(tp, h, 512, hipMemcpyHostToDevice);
  double* out; unsigned long long* tm; hipMalloc(&out, 256 * 256 * 8); hipMalloc(&tm, 16);
  const size_t lds = (256 + 64 * 64 + (size_t)prog.nslots * 4 * 256) * 8;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int threads : {64, 256}) for (int blocks : {1, 135}) {
    k<<<blocks, threads, lds>>>(out, tm, prog, tp, P, 9); hipDeviceSynchronize();
    unsigned long long r[2]; hipMemcpy(r, tm, 16, hipMemcpyDeviceToHost);
    printf("%3d blocks x %3d thr: forward sweep %llu cycles, backward sweep %llu cycles (nslots %d, lds %zu)\n", blocks, threads, r[0], r[1], prog.nslots, lds);
  }
  return 0;
}
