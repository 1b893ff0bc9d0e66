// Issue interval of f64 VALU instructions on one wave per SIMD: NCH independent chains of one op, timed with the
// shader clock (clock64) and the 100 MHz realtime counter.  Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
enum { OP_FMA, OP_MUL, OP_ADD, OP_RCP, OP_LDEXP, OP_RNDNE, OP_MAX, OP_FMA32 };
template <int OP, int NCH>
__global__ void k(double* out, unsigned long long* tm, int iters) {
  double x[NCH];
  float xf[NCH];
  for (int i = 0; i < NCH; ++i) { x[i] = 1.0 + threadIdx.x * 1e-3 + i; xf[i] = (float)x[i]; }
  const double y = 0.999999, z = 1e-9;
  unsigned long long t0 = clock64(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep)
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (OP == OP_FMA) x[i] = fma(x[i], y, z);
        if (OP == OP_MUL) x[i] = x[i] * y;
        if (OP == OP_ADD) x[i] = x[i] + z;
        if (OP == OP_RCP) x[i] = __builtin_amdgcn_rcp(x[i]);
        if (OP == OP_LDEXP) x[i] = ldexp(x[i], (int)threadIdx.x & 1);
        if (OP == OP_RNDNE) x[i] = rint(x[i]);
        if (OP == OP_MAX) x[i] = fmax(x[i], z);
        if (OP == OP_FMA32) xf[i] = fmaf(xf[i], 0.999f, 1e-6f);
        asm volatile("" : "+v"(x[i]), "+v"(xf[i]));
      }
  }
  unsigned long long t1 = clock64(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NCH; ++i) s += x[i] + xf[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { tm[0] = t1 - t0; tm[1] = r1 - r0; }
}
template <int OP, int NCH> void run(const char* name, int threads = 64) {
  double* out; unsigned long long* tm; hipMalloc(&out, 1024 * 8); hipMalloc(&tm, 16);
  const int iters = 500;
  for (int w = 0; w < 2; ++w) { k<OP, NCH><<<1, threads>>>(out, tm, iters); hipDeviceSynchronize(); }
  unsigned long long h[2]; hipMemcpy(h, tm, 16, hipMemcpyDeviceToHost);
  const double n = (double)iters * 8 * NCH;
  printf("%-8s chains %d, %d thr: %.2f cycles/instr  %.2f ns/instr  (clock %.2f GHz)\n", name, NCH, threads, h[0] / n, h[1] * 10.0 / n, h[0] / (h[1] * 10.0));
  hipFree(out); hipFree(tm);
}
int main() {
  run<OP_FMA, 1>("fma"); run<OP_FMA, 2>("fma"); run<OP_FMA, 4>("fma"); run<OP_FMA, 8>("fma"); run<OP_FMA, 8>("fma", 256); run<OP_FMA, 8>("fma", 512);
  run<OP_MUL, 1>("mul"); run<OP_MUL, 8>("mul");
  run<OP_ADD, 1>("add"); run<OP_ADD, 8>("add");
  run<OP_MAX, 1>("max"); run<OP_MAX, 8>("max");
  run<OP_RCP, 1>("rcp"); run<OP_RCP, 8>("rcp");
  run<OP_LDEXP, 1>("ldexp"); run<OP_LDEXP, 8>("ldexp");
  run<OP_RNDNE, 1>("rndne"); run<OP_RNDNE, 8>("rndne");
  run<OP_FMA32, 1>("fma32"); run<OP_FMA32, 8>("fma32");
  return 0;
}
