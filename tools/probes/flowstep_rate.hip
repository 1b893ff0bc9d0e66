// Cycles of one StepTanh forward step body (4 nodes, staged) and of its pieces, one wave, registers only.
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../tgp/pytorch_amd/csrc -I../../include flowstep_rate.hip -o flowstep_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "tgp_dev.hpp"
using namespace tgp;
template <int WHAT, int NB>
__global__ void k(double* out, unsigned long long* tm, int iters, double c, double idt, double a, double bt) {
  double f[NB], g[NB];
  for (int u = 0; u < NB; ++u) { f[u] = 0.1 * u + 1e-3 * threadIdx.x; g[u] = 0; }
  unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    double e[NB];
    if (WHAT == 0) {  // whole step
      TGP_EACH(u, NB) e[u] = 2.0 * (f[u] - c) * idt;
      exp_fast_n<NB>(e);
      TGP_EACH(u, NB) e[u] += 1.0;
      rcp_fast_n<NB>(e);
      TGP_EACH(u, NB) { const double th = 1.0 - 2.0 * e[u]; g[u] += a + bt * th; f[u] += 1e-3 * th; }
    } else if (WHAT == 1) {  // exp only
      TGP_EACH(u, NB) e[u] = f[u];
      exp_fast_n<NB>(e);
      TGP_EACH(u, NB) f[u] = e[u] * 0.5;
    } else if (WHAT == 2) {  // rcp only
      TGP_EACH(u, NB) e[u] = f[u] + 1.0;
      rcp_fast_n<NB>(e);
      TGP_EACH(u, NB) f[u] = e[u];
    } else {  // scalar exp_fast per node (old form)
      TGP_EACH(u, NB) f[u] = exp_fast(f[u]) * 0.5;
    }
  }
  unsigned long long t1 = clock64();
  double s = 0;
  for (int u = 0; u < NB; ++u) s += f[u] + g[u];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) tm[0] = t1 - t0;
}
template <int WHAT, int NB> void run(const char* name) {
  double* out; unsigned long long* tm; hipMalloc(&out, 64 * 8); hipMalloc(&tm, 8);
  const int iters = 1000;
  for (int w = 0; w < 2; ++w) { k<WHAT, NB><<<1, 64>>>(out, tm, iters, 0.3, 1.7, 0.1, 0.2); hipDeviceSynchronize(); }
  unsigned long long h; hipMemcpy(&h, tm, 8, hipMemcpyDeviceToHost);
  printf("%-22s NB %d: %.0f cycles per iteration (%.0f per node)\n", name, NB, (double)h / iters, (double)h / iters / NB);
  hipFree(out); hipFree(tm);
}
int main() {
  run<0, 4>("tanh step"); run<0, 8>("tanh step"); run<0, 2>("tanh step"); run<0, 1>("tanh step");
  run<1, 4>("exp_fast_n"); run<1, 8>("exp_fast_n"); run<1, 1>("exp_fast_n");
  run<2, 4>("rcp_fast_n"); run<2, 8>("rcp_fast_n");
  run<3, 4>("exp_fast scalar x NB");
  return 0;
}
