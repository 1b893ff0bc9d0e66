// sc1_rate.hip -- cost of agent-scope-coherent (sc1) stores / loads issued by ONE wave, against plain ones.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 sc1_rate.hip -o sc1_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int MODE>
__global__ void k(double* buf, unsigned long long* tm, int n) {
  const int lane = threadIdx.x & 63;
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
  __syncthreads();
  if (threadIdx.x >= 64) return;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  double acc = 0.0;
  for (int it = 0; it < n; ++it) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = lds[(it * 8 + u) * 64 % 4096 + lane];
    double* p = buf + (size_t)it * 8 * 112 + (lane >> 4) * 112 + (lane & 15);
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u * 4 * 112] = v[u];
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) st_agent(p + u * 4 * 112, v[u]);
    } else if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += p[u * 4 * 112];
    } else {
      double w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = ld_agent(p + u * 4 * 112);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += w[u];
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) { tm[0] = t1 - t0; tm[1] = t2 - t0; }
  if (acc == 12345.678) buf[0] = acc;
}
template <int MODE> void run(const char* name, double* buf, unsigned long long* tm, int n) {
  for (int w = 0; w < 3; ++w) { k<MODE><<<1, 256>>>(buf, tm, n); hipDeviceSynchronize(); }
  unsigned long long h[2]; hipMemcpy(h, tm, 16, hipMemcpyDeviceToHost);
  printf("%-28s n=%3d groups of 8: issue %.2f us, +drain %.2f us  (%.0f ns per 8-store/load group)\n", name, n, h[0] * 0.01, h[1] * 0.01, h[1] * 10.0 / n);
}
int main() {
  double* buf; unsigned long long* tm;
  hipMalloc(&buf, 64 << 20); hipMalloc(&tm, 16); hipMemset(buf, 0, 64 << 20);
  for (int n : {1, 8, 32}) {
    run<0>("plain stores", buf, tm, n);
    run<1>("sc1 (agent atomic) stores", buf, tm, n);
    run<2>("plain loads", buf, tm, n);
    run<3>("sc1 (agent atomic) loads", buf, tm, n);
  }
  return 0;
}
