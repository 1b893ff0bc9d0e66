// What does the FIRST global load of a launch cost?  Kernel W writes a buffer (plain stores); kernel R (launched right behind
// it, 256 workgroups of one wave) times, per workgroup, a chain of dependent loads with s_memrealtime (100 MHz):
//   t0: the workgroup's first load (a line nobody in this launch has touched)       t1: the next line of the same 4 KB page
//   t2: a line 64 KB away (same 2 MB fragment)     t3: a line 4 MB away     t4: the first line again (now in L2 / L1)
//   t5: a line another workgroup of the same XCD loaded first (id + 8)
// hipcc --offload-arch=gfx950 -O3 -o first_load_latency first_load_latency.hip && ./first_load_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k_write(double* buf, size_t n) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = (double)(i & 1023);
}
__device__ __forceinline__ double ld(const double* p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__global__ void k_read(const double* buf, double* out, int mode) {
  if (threadIdx.x != 0) return;
  const size_t base = (size_t)blockIdx.x * (1 << 16) / 8 * 3;      // 192 KB apart: every workgroup its own lines
  const double* p = buf + base;
  unsigned long long t[8];
  double acc = 0.0;
  t[0] = __builtin_amdgcn_s_memrealtime();
  acc += ld(p);
  t[1] = __builtin_amdgcn_s_memrealtime();
  acc += ld(p + 16 + (long)acc % 2);
  t[2] = __builtin_amdgcn_s_memrealtime();
  acc += ld(p + 8192 + (long)acc % 2);
  t[3] = __builtin_amdgcn_s_memrealtime();
  acc += ld(p + 524288 + (long)acc % 2);
  t[4] = __builtin_amdgcn_s_memrealtime();
  acc += ld(p + (long)acc % 2);
  t[5] = __builtin_amdgcn_s_memrealtime();
  const size_t other = (size_t)((blockIdx.x + 8) % gridDim.x) * (1 << 16) / 8 * 3;
  acc += ld(buf + other + (long)acc % 2);
  t[6] = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < 6; ++i) out[blockIdx.x * 8 + i] = (double)(t[i + 1] - t[i]);
  out[blockIdx.x * 8 + 7] = acc;
}
int main() {
  const size_t n = (size_t)96 << 20;   // 768 MB of doubles? no: 96 Mi doubles = 768 MB is too much; use 48 Mi
  const size_t nd = (size_t)12 << 20;  // 12 Mi doubles = 96 MB
  double *buf, *out;
  hipMalloc(&buf, nd * 8); hipMalloc(&out, 256 * 8 * 8);
  std::vector<double> h(256 * 8);
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, 0, buf, nd);
    hipLaunchKernelGGL(k_read, dim3(256), dim3(64), 0, 0, buf, out, 0);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    const char* names[6] = {"first load", "next line, same page", "64 KB away", "4 MB away", "first line again", "line of workgroup id+8"};
    printf("rep %d (units of 10 ns, median / min / max over 256 workgroups)\n", rep);
    for (int i = 0; i < 6; ++i) {
      std::vector<double> v; for (int b = 0; b < 256; ++b) v.push_back(h[b * 8 + i]);
      std::sort(v.begin(), v.end());
      printf("  %-26s %6.0f / %6.0f / %6.0f\n", names[i], v[128], v[0], v[255]);
    }
  }
  (void)n;
  return 0;
}
