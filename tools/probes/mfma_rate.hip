#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, unsigned long long* tm, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { tm[0] = t1 - t0; tm[1] = r1 - r0; }
}
// f64 FMA dependent chain and exp throughput
__global__ void kfma(double* out, unsigned long long* tm, int iters) {
  double x = threadIdx.x * 1e-3, y = 1.000001;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) x = fma(x, y, 1e-9);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double e = 0; double z = threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) { e += exp(-z); z += 1e-3; }
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + e;
  if (threadIdx.x == 0 && blockIdx.x == 0) { tm[0] = t1 - t0; tm[1] = t2 - t1; }
}
template <int NACC> void run(int blocks, int threads, const char* name) {
  double* out; unsigned long long* tm; hipMalloc(&out, blocks * threads * 8); hipMalloc(&tm, 16);
  const int iters = 2000;
  k<NACC><<<blocks, threads>>>(out, tm, iters); hipDeviceSynchronize();
  k<NACC><<<blocks, threads>>>(out, tm, iters); hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, tm, 16, hipMemcpyDeviceToHost);
  double cyc = (double)h[0] / (iters * NACC), us = h[1] * 0.01;
  printf("%-28s blocks %4d x %4d thr: %.1f memtime-ticks/MFMA, %.1f ns/MFMA, tick rate %.2f GHz\n", name, blocks, threads, cyc, us * 1e3 / (iters * NACC), h[0] / (us * 1e3));
  hipFree(out); hipFree(tm);
}
int main() {
  run<1>(1, 64, "1 wave, 1 acc (dependent)");
  run<2>(1, 64, "1 wave, 2 acc");
  run<4>(1, 64, "1 wave, 4 acc");
  run<4>(1, 256, "4 waves/CU, 4 acc");
  run<4>(135, 256, "135 CUs x 4 waves, 4 acc");
  run<4>(256, 256, "256 CUs x 4 waves, 4 acc");
  run<4>(256, 512, "256 CUs x 8 waves, 4 acc");
  double* out; unsigned long long* tm; hipMalloc(&out, 256 * 8); hipMalloc(&tm, 16);
  kfma<<<1, 64>>>(out, tm, 4000); hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, tm, 16, hipMemcpyDeviceToHost);
  printf("f64 dependent FMA: %.1f ticks each; exp(double): %.1f ticks each (1 wave)\n", h[0] / 4000.0, h[1] / 4000.0);
  return 0;
}
