// How the f64 matrix pipe of one SIMD serves 1, 2, 3 co-resident waves (one workgroup of 256 / 512 / 768 threads on one
// CU: waves w, w+4, w+8 share a SIMD), each wave running ONE dependent chain of v_mfma_f64_16x16x4_f64 (the shape of the
// row kernels' triangular products), and the same with a VALU-only partner (f64 FMA chain) to see whether matrix and
// vector f64 work of DIFFERENT waves overlap.   hipcc -O3 --offload-arch=gfx950 mfma_waves.hip -o mfma_waves
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// mode 0: every wave an MFMA chain; mode 1: waves >= 4 run an f64 FMA chain (4 independent chains) instead
__global__ void k(double* out, unsigned long long* tm, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  d4 acc = {0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mode == 0 || wave < 4) {
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  } else {
    for (int it = 0; it < 4 * iters; ++it) {   // 16 f64 FMAs per MFMA of the partner: ~ the same time alone
      f0 = fma(f0, b, 1e-9); f1 = fma(f1, b, 1e-9); f2 = fma(f2, b, 1e-9); f3 = fma(f3, b, 1e-9);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + f0 + f1 + f2 + f3;
  if ((threadIdx.x & 63) == 0) tm[wave] = t1 - t0;
}

int main() {
  double* out; unsigned long long* tm; hipMalloc(&out, 1024 * 8); hipMalloc(&tm, 16 * 8);
  const int iters = 4000;
  for (int mode = 0; mode < 2; ++mode)
    for (int threads = 256; threads <= 768; threads += 256) {
      if (mode == 1 && threads == 256) continue;
      k<<<1, threads>>>(out, tm, iters, mode); hipDeviceSynchronize();
      k<<<1, threads>>>(out, tm, iters, mode); hipDeviceSynchronize();
      unsigned long long h[16]; hipMemcpy(h, tm, 16 * 8, hipMemcpyDeviceToHost);
      printf("%s, %d waves per SIMD: ", mode == 0 ? "all waves MFMA chains" : "waves 0-3 MFMA chain, the others f64 FMA chains (16 FMA per MFMA)", threads / 256);
      for (int w = 0; w < threads / 64; w += 4) printf(" wave %d: %.1f cycles per %s;", w, (double)h[w] / iters, (mode == 1 && w >= 4) ? "16 FMAs" : "MFMA");
      printf("   => SIMD: %.1f cycles per MFMA issued\n", (double)h[0] / iters / (mode == 0 ? threads / 256 : 1));
    }
  return 0;
}
