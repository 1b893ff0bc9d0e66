// Probe: verify v_mfma_f64_16x16x4_f64 operand/result lane maps on gfx950 with exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* C, int K) {
  // A: 16 x K row-major, B: K x 16 row-major, C: 16x16 row-major
  int l = threadIdx.x;
  d4 acc = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 4) {
    double a = A[(l & 15) * K + k0 + (l >> 4)];
    double b = B[(k0 + (l >> 4)) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
// chained: D = A2 * (A1*B) using accumulator regs directly as B operand (k-step r <- reg r)
__global__ void k2(const double* A1, const double* A2, const double* B, double* C) {
  int l = threadIdx.x;
  d4 acc = {0, 0, 0, 0};
  for (int k0 = 0; k0 < 16; k0 += 4) {
    double a = A1[(l & 15) * 16 + k0 + (l >> 4)];
    double b = B[(k0 + (l >> 4)) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  d4 acc2 = {0, 0, 0, 0};
  for (int r = 0; r < 4; ++r) {
    double a = A2[(l & 15) * 16 + 4 * r + (l >> 4)];
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[r], acc2, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc2[r];
}
__global__ void katom(double* p) { atomicAdd(p, 1.0); unsafeAtomicAdd(p + 1, 2.0); }
int main() {
  const int K = 8;
  std::vector<double> A(16 * K), B(K * 16), C(256), R(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < K; ++k) A[i * K + k] = (i * 3 + k * 7) % 11 - 5;
  for (int k = 0; k < K; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k * 5 + j * 2 + k * j) % 13 - 6;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < K; ++k) R[i * 16 + j] += A[i * K + k] * B[k * 16 + j];
  double *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, 256 * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dC, K);
  hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) bad += (C[i] != R[i]);
  printf("mfma_f64 layout: %s (bad=%d)\n", bad ? "FAIL" : "OK", bad);
  // chained
  std::vector<double> A1(256), A2(256), B2(256), R1(256, 0.0), R2(256, 0.0);
  for (int i = 0; i < 256; ++i) { A1[i] = (i * 7) % 5 - 2; A2[i] = (i * 11 + i / 16) % 7 - 3; B2[i] = (i * 13 + 3 * (i / 16)) % 9 - 4; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 16; ++k) R1[i * 16 + j] += A1[i * 16 + k] * B2[k * 16 + j];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 16; ++k) R2[i * 16 + j] += A2[i * 16 + k] * R1[k * 16 + j];
  double *dA1, *dA2, *dB2;
  hipMalloc(&dA1, 2048); hipMalloc(&dA2, 2048); hipMalloc(&dB2, 2048);
  hipMemcpy(dA1, A1.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dA2, A2.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB2, B2.data(), 2048, hipMemcpyHostToDevice);
  k2<<<1, 64>>>(dA1, dA2, dB2, dC);
  hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost);
  bad = 0; for (int i = 0; i < 256; ++i) bad += (C[i] != R2[i]);
  printf("acc-as-B chaining: %s (bad=%d)\n", bad ? "FAIL" : "OK", bad);
  double h[2] = {0, 0}; hipMemcpy(dC, h, 16, hipMemcpyHostToDevice);
  katom<<<4, 64>>>(dC); hipMemcpy(h, dC, 16, hipMemcpyDeviceToHost);
  printf("atomics: %g %g (expect 256 512)\n", h[0], h[1]);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("dev %s CUs %d clock %d kHz lds/block %zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  return 0;
}
