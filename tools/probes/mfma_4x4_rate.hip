// mfma_4x4_rate.hip -- VERDICT r4 #1(a): does v_mfma_f64_4x4x4_4b_f64 keep the f64 matrix rate on gfx950 with a 4-row N?
// Prints (i) cycles per instruction of v_mfma_f64_4x4x4f64 as ONE accumulator chain, as 2/4/8/16 independent chains and
// as a chain through the B operand (D of one instruction = B of the next: the substitution pattern of k_rows), next to
// the same three shapes of v_mfma_f64_16x16x4f64; (ii) the operand / result layout of the 4-block form, found with exact
// integer data (which lane holds A[b][i][k], B[b][k][j], D[b][i][j]).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_4x4_rate mfma_4x4_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef double d4 __attribute__((ext_vector_type(4)));

#define M44(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)
#define M16(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

template <int NACC>
__global__ void k44(double* out, unsigned long long* tm, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = M44(a, b, acc[i]);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { tm[0] = t1 - t0; tm[1] = r1 - r0; }
}
// chain through the B operand: b <- D (NCH independent such chains)
template <int NCH>
__global__ void k44b(double* out, unsigned long long* tm, int iters) {
  double b[NCH];
  for (int i = 0; i < NCH; ++i) b[i] = 1.0 + threadIdx.x * 1e-4 + i;
  double a = 1e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) b[i] = M44(a, b[i], 0.0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NCH; ++i) s += b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { tm[0] = t1 - t0; tm[1] = r1 - r0; }
}
template <int NACC>
__global__ void k16(double* out, unsigned long long* tm, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = M16(a, b, acc[i]);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { tm[0] = t1 - t0; tm[1] = r1 - r0; }
}
// 16x16x4 chain through the B operand: b <- D[r] (the close of a substitution tile in k_rows)
template <int NCH>
__global__ void k16b(double* out, unsigned long long* tm, int iters) {
  double b[NCH];
  for (int i = 0; i < NCH; ++i) b[i] = 1.0 + threadIdx.x * 1e-4 + i;
  double a = 1e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) { d4 z = {0, 0, 0, 0}; z = M16(a, b[i], z); b[i] = z[it & 3]; }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NCH; ++i) s += b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { tm[0] = t1 - t0; tm[1] = r1 - r0; }
}

// layout: out[l] = D of lane l for A = indicator(lane == la), B = indicator(lane == lb)
__global__ void klayout(double* out, int la, int lb) {
  const double a = (int)threadIdx.x == la ? 1.0 : 0.0, b = (int)threadIdx.x == lb ? 1.0 : 0.0;
  out[threadIdx.x] = M44(a, b, 0.0);
}

template <class K>
static void run(K kern, int nper, int blocks, int threads, const char* name, double flop_per) {
  double* out; unsigned long long* tm; hipMalloc(&out, (size_t)blocks * threads * 8); hipMalloc(&tm, 16);
  const int iters = 4000;
  kern<<<blocks, threads>>>(out, tm, iters); hipDeviceSynchronize();
  kern<<<blocks, threads>>>(out, tm, iters); hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, tm, 16, hipMemcpyDeviceToHost);
  const double cyc = (double)h[0] / ((double)iters * nper), us = h[1] * 0.01;
  // s_memtime ticks at 100 MHz on gfx950 (constant), s_memrealtime too: report ns and derive cycles at the shader clock
  printf("%-44s %3d blk x %3d thr: %7.2f ns/instr  = %6.1f flop/ns/SIMD  (memtime %.2f ticks/instr)\n", name, blocks, threads,
         us * 1e3 / ((double)iters * nper), flop_per / (us * 1e3 / ((double)iters * nper)), cyc);
  hipFree(out); hipFree(tm);
}

int main() {
  printf("== v_mfma_f64_4x4x4_4b (512 flop) vs v_mfma_f64_16x16x4 (2048 flop), one wave ==\n");
  run(k44<1>, 1, 1, 64, "4x4x4  1 accumulator chain", 512);
  run(k44<2>, 2, 1, 64, "4x4x4  2 accumulators", 512);
  run(k44<4>, 4, 1, 64, "4x4x4  4 accumulators", 512);
  run(k44<8>, 8, 1, 64, "4x4x4  8 accumulators", 512);
  run(k44<16>, 16, 1, 64, "4x4x4 16 accumulators", 512);
  run(k44b<1>, 1, 1, 64, "4x4x4  D->B chain x1", 512);
  run(k44b<4>, 4, 1, 64, "4x4x4  D->B chain x4", 512);
  run(k16<1>, 1, 1, 64, "16x16x4 1 accumulator chain", 2048);
  run(k16<4>, 4, 1, 64, "16x16x4 4 accumulators", 2048);
  run(k16b<1>, 1, 1, 64, "16x16x4 D->B chain x1", 2048);
  run(k16b<4>, 4, 1, 64, "16x16x4 D->B chain x4", 2048);
  printf("== whole chip, 4 waves per CU ==\n");
  run(k44<8>, 8, 256, 256, "4x4x4  8 accumulators", 512);
  run(k16<4>, 4, 256, 256, "16x16x4 4 accumulators", 2048);
  printf("== 2 waves per SIMD ==\n");
  run(k44<8>, 8, 256, 512, "4x4x4  8 accumulators", 512);
  run(k44<1>, 1, 256, 512, "4x4x4  1 accumulator chain", 512);

  // layout
  double* out; hipMalloc(&out, 64 * 8);
  double h[64];
  printf("== layout of the 4-block form: A lane la, B lane lb -> lanes with D = 1 ==\n");
  const int probes[][2] = {{0, 0}, {1, 0}, {4, 0}, {0, 1}, {0, 4}, {4, 4}, {5, 6}, {16, 16}, {16, 0}, {0, 16}, {21, 22}, {37, 38}, {53, 54}, {12, 3}, {3, 12}};
  for (auto& pr : probes) {
    klayout<<<1, 64>>>(out, pr[0], pr[1]); hipDeviceSynchronize();
    hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost);
    printf("A@%2d B@%2d :", pr[0], pr[1]);
    int n = 0;
    for (int l = 0; l < 64; ++l) if (h[l] != 0.0) { printf(" %d", l); ++n; }
    if (!n) printf(" (none)");
    printf("\n");
  }
  return 0;
}
