// (Measured at commit 9d413c0, whose k_gemm had the tile orders `xcd 5` / `xcd 6` and whose launcher took a triangular op(A); the
//  shipped library is round 5's GEMM again -- see profiles/NOTES.md round 6.)
// Round 6 (VERDICT r5 #1c): the chunk products of the general-M path in the layout the pipeline ships (chunk matrices [NC][MP],
// "n-major": the big operand is op(A), x-major) against the m-major layout (chunk matrices [MP][NC]: the small M x M factor is
// op(A), the big operand op(B) stored [k][n], k-major), with the triangular trimming and tile orders each would launch with.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 gemm_mmajor.hip -o gemm_mmajor -L../../tgp/pytorch_amd -ltgp_hip -Wl,-rpath,'$ORIGIN/../../tgp/pytorch_amd'
//   ./gemm_mmajor [NC]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../tgp/pytorch_amd/csrc/tgp_gemm.hpp"
using namespace tgp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

static float runl(bool ta, bool tb, GemmArgs g, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) if (int rc = launch_gemm(ta, tb, g, 0)) { printf("launch rc %d\n", rc); return -1; }
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch_gemm(ta, tb, g, 0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int MP = 1024;
  const size_t cap = (size_t)16384 * MP;
  double *Kc, *A, *B, *J, *V;
  CK(hipMalloc(&Kc, cap * 8)); CK(hipMalloc(&A, cap * 8)); CK(hipMalloc(&B, cap * 8 * 2));
  CK(hipMalloc(&J, (size_t)MP * MP * 8)); CK(hipMalloc(&V, 16384 * 8));
  std::vector<double> h(cap);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  CK(hipMemcpy(Kc, h.data(), cap * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(A, h.data(), cap * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(J, h.data(), (size_t)MP * MP * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(V, h.data(), 16384 * 8, hipMemcpyHostToDevice));
  std::vector<int> ncs;
  for (int i = 1; i < argc; ++i) ncs.push_back(atoi(argv[i]));
  if (ncs.empty()) { ncs.push_back(15744); ncs.push_back(10112); }
  for (int NC : ncs) {
    const double tri_f = 2.0 * NC * (double)MP * MP * 0.5625;   // useful flops of a triangular product (8 tile columns: 36 / 64)
    printf("---- NC = %d, M = %d (TF/s of useful flops) ----\n", NC, MP);
    // ---- shipped layout: chunk matrices [NC][MP] ----
    {
      GemmArgs g = gemm_args(Kc, MP, J, MP, A, MP, NC, MP, MP, 1.0, 0.0, TRI_B_UPPER); g.xcd = 1;
      float t = runl(false, true, g, 10);
      printf("n-major  A' = K' J^T   (NT, tri B upper, xcd 1)      %.3f ms  %.1f\n", t, tri_f / t / 1e9);
      g = gemm_args(A, MP, J, MP, B, MP, NC, MP, MP, 1.0, 0.0, TRI_B_LOWER); g.xcd = 1;
      t = runl(false, false, g, 10);
      printf("n-major  B' = A' Lq    (NN, tri B lower, xcd 1)      %.3f ms  %.1f\n", t, tri_f / t / 1e9);
      for (int ks : {14, 28}) {
        g = gemm_args(A, MP, A, MP, B, MP, MP, MP, NC, 1.0, 1.0, TRI_C_LOWER);
        g.ksplit = ks; g.cz = (size_t)MP * MP; g.xcd = 2; g.k_scale = V;
        t = runl(true, false, g, 10);
        printf("n-major  G SYRK        (TN, split-K %2d, k_scale)     %.3f ms  %.1f\n", ks, t, tri_f / t / 1e9);
      }
    }
    // ---- m-major layout: chunk matrices [MP][NC] ----
    for (int xcd : {0, 5}) {
      // A'^T = J K'^T : op(A) = J lower, stored [m][k] (x-major, small) or J^T stored [k][m] (k-major, TA)
      GemmArgs g = gemm_args(J, MP, Kc, NC, A, NC, MP, NC, MP, 1.0, 0.0, TRI_A_LOWER); g.xcd = xcd;
      float t = runl(false, false, g, 10);
      printf("m-major  A'^T = J K'^T   (NN, tri A lower, xcd %d)    %.3f ms  %.1f\n", xcd, t, tri_f / t / 1e9);
      g = gemm_args(J, MP, Kc, NC, A, NC, MP, NC, MP, 1.0, 0.0, TRI_A_LOWER); g.xcd = xcd;
      t = runl(true, false, g, 10);
      printf("m-major  A'^T = (J^T)^T K'^T (TN, tri A lower, xcd %d) %.3f ms  %.1f\n", xcd, t, tri_f / t / 1e9);
      g = gemm_args(J, MP, A, NC, B, NC, MP, NC, MP, 1.0, 0.0, TRI_A_UPPER); g.xcd = xcd;
      t = runl(true, false, g, 10);
      printf("m-major  B'^T = Lq^T A'^T  (TN, tri A upper, xcd %d)   %.3f ms  %.1f\n", xcd, t, tri_f / t / 1e9);
      g = gemm_args(J, MP, A, NC, B, NC, MP, NC, MP, 1.0, 0.0, TRI_A_UPPER); g.xcd = xcd;
      t = runl(false, false, g, 10);
      printf("m-major  B'^T = Lq^T A'^T  (NN, tri A upper, xcd %d)   %.3f ms  %.1f\n", xcd, t, tri_f / t / 1e9);
    }
    for (int ks : {14, 28}) {
      // G = A'^T diag(v) A' with A'^T stored [MP][NC]: both operands x-major, k = NC
      GemmArgs g = gemm_args(A, NC, A, NC, B, MP, MP, MP, NC, 1.0, 1.0, TRI_C_LOWER);
      g.ksplit = ks; g.cz = (size_t)MP * MP; g.xcd = 2; g.k_scale = V;
      float t = runl(false, true, g, 10);
      printf("m-major  G SYRK        (NT, split-K %2d, k_scale)     %.3f ms  %.1f\n", ks, t, tri_f / t / 1e9);
    }
  }
  return 0;
}
