// standalone timing of tgp::k_gemm variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../tgp/pytorch_amd/csrc/tgp_gemm.hpp"
using namespace tgp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

float runl(bool ta, bool tb, GemmArgs g, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) if (int rc = launch_gemm(ta, tb, g, 0)) { printf("launch rc %d\n", rc); return -1; }
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch_gemm(ta, tb, g, 0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}
template <bool TA, bool TB>
float run(GemmArgs g, int reps) { return runl(TA, TB, g, reps); }

int main(int argc, char** argv) {
  const int n = 4096;  // buffers: 4096^2 doubles = 16.7M >= 15744*1024
  double *A, *B, *C, *V;
  CK(hipMalloc(&A, (size_t)n * n * 8)); CK(hipMalloc(&B, (size_t)n * n * 8)); CK(hipMalloc(&C, (size_t)n * n * 8 * 2));
  CK(hipMalloc(&V, (size_t)n * 8));
  std::vector<double> h((size_t)n * n);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
  CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(V, h.data(), n * 8, hipMemcpyHostToDevice));
  {
    // pipeline shapes: NC x MP x MP triangular (n-major layout), SYRK and T
    const int NC = argc > 1 ? atoi(argv[1]) : 15744, MP = 1024;
    const int LDP = argc > 2 ? atoi(argv[2]) : 0;   // leading-dimension padding of the x-major operands (doubles): 0 = the power-of-two stride
    struct { const char* name; bool ta, tb; int tri, xcd; } cs[] = {
      {"NT full        ", false, true, 0, 0}, {"NT full xcd1   ", false, true, 0, 1},
      {"NT triBU       ", false, true, TRI_B_UPPER, 0}, {"NT triBU xcd1  ", false, true, TRI_B_UPPER, 1},
      {"NN triBL       ", false, false, TRI_B_LOWER, 0}, {"NN triBL xcd1  ", false, false, TRI_B_LOWER, 1},
      {"TN full (m-maj)", true, false, 0, 0},
    };
    for (auto& c : cs) {
      GemmArgs g;
      if (c.ta) g = gemm_args(A, MP, B, NC, C, NC, MP, NC, MP, 1.0, 0.0, c.tri);  // m-major: C[MP][NC] = J^T-like [k][m] x Kc [k][NC]
      else g = gemm_args(A, MP + LDP, B, MP + LDP, C, MP + LDP, NC, MP, MP, 1.0, 0.0, c.tri);
      g.xcd = c.xcd;
      float t = c.ta ? run<true, false>(g, 10) : (c.tb ? run<false, true>(g, 10) : run<false, false>(g, 10));
      const double f = 2.0 * NC * (double)MP * MP * (c.tri ? 0.5625 : 1.0);
      printf("%s %.3f ms  %.1f TF/s (useful)\n", c.name, t, f / t / 1e9);
    }
    {
      // the NT product at the TN line's GRID SHAPE (8 tile rows x NC / 128 tile columns): small x-major A [MP][k], big x-major
      // B [NC][k] (op(B) = B^T), C [MP][NC] -- separates "x-major operands" from "128 x 8 grid" as the cause of the NT / TN gap
      for (int xcd : {0, 1}) {
        GemmArgs g = gemm_args(A, MP, B, MP, C, NC, MP, NC, MP, 1.0, 0.0, 0);
        g.xcd = xcd;
        float t = run<false, true>(g, 10);
        printf("NT full, m = MP, n = NC (swapped shape) xcd%d  %.3f ms  %.1f TF/s (useful)\n", xcd, t, 2.0 * NC * (double)MP * MP / t / 1e9);
      }
      // and the TN product at the NT lines' shape: C [NC][MP] = A^T ([k][NC] k-major, big) x B ([k][MP] k-major, small)
      for (int xcd : {0, 1}) {
        GemmArgs g = gemm_args(A, NC, B, MP, C, MP, NC, MP, MP, 1.0, 0.0, 0);
        g.xcd = xcd;
        float t = run<true, false>(g, 10);
        printf("TN full, m = NC, n = MP (the NT lines' shape) xcd%d  %.3f ms  %.1f TF/s (useful)\n", xcd, t, 2.0 * NC * (double)MP * MP / t / 1e9);
      }
    }
    for (int ks : {8, 12, 14, 16, 28}) for (int xcd : {2}) for (int sc = 0; sc < 2; ++sc) {
      GemmArgs g = gemm_args(A, MP, A, MP, C, MP, MP, MP, NC, 1.0, 1.0, TRI_C_LOWER);
      g.ksplit = ks; g.cz = (size_t)MP * MP; g.xcd = xcd; g.k_scale = sc ? V : nullptr;
      float t = run<true, false>(g, 10);
      printf("SYRK ks=%d xcd=%d kscale=%d  %.3f ms  %.1f TF/s (useful)\n", ks, xcd, sc, t, 2.0 * NC * (double)MP * MP * 0.5625 / t / 1e9);
    }
    for (int am = 0; am < 2; ++am) {
      GemmArgs g = gemm_args(A, MP, B, 128, C, 128, MP, 128, NC, 1.0, 1.0);
      g.ksplit = 32; g.cz = (size_t)MP * 128; g.xcd = 2; g.a_mul = am ? B : nullptr;
      float t = run<true, false>(g, 10);
      printf("T gemm a_mul=%d  %.3f ms  %.1f TF/s\n", am, t, 2.0 * NC * (double)MP * 128 / t / 1e9);
    }
  }
  return 0;
}
