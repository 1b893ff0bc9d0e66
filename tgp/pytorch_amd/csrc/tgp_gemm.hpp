// tgp_gemm.hpp -- tiled float64 GEMM on v_mfma_f64_16x16x4_f64 for the general-M path (M > 128), where the
// per-row operands no longer fit in registers and A = L^-1 K_MN is materialised per row chunk (tgp_big.hip).
//
//   C[m x n] = epilogue( alpha * op(A)[m x k] * op(B)[k x n] )           (row-major, leading dimensions lda/ldb/ldc)
//
// Workgroup = 256 threads = 4 waves (2 x 2), output tile 128 x 128, each wave 64 x 64 = 4 x 4 MFMA tiles
// (16 accumulators); k advances 16 per stage through two LDS buffers per operand, stored k-major
// ([16][128 + pad]) so that every fragment read is 16 lanes x 8 contiguous bytes and the transposing store of a
// non-transposed operand is conflict-free (pad = 1).  Global loads of stage s+1 are in flight while stage s
// feeds the matrix pipe.  All dimensions must be multiples of 128 (m, n) / 16 (k): callers pad.
//
// `tri` trims the k range per output tile for triangular operands (the Cholesky / inverse factors):
//   TRI_A_LOWER: op(A) lower triangular  -> k < i_tile_end        TRI_A_UPPER: op(A) upper -> k >= i_tile_begin
//   TRI_B_LOWER: op(B) lower triangular  -> k >= j_tile_begin     TRI_B_UPPER: op(B) upper -> k < j_tile_end
//   TRI_C_LOWER: only output tiles with i_tile >= j_tile are computed (symmetric results)
// Operand modifiers (fused elementwise work, so that no N-sized temporary is written for them):
//   a_mul   : op(A) element is multiplied by the element of a second matrix with A's layout   (K_bar o K)
//   k_scale : op(A)[i][k] is multiplied by k_scale[k]                                          (A diag(v_bar) A^T)
// Epilogue:  x = alpha*acc + gamma*add[i][j];  x *= col_scale[j];  x += rowv[i]*colv[j];  C = x + beta*C
// Split-K:   gridDim.z slabs of the k range; slab z writes C + z*cz (the caller reduces the slabs, fixed order).
#pragma once
#include "tgp_dev.hpp"

namespace tgp {

enum { TRI_A_LOWER = 1, TRI_A_UPPER = 2, TRI_B_LOWER = 4, TRI_B_UPPER = 8, TRI_C_LOWER = 16 };

struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  int m, n, k, lda, ldb, ldc;
  double alpha, beta;
  int tri;
  // optional (nullptr / 0 = off)
  const double* a_mul;
  const double* k_scale;
  const double* add;   // ld = ldadd
  int ldadd;
  double gamma;
  const double* col_scale;
  const double* rowv;
  const double* colv;
  int ksplit;          // >= 1
  size_t cz;           // doubles between slab outputs
};

inline GemmArgs gemm_args(const double* A, int lda, const double* B, int ldb, double* C, int ldc, int m, int n, int k,
                          double alpha = 1.0, double beta = 0.0, int tri = 0) {
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.m = m; g.n = n; g.k = k; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta; g.tri = tri;
  g.a_mul = nullptr; g.k_scale = nullptr; g.add = nullptr; g.ldadd = 0; g.gamma = 0.0;
  g.col_scale = nullptr; g.rowv = nullptr; g.colv = nullptr; g.ksplit = 1; g.cz = 0;
  return g;
}

#define GT 128        /* output tile edge */
#define GK 16         /* k per stage */
#define GLD (GT + 1)  /* LDS row stride (doubles) */
#define GEMM_LDS_BYTES (4 * GK * GLD * sizeof(double))

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void k_gemm(GemmArgs g) {
  // 4 stage buffers of 16 x 129 doubles = 66 KB (dynamic: above the 64 KB static limit).  GLD odd keeps the
  // transposing stores conflict-free; the fragment reads are then 2-way conflicted, ~1/8 of the MFMA time.
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  double (*As)[GK * GLD] = reinterpret_cast<double (*)[GK * GLD]>(gemm_smem);
  double (*Bs)[GK * GLD] = reinterpret_cast<double (*)[GK * GLD]>(gemm_smem + 2 * GK * GLD * sizeof(double));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  // heaviest tile rows first when the k range grows with the tile row
  const int by = (g.tri & TRI_A_LOWER) ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
  const int i0 = by * GT, j0 = blockIdx.x * GT;
  if ((g.tri & TRI_C_LOWER) && j0 > i0) return;
  int kb = 0, ke = g.k;
  if (g.tri & TRI_A_LOWER) ke = min(ke, i0 + GT);
  if (g.tri & TRI_A_UPPER) kb = max(kb, i0);
  if (g.tri & TRI_B_LOWER) kb = max(kb, j0);
  if (g.tri & TRI_B_UPPER) ke = min(ke, j0 + GT);
  double* __restrict__ C = g.C;
  if (g.ksplit > 1) {
    const int stages = (ke - kb + GK - 1) / GK, per = (stages + g.ksplit - 1) / g.ksplit;
    const int z = blockIdx.z;
    const int b2 = kb + z * per * GK, e2 = min(ke, b2 + per * GK);
    kb = b2; ke = e2;
    C += (size_t)z * g.cz;
  }
  const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
  d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = {0, 0, 0, 0};

  // staging: each operand stage is 16 x 128 doubles = 2048 values = 8 per thread
  double sa[8], sb[8];
  const double* __restrict__ Ag = g.A;
  const double* __restrict__ Bg = g.B;
  auto load = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + 256 * u;
      size_t ia;
      int kk;
      if (TA) {  // A stored [k][m]: row k0 + e/128, col i0 + e%128  (coalesced along m)
        kk = k0 + (e >> 7);
        ia = (size_t)kk * g.lda + i0 + (e & 127);
      } else {   // A stored [m][k]: row i0 + e/16, col k0 + e%16     (128-byte segments along k)
        kk = k0 + (e & 15);
        ia = (size_t)(i0 + (e >> 4)) * g.lda + kk;
      }
      double x = Ag[ia];
      if (g.a_mul) x *= g.a_mul[ia];
      if (g.k_scale) x *= g.k_scale[kk];
      sa[u] = x;
      if (TB) {  // B stored [n][k]
        sb[u] = Bg[(size_t)(j0 + (e >> 4)) * g.ldb + k0 + (e & 15)];
      } else {   // B stored [k][n]
        sb[u] = Bg[(size_t)(k0 + (e >> 7)) * g.ldb + j0 + (e & 127)];
      }
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + 256 * u;
      if (TA) As[buf][(e >> 7) * GLD + (e & 127)] = sa[u];
      else As[buf][(e & 15) * GLD + (e >> 4)] = sa[u];
      if (TB) Bs[buf][(e & 15) * GLD + (e >> 4)] = sb[u];
      else Bs[buf][(e >> 7) * GLD + (e & 127)] = sb[u];
    }
  };

  if (kb < ke) {
    load(kb);
    int buf = 0;
    for (int k0 = kb; k0 < ke; k0 += GK) {
      store(buf);
      __syncthreads();
      if (k0 + GK < ke) load(k0 + GK);
#pragma unroll
      for (int s = 0; s < GK / 4; ++s) {
        double af[4], bf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) af[a] = As[buf][(4 * s + q) * GLD + wi + 16 * a + r];
#pragma unroll
        for (int b = 0; b < 4; ++b) bf[b] = Bs[buf][(4 * s + q) * GLD + wj + 16 * b + r];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = TGP_MFMA(af[a], bf[b], acc[a][b]);
      }
      buf ^= 1;
    }
  }
  // epilogue: C/D layout row = q + 4 rr, col = r
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int col = j0 + wj + 16 * b + r;
    const double cs = g.col_scale ? g.col_scale[col] : 1.0;
    const double cv = g.colv ? g.colv[col] : 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = i0 + wi + 16 * a + q + 4 * rr;
        double x = g.alpha * acc[a][b][rr];
        if (g.add) x += g.gamma * g.add[(size_t)row * g.ldadd + col];
        x *= cs;
        if (g.rowv) x += g.rowv[row] * cv;
        double* c = C + (size_t)row * g.ldc + col;
        *c = (g.beta == 0.0) ? x : x + g.beta * *c;
      }
  }
}

int launch_gemm(bool ta, bool tb, const GemmArgs& g, hipStream_t st);  // tgp_big.hip

}  // namespace tgp
