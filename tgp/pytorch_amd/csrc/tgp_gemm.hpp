// tgp_gemm.hpp -- tiled float64 GEMM on v_mfma_f64_16x16x4_f64 for the general-M path (M > 128), where the
// per-row operands no longer fit in registers and A = L^-1 K_MN is materialised per row chunk (tgp_big.hip).
//
//   C[m x n] = epilogue( alpha * op(A)[m x k] * op(B)[k x n] )           (row-major, leading dimensions lda/ldb/ldc)
//
// Workgroup = 256 threads = 4 waves (2 x 2), output tile 128 x 128, each wave 64 x 64 = 4 x 4 MFMA tiles
// (16 accumulators); k advances 16 per stage through two LDS buffers per operand.  An operand stored k-major in
// memory keeps that layout in LDS ([16][144]); an operand stored x-major is kept x-major ([128][17]) -- both give
// conflict-free stores and natural-rate fragment reads without a transposing store.  Global loads (16 bytes per
// lane) of stage s+1 are in flight while stage s feeds the matrix pipe.  All dimensions must be multiples of
// 128 (m, n) / 16 (k) and leading dimensions even: callers pad.
//
// `tri` trims the k range per output tile for triangular operands (the Cholesky / inverse factors):
//   TRI_A_LOWER: op(A) lower triangular  -> k < i_tile_end        TRI_A_UPPER: op(A) upper -> k >= i_tile_begin
//   TRI_B_LOWER: op(B) lower triangular  -> k >= j_tile_begin     TRI_B_UPPER: op(B) upper -> k < j_tile_end
//   TRI_C_LOWER: only output tiles with i_tile >= j_tile are computed (symmetric results)
// Operand modifiers (fused elementwise work, so that no N-sized temporary is written for them):
//   a_mul   : op(A) element is multiplied by the element of a second matrix with A's layout   (K_bar o K)
//   k_scale : op(A)[i][k] is multiplied by k_scale[k]                                          (A diag(v_bar) A^T)
// Epilogue:  x = alpha*acc + gamma*add[i][j];  x *= col_scale[j] * row_scale[i];  x += rowv[i]*colv[j];  C = x + beta*C
//            (beta and add are mutually exclusive)
// Split-K:   gridDim.z slabs of the k range; slab z writes C + z*cz (the caller reduces the slabs, fixed order).
// XCD placement: consecutive workgroup ids go round-robin over the 8 XCDs (private 4 MB L2 each), so the natural
//   id -> tile order spreads the 8 workgroups that share one operand block over 8 different L2s.  `xcd` remaps ids so
//   that sharers carry the same id mod 8 (bijective; pure speed choice, results do not depend on it).
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
  const double* row_scale;
  const double* rowv;
  const double* colv;
  int ksplit;          // >= 1
  size_t cz;           // doubles between slab outputs
  int pair;            // set by the launcher: triangular op(B), column tiles j and n/128-1-j handled by one workgroup
  int xcd;             // workgroup -> tile mapping: 0 natural, 1 all column tiles of a tile ROW share an XCD (they
                       // re-read the same op(A) rows), 2 all tiles of a k SLAB share an XCD (split-K operands),
                       // 3 (set by the launcher for TRI_C_LOWER + split-K) compact lower-triangle enumeration,
                       // 4 (set by the launcher for an unpaired triangular op(B)) per-XCD heavy-column-first order
};

inline GemmArgs gemm_args(const double* A, int lda, const double* B, int ldb, double* C, int ldc, int m, int n, int k,
                          double alpha = 1.0, double beta = 0.0, int tri = 0) {
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.m = m; g.n = n; g.k = k; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta; g.tri = tri;
  g.a_mul = nullptr; g.k_scale = nullptr; g.add = nullptr; g.ldadd = 0; g.gamma = 0.0;
  g.col_scale = nullptr; g.row_scale = nullptr; g.rowv = nullptr; g.colv = nullptr; g.ksplit = 1; g.cz = 0; g.xcd = 0; g.pair = 0;
  return g;
}

#define GT 128        /* output tile edge */
#define GK 16         /* k per stage */
#define GLDK 144      /* k-major LDS stage [16][144]: row stride = 16 mod 32 doubles -> the 4 k-rows of a fragment read
                         land on disjoint bank halves (natural ds_read_b64 rate), b128 stores stay aligned */
#define GLDX 17       /* x-major LDS stage [128][17]: odd stride, conflict-free stores of 128-byte global row segments */
#define GSTAGE 2304   /* doubles per stage buffer = max(16*144, 128*17) */
#define GEMM_LDS_BYTES (4 * GSTAGE * sizeof(double))

typedef double d2 __attribute__((ext_vector_type(2)));

// Global-memory accesses of gemm_tile, with the address space spelled out.  The tile function re-reads its arguments from
// the kernarg segment, so the operand pointers are plain (generic) pointers to the compiler and it emitted FLAT loads --
// which count against lgkmcnt as well as vmcnt: every `s_waitcnt lgkmcnt(0)` in front of an MFMA batch (meant for the
// LDS fragment reads) also waited for the NEXT stage's operand loads, i.e. the global latency the double buffering is
// there to hide was paid in every stage.  global_load / global_store only touch vmcnt.
typedef const d2 __attribute__((address_space(1)))* gemm_gp2;
typedef const double __attribute__((address_space(1)))* gemm_gp1;
typedef double __attribute__((address_space(1)))* gemm_gpw;
__device__ __forceinline__ d2 gemm_ld2(const double* p) { return *(gemm_gp2)(p); }
__device__ __forceinline__ double gemm_ld1(const double* p) { return *(gemm_gp1)(p); }
__device__ __forceinline__ void gemm_st1(double* p, double x) { *(gemm_gpw)(p) = x; }

// One operand stage (16 k x 128 x) global -> registers -> LDS.  KMAJ: stored [k][x] in memory (coalesced 1 KB rows);
// else stored [x][k] (128-byte row segments, 8 rows per wave load).  16-byte loads throughout.  Addresses are
// (uniform base of the stage) + (per-thread 32-bit offset that never changes): one VGPR per operand, the rest SGPRs.
template <bool KMAJ>
__device__ __forceinline__ int gemm_voff(int ld, int tid) {
  return KMAJ ? (tid >> 6) * ld + 2 * (tid & 63) : (tid >> 3) * ld + 2 * (tid & 7);
}
template <bool KMAJ>
__device__ __forceinline__ void gemm_load(const double* __restrict__ G, int ld, int x0, int k0, int voff, d2 (&s)[4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const double* __restrict__ b = KMAJ ? G + (size_t)(k0 + 4 * u) * ld + x0 : G + (size_t)(x0 + 32 * u) * ld + k0;
    s[u] = gemm_ld2(b + voff);
  }
}
// modifier values of the same stage (raw; multiplied in at LDS-store time so that the loads stay in flight
// across the MFMA block): sm = a_mul elements, sk = k_scale entries
template <bool KMAJ>
__device__ __forceinline__ void gemm_load_mod(const double* __restrict__ mul, const double* __restrict__ ksc, int ld, int x0,
                                              int k0, int voff, int tid, d2 (&sm)[4], d2 (&sk)[4]) {
  const d2 one = {1.0, 1.0};
  if (mul) gemm_load<KMAJ>(mul, ld, x0, k0, voff, sm);
  else {
#pragma unroll
    for (int u = 0; u < 4; ++u) sm[u] = one;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (ksc) {
      if (KMAJ) { const double t = gemm_ld1(ksc + k0 + (tid >> 6) + 4 * u); sk[u] = d2{t, t}; }
      else sk[u] = gemm_ld2(ksc + k0 + 2 * (tid & 7));
    } else {
      sk[u] = one;
    }
  }
}
template <bool KMAJ>
__device__ __forceinline__ void gemm_store(double* __restrict__ L, int tid, const d2 (&s)[4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u;
    if (KMAJ) {
      *reinterpret_cast<d2*>(L + (e >> 6) * GLDK + 2 * (e & 63)) = s[u];
    } else {
      double* d = L + (e >> 3) * GLDX + 2 * (e & 7);
      d[0] = s[u][0];
      d[1] = s[u][1];
    }
  }
}
// fragment element (k, x) of a stage
template <bool KMAJ>
__device__ __forceinline__ double gemm_frag(const double* __restrict__ L, int k, int x) {
  return KMAJ ? L[k * GLDK + x] : L[x * GLDX + k];
}

// MOD: a_mul / k_scale in use; EPI: the epilogue reads beta*C or gamma*add (separate instantiations: the plain
// GEMM keeps its registers).
//
// One 128 x 128 output tile.  Kept out of line on purpose: the kernel calls it once or twice (tile pairing) and the
// register allocator, given the two calls inlined in a loop, spilled ~80 VGPRs of accumulator state.
template <bool TA, bool TB, bool MOD, bool EPI>
__device__ __attribute__((noinline)) void gemm_tile(uint64_t kernarg, int i0, int j0, int bz, double* As, double* Bs) {
  // the kernel's only parameter, re-read from the kernarg segment with scalar loads instead of travelling through
  // the call in vector registers or scratch (the address arrives in VGPRs: make it provably uniform first)
  union { GemmArgs g; uint32_t w[sizeof(GemmArgs) / 4]; } ka;
  {
    const uint32_t klo = __builtin_amdgcn_readfirstlane((uint32_t)kernarg);
    const uint32_t khi = __builtin_amdgcn_readfirstlane((uint32_t)(kernarg >> 32));
    const __attribute__((address_space(4))) uint32_t* kp =
        (const __attribute__((address_space(4))) uint32_t*)(((uint64_t)khi << 32) | klo);
#pragma unroll
    for (unsigned i = 0; i < sizeof(GemmArgs) / 4; ++i) ka.w[i] = kp[i];
  }
  const GemmArgs& g = ka.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
  const int voa = gemm_voff<TA>(g.lda, tid), vob = gemm_voff<!TB>(g.ldb, tid);
  {
    if ((g.tri & TRI_C_LOWER) && j0 > i0) return;
    int kb = 0, ke = g.k;
    if (g.tri & TRI_A_LOWER) ke = min(ke, i0 + GT);
    if (g.tri & TRI_A_UPPER) kb = max(kb, i0);
    if (g.tri & TRI_B_LOWER) kb = max(kb, j0);
    if (g.tri & TRI_B_UPPER) ke = min(ke, j0 + GT);
    double* __restrict__ C = g.C;
    if (g.ksplit > 1) {
      const int stages = (ke - kb + GK - 1) / GK, per = (stages + g.ksplit - 1) / g.ksplit;
      const int b2 = kb + bz * per * GK, e2 = min(ke, b2 + per * GK);
      kb = b2; ke = e2;
      C += (size_t)bz * g.cz;
    }
    d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = {0, 0, 0, 0};

    // op(A)[i][k]: TA -> A stored [k][m] (k-major); else stored [m][k].  op(B)[k][j]: TB -> B stored [n][k]; else k-major.
    d2 sa[4], sb[4], sm[4], sk[4];
    if (kb < ke) {
      gemm_load<TA>(g.A, g.lda, i0, kb, voa, sa);
      if (MOD) gemm_load_mod<TA>(g.a_mul, g.k_scale, g.lda, i0, kb, voa, tid, sm, sk);
      gemm_load<!TB>(g.B, g.ldb, j0, kb, vob, sb);
      int buf = 0;
      for (int k0 = kb; k0 < ke; k0 += GK) {
        double* Ab = As + buf * GSTAGE;
        double* Bb = Bs + buf * GSTAGE;
        if (MOD) {
#pragma unroll
          for (int u = 0; u < 4; ++u) sa[u] = sa[u] * sm[u] * sk[u];
        }
        gemm_store<TA>(Ab, tid, sa);
        gemm_store<!TB>(Bb, tid, sb);
        __syncthreads();
        if (k0 + GK < ke) {
          gemm_load<TA>(g.A, g.lda, i0, k0 + GK, voa, sa);
          if (MOD) gemm_load_mod<TA>(g.a_mul, g.k_scale, g.lda, i0, k0 + GK, voa, tid, sm, sk);
          gemm_load<!TB>(g.B, g.ldb, j0, k0 + GK, vob, sb);
        }
#pragma unroll
        for (int s = 0; s < GK / 4; ++s) {
          double af[4], bf[4];
#pragma unroll
          for (int a = 0; a < 4; ++a) af[a] = gemm_frag<TA>(Ab, 4 * s + q, wi + 16 * a + r);
#pragma unroll
          for (int b = 0; b < 4; ++b) bf[b] = gemm_frag<!TB>(Bb, 4 * s + q, wj + 16 * b + r);
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = TGP_MFMA(af[a], bf[b], acc[a][b]);
        }
        buf ^= 1;
      }
    }
    // epilogue: C/D layout row = q + 4 rr, col = r.
    // Plain (EPI = 0): C = alpha * acc, nothing but stores -- on gfx9 stores and loads share vmcnt, so a load
    // between two stores makes the second wait for the first to complete (one memory round trip per element).
    // EPI = 1: every optional read (row vectors once, column vectors and the beta*C / gamma*add matrix per
    // 16-column group) is issued ahead of the stores it feeds.  The launcher never sets beta and add together.
    if (!EPI) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            gemm_st1(C + (size_t)(i0 + wi + 16 * a + q + 4 * rr) * g.ldc + j0 + wj + 16 * b + r, g.alpha * acc[a][b][rr]);
    } else {
      const bool ha = g.add != nullptr;
      const double* __restrict__ E = ha ? g.add : C;
      const int lde = ha ? g.ldadd : g.ldc;
      const bool he = ha || g.beta != 0.0;
      double rs[4][4], rv[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = i0 + wi + 16 * a + q + 4 * rr;
          rs[a][rr] = g.row_scale ? gemm_ld1(g.row_scale + row) : 1.0;
          rv[a][rr] = g.rowv ? gemm_ld1(g.rowv + row) : 0.0;
        }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int col = j0 + wj + 16 * b + r;
        const double cs = g.col_scale ? gemm_ld1(g.col_scale + col) : 1.0;
        const double cv = g.colv ? gemm_ld1(g.colv + col) : 0.0;
        double ein[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            ein[a][rr] = he ? gemm_ld1(E + (size_t)(i0 + wi + 16 * a + q + 4 * rr) * lde + col) : 0.0;
        const double ca = ha ? g.gamma : 0.0, cb = ha ? 0.0 : g.beta;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            double x = g.alpha * acc[a][b][rr] + ca * ein[a][rr];
            x = x * cs * rs[a][rr] + rv[a][rr] * cv + cb * ein[a][rr];
            gemm_st1(C + (size_t)(i0 + wi + 16 * a + q + 4 * rr) * g.ldc + col, x);
          }
      }
    }
  }
}

template <bool TA, bool TB, bool MOD, bool EPI>
__global__ __launch_bounds__(256, 2) void k_gemm(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  double* As = reinterpret_cast<double*>(gemm_smem);  // 2 stage buffers
  double* Bs = As + 2 * GSTAGE;                       // 2 stage buffers
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (g.xcd == 1) {
    const int gx = gridDim.x, gy8 = (int)gridDim.y & ~7, L = bx + gx * by;
    if (L < gx * gy8) {  // whole groups of 8 tile rows; the ragged tail keeps the natural order
      const int s = L >> 3;
      by = (L & 7) + 8 * (s / gx);
      bx = s % gx;
    }
  } else if (g.xcd == 2 && (gridDim.z & 7) == 0) {
    const int gx = gridDim.x, per = gx * (int)gridDim.y, L = bx + gx * (by + (int)gridDim.y * bz), s = L >> 3;
    bz = (L & 7) + 8 * (s / per);
    const int rest = s % per;
    by = rest / gx;
    bx = rest % gx;
  } else if (g.xcd == 4 && ((int)gridDim.y & 7) == 0) {
    // unpaired triangular op(B): every XCD (id = linear id mod 8) owns the tile rows by = xcd, xcd + 8, ... and walks them
    // column tile by column tile, HEAVIEST column first -- the column tiles of a row still start together (they share the
    // op(A) rows in that XCD's L2) and what is left for the end of the launch are the short tiles
    const int gx = gridDim.x, L = bx + gx * by, xc = L & 7, sq = L >> 3, nrx = (int)gridDim.y >> 3;
    const int w = sq / nrx;
    by = xc + 8 * (sq % nrx);
    bx = (g.tri & TRI_B_LOWER) ? w : gx - 1 - w;
    if (by * GT >= g.m) return;   // the launcher rounds the tile rows up to a multiple of 8
  } else if (g.xcd == 3) {
    // compact split-K enumeration of the lower block triangle: gridDim.x = ntl * ksplit, slab-major, and the ids
    // that share an XCD (same id mod 8) take a contiguous run of (slab, tile) pairs -- equal load per XCD and the
    // tiles of one slab (same operand rows) mostly on one L2
    const int W = gridDim.x, L = blockIdx.x, qq = W >> 3, rr8 = W & 7, xc = L & 7;
    const int o = (xc < rr8 ? xc * (qq + 1) : rr8 * (qq + 1) + (xc - rr8) * qq) + (L >> 3);
    const int nb = g.m / GT, ntl = nb * (nb + 1) / 2;
    bz = o / ntl;
    const int t = o % ntl;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    by = ti;
    bx = t - ti * (ti + 1) / 2;
  } else if (g.tri & TRI_A_LOWER) {
    by = (int)gridDim.y - 1 - by;  // heaviest tile rows first when the k range grows with the tile row
  }
  // Triangular op(B): the k range grows (or shrinks) linearly with the column tile, so a workgroup takes the column
  // tiles j and nj-1-j one after the other -- every workgroup then runs the same number of k stages (the launcher
  // halves gridDim.x).  Both tiles re-read the same op(A) row block.
  const int nj = g.n / GT;
  const uint64_t kernarg = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  gemm_tile<TA, TB, MOD, EPI>(kernarg, by * GT, bx * GT, bz, As, Bs);
  if (g.pair && nj - 1 - bx > bx) {
    __syncthreads();  // the first tile's last stage may still be read by slower waves
    gemm_tile<TA, TB, MOD, EPI>(kernarg, by * GT, (nj - 1 - bx) * GT, bz, As, Bs);
  }
}

// Two independent products in ONE launch (the first `na` workgroups run product a, the rest product b; natural tile
// order, no split-K): the blocked factorisation's chain of small launches leaves most CUs idle, and the block-row inverse
// that does not depend on the current step rides in the same launches instead of costing launches -- or cross-queue
// graph dependencies, ~10 us each under replay -- of its own.  The second GemmArgs sits behind the first in the kernel
// argument segment (gemm_tile reads its arguments through scalar loads from that segment).
template <bool TA1, bool TB1, bool MOD1, bool EPI1, bool TA2, bool TB2, bool MOD2, bool EPI2>
__global__ __launch_bounds__(256, 2) void k_gemm_pair(GemmArgs a, GemmArgs b, int na, int gxa, int gya, int gxb, int gyb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  double* As = reinterpret_cast<double*>(gemm_smem);
  double* Bs = As + 2 * GSTAGE;
  const uint64_t kernarg = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  int L = blockIdx.x;
  if (L < na) {
    int bx = L % gxa, by = L / gxa;
    if (a.tri & TRI_A_LOWER) by = gya - 1 - by;
    const int nj = a.n / GT;
    gemm_tile<TA1, TB1, MOD1, EPI1>(kernarg, by * GT, bx * GT, 0, As, Bs);
    if (a.pair && nj - 1 - bx > bx) {
      __syncthreads();
      gemm_tile<TA1, TB1, MOD1, EPI1>(kernarg, by * GT, (nj - 1 - bx) * GT, 0, As, Bs);
    }
  } else {
    L -= na;
    int bx = L % gxb, by = L / gxb;
    if (b.tri & TRI_A_LOWER) by = gyb - 1 - by;
    const int nj = b.n / GT;
    const uint64_t kb = kernarg + sizeof(GemmArgs);
    gemm_tile<TA2, TB2, MOD2, EPI2>(kb, by * GT, bx * GT, 0, As, Bs);
    if (b.pair && nj - 1 - bx > bx) {
      __syncthreads();
      gemm_tile<TA2, TB2, MOD2, EPI2>(kb, by * GT, (nj - 1 - bx) * GT, 0, As, Bs);
    }
  }
}

int launch_gemm(bool ta, bool tb, const GemmArgs& g, hipStream_t st);  // tgp_big.hip (checks, tiling choice)
// tgp_gemm128.hip: the 128 x 128 kernels' launchers, on arguments launch_gemm has normalised
int launch_gemm128(bool ta, bool tb, bool mod, bool epi, const GemmArgs& g, hipStream_t st);
int launch_gemm128_pair_ft_ff(bool epi, const GemmArgs& a, const GemmArgs& b, int na, int gxa, int gya, int gxb, int gyb,
                              hipStream_t st);

}  // namespace tgp
