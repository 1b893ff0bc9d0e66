// tgp_big.hip -- the general-M ELBO step (128 < M <= TGP_BIG_MAX_M), same algebra as the fused path.
//
// At M = 1000 (SURVEY config C5) neither the M x M operators nor a row block's A = L^-1 K_MN column fit in LDS/registers,
// and the work is dense contraction proper: 5.25 M^2 N flops per step against 8 (D+1) N compulsory bytes.  So this
// path is organised around ONE tiled MFMA GEMM (tgp_gemm.hpp) with fused operand/epilogue modifiers:
//
//   prepare   : K_MM, blocked right-looking Cholesky (128-wide: in-LDS potrf+trtri of the diagonal block, panel and
//               trailing update as GEMMs), block-row inverse J = L^-1, S = Lq Lq^T, H' = J^T (S - I), w = J^T m, KL
//   row chunks: rows are processed NC (<= 16384) at a time; the chunk matrices are the TRANSPOSES of the DESIGN.md
//               operands, laid out [NC][MP] (one data row = 8 KB contiguous), so that the triangular GEMMs stream
//               contiguous 1 MB row blocks and the two reductions over rows (G, T) read k-major, fully coalesced:
//               K' = K_NM -> A' = K' J^T -> B' = A' Lq -> (mu, v) -> likelihood (k_ell_gauss / k_ell_flow) ->
//               Abar' = vbar o (2 B' Lq^T - 2 A') + mubar m^T (GEMM epilogue) -> Kbar' = Abar' J ->
//               T += (Kbar' o K')^T [xs, xs^2, 1] (split-K)   G += A'^T diag(vbar) A' (split-K SYRK)   s += A'^T mubar
//   backward  : Lbar = -tril(w s^T + 2 H' G), Lambar = 2 tril(G Lq) - kl(...), Q = Phi(L^T Lbar) + Phi(.)^T,
//               Kbar_MM = 1/2 J^T Q J, U = (Kbar_MM o K_MM) [Zs, Zs^2, 1], parameter gradients.
// Replaces the same reference lines as tgp_mm.hip / tgp_rows.hpp (models/sparse_MF_SP.py:274-431,552-626).
#include "tgp_dev.hpp"
#include "tgp_gemm.hpp"
#include "tgp_launch.hpp"
#include "tgp_prep.hpp"   // hand-off primitives (sync_wait / sync_add / st_agent / ld_agent)

namespace tgp {

#define LAUNCH_CHECK()                                              \
  do {                                                              \
    hipError_t e_ = hipGetLastError();                              \
    if (e_ != hipSuccess) return set_error(e_, __FILE__, __LINE__); \
  } while (0)

#define BIG_XW 128    /* width of the augmented coordinate matrices [xs, xs^2, 1, 0...] */
#define BIG_KST 32    /* split-K slabs of the T statistics GEMM */
#define BIG_NKL 64    /* KL partial blocks */
#define BIG_SSL 32    /* row slabs of the s = A'^T mubar partial sums */
#define BIG_NCMAX 16384

struct BigPlan {
  int N, D, M, S, nblk, P, RP, lik, kernel;
  int MP, DP, NC, nchunks, NP, LS;
  int ksg;  // split-K slabs of the G SYRK: about two workgroups per CU over the lower block triangle
  size_t hdr, ils, ls, Zs, mpad, w, sv, klpart, svb, spart;
  size_t Kmm, Lm, J, Lq, S_, Hp, G, Q, R1, tmp;
  size_t Zaug, U, T, Xaug;
  size_t Kc, A, B, Ab;
  size_t mu, v, mub, vb;
  size_t Gpart, Tpart, likslot, likws;
  size_t cstride;  // second set of chunk buffers {Xaug, K', A', B', Abar'} (0 = none): forward of chunk c+1 overlaps backward of chunk c
  size_t Sk;  // split-K slabs of the M x M products
  size_t Kmmg, TpartK, TK, UK;  // MATERN32 only: derivative-weight K_MM, statistics of (Kbar o K) next to those of (Kbar o K_g)
  size_t total;
};

// tgp_model.plan (include/tgp_hip.h): chunk size and chunk-pipeline overlap are properties of the CALL (they
// were process-global environment switches read once)
static int big_chunk_max(int plan) {
  int x = TGP_PLAN_CHUNK_OF(plan);
  if (x <= 0) x = BIG_NCMAX;
  if (x < 128) x = 128;
  return (x + 127) / 128 * 128;
}

// helper stream + events for the chunk pipeline (created once per host thread)
struct BigAux {
  hipStream_t fwd;
  hipEvent_t e0, eF[2], eB[2];
  bool ok;
};
static BigAux* big_aux() {
  // one set per HOST THREAD: calls from different threads never share helper streams or events (the ABI is re-entrant
  // across threads); two engines driven by one thread share a set, which is correct (stream order) and, once their steps
  // are captured, irrelevant (the forked branches are nodes of each engine's own graph)
  static thread_local BigAux a;
  static thread_local bool init = false;
  if (!init) {
    init = true;
    a.ok = hipStreamCreateWithFlags(&a.fwd, hipStreamNonBlocking) == hipSuccess;
    hipEvent_t* ev[5] = {&a.e0, &a.eF[0], &a.eF[1], &a.eB[0], &a.eB[1]};
    for (int i = 0; i < 5 && a.ok; ++i) a.ok = hipEventCreateWithFlags(ev[i], hipEventDisableTiming) == hipSuccess;
    if (!a.ok) (void)hipGetLastError();
  }
  return a.ok ? &a : nullptr;
}

static BigPlan plan_parity(const BigPlan& p, int par) {
  BigPlan q = p;
  const size_t d = (size_t)par * p.cstride;
  q.Xaug += d; q.Kc += d; q.A += d; q.B += d; q.Ab += d;
  return q;
}

static int make_big_plan(BigPlan& p, int N, int D, int M, int S, int nblk, int P, int RP, int lik, int kernel, int plan = 0) {
  if (D < 1 || D > 16) return -2;
  if (M < 1 || M > TGP_BIG_MAX_M) return TGP_E_UNSUPPORTED;
  // (M <= 128 with the RBF kernel normally takes the fused path; tgp_api.hip sends it here when the flow stack of a training
  //  step does not fit a CU's LDS beside the fused row kernel's tiles)
  p.kernel = kernel;
  p.N = N; p.D = D; p.M = M; p.S = S; p.nblk = nblk; p.P = P; p.RP = RP; p.lik = lik;
  p.MP = (M + 127) / 128 * 128;
  p.DP = D <= 4 ? 4 : (D <= 8 ? 8 : 16);
  const int ncmax = big_chunk_max(plan);
  p.nchunks = (N + ncmax - 1) / ncmax;
  if (p.nchunks < 1) p.nchunks = 1;
  p.NC = (int)rup((size_t)(N + p.nchunks - 1) / p.nchunks, 128);
  p.NP = p.NC * p.nchunks;
  p.LS = (int)rup(2 + P, 8);
  {
    const int nb = p.MP / 128, ntiles = nb * (nb + 1) / 2;
    p.ksg = 512 / ntiles;  // one resident round: two workgroups per CU
    // ... but no thinner than 256 rows per slab: a rank's 1 250 rows of an 8-GPU minibatch cut 14 ways were 90-row slabs
    // whose partials (14 x M^2) cost k_big_reduce more than the SYRK itself (29 -> 19 us)
    if (p.ksg > p.NC / 256) p.ksg = p.NC / 256;
    if (p.ksg < 1) p.ksg = 1;
    if (p.ksg > 32) p.ksg = 32;
  }
  size_t o = 0;
  const size_t mm = (size_t)p.MP * p.MP, mn = (size_t)p.MP * p.NC;
  p.hdr = o; o += H_N;
  p.ils = o; o += 16;
  p.ls = o; o += 16;
  p.Zs = o; o += (size_t)p.MP * p.DP;
  p.mpad = o; o += p.MP;
  p.w = o; o += p.MP;
  p.sv = o; o += p.MP;
  p.klpart = o; o += BIG_NKL;
  p.svb = o; o += 16;
  p.spart = o; o += (size_t)BIG_SSL * p.MP;
  p.Kmm = o; o += mm; p.Lm = o; o += mm; p.J = o; o += mm; p.Lq = o; o += mm; p.S_ = o; o += mm;
  p.Hp = o; o += mm; p.G = o; o += mm; p.Q = o; o += mm; p.R1 = o; o += mm;
  p.tmp = o; o += (size_t)128 * p.MP;
  p.Zaug = o; o += (size_t)p.MP * BIG_XW;
  p.U = o; o += (size_t)p.MP * BIG_XW;
  p.T = o; o += (size_t)p.MP * BIG_XW;
  p.Xaug = o; o += (size_t)p.NC * BIG_XW;
  p.Kc = o; o += mn; p.A = o; o += mn; p.B = o; o += mn; p.Ab = o; o += mn;
  p.cstride = 0;
  if (p.nchunks >= 2 && !(plan & TGP_PLAN_NO_CHUNK_OVERLAP)) {
    p.cstride = o - p.Xaug;  // Xaug, Kc, A, B, Ab are contiguous
    o += p.cstride;
  }
  p.mu = o; o += p.NP; p.v = o; o += p.NP; p.mub = o; o += p.NP; p.vb = o; o += p.NP;
  p.Gpart = o; o += (size_t)p.ksg * mm;
  p.Tpart = o; o += (size_t)BIG_KST * p.MP * BIG_XW;
  p.likslot = o; o += (size_t)p.nchunks * p.LS;
  p.likws = o; o += rup(lik_workspace_doubles(p.NC, P, RP), 16);
  {
    const int nb = p.MP / 128;
    int ks = 256 / (nb * nb);
    ks = ks < 1 ? 1 : (ks > 8 ? 8 : ks);
    p.Sk = o; o += (size_t)(ks > 1 ? 8 : 0) * (nb <= 4 ? mm : mm / 2);
  }
  p.Kmmg = p.TpartK = p.TK = p.UK = 0;
  if (kernel != TGP_KERNEL_SCALE_RBF) {
    p.Kmmg = o; o += mm;
    p.TpartK = o; o += (size_t)BIG_KST * p.MP * BIG_XW;
    p.TK = o; o += (size_t)p.MP * BIG_XW;
    p.UK = o; o += (size_t)p.MP * BIG_XW;
  }
  p.total = o;
  return 0;
}

size_t big_workspace_doubles(int N, int D, int M, int S, int nblk, int P, int RP, int kernel, int plan) {
  BigPlan p;
  if (make_big_plan(p, N, D, M, S, nblk, P, RP, TGP_LIK_FLOW, kernel, plan) != 0) return 0;
  return p.total;
}

// ---------------------------------------------------------------------------------------------------
// GEMM launcher
// ---------------------------------------------------------------------------------------------------
// (the 128 x 128 kernels and their launchers are a translation unit of their own, tgp_gemm128.hip: they are built with
// another instruction scheduler -- see the Makefile)

// argument checks and the launcher-set fields (pair, xcd); 0 or -1
static int gemm_normalise(GemmArgs& g) {
  if (g.m % GT || g.n % GT || g.k % GK || g.m < 1 || g.n < 1 || g.ksplit < 1) return -1;
  if ((g.lda | g.ldb) & 1) return -1;  // 16-byte operand loads
  if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B) | reinterpret_cast<uintptr_t>(g.a_mul) |
       reinterpret_cast<uintptr_t>(g.k_scale)) & 15)
    return -1;
  if (g.add != nullptr && g.beta != 0.0) return -1;  // the epilogue reads one extra matrix, not two
  if (g.xcd == 4) g.xcd = 1;   // 4 is the launcher's own choice (below)
  // triangular op(B) without split-K: pair column tiles so that all workgroups run the same number of stages
  g.pair = ((g.tri & (TRI_B_LOWER | TRI_B_UPPER)) && !(g.tri & (TRI_A_LOWER | TRI_A_UPPER | TRI_C_LOWER)) && g.ksplit == 1 &&
            g.n / GT > 1) ? 1 : 0;
  if (g.pair) {
    // Pairing makes every workgroup equal (nj + 1 k-blocks) but halves their number: with 1-2 workgroups per CU that can
    // leave a quarter of the chip with twice the work (a 10 000-row minibatch at M = 1000: 316 workgroups of 9 blocks on
    // 256 CUs = 18 block times against 2844 / 256 = 11.1).  Estimate both in k-block times; where the unpaired launch is
    // better it runs in the per-XCD heaviest-column-first order (xcd 4): the long workgroups start at once, the short
    // ones fill the end of the launch (measured on that minibatch: 317 us paired, 300 unpaired in row order, 250 in
    // this order; one workgroup per CU instead of two was slower).
    const long nrow = g.m / GT, nj = g.n / GT;
    const long paired = ((nrow * ((nj + 1) / 2) + 255) / 256) * (nj + 1);
    long unpaired = (nrow * nj * (nj + 1) / 2 + 255) / 256;
    if (unpaired < nj) unpaired = nj;
    if (unpaired * 108 < paired * 100) {
      g.pair = 0;
      g.xcd = 4;
    }
  }
  if ((g.tri & TRI_C_LOWER) && g.ksplit > 1 && g.m == g.n && !(g.tri & ~TRI_C_LOWER)) g.xcd = 3;
  else if (g.xcd == 3) g.xcd = 0;
  return 0;
}
static bool gemm_has_mod(const GemmArgs& g) { return g.a_mul != nullptr || g.k_scale != nullptr; }
static bool gemm_has_epi(const GemmArgs& g) {
  return g.add != nullptr || g.beta != 0.0 || g.col_scale || g.row_scale || g.rowv || g.colv;
}

static int launch_gemm64(bool tb, const GemmArgs& g, hipStream_t st);

int launch_gemm(bool ta, bool tb, const GemmArgs& g_in, hipStream_t st) {
  GemmArgs g = g_in;
  if (gemm_normalise(g)) return -1;
  {
    // products without operand modifiers, with a (possibly triangular) op(B), whose 128 x 128 tiling leaves the chip mostly
    // idle: 64 x 64 tiles.
    // Times in units of one 128^3 k-block on a CU: the big tiling's longest workgroup / balance, against the small
    // tiling's (a 64^3 block is 1/8 of it, two resident workgroups share a CU).
    const bool plain = !ta && !gemm_has_mod(g) && g.ksplit == 1 && !(g.tri & ~(TRI_B_LOWER | TRI_B_UPPER)) && g.k % 64 == 0;
    if (plain) {
      const long nrow = g.m / GT, nj = g.n / GT, kb = g.k / GT;
      const bool tri = (g.tri & (TRI_B_LOWER | TRI_B_UPPER)) != 0;
      const long per_row = tri ? nj * (nj + 1) / 2 : nj * kb;
      const long longest = tri ? nj : kb;
      long big = g.pair ? ((nrow * ((nj + 1) / 2) + 255) / 256) * (nj + 1) : (nrow * per_row + 255) / 256;
      if (!g.pair && big < longest) big = longest;
      // small tiling, in eighths of a block time: total / 256 CUs, or the longest tile at half rate
      const long nj2 = 2 * nj, rows2 = (g.m + 63) / 64;
      const long per_row2 = tri ? nj2 * (nj2 + 1) / 2 : nj2 * 2 * kb;
      long small8 = (rows2 * per_row2 + 255) / 256;
      const long longest2 = 2 * (tri ? nj2 : 2 * kb);
      if (small8 < longest2) small8 = longest2;
      if (small8 * 10 < big * 8 * 7) return launch_gemm64(tb, g, st);   // predicted at least 30 % faster
    }
  }
  const bool mod = g.a_mul != nullptr || g.k_scale != nullptr;
  const bool epi = g.add != nullptr || g.beta != 0.0 || g.col_scale || g.row_scale || g.rowv || g.colv;
  return launch_gemm128(ta, tb, mod, epi, g, st);
}

// ---------------------------------------------------------------------------------------------------
// The k = 128 products of the blocked factorisation (panel, trailing update, inverse row, inverse push) on SMALL tiles.
// k_gemm's 128 x 128 tile is 14 us of fp64 MFMA on one CU all by itself, and these products have 1-30 such tiles: the
// launches were bound by tile size while 200+ CUs idled.  Here a workgroup = 4 waves x (32 x 32) computes a 32x128,
// 128x32 or 64x64 tile (3.6 us of MFMA), so a product spreads over up to 4x the CUs.  The two in-place products keep
// their safety by tile SHAPE: the panel (C = A J_kk^T over A) takes whole rows (32 x 128), the inverse row (C = -J_kk B
// over B) whole columns (128 x 32) -- a workgroup reads exactly the operand region it overwrites, all of it before its
// first store.  K is staged in two halves of 64 through LDS ([x][68]: fragment reads on 32 distinct 8-byte banks per
// half wave), the second half's global loads in flight under the first half's MFMAs.  Two jobs per launch, as before.
struct FacJob {
  const double* A;   // [m][128] row-major, leading dimension lda
  const double* B;   // tb: [n][128] row-major (op(B) = B^T) ; else [128][n] row-major
  double* C;
  int lda, ldb, ldc, m, n, tb, shape, lower;   // shape 0: 32 x 128 tiles, 1: 128 x 32, 2: 64 x 64 ; lower: skip tiles above the diagonal
  double alpha, beta;
};
#define FAC_LD 68
enum { FP_WORD = 4 };   // status[4]: producers of the diagonal block a workgroup of the same launch waits for (k_big_kmm_potrf, k_fac_potrf)
#define FAC_LDS_BYTES ((size_t)160 * FAC_LD * sizeof(double))

template <int WM, int WN, bool TB>
__device__ __forceinline__ void fac_tile(const FacJob& g, int i0, int j0, double* As, double* Bs) {
  constexpr int TM = 32 * WM, TN = 32 * WN;
  constexpr int NA = TM * 32 / 256;                 // d2 loads per thread for a [TM][64] half of A (8 .. 16 .. 4)
  constexpr int NB = TN * 32 / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int wi = (wave / WN) * 32, wj = (wave % WN) * 32;
  d2 ra[NA], rb[NB];
  auto load_half = [&](int kh) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int e = tid + 256 * u, x = e >> 5, k2 = e & 31;          // 32 d2 per row of 64 k
      ra[u] = *reinterpret_cast<const d2*>(g.A + (size_t)(i0 + x) * g.lda + 64 * kh + 2 * k2);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int e = tid + 256 * u;
      if (TB) {
        const int x = e >> 5, k2 = e & 31;
        rb[u] = *reinterpret_cast<const d2*>(g.B + (size_t)(j0 + x) * g.ldb + 64 * kh + 2 * k2);
      } else {
        const int kk = e / (TN / 2), x2 = e % (TN / 2);                // TN / 2 d2 per k row
        rb[u] = *reinterpret_cast<const d2*>(g.B + (size_t)(64 * kh + kk) * g.ldb + j0 + 2 * x2);
      }
    }
  };
  auto store_half = [&]() {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int e = tid + 256 * u, x = e >> 5, k2 = e & 31;
      *reinterpret_cast<d2*>(As + x * FAC_LD + 2 * k2) = ra[u];
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int e = tid + 256 * u;
      if (TB) {
        const int x = e >> 5, k2 = e & 31;
        *reinterpret_cast<d2*>(Bs + x * FAC_LD + 2 * k2) = rb[u];
      } else {
        const int kk = e / (TN / 2), x2 = e % (TN / 2);
        Bs[(2 * x2) * FAC_LD + kk] = rb[u][0];
        Bs[(2 * x2 + 1) * FAC_LD + kk] = rb[u][1];
      }
    }
  };
  // the epilogue's read of C (beta != 0) is requested with the first operands
  double cold[2][2][4];
  const bool hb = g.beta != 0.0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        cold[a][b][rr] = hb ? g.C[(size_t)(i0 + wi + 16 * a + q + 4 * rr) * g.ldc + j0 + wj + 16 * b + r] : 0.0;
  d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = {0, 0, 0, 0};
  load_half(0);
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    store_half();
    __syncthreads();
    if (kh == 0) load_half(1);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      double af[2], bf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = As[(wi + 16 * a + r) * FAC_LD + 4 * s + q];
#pragma unroll
      for (int b = 0; b < 2; ++b) bf[b] = Bs[(wj + 16 * b + r) * FAC_LD + 4 * s + q];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = TGP_MFMA(af[a], bf[b], acc[a][b]);
    }
    __syncthreads();   // every wave is done with this half before the next one overwrites it
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        g.C[(size_t)(i0 + wi + 16 * a + q + 4 * rr) * g.ldc + j0 + wj + 16 * b + r] = g.alpha * acc[a][b][rr] + g.beta * cold[a][b][rr];
}

// Round 6: the panel and the inverse row as SINGLE-STAGE tiles (all 128 k in LDS at once: one memory round trip instead of two
// -- these launches sit on the factorisation's chain and are latency, not work).  16 x 128 resp. 128 x 16 output tiles keep the
// in-place safety of the 32-wide ones (whole rows resp. whole columns) and fit: (128 + 16) rows x FS_LD doubles = 152 KB.
#define FS_LD 132
#define FS_LDS_BYTES ((size_t)144 * FS_LD * sizeof(double))
// shape 3: C[i0 .. i0+16, 0 .. 128) = alpha A[i0 .. i0+16, :] B^T, B stored [n][k] (tb), in place over A
__device__ __forceinline__ void fac_tile_pan16(const FacJob& g, int i0, double* As, double* Bs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  d2 ra[4], rb[32];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u, x = e >> 6, k2 = e & 63;
    ra[u] = *reinterpret_cast<const d2*>(g.A + (size_t)(i0 + x) * g.lda + 2 * k2);
  }
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int e = tid + 256 * u, x = e >> 6, k2 = e & 63;
    rb[u] = *reinterpret_cast<const d2*>(g.B + (size_t)x * g.ldb + 2 * k2);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u, x = e >> 6, k2 = e & 63;
    *reinterpret_cast<d2*>(As + x * FS_LD + 2 * k2) = ra[u];
  }
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int e = tid + 256 * u, x = e >> 6, k2 = e & 63;
    *reinterpret_cast<d2*>(Bs + x * FS_LD + 2 * k2) = rb[u];
  }
  __syncthreads();
  d4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const double af = As[r * FS_LD + 4 * s + q];
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[b] = TGP_MFMA(af, Bs[(32 * wave + 16 * b + r) * FS_LD + 4 * s + q], acc[b]);
  }
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) g.C[(size_t)(i0 + q + 4 * rr) * g.ldc + 32 * wave + 16 * b + r] = g.alpha * acc[b][rr];
}
// shape 4: C[0 .. 128, j0 .. j0+16) = alpha A B[:, j0 .. j0+16), A [128][128], B stored [k][n], in place over B
__device__ __forceinline__ void fac_tile_row16(const FacJob& g, int j0, double* As, double* Bs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  d2 ra[32], rb[4];
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int e = tid + 256 * u, x = e >> 6, k2 = e & 63;
    ra[u] = *reinterpret_cast<const d2*>(g.A + (size_t)x * g.lda + 2 * k2);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u, kk = e >> 3, x2 = e & 7;            // 8 d2 per k row of 16 columns
    rb[u] = *reinterpret_cast<const d2*>(g.B + (size_t)kk * g.ldb + j0 + 2 * x2);
  }
#pragma unroll
  for (int u = 0; u < 32; ++u) {
    const int e = tid + 256 * u, x = e >> 6, k2 = e & 63;
    *reinterpret_cast<d2*>(As + x * FS_LD + 2 * k2) = ra[u];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + 256 * u, kk = e >> 3, x2 = e & 7;
    Bs[(2 * x2) * FS_LD + kk] = rb[u][0];
    Bs[(2 * x2 + 1) * FS_LD + kk] = rb[u][1];
  }
  __syncthreads();
  d4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const double bf = Bs[r * FS_LD + 4 * s + q];
#pragma unroll
    for (int a = 0; a < 2; ++a) acc[a] = TGP_MFMA(As[(32 * wave + 16 * a + r) * FS_LD + 4 * s + q], bf, acc[a]);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) g.C[(size_t)(32 * wave + 16 * a + q + 4 * rr) * g.ldc + j0 + r] = g.alpha * acc[a][rr];
}

__device__ __forceinline__ void fac_job(const FacJob& g, int L, double* As, double* Bs) {
  if (g.shape == 3) { fac_tile_pan16(g, 16 * L, As, As + 16 * FS_LD); return; }
  if (g.shape == 4) { fac_tile_row16(g, 16 * L, As, As + 128 * FS_LD); return; }
  const int TM = g.shape == 0 ? 32 : (g.shape == 1 ? 128 : 64), TN = g.shape == 0 ? 128 : (g.shape == 1 ? 32 : 64);
  const int gx = g.n / TN, bx = L % gx, by = L / gx;
  const int i0 = by * TM, j0 = bx * TN;
  if (g.lower && j0 > i0) return;
  if (g.shape == 0) { if (g.tb) fac_tile<1, 4, true>(g, i0, j0, As, Bs); else fac_tile<1, 4, false>(g, i0, j0, As, Bs); }
  else if (g.shape == 1) { if (g.tb) fac_tile<4, 1, true>(g, i0, j0, As, Bs); else fac_tile<4, 1, false>(g, i0, j0, As, Bs); }
  else { if (g.tb) fac_tile<2, 2, true>(g, i0, j0, As, Bs); else fac_tile<2, 2, false>(g, i0, j0, As, Bs); }
}

__global__ __launch_bounds__(256) void k_fac_pair(FacJob a, FacJob b, int na) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fac_smem[];
  double* As = reinterpret_cast<double*>(fac_smem);
  double* Bs = As + 128 * FAC_LD;                    // A tile: at most 128 rows; B tile: at most 128
  const int L = blockIdx.x;
  if (L < na) fac_job(a, L, As, Bs);
  else fac_job(b, L - na, As, Bs);
}

static int fac_tiles(const FacJob& g) {
  if (g.shape == 3) return g.m / 16;
  if (g.shape == 4) return g.n / 16;
  const int TM = g.shape == 0 ? 32 : (g.shape == 1 ? 128 : 64), TN = g.shape == 0 ? 128 : (g.shape == 1 ? 32 : 64);
  return (g.m / TM) * (g.n / TN);
}
static FacJob fac_job_args(const double* A, int lda, const double* B, int ldb, double* C, int ldc, int m, int n, int tb, int shape,
                           double alpha, double beta, int lower = 0) {
  FacJob g;
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.m = m; g.n = n; g.tb = tb; g.shape = shape; g.lower = lower;
  g.alpha = alpha; g.beta = beta;
  return g;
}
// one or two jobs in one launch (b.m == 0: only a)
static int launch_fac_pair(const FacJob& a, const FacJob& b, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fac_pair), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)FS_LDS_BYTES);
    if (e != hipSuccess) { (void)hipGetLastError(); return set_error(e, __FILE__, __LINE__); }
    attr_done = true;
  }
  const int na = fac_tiles(a), nb2 = b.m > 0 ? fac_tiles(b) : 0;
  static_assert(FS_LDS_BYTES >= 2 * 128 * FAC_LD * sizeof(double), "both tile families stage through one allocation");
  hipLaunchKernelGGL(k_fac_pair, dim3(na + nb2), dim3(256), FS_LDS_BYTES, st, a, b, na);
  LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// k_gemm64: C = alpha op(A) op(B) on 64 x 64 tiles (4 waves x 32 x 32), any k (multiple of 64), A stored [m][k], B
// stored [n][k] (TB) or [k][n]; triangular op(B) trims the k range per column tile.  For products whose 128 x 128 tiling
// cannot fill the chip (a rank's 1 250 rows of an 8-GPU minibatch: 10 tile rows x 8 columns = 80 workgroups, the longest
// walking 8 k-blocks of 16.5 us one after the other): four times the workgroups, a quarter of the time per k-block.
// Unpaired, per-XCD heaviest-column-first order (as k_gemm's xcd 4).  K advances 64 per stage through one LDS buffer per
// operand, the next stage's global loads in flight under the current stage's MFMAs.
// ---------------------------------------------------------------------------------------------------
struct G64Args {
  const double* A;
  const double* B;
  double* C;
  int lda, ldb, ldc, m, n, k, tb, tri;
  double alpha;
  // epilogue of k_gemm (EPI): x = alpha acc + gamma add ; C = x col_scale row_scale + rowv colv + beta C
  const double* add;
  int ldadd;
  double gamma, beta;
  const double* col_scale;
  const double* row_scale;
  const double* rowv;
  const double* colv;
};

// Round 6: 8 waves per workgroup.  The 4-wave form ran one wave per SIMD whenever a tile had its CU to itself -- the heavy tiles of a
// triangular product in the tail of the launch -- and a lone wave exposes every LDS store, barrier and fragment read of the
// single-buffered stage: the longest tile (16 k-blocks at M = 1024) took 60 us where its MFMAs are 30, and set the launch's time
// (72 us for 20 us of work per CU at 1 280 rows).  Now the k range of a tile is split over TWO groups of 4 waves (own stage buffers,
// two waves per SIMD: one group's MFMAs cover the other's staging), the second group's accumulators reach the first through LDS
// (fixed order: group 0 + group 1) and group 0 runs the epilogue.  One workgroup per CU.
#define G64_THREADS 512
#define G64_LDK 80                                      /* row length of the k-major B stage */
#define G64_GRP ((size_t)64 * FAC_LD + 64 * G64_LDK)    /* doubles of stage buffers per wave group */
#define G64_LDS_BYTES (2 * G64_GRP * sizeof(double))
template <bool TB, bool EPI>
__device__ __forceinline__ void g64_tile(const G64Args& g, int i0, int j0, double* smem) {
  const int tid8 = threadIdx.x, grp = tid8 >> 8, tid = tid8 & 255, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  double* As = smem + (size_t)grp * G64_GRP;
  double* Bs = As + 64 * FAC_LD;     // TB: [x][FAC_LD] like As ; else k-major [k][G64_LDK] (no transposing store: see store_stage)
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  int kb = 0, ke = g.k;
  if (g.tri & TRI_B_LOWER) kb = max(kb, j0);
  if (g.tri & TRI_B_UPPER) ke = min(ke, j0 + 64);
  // group 0: the first n0 = ceil(nblk / 2) k-blocks, group 1 the rest; both run n0 trips (the barriers are the workgroup's)
  const int nblk = (ke - kb) / 64, n0 = (nblk + 1) / 2, nmine = grp ? nblk - n0 : n0;
  const int kmine = kb + (grp ? n0 * 64 : 0);
  d2 ra[8], rb[8];
  auto load_stage = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + 256 * u, x = e >> 5, k2 = e & 31;
      ra[u] = *reinterpret_cast<const d2*>(g.A + (size_t)(i0 + x) * g.lda + k0 + 2 * k2);
      if (TB) rb[u] = *reinterpret_cast<const d2*>(g.B + (size_t)(j0 + x) * g.ldb + k0 + 2 * k2);
      else rb[u] = *reinterpret_cast<const d2*>(g.B + (size_t)(k0 + (e >> 5)) * g.ldb + j0 + 2 * (e & 31));
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + 256 * u, x = e >> 5, k2 = e & 31;
      *reinterpret_cast<d2*>(As + x * FAC_LD + 2 * k2) = ra[u];
      if (TB) *reinterpret_cast<d2*>(Bs + x * FAC_LD + 2 * k2) = rb[u];
      // (round 6: B stored [k][n] keeps that layout in LDS -- the transposing pair of scalar stores of rounds 3-5 put the 32
      //  lanes of a row on 4 banks; rows of 80 doubles: 16-byte stores and the fragment reads of a half wave are conflict-free)
      else *reinterpret_cast<d2*>(Bs + x * G64_LDK + 2 * k2) = rb[u];       // here x is the k row (e >> 5) and k2 the column pair
    }
  };
  d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = {0, 0, 0, 0};
  if (nmine > 0) load_stage(kmine);
  for (int it = 0; it < n0; ++it) {
    const bool act = it < nmine;
    if (act) store_stage();
    __syncthreads();
    if (it + 1 < nmine) load_stage(kmine + 64 * (it + 1));
    if (act) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        double af[2], bf[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) af[a] = As[(wi + 16 * a + r) * FAC_LD + 4 * s + q];
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[b] = TB ? Bs[(wj + 16 * b + r) * FAC_LD + 4 * s + q] : Bs[(4 * s + q) * G64_LDK + wj + 16 * b + r];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = TGP_MFMA(af[a], bf[b], acc[a][b]);
      }
    }
    __syncthreads();
  }
  // group 1's accumulators -> LDS (its own, now idle, stage buffers: 4 waves x 64 lanes x 16 doubles), group 0 adds them
  double* xch = smem + G64_GRP + (size_t)(wave * 64 + lane) * 17;
  if (grp) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) xch[(2 * a + b) * 4 + rr] = acc[a][b][rr];
  }
  __syncthreads();
  if (grp) return;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[a][b][rr] += xch[(2 * a + b) * 4 + rr];
  if (!EPI) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          g.C[(size_t)(i0 + wi + 16 * a + q + 4 * rr) * g.ldc + j0 + wj + 16 * b + r] = g.alpha * acc[a][b][rr];
  } else {
    // the same arithmetic, in the same order, as k_gemm's EPI epilogue (every read issued ahead of the stores)
    const bool ha = g.add != nullptr;
    const double* __restrict__ E = ha ? g.add : g.C;
    const int lde = ha ? g.ldadd : g.ldc;
    const bool he = ha || g.beta != 0.0;
    const double ca = ha ? g.gamma : 0.0, cb = ha ? 0.0 : g.beta;
    double rs[2][4], rv[2][4], ein[2][2][4], cs[2], cv[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = i0 + wi + 16 * a + q + 4 * rr;
        rs[a][rr] = g.row_scale ? g.row_scale[row] : 1.0;
        rv[a][rr] = g.rowv ? g.rowv[row] : 0.0;
      }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = j0 + wj + 16 * b + r;
      cs[b] = g.col_scale ? g.col_scale[col] : 1.0;
      cv[b] = g.colv ? g.colv[col] : 0.0;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) ein[a][b][rr] = he ? E[(size_t)(i0 + wi + 16 * a + q + 4 * rr) * lde + col] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          double x = g.alpha * acc[a][b][rr] + ca * ein[a][b][rr];
          x = x * cs[b] * rs[a][rr] + rv[a][rr] * cv[b] + cb * ein[a][b][rr];
          g.C[(size_t)(i0 + wi + 16 * a + q + 4 * rr) * g.ldc + j0 + wj + 16 * b + r] = x;
        }
  }
}

template <bool EPI>
__global__ __launch_bounds__(G64_THREADS) void k_gemm64(G64Args g, int gx, int gy8) {
  extern __shared__ __attribute__((aligned(16))) unsigned char g64_smem[];
  double* smem = reinterpret_cast<double*>(g64_smem);
  // per-XCD heaviest-column-first order: XCD xc owns the tile rows xc, xc + 8, ...
  const int L = blockIdx.x, xc = L & 7, sq = L >> 3, nrx = gy8 >> 3;
  const int w = sq / nrx, by = xc + 8 * (sq % nrx);
  int bx = w;
  if (g.tri & TRI_B_UPPER) bx = gx - 1 - w;
  if (by * 64 >= g.m) return;
  if (g.tb) g64_tile<true, EPI>(g, by * 64, bx * 64, smem);
  else g64_tile<false, EPI>(g, by * 64, bx * 64, smem);
}

static int launch_gemm64(bool tb, const GemmArgs& g, hipStream_t st) {
  static bool attr_done[2] = {false, false};
  const size_t lds = G64_LDS_BYTES;
  const bool epi = gemm_has_epi(g);
  const void* f = epi ? reinterpret_cast<const void*>(k_gemm64<true>) : reinterpret_cast<const void*>(k_gemm64<false>);
  if (!attr_done[epi]) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return set_error(e, __FILE__, __LINE__); }
    attr_done[epi] = true;
  }
  G64Args a;
  a.A = g.A; a.B = g.B; a.C = g.C; a.lda = g.lda; a.ldb = g.ldb; a.ldc = g.ldc; a.m = g.m; a.n = g.n; a.k = g.k; a.tb = tb ? 1 : 0;
  a.tri = g.tri; a.alpha = g.alpha;
  a.add = g.add; a.ldadd = g.ldadd; a.gamma = g.gamma; a.beta = g.beta; a.col_scale = g.col_scale; a.row_scale = g.row_scale;
  a.rowv = g.rowv; a.colv = g.colv;
  const int gx = g.n / 64, gy8 = (g.m / 64 + 7) & ~7;
  if (epi) hipLaunchKernelGGL(k_gemm64<true>, dim3((unsigned)(gx * gy8)), dim3(G64_THREADS), lds, st, a, gx, gy8);
  else hipLaunchKernelGGL(k_gemm64<false>, dim3((unsigned)(gx * gy8)), dim3(G64_THREADS), lds, st, a, gx, gy8);
  LAUNCH_CHECK();
  return 0;
}

// a (op(B) transposed) and b (no transposition) in one launch, both plain or both with the C epilogue -- the shapes the
// blocked factorisation pairs up; anything else: two launches
static int launch_gemm_pair_ft_ff(const GemmArgs& a_in, const GemmArgs& b_in, hipStream_t st) {
  GemmArgs a = a_in, b = b_in;
  if (gemm_normalise(a) || gemm_normalise(b)) return -1;
  const bool epi = gemm_has_epi(a);
  const bool ok = !gemm_has_mod(a) && !gemm_has_mod(b) && gemm_has_epi(b) == epi && a.ksplit == 1 && b.ksplit == 1;
  if (!ok) {
    if (int rc = launch_gemm(false, true, a_in, st)) return rc;
    return launch_gemm(false, false, b_in, st);
  }
  a.xcd = 0; b.xcd = 0;
  const int gxa = a.pair ? (a.n / GT + 1) / 2 : a.n / GT, gya = a.m / GT;
  const int gxb = b.pair ? (b.n / GT + 1) / 2 : b.n / GT, gyb = b.m / GT;
  const int na = gxa * gya, nbk = gxb * gyb;
  return launch_gemm128_pair_ft_ff(epi, a, b, na, gxa, gya, gxb, gyb, st);
}

// ---------------------------------------------------------------------------------------------------
// prepare kernels
// ---------------------------------------------------------------------------------------------------
// (block 0 of k_big_hdr_zs: nothing below depends on the other blocks of that launch, nor they on it)
__device__ __forceinline__ void big_hdr_block(const BigPlan& p, const tgp_model& md, double* __restrict__ ws, int32_t* __restrict__ status) {
  const int tid = threadIdx.x;
  if (tid < 16) {
    const double l = tid < p.D ? softplus_d(md.raw_ls[tid]) : 1.0;
    ws[p.ls + tid] = l;
    ws[p.ils + tid] = tid < p.D ? 1.0 / l : 0.0;
  }
  if (tid == 0) {
    double* hdr = ws + p.hdr;
    hdr[H_S2] = softplus_d(md.raw_os[0]);
    hdr[H_ETA] = md.log_var_noise[0];
    hdr[H_EINV] = exp(-md.log_var_noise[0]);
    hdr[H_SIG_OS] = sigmoid_d(md.raw_os[0]);
    status[0] = 0;
    status[1] = 0;
    status[2] = 0;
  }
  // zero the adjoints of the padding rows of the last chunk
  for (int i = p.N + tid; i < p.NP; i += 256) { ws[p.mub + i] = 0.0; ws[p.vb + i] = 0.0; }
}

// Zs, padded m, Zaug = [Zs, Zs^2, 1, 0...]; one thread per Zaug element.  Block 0 also writes the header scalars and clears the
// status words (round 6: one launch instead of k_big_hdr -> k_big_zs; the two never depended on each other)
__global__ __launch_bounds__(256) void k_big_hdr_zs(BigPlan p, tgp_model md, double* __restrict__ ws, int32_t* __restrict__ status) {
  if (blockIdx.x == 0) big_hdr_block(p, md, ws, status);
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int row = (int)(e >> 7), c = (int)(e & 127);
  if (row >= p.MP) return;
  const int DP = p.DP;
  double x = 0.0;
  if (row < p.M) {
    const int d = c < DP ? c : c - DP;
    if (c < 2 * DP) {
      if (d < p.D) {
        const double z = md.Z[(size_t)row * p.D + d] * (1.0 / softplus_d(md.raw_ls[d]));
        x = c < DP ? z : z * z;
      }
    } else if (c == 2 * DP) {
      x = 1.0;
    }
  }
  ws[p.Zaug + e] = x;
  if (c < DP) ws[p.Zs + (size_t)row * DP + c] = x;
  if (c == 0) ws[p.mpad + row] = row < p.M ? md.m[row] : 0.0;
}

// whitened KL partial sums (models/sparse_MF_SP.py:406-431)
__global__ __launch_bounds__(256) void k_big_kl(BigPlan p, tgp_model md, double* __restrict__ ws) {
  __shared__ double red[4];
  const int M = p.M;
  double part = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)M * M; i += (size_t)BIG_NKL * 256) {
    const int rr = (int)(i / M), cc = (int)(i % M);
    if (cc <= rr) {
      const double x = md.Lam[i];
      part += x * x;
      if (cc == rr) part -= log(x * x);
    }
  }
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < M; i += 256) part += md.m[i] * md.m[i];
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) ws[p.klpart + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// K_MM (+ jitter, identity on the padding), the to-be-factorised copy (lower block triangle), J = 0, masked L_q: element e.
// The first diagonal block of the to-be-factorised copy and of J belongs to the workgroup that factorises it in the same launch
// (k_big_kmm_potrf): nothing of Lm is written there, of J only the tiles above the diagonal tiles (which that workgroup never writes).
// (DPC: the padded input dimension as a compile-time constant -- 2 DPC loads in flight at once instead of a loop of dependent
//  round trips; with one workgroup per CU there is no other wave to hide them behind)
// FIRST: only the element of the to-be-factorised copy, written through (st_agent) -- the first diagonal block on its way to the
// workgroup that factorises it in the same launch.
template <int DPC, bool FIRST = false>
__device__ __forceinline__ void big_kmm_element(const BigPlan& p, const tgp_model& md, double* __restrict__ ws,
                                                int32_t* __restrict__ status, int row, int col) {
  const int MP = p.MP, M = p.M, DP = DPC;
  const size_t e = (size_t)row * MP + col;
  double k, lq = 0.0;
  if (row < M && col < M) {
    const double* zr = ws + p.Zs + (size_t)row * DP;
    const double* zc = ws + p.Zs + (size_t)col * DP;
    double zrv[DPC], zcv[DPC];
#pragma unroll
    for (int d = 0; d < DPC; ++d) { zrv[d] = zr[d]; zcv[d] = zc[d]; }
    double d2 = 0.0;
#pragma unroll
    for (int d = 0; d < DPC; ++d) {
      const double t = zrv[d] - zcv[d];
      d2 += t * t;
    }
    k = cov_value(p.kernel, ws[p.hdr + H_S2], d2);
    if (row == col) k += md.jitter;
    if constexpr (!FIRST) {
      if (p.kernel != TGP_KERNEL_SCALE_RBF) ws[p.Kmmg + e] = cov_gweight(p.kernel, ws[p.hdr + H_S2], d2);
      if (k != k) status[1] = 1;
      if (col <= row) lq = md.Lam[(size_t)row * M + col];
    }
  } else {
    k = row == col ? 1.0 : 0.0;
    if constexpr (!FIRST)
      if (p.kernel != TGP_KERNEL_SCALE_RBF) ws[p.Kmmg + e] = 0.0;
  }
  if constexpr (FIRST) {
    st_agent(ws + p.Lm + e, k);
    return;
  }
  ws[p.Kmm + e] = k;
  const bool first = row < 128 && col < 128;
  if (!first) ws[p.Lm + e] = (row >> 7) >= (col >> 7) ? k : 0.0;
  if (!first || (col >> 4) > (row >> 4)) ws[p.J + e] = 0.0;
  ws[p.Lq + e] = lq;
}

// ---------------------------------------------------------------------------------------------------
// Cholesky + inverse of one 128 x 128 diagonal block, in LDS (8 waves; same lookahead schedule as k_prep_a).
//   in : lower triangle of Lm[o.., o..] (o = 128 kb), already updated by the previous block columns
//   out: L_kk (zero above the diagonal) in place, J_kk = L_kk^-1 into J's diagonal block
// ---------------------------------------------------------------------------------------------------
#ifdef TGP_STAMPS
// diagnostic build (tools/probes/stamp_big.py): s_memrealtime (100 MHz) of every wave at [0] kernel entry, [1] block in
// LDS, [2 + 3 j ..] window j: entry / own work done / after the window's barrier; [40] end.  Block kb = 1 of the last
// factorisation.
__device__ unsigned long long g_potrf_stamps[8 * 48];
#ifndef TGP_STAMP_KB
#define TGP_STAMP_KB 1
#endif
#define BSTAMP(i) do { if (kb == TGP_STAMP_KB && lane == 0) g_potrf_stamps[wave * 48 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif
#define POTRF_THREADS 512
#define POTRF_LD 129
#define POTRF_LDS_BYTES ((128 * POTRF_LD + 8 * 256) * sizeof(double))

// Where the block comes from.  PB_SHARED: written by workgroups of the SAME launch (k_fac_potrf, k_big_kmm_potrf) -- read past
// the L2 (ld_agent).
enum { PB_GLOBAL = 0, PB_SHARED = 1 };
template <int IN>
__device__ __forceinline__ void potrf_block(double* __restrict__ Lm, double* __restrict__ Jm, int ld, int kb,
                                            int32_t* __restrict__ status, unsigned char* smem_raw) {
  double* A = reinterpret_cast<double*>(smem_raw);  // 128 x 129: lower = block -> L ; strict-upper TILES hold J^T tiles
  double* Dt = A + 128 * POTRF_LD;                  // 8 x 256: inverses of the diagonal 16x16 tiles
  __shared__ int s_info, s_next;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: task indices and their branches stay scalar
  constexpr int LD = POTRF_LD, MT = 8, NW = POTRF_THREADS / 64;
  const size_t o = (size_t)kb * 128;
  double* Lb = Lm + o * ld + o;
  double* Jb = Jm + o * ld + o;
  if (tid == 0) { s_info = 0; s_next = 0; }
  BSTAMP(0);
  // Round 4: the schedule of k_prep_a's factorisation blocks as it stands after round 4 (tgp_mm.hip), on 8 x 8 tiles of
  // 16 with the block read from global memory: right-looking, ONE register pass of one wave per 16-column panel
  // (potrf_panel16: diagonal tile + every row below it; no inverse, triangular solve or panel product on the chain), one
  // 4-MFMA update per tile of the next block column between two panels, and everything else taken from a task counter in
  // LDS by the other waves in the pass's shadow, each task straight-line code with its LDS reads ahead of its MFMAs:
  //   window j:  Dinv_{j-1} (trtri16 by a task wave, no longer in the pass: 3.4 k instead of 4.1 k cycles per panel)
  //              block column j+1 -= block columns 0 .. j-1, ONE task per half column with a compile-time depth
  //              row j-2 of J = L^-1, a tile per task, stored to LDS (for the rows below) and to global memory at once
  //              tile row j-1 of L -> global memory, 16-byte stores
  // (round 3 took the catch-up tile by tile with run-time k loops, carried the inverse in the pass, wrote J out in tasks
  //  of its own and loaded the whole block before the first pass: 39.5 us per diagonal block.)
  // the block's lower tiles -> LDS: 32 rows of 4 per thread, every load in flight before the first LDS store (one memory
  // round trip for the whole block: it was written by other CUs' trailing update a launch ago and comes from beyond
  // this XCD's L2 -- fetched column by column in the windows, each fetch was a 2-3 us task and bound its window)
  {
    const int col = tid & 127, rsub = tid >> 7, ct = col >> 4;
    double v[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        v[k] = 0.0;
        if (ct <= (k >> 2)) v[k] = ld_maybe<IN == PB_SHARED>(Lb + (size_t)(4 * k + rsub) * ld + col);
      }
#pragma unroll
    for (int k = 0; k < 32; ++k)
      if (ct <= (k >> 2)) A[(4 * k + rsub) * LD + col] = v[k];
  }
  __syncthreads();
  BSTAMP(1);
  // tile (i, c) -= L[i rows, k0 .. k0+15] L[c rows, k0 .. k0+15]^T : one block column's contribution, 4 MFMAs
  auto sub16 = [&](int i, int c, int k0) {
    double a4[4], b4[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) { a4[s4] = A[(16 * i + r) * LD + k0 + 4 * s4 + q]; b4[s4] = A[(16 * c + r) * LD + k0 + 4 * s4 + q]; }
    double cur[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) cur[rr] = A[(16 * i + q + 4 * rr) * LD + 16 * c + r];
    d4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = TGP_MFMA(a4[s4], b4[s4], acc);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) A[(16 * i + q + 4 * rr) * LD + 16 * c + r] = cur[rr] - acc[rr];
  };
  // tiles (i, c), i = i0, i0 + istep, ... (two per trip), -= L[i rows, 0 .. 16 NC) L[c rows, 0 .. 16 NC)^T (all final)
  auto catch_up_nc = [&](int c, auto ncc, int i0, int istep) {
    constexpr int NC = decltype(ncc)::value;
    double bf[4 * NC];
#pragma unroll
    for (int s4 = 0; s4 < 4 * NC; ++s4) bf[s4] = A[(16 * c + r) * LD + 4 * s4 + q];
    for (int i = i0; i < MT; i += istep) {
      const bool two = i + 1 < MT;
      const int i1 = two ? i + 1 : i;
      if constexpr (NC > 3) {   // deep columns: a tile per trip (two tiles' operands + the column's = 72 doubles: spills)
        for (int ii = i; ii <= i1; ++ii) {
          double af[4 * NC], cur[4];
#pragma unroll
          for (int s4 = 0; s4 < 4 * NC; ++s4) af[s4] = A[(16 * ii + r) * LD + 4 * s4 + q];
#pragma unroll
          for (int u = 0; u < 4; ++u) cur[u] = A[(16 * ii + q + 4 * u) * LD + 16 * c + r];
          d4 acc = {0, 0, 0, 0};
#pragma unroll
          for (int s4 = 0; s4 < 4 * NC; ++s4) acc = TGP_MFMA(af[s4], bf[s4], acc);
#pragma unroll
          for (int u = 0; u < 4; ++u) A[(16 * ii + q + 4 * u) * LD + 16 * c + r] = cur[u] - acc[u];
        }
        continue;
      }
      double af0[4 * NC], af1[4 * NC], cur0[4], cur1[4];
#pragma unroll
      for (int s4 = 0; s4 < 4 * NC; ++s4) { af0[s4] = A[(16 * i + r) * LD + 4 * s4 + q]; af1[s4] = A[(16 * i1 + r) * LD + 4 * s4 + q]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { cur0[u] = A[(16 * i + q + 4 * u) * LD + 16 * c + r]; cur1[u] = A[(16 * i1 + q + 4 * u) * LD + 16 * c + r]; }
      d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
      for (int s4 = 0; s4 < 4 * NC; ++s4) acc0 = TGP_MFMA(af0[s4], bf[s4], acc0);
#pragma unroll
      for (int s4 = 0; s4 < 4 * NC; ++s4) acc1 = TGP_MFMA(af1[s4], bf[s4], acc1);
#pragma unroll
      for (int u = 0; u < 4; ++u) A[(16 * i + q + 4 * u) * LD + 16 * c + r] = cur0[u] - acc0[u];
      if (two) {
#pragma unroll
        for (int u = 0; u < 4; ++u) A[(16 * i1 + q + 4 * u) * LD + 16 * c + r] = cur1[u] - acc1[u];
      }
    }
  };
  auto catch_up_col = [&](int c, int ncol, int h, int halves) {
    static_for<MT - 2>([&](auto k) {
      if (ncol == decltype(k)::value + 1)
        catch_up_nc(c, std::integral_constant<int, decltype(k)::value + 1>{}, c + 2 * h, 2 * halves);
    });
  };
  // Dinv_jt (one wave, trtri16 on its own copy of L_jj) -> LDS and the diagonal tile of J
  auto inv_diag = [&](int jt) {
    double dgv[16], xv[16];
    const double rd = Dt[jt * 256 + r];   // 1 / L_jj[r][r], left here by the pass
#pragma unroll
    for (int c = 0; c < 16; ++c) dgv[c] = A[(16 * jt + r) * LD + 16 * jt + c];
    trtri16<true, true>(dgv, xv, r, rd);
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // xv[c] = Dinv[c][r]; the four 16-lane rows hold the same values: row q stores rows 4u + q
      const double xv4 = q == 0 ? xv[4 * u] : (q == 1 ? xv[4 * u + 1] : (q == 2 ? xv[4 * u + 2] : xv[4 * u + 3]));
      Dt[jt * 256 + (4 * u + q) * 16 + r] = xv4;
      Jb[(size_t)(16 * jt + 4 * u + q) * ld + 16 * jt + r] = xv4;
    }
  };
  // tile (jr, c) of J = -Dinv_jr (L[jr, c] Dinv_c + sum_{c < k < jr} L[jr, k] J[k, c]), NK = jr - c - 1 inner tiles:
  // transposed into the strict-upper tile (c, jr) of A for the rows below, and to global memory from the registers
  auto inv_tile_nk = [&](int jr, int c, auto nkc) {
    constexpr int NK = decltype(nkc)::value;
    const int j0 = 16 * jr, c0 = 16 * c;
    d4 acc = {0, 0, 0, 0};
    {
      double af[4], bf[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) { af[s4] = A[(j0 + r) * LD + c0 + 4 * s4 + q]; bf[s4] = Dt[c * 256 + (4 * s4 + q) * 16 + r]; }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = TGP_MFMA(af[s4], bf[s4], acc);
    }
    static_for<(NK + 2) / 3>([&](auto gc) {   // the inner tiles three at a time: 24 operand doubles live, reads ahead of the MFMAs
      constexpr int g0 = 3 * decltype(gc)::value, gn = (NK - g0 < 3 ? NK - g0 : 3);
      double af[4 * gn], bf[4 * gn];
#pragma unroll
      for (int s4 = 0; s4 < 4 * gn; ++s4) {
        const int k = c0 + 16 + 16 * g0 + 4 * s4;
        af[s4] = A[(j0 + r) * LD + k + q];
        bf[s4] = A[(c0 + r) * LD + k + q];
      }
#pragma unroll
      for (int s4 = 0; s4 < 4 * gn; ++s4) acc = TGP_MFMA(af[s4], bf[s4], acc);
    });
    double dj[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) dj[s4] = Dt[jr * 256 + r * 16 + 4 * s4 + q];
    d4 out = {0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) out = TGP_MFMA(dj[s4], acc[s4], out);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      A[(c0 + r) * LD + j0 + q + 4 * rr] = -out[rr];
      Jb[(size_t)(j0 + q + 4 * rr) * ld + c0 + r] = -out[rr];
    }
  };
  auto inv_tile = [&](int jr, int c) {
    static_for<MT - 1>([&](auto k) {
      if (jr - c - 1 == decltype(k)::value) inv_tile_nk(jr, c, k);
    });
  };
  // lower tile (ti, tj) of L -> global memory: two 16-byte stores per lane (lane = row 8 h + (lane >> 3), column pair
  // 2 (lane & 7)); the strict upper part of a diagonal tile as zeros
  const int rw = lane >> 3, cp = 2 * (lane & 7);
  auto write_L = [&](int ti, int tj) {
    double a[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      a[h][0] = A[(16 * ti + 8 * h + rw) * LD + 16 * tj + cp];
      a[h][1] = A[(16 * ti + 8 * h + rw) * LD + 16 * tj + cp + 1];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * h + rw;
      if (ti == tj) {
        if (cp > row) a[h][0] = 0.0;
        if (cp + 1 > row) a[h][1] = 0.0;
      }
      *reinterpret_cast<double2*>(Lb + (size_t)(16 * ti + row) * ld + 16 * tj + cp) = make_double2(a[h][0], a[h][1]);
    }
  };
  // zeros for the tiles of L right of the diagonal in tile row ti (the block arrives with K's symmetric copy resp. the
  // trailing update's values there; J's are zero from the fill kernel and nothing else writes them)
  auto zero_row = [&](int ti) {
    for (int tj = ti + 1; tj < MT; ++tj) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<double2*>(Lb + (size_t)(16 * ti + 8 * h + rw) * ld + 16 * tj + cp) = make_double2(0.0, 0.0);
    }
  };
#define POTRF_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  auto cu_tasks = [&](int j) { return (j >= 1 && j + 1 < MT) ? (MT - (j + 1) > 2 ? 2 : 1) : 0; };
  int tbase = 0;
  // windows MT and MT+1 have no panel left: every wave takes tasks -- the last diagonal tile's inverse, the last tile row
  // of L and row MT-2 of J, then row MT-1 of J (one loop: every task body has ONE call site and is inlined; called from a
  // tail of its own as well, the J tile came out as a function with its closure in scratch memory)
  for (int j = 0; j < MT + 2; ++j) {
    const int j0 = 16 * j;
    const int npan = (MT - 1 - j) * 16;
    const int npw = j >= MT ? 0 : (npan > 64 ? 2 : 1);   // panel waves (window MT-1: wave 0 with the diagonal tile alone)
    double ltile[4] = {0.0, 0.0, 0.0, 0.0};
    bool did_diag = false;
    // window j's tasks, heaviest first
    const int nt = (j >= 1 && j <= MT) ? 1 : 0;                      // Dinv_{j-1}
    const int ncu = cu_tasks(j);                                     // column j+1 -= columns 0 .. j-1
    const int nir = j >= 2 ? j - 2 : 0;                              // row j-2 of J left of its diagonal tile
    const int nwl = j <= MT ? j : 0;                                 // tile row j-1 of L
    const int nz = j == 0 ? MT - 1 : 0;
    const int ntask = nt + ncu + nir + nwl + nz;
    BSTAMP(2 + 3 * j);
    if (wave < npw) {
      __builtin_amdgcn_s_setprio(3);
      const int l0 = wave * 64 + lane;
      const bool has = l0 < npan;
      const int prow = j0 + (npan > 0 ? 16 : 0) + (has ? l0 : 0);
      double dg[16], a[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) dg[c] = A[(j0 + r) * LD + j0 + c];
      if (npan > 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = A[prow * LD + j0 + c];
      }
      int bad = 0;
      // (wave 0 keeps the pass's own reciprocals of the diagonal for inv_diag: the inverse then comes out exactly as when
      //  it rode in the pass -- a borderline pivot of a later block, K_MM with a duplicated inducing point, depends on it)
      double rd = 0.0;
      if (wave != 0) bad = potrf_panel16<true>(dg, a);
      else if (npan > 0) bad = potrf_panel16<true, false, true>(dg, a, nullptr, r, &rd);
      else bad = potrf_panel16<false, false, true>(dg, a, nullptr, r, &rd);
      if (wave == 0 && lane < 16) Dt[j * 256 + lane] = rd;
      if (has && npan > 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) A[prow * LD + j0 + c] = a[c];
      }
      if (wave == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          ltile[u] = q == 0 ? dg[4 * u] : (q == 1 ? dg[4 * u + 1] : (q == 2 ? dg[4 * u + 2] : dg[4 * u + 3]));
        did_diag = true;
        if (lane == 0 && bad != 0 && s_info == 0) s_info = j0 + bad;
      }
      __builtin_amdgcn_s_setprio(0);
    } else {
      for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(&s_next, 1);
        t = __builtin_amdgcn_readfirstlane(t) - tbase;
        if (t >= ntask) break;
        if (t < nt) { inv_diag(j - 1); continue; }
        t -= nt;
        if (t < ncu) { catch_up_col(j + 1, j, t, ncu); continue; }
        t -= ncu;
        if (t < nir) { inv_tile(j - 2, t); continue; }
        t -= nir;
        if (t < nwl) { write_L(j - 1, t); continue; }
        t -= nwl;
        zero_row(t);
      }
    }
    tbase += ntask + (NW - npw);   // the tasks + one over-grab per task wave
    BSTAMP(3 + 3 * j);
    POTRF_BARRIER();
    BSTAMP(4 + 3 * j);
    if (j >= MT) continue;
    if (did_diag) {
#pragma unroll
      for (int u = 0; u < 4; ++u) A[(j0 + r) * LD + j0 + 4 * u + q] = ltile[u];
    }
    if (j + 1 < MT) {
      for (int i = j + 1 + wave; i < MT; i += NW) sub16(i, j + 1, j0);
    }
    POTRF_BARRIER();
  }
#undef POTRF_BARRIER
  __syncthreads();
  BSTAMP(40);
  if (tid == 0 && s_info != 0 && status[0] == 0) status[0] = (int)o + s_info;
}

__global__ __launch_bounds__(POTRF_THREADS) void k_big_potrf(double* __restrict__ Lm, double* __restrict__ Jm, int ld, int kb,
                                                              int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  potrf_block<PB_GLOBAL>(Lm, Jm, ld, kb, status, smem_raw);
}

// K_MM and the factorisation of its first diagonal block in ONE launch (round 6).  Workgroups 0-31 first form the 128 x 128
// block (4 rows each, one element per thread, written through) and count themselves in status[4]; workgroup 32 polls that word and
// factorises the block (PB_SHARED) while all the others (grid-stride over whole rows, one workgroup per CU: the block's LDS sets
// the occupancy) write K_MM, the to-be-factorised copy outside that block, J = 0 and the masked L_q.  Same protocol and the same
// dispatch-order argument as k_fac_potrf (the producers are workgroups 0-31, the consumer is workgroup 32).  The chain's head was k_big_hdr -> k_big_zs -> k_big_kmm -> k_big_potrf
// (4.7 + 5.0 + 18.4 + 4.6 gap + 33 us); now k_big_hdr_zs -> this.
enum { KP_NFIRST = 32 };
template <int DPC>
__device__ __forceinline__ void big_kmm_rows(const BigPlan& p, const tgp_model& md, double* __restrict__ ws,
                                             int32_t* __restrict__ status, int w, int nw) {
  if (w < KP_NFIRST) {
    big_kmm_element<DPC, true>(p, md, ws, status, 4 * w + (int)(threadIdx.x >> 7), (int)(threadIdx.x & 127));
    handoff_barrier();
    if (threadIdx.x == 0) sync_add(status + FP_WORD, 1);
  }
  for (int row = w; row < p.MP; row += nw)
    for (int col = threadIdx.x; col < p.MP; col += POTRF_THREADS) big_kmm_element<DPC>(p, md, ws, status, row, col);
}
__global__ __launch_bounds__(POTRF_THREADS) void k_big_kmm_potrf(BigPlan p, tgp_model md, double* __restrict__ ws,
                                                                  int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (blockIdx.x == KP_NFIRST) {   // (behind its producers in dispatch order)
    if (threadIdx.x < 64) {
      const int v = sync_wait(status + FP_WORD, [](int x) { return x >= KP_NFIRST; });
      if (threadIdx.x == 0) {
        if (v == (int)0x80000000) {
          __hip_atomic_store(status, (int32_t)TGP_STATUS_SYNC_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(status + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sync_st(status + FP_WORD, 0);
      }
    }
    __syncthreads();
    potrf_block<PB_SHARED>(ws + p.Lm, ws + p.J, p.MP, 0, status, smem_raw);
    return;
  }
  const int w = (int)blockIdx.x - ((int)blockIdx.x > KP_NFIRST ? 1 : 0), nw = (int)gridDim.x - 1;
  if (p.DP == 4) big_kmm_rows<4>(p, md, ws, status, w, nw);
  else if (p.DP == 8) big_kmm_rows<8>(p, md, ws, status, w, nw);
  else big_kmm_rows<16>(p, md, ws, status, w, nw);
}

// ---------------------------------------------------------------------------------------------------
// Round 6: step kb's trailing update (+ the inverse's push) and the NEXT diagonal block's factorisation in ONE launch.
// k_big_potrf(kb + 1) needs the update of the next diagonal block only and used to wait for all of the update's tiles (105 at
// the first step) and for a launch boundary; now that block's ten lower 32 x 32 tiles are workgroups 0-9 (diag_tile32), they
// store write-through and count themselves in status[4] (the hand-off protocol of tgp_prep.hpp), workgroup 10 polls that word
// and runs the diagonal block beside the rest of the update.  Producers carry the lowest indices (dispatch order: no deadlock); the
// potrf workgroup clears the word (the next launch finds it zero); a wait that expires is reported like the fused path's
// (status[0] = TGP_STATUS_SYNC_TIMEOUT, sticky count in status[3]).  One block size for both roles: the k = 128 products are
// 4-wave tiles, so waves 4-7 of their workgroups leave at once (S_BARRIER waits on the surviving waves of a workgroup only).
// Per step: 3 launches -> 2, and the diagonal block off the update's tail.
// ---------------------------------------------------------------------------------------------------
enum { FP_NDIAG = 10 };   // the next diagonal block's 10 lower 32 x 32 tiles count themselves in status[FP_WORD]
#define FD_LD 132
// One 32 x 32 tile (i0, j0) of C -= A A^T over k = 128 in ONE stage (4 waves x 16 x 16; A = the panel, rows i0.. and j0..):
// the tiles in front of the next diagonal block are latency, not work -- one memory round trip, 32 MFMAs per wave, a
// write-through store (a 64 x 64 tile of fac_tile: two round trips and 128 MFMAs per wave, as long as the whole update launch).
__device__ __forceinline__ void diag_tile32(const FacJob& g, int i0, int j0, double* As, double* Bs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int wi = (wave >> 1) * 16, wj = (wave & 1) * 16;
  d2 ra[8], rb[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = tid + 256 * u, x = e >> 6, k2 = e & 63;            // 64 d2 per row of 128 k
    ra[u] = *reinterpret_cast<const d2*>(g.A + (size_t)(i0 + x) * g.lda + 2 * k2);
    rb[u] = *reinterpret_cast<const d2*>(g.B + (size_t)(j0 + x) * g.ldb + 2 * k2);
  }
  double cold[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) cold[rr] = g.C[(size_t)(i0 + wi + q + 4 * rr) * g.ldc + j0 + wj + r];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = tid + 256 * u, x = e >> 6, k2 = e & 63;
    *reinterpret_cast<d2*>(As + x * FD_LD + 2 * k2) = ra[u];
    *reinterpret_cast<d2*>(Bs + x * FD_LD + 2 * k2) = rb[u];
  }
  __syncthreads();
  d4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 32; ++s) acc = TGP_MFMA(As[(wi + r) * FD_LD + 4 * s + q], Bs[(wj + r) * FD_LD + 4 * s + q], acc);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
    st_agent(g.C + (size_t)(i0 + wi + q + 4 * rr) * g.ldc + j0 + wj + r, g.alpha * acc[rr] + g.beta * cold[rr]);
}

__global__ __launch_bounds__(POTRF_THREADS) void k_fac_potrf(FacJob a, FacJob b, int na, double* __restrict__ Lm,
                                                              double* __restrict__ Jm, int ld, int kb, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int id = blockIdx.x;
  if (id == FP_NDIAG) {
    if (threadIdx.x < 64) {
      const int v = sync_wait(status + FP_WORD, [](int x) { return x >= FP_NDIAG; });
      if (threadIdx.x == 0) {
        if (v == (int)0x80000000) {
          __hip_atomic_store(status, (int32_t)TGP_STATUS_SYNC_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(status + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sync_st(status + FP_WORD, 0);
      }
    }
    __syncthreads();
    potrf_block<PB_SHARED>(Lm, Jm, ld, kb, status, smem_raw);
    return;
  }
  if (threadIdx.x >= 256) return;
  double* As = reinterpret_cast<double*>(smem_raw);
  const int gx = a.n / 64;
  if (id < FP_NDIAG) {
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;          // lower 32 x 32 tile (ti, tj) of the next diagonal block
    const int tj = id - ti * (ti + 1) / 2;
    diag_tile32(a, 32 * ti, 32 * tj, As, As + 32 * FD_LD);
    handoff_barrier();
    if (threadIdx.x == 0) sync_add(status + FP_WORD, 1);
    return;
  }
  double* Bs = As + 128 * FAC_LD;
  const int t = id - FP_NDIAG - 1, nd = na - 3;          // the update's other tiles: (0,0), (1,0), (1,1) of 64 are the block's
  if (t < nd) fac_job(a, t < gx - 1 ? t + 1 : t + 3, As, Bs);
  else fac_job(b, t - nd, As, Bs);
}
#ifdef TGP_STAMPS
}  // namespace tgp
extern "C" int tgp_debug_potrf_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgp::g_potrf_stamps), sizeof(unsigned long long) * 8 * 48);
}
namespace tgp {
#endif

__global__ __launch_bounds__(256) void k_big_sub_eye(double* __restrict__ S, int MP) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < MP) S[(size_t)i * MP + i] -= 1.0;
}

// w = J^T m (J lower): thread per column
// w = J^T m.  One workgroup = 64 columns x 16 row ranges (a wave per range, 16 loads in flight per thread, the 16 range
// sums added in LDS in a fixed order): one thread per column walking all MP rows four loads at a time was 88 us of
// dependent L2 round trips in every prepare phase (2 % of the 10 000-row minibatch step).
#define WVEC_THREADS 1024
__global__ __launch_bounds__(WVEC_THREADS) void k_big_wvec(BigPlan p, double* __restrict__ ws) {
  __shared__ double part[16][64];
  const int jl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl, MP = p.MP;
  const double* __restrict__ J = ws + p.J;
  const double* __restrict__ m = ws + p.mpad;
  const int ib = blockIdx.x * 64;               // J[i][j] = 0 for i < j: start at the block's first column
  const int rows = MP - ib, per = (rows + 15) / 16;
  const int i0 = ib + rg * per, i1 = (i0 + per < MP) ? i0 + per : MP;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int i = i0; i < i1; i += 16) {
    double jv[16], mv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int iu = i + u < i1 ? i + u : i1 - 1;
      jv[u] = J[(size_t)iu * MP + j];
      mv[u] = i + u < i1 ? m[iu] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      s0 = fma(jv[u], mv[u], s0); s1 = fma(jv[u + 1], mv[u + 1], s1);
      s2 = fma(jv[u + 2], mv[u + 2], s2); s3 = fma(jv[u + 3], mv[u + 3], s3);
    }
  }
  part[rg][jl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0) {
    double s = 0.0;
#pragma unroll
    for (int r_ = 0; r_ < 16; ++r_) s += part[r_][jl];
    ws[p.w + j] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// row-chunk kernels (matrices [MP][NC], column n = data row c0 + n)
// ---------------------------------------------------------------------------------------------------
// Xaug[n][:] = [xs, xs^2, 1, 0...] (zero rows for the padding)
__global__ __launch_bounds__(256) void k_big_xaug(BigPlan p, const double* __restrict__ X, int nrows, double* __restrict__ ws) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int n = (int)(e >> 7), c = (int)(e & 127), DP = p.DP;
  if (n >= p.NC) return;
  double x = 0.0;
  if (n < nrows) {
    const int d = c < DP ? c : c - DP;
    if (c < 2 * DP) {
      if (d < p.D) {
        const double t = X[(size_t)n * p.D + d] * ws[p.ils + d];
        x = c < DP ? t : t * t;
      }
    } else if (c == 2 * DP) {
      x = 1.0;
    }
  }
  ws[p.Xaug + e] = x;
}

// K'[n][m] = k(xs_n, zs_m) (gweight: its derivative weight k_g instead); block = 32 data rows x 128 inducing columns
__global__ __launch_bounds__(256) void k_big_knm(BigPlan p, const double* __restrict__ X, int nrows, double* __restrict__ ws,
                                                  int gweight) {
  __shared__ double xl[32 * 16];
  const int tid = threadIdx.x, c = tid & 127, rg = tid >> 7, DP = p.DP;
  const int m = blockIdx.x * 128 + c, n0 = blockIdx.y * 32;
  for (int i = tid; i < 32 * DP; i += 256) {
    const int nl = i / DP, d = i % DP;
    int n = n0 + nl;
    n = n < nrows ? n : nrows - 1;  // padding rows repeat the last row: finite values, zero adjoints
    xl[i] = d < p.D ? X[(size_t)n * p.D + d] * ws[p.ils + d] : 0.0;
  }
  double zs[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) zs[d] = d < DP ? ws[p.Zs + (size_t)m * DP + d] : 0.0;
  __syncthreads();
  const double s2 = ws[p.hdr + H_S2];
  double* __restrict__ Kc = ws + p.Kc;
  for (int u = 0; u < 16; ++u) {
    const int nl = rg * 16 + u;
    double d2 = 0.0;
    for (int d = 0; d < DP; ++d) {
      const double t = xl[nl * DP + d] - zs[d];
      d2 += t * t;
    }
    Kc[(size_t)(n0 + nl) * p.MP + m] = m < p.M ? (gweight ? cov_gweight(p.kernel, s2, d2) : cov_value(p.kernel, s2, d2)) : 0.0;
  }
}

// mu_n = sum_m m_m A'_nm ; v_n = s2 - sum_m A'_nm^2 + sum_m B'_nm^2   (sparse_MF_SP.py:354-355,376-382); wave per row
__global__ __launch_bounds__(256) void k_big_moments(BigPlan p, double* __restrict__ ws, double* __restrict__ mu,
                                                      double* __restrict__ v, int nrows) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= nrows) return;
  const double* __restrict__ A = ws + p.A + (size_t)n * p.MP;
  const double* __restrict__ B = ws + p.B + (size_t)n * p.MP;
  const double* __restrict__ mp = ws + p.mpad;
  double sm = 0.0, sa = 0.0, sb = 0.0;
  for (int m = 2 * lane; m < p.MP; m += 128) {
    const d2 a = *reinterpret_cast<const d2*>(A + m), b = *reinterpret_cast<const d2*>(B + m);
    const d2 mm_ = *reinterpret_cast<const d2*>(mp + m);
    sm = fma(mm_[0], a[0], sm); sm = fma(mm_[1], a[1], sm);
    sa = fma(a[0], a[0], sa); sa = fma(a[1], a[1], sa);
    sb = fma(b[0], b[0], sb); sb = fma(b[1], b[1], sb);
  }
  sm = wave_sum(sm); sa = wave_sum(sa); sb = wave_sum(sb);
  if (lane == 0) {
    mu[n] = sm;
    v[n] = ws[p.hdr + H_S2] - sa + sb;
  }
}

// s_m partial (+)= sum_{n in row slab} A'[n][m] mubar_n : grid (MP/64, BIG_SSL), block = 64 columns x 4 row phases;
// block (0, BIG_SSL) accumulates sum_n vbar_n
__global__ __launch_bounds__(256) void k_big_coldot(BigPlan p, double* __restrict__ ws, size_t c0, int accumulate) {
  __shared__ double red[4][64];
  const int tid = threadIdx.x, c = tid & 63, g = tid >> 6;
  if ((int)blockIdx.y == BIG_SSL) {
    if (blockIdx.x != 0) return;
    const double* __restrict__ vb = ws + p.vb + c0;
    double s = 0.0;
    for (int n = tid; n < p.NC; n += 256) s += vb[n];
    s = wave_sum(s);
    if (c == 0) red[g][0] = s;
    __syncthreads();
    if (tid == 0) ws[p.svb] = (accumulate ? ws[p.svb] : 0.0) + ((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]));
    return;
  }
  const int m = blockIdx.x * 64 + c;
  const int per = (p.NC + BIG_SSL - 1) / BIG_SSL, n0 = blockIdx.y * per, n1 = min(p.NC, n0 + per);
  const double* __restrict__ A = ws + p.A;
  const double* __restrict__ mub = ws + p.mub + c0;
  double s0 = 0.0, s1 = 0.0;
  int n = n0 + g;
  for (; n + 4 < n1; n += 8) {
    s0 = fma(A[(size_t)n * p.MP + m], mub[n], s0);
    s1 = fma(A[(size_t)(n + 4) * p.MP + m], mub[n + 4], s1);
  }
  for (; n < n1; n += 4) s0 = fma(A[(size_t)n * p.MP + m], mub[n], s0);
  red[g][c] = s0 + s1;
  __syncthreads();
  if (g == 0) {
    double* o = ws + p.spart + (size_t)blockIdx.y * p.MP + m;
    *o = (accumulate ? *o : 0.0) + ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
  }
}

// slab reduction: G (lower -> full symmetric), T, s
__global__ __launch_bounds__(256) void k_big_reduce(BigPlan p, double* __restrict__ ws) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t mm = (size_t)p.MP * p.MP;
  if (e < mm) {
    const int i = (int)(e / p.MP), j = (int)(e % p.MP);
    if (j > i) return;
    double s = 0.0;
    for (int z = 0; z < p.ksg; ++z) s += ws[p.Gpart + z * mm + e];
    ws[p.G + e] = s;
    ws[p.G + (size_t)j * p.MP + i] = s;
  } else {
    const size_t t = e - mm;
    if (t < (size_t)p.MP * BIG_XW) {
      double s = 0.0;
      for (int z = 0; z < BIG_KST; ++z) s += ws[p.Tpart + (size_t)z * p.MP * BIG_XW + t];
      ws[p.T + t] = s;
      if (p.kernel != TGP_KERNEL_SCALE_RBF) {
        double sk = 0.0;
        for (int z = 0; z < BIG_KST; ++z) sk += ws[p.TpartK + (size_t)z * p.MP * BIG_XW + t];
        ws[p.TK + t] = sk;
      }
    } else if (t < (size_t)p.MP * BIG_XW + p.MP) {
      const size_t m = t - (size_t)p.MP * BIG_XW;
      double s = 0.0;
      for (int z = 0; z < BIG_SSL; ++z) s += ws[p.spart + (size_t)z * p.MP + m];
      ws[p.sv + m] = s;
    }
  }
}

// dst = sum of `ns` slabs of length len
__global__ __launch_bounds__(256) void k_big_sum_slabs(const double* __restrict__ src, int ns, size_t len, double* __restrict__ dst) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= len) return;
  double s = 0.0;
  for (int z = 0; z < ns; ++z) s += src[(size_t)z * len + e];
  dst[e] = s;
}

// C = beta C + sum of `ns` compact slabs [m][n]  (split-K reduction of an M x M product, fixed order), with the elementwise
// step that follows the product in the backward chain applied on the way out (a launch and a pass over M x M less each):
//   OP 1: C = -tril(u v^T + x)                (Lbar, k_big_lbar)
//   OP 2: C = Phi(x) + Phi(x)^T               (k_big_phisym: the lower triangle mirrored; strictly-upper sums are not formed)
template <int OP>
__global__ __launch_bounds__(256) void k_big_sum_slabs2d(const double* __restrict__ src, int ns, int m, int n, double* __restrict__ C,
                                                          int ldc, double beta, const double* __restrict__ u,
                                                          const double* __restrict__ v) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x, len = (size_t)m * n;
  if (e >= len) return;
  const int i = (int)(e / n), j = (int)(e % n);
  if (OP == 2 && j > i) return;
  double s = 0.0;
  for (int z0 = 0; z0 < ns; z0 += 8) {   // up to eight slabs requested together, added in slab order
    double x[8];
#pragma unroll
    for (int z = 0; z < 8; ++z) x[z] = z0 + z < ns ? src[(size_t)(z0 + z) * len + e] : 0.0;
#pragma unroll
    for (int z = 0; z < 8; ++z) s += x[z];
  }
  double* c = C + (size_t)i * ldc + j;
  if (OP == 1) {
    *c = j <= i ? -(u[i] * v[j] + s) : 0.0;
  } else if (OP == 2) {
    *c = s;
    if (j < i) C[(size_t)j * ldc + i] = s;
  } else {
    *c = beta == 0.0 ? s : s + beta * *c;
  }
}

// ---------------------------------------------------------------------------------------------------
// backward M x M elementwise kernels
// ---------------------------------------------------------------------------------------------------
// R1 (= 2 H' G) -> Lbar = -tril(w s^T + R1)
__global__ __launch_bounds__(256) void k_big_lbar(BigPlan p, double* __restrict__ ws) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(e / p.MP), j = (int)(e % p.MP);
  double x = 0.0;
  if (j <= i) x = -(ws[p.w + i] * ws[p.sv + j] + ws[p.R1 + e]);
  ws[p.R1 + e] = x;
}

// dELBO/dLam = 2 tril(G Lq) - kl (Lq - diag(1/Lam_ii)); R2 = 2 G Lq is in S_
// `upd`: torch.optim.Adam on Lam's entries in the thread that forms their gradient (k_adam_dev's arithmetic; the step counter
// is read here and advanced by the tail update behind k_big_final) -- M^2 of the parameters leave the end of the step's chain
__global__ __launch_bounds__(256) void k_big_glam(BigPlan p, tgp_model md, double* __restrict__ gLam, const double* __restrict__ GL,
                                                  AdamDev ad, int upd) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int M = p.M;
  if (e >= (size_t)M * M) return;
  const int row = (int)(e / M), col = (int)(e % M);
  double x = 0.0;
  if (col <= row) {
    const double lam = md.Lam[e];
    x = GL[(size_t)row * p.MP + col] - md.kl_scale * (col == row ? lam - 1.0 / lam : lam);   // GL = 2 G Lq
  }
  gLam[e] = x;
  if (upd) {
    const double step = (double)(ad.step_dev[0] + 1);
    const double bc1 = 1.0 - exp_fast(step * ad.ln_b1), bc2s = sqrt(1.0 - exp_fast(step * ad.ln_b2));
    const long i = ad.lam_off + (long)e;
    const double gi = ad.sign * x;
    const double mi = ad.b1 * ad.m[i] + (1.0 - ad.b1) * gi;
    const double vi = ad.b2 * ad.v[i] + (1.0 - ad.b2) * gi * gi;
    ad.m[i] = mi;
    ad.v[i] = vi;
    ad.p[i] -= (ad.lr / bc1) * mi / (sqrt(vi) / bc2s + ad.eps);
  }
}

// Q <- Phi(Q) + Phi(Q)^T in place (Phi: lower triangle, diagonal halved)
__global__ __launch_bounds__(256) void k_big_phisym(BigPlan p, double* __restrict__ ws) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(e / p.MP), j = (int)(e % p.MP);
  if (j >= i) return;  // diagonal: 2 * (1/2) Q_ii = Q_ii, unchanged
  ws[p.Q + (size_t)j * p.MP + i] = ws[p.Q + e];
}

// parameter gradients and the scalars (single block).  ONE workgroup on a cold instruction cache: its time is the number of
// memory round trips on its critical path (operands written by other XCDs a moment ago) plus the code it has to fetch, so
// the kernel is written for few of both -- one thread per (inducing point, dimension) element, four points in flight per
// thread, a small loop body instead of the 16-way unrolled per-dimension code of the first version (2 600 instructions
// fetched once each by a single workgroup: 39-50 us); the scalar partial sums are lane-parallel (a serial loop of global
// loads is a round trip per term), the column sums go through a wave butterfly + one LDS hop.
#define FINAL_THREADS 1024
__global__ __launch_bounds__(FINAL_THREADS) void k_big_final(BigPlan p, tgp_model md, tgp_grads g, double* __restrict__ out,
                                                              double* __restrict__ ws) {
  constexpr int NWV = FINAL_THREADS / 64;
  __shared__ double red[NWV][17];
  __shared__ double sc[4];
  static_assert(BIG_NKL <= 64, "one wave sums the KL partials");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, M = p.M, D = p.D, DP = p.DP;
  const int sh = DP == 4 ? 2 : (DP == 8 ? 3 : 4), d = tid & (DP - 1), jl = tid >> sh, rows = FINAL_THREADS >> sh;
  const bool vd = d < D;
  const double* __restrict__ hdr = ws + p.hdr;
  const double* __restrict__ Tm = ws + p.T;
  const double* __restrict__ Um = ws + p.U;
  const double* __restrict__ Zs = ws + p.Zs;
  const double* __restrict__ lik = ws + p.likslot;
  const double* __restrict__ svp = ws + p.sv;
  const bool rbf = p.kernel == TGP_KERNEL_SCALE_RBF;
  // ---- requests: scalars, this wave's partials ----
  const double s2 = hdr[H_S2], sig_os = hdr[H_SIG_OS], svb = ws[p.svb];
  const double ilsd = ws[p.ils + (vd ? d : 0)];
  const double rawls = md.raw_ls[tid < D ? tid : 0];
  double part = 0.0;
  if (wave == 1) part = lane < BIG_NKL ? ws[p.klpart + lane] : 0.0;
  if (wave == 2 || wave == 3)
    for (int c = lane; c < p.nchunks; c += 64) part += lik[(size_t)c * p.LS + (wave - 2)];
  double a = 0.0, aos = 0.0;
  for (int j0 = 0; j0 < M; j0 += 4 * rows) {
    double t1[4], t2[4], R[4], z[4], t0[4], cs[4], sv[4], mj[4], tk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * rows + jl, jc = j < M ? j : M - 1;
      const double* __restrict__ Tj = Tm + (size_t)jc * BIG_XW;
      const double* __restrict__ Uj = Um + (size_t)jc * BIG_XW;
      const int dc = vd ? d : 0;
      t1[u] = Tj[dc]; t2[u] = Tj[DP + dc]; R[u] = Uj[dc]; z[u] = Zs[(size_t)jc * DP + dc];
      t0[u] = Tj[2 * DP]; cs[u] = Uj[2 * DP];
      sv[u] = svp[jc]; mj[u] = md.m[jc];
      // d/d outputscale needs sum (Kbar o K); for the RBF K_g = K and it is the same ones-column
      tk[u] = rbf ? 0.0 : ws[p.TK + (size_t)jc * BIG_XW + 2 * DP] + ws[p.UK + (size_t)jc * BIG_XW + 2 * DP];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * rows + jl;
      const bool vj = j < M;
      const double gz = (t1[u] - z[u] * t0[u] + 2.0 * (R[u] - z[u] * cs[u])) * ilsd;
      const double c = (t2[u] - 2.0 * z[u] * t1[u] + z[u] * z[u] * t0[u]) + 2.0 * z[u] * (z[u] * cs[u] - R[u]);
      if (vj && vd) g.Z[(size_t)j * D + d] = gz;
      a += (vj && vd) ? c : 0.0;
      if (vj && d == 0) {
        aos += rbf ? cs[u] + t0[u] : tk[u];
        g.m[j] = sv[u] - md.kl_scale * mj[u];
      }
    }
  }
  if (g.theta != nullptr)
    for (int i = tid; i < p.P; i += FINAL_THREADS) {
      double s = 0.0;
      for (int c0 = 0; c0 < p.nchunks; c0 += 8) {   // eight chunks' slots in flight
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = c0 + u < p.nchunks ? lik[(size_t)(c0 + u) * p.LS + 2 + i] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += x[u];
      }
      g.theta[i] = s;
    }
  part = wave_sum(part);
  if (lane == 0 && wave < 4) sc[wave] = part;
  for (int o = DP; o < 64; o <<= 1) a += __shfl_xor(a, o);   // over the lanes of this wave that hold dimension d
  aos = wave_sum(aos);
  if (lane < DP) red[wave][lane] = a;
  if (lane == 0) red[wave][16] = aos;
  __syncthreads();
  if (tid < 17 && (tid < D || tid == 16)) {
    double s = 0.0;
    for (int w = 0; w < NWV; ++w) s += red[w][tid];
    if (tid < D) {
      g.raw_ls[tid] = s * ilsd * sigmoid_d(rawls);
    } else {
      const double kls = sc[1], ell = sc[2], etab = sc[3];
      const double kl = 0.5 * (kls - (double)M);
      g.raw_os[0] = (svb + s / s2) * sig_os;
      g.log_var_noise[0] = etab;
      out[0] = ell - kl;
      out[1] = ell;
      out[2] = kl;
      out[3] = 0.0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------------
#define GEMM(ta, tb, args)                                  \
  do {                                                      \
    if (int rc_ = launch_gemm((ta), (tb), (args), st)) return rc_; \
  } while (0)

// An M x M product has at most (MP/128)^2 <= 64 output tiles at M = 1000 -- a quarter of the CUs, each running the
// whole k loop.  Split k so that about 256 workgroups run, partial sums into compact slabs, then one fixed-order
// reduction into C (150 us -> ~50 us per product at MP = 1024).
// `op`: elementwise step fused into the slab reduction (see k_big_sum_slabs2d); returns 0 with *fused = false when the
// product was not split (the caller then runs the stand-alone elementwise kernel)
static int gemm_mm_on(bool ta, bool tb, GemmArgs g, double* scratch, size_t cap, hipStream_t st, int op = 0,
                      const double* u = nullptr, const double* v = nullptr, bool* fused = nullptr) {
  if (fused) *fused = false;
  const int tiles = (g.m / 128) * (g.n / 128);
  int ks = 256 / (tiles > 0 ? tiles : 1);
  if (ks > 8) ks = 8;
  if (ks > g.k / 64) ks = g.k / 64;  // at least four k stages per slab
  const size_t slab = (size_t)g.m * g.n;
  if (ks <= 1 || (size_t)ks * slab > cap || g.add || g.row_scale || g.col_scale || g.rowv) return launch_gemm(ta, tb, g, st);
  double* C = g.C;
  const int ldc = g.ldc;
  const double beta = g.beta;
  g.C = scratch; g.ldc = g.n; g.beta = 0.0; g.ksplit = ks; g.cz = slab;
  if (int rc = launch_gemm(ta, tb, g, st)) return rc;
  const dim3 grid((unsigned)((slab + 255) / 256));
  if (op == 1 && beta == 0.0) {
    hipLaunchKernelGGL(k_big_sum_slabs2d<1>, grid, dim3(256), 0, st, scratch, ks, g.m, g.n, C, ldc, beta, u, v);
    if (fused) *fused = true;
  } else if (op == 2 && beta == 0.0 && g.m == g.n) {
    hipLaunchKernelGGL(k_big_sum_slabs2d<2>, grid, dim3(256), 0, st, scratch, ks, g.m, g.n, C, ldc, beta, u, v);
    if (fused) *fused = true;
  } else {
    hipLaunchKernelGGL(k_big_sum_slabs2d<0>, grid, dim3(256), 0, st, scratch, ks, g.m, g.n, C, ldc, beta, u, v);
  }
  LAUNCH_CHECK();
  return 0;
}
static int gemm_mm(bool ta, bool tb, const GemmArgs& g, const BigPlan& p, double* ws, hipStream_t st, int op = 0,
                   const double* u = nullptr, const double* v = nullptr, bool* fused = nullptr) {
  const size_t cap = (size_t)8 * ((p.MP / 128) <= 4 ? (size_t)p.MP * p.MP : (size_t)p.MP * p.MP / 2);
  return gemm_mm_on(ta, tb, g, ws + p.Sk, cap, st, op, u, v, fused);
}

// ---------------------------------------------------------------------------------------------------
// psd_safe_cholesky's retry ladder (dsp/utils.py:256-269) for the general-M path, ON THE DEVICE: a captured step
// cannot ask the host to retry.  One launch after the blocked factorisation; it returns at once unless that
// factorisation reported a non-positive pivot (status[0] != 0) -- the rare case, served by a plain single-workgroup
// code path: K_MM + jitter_ladder * 10^i (i = 0..2) is refactorised column by column in global memory (right-looking,
// one barrier pair per column) and inverted by forward substitution (thread c owns column c of J = L^-1), tens of
// milliseconds at M = 1000.  status[2] = ladder level that succeeded, status[0] = 0 then; otherwise the last failing
// pivot stays in status[0] (the reference raises after its third retry).
// ---------------------------------------------------------------------------------------------------
#define LADDER_THREADS 1024
__global__ __launch_bounds__(LADDER_THREADS) void k_big_ladder(BigPlan p, tgp_model md, double* __restrict__ ws,
                                                               int32_t* __restrict__ status) {
  if (status[0] == 0 || status[1] != 0 || !(md.jitter_ladder > 0.0)) return;
  const int tid = threadIdx.x, MP = p.MP, M = p.M;
  const int ty = tid >> 6, tx = tid & 63;
  double* Lm = ws + p.Lm;
  double* J = ws + p.J;
  const double* __restrict__ K = ws + p.Kmm;
  __shared__ int s_bad;
  int level = 0, bad = 0;
  for (int attempt = 1; attempt <= 3 && level == 0; ++attempt) {
    const double add = md.jitter_ladder * (attempt == 1 ? 1.0 : (attempt == 2 ? 10.0 : 100.0));
    // working copy: lower triangle of K_MM (md.jitter is in it already) + the ladder's jitter on the diagonal
    for (int row = ty; row < MP; row += LADDER_THREADS / 64)
      for (int col = tx; col < MP; col += 64)
        Lm[(size_t)row * MP + col] = col <= row ? K[(size_t)row * MP + col] + ((row == col && row < M) ? add : 0.0) : 0.0;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int j = 0; j < MP; ++j) {
      const double d = Lm[(size_t)j * MP + j];           // the same value in every thread
      if (!(d > 0.0)) {
        if (tid == 0) s_bad = j + 1;
        break;
      }
      const double dj = sqrt(d), inv = 1.0 / dj;
      __syncthreads();                                    // everybody has read the pivot
      for (int i = j + tid; i < MP; i += LADDER_THREADS) Lm[(size_t)i * MP + j] = (i == j) ? dj : Lm[(size_t)i * MP + j] * inv;
      __syncthreads();
      for (int i = j + 1 + ty; i < MP; i += LADDER_THREADS / 64) {
        const double lij = Lm[(size_t)i * MP + j];
        for (int k = j + 1 + tx; k <= i; k += 64) Lm[(size_t)i * MP + k] -= lij * Lm[(size_t)k * MP + j];
      }
      __syncthreads();
    }
    __syncthreads();
    bad = s_bad;
    __syncthreads();
    if (bad == 0) level = attempt;
  }
  if (level != 0) {
    // J = L^-1 by forward substitution: thread c owns column c (it only re-reads what it wrote itself)
    for (int c = tid; c < MP; c += LADDER_THREADS) {
      J[(size_t)c * MP + c] = 1.0 / Lm[(size_t)c * MP + c];
      for (int i = c + 1; i < MP; ++i) {
        double s0 = 0.0, s1 = 0.0;
        int k = c;
        for (; k + 1 < i; k += 2) {
          s0 += Lm[(size_t)i * MP + k] * J[(size_t)k * MP + c];
          s1 += Lm[(size_t)i * MP + k + 1] * J[(size_t)(k + 1) * MP + c];
        }
        if (k < i) s0 += Lm[(size_t)i * MP + k] * J[(size_t)k * MP + c];
        J[(size_t)i * MP + c] = -(s0 + s1) / Lm[(size_t)i * MP + i];
      }
    }
  }
  if (tid == 0) {
    status[0] = level != 0 ? 0 : bad;
    status[2] = level;
  }
}

#define GEMM_MM(ta, tb, args)                                          \
  do {                                                                 \
    if (int rc_ = gemm_mm((ta), (tb), (args), p, ws, st)) return rc_;  \
  } while (0)

// A second stream inside one C-ABI call.  The M x M phases of this path are chains of small launches (one workgroup
// factorising a 128-column block, products with 8-64 output tiles) that leave most of the chip idle, and several of the
// chains do not depend on each other; they are forked onto an auxiliary stream by event record / wait -- valid in eager
// mode and under stream capture of the caller's stream (the auxiliary stream joins the capture through the event and is
// joined back before the call returns) -- so a captured step replays them as parallel branches of the graph.
struct BigFork {
  hipStream_t aux = nullptr;
  hipEvent_t ev[2] = {};    // 0: fork, 1: join
  int init() {
    if (aux != nullptr) return 0;
    if (hipError_t e = hipStreamCreateWithFlags(&aux, hipStreamNonBlocking); e != hipSuccess) return set_error(e, __FILE__, __LINE__);
    for (auto& x : ev)
      if (hipError_t e = hipEventCreateWithFlags(&x, hipEventDisableTiming); e != hipSuccess) return set_error(e, __FILE__, __LINE__);
    return 0;
  }
  // `to` continues after everything issued on `from` so far
  int after(int i, hipStream_t from, hipStream_t to) {
    if (hipError_t e = hipEventRecord(ev[i], from); e != hipSuccess) return set_error(e, __FILE__, __LINE__);
    if (hipError_t e = hipStreamWaitEvent(to, ev[i], 0); e != hipSuccess) return set_error(e, __FILE__, __LINE__);
    return 0;
  }
};
static BigFork& big_fork() {
  static thread_local BigFork f;  // per host thread, like big_aux()
  return f;
}

// the trailing update a (+ the inverse's push b, b.m == 0: none) and the factorisation of diagonal block kb in one launch
static int launch_fac_potrf(const FacJob& a, const FacJob& b, double* Lm, double* J, int MP, int kb, int32_t* status, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fac_potrf), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)POTRF_LDS_BYTES);
    if (e != hipSuccess) { (void)hipGetLastError(); return set_error(e, __FILE__, __LINE__); }
    attr_done = true;
  }
  static_assert(POTRF_LDS_BYTES >= 2 * 128 * FAC_LD * sizeof(double), "the product tiles stage through the potrf block's LDS");
  const int na = fac_tiles(a), nb2 = b.m > 0 ? fac_tiles(b) : 0;
  hipLaunchKernelGGL(k_fac_potrf, dim3(na - 3 + nb2 + FP_NDIAG + 1), dim3(POTRF_THREADS), POTRF_LDS_BYTES, st, a, b, na, Lm, J, MP, kb, status);
  LAUNCH_CHECK();
  return 0;
}

// Blocked right-looking Cholesky of the padded matrix in p.Lm (lower block triangle filled, J zeroed) and, if wanted, the
// inverse J = L^-1 (torch.cholesky, dsp/utils.py:239).  Diagonal blocks of J are always produced (the panel solve
// multiplies by them).  The inverse is right-looking too, J[i,j] = -J_ii sum_{j <= k < i} L[i,k] J[k,j]:
//   after k_big_potrf(k):  J[k, 0:k]  = -J_kk Acc[k, 0:k]                       (row k is final)
//   after the panel of k:  Acc[i, 0:k+1] += L[i,k] J[k, 0:k+1]  for all i > k    (Acc lives in J's own blocks below row k)
// so every product has k = 128 (no split-k, no scratch) and each rides in a launch the factorisation makes anyway
// (launch_gemm_pair_ft_ff: the row product beside the panel product, the push beside the trailing update): the inverse
// costs ONE launch at the very end instead of 3 (nb - 1) launches after the factorisation (330 us at M = 1000).
// Launches per 128-column step (round 6): [panel + inverse row] (k_fac_pair), [trailing update + push + the NEXT diagonal
// block] (k_fac_potrf); k_big_potrf alone only for block 0.
// `first_done`: block 0 was factorised by the launch that wrote the matrix (k_big_kmm_potrf).
static int big_factorise(const BigPlan& p, double* ws, int32_t* status, bool want_inverse, hipStream_t st, bool first_done = false) {
  const int MP = p.MP, nb = MP / 128;
  double* Lm = ws + p.Lm;
  double* J = ws + p.J;
  FacJob none;
  none.m = 0; none.n = 0; none.shape = 2;
  for (int kb = 0; kb < nb; ++kb) {
    if (kb == 0 && !first_done) {   // (the later diagonal blocks ride in the previous step's update launch: k_fac_potrf)
      hipLaunchKernelGGL(k_big_potrf, dim3(1), dim3(POTRF_THREADS), POTRF_LDS_BYTES, st, Lm, J, MP, kb, status);
      LAUNCH_CHECK();
    }
    const int rem = MP - (kb + 1) * 128;
    double* Jk = J + (size_t)kb * 128 * MP;                          // block row kb of J
    const double* Jkk = Jk + (size_t)kb * 128;
    // J[kb, 0:kb] = -J_kk Acc[kb, 0:kb]  in place (128 x 16 single-stage tiles: a workgroup owns its columns)
    const FacJob inv_row = fac_job_args(Jkk, MP, Jk, MP, Jk, MP, 128, 128 * kb, 0, 4, -1.0, 0.0);
    const bool row = want_inverse && kb >= 1;
    if (rem > 0) {
      double* panel = Lm + (size_t)(kb + 1) * 128 * MP + (size_t)kb * 128;
      // L[i,kb] = K[i,kb] J_kk^T  in place (16 x 128 single-stage tiles: a workgroup owns its rows)
      const FacJob pan = fac_job_args(panel, MP, Jkk, MP, panel, MP, rem, 128, 1, 3, 1.0, 0.0);
      // trailing update K[i,j] -= L[i,kb] L[j,kb]^T, lower part (64 x 64 tiles)
      double* trail = Lm + (size_t)(kb + 1) * 128 * MP + (size_t)(kb + 1) * 128;
      const FacJob upd = fac_job_args(panel, MP, panel, MP, trail, MP, rem, rem, 1, 2, -1.0, 1.0, 1);
      // Acc[i, 0:kb+1] += L[i,kb] J[kb, 0:kb+1]
      const FacJob push = fac_job_args(panel, MP, Jk, MP, J + (size_t)(kb + 1) * 128 * MP, MP, rem, 128 * (kb + 1), 0, 2, 1.0, 1.0);
      if (int rc = launch_fac_pair(pan, row ? inv_row : none, st)) return rc;
      if (int rc = launch_fac_potrf(upd, want_inverse ? push : none, Lm, J, MP, kb + 1, status, st)) return rc;
    } else if (row) {
      if (int rc = launch_fac_pair(inv_row, none, st)) return rc;
    }
  }
  return 0;
}

// `defer_hw`: H' = J^T S and w = J^T m (used by the backward M x M chain only) are left running on the auxiliary stream;
// the caller joins it (big_join) before that chain -- they then run beside the first row chunk instead of in front of it.
static int big_join(hipStream_t st) {
  BigFork& fk = big_fork();
  return fk.after(1, fk.aux, st);
}
static int big_chunk_kernel(const BigPlan& p, const double* Xc, int nrows, double* ws, bool train, hipStream_t st);

// `X0` (training step that runs prepare and rows in one call): the first chunk's K' tiles need nothing of the
// factorisation, only the scaled inducing points -- they are generated on the auxiliary stream under the first diagonal
// blocks (one workgroup factorises there, 255 CUs idle) instead of in front of the first row product.
static int big_prepare(const BigPlan& p, const tgp_model& md, double* ws, int32_t* status, bool train, hipStream_t st,
                       bool defer_hw = false, const double* X0 = nullptr, int nrows0 = 0) {
  const int MP = p.MP;
  const size_t mm = (size_t)MP * MP;
  BigFork& fk = big_fork();
  if (int rc = fk.init()) return rc;
  hipStream_t sx = fk.aux;
  hipLaunchKernelGGL(k_big_hdr_zs, dim3((unsigned)((size_t)MP * BIG_XW / 256)), dim3(256), 0, st, p, md, ws, status);
  LAUNCH_CHECK();
  {
    static bool attr_done = false;
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_big_kmm_potrf), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)POTRF_LDS_BYTES);
      if (e != hipSuccess) { (void)hipGetLastError(); return set_error(e, __FILE__, __LINE__); }
      attr_done = true;
    }
    hipLaunchKernelGGL(k_big_kmm_potrf, dim3(256), dim3(POTRF_THREADS), POTRF_LDS_BYTES, st, p, md, ws, status);
    LAUNCH_CHECK();
  }
  // ---- fork: what needs only the variational parameters (KL, S = Lq Lq^T - I) runs beside the factorisation.  The
  //      factorisation is ISSUED FIRST: a captured graph keeps the first-created successor of a fork on the parent's
  //      hardware queue and moves the later ones to another -- with the auxiliary branch created first the critical
  //      chain hopped queues behind k_big_kmm and again in front of the row phase, ~10 us of dependency latency each
  //      time (q1 -> q4 -> q1 -> q4 in the kernel timeline; now one hop, where the row phase meets the early K') ----
  if (hipError_t e = hipEventRecord(fk.ev[0], st); e != hipSuccess) return set_error(e, __FILE__, __LINE__);
  if (int rc = big_factorise(p, ws, status, true, st, true)) return rc;
  if (md.jitter_ladder > 0.0) {  // the device-side retry ladder: returns at once unless the factorisation failed
    hipLaunchKernelGGL(k_big_ladder, dim3(1), dim3(LADDER_THREADS), 0, st, p, md, ws, status);
    LAUNCH_CHECK();
  }
  if (hipError_t e = hipStreamWaitEvent(sx, fk.ev[0], 0); e != hipSuccess) return set_error(e, __FILE__, __LINE__);
  if (train) {
    hipLaunchKernelGGL(k_big_kl, dim3(BIG_NKL), dim3(256), 0, sx, p, md, ws);
    LAUNCH_CHECK();
    const double* Lq = ws + p.Lq;
    // (split-k through the row phase's G partials, idle until the first chunk: the scratch of gemm_mm belongs to the main
    //  stream; split, the product fits under the first diagonal block's factorisation instead of slowing the first panel)
    if (int rc = gemm_mm_on(false, true, gemm_args(Lq, MP, Lq, MP, ws + p.S_, MP, MP, MP, MP, 1.0, 0.0, TRI_A_LOWER | TRI_B_UPPER),
                            ws + p.Gpart, (size_t)p.ksg * mm, sx))
      return rc;
    hipLaunchKernelGGL(k_big_sub_eye, dim3(MP / 256 + 1), dim3(256), 0, sx, ws + p.S_, MP);
    LAUNCH_CHECK();
    if (X0 != nullptr)
      if (int rc = big_chunk_kernel(p, X0, nrows0, ws, true, sx)) return rc;
  }
  if (int rc = fk.after(1, sx, st)) return rc;   // join
  if (!train) return 0;
  // H' = J^T S (after the ladder: a retry rewrites J)
  hipStream_t sh = st;
  if (defer_hw) {
    if (int rc = fk.after(0, st, sx)) return rc;
    sh = sx;
  }
  // (split-k either way: the scratch of gemm_mm is not used by the row phase, and the main stream's next user -- the
  //  backward chain -- comes after the join; unsplit, the 64 long workgroups slowed the first row product by 90 us)
  if (int rc = gemm_mm(true, false, gemm_args(ws + p.J, MP, ws + p.S_, MP, ws + p.Hp, MP, MP, MP, MP, 1.0, 0.0, TRI_A_UPPER), p, ws, sh))
    return rc;
  hipLaunchKernelGGL(k_big_wvec, dim3(MP / 64), dim3(WVEC_THREADS), 0, sh, p, ws);
  LAUNCH_CHECK();
  return 0;
}

// K' tiles of one chunk (and, for a training step, its augmented coordinates)
static int big_chunk_kernel(const BigPlan& p, const double* Xc, int nrows, double* ws, bool train, hipStream_t st) {
  const int MP = p.MP, NC = p.NC;
  if (train) {
    hipLaunchKernelGGL(k_big_xaug, dim3((unsigned)((size_t)NC * BIG_XW / 256)), dim3(256), 0, st, p, Xc, nrows, ws);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_big_knm, dim3(MP / 128, NC / 32), dim3(256), 0, st, p, Xc, nrows, ws, 0);
  LAUNCH_CHECK();
  return 0;
}

// forward part of one chunk: Kc (unless `have_k`: made during the factorisation), A, B, moments
static int big_chunk_forward(const BigPlan& p, const double* Xc, int nrows, double* ws, double* mu, double* v, bool train,
                             hipStream_t st, bool have_k = false) {
  const int MP = p.MP, NC = p.NC;
  if (!have_k)
    if (int rc = big_chunk_kernel(p, Xc, nrows, ws, train, st)) return rc;
  // A' = K' J^T (J^T upper), B' = A' Lq (Lq lower); the 8 column tiles of a row block share an XCD
  GemmArgs a1 = gemm_args(ws + p.Kc, MP, ws + p.J, MP, ws + p.A, MP, NC, MP, MP, 1.0, 0.0, TRI_B_UPPER);
  a1.xcd = 1;
  GEMM(false, true, a1);
  GemmArgs a2 = gemm_args(ws + p.A, MP, ws + p.Lq, MP, ws + p.B, MP, NC, MP, MP, 1.0, 0.0, TRI_B_LOWER);
  a2.xcd = 1;
  GEMM(false, false, a2);
  hipLaunchKernelGGL(k_big_moments, dim3((nrows + 3) / 4), dim3(256), 0, st, p, ws, mu, v, nrows);
  LAUNCH_CHECK();
  return 0;
}

int launch_big_moments(const tgp_model& md, const double* X, double* mu, double* v, int32_t* status, double* ws,
                       size_t ws_doubles, hipStream_t st) {
  BigPlan p;
  if (int rc = make_big_plan(p, md.N, md.D, md.M, 1, 0, 0, 0, TGP_LIK_GAUSS, md.kernel, md.plan)) return rc;
  if (ws_doubles < p.total) return TGP_E_WORKSPACE;
  if (int rc = big_prepare(p, md, ws, status, false, st)) return rc;
  for (int ci = 0; ci < p.nchunks; ++ci) {
    const size_t c0 = (size_t)ci * p.NC;
    const int nrows = (int)((size_t)p.N - c0 < (size_t)p.NC ? (size_t)p.N - c0 : (size_t)p.NC);
    if (int rc = big_chunk_forward(p, X + c0 * p.D, nrows, ws, mu + c0, v + c0, false, st)) return rc;
  }
  return 0;
}

int launch_big_step(const tgp_model& md, const FlowProg& fp, const double* X, const double* Y, const double* rowp,
                    double* out, const tgp_grads& g, double* mu, double* v, int32_t* status, double* ws, size_t ws_doubles,
                    uint32_t phases, hipStream_t st, const AdamDev* adam) {
  BigPlan p;
  if (int rc = make_big_plan(p, md.N, md.D, md.M, md.S, md.nblk, md.P, md.RP, md.lik, md.kernel, md.plan)) return rc;
  if (ws_doubles < p.total) return TGP_E_WORKSPACE;
  const int MP = p.MP, NC = p.NC;
  const size_t mm = (size_t)MP * MP;
  const bool defer_hw = (phases & TGP_PHASE_PREPARE) && (phases & TGP_PHASE_ROWS) && (phases & TGP_PHASE_BACKWARD);
  const bool early_k = (phases & TGP_PHASE_PREPARE) && (phases & TGP_PHASE_ROWS);
  if (phases & TGP_PHASE_PREPARE) {
    const int n0 = p.N < NC ? p.N : NC;
    if (int rc = big_prepare(p, md, ws, status, true, st, defer_hw, early_k ? X : nullptr, n0)) return rc;
  }
  if (phases & TGP_PHASE_ROWS) {
    // Chunk pipeline.  With a second set of chunk buffers the forward half of chunk c+1 (K' tiles, A', B', moments,
    // likelihood -- a third of it bandwidth/latency-bound kernels that leave the matrix cores idle) runs on a helper
    // stream while the caller's stream runs the backward half of chunk c; events order buffer reuse.  Under hipGraph
    // capture the helper stream forks from and rejoins the capturing stream.
    BigAux* aux = p.cstride ? big_aux() : nullptr;
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return set_error(e_, __FILE__, __LINE__); } while (0)
    if (aux) {
      HIPCK(hipEventRecord(aux->e0, st));
      HIPCK(hipStreamWaitEvent(aux->fwd, aux->e0, 0));
    }
    for (int ci = 0; ci < p.nchunks; ++ci) {
      const size_t c0 = (size_t)ci * NC;
      const int nrows = (int)((size_t)p.N - c0 < (size_t)NC ? (size_t)p.N - c0 : (size_t)NC);
      const int par = aux ? (ci & 1) : 0;
      const BigPlan pc = plan_parity(p, par);
      hipStream_t sf = aux ? aux->fwd : st;
      if (aux && ci >= 2) HIPCK(hipStreamWaitEvent(sf, aux->eB[par], 0));  // chunk ci-2 is done with these buffers
      if (int rc = big_chunk_forward(pc, X + c0 * p.D, nrows, ws, ws + p.mu + c0, ws + p.v + c0, true, sf, early_k && ci == 0)) return rc;
      // likelihood of the chunk: partial (scale*ELL, scale*eta_bar, theta_bar) into this chunk's slot
      double* slot = ws + p.likslot + (size_t)ci * p.LS;
      if (md.lik == TGP_LIK_FLOW) {
        tgp_model mc = md;
        mc.N = nrows;
        if (int rc = launch_ell_flow(mc, fp, Y + c0, ws + p.mu + c0, ws + p.v + c0, rowp ? rowp + c0 * md.RP : nullptr, slot,
                                     ws + p.mub + c0, ws + p.vb + c0, slot + 2, g.rowp ? g.rowp + c0 * md.RP : nullptr,
                                     ws + p.likws, sf))
          return rc;
      } else if (md.lik == TGP_LIK_ADJOINT) {
        // tgp_qf_moments_bwd_f64: the adjoints are the caller's (mu_bar in the Y slot, v_bar in the rowp slot)
        hipError_t e = hipMemcpyAsync(ws + p.mub + c0, Y + c0, (size_t)nrows * sizeof(double), hipMemcpyDeviceToDevice, sf);
        if (e == hipSuccess) e = hipMemcpyAsync(ws + p.vb + c0, rowp + c0, (size_t)nrows * sizeof(double), hipMemcpyDeviceToDevice, sf);
        if (e == hipSuccess) e = hipMemsetAsync(slot, 0, (size_t)p.LS * sizeof(double), sf);
        if (e != hipSuccess) return set_error(e, __FILE__, __LINE__);
      } else {
        if (int rc = launch_ell_gauss(Y + c0, ws + p.mu + c0, ws + p.v + c0, nrows, md.log_var_noise, md.scale, slot,
                                      ws + p.mub + c0, ws + p.vb + c0, ws + p.likws, sf))
          return rc;
      }
      if (aux) {
        HIPCK(hipEventRecord(aux->eF[par], sf));
        HIPCK(hipStreamWaitEvent(st, aux->eF[par], 0));
      }
      // ---- backward half, caller's stream ----
      // (the G SYRK and s = A'^T mubar need only A', vbar, mubar and could run beside the chain Abar' -> Kbar' -> T on the
      //  fork stream: measured, the two branches take exactly the sum of their solo times -- the launches are bound by
      //  matrix throughput, not by idle CUs -- so they stay in line)
      // Abar' = vbar o (2 B' Lq^T - 2 A') + mubar m^T
      GemmArgs a3 = gemm_args(ws + pc.B, MP, ws + p.Lq, MP, ws + pc.Ab, MP, NC, MP, MP, 2.0, 0.0, TRI_B_UPPER);
      a3.add = ws + pc.A; a3.ldadd = MP; a3.gamma = -2.0;
      a3.row_scale = ws + p.vb + c0; a3.rowv = ws + p.mub + c0; a3.colv = ws + p.mpad;
      a3.xcd = 1;
      GEMM(false, true, a3);
      // Kbar' = Abar' J  (into the B' buffer)
      GemmArgs a4 = gemm_args(ws + pc.Ab, MP, ws + p.J, MP, ws + pc.B, MP, NC, MP, MP, 1.0, 0.0, TRI_B_LOWER);
      a4.xcd = 1;
      GEMM(false, false, a4);
      // T slabs (+)= (Kbar' o K'_g)^T Xaug.  RBF: K'_g = K'.  MATERN32: first the statistics with K' itself (only their
      // ones column is used: d/d outputscale), then K' is overwritten by its derivative weight K'_g
      GemmArgs at = gemm_args(ws + pc.B, MP, ws + pc.Xaug, BIG_XW, ws + p.Tpart, BIG_XW, MP, BIG_XW, NC, 1.0, ci ? 1.0 : 0.0);
      at.a_mul = ws + pc.Kc; at.ksplit = BIG_KST; at.cz = (size_t)MP * BIG_XW; at.xcd = 2;
      if (p.kernel != TGP_KERNEL_SCALE_RBF) {
        GemmArgs atk = at;
        atk.C = ws + p.TpartK;
        GEMM(true, false, atk);
        hipLaunchKernelGGL(k_big_knm, dim3(MP / 128, NC / 32), dim3(256), 0, st, pc, X + c0 * p.D, nrows, ws, 1);
        LAUNCH_CHECK();
      }
      GEMM(true, false, at);
      // G slabs (+)= A'^T diag(vbar) A', lower block triangle
      GemmArgs ag = gemm_args(ws + pc.A, MP, ws + pc.A, MP, ws + p.Gpart, MP, MP, MP, NC, 1.0, ci ? 1.0 : 0.0, TRI_C_LOWER);
      ag.k_scale = ws + p.vb + c0; ag.ksplit = p.ksg; ag.cz = mm; ag.xcd = 2;
      GEMM(true, false, ag);
      hipLaunchKernelGGL(k_big_coldot, dim3(MP / 64, BIG_SSL + 1), dim3(256), 0, st, pc, ws, c0, ci ? 1 : 0);
      LAUNCH_CHECK();
      if (aux) HIPCK(hipEventRecord(aux->eB[par], st));
    }
#undef HIPCK
    hipLaunchKernelGGL(k_big_reduce, dim3((unsigned)((mm + (size_t)MP * BIG_XW + MP + 255) / 256)), dim3(256), 0, st, p, ws);
    LAUNCH_CHECK();
    if (mu != nullptr) {
      hipError_t e = hipMemcpyAsync(mu, ws + p.mu, (size_t)p.N * sizeof(double), hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(v, ws + p.v, (size_t)p.N * sizeof(double), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) return set_error(e, __FILE__, __LINE__);
    }
  }
  if (phases & TGP_PHASE_BACKWARD) {
    const unsigned gmm = (unsigned)(mm / 256);
    BigFork& fk = big_fork();
    if (int rc = fk.init()) return rc;
    if (defer_hw)
      if (int rc = big_join(st)) return rc;   // H', w
    // dLam = tril(2 G Lq) - ... needs G only: on the auxiliary stream, beside the chain Lbar -> Q -> Kbar_MM -> U.  The fork
    // point is recorded here, the auxiliary branch is ISSUED AFTER the chain (see big_prepare: the first-created successor
    // of a fork keeps the parent's hardware queue under graph replay, and that must be the critical chain).
    if (hipError_t e = hipEventRecord(fk.ev[0], st); e != hipSuccess) return set_error(e, __FILE__, __LINE__);
    // Lbar = -tril(w s^T + 2 H' G)
    bool fused = false;
    if (int rc = gemm_mm(false, false, gemm_args(ws + p.Hp, MP, ws + p.G, MP, ws + p.R1, MP, MP, MP, MP, 2.0, 0.0), p, ws, st, 1,
                         ws + p.w, ws + p.sv, &fused))
      return rc;
    if (!fused) {
      hipLaunchKernelGGL(k_big_lbar, dim3(gmm), dim3(256), 0, st, p, ws);
      LAUNCH_CHECK();
    }
    // Q = Phi(L^T Lbar) + Phi(.)^T
    if (int rc = gemm_mm(true, false, gemm_args(ws + p.Lm, MP, ws + p.R1, MP, ws + p.Q, MP, MP, MP, MP, 1.0, 0.0, TRI_A_UPPER | TRI_B_LOWER),
                         p, ws, st, 2, nullptr, nullptr, &fused))
      return rc;
    if (!fused) {
      hipLaunchKernelGGL(k_big_phisym, dim3(gmm), dim3(256), 0, st, p, ws);
      LAUNCH_CHECK();
    }
    // Kbar_MM = 1/2 J^T Q J
    GEMM_MM(false, false, gemm_args(ws + p.Q, MP, ws + p.J, MP, ws + p.S_, MP, MP, MP, MP, 1.0, 0.0, TRI_B_LOWER));
    GEMM_MM(true, false, gemm_args(ws + p.J, MP, ws + p.S_, MP, ws + p.R1, MP, MP, MP, MP, 0.5, 0.0, TRI_A_UPPER));
    // U = (Kbar_MM o K_MM,g) Zaug  (MATERN32: also with K_MM itself, for d/d outputscale)
    GemmArgs au = gemm_args(ws + p.R1, MP, ws + p.Zaug, BIG_XW, ws + p.Tpart, BIG_XW, MP, BIG_XW, MP, 1.0, 0.0);
    au.ksplit = 8; au.cz = (size_t)MP * BIG_XW;
    if (p.kernel != TGP_KERNEL_SCALE_RBF) {
      au.a_mul = ws + p.Kmm;
      GEMM(false, false, au);
      hipLaunchKernelGGL(k_big_sum_slabs, dim3((unsigned)((size_t)MP * BIG_XW / 256)), dim3(256), 0, st, ws + p.Tpart, 8,
                         (size_t)MP * BIG_XW, ws + p.UK);
      LAUNCH_CHECK();
      au.a_mul = ws + p.Kmmg;
    } else {
      au.a_mul = ws + p.Kmm;
    }
    GEMM(false, false, au);
    hipLaunchKernelGGL(k_big_sum_slabs, dim3((unsigned)((size_t)MP * BIG_XW / 256)), dim3(256), 0, st, ws + p.Tpart, 8,
                       (size_t)MP * BIG_XW, ws + p.U);
    LAUNCH_CHECK();
    if (hipError_t e = hipStreamWaitEvent(fk.aux, fk.ev[0], 0); e != hipSuccess) return set_error(e, __FILE__, __LINE__);
    {
      // (product and split-k slabs in the row phase's G partials, reduced into G by now)
      double* dl = ws + p.Gpart;
      const size_t cap = p.ksg > 1 ? (size_t)(p.ksg - 1) * mm : 0;
      if (int rc = gemm_mm_on(false, false, gemm_args(ws + p.G, MP, ws + p.Lq, MP, dl, MP, MP, MP, MP, 2.0, 0.0, TRI_B_LOWER), dl + mm, cap,
                              fk.aux))
        return rc;
      hipLaunchKernelGGL(k_big_glam, dim3((unsigned)(((size_t)p.M * p.M + 255) / 256)), dim3(256), 0, fk.aux, p, md, g.Lam, dl,
                         adam != nullptr ? *adam : AdamDev(), adam != nullptr ? 1 : 0);
      LAUNCH_CHECK();
    }
    if (int rc = big_join(st)) return rc;   // dLam
    hipLaunchKernelGGL(k_big_final, dim3(1), dim3(FINAL_THREADS), 0, st, p, md, g, out, ws);
    LAUNCH_CHECK();
  }
  return 0;
}

// ---- stand-alone Cholesky for M > 128 (tgp_cholesky_f64): pad, factorise, unpad --------------------------------
__global__ __launch_bounds__(256) void k_big_chol_in(BigPlan p, const double* __restrict__ A, double* __restrict__ ws,
                                                     int32_t* __restrict__ status) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int MP = p.MP, M = p.M;
  const int row = (int)(e / MP), col = (int)(e % MP);
  if (e == 0) { status[0] = 0; status[1] = 0; }
  double k = row == col ? 1.0 : 0.0;
  if (row < M && col < M) {
    k = A[(size_t)row * M + col];
    if (k != k) status[1] = 1;
  }
  ws[p.Lm + e] = (row >> 7) >= (col >> 7) ? k : 0.0;
  ws[p.J + e] = 0.0;
}
__global__ __launch_bounds__(256) void k_big_chol_out(BigPlan p, const double* __restrict__ ws, double* __restrict__ Lo,
                                                      double* __restrict__ Jo) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int M = p.M;
  if (e >= (size_t)M * M) return;
  const int row = (int)(e / M), col = (int)(e % M);
  const size_t src = (size_t)row * p.MP + col;
  Lo[e] = col <= row ? ws[p.Lm + src] : 0.0;
  if (Jo) Jo[e] = col <= row ? ws[p.J + src] : 0.0;
}

size_t big_cholesky_workspace_doubles(int M) {
  BigPlan p;
  if (make_big_plan(p, 128, 1, M, 1, 0, 0, 0, TGP_LIK_GAUSS, TGP_KERNEL_SCALE_MATERN32) != 0) return 0;  // any-M plan
  return p.total;
}

int launch_big_cholesky(const double* A, int M, double* Lo, double* Jo, int32_t* status, double* ws, size_t ws_doubles,
                        hipStream_t st) {
  BigPlan p;
  if (int rc = make_big_plan(p, 128, 1, M, 1, 0, 0, 0, TGP_LIK_GAUSS, TGP_KERNEL_SCALE_MATERN32)) return rc;
  if (ws_doubles < p.total) return TGP_E_WORKSPACE;
  const size_t mm = (size_t)p.MP * p.MP;
  hipLaunchKernelGGL(k_big_chol_in, dim3((unsigned)(mm / 256)), dim3(256), 0, st, p, A, ws, status);
  LAUNCH_CHECK();
  if (int rc = big_factorise(p, ws, status, Jo != nullptr, st)) return rc;
  hipLaunchKernelGGL(k_big_chol_out, dim3((unsigned)(((size_t)M * M + 255) / 256)), dim3(256), 0, st, p, ws, Lo, Jo);
  LAUNCH_CHECK();
  return 0;
}

// ---- adjoint of the Cholesky factorisation (tgp_cholesky_bwd_f64), any M: pad, the backward chain's three products, unpad ----
//   Abar = 1/2 J^T (Phi(L^T Lbar) + Phi(.)^T) J,   J = L^-1     (symmetric: what torch's cholesky backward returns)
__global__ __launch_bounds__(256) void k_big_cholbwd_in(BigPlan p, const double* __restrict__ L, const double* __restrict__ Li,
                                                        const double* __restrict__ Lbar, double* __restrict__ ws) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int MP = p.MP, M = p.M;
  const int row = (int)(e / MP), col = (int)(e % MP);
  const bool in = row < M && col < M && col <= row;
  const size_t s = (size_t)row * M + col;
  ws[p.Lm + e] = in ? L[s] : (row == col ? 1.0 : 0.0);   // identity on the padding: L^T Lbar and J stay block diagonal there
  ws[p.J + e] = in ? Li[s] : (row == col ? 1.0 : 0.0);
  ws[p.R1 + e] = in ? Lbar[s] : 0.0;                      // the factor's adjoint counts on and below the diagonal only
}
__global__ __launch_bounds__(256) void k_big_cholbwd_out(BigPlan p, const double* __restrict__ ws, double* __restrict__ Abar) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int M = p.M;
  if (e >= (size_t)M * M) return;
  Abar[e] = ws[p.R1 + (size_t)(e / M) * p.MP + e % M];
}

int launch_big_cholesky_bwd(const double* L, const double* Linv, const double* Lbar, int M, double* Abar, double* ws,
                            size_t ws_doubles, hipStream_t st) {
  BigPlan p;
  if (int rc = make_big_plan(p, 128, 1, M, 1, 0, 0, 0, TGP_LIK_GAUSS, TGP_KERNEL_SCALE_MATERN32)) return rc;
  if (ws_doubles < p.total) return TGP_E_WORKSPACE;
  const int MP = p.MP;
  const size_t mm = (size_t)MP * MP;
  const unsigned gmm = (unsigned)(mm / 256);
  hipLaunchKernelGGL(k_big_cholbwd_in, dim3(gmm), dim3(256), 0, st, p, L, Linv, Lbar, ws);
  LAUNCH_CHECK();
  bool fused = false;
  if (int rc = gemm_mm(true, false, gemm_args(ws + p.Lm, MP, ws + p.R1, MP, ws + p.Q, MP, MP, MP, MP, 1.0, 0.0, TRI_A_UPPER | TRI_B_LOWER),
                       p, ws, st, 2, nullptr, nullptr, &fused))
    return rc;
  if (!fused) {
    hipLaunchKernelGGL(k_big_phisym, dim3(gmm), dim3(256), 0, st, p, ws);
    LAUNCH_CHECK();
  }
  GEMM_MM(false, false, gemm_args(ws + p.Q, MP, ws + p.J, MP, ws + p.S_, MP, MP, MP, MP, 1.0, 0.0, TRI_B_LOWER));
  GEMM_MM(true, false, gemm_args(ws + p.J, MP, ws + p.S_, MP, ws + p.R1, MP, MP, MP, MP, 0.5, 0.0, TRI_A_UPPER));
  hipLaunchKernelGGL(k_big_cholbwd_out, dim3((unsigned)(((size_t)M * M + 255) / 256)), dim3(256), 0, st, p, ws, Abar);
  LAUNCH_CHECK();
  return 0;
}

// diagnostic / test entry: plain GEMM on padded operands
int launch_gemm_plain(bool ta, bool tb, int tri, int m, int n, int k, double alpha, const double* A, int lda, const double* B,
                      int ldb, double beta, double* C, int ldc, hipStream_t st) {
  return launch_gemm(ta, tb, gemm_args(A, lda, B, ldb, C, ldc, m, n, k, alpha, beta, tri), st);
}

}  // namespace tgp
