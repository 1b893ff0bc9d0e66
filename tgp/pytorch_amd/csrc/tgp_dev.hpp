// tgp_dev.hpp -- shared device helpers and the workspace layout (gfx950 only, float64).
//
// Wave = 64 lanes.  All dense contraction goes through v_mfma_f64_16x16x4_f64:
//   A operand: lane l holds A[i = l&15][k = l>>4]        (one f64)
//   B operand: lane l holds B[k = l>>4][j = l&15]        (one f64)
//   C/D      : lane l, register r holds D[row = (l>>4) + 4r][col = l&15]
// (layout verified on hardware with exact integer data, tools/probes/mfma_probe.hip).  Consequence used
// everywhere below: accumulator register r of a 16x16 tile IS the B operand of k-step r of a product
// that contracts over the tile's ROW index -- GEMM chains need no LDS round trip.
#pragma once
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/tgp_hip.h"

typedef double d4 __attribute__((ext_vector_type(4)));
#define TGP_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#define TGP_ROWS_PER_BLOCK 64 /* 4 waves x 16 rows */
#define TGP_TILE_LD 66        /* LDS row stride (f64) of the [m][64 rows] transposition tile: conflict-free ds_read_b64 */
#define TGP_MAX_MT 8
#ifndef TGP_RSPLIT
#define TGP_RSPLIT 4 /* the slab reduction sums every element in this many contiguous shares of the slabs (k_reduce) */
#endif
#define TGP_LOG_2PI_REF 1.8378770942368803 /* log(2*float32(pi)): the reference's cg.pi is a float32 tensor (dsp/config.py:71) */

// Diagnostic build only (-DTGP_STAMPS): thread 0 of block 0 writes the 100 MHz s_memrealtime counter into the
// workspace header at phase boundaries.  The shipped library never executes a stamp.
#ifdef TGP_STAMPS
#define TGP_STAMP(wsp, plan, i)                                                                       \
  do {                                                                                                \
    if (blockIdx.x == 0 && threadIdx.x == 0)                                                          \
      (wsp)[(plan).hdr + tgp::H_STAMP + (i)] = (double)__builtin_amdgcn_s_memrealtime();              \
  } while (0)
#else
#define TGP_STAMP(wsp, plan, i) \
  do {                          \
  } while (0)
#endif

namespace tgp {

// Write-through store (global_store ... sc1) for results the NEXT launch consumes.  A launch ends with the write-back of every
// dirty L2 line its workgroups left behind, and that flush sits on the chain between two dependent launches: the row kernel's
// 15.5 MB of slabs cost ~3 us after its last wave had finished (round 5: step 100.7 -> 95.9 us with the slabs stored this way
// -- the data leaves while other workgroups still compute, and the consumer finds it in the Infinity Cache).  Measured one
// group of stores at a time on one box: it pays for the slabs and (a little) for k_reduce's sums; the small outputs (L, J, H'^T,
// gradients, optimiser state, rowp) are neutral, the MLP's read-modify-written weight-gradient partials lose (ID_TGP 7 250 -> 6 950).
__device__ __forceinline__ void st_wt(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------
// problem plan: derived sizes + workspace offsets (in doubles).  Host and device agree through this.
// ---------------------------------------------------------------------------------------------------
struct Plan {
  int N, D, M, S, nblk, P, RP, lik;
  int MT, MP, DP, CT, CT16, ntri, nblocks;
  int nw4;     // 0: the 16-rows-per-wave row kernel (k_rows, 64 rows per block); else k_rows4 with nw4 waves per block
  int rw;      // data rows per wave of k_rows: 16, or 10 (tgp_rows.hpp, RW) -- meaningful when nw4 == 0
  int rpb;     // data rows per row block (= per slab): 4 * rw or 4 * nw4
  int nslots;  // store-mode flow stack slots
  int zs_lds;  // k_prep_a keeps Zs in LDS (set by the launcher from the LDS budget)
  size_t slab_G, slab_T, slab_S, slab_C, slab_len;  // offsets inside one slab / slab length
  // workspace offsets (doubles)
  size_t hdr, ils, ls, Zs, mpad, w, tp, tg;
  size_t Kmm, L, J, LT, Lq, LqT, S_, HpT, Q;
  size_t nD;     // MT x 256: minus the inverses of the diagonal tiles of L (row-major 16 x 16, zero above the diagonal)
  size_t Gp;     // G = sum over the row blocks' slabs, expanded to a full symmetric MP x MP matrix
  size_t redp;   // the sum of the slab tail (T, s, scalars); slab layout, G part unused
  size_t PP;     // MT x MP x PPW per-row-block partials of (Kbar_MM o K_MM) [Zs, 1]
  int PPW;
  size_t dbg;    // 256 doubles for the diagnostic (-DTGP_STAMPS) builds
  size_t slabs;  // nblocks * slab_len
  size_t total;  // doubles
};

// hdr slots
enum { H_S2 = 0, H_KL = 1, H_ETA = 2, H_EINV = 3, H_SIG_OS = 4, H_STEP = 5, H_STAMP = 8, H_PSTAMP = 32, H_OUT = 60, H_N = 64 };
// (H_OUT .. H_OUT+3: where tgp_qf_moments_bwd_f64 lets the backward chain put its four ELBO scalars -- it has no `out`)
// slab scalar slots
enum { C_ELL = 0, C_ETAB = 1, C_SVB = 2, C_PAD = 3, C_THETA = 4 };

inline size_t rup(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Waves per workgroup the 4-rows-per-wave row kernel (tgp_rows4.hpp) uses at N rows, 0 = not a candidate: the smallest
// workgroup (4 waves = 16 rows, then 8 waves = 32 rows) for which the row blocks AND the MT passenger blocks of the launch
// are at most one per CU -- one more and the passengers wait for a row block to retire (measured: 34.3 -> 44.1 us from 3 500 to
// 4 096 rows with 4-wave workgroups, 42.1 -> 54.1 us from 7 000 to 8 192 rows with 8-wave ones).  MEASURED (ROWS phase alone,
// M = 100, StepTanhL 3 x 2, profiles/r05_rows4_vs_rows16.txt): 32-34 us with 4-wave workgroups and 40-42 us with 8-wave ones
// against 52-53 us for the 16-row kernel; with 12-wave workgroups (three waves per SIMD: 168 registers, spills) it loses
// (67 us at the full Power batch), so beyond 32 (255 - MT) rows the launch stays on k_rows.
// (<= 255, not 256: with every CU needed, one CU that is late -- measured on one box of the pool, +11 us at 249 + 7
//  workgroups -- costs a second round; one spare CU costs 16 / 32 rows of range)
inline int rows4_waves(int N, int MT) {
  if ((N + 15) / 16 + MT <= 255) return 4;
  if ((N + 31) / 32 + MT <= 255) return 8;
  return 0;
}
// Data rows per wave k_rows uses at N rows when the launch qualifies (training, flow likelihood, shared flow parameters;
// tgp_rows.hip rows_per_wave): 10 where 4 x 10-row workgroups still number at most one per CU and 16-row waves would leave
// SIMDs idle (Power: 862 waves in 216 workgroups instead of 539 in 135), else 16.
inline int rows_rw(int N) { return (N > 4000 && (N + 39) / 40 <= 256) ? 10 : 16; }
// slabs a workspace must hold whichever row kernel runs
inline int plan_alloc_blocks(int N) {
  int nb = (N + TGP_ROWS_PER_BLOCK - 1) / TGP_ROWS_PER_BLOCK;
  const int nw = rows4_waves(N, 1);      // (MT = 1: the most blocks any model of this N can get)
  if (nw > 0 && (N + 4 * nw - 1) / (4 * nw) > nb) nb = (N + 4 * nw - 1) / (4 * nw);
  const int rw = rows_rw(N);
  if ((N + 4 * rw - 1) / (4 * rw) > nb) nb = (N + 4 * rw - 1) / (4 * rw);
  return nb < 1 ? 1 : nb;
}

inline int make_plan(Plan& p, int N, int D, int M, int S, int nblk, int P, int RP, int lik, int nw4 = 0, int rw = 16) {
  if (D < 1 || D > 16) return -2;
  if (M < 1 || M > 16 * TGP_MAX_MT) return TGP_E_UNSUPPORTED;
  p.N = N; p.D = D; p.M = M; p.S = S; p.nblk = nblk; p.P = P; p.RP = RP; p.lik = lik; p.nslots = 0; p.zs_lds = 0;
  p.MT = (M + 15) / 16; p.MP = p.MT * 16;
  p.DP = D <= 4 ? 4 : (D <= 8 ? 8 : 16);
  p.CT = (2 * p.DP + 1 + 15) / 16; p.CT16 = p.CT * 16;
  p.ntri = p.MT * (p.MT + 1) / 2;
  p.nw4 = nw4;
  p.rw = rw;
  p.rpb = nw4 > 0 ? 4 * nw4 : 4 * rw;
  p.nblocks = (N + p.rpb - 1) / p.rpb;
  if (p.nblocks < 1) p.nblocks = 1;
  p.slab_G = 0;
  p.slab_T = p.slab_G + (size_t)p.ntri * 256;
  p.slab_S = p.slab_T + (size_t)p.MP * p.CT16;
  p.slab_C = p.slab_S + p.MP;
  p.slab_len = rup(p.slab_C + C_THETA + P, 16);
  size_t o = 0;
  const size_t mm = (size_t)p.MP * p.MP;
  p.hdr = o; o += H_N;
  p.ils = o; o += 16;
  p.ls = o; o += 16;
  p.Zs = o; o += (size_t)p.MP * p.DP;
  p.mpad = o; o += p.MP;
  p.w = o; o += p.MP;
  p.tp = o; o += rup(P + 1, 16);
  p.tg = o; o += rup(P + 1, 16);
  p.Kmm = o; o += mm; p.L = o; o += mm; p.J = o; o += mm; p.LT = o; o += mm;
  p.Lq = o; o += mm; p.LqT = o; o += mm; p.S_ = o; o += mm; p.HpT = o; o += mm; p.Q = o; o += mm;
  p.nD = o; o += (size_t)p.MT * 256;
  p.Gp = o; o += mm;
  p.redp = o; o += p.slab_len;
  p.PPW = p.DP + 2;
  p.PP = o; o += (size_t)2 * p.MT * p.MP * p.PPW;   // two partials per row block (k_bwd's row role is split in two)
  p.dbg = o; o += 256;
  p.slabs = o; o += (size_t)p.nblocks * p.slab_len;
  p.total = o;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// scalar helpers (torch semantics)
// ---------------------------------------------------------------------------------------------------

// exp(x) with a SHORT dependency chain.  A dependent f64 FMA costs ~32 cycles on gfx950 (measured,
// tools/probes/mfma_rate.hip) and the row kernel runs one wave per SIMD, so the ~25-deep chain of the library exp is
// what the K tiles and the flow quadrature were waiting on.  Cody-Waite reduction x = k ln2 + r, |r| <= 0.347,
// degree-13 Taylor polynomial evaluated by Estrin's scheme (depth 4), scaled by v_ldexp_f64.
// Truncation r^14/14! < 5e-18; total error a few ulp.  Arguments are clamped to the finite range.
__device__ __forceinline__ double exp_fast(double x) {
  x = fmin(fmax(x, -745.0), 709.0);
  const double k = rint(x * 1.4426950408889634074);
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
  const double p01 = 1.0 + r;
  const double p23 = fma(r, 1.0 / 6.0, 0.5);
  const double p45 = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  const double p67 = fma(r, 1.0 / 5040.0, 1.0 / 720.0);
  const double p89 = fma(r, 1.0 / 362880.0, 1.0 / 40320.0);
  const double pab = fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0);
  const double pcd = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
  const double q0 = fma(p23, r2, p01), q1 = fma(p67, r2, p45), q2 = fma(pab, r2, p89);
  const double s0 = fma(q1, r4, q0), s1 = fma(pcd, r4, q2);
  return ldexp(fma(s1, r8, s0), (int)k);
}

// Covariance value from the squared scaled distance d2 = |(x - z)/l|^2, and the weight of -1/2 d(d2) in its
// derivative (k_g = -2 dk/d(d2)):   RBF      k = s2 exp(-d2/2)                       k_g = k
//                                   MATERN32 k = s2 (1 + a r) exp(-a r), a = sqrt 3    k_g = 3 s2 exp(-a r)
// with r = sqrt(max(d2, 1e-30)) as gpytorch's covar_dist clamps it (third-party formula, gpytorch 1.1.1 MaternKernel).
#define TGP_SQRT3 1.7320508075688772
__device__ __forceinline__ double cov_value(int kernel, double s2, double d2) {
  if (kernel == TGP_KERNEL_SCALE_MATERN32) {
    const double ar = TGP_SQRT3 * sqrt(fmax(d2, 1e-30));
    return s2 * (1.0 + ar) * exp_fast(-ar);
  }
  return s2 * exp_fast(-0.5 * d2);
}
__device__ __forceinline__ double cov_gweight(int kernel, double s2, double d2) {
  if (kernel == TGP_KERNEL_SCALE_MATERN32) return 3.0 * s2 * exp_fast(-TGP_SQRT3 * sqrt(fmax(d2, 1e-30)));
  return s2 * exp_fast(-0.5 * d2);
}

// 1/sqrt(d): v_rsq_f64 estimate + two Newton steps (same scheme as rsqrt_nr below, declared here for the flows)
__device__ __forceinline__ double rsqrt_nr_fwd(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}

// log(x) for finite normal x > 0 with a short instruction stream (the library log is ~70 f64 instructions of
// double-double arithmetic; the SAL blocks call it once per quadrature node).  x = m 2^e, m in [sqrt(1/2), sqrt 2):
// log m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716, odd series to s^19 (truncation 4e-18), Estrin in z = s^2.
__device__ __forceinline__ double log_fast(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double den = m + 1.0;
  double y = __builtin_amdgcn_rcp(den);
  y = fma(fma(-den, y, 1.0), y, y);
  y = fma(fma(-den, y, 1.0), y, y);
  const double num = m - 1.0;
  double s_ = num * y;
  s_ = fma(fma(-den, s_, num), y, s_);  // one correction step of the quotient
  const double z = s_ * s_, z2 = z * z, z4 = z2 * z2;
  const double p01 = fma(z, 2.0 / 5.0, 2.0 / 3.0);
  const double p23 = fma(z, 2.0 / 9.0, 2.0 / 7.0);
  const double p45 = fma(z, 2.0 / 13.0, 2.0 / 11.0);
  const double p67 = fma(z, 2.0 / 17.0, 2.0 / 15.0);
  const double p8 = 2.0 / 19.0;
  const double q0 = fma(p23, z2, p01), q1 = fma(p67, z2, p45);
  const double P = fma(fma(p8, z4, q1), z4, q0);
  const double lm = fma(s_ * z, P, s_ + s_);
  const double ed = (double)e;
  return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lm));
}

// 1/x from v_rcp_f64 + two Newton steps (no v_div_scale / v_div_fixup: operands here are finite, normal, non-zero)
__device__ __forceinline__ double rcp_fast(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}

// F.softplus (threshold 20) and sigmoid on the short-chain exp / log above.  The library log1p(exp(x)) is ~250
// dependent f64 instructions, and the lengthscale transform sits at the very start of k_prep_a's critical chain.
// log1p(e) = log(u) e / (u - 1), u = 1 + e (exact when u == 1 is handled apart): accurate for tiny e too.
__device__ __forceinline__ double softplus_d(double x) {
  if (x > 20.0) return x;
  const double e = exp_fast(x), u = 1.0 + e;
  if (u == 1.0) return e;
  return log_fast(u) * (e * rcp_fast(u - 1.0));
}
__device__ __forceinline__ double sigmoid_d(double x) { return 1.0 / (1.0 + exp_fast(-x)); }

// ---- N independent evaluations written STAGE BY STAGE.  The row kernel runs one wave per SIMD, so the only latency
// hiding the f64 pipe gets is independent instructions of the same wave; unrolled node-by-node the chains below were
// emitted back to back (hipcc keeps source order inside a block: ~9.5 cycles per instruction measured instead of 4).
// Same operations in the same order per element as the scalar functions: bitwise the same results. ----
#define TGP_EACH(u, N) _Pragma("unroll") for (int u = 0; u < (N); ++u)
template <int N>
__device__ __forceinline__ void exp_fast_n(double (&x)[N]) {
  double k[N], r[N], r2[N], r4[N], r8[N], p01[N], p23[N], p45[N], p67[N], p89[N], pab[N], pcd[N];
  TGP_EACH(u, N) x[u] = fmin(fmax(x[u], -745.0), 709.0);
  TGP_EACH(u, N) k[u] = rint(x[u] * 1.4426950408889634074);
  TGP_EACH(u, N) r[u] = fma(-k[u], 6.93147180369123816490e-01, x[u]);
  TGP_EACH(u, N) r[u] = fma(-k[u], 1.90821492927058770002e-10, r[u]);
  TGP_EACH(u, N) r2[u] = r[u] * r[u];
  TGP_EACH(u, N) p01[u] = 1.0 + r[u];
  TGP_EACH(u, N) p23[u] = fma(r[u], 1.0 / 6.0, 0.5);
  TGP_EACH(u, N) p45[u] = fma(r[u], 1.0 / 120.0, 1.0 / 24.0);
  TGP_EACH(u, N) p67[u] = fma(r[u], 1.0 / 5040.0, 1.0 / 720.0);
  TGP_EACH(u, N) p89[u] = fma(r[u], 1.0 / 362880.0, 1.0 / 40320.0);
  TGP_EACH(u, N) pab[u] = fma(r[u], 1.0 / 39916800.0, 1.0 / 3628800.0);
  TGP_EACH(u, N) pcd[u] = fma(r[u], 1.0 / 6227020800.0, 1.0 / 479001600.0);
  TGP_EACH(u, N) r4[u] = r2[u] * r2[u];
  TGP_EACH(u, N) p01[u] = fma(p23[u], r2[u], p01[u]);   // q0
  TGP_EACH(u, N) p45[u] = fma(p67[u], r2[u], p45[u]);   // q1
  TGP_EACH(u, N) p89[u] = fma(pab[u], r2[u], p89[u]);   // q2
  TGP_EACH(u, N) r8[u] = r4[u] * r4[u];
  TGP_EACH(u, N) p01[u] = fma(p45[u], r4[u], p01[u]);   // s0
  TGP_EACH(u, N) p89[u] = fma(pcd[u], r4[u], p89[u]);   // s1
  TGP_EACH(u, N) x[u] = ldexp(fma(p89[u], r8[u], p01[u]), (int)k[u]);
}
template <int N>
__device__ __forceinline__ void rcp_fast_n(double (&x)[N]) {
  double y[N];
  TGP_EACH(u, N) y[u] = __builtin_amdgcn_rcp(x[u]);
  TGP_EACH(u, N) y[u] = fma(fma(-x[u], y[u], 1.0), y[u], y[u]);
  TGP_EACH(u, N) x[u] = fma(fma(-x[u], y[u], 1.0), y[u], y[u]);
}
template <int N>
__device__ __forceinline__ void rsqrt_nr_fwd_n(const double (&d)[N], double (&y)[N]) {
  TGP_EACH(u, N) y[u] = __builtin_amdgcn_rsq(d[u]);
  TGP_EACH(u, N) y[u] = y[u] * fma(-(0.5 * d[u]) * y[u], y[u], 1.5);
  TGP_EACH(u, N) y[u] = y[u] * fma(-(0.5 * d[u]) * y[u], y[u], 1.5);
}
template <int N>
__device__ __forceinline__ void log_fast_n(double (&x)[N]) {
  double m[N], den[N], y[N], num[N], s_[N], z[N], z2[N], z4[N], p01[N], p23[N], p45[N], p67[N], ed[N];
  TGP_EACH(u, N) {
    m[u] = __builtin_amdgcn_frexp_mant(x[u]);
    int e = __builtin_amdgcn_frexp_exp(x[u]);
    const bool lo = m[u] < 0.70710678118654752440;
    m[u] = lo ? m[u] + m[u] : m[u];
    e = lo ? e - 1 : e;
    ed[u] = (double)e;
  }
  TGP_EACH(u, N) den[u] = m[u] + 1.0;
  TGP_EACH(u, N) num[u] = m[u] - 1.0;
  TGP_EACH(u, N) y[u] = __builtin_amdgcn_rcp(den[u]);
  TGP_EACH(u, N) y[u] = fma(fma(-den[u], y[u], 1.0), y[u], y[u]);
  TGP_EACH(u, N) y[u] = fma(fma(-den[u], y[u], 1.0), y[u], y[u]);
  TGP_EACH(u, N) s_[u] = num[u] * y[u];
  TGP_EACH(u, N) s_[u] = fma(fma(-den[u], s_[u], num[u]), y[u], s_[u]);
  TGP_EACH(u, N) z[u] = s_[u] * s_[u];
  TGP_EACH(u, N) z2[u] = z[u] * z[u];
  TGP_EACH(u, N) p01[u] = fma(z[u], 2.0 / 5.0, 2.0 / 3.0);
  TGP_EACH(u, N) p23[u] = fma(z[u], 2.0 / 9.0, 2.0 / 7.0);
  TGP_EACH(u, N) p45[u] = fma(z[u], 2.0 / 13.0, 2.0 / 11.0);
  TGP_EACH(u, N) p67[u] = fma(z[u], 2.0 / 17.0, 2.0 / 15.0);
  TGP_EACH(u, N) z4[u] = z2[u] * z2[u];
  TGP_EACH(u, N) p01[u] = fma(p23[u], z2[u], p01[u]);  // q0
  TGP_EACH(u, N) p45[u] = fma(p67[u], z2[u], p45[u]);  // q1
  TGP_EACH(u, N) p45[u] = fma(2.0 / 19.0, z4[u], p45[u]);
  TGP_EACH(u, N) p01[u] = fma(p45[u], z4[u], p01[u]);  // P
  TGP_EACH(u, N) p01[u] = fma(s_[u] * z[u], p01[u], s_[u] + s_[u]);  // log m
  TGP_EACH(u, N) x[u] = fma(ed[u], 6.93147180369123816490e-01, fma(ed[u], 1.90821492927058770002e-10, p01[u]));
}

// sum over the four 16-lane groups (lanes l, l^16, l^32, l^48): every lane gets the total
// (gfx950 v_permlane16_swap / v_permlane32_swap: VALU lane exchanges, no trip through the LDS crossbar.  With both
//  operands = x the pair returned is {x of the even rows / low half, x of the odd rows / high half} in every lane.)
__device__ __forceinline__ double xor_sum16(double x) {
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double xor_sum32(double x) {
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double quad_sum(double x) { return xor_sum32(xor_sum16(x)); }
// x[l] + x[(l + R) % 16 within its row of 16 lanes] on DPP (row_ror:R); equals x[l] + x[l ^ R] once x is 2R-periodic
template <int R>
__device__ __forceinline__ double ror_sum(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const int l2 = __builtin_amdgcn_update_dpp(0, lo, 0x120 + R, 0xf, 0xf, false);
  const int h2 = __builtin_amdgcn_update_dpp(0, hi, 0x120 + R, 0xf, 0xf, false);
  return x + __hiloint2double(h2, l2);
}
// butterfly sum over the 64 lanes in the order xor 32, 16, 8, 4, 2, 1 (every lane gets the total), without LDS traffic
__device__ __forceinline__ double wave_sum(double x) {
  x = xor_sum16(xor_sum32(x));
  return ror_sum<1>(ror_sum<2>(ror_sum<4>(ror_sum<8>(x))));
}

// ---------------------------------------------------------------------------------------------------
// one-wave 16x16 Cholesky + triangular inverse (the serial core of both blocked factorisations)
// ---------------------------------------------------------------------------------------------------
// Broadcast of lane `src` OF EACH ROW OF 16 LANES to that row (src a constant after unrolling): one v_mov_b64_dpp
// row_newbcast instead of two v_readlane_b32 -- the readlanes were half the instructions of the tile factorisation,
// which is bound by instruction issue, not by its dependency chain (tools/probes/potrf_rate.hip).  The callers give
// all four rows of the wave the same data (lane & 15), so a per-row broadcast is a wave-wide one.
// (measured, one 16x16 tile: 7059 cycles with readlanes, 4569 with this; folding the broadcast into v_fmac_f64_dpp by
//  hand needs an s_nop per instruction for the DPP read hazard and came out slower, 5435)
template <int K>
__device__ __forceinline__ double bcast_row_k(double x) {
  return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + K, 0xf, 0xf, true);
}
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{}) -- the DPP control
// word is an immediate, so the lane index has to be a constant expression (a `#pragma unroll` index is not one)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// 1/sqrt(d) from the hardware estimate v_rsq_f64 plus two Newton steps (y <- y (1.5 - 0.5 d y^2)): the diagonal
// tile factorisation is ONE serial dependency chain, so the ~40-instruction 1.0/sqrt(d) sequence was its main cost.
__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}

// Cholesky of one 16x16 diagonal tile + its inverse, executed by ONE wave; lane i (< 16) owns row i.
//   in : a[c] = A[i][c] (lower part valid)
//   out: a[c] = L[i][c] ; x[r] = (L^-1)[r][lane] (column `lane` of the inverse) ; returns first bad pivot (0 = ok)
__device__ __forceinline__ int potrf_trtri16(double (&a)[16], double (&x)[16], int lane) {
  int bad = 0;
  // Row k of X = L^-1 needs only L[k][0..k-1] (final after pivot k-1), the rows of X above it and 1/L_kk, so it is
  // formed inside step k: its dot product is independent of the pivot's rsqrt/Newton chain and of the rank-1 update,
  // and the fully unrolled code lets the scheduler interleave the two dependency chains (the factorisation alone
  // sets the latency; the inverse rides in its shadow).
  static_for<16>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const double d = bcast_row_k<k>(a[k]);
    if (!(d > 0.0) && bad == 0) bad = k + 1;  // the same in every lane; also catches NaN
    const double rinv = rsqrt_nr(d);
    // x_k = -(sum_{m<k} L_km x_m) / L_kk for column `lane` (two partial sums halve the dependent-add chain)
    double s0 = 0.0, s1 = 0.0;
    static_for<16>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (m < k) {
        if constexpr ((m & 1) == 0) s0 = fma(bcast_row_k<k>(a[m]), x[m], s0);
        else s1 = fma(bcast_row_k<k>(a[m]), x[m], s1);
      }
    });
    x[k] = (lane == k) ? rinv : (lane < k ? -(s0 + s1) * rinv : 0.0);
    a[k] = (lane == k) ? d * rinv : a[k] * rinv;  // column k of L (rows >= k meaningful)
    static_for<16>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j > k) {
        a[j] = fma(-a[k], bcast_row_k<j>(a[k]), a[j]);  // a_ij -= L_ik L_jk (rows i >= j)
        // the update is DONE here: without the pin hipcc sinks it to the step that consumes a[j] and keeps the
        // broadcast alive until then -- 15 live doubles per pivot, 246 VGPRs, spills in the callers
        asm volatile("" : "+v"(a[j]));
      }
    });
  });
  return bad;
}

// Cholesky of a 16-COLUMN PANEL by one wave, registers only (round 3: the critical chain of both blocked factorisations).
//   dg : the diagonal tile, replicated in each of the wave's four 16-lane rows: lane (l & 15) owns row l & 15
//        in : dg[c] = A_jj[l & 15][c] (lower part valid)     out: dg[c] = L_jj[l & 15][c]
//   a  : one row of the panel BELOW the diagonal tile per lane (64 rows per wave; PANEL = false: none)
//        in : a[c] = A[row_l][c], already updated by the block columns to the left
//        out: a[c] = L[row_l][c] = (A_panel L_jj^-T)[row_l][c]
// The rank-1 update of pivot k is applied to the panel rows in the same instructions' shadow (the multiplier L[j][k] is
// the broadcast the diagonal tile needs anyway), so a whole block column of L comes out of ONE pass: no inverse of the
// diagonal tile and no triangular solve on the chain (the left-looking schedule of rounds 1-2 paid potrf + trtri
// interleaved in one wave -- 800 instructions, issue-bound -- then a panel product and a diagonal update per block column).
// Any wave can run this on its own 64 panel rows: it factorises its own copy of the diagonal tile redundantly and needs
// nothing from the other waves.  Returns the first bad pivot (1-based; 0 = ok), the same in every lane.
// RDIAG: *rd = the pass's own 1 / L_jj[lane][lane] (lane = l & 15), for a later trtri16<.., true> that then reproduces the
// inverse the TRTRI form computes in the pass, bit for bit.
template <bool PANEL, bool TRTRI = false, bool RDIAG = false>
__device__ __forceinline__ int potrf_panel16(double (&dg)[16], double (&a)[16], double* x = nullptr, int lane = 0,
                                             double* rd = nullptr) {
  int bad = 0;
  static_for<16>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const double d = bcast_row_k<k>(dg[k]);
    if (!(d > 0.0) && bad == 0) bad = k + 1;  // the same in every lane; also catches NaN
    const double rinv = rsqrt_nr(d);
    if constexpr (RDIAG) {
      if (lane == k) *rd = rinv;
    }
    if constexpr (TRTRI) {
      // row k of X = L_jj^-1 rides in the pass's shadow as in potrf_trtri16: x[r] = X[r][lane & 15]
      double s0 = 0.0, s1 = 0.0;
      static_for<16>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m < k) {
          if constexpr ((m & 1) == 0) s0 = fma(bcast_row_k<k>(dg[m]), x[m], s0);
          else s1 = fma(bcast_row_k<k>(dg[m]), x[m], s1);
        }
      });
      x[k] = (lane == k) ? rinv : (lane < k ? -(s0 + s1) * rinv : 0.0);
    }
    dg[k] *= rinv;  // column k of L_jj (lane k: d / sqrt(d))
    if constexpr (PANEL) a[k] *= rinv;
    static_for<16>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (j > k) {
        const double b = bcast_row_k<j>(dg[k]);  // L_jj[j][k]
        dg[j] = fma(-dg[k], b, dg[j]);
        if constexpr (PANEL) a[j] = fma(-a[k], b, a[j]);
        asm volatile("" : "+v"(dg[j]));  // done here (see potrf_trtri16)
        if constexpr (PANEL) asm volatile("" : "+v"(a[j]));
      }
    });
  });
  return bad;
}

// Inverse of a 16x16 lower-triangular tile by one wave: in dg[c] = L[l & 15][c]; out x[r] = (L^-1)[r][l & 15] (column
// `lane` of the inverse, zero above the diagonal).  Off the factorisation's chain (a helper wave runs it one block
// column behind).
// GATED = false (a caller with registers to spare: the fused launch's chain blocks, 512 per wave): the 120 broadcasts
// and the 16 reciprocals are left free to be issued ahead of the substitution chain.
// RD = true: rd = 1 / L[lane][lane] as the factorisation pass formed it (potrf_panel16<.., .., true>) instead of a
// reciprocal of its own.
template <bool GATED = true, bool RD = false>
__device__ __forceinline__ void trtri16(const double (&dg)[16], double (&x)[16], int lane, double rd = 0.0) {
  double gate = 0.0;  // x[k-1]: the broadcasts of step k are tied behind it (below)
  double rinvs[16];
  if constexpr (!GATED && !RD) {
    static_for<16>([&](auto kc) { constexpr int k = decltype(kc)::value; rinvs[k] = bcast_row_k<k>(dg[k]); });
    rcp_fast_n<16>(rinvs);
  }
  static_for<16>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    double rinv;
    if constexpr (RD) rinv = bcast_row_k<k>(rd);
    else if constexpr (GATED) rinv = rcp_fast(bcast_row_k<k>(dg[k]));
    else rinv = rinvs[k];
    double s0 = 0.0, s1 = 0.0;
    static_for<16>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (m < k) {
        // None of the 120 broadcasts depends on x: left alone hipcc issues them all up front and keeps them alive
        // (240 VGPRs: the callers spilled).  The empty asm makes step k's operands wait for x[k-1].
        double t = dg[m];
        if constexpr (GATED) asm volatile("" : "+v"(t) : "v"(gate));
        if constexpr ((m & 1) == 0) s0 = fma(bcast_row_k<k>(t), x[m], s0);
        else s1 = fma(bcast_row_k<k>(t), x[m], s1);
      }
    });
    x[k] = (lane == k) ? rinv : (lane < k ? -(s0 + s1) * rinv : 0.0);
    gate = x[k];
  });
}

// ---------------------------------------------------------------------------------------------------
// 16x16 output tile of opA(A) * opB(B) over k in [k0,k1) (multiples of 4), operands in global memory,
// leading dimension ld.  TA: opA = A^T, TB: opB = B^T.  One wave.
// ---------------------------------------------------------------------------------------------------
template <bool TA, bool TB>
__device__ __forceinline__ d4 tile_mm(const double* __restrict__ A, const double* __restrict__ B, int ld, int i0,
                                      int j0, int k0, int k1, d4 acc) {
  const int l = threadIdx.x & 63, r = l & 15, q = l >> 4;
  for (int k = k0; k < k1; k += 4) {
    const double a = TA ? A[(size_t)(k + q) * ld + i0 + r] : A[(size_t)(i0 + r) * ld + k + q];
    const double b = TB ? B[(size_t)(j0 + r) * ld + k + q] : B[(size_t)(k + q) * ld + j0 + r];
    acc = TGP_MFMA(a, b, acc);
  }
  return acc;
}

// Same product with caller-supplied operand fetchers fa(k), fb(k) (k = first row of the 4-deep k-step; the
// fetcher adds the lane's own q).  Loads of 8 k-steps are issued before their MFMAs so the (L2/LDS) latency
// of one batch overlaps the matrix pipe instead of serialising load -> mfma -> load.
template <int BATCH = 8, class FA, class FB>
__device__ __forceinline__ d4 tile_mm_f(FA fa, FB fb, int k0, int k1, d4 acc) {
  int k = k0;
  for (; k + 4 * BATCH <= k1; k += 4 * BATCH) {
    double a[BATCH], b[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) { a[u] = fa(k + 4 * u); b[u] = fb(k + 4 * u); }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) acc = TGP_MFMA(a[u], b[u], acc);
    // pin the batch shape: every load (LDS or global) of the batch before its MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100 | 0x020, 3 * BATCH, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, BATCH, 0);
  }
  for (; k + 16 <= k1; k += 16) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = fa(k + 4 * u); b[u] = fb(k + 4 * u); }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = TGP_MFMA(a[u], b[u], acc);
  }
  for (; k < k1; k += 4) acc = TGP_MFMA(fa(k), fb(k), acc);
  return acc;
}

// Global-memory operands (L2 round trip ~1 us): one big batch so a whole tile costs one or two round trips.
#define TGP_GBATCH 28

// ---------------------------------------------------------------------------------------------------
// flows (models/flow.py).  `tp` = shared parameters after their positivity transform, `tg` = d(tp)/d(raw)
// (both prepared once per step by k_prep_a); per-row parameters are transformed on the fly.
// ---------------------------------------------------------------------------------------------------
#define TGP_MAX_BLOCKS 64
// the flow program travels to the kernels BY VALUE (kernel argument): the ABI takes it as a small host array
struct FlowProg {
  int32_t nblk;
  int32_t nslots;  // store-mode stack slots (flow_slots)
  int32_t blk[4 * TGP_MAX_BLOCKS];
};

struct FlowDev {
  const int32_t* prog;  // nblk x 4
  int nblk;
  const double* tp;
  const double* tg;
  const double* ti = nullptr;  // optional: 1 / tp[i] (rcp_fast) for every i, prepared once by the caller
};
// Block descriptor b of an LDS-resident program (16-byte aligned, 4 x int32 per block) in one read.  The store-mode
// sweeps request descriptor b+1 (b-1) BEFORE working on block b: with one wave per SIMD an un-prefetched descriptor
// costs a full LDS round trip per block.
struct FlowBlk { int kind, K, poff, flags; };
__device__ __forceinline__ FlowBlk flow_blk(const int32_t* prog, int b) {
  const int4 v = *reinterpret_cast<const int4*>(prog + 4 * b);
  return FlowBlk{v.x, v.y, v.z, v.w};
}
// 1 / tp[i]: from the caller's table when there is one (the row kernel: one reciprocal chain less per tanh step)
__device__ __forceinline__ double flow_rcp_param(const FlowDev& F, int i) {
  return F.ti != nullptr ? F.ti[i] : rcp_fast(F.tp[i]);
}
// accumulator update in LDS without the read round trip of `*p += v` (ds_add_f64; one wave per SIMD has nothing to
// hide that latency with).  Each address is only ever touched by one lane: the sum order stays fixed.
__device__ __forceinline__ void lds_acc(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// asinh exactly as the reference writes it (flow.py:904-905)
__device__ __forceinline__ double asinh_ref(double f) { return log(f + sqrt(f * f + 1.0)); }

// Parameter `poff` of a block: the shared value in LDS, or this row's value in global memory (`pr`).  Written as two
// loads in two address spaces on purpose: `pr ? rp[poff] : F.tp[poff]` became a select of the two POINTERS followed by one
// FLAT load -- LDS parameters fetched through the flat path, counted against vmcnt and lgkmcnt at once.
__device__ __forceinline__ double flow_param(const FlowDev& F, const double* rp, int poff, bool pr) {
  double a = F.tp[poff];
  if (pr) a = *(const double __attribute__((address_space(1)))*)(rp + poff);
  return a;
}

// Forward through all blocks (evaluation / prediction).  If dG != nullptr it receives dG/df.
__device__ inline double flow_forward(const FlowDev& F, double f, const double* __restrict__ rp, double* stack,
                                      int sstride, double* dG) {
  double der = 1.0;
  for (int b = 0; b < F.nblk; ++b) {
    const int kind = F.prog[4 * b], K = F.prog[4 * b + 1], poff = F.prog[4 * b + 2], flags = F.prog[4 * b + 3];
    const bool pr = flags & TGP_FLAG_PER_ROW;
    if (stack) stack[b * sstride] = f;
    if (kind == TGP_FLOW_AFFINE) {
      double a = flow_param(F, rp, poff, pr);
      if (pr && (flags & TGP_FLAG_RESTRICT)) a = softplus_d(a);
      const double bb = flow_param(F, rp, poff + 1, pr);
      f = a * f + bb;
      der *= a;
    } else if (kind == TGP_FLOW_SAL) {
      const double a = flow_param(F, rp, poff, pr);
      double bb = flow_param(F, rp, poff + 1, pr);
      if (pr && (flags & TGP_FLAG_RESTRICT)) bb = softplus_d(bb);
      const double t = bb * asinh_ref(f) - a;
      double g = sinh(t);
      double gp = bb * cosh(t) / sqrt(1.0 + f * f);
      if (flags & TGP_FLAG_ADD_F0) { g += f; gp += 1.0; }
      f = g;
      der *= gp;
    } else {  // STEPTANH (shared parameters only; the reference raises NotImplementedError for per-row step flows)
      const bool addf = flags & TGP_FLAG_ADD_F0;
      double g = addf ? f : 0.0, gp = addf ? 1.0 : 0.0;
      for (int k = 0; k < K; ++k) {
        const double a = F.tp[poff + 4 * k], bt = F.tp[poff + 4 * k + 1], c = F.tp[poff + 4 * k + 2],
                     dt = F.tp[poff + 4 * k + 3];
        const double th = tanh((f - c) / dt);
        g += a + bt * th;
        gp += bt * (1.0 - th * th) / dt;
      }
      f = g;
      der *= gp;
    }
  }
  if (dG) *dG = der;
  return f;
}

// Forward through all blocks for NB independent elements at once (evaluation / prediction kernels): value and, when DER,
// dG/df, on the short-chain exp / log / rcp above, written stage by stage over the NB elements (the scalar flow_forward
// with the library tanh / sinh / cosh / division is ~4x the instructions).  rp[u] = per-row parameters of element u.
// Unlike the training sweeps (exp clamped to the finite range), an exponent past the float64 range gives +inf here: the
// reference's evaluation pushes every node through the naive sinh(b asinh f - a) and reports the inf.
template <int NB, bool DER>
__device__ inline void flow_forward_n(const FlowDev& F, double (&f)[NB], const double* const (&rp)[NB], double (&der)[NB]) {
  TGP_EACH(u, NB) der[u] = 1.0;
  for (int b = 0; b < F.nblk; ++b) {
    const int kind = F.prog[4 * b], K = F.prog[4 * b + 1], poff = F.prog[4 * b + 2], flags = F.prog[4 * b + 3];
    const bool pr = flags & TGP_FLAG_PER_ROW;
    if (kind == TGP_FLOW_AFFINE) {
      TGP_EACH(u, NB) {
        double a = flow_param(F, rp[u], poff, pr);
        if (pr && (flags & TGP_FLAG_RESTRICT)) a = softplus_d(a);
        const double bb = flow_param(F, rp[u], poff + 1, pr);
        f[u] = a * f[u] + bb;
        if (DER) der[u] *= a;
      }
    } else if (kind == TGP_FLOW_SAL) {
      const bool addf = flags & TGP_FLAG_ADD_F0;
      double a[NB], bb[NB], q1[NB], isf[NB], sf[NB], uu[NB], e[NB], ei[NB], t[NB];
      TGP_EACH(u, NB) {
        a[u] = flow_param(F, rp[u], poff, pr);
        bb[u] = flow_param(F, rp[u], poff + 1, pr);
        if (pr && (flags & TGP_FLAG_RESTRICT)) bb[u] = softplus_d(bb[u]);
      }
      TGP_EACH(u, NB) q1[u] = f[u] * f[u] + 1.0;
      rsqrt_nr_fwd_n<NB>(q1, isf);
      TGP_EACH(u, NB) sf[u] = q1[u] * isf[u];
      TGP_EACH(u, NB) sf[u] = fma(fma(-sf[u], sf[u], q1[u]), 0.5 * isf[u], sf[u]);
      TGP_EACH(u, NB) uu[u] = f[u] + sf[u];
      log_fast_n<NB>(uu);                                   // asinh as the reference writes it (flow.py:904-905)
      TGP_EACH(u, NB) t[u] = bb[u] * uu[u] - a[u];
      TGP_EACH(u, NB) e[u] = t[u];
      exp_fast_n<NB>(e);
      TGP_EACH(u, NB) e[u] = t[u] > 709.78 ? INFINITY : e[u];
      TGP_EACH(u, NB) ei[u] = fmax(e[u], 1e-320);
      rcp_fast_n<NB>(ei);
      TGP_EACH(u, NB) ei[u] = t[u] > 709.78 ? 0.0 : (t[u] < -709.78 ? INFINITY : ei[u]);
      TGP_EACH(u, NB) {
        double g = 0.5 * (e[u] - ei[u]), gp = bb[u] * (0.5 * (e[u] + ei[u])) * isf[u];
        if (addf) { g += f[u]; gp += 1.0; }
        f[u] = g;
        if (DER) der[u] *= gp;
      }
    } else {  // STEPTANH (shared parameters only)
      const bool addf = flags & TGP_FLAG_ADD_F0;
      double g[NB], gp[NB];
      TGP_EACH(u, NB) { g[u] = addf ? f[u] : 0.0; gp[u] = addf ? 1.0 : 0.0; }
      for (int k = 0; k < K; ++k) {
        const double a = F.tp[poff + 4 * k], bt = F.tp[poff + 4 * k + 1], c = F.tp[poff + 4 * k + 2],
                     idt = flow_rcp_param(F, poff + 4 * k + 3);
        double e[NB];
        TGP_EACH(u, NB) e[u] = 2.0 * (f[u] - c) * idt;
        exp_fast_n<NB>(e);
        TGP_EACH(u, NB) e[u] += 1.0;
        rcp_fast_n<NB>(e);
        TGP_EACH(u, NB) {
          const double th = 1.0 - 2.0 * e[u];
          g[u] += a + bt * th;
          if (DER) gp[u] += bt * (1.0 - th * th) * idt;
        }
      }
      TGP_EACH(u, NB) {
        f[u] = g[u];
        if (DER) der[u] *= gp[u];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// "store" evaluation: NB quadrature nodes in flight per lane (independent dependency chains for the
// single-wave-per-SIMD row kernel), forward keeps what the reverse sweep needs so that the reverse sweep
// contains no transcendental.  Stack slots per block: AFFINE 1 {f_in}; SAL 3 {u, cosh t, g'}; STEPTANH 1+K
// {f_in, tanh_k}.  Slot s of node u lives at stack[(s*NB + u) * sstride].
// sinh/cosh/tanh are formed from ONE exp (+1 division): absolute error ~1e-16 * max(1, cosh), which is what the
// residual y - G(f) needs; asinh keeps the reference's log form.
// ---------------------------------------------------------------------------------------------------
__host__ __device__ inline int flow_slots(const int32_t* prog, int nblk) {
  int s = 0;
  for (int b = 0; b < nblk; ++b) s += prog[4 * b] == TGP_FLOW_AFFINE ? 1 : (prog[4 * b] == TGP_FLOW_SAL ? 3 : 1 + prog[4 * b + 1]);
  return s;
}

// PN (k_rows<.., RW < 16>: a lane's NB nodes belong to DIFFERENT data rows): the per-row parameters of node u come from
// rpn[u], an LDS table of the row's TRANSFORMED parameters the caller staged -- entry 2 c = value of column c (softplus
// already applied where the block restricts it), entry 2 c + 1 = d(value)/d(raw) -- so the sweeps neither wait for global
// memory nor evaluate a softplus per node.  Supported for SAL blocks (the per-row flows of this package); `rp` is unused.
template <int NB, bool PN = false>
__device__ inline void flow_forward_store(const FlowDev& F, double (&f)[NB], const double* __restrict__ rp,
                                          double* stack, int sstride, const double* const* rpn = nullptr) {
  int sl = 0;
  FlowBlk nx = flow_blk(F.prog, 0);
  for (int b = 0; b < F.nblk; ++b) {
    const int kind = nx.kind, K = nx.K, poff = nx.poff, flags = nx.flags;
    nx = flow_blk(F.prog, b + 1 < F.nblk ? b + 1 : b);
    const bool pr = flags & TGP_FLAG_PER_ROW;
    if (kind == TGP_FLOW_AFFINE) {
      double a = flow_param(F, rp, poff, pr);
      if (pr && (flags & TGP_FLAG_RESTRICT)) a = softplus_d(a);
      const double bb = flow_param(F, rp, poff + 1, pr);
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        stack[(sl * NB + u) * sstride] = f[u];
        f[u] = a * f[u] + bb;
      }
      sl += 1;
    } else if (kind == TGP_FLOW_SAL) {
      double av[NB], bv[NB];
      if constexpr (PN) {
        TGP_EACH(u, NB) {
          av[u] = pr ? rpn[u][2 * poff] : F.tp[poff];
          bv[u] = pr ? rpn[u][2 * (poff + 1)] : F.tp[poff + 1];
        }
      } else {
        const double a = flow_param(F, rp, poff, pr);
        double bb = flow_param(F, rp, poff + 1, pr);
        if (pr && (flags & TGP_FLAG_RESTRICT)) bb = softplus_d(bb);
        TGP_EACH(u, NB) { av[u] = a; bv[u] = bb; }
      }
      const bool addf = flags & TGP_FLAG_ADD_F0;
      // sqrt(f^2+1) and its reciprocal from one v_rsq_f64 + Newton (1/sf is needed anyway); asinh keeps the
      // reference's log(f + sqrt(f^2+1)) form (flow.py:904-905) on the short-chain log.  Stage by stage over the nodes.
      double q1[NB], isf[NB], sf[NB], uu[NB], e[NB], ei[NB];
      TGP_EACH(u, NB) q1[u] = f[u] * f[u] + 1.0;
      rsqrt_nr_fwd_n<NB>(q1, isf);
      TGP_EACH(u, NB) sf[u] = q1[u] * isf[u];
      TGP_EACH(u, NB) sf[u] = fma(fma(-sf[u], sf[u], q1[u]), 0.5 * isf[u], sf[u]);
      TGP_EACH(u, NB) uu[u] = f[u] + sf[u];
      log_fast_n<NB>(uu);
      TGP_EACH(u, NB) e[u] = bv[u] * uu[u] - av[u];
      exp_fast_n<NB>(e);
      TGP_EACH(u, NB) ei[u] = e[u];
      rcp_fast_n<NB>(ei);
      TGP_EACH(u, NB) {
        const double ch = 0.5 * (e[u] + ei[u]);
        double g = 0.5 * (e[u] - ei[u]), gp = bv[u] * ch * isf[u];
        if (addf) { g += f[u]; gp += 1.0; }
        stack[((sl + 0) * NB + u) * sstride] = uu[u];
        stack[((sl + 1) * NB + u) * sstride] = ch;
        stack[((sl + 2) * NB + u) * sstride] = gp;
        f[u] = g;
      }
      sl += 3;
    } else {
      const bool addf = flags & TGP_FLAG_ADD_F0;
      double g[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        stack[(sl * NB + u) * sstride] = f[u];
        g[u] = addf ? f[u] : 0.0;
      }
      for (int k = 0; k < K; ++k) {
        const double a = F.tp[poff + 4 * k], bt = F.tp[poff + 4 * k + 1], c = F.tp[poff + 4 * k + 2],
                     idt = flow_rcp_param(F, poff + 4 * k + 3);
        double e[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) e[u] = 2.0 * (f[u] - c) * idt;
        exp_fast_n<NB>(e);
        TGP_EACH(u, NB) e[u] += 1.0;
        rcp_fast_n<NB>(e);
        TGP_EACH(u, NB) {
          const double th = 1.0 - 2.0 * e[u];
          stack[((sl + 1 + k) * NB + u) * sstride] = th;
          g[u] += a + bt * th;
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) f[u] = g[u];
      sl += 1 + K;
    }
  }
}

// Reverse sweep for NB nodes: c[u] = d(objective)/dG on entry, d(objective)/df0 on exit.  Shared-parameter
// partials are summed over the NB nodes, then over the four lanes that share a data row (quad_sum), and lanes
// with q == 0 accumulate them into accq[slot * qstride]; per-row parameter partials go to the lane-private
// accr[(poff + j) * rstride].  All lanes of the wave must call this together (cross-lane sums inside).
// value of lane l ^ 32 / l ^ 16 (gfx950 permlane swaps, see xor_sum32 / xor_sum16)
__device__ __forceinline__ double lane_xor32(double x, bool hi) {
  const unsigned lo = __double2loint(x), h = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(h, h, false, false);
  return hi ? __hiloint2double(b[0], a[0]) : __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double lane_xor16(double x, bool odd) {
  const unsigned lo = __double2loint(x), h = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(h, h, false, false);
  return odd ? __hiloint2double(b[0], a[0]) : __hiloint2double(b[1], a[1]);
}
// Wave sums of FOUR values for the price of about one and a half: a reduce-scatter over the two top lane bits (after the
// xor-32 step a lane carries two of the four partial sums, after the xor-16 step one), then the four rotations inside
// the row of 16 lanes.  Lane 16 q (q = 0..3) ends with the wave's total of value q; fixed order, bit-reproducible.
__device__ __forceinline__ double wave_sum4(double p0, double p1, double p2, double p3, int lane) {
  const bool b5 = lane & 32, b4 = lane & 16;
  const double ka = b5 ? p2 : p0, sa = b5 ? p0 : p2, kb = b5 ? p3 : p1, sb = b5 ? p1 : p3;
  const double ra = ka + lane_xor32(sa, b5), rb = kb + lane_xor32(sb, b5);
  const double k = b4 ? rb : ra, sd = b4 ? ra : rb;
  const double v = k + lane_xor16(sd, b4);
  return ror_sum<1>(ror_sum<2>(ror_sum<4>(ror_sum<8>(v))));
}
// ... of TWO values: lane 0 ends with the total of p0, lane 32 with the total of p1
__device__ __forceinline__ double wave_sum2(double p0, double p1, int lane) {
  const bool b5 = lane & 32;
  const double k = b5 ? p1 : p0, sd = b5 ? p0 : p1;
  double v = k + lane_xor32(sd, b5);
  v = xor_sum16(v);
  return ror_sum<1>(ror_sum<2>(ror_sum<4>(ror_sum<8>(v))));
}

// RED: how far the shared-parameter partials are summed before the leader lanes add them to `accq`: 0 = over the four
// lanes l, l^16, l^32, l^48 (k_rows: they share a data row; leaders = lanes 0..15), 2 = over the whole wave (k_rows4:
// leader = lane 0; one accumulator column per wave instead of sixteen -- its LDS is the flow stack's).
template <int RED>
__device__ __forceinline__ double flow_red(double x) {
  if constexpr (RED == 2) return wave_sum(x);
  else return quad_sum(x);
}
// PN (see flow_forward_store): the per-row partials of a SAL block are PER NODE -- each node of a lane belongs to another data
// row -- and go to the block's own, now dead, stack slots (slot + 0: d/da, slot + 1: d/db of node u) through `gst`, the
// writable alias of `stack`; the caller gathers them per row afterwards.  accr is unused then.
template <int NB, int RED = 0, bool PN = false>
__device__ inline void flow_backward_store(const FlowDev& F, double (&c)[NB], const double* __restrict__ rp,
                                           const double* stack, int sstride, int nslots, double* accq, int qstride,
                                           bool qlead, double* accr, int rstride, const double* const* rpn = nullptr,
                                           double* gst = nullptr) {
  // (accq / accr live in LDS: lds_acc.  Every block reads all its LDS operands -- parameters AND stack slots -- before
  //  the first use: one exposed LDS round trip per block step instead of three or four.)
  int sl = nslots;
  FlowBlk nx = flow_blk(F.prog, F.nblk - 1);
  for (int b = F.nblk - 1; b >= 0; --b) {
    const int kind = nx.kind, K = nx.K, poff = nx.poff, flags = nx.flags;
    nx = flow_blk(F.prog, b > 0 ? b - 1 : 0);
    const bool pr = flags & TGP_FLAG_PER_ROW;
    if (kind == TGP_FLOW_AFFINE) {
      sl -= 1;
      double a, fa, fin[NB];
      TGP_EACH(u, NB) fin[u] = stack[(sl * NB + u) * sstride];
      if (pr) {
        a = rp[poff]; fa = 1.0;
        if (flags & TGP_FLAG_RESTRICT) { fa = sigmoid_d(a); a = softplus_d(a); }
      } else {
        a = F.tp[poff]; fa = F.tg[poff];
      }
      double pa = 0.0, pb = 0.0;
      TGP_EACH(u, NB) {
        pa += c[u] * fin[u];
        pb += c[u];
        c[u] *= a;
      }
      pa *= fa;
      if (pr) {
        lds_acc(accr + (poff + 0) * rstride, pa);
        lds_acc(accr + (poff + 1) * rstride, pb);
      } else {
        if constexpr (RED == 2) {
          const int ln = threadIdx.x & 63;
          const double vv = wave_sum2(pa, pb, ln);
          if ((ln & 31) == 0) lds_acc(accq + (poff + (ln >> 5)) * qstride, vv);
        } else {
          pa = flow_red<RED>(pa); pb = flow_red<RED>(pb);
          if (qlead) { lds_acc(accq + (poff + 0) * qstride, pa); lds_acc(accq + (poff + 1) * qstride, pb); }
        }
      }
    } else if (kind == TGP_FLOW_SAL) {
      sl -= 3;
      double uu[NB], ch[NB], gp[NB];
      TGP_EACH(u, NB) {
        uu[u] = stack[((sl + 0) * NB + u) * sstride];
        ch[u] = stack[((sl + 1) * NB + u) * sstride];
        gp[u] = stack[((sl + 2) * NB + u) * sstride];
      }
      if constexpr (PN) {
        if (pr) {
          TGP_EACH(u, NB) {
            const double fbu = rpn[u][2 * (poff + 1) + 1];
            gst[((sl + 0) * NB + u) * sstride] = -c[u] * ch[u];
            gst[((sl + 1) * NB + u) * sstride] = c[u] * uu[u] * ch[u] * fbu;
            c[u] *= gp[u];
          }
          continue;
        }
      }
      double fb = 1.0;
      if (pr) {
        if (flags & TGP_FLAG_RESTRICT) fb = sigmoid_d(rp[poff + 1]);
      } else {
        fb = F.tg[poff + 1];
      }
      double pa = 0.0, pb = 0.0;
      TGP_EACH(u, NB) {
        pa -= c[u] * ch[u];
        pb += c[u] * uu[u] * ch[u];
        c[u] *= gp[u];
      }
      pb *= fb;
      if (pr) {
        lds_acc(accr + (poff + 0) * rstride, pa);
        lds_acc(accr + (poff + 1) * rstride, pb);
      } else {
        if constexpr (RED == 2) {
          const int ln = threadIdx.x & 63;
          const double vv = wave_sum2(pa, pb, ln);
          if ((ln & 31) == 0) lds_acc(accq + (poff + (ln >> 5)) * qstride, vv);
        } else {
          pa = flow_red<RED>(pa); pb = flow_red<RED>(pb);
          if (qlead) { lds_acc(accq + (poff + 0) * qstride, pa); lds_acc(accq + (poff + 1) * qstride, pb); }
        }
      }
    } else {
      sl -= 1 + K;
      double gp[NB], fin[NB];
      TGP_EACH(u, NB) {
        gp[u] = (flags & TGP_FLAG_ADD_F0) ? 1.0 : 0.0;
        fin[u] = stack[(sl * NB + u) * sstride];
      }
      for (int k = 0; k < K; ++k) {
        const int o = poff + 4 * k;
        double th[NB];
        TGP_EACH(u, NB) th[u] = stack[((sl + 1 + k) * NB + u) * sstride];
        const double bt = F.tp[o + 1], cc = F.tp[o + 2], idt = flow_rcp_param(F, o + 3), g1 = F.tg[o + 1], g3 = F.tg[o + 3];
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        double se[NB], cs[NB], dz[NB];
        TGP_EACH(u, NB) se[u] = 1.0 - th[u] * th[u];
        TGP_EACH(u, NB) dz[u] = fin[u] - cc;
        TGP_EACH(u, NB) se[u] = bt * se[u] * idt;  // d/df of this step
        TGP_EACH(u, NB) cs[u] = c[u] * se[u];
        TGP_EACH(u, NB) dz[u] = cs[u] * dz[u] * idt;
        TGP_EACH(u, NB) {
          p0 += c[u];
          p1 += c[u] * th[u];
          p2 -= cs[u];
          p3 -= dz[u];
          gp[u] += se[u];
        }
        if constexpr (RED == 2) {
          const int ln = threadIdx.x & 63;
          const double vv = wave_sum4(p0, p1 * g1, p2, p3 * g3, ln);
          if ((ln & 15) == 0) lds_acc(accq + (o + (ln >> 4)) * qstride, vv);
        } else {
          p0 = flow_red<RED>(p0); p1 = flow_red<RED>(p1 * g1); p2 = flow_red<RED>(p2); p3 = flow_red<RED>(p3 * g3);
          if (qlead) {
            lds_acc(accq + (o + 0) * qstride, p0);
            lds_acc(accq + (o + 1) * qstride, p1);
            lds_acc(accq + (o + 2) * qstride, p2);
            lds_acc(accq + (o + 3) * qstride, p3);
          }
        }
      }
      TGP_EACH(u, NB) c[u] *= gp[u];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// "checkpoint" evaluation (stand-alone likelihood kernel, general-M path): the forward keeps ONE value per block and
// node -- the block's input -- and the reverse sweep recomputes the block from it.  Twice the transcendentals of the
// store mode, but 5 instead of 35 stack slots for the 5 x 6 tanh flow: the kernel is no longer limited to one
// workgroup per CU by the LDS stack, and 3 workgroups x NB = 4 nodes per lane give the f64 pipes the ~12 independent
// dependency chains per SIMD they need (the store-mode kernel ran at 1/18 of its instruction-issue bound).
// Shared-parameter partials: summed over the NB nodes in registers, over the wave by shuffles (fixed tree), added by
// lane 0 into the wave's own accumulator row accw[param] -- no atomics, fixed order.
// ---------------------------------------------------------------------------------------------------
template <int NB>
__device__ inline void flow_forward_ckpt(const FlowDev& F, double (&f)[NB], const double* __restrict__ rp, double* stack,
                                         int sstride) {
  for (int b = 0; b < F.nblk; ++b) {
    const int kind = F.prog[4 * b], K = F.prog[4 * b + 1], poff = F.prog[4 * b + 2], flags = F.prog[4 * b + 3];
    const bool pr = flags & TGP_FLAG_PER_ROW;
#pragma unroll
    for (int u = 0; u < NB; ++u) stack[(b * NB + u) * sstride] = f[u];
    if (kind == TGP_FLOW_AFFINE) {
      double a = flow_param(F, rp, poff, pr);
      if (pr && (flags & TGP_FLAG_RESTRICT)) a = softplus_d(a);
      const double bb = flow_param(F, rp, poff + 1, pr);
#pragma unroll
      for (int u = 0; u < NB; ++u) f[u] = a * f[u] + bb;
    } else if (kind == TGP_FLOW_SAL) {
      const double a = flow_param(F, rp, poff, pr);
      double bb = flow_param(F, rp, poff + 1, pr);
      if (pr && (flags & TGP_FLAG_RESTRICT)) bb = softplus_d(bb);
      const bool addf = flags & TGP_FLAG_ADD_F0;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const double q1 = f[u] * f[u] + 1.0;
        const double isf = rsqrt_nr_fwd(q1);
        double sf = q1 * isf;
        sf = fma(fma(-sf, sf, q1), 0.5 * isf, sf);
        const double e = exp_fast(bb * log_fast(f[u] + sf) - a), ei = rcp_fast(e);
        const double g = 0.5 * (e - ei);
        f[u] = addf ? g + f[u] : g;
      }
    } else {
      const bool addf = flags & TGP_FLAG_ADD_F0;
      double g[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) g[u] = addf ? f[u] : 0.0;
      for (int k = 0; k < K; ++k) {
        const double a = F.tp[poff + 4 * k], bt = F.tp[poff + 4 * k + 1], c = F.tp[poff + 4 * k + 2],
                     idt = flow_rcp_param(F, poff + 4 * k + 3);
        // (stage by stage over the nodes in flight: independent chains for the scheduler, as in the store-mode sweep)
        double e[NB];
        TGP_EACH(u, NB) e[u] = 2.0 * (f[u] - c) * idt;
        exp_fast_n<NB>(e);
        TGP_EACH(u, NB) e[u] += 1.0;
        rcp_fast_n<NB>(e);
        TGP_EACH(u, NB) g[u] += a + bt * (1.0 - 2.0 * e[u]);
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) f[u] = g[u];
    }
  }
}

// c[u] = d(objective)/dG on entry, d(objective)/df0 on exit.  All lanes of the wave must call this together.
template <int NB>
__device__ inline void flow_backward_ckpt(const FlowDev& F, double (&c)[NB], const double* __restrict__ rp, const double* stack,
                                          int sstride, double* accw, int lane, double* accr, int rstride) {
  for (int b = F.nblk - 1; b >= 0; --b) {
    const int kind = F.prog[4 * b], K = F.prog[4 * b + 1], poff = F.prog[4 * b + 2], flags = F.prog[4 * b + 3];
    const bool pr = flags & TGP_FLAG_PER_ROW;
    double fin[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) fin[u] = stack[(b * NB + u) * sstride];
    if (kind == TGP_FLOW_AFFINE) {
      double a, fa;
      if (pr) {
        a = rp[poff]; fa = 1.0;
        if (flags & TGP_FLAG_RESTRICT) { fa = sigmoid_d(a); a = softplus_d(a); }
      } else {
        a = F.tp[poff]; fa = F.tg[poff];
      }
      double pa = 0.0, pb = 0.0;
#pragma unroll
      for (int u = 0; u < NB; ++u) { pa += c[u] * fin[u]; pb += c[u]; c[u] *= a; }
      pa *= fa;
      if (pr) {
        accr[(poff + 0) * rstride] += pa;
        accr[(poff + 1) * rstride] += pb;
      } else {
        const double vv = wave_sum2(pa, pb, lane);
        if ((lane & 31) == 0) accw[poff + (lane >> 5)] += vv;
      }
    } else if (kind == TGP_FLOW_SAL) {
      const double a = flow_param(F, rp, poff, pr);
      double bb = flow_param(F, rp, poff + 1, pr), fb = pr ? 1.0 : F.tg[poff + 1];
      if (pr && (flags & TGP_FLAG_RESTRICT)) { fb = sigmoid_d(bb); bb = softplus_d(bb); }
      const bool addf = flags & TGP_FLAG_ADD_F0;
      double pa = 0.0, pb = 0.0;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const double q1 = fin[u] * fin[u] + 1.0;
        const double isf = rsqrt_nr_fwd(q1);
        double sf = q1 * isf;
        sf = fma(fma(-sf, sf, q1), 0.5 * isf, sf);
        const double uu = log_fast(fin[u] + sf);
        const double e = exp_fast(bb * uu - a), ei = rcp_fast(e);
        const double ch = 0.5 * (e + ei);
        double gp = bb * ch * isf;
        if (addf) gp += 1.0;
        pa -= c[u] * ch;
        pb += c[u] * uu * ch;
        c[u] *= gp;
      }
      pb *= fb;
      if (pr) {
        accr[(poff + 0) * rstride] += pa;
        accr[(poff + 1) * rstride] += pb;
      } else {
        const double vv = wave_sum2(pa, pb, lane);
        if ((lane & 31) == 0) accw[poff + (lane >> 5)] += vv;
      }
    } else {
      double gp[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) gp[u] = (flags & TGP_FLAG_ADD_F0) ? 1.0 : 0.0;
      for (int k = 0; k < K; ++k) {
        const int o = poff + 4 * k;
        const double bt = F.tp[o + 1], cc = F.tp[o + 2], idt = flow_rcp_param(F, o + 3);
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        double t[NB], e[NB];
        TGP_EACH(u, NB) t[u] = (fin[u] - cc) * idt;
        TGP_EACH(u, NB) e[u] = 2.0 * t[u];
        exp_fast_n<NB>(e);
        TGP_EACH(u, NB) e[u] += 1.0;
        rcp_fast_n<NB>(e);
        TGP_EACH(u, NB) {
          const double th = 1.0 - 2.0 * e[u];
          const double se = bt * (1.0 - th * th) * idt;
          p0 += c[u];
          p1 += c[u] * th;
          p2 -= c[u] * se;
          p3 -= c[u] * se * t[u];
          gp[u] += se;
        }
        // the four partials of this step in ONE reduce-scatter over the wave (wave_sum4: lane 16 q ends with the total of value
        // q) instead of four butterflies -- round 6, VERDICT r5 #9: the reductions were a fifth of the sweep's instructions
        const double vv = wave_sum4(p0, p1 * F.tg[o + 1], p2, p3 * F.tg[o + 3], lane);
        if ((lane & 15) == 0) accw[o + (lane >> 4)] += vv;
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) c[u] *= gp[u];
    }
  }
}

}  // namespace tgp
