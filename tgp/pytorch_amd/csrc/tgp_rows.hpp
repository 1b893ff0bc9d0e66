// tgp_rows.hpp -- the fused row kernel of the ELBO step (one launch covers forward AND backward of
// everything that scales with the number of rows).
//
// Replaces, per minibatch row n (reference: models/sparse_MF_SP.py:313-396, likelihoods/*.py,
// models/flow.py and the autograd replay of all of it, trainers/trainer_base.py:341):
//   K_n  = sigma^2 exp(-1/2 |xs_n - zs_j|^2)                 never written to HBM
//   A    = L^-1 K_MN      (forward substitution on MFMA)     reference: triangular_solve :380
//   B    = L_q^T A        (upper-triangular MFMA GEMM)       reference: S = LqLq^T, matmul(S, rhs) :346,382
//   mu   = m^T A ;  v = sigma^2 - sum A^2 + sum B^2          reference: :354-355, :376-382
//   ell_n, d ell/d mu, d ell/d v, d ell/d theta, d ell/d eta  Gauss-Hermite through the flow (or closed form)
//   Abar = m mubar^T - 2 A vbar + 2 L_q (B vbar) ; Kbar = L^-T Abar   (triangular MFMA GEMM + back substitution)
// The two solves with L are SUBSTITUTIONS over 16-row tiles (round 4): A_i = Dinv_i (K_i - sum_{kb<i} L_i,kb A_kb) and
// Kbar_i = Dinv_i^T (Abar_i - sum_{kb>i} L_kb,i^T Kbar_kb), the same MT (MT+1)/2 tile products as a product with an
// explicit J = L^-1 and the reference's own solve shape -- so the prepare launch hands over only L, L^T and the
// inverses of the 16 x 16 diagonal tiles, and J (which only the backward M x M chain still wants) is formed by this
// launch's passenger blocks on CUs the row tiles leave idle.
//   row statistics  G = A diag(vbar) A^T, s = A mubar, T = (Kbar o K) [xs, xs^2, 1]   (MFMA, via LDS transpose)
// One workgroup = 4 waves = 64 rows; one wave owns 16 rows and keeps K, A, B, Abar, Kbar in registers:
// the accumulator layout of v_mfma_f64_16x16x4 is directly the B-operand layout of the next product.
// Per-block statistics go to a slab in HBM (deterministic two-pass reduction, no atomics).
#pragma once
#include "tgp_dev.hpp"
#include "tgp_prep.hpp"

namespace tgp {

#define TGP_NODES_IN_FLIGHT 4

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).  __syncthreads() also drains vmcnt:
// the operand panels requested two ahead (global loads meant to land under the MFMA chain of the current panel) were
// waited for AT every barrier, and so were the slab stores of the epilogue.  Everything the waves of a row kernel
// exchange goes through LDS; what they store to global memory is read by no wave of the same launch.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct RowArgs {
  Plan p;
  const double* X;
  const double* Y;
  const double* rowp;
  double* g_rowp;
  double* mu;
  double* v;
  double* ws;
  FlowProg prog;
  const double* xs;
  const double* wn;
  double scale;
};

// LDS carve-up (offsets in doubles)
struct RowLds {
  size_t zs, ils, mv, tile, xt, vbs, mbs, tp, tg, ti, xs, wn, stack, acc, red, prog, rin, cbuf, rpl, total;
};
#define TGP_RW_SMALL 10          /* data rows per wave of the k_rows<.., RW = 10> variant (see the kernel) */
#define TGP_RW_NODES 5           /* its quadrature nodes in flight per lane: 10 rows x 32 nodes = 64 lanes x 5 */

// mode: 0 = moments only; 1 = training, TGP_NODES_IN_FLIGHT quadrature nodes in flight per lane; 2 = training, one
// node in flight (fallback when the 4-node stack of a long flow does not fit in LDS); nslots = flow_slots(program)
__host__ __device__ inline RowLds row_lds(const Plan& p, int mode, int nslots, int rw = 16) {
  const bool train = mode != 0;
  RowLds L;
  size_t o = 0;
  auto take = [&o](size_t n) { size_t r = o; o += (n + 1) / 2 * 2; return r; };  // keep 16-byte alignment
  L.zs = take((size_t)p.MP * p.DP);
  L.ils = take(16);
  L.mv = take(p.MP);
  L.tp = take(p.P + 1);
  L.tg = take(p.P + 1);
  L.ti = take(p.P + 1);
  L.xs = take(p.S + 1);
  L.wn = take(p.S + 1);
  L.prog = take((size_t)2 * p.nblk + 2);  // 4 int32 per block
  L.red = take(32);
  if (train) {
    L.tile = take((size_t)p.MP * TGP_TILE_LD);
    L.xt = take((size_t)TGP_ROWS_PER_BLOCK * p.CT16);
    L.vbs = take(TGP_ROWS_PER_BLOCK);
    L.mbs = take(TGP_ROWS_PER_BLOCK);
    // the flow stack (nodes-in-flight x slots x 256 lanes) shares its space with the transposition tile / the operand
    // panels: the flow phase runs strictly between GEMM 2 and GEMM 3
    // (tile, xt, vbs, mbs are contiguous and all dead during the flow phase: the stack may cover all of them)
    const size_t st = (size_t)(nslots > 0 ? nslots : 1) * (mode == 1 ? (rw < 16 ? TGP_RW_NODES : TGP_NODES_IN_FLIGHT) : 1) * 256;
    if (L.tile + st > o) take(L.tile + st - o);
    L.stack = L.tile;
    L.acc = take((size_t)(p.P > 0 ? p.P : 1) * 64 + (size_t)p.RP * 256);  // [P][64] quad-reduced + [RP][256] per lane
    L.rin = L.cbuf = L.rpl = o;
    if (rw < 16) {   // the quadrature's (row, node) pairs are dealt over all 64 lanes: per-wave row inputs and pair results
      L.rin = take(4 * 48);
      L.cbuf = take((size_t)4 * 64 * TGP_RW_NODES);
      L.rpl = take((size_t)4 * 16 * 2 * p.RP);   // per wave and row: (transformed value, d value / d raw) of every per-row column
    }
  } else {
    L.tile = take((size_t)p.MP * 32);  // only the two operand panels (2 x MP x 16)
    L.xt = L.vbs = L.mbs = L.stack = L.acc = L.rin = L.cbuf = L.rpl = o;
  }
  L.total = o;
  return L;
}

// One output tile of a triangular GEMM of the row kernel: NSTEPS k-steps (4 rows of the operand panel each) starting
// at k-step `step0`.  `a0` = &panel[q*16 + nl] (k-step s is 64 doubles further), fb(s) = B operand of local step s.
// Operands are read from LDS in batches of 8 BEFORE their MFMAs (left to itself hipcc emits ds_read -> wait -> mfma
// one by one: ~120 cycles per MFMA instead of 64).  ONE accumulator: back-to-back MFMAs on the same accumulator are
// forwarded inside the matrix pipe (64 cycles apart); two alternating accumulators measured slower (gemmphase_rate.hip).
// (nsteps is a constant after the caller's tile loop is unrolled; MAXSTEPS bounds the unrolling.)
template <int MAXSTEPS, class FB>
__device__ __forceinline__ d4 mfma_chain(const double* a0, int step0, int nsteps, FB fb) {
  d4 c = {0, 0, 0, 0};
#pragma unroll
  for (int s0 = 0; s0 < MAXSTEPS; s0 += 8) {
    double av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < nsteps) av[u] = a0[(step0 + s0 + u) * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < nsteps) c = TGP_MFMA(av[u], fb(s0 + u), c);
    // pin the shape of this batch in the emitted code: all LDS reads first, then the MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
  }
  return c;
}

// One tile of a substitution with L (forward or backward): the running right-hand side `c` (accumulator layout) first
// collects nsum k-steps  c += panel[fa(s)] * fb(s)  (fb = register r of an earlier result tile: the accumulator layout
// IS the B-operand layout), then the tile closes with the four k-steps of the diagonal block at panel k-step dstep0,
// whose B operand is c itself:  out = Dblock * c.  With c = -(rhs tile), the panel holding +L and the diagonal block
// holding -Dinv this is out = Dinv (rhs - sum L x).  Operand reads batched eight ahead of their MFMAs as in mfma_chain.
template <int MAXSTEPS, class FA, class FB>
__device__ __forceinline__ d4 subst_chain(const double* a0, int nsum, int dstep0, d4 c, FA fa, FB fb) {
  d4 out = {0, 0, 0, 0};
#pragma unroll
  for (int s0 = 0; s0 < MAXSTEPS; s0 += 8) {
    double av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u;
      if (s < nsum) av[u] = a0[fa(s) * 64];
      else if (s < nsum + 4) av[u] = a0[(dstep0 + s - nsum) * 64];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u;
      if (s < nsum) c = TGP_MFMA(av[u], fb(s), c);
      else if (s < nsum + 4) out = TGP_MFMA(av[u], c[s - nsum], out);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
  }
  return out;
}

#ifdef TGP_STAMPS
#define ROW_STAMP(wsp, plan, i)                                                                       \
  do {                                                                                                \
    if (bid == 0 && threadIdx.x == 0)                                                                 \
      (wsp)[(plan).hdr + tgp::H_STAMP + (i)] = (double)__builtin_amdgcn_s_memrealtime();              \
  } while (0)
#else
#define ROW_STAMP(wsp, plan, i) \
  do {                          \
  } while (0)
#endif

// One wave per SIMD by construction (4 waves per workgroup, one workgroup per CU): tell the register allocator and the
// scheduler so, otherwise hipcc schedules to minimise VGPRs and serialises every LDS read behind its MFMA.
// RW = data rows per wave: 16 (every column of the 16 x 16 x 4 tiles carries a row), or TGP_RW_SMALL = 10 (round 5; training
// with the flow likelihood and shared flow parameters only).  The tile chains of a wave cost the same whatever the number
// of columns that carry rows, but the quadrature and the row statistics do not: at Power size 16 rows per wave are 539 waves
// for 1 024 SIMDs, 10 rows per wave are 862 waves in 216 workgroups -- 10 x 32 (row, node) pairs are exactly 5 per lane,
// dealt over ALL 64 lanes (pair p = lane + 64 u: row p % 10, node p / 10) with five nodes in flight per lane instead of two
// trips of four (row inputs and pair results cross the wave through LDS), and the statistics contract over 40 columns
// instead of 64.
template <int MT, int DP, int MODE, int RW = 16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_rows(RowArgs a) {
  static_assert(RW == 16 || (RW == TGP_RW_SMALL && MODE == 1), "rows per wave: 16, or TGP_RW_SMALL in training mode 1");
  constexpr bool TRAIN = MODE != 0;
  constexpr int RBK = 4 * RW;   // data rows (= statistics columns) per workgroup
  constexpr int MP = MT * 16;
  constexpr int CT = (2 * DP + 1 + 15) / 16, CT16 = CT * 16;
  constexpr int LD = TGP_TILE_LD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const Plan& p = a.p;
  const RowLds L = row_lds(p, MODE, a.prog.nslots, RW);
  double* zs = sm + L.zs;
  double* ils = sm + L.ils;
  double* mv = sm + L.mv;
  double* tpL = sm + L.tp;
  double* tgL = sm + L.tg;
  double* tiL = sm + L.ti;
  double* xsL = sm + L.xs;
  double* wnL = sm + L.wn;
  int32_t* progL = reinterpret_cast<int32_t*>(sm + L.prog);
  double* red = sm + L.red;
  double* tile = sm + L.tile;
  double* xt = sm + L.xt;
  double* vbs = sm + L.vbs;
  double* mbs = sm + L.mbs;
  double* stack = sm + L.stack;
  double* acc = sm + L.acc;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 15, q = lane >> 4;
  const double* __restrict__ ws = a.ws;
  const int N = p.N, D = p.D, M = p.M, P = p.P, RP = p.RP;
  const int bid = blockIdx.x;                          // index among the row blocks, then the passengers

  if (TRAIN && bid >= p.nblocks) {
    // ---- passenger blocks (one per 16-column block c of J): what only the backward M x M chain needs -- J = L^-1
    //      (k_bwd's row blocks), H'^T = (J^T (S - I))^T (its column blocks) and w = J^T m -- formed on CUs the row tiles leave idle instead
    //      of on the prepare launch's critical chain.  Column block c of J is the forward substitution with the unit
    //      columns 16c .. 16c+15 as right-hand side, J_cc = Dinv_c, J_ic = -Dinv_i sum_{c<=kb<i} L_i,kb J_kb,c : the same
    //      register chain as the row waves' (a finished tile IS the next product's B operand), every L fragment it
    //      needs requested up front.  All four waves run the chain (nothing to exchange); wave 0 stores the column of
    //      J, wave w the tiles (c, w), (c, w+4) of H' -- whose A operand is again the accumulator layout of the J tiles,
    //      now read as J^T -- and wave 3 the 16 entries of w.
    const int c = __builtin_amdgcn_readfirstlane(bid - p.nblocks);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const double* __restrict__ LTg = ws + p.LT;
    const double* __restrict__ nDg = ws + p.nD;
    const double* __restrict__ Sg = ws + p.S_;
    constexpr int NLF = MT > 1 ? MT * (MT - 1) / 2 : 1;
    double lf[NLF][4], df[MT][4];
    d4 Jt[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i < c) continue;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) df[i][s4] = *(nDg + i * 256 + nl * 16 + 4 * s4 + q);   // A operand of -Dinv_i
#pragma unroll
      for (int kb = 0; kb < i; ++kb) {
        if (kb < c) continue;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) lf[i * (i - 1) / 2 + kb][s4] = *(LTg + (size_t)(16 * kb + 4 * s4 + q) * MP + 16 * i + nl);
      }
    }
    d4 dc;  // Dinv_c in accumulator layout
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) dc[rr] = -*(nDg + c * 256 + (4 * rr + q) * 16 + nl);
    double mf[MT][4];  // this lane's entries of m (for w)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) mf[i][rr] = *(ws + p.mpad + 16 * i + 4 * rr + q);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      Jt[i] = d4{0, 0, 0, 0};
      if (i < c) continue;
      if (i == c) { Jt[i] = dc; continue; }
      d4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < i; ++kb) {
        if (kb < c) continue;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = TGP_MFMA(lf[i * (i - 1) / 2 + kb][s4], Jt[kb][s4], acc);
      }
      d4 o = {0, 0, 0, 0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) o = TGP_MFMA(df[i][s4], acc[s4], o);
      Jt[i] = o;
    }
    if (wv == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i < c) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a.ws[p.J + (size_t)(16 * i + 4 * rr + q) * MP + 16 * c + nl] = Jt[i][rr];
      }
    }
    for (int jj = wv; jj < MT; jj += 4) {
      double bf[MT][4];
#pragma unroll
      for (int kb = 0; kb < MT; ++kb) {
        if (kb < c) continue;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int kk = 16 * kb + 4 * s4 + q;
          bf[kb][s4] = *(Sg + (size_t)kk * MP + 16 * jj + nl) - (kk == 16 * jj + nl ? 1.0 : 0.0);
        }
      }
      d4 h = {0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < MT; ++kb) {
        if (kb < c) continue;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) h = TGP_MFMA(Jt[kb][s4], bf[kb][s4], h);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) a.ws[p.HpT + (size_t)(16 * jj + nl) * MP + 16 * c + 4 * rr + q] = h[rr];
    }
    if (wv == 3) {
      double sw = 0.0;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i < c) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) sw += Jt[i][rr] * mf[i][rr];
      }
      sw = quad_sum(sw);
      if (q == 0) a.ws[p.w + 16 * c + nl] = sw;
    }
    return;
  }

  ROW_STAMP(a.ws, p, 0);
#ifdef TGP_STAMPS
  if (bid == 0 && threadIdx.x == 0) a.ws[p.hdr + H_STAMP + 11] = (double)clock64();
#endif
  // this lane's data row and the step header: requested with the first staging loads (they need nothing from LDS;
  // behind the staging barrier they cost a memory round trip of their own)
  const int n = bid * RBK + wave * RW + nl;
  const bool valid = nl < RW && n < N;
  const int nc = valid ? n : N - 1;
  double xraw[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) xraw[d] = d < D ? a.X[(size_t)nc * D + d] : 0.0;
  const double y = TRAIN ? a.Y[nc] : 0.0;
  // (RW < 16 with per-row flow parameters: this lane's columns q, q + 4 of its row, requested with the first loads of the
  //  kernel -- at the head of the quadrature they cost a memory round trip of their own)
  double rraw[2] = {0.0, 0.0};
  if constexpr (RW < 16) {
    if (a.rowp != nullptr && RP > 0 && RP <= 8) {
      rraw[0] = q < RP ? a.rowp[(size_t)nc * RP + q] : 0.0;
      rraw[1] = q + 4 < RP ? a.rowp[(size_t)nc * RP + q + 4] : 0.0;
    }
  }
  const double s2 = *(ws + p.hdr + H_S2), eta = *(ws + p.hdr + H_ETA),
               einv = *(ws + p.hdr + H_EINV);

  // ---- operand panels: the A operands of the four triangular products (16 columns x up to MP rows of L^T, Lq, Lq^T, L)
  //      are staged by the whole workgroup through two LDS buffers (aliased on the transposition tile, which is
  //      only used after the products): coalesced 128-byte row segments in, conflict-free 512-byte wave reads out.
  //      Panels are prefetched TWO ahead through two register sets (the early panels feed only 4-8 MFMAs, far less
  //      than one L2 round trip), and the first two are requested here, at the top of the kernel: the K tile takes
  //      about 1 us, far less than their round trip.
  //      Panel kinds:  0 = lower-type panel i of L^T (rows [0, 16 i)) closed by -Dinv_i^T as row block i   (A = L^-1 K)
  //                    1 = upper-type panel i of Lq  (rows [16 i, MP))                                      (B = Lq^T A)
  //                    2 = lower-type panel i of Lq^T (rows [0, 16 (i+1)))                                  (C = Lq (B vbar))
  //                    3 = upper-type panel i of L (rows [16 (i+1), MP)) headed by -Dinv_i as row block i   (Kbar = L^-T Abar)
  const double* __restrict__ LTm = ws + p.LT;
  const double* __restrict__ Lm = ws + p.L;
  const double* __restrict__ nD = ws + p.nD;
  const double* __restrict__ Lq = ws + p.Lq;
  const double* __restrict__ LqT = ws + p.LqT;
  double* pan = tile;  // 2 x (MP x 16)
  double stg[2][MT];
  // (row-block u of a panel exists iff u < nb with nb = i+1 (lower) / MT-i (upper): a compile-time predicate once the
  //  tile loops are unrolled -- a `row < r1` test would cost an exec-mask branch around every load and store)
  auto issue = [&](int kind, int i, double (&st)[MT]) {
    const bool lower = kind == 0 || kind == 2;
    const double* __restrict__ Mt = (kind == 0 ? LTm : (kind == 1 ? Lq : (kind == 2 ? LqT : Lm))) +
                                    (size_t)((lower ? 0 : 16 * i) + (tid >> 4)) * MP + 16 * i + (tid & 15);
    const int nb = lower ? i + 1 : MT - i;
#pragma unroll
    for (int u = 0; u < MT; ++u)
      if (u < nb) {
        if (kind == 0 && u == i) st[u] = *(nD + i * 256 + (tid & 15) * 16 + (tid >> 4));        // (-Dinv_i)^T
        else if (kind == 3 && u == 0) st[u] = *(nD + i * 256 + (tid >> 4) * 16 + (tid & 15));   // -Dinv_i
        else st[u] = *(Mt + (size_t)16 * u * MP);
      }
  };
  auto commit = [&](int par, bool lower, int i, const double (&st)[MT]) {
    const int nb = lower ? i + 1 : MT - i;
    double* buf = pan + par * (MP * 16) + ((lower ? 0 : 16 * i) + (tid >> 4)) * 16 + (tid & 15);
#pragma unroll
    for (int u = 0; u < MT; ++u)
      if (u < nb) buf[16 * u * 16] = st[u];
  };
  // panel sequences of the two phases (pp = 0 .. 2 MT - 1; buffer and register set = pp & 1)
  auto issue1 = [&](int pp, double (&st)[MT]) { if (pp < MT) issue(0, pp, st); else issue(1, pp - MT, st); };
  auto issue2 = [&](int pp, double (&st)[MT]) { if (pp < MT) issue(2, pp, st); else issue(3, 2 * MT - 1 - pp, st); };
  issue1(0, stg[0]);
  if (MT * 2 > 1) issue1(1, stg[1]);
  // ---- stage the small shared operands ----
  // (the first slice of every array is requested before anything is stored: the loops below, one after the other,
  //  paid one L2 round trip each)
  {
    constexpr int NZ = (MP * DP + 255) / 256;
    double zv[NZ];
#pragma unroll
    for (int u = 0; u < NZ; ++u) zv[u] = tid + 256 * u < MP * DP ? *(ws + p.Zs + tid + 256 * u) : 0.0;
    const double mv0 = tid < MP ? *(ws + p.mpad + tid) : 0.0;
    const double il0 = tid < 16 ? *(ws + p.ils + tid) : 0.0;
    const bool fl = p.lik == TGP_LIK_FLOW;
    const double tp0 = (fl && tid < P) ? *(ws + p.tp + tid) : 0.0, tg0 = (fl && tid < P) ? *(ws + p.tg + tid) : 0.0;
    const double xs0 = (fl && tid < p.S) ? a.xs[tid] : 0.0, wn0 = (fl && tid < p.S) ? a.wn[tid] : 0.0;
#pragma unroll
    for (int u = 0; u < NZ; ++u)
      if (tid + 256 * u < MP * DP) zs[tid + 256 * u] = zv[u];
    if (tid < MP) mv[tid] = mv0;
    if (tid < 16) ils[tid] = il0;
    if (fl) {
      if (tid < P) { tpL[tid] = tp0; tgL[tid] = tg0; tiL[tid] = rcp_fast(tp0); }
      if (tid < p.S) { xsL[tid] = xs0; wnL[tid] = wn0; }
      for (int i = tid + 256; i < P; i += 256) { tpL[i] = *(ws + p.tp + i); tgL[i] = *(ws + p.tg + i); tiL[i] = rcp_fast(tpL[i]); }
      for (int i = tid + 256; i < p.S; i += 256) { xsL[i] = a.xs[i]; wnL[i] = a.wn[i]; }
      for (int i = tid; i < 4 * p.nblk; i += 256) progL[i] = a.prog.blk[i];
    }
  }
  if (TRAIN) {
    const int nacc = P * 64 + RP * 256;
    for (int i = tid; i < nacc; i += 256) acc[i] = 0.0;
  }
  lds_barrier();

  ROW_STAMP(a.ws, p, 1);
  double x[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) x[d] = d < D ? xraw[d] * ils[d] : 0.0;


  // ---- K tile in B-operand layout: Kr[ks] = K[m = 4 ks + q][row nl] ----
  double Kr[4 * MT];
  // (four tiles' worth of exponentials at a time, stage by stage: independent chains for the single wave of this SIMD;
  //  padding rows mm >= M are computed from the zero-filled Zs rows and multiplied by 0)
#pragma unroll
  for (int k0 = 0; k0 < 4 * MT; k0 += 4) {
    double e[4];
    TGP_EACH(u, 4) {
      const int mm = 4 * (k0 + u) + q;
      double d2 = 0.0;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const double t = x[d] - zs[mm * DP + d];
        d2 += t * t;
      }
      e[u] = -0.5 * d2;
    }
    exp_fast_n<4>(e);
    // (a masked multiplier, not a select: hipcc turns `cond ? s2 * e : 0` back into a branch around the whole chain)
    TGP_EACH(u, 4) Kr[k0 + u] = (4 * (k0 + u) + q < M ? s2 : 0.0) * e[u];
  }

  ROW_STAMP(a.ws, p, 2);
  d4 Aa[MT], Ba[MT];
  // ---- A = L^-1 K by forward substitution: A_i = Dinv_i (K_i - sum_{kb < i} L[i,kb] A_kb)  (A operand = rows of L^T,
  //      then of -Dinv_i^T; the running right-hand side starts as -K_i, which already sits in accumulator layout) ----
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const double* buf = pan + (i & 1) * (MP * 16);
    commit(i & 1, true, i, stg[i & 1]);
    lds_barrier();
    if (i + 2 < 2 * MT) issue1(i + 2, stg[i & 1]);
    const d4 c0 = {-Kr[4 * i], -Kr[4 * i + 1], -Kr[4 * i + 2], -Kr[4 * i + 3]};
    Aa[i] = subst_chain<4 * MT>(buf + q * 16 + nl, 4 * i, 4 * i, c0, [&](int st) { return st; },
                                [&](int st) { return Aa[st / 4][st % 4]; });
  }
  ROW_STAMP(a.ws, p, 3);
  // ---- B = Lq^T A : B_i = sum_{kb >= i} Lq[kb,i]^T A_kb ; accumulator register r of A_kb is k-step r ----
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const double* buf = pan + ((MT + i) & 1) * (MP * 16);
    commit((MT + i) & 1, false, i, stg[(MT + i) & 1]);
    lds_barrier();
    if (MT + i + 2 < 2 * MT) issue1(MT + i + 2, stg[(MT + i) & 1]);
    Ba[i] = mfma_chain<4 * MT>(buf + q * 16 + nl, 4 * i, 4 * (MT - i), [&](int st) { return Aa[i + st / 4][st % 4]; });
  }
  ROW_STAMP(a.ws, p, 4);
  // ---- mu, v ----
  double pm = 0.0, pa = 0.0, pb = 0.0;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pm += mv[16 * i + 4 * r + q] * Aa[i][r];
      pa += Aa[i][r] * Aa[i][r];
      pb += Ba[i][r] * Ba[i][r];
    }
  pm = quad_sum(pm); pa = quad_sum(pa); pb = quad_sum(pb);
  const double mu = pm, v = s2 - pa + pb;
  if (a.mu != nullptr && q == 0 && valid) { a.mu[n] = mu; a.v[n] = v; }
  if (!TRAIN) return;

  ROW_STAMP(a.ws, p, 5);
  // ---- expected log-likelihood and its adjoints ----
  double mub = 0.0, vb = 0.0, ellp = 0.0, etap = 0.0;
  if (p.lik == TGP_LIK_GAUSS) {
    // GaussianLinearMean.expected_log_prob (likelihoods/GaussianLinearMean.py:81-87)
    const double r = y - mu;
    mub = a.scale * einv * r;
    vb = -0.5 * a.scale * einv;
    if (q == 0) {
      ellp = -0.5 * TGP_LOG_2PI_REF - 0.5 * eta - 0.5 * einv * (r * r + v);
      etap = -0.5 + 0.5 * einv * (r * r + v);
    }
  } else if (p.lik == TGP_LIK_ADJOINT) {
    // tgp_qf_moments_bwd_f64: the adjoints of (mu, v) are the caller's (mu_bar in the Y slot, v_bar in the rowp slot)
    mub = y;
    vb = a.rowp[nc];
  } else {
    FlowDev F{progL, p.nblk, tpL, tgL, tiL};
    const double sq = sqrt(2.0 * v);
    if constexpr (RW < 16) {
      // GaussianNonLinearMean.expected_log_prob (likelihoods/GaussianNonLinearMean.py:91-148) with the wave's RW x S
      // (row, node) pairs dealt over all 64 lanes, NB per lane in ONE trip (the launcher checked RW * S <= 64 NB and
      // shared flow parameters): pair p = lane + 64 u is node p / RW of row p % RW.  The row lanes publish (mu, sqrt(2 v),
      // y) through the wave's LDS words, every pair's d(ell)/d(f0) goes back the same way and the row lanes add their
      // row's nodes in a fixed order.
      constexpr int NB = TGP_RW_NODES;
      double* rin = sm + L.rin + wave * 48;
      double* cb = sm + L.cbuf + wave * (64 * NB);
      if (q == 0) { rin[nl] = mu; rin[16 + nl] = sq; rin[32 + nl] = y; }
      double* rpl = sm + L.rpl + wave * (16 * 2 * RP);
      if (a.rowp != nullptr && RP > 0 && nl < RW) {
        // this row's per-row flow parameters, transformed ONCE (lane q takes the columns q, q + 4, ...)
        for (int col = q; col < RP; col += 4) {
          const double raw = RP <= 8 ? rraw[col >> 2] : a.rowp[(size_t)nc * RP + col];
          double val = raw, der = 1.0;
          for (int b = 0; b < p.nblk; ++b) {
            const int kind = progL[4 * b], poff = progL[4 * b + 2], flags = progL[4 * b + 3];
            if ((flags & TGP_FLAG_PER_ROW) && (flags & TGP_FLAG_RESTRICT) && col == poff + (kind == TGP_FLOW_AFFINE ? 0 : 1)) {
              val = softplus_d(raw);
              der = sigmoid_d(raw);
            }
          }
          rpl[(nl * RP + col) * 2] = val;
          rpl[(nl * RP + col) * 2 + 1] = der;
        }
      }
      double* accq = acc + wave * 16 + nl;
      double* accr = acc + (size_t)P * 64 + tid;
      lds_barrier();  // the stack aliases the operand panels: all waves must be done with GEMM 2 (and rin is visible)
      double xn[NB], wq[NB], f[NB], c[NB], yv[NB];
      const double* rpn[NB];   // LDS table of each pair's row (ID_TGP; the launcher admits per-row SAL blocks only)
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int pp = lane + 64 * u, row = pp % RW, sn = pp / RW;
        const int nr = bid * RBK + wave * RW + row;
        const bool ok = sn < p.S && nr < N;
        xn[u] = xsL[sn < p.S ? sn : 0];
        wq[u] = ok ? wnL[sn < p.S ? sn : 0] : 0.0;
        f[u] = rin[row] + rin[16 + row] * xn[u];
        yv[u] = rin[32 + row];
        rpn[u] = rpl + row * RP * 2;
      }
      flow_forward_store<NB, true>(F, f, nullptr, stack + tid, 256, rpn);
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const double r = yv[u] - f[u];
        ellp += wq[u] * (-0.5 * TGP_LOG_2PI_REF - 0.5 * eta - 0.5 * einv * r * r);
        etap += wq[u] * (-0.5 + 0.5 * einv * r * r);
        c[u] = a.scale * einv * wq[u] * r;
      }
      flow_backward_store<NB, 0, true>(F, c, nullptr, stack + tid, 256, a.prog.nslots, accq, 64, q == 0, accr, 256, rpn, stack + tid);
#pragma unroll
      for (int u = 0; u < NB; ++u) cb[lane + 64 * u] = c[u];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (a.g_rowp != nullptr && RP > 0) {
        // per-row parameter gradients: the pairs of a per-row SAL block left (d/da, d/db) in the block's two first stack
        // slots; row lane (nl, q) adds the nodes q, q + 4, ... of its row, the quad completes the sum (fixed order)
        int sl = 0;
        for (int b = 0; b < p.nblk; ++b) {
          const int kind = progL[4 * b], K = progL[4 * b + 1], poff = progL[4 * b + 2], flags = progL[4 * b + 3];
          if (kind == TGP_FLOW_SAL && (flags & TGP_FLAG_PER_ROW)) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              double gsum = 0.0;
              if (nl < RW) {
                for (int sn = q; sn < p.S; sn += 4) {
                  const int pp = sn * RW + nl;
                  gsum += stack[((sl + jj) * NB + (pp >> 6)) * 256 + wave * 64 + (pp & 63)];
                }
              }
              gsum = quad_sum(gsum);
              if (q == 0 && valid) a.g_rowp[(size_t)n * RP + poff + jj] = gsum;
            }
          }
          sl += kind == TGP_FLOW_AFFINE ? 1 : (kind == TGP_FLOW_SAL ? 3 : 1 + K);
        }
      }
      // row lane (nl, q): the nodes q, q + 4, ... of row nl (pair index node * RW + row)
      double cm = 0.0, cv = 0.0;
      if (nl < RW) {
        for (int sn = q; sn < p.S; sn += 4) {
          const double cc = cb[sn * RW + nl];
          cm += cc;
          cv += cc * xsL[sn];
        }
      }
      mub = quad_sum(cm);
      vb = quad_sum(cv) / sq;
      if (!valid) { mub = 0.0; vb = 0.0; }   // (ellp / etap of this lane belong to its PAIRS and are already weighted)
    } else {
      // GaussianNonLinearMean.expected_log_prob (likelihoods/GaussianNonLinearMean.py:91-148): this lane takes the
      // quadrature nodes s = q, q+4, q+8, ... of its row
      const double* rp = (a.rowp != nullptr && RP > 0) ? a.rowp + (size_t)nc * RP : nullptr;
      double cm = 0.0, cv = 0.0;
      {
        // NB nodes in flight per lane (independent dependency chains); every lane runs the same trip count
        // (cross-lane sums inside the reverse sweep), out-of-range nodes and padding rows carry weight 0
        constexpr int NB = MODE == 1 ? TGP_NODES_IN_FLIGHT : 1;
        double* accq = acc + wave * 16 + nl;
        double* accr = acc + (size_t)P * 64 + tid;
        const int ntrip = (p.S + 4 * NB - 1) / (4 * NB);
        lds_barrier();  // the stack aliases the operand panels: all waves must be done with GEMM 2
        for (int it = 0; it < ntrip; ++it) {
          double xn[NB], wq[NB], f[NB], c[NB];
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const int sn = q + 4 * (NB * it + u);
            xn[u] = xsL[sn < p.S ? sn : 0];
            wq[u] = (valid && sn < p.S) ? wnL[sn] : 0.0;
            f[u] = mu + sq * xn[u];
          }
          flow_forward_store<NB>(F, f, rp, stack + tid, 256);
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const double r = y - f[u];
            ellp += wq[u] * (-0.5 * TGP_LOG_2PI_REF - 0.5 * eta - 0.5 * einv * r * r);
            etap += wq[u] * (-0.5 + 0.5 * einv * r * r);
            c[u] = a.scale * einv * wq[u] * r;
          }
          flow_backward_store<NB>(F, c, rp, stack + tid, 256, a.prog.nslots, accq, 64, q == 0, accr, 256);
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            cm += c[u];
            cv += c[u] * xn[u];
          }
        }
      }
      mub = quad_sum(cm);
      vb = quad_sum(cv) / sq;
      if (!valid) { mub = 0.0; vb = 0.0; ellp = 0.0; etap = 0.0; }
    }
  }
  if (p.lik != TGP_LIK_FLOW && !valid) { mub = 0.0; vb = 0.0; ellp = 0.0; etap = 0.0; }

  ROW_STAMP(a.ws, p, 6);
  // ---- Abar = m mubar^T - 2 A vbar + 2 Lq (B vbar) ;  Kbar = L^-T Abar ----
#pragma unroll
  for (int i = 0; i < MT; ++i) Ba[i] *= vb;
  d4 Ca[MT];
  lds_barrier();  // every wave is done with the flow stack / forward panels before the region is overwritten
  issue2(0, stg[0]);
  issue2(1, stg[1]);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const double* buf = pan + (i & 1) * (MP * 16);
    commit(i & 1, true, i, stg[i & 1]);
    lds_barrier();
    if (i + 2 < 2 * MT) issue2(i + 2, stg[i & 1]);
    Ca[i] = mfma_chain<4 * MT>(buf + q * 16 + nl, 0, 4 * (i + 1), [&](int st) { return Ba[st / 4][st % 4]; });
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) Ca[i][r] = mv[16 * i + 4 * r + q] * mub - 2.0 * Aa[i][r] * vb + 2.0 * Ca[i][r];
  // back substitution, last tile first: Kbar_i = Dinv_i^T (Abar_i - sum_{kb > i} L[kb,i]^T Kbar_kb), the finished tiles
  // taken in the order kb = MT-1 .. i+1 so that the freshest one is the last operand
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int i = MT - 1 - t, pp = MT + t;
    const double* buf = pan + (pp & 1) * (MP * 16);
    commit(pp & 1, false, i, stg[pp & 1]);
    lds_barrier();
    if (pp + 2 < 2 * MT) issue2(pp + 2, stg[pp & 1]);
    Ba[i] = subst_chain<4 * MT>(buf + q * 16 + nl, 4 * (MT - 1 - i), 4 * i, -Ca[i],
                                [&](int st) { return 4 * (MT - 1 - st / 4) + st % 4; },
                                [&](int st) { return Ba[MT - 1 - st / 4][st % 4]; });  // Kbar
  }
  lds_barrier();  // panels dead: the region becomes the transposition tile

  ROW_STAMP(a.ws, p, 7);
  double* slab = a.ws + p.slabs + (size_t)bid * p.slab_len;
  const int col = wave * RW + nl;   // this lane's data row among the workgroup's RBK statistics columns (nl < RW)

  // ---- phase 1: E = Kbar o K through LDS (transposed), T = E [xs, xs^2, 1] on MFMA ----
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (RW == 16 || nl < RW) tile[(16 * i + 4 * r + q) * LD + col] = Ba[i][r] * Kr[4 * i + r];
#pragma unroll
  for (int c = 0; c < CT16; ++c) {
    if ((c & 3) == q && (RW == 16 || nl < RW)) {
      double val = 0.0;
      if (c < DP) val = x[c < DP ? c : 0];
      else if (c < 2 * DP) val = x[(c - DP) < DP ? (c - DP) : 0] * x[(c - DP) < DP ? (c - DP) : 0];
      else if (c == 2 * DP) val = 1.0;
      xt[col * CT16 + c] = val;
    }
  }
  lds_barrier();
  for (int t = wave; t < MT * CT; t += 4) {
    const int ti = t / CT, tc = t % CT;
    d4 c = {0, 0, 0, 0};
    c = tile_mm_f([&](int k) { return tile[(16 * ti + nl) * LD + k + q]; },
                  [&](int k) { return xt[(k + q) * CT16 + 16 * tc + nl]; }, 0, RBK, c);
#pragma unroll
    for (int r = 0; r < 4; ++r) st_wt(&slab[p.slab_T + (size_t)(16 * ti + q + 4 * r) * CT16 + 16 * tc + nl], c[r]);
  }
  lds_barrier();

  ROW_STAMP(a.ws, p, 8);
  // ---- phase 2: A through LDS, G = A diag(vbar) A^T (lower tiles), s = A mubar ----
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (RW == 16 || nl < RW) tile[(16 * i + 4 * r + q) * LD + col] = Aa[i][r];
  if (q == 0 && (RW == 16 || nl < RW)) { vbs[col] = vb; mbs[col] = mub; }
  lds_barrier();
  ROW_STAMP(a.ws, p, 17);
  // Row-blocks of G are dealt to the waves in balanced groups -- MT odd: {MT-1}, {MT-2, 0}, {MT-3, 1}, ...; MT even:
  // {MT-1, 0}, {MT-2, 1}, ... (every group holds MT or MT+1 of the MT(MT+1)/2 lower tiles) -- so that the 16 A-operand
  // fragments of a row-block are read from LDS ONCE, kept in registers (first raw for s = A mubar, then scaled by vbar for
  // the G tiles of that row) and only the B operand streams from LDS: with all operands re-read per MFMA the four waves
  // saturated the LDS (61 ns per MFMA measured instead of 27).
  {
    const int ngroups = (MT + 1) / 2;
    if (wave < ngroups) {
      int rows[2], nrows;
      if (MT & 1) {
        if (wave == 0) { rows[0] = MT - 1; nrows = 1; }
        else { rows[0] = MT - 1 - wave; rows[1] = wave - 1; nrows = 2; }
      } else {
        rows[0] = MT - 1 - wave; rows[1] = wave; nrows = 2;
      }
      for (int g = 0; g < nrows; ++g) {
        const int ti = rows[g];
        double af[RW];
#pragma unroll
        for (int nk = 0; nk < RW; ++nk) af[nk] = tile[(16 * ti + nl) * LD + 4 * nk + q];
        {  // s = A mubar (B operand = mubar in column 0)
          d4 c = {0, 0, 0, 0};
          double bm[RW];
#pragma unroll
          for (int nk = 0; nk < RW; ++nk) bm[nk] = nl == 0 ? mbs[4 * nk + q] : 0.0;
#pragma unroll
          for (int nk = 0; nk < RW; ++nk) c = TGP_MFMA(af[nk], bm[nk], c);
          if (nl == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st_wt(&slab[p.slab_S + 16 * ti + q + 4 * r], c[r]);
          }
        }
#pragma unroll
        for (int nk = 0; nk < RW; ++nk) af[nk] *= vbs[4 * nk + q];
        for (int tj = 0; tj <= ti; ++tj) {
          const int t = ti * (ti + 1) / 2 + tj;
          d4 c = {0, 0, 0, 0};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            constexpr int HB = (RW + 1) / 2;   // k-steps per batch (two batches cover the RW k-steps of the workgroup's columns)
            double bv[HB];
#pragma unroll
            for (int u = 0; u < HB; ++u)
              if (HB * h + u < RW) bv[u] = tile[(16 * tj + nl) * LD + 4 * (HB * h + u) + q];
#pragma unroll
            for (int u = 0; u < HB; ++u)
              if (HB * h + u < RW) c = TGP_MFMA(af[HB * h + u], bv[u], c);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) st_wt(&slab[p.slab_G + (size_t)t * 256 + (q + 4 * r) * 16 + nl], c[r]);
        }
      }
    }
  }
  ROW_STAMP(a.ws, p, 18);

  ROW_STAMP(a.ws, p, 9);
  // ---- scalars, flow parameter gradients ----
  const double e1 = wave_sum(ellp), e2 = wave_sum(etap), e3 = wave_sum(q == 0 ? vb : 0.0);
  if (lane == 0) { red[wave * 4] = e1; red[wave * 4 + 1] = e2; red[wave * 4 + 2] = e3; }
  lds_barrier();
  if (tid == 0) {
    st_wt(&slab[p.slab_C + C_ELL], a.scale * (red[0] + red[4] + red[8] + red[12]));
    st_wt(&slab[p.slab_C + C_ETAB], a.scale * (red[1] + red[5] + red[9] + red[13]));
    st_wt(&slab[p.slab_C + C_SVB], red[2] + red[6] + red[10] + red[14]);
    st_wt(&slab[p.slab_C + C_PAD], 0.0);
  }
  for (int j = wave; j < P; j += 4) {
    const double s = wave_sum(acc[j * 64 + lane]);
    if (lane == 0) st_wt(&slab[p.slab_C + C_THETA + j], s);
  }
  ROW_STAMP(a.ws, p, 10);
#ifdef TGP_STAMPS
  if (bid == 0 && threadIdx.x == 0) a.ws[p.hdr + H_STAMP + 23] = (double)clock64();
#endif
  for (size_t i = p.slab_C + C_THETA + P + tid; i < p.slab_len; i += 256) st_wt(&slab[i], 0.0);
  if (RW == 16 && a.g_rowp != nullptr && q == 0 && valid) {
    const double* rbase = acc + (size_t)P * 64;
    for (int jr = 0; jr < RP; ++jr) {
      const double* ap = rbase + (size_t)jr * 256 + wave * 64 + nl;
      a.g_rowp[(size_t)n * RP + jr] = ap[0] + ap[16] + ap[32] + ap[48];
    }
  }
}

}  // namespace tgp
