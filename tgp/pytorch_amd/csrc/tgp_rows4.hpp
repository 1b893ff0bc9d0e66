// tgp_rows4.hpp -- the row kernel at FOUR data rows per wave, on v_mfma_f64_4x4x4_4b (round 5).
//
// Same algebra, same references as tgp_rows.hpp (models/sparse_MF_SP.py:313-396: K_NM, the two solves with L, the
// products with L_q, mu / v; likelihoods/GaussianLinearMean.py:60-87, GaussianNonLinearMean.py:64-150, models/flow.py; the
// autograd replay of all of it, trainers/trainer_base.py:341) -- what changes is the work granularity.  k_rows gives a
// wave 16 rows (the N of v_mfma_f64_16x16x4): 539 waves at Power size for 1 024 SIMDs, one wave per SIMD at 378
// registers, and its chains cost the same whether 16 or 9 of the columns carry rows.  Here a wave owns 4 rows and the four
// BLOCKS of the 4x4x4 form are four 4-row chunks of the M dimension, so the matrix work of a wave is proportional to its
// rows and 2 to 3 waves share a SIMD (K, A, B of a wave are 3 MT doubles per lane, not 12 MT):
//   lane l = 16 i + 4 b + j :  j = data row of the wave, b = block, i = row inside the 4 x 4 chunk
//   an M x 4 column strip X lives in MT registers: register t, lane (i, b, j) = X[m = 16 t + 4 b + i][row j]
//   (the D layout of the instruction, probed in tools/probes/mfma_4x4_rate.hip: A operand lane 16 k + 4 b + i,
//    B operand lane 16 k + 4 b + j, D lane 16 i + 4 b + j -- so a result chunk IS a B operand of its own block)
//   out[t'] += Op[rows of group t'][cols of chunk kc] * X[chunk kc] is ONE instruction: the A operand is the 16 x 4 strip
//   of the operator, the B operand is chunk kc of X REPLICATED in the four blocks (a broadcast through a 512-byte
//   per-wave LDS scratch: one write per register, one read per chunk).
// The substitutions run RIGHT-looking (a finished group of 16 rows updates every right-hand side below it: independent
// accumulators -- the 4x4x4 form issues every 18 cycles with >= 4 of them against 52 as one chain).  The operand of a
// phase (a triangular matrix, MT (MT + 1) / 2 tiles of 2 KB) is staged WHOLE by the workgroup, so inside a phase the
// waves run free of each other (the first version walked 16-column panels in lockstep, one barrier per panel: every wave
// of a SIMD sat in its LDS round trips and its 4-instruction closing chain at the same time -- 1 us per panel step, the
// matrix pipe idle for 45 % of it, profiles/r05_rows4_prototype.txt).
#pragma once
#include "tgp_rows.hpp"

namespace tgp {

#define TGP_MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)
#define TGP_R4_NB 2 /* quadrature nodes in flight per lane when the stack fits (16 lanes per data row: one trip = 32 nodes); else 1 */

// LDS carve-up (offsets in doubles).  `region` is used in turn by: the staged operand image of a phase ([tile][kp = chunk
// pair][lane][2]: the A operands of chunks 2 kp and 2 kp + 1 of a lane side by side -- ONE conflict-free ds_read_b128
// feeds two instructions) with the per-wave broadcast scratch and the scaled inducing points behind it, the flow stack
// (nothing of the image phases survives it: the scratch is transient, Zs is only read for the K strip), and the
// transposition tiles of the row statistics.
struct Row4Lds {
  size_t ils, mv, tp, tg, ti, xs, wn, prog, red, acc, region, bc, zs, tileA, xt, vbs, mbs, total;
  int LD;   // row stride of the [m][4 NW data rows] transposition tiles: 2 x odd -> conflict-free fragment reads
  int nb;   // quadrature nodes in flight per lane (2, or 1 when two do not fit the LDS), 0: not even one fits
};
__host__ __device__ inline Row4Lds row4_lds(const Plan& p, bool train, int nslots, int nw) {
  Row4Lds L;
  size_t o = 0;
  auto take = [&o](size_t n) { size_t r = o; o += (n + 1) / 2 * 2; return r; };
  L.ils = take(16);
  L.mv = take(p.MP);
  L.tp = take(p.P + 1);
  L.tg = take(p.P + 1);
  L.ti = take(p.P + 1);
  L.xs = take(p.S + 1);
  L.wn = take(p.S + 1);
  L.prog = take((size_t)2 * p.nblk + 2);
  L.red = take((size_t)4 * nw + 8);
  L.LD = 4 * nw + 2;
  L.nb = TGP_R4_NB;
  const size_t imgsz = (size_t)2 * p.ntri * 256;   // TWO operand images: the two phases of a pass run without a barrier between them
  size_t reg = imgsz + (size_t)nw * 64 + (size_t)p.MP * p.DP;
  L.acc = o;
  if (train) {
    L.acc = take((size_t)(p.P > 0 ? p.P : 1) * nw + (size_t)p.RP * nw * 64);   // [P][NW] wave-reduced + [RP][NT] per lane
    const size_t stats = (size_t)2 * p.MP * L.LD + (size_t)4 * nw * p.CT16 + (size_t)8 * nw;
    if (stats > reg) reg = stats;
    const size_t lim = (160 * 1024 - 2048) / sizeof(double);
    const size_t slot = (size_t)(nslots > 0 ? nslots : 1) * nw * 64;
    if (p.lik == TGP_LIK_FLOW) {
      if (o + (size_t)TGP_R4_NB * slot > lim) L.nb = 1;
      if (o + (size_t)L.nb * slot > lim) L.nb = 0;
      if ((size_t)L.nb * slot > reg) reg = (size_t)L.nb * slot;
    }
  }
  L.region = take(reg);
  L.bc = L.region + imgsz;
  L.zs = L.bc + (size_t)nw * 64;
  L.tileA = L.region + (size_t)p.MP * L.LD;
  L.xt = L.tileA + (size_t)p.MP * L.LD;
  L.vbs = L.xt + (size_t)4 * nw * p.CT16;
  L.mbs = L.vbs + (size_t)4 * nw;
  L.total = o;
  return L;
}

// sum over the 16 lanes that share a data row j = l & 3 (bits 2..5 of the lane index): every lane gets the total
__device__ __forceinline__ double row4_sum(double x) {
  x = xor_sum16(xor_sum32(x));          // bits 5, 4
  return ror_sum<4>(ror_sum<8>(x));     // bits 3, 2 (rotations inside the row of 16 lanes)
}

#ifdef TGP_STAMPS
#define R4_STAMP(i)                                                                                          \
  do {                                                                                                       \
    if (blockIdx.x == 0 && lane == 0 && wave < 12) a.ws[a.p.dbg + wave * 20 + (i)] = (double)__builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define R4_STAMP(i) \
  do {              \
  } while (0)
#endif

// TRAIN = false: moments only (tgp_qf_moments_f64).  NW = waves per workgroup (4 NW data rows; Plan.nblocks = ceil(N / 4 NW)).
template <int MT, int DP, bool TRAIN, int NW>
__global__ __launch_bounds__(NW * 64) void k_rows4(RowArgs a) {
  constexpr int MP = MT * 16, NT = NW * 64, NTRI = MT * (MT + 1) / 2, NIMG = NTRI * 256;
  constexpr int CT = (2 * DP + 1 + 15) / 16, CT16 = CT * 16, RB = 4 * NW;
  typedef double d2v __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const Plan& p = a.p;
  const Row4Lds L = row4_lds(p, TRAIN, a.prog.nslots, NW);
  double* zs = sm + L.zs;   // (inside `region`, behind the operand image)
  double* ils = sm + L.ils;
  double* mv = sm + L.mv;
  double* tpL = sm + L.tp;
  double* tgL = sm + L.tg;
  double* tiL = sm + L.ti;
  double* xsL = sm + L.xs;
  double* wnL = sm + L.wn;
  int32_t* progL = reinterpret_cast<int32_t*>(sm + L.prog);
  double* red = sm + L.red;
  double* acc = sm + L.acc;
  double* img = sm + L.region;
  double* stack = sm + L.region;
  const int LD = L.LD;
  double* tileE = sm + L.region;
  double* tileA = sm + L.tileA;
  double* xt = sm + L.xt;
  double* vbs = sm + L.vbs;
  double* mbs = sm + L.mbs;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* bc = sm + L.bc + wave * 64;   // (inside `region`, behind the operand image)
  const int j = lane & 3, kq = lane >> 4, l16 = lane >> 2;   // data row; k / i index; lane among the 16 of its data row
  const double* __restrict__ ws = a.ws;
  const int N = p.N, D = p.D, M = p.M, P = p.P, RP = p.RP;
  const int bid = blockIdx.x;

  if (TRAIN && bid >= p.nblocks) {
    // ---- passenger blocks (one per 16-column block c of J = L^-1): J, H'^T = (J^T (S - I))^T, w = J^T m for the backward
    //      M x M chain, on v_mfma_f64_16x16x4 like k_rows' (tgp_rows.hpp) but with the L fragments of a tile row
    //      requested when that row is due, not all up front (this kernel runs at <= 168 registers; the passengers have
    //      tens of microseconds to spare).  Waves 0..3 work, the others leave.
    if (wave >= 4) return;
    const int c = __builtin_amdgcn_readfirstlane(bid - p.nblocks);
    const int nl = lane & 15, q = lane >> 4;
    const double* __restrict__ LTg = ws + p.LT;
    const double* __restrict__ nDg = ws + p.nD;
    const double* __restrict__ Sg = ws + p.S_;
    d4 Jt[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) Jt[i] = d4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i < c) continue;
      if (i == c) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Jt[i][rr] = -nDg[c * 256 + (4 * rr + q) * 16 + nl];
        continue;
      }
      double df[4], lf[MT][4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) df[s4] = nDg[i * 256 + nl * 16 + 4 * s4 + q];
#pragma unroll
      for (int kb = 0; kb < MT; ++kb) {
        if (kb < c || kb >= i) continue;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) lf[kb][s4] = LTg[(size_t)(16 * kb + 4 * s4 + q) * MP + 16 * i + nl];
      }
      d4 ac = {0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < MT; ++kb) {
        if (kb < c || kb >= i) continue;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) ac = TGP_MFMA(lf[kb][s4], Jt[kb][s4], ac);
      }
      d4 o = {0, 0, 0, 0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) o = TGP_MFMA(df[s4], ac[s4], o);
      Jt[i] = o;
    }
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i < c) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a.ws[p.J + (size_t)(16 * i + 4 * rr + q) * MP + 16 * c + nl] = Jt[i][rr];
      }
    }
    for (int jj = wave; jj < MT; jj += 4) {
      d4 h = {0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < MT; ++kb) {
        if (kb < c) continue;
        double bf[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int kk = 16 * kb + 4 * s4 + q;
          bf[s4] = Sg[(size_t)kk * MP + 16 * jj + nl] - (kk == 16 * jj + nl ? 1.0 : 0.0);
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) h = TGP_MFMA(Jt[kb][s4], bf[s4], h);
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) a.ws[p.HpT + (size_t)(16 * jj + nl) * MP + 16 * c + 4 * rr + q] = h[rr];
    }
    if (wave == 3) {
      double sw = 0.0;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i < c) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) sw += Jt[i][rr] * ws[p.mpad + 16 * i + 4 * rr + q];
      }
      sw = quad_sum(sw);
      if (q == 0) a.ws[p.w + 16 * c + nl] = sw;
    }
    return;
  }

  R4_STAMP(0);
  const int n = (bid * NW + wave) * 4 + j;
  const bool valid = n < N;
  const int nc = valid ? n : N - 1;
  double xraw[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) xraw[d] = d < D ? a.X[(size_t)nc * D + d] : 0.0;
  const double y = TRAIN ? a.Y[nc] : 0.0;
  const double s2 = ws[p.hdr + H_S2], eta = ws[p.hdr + H_ETA], einv = ws[p.hdr + H_EINV];

  // ---- operand images.  Tile (tt, t) of a phase = rows of group tt, columns of group t of its operator:
  //        kind 0  A = L^-1 K       tt >= t : L[tt][t], and -Dinv_t on the diagonal            index col_off(t) + tt - t
  //        kind 1  B = Lq^T A       tt <= t : Lq^T[tt][t]                                        index t (t + 1) / 2 + tt
  //        kind 2  C = Lq (B vbar)  tt >= t : Lq[tt][t]                                          index col_off(t) + tt - t
  //        kind 3  Kbar = L^-T Abar tt <= t : L^T[tt][t], and (-Dinv_t)^T on the diagonal        index t (t + 1) / 2 + tt
  //      In LDS a tile is [kp][lane][2]: element (row rr, column c = 4 kc + k) at kp = kc >> 1, lane 16 k + rr, kc & 1.
  const double* __restrict__ Lm = ws + p.L;
  const double* __restrict__ LTm = ws + p.LT;
  const double* __restrict__ nD = ws + p.nD;
  const double* __restrict__ Lq = ws + p.Lq;
  const double* __restrict__ LqT = ws + p.LqT;
  auto col_off = [](int t) { return t * MT - t * (t - 1) / 2; };
  // Staging in 16-byte loads: 128 threads per tile (row rr, column pair 2 cp), so a wave-instruction covers 8 rows x 128 B and
  // an image is half the load instructions of the 8-byte version -- what a CU pulls per microsecond is set by its requests in
  // flight, not by bytes.  The tile index, its (tt, t) and every base address are wave-uniform (scalar registers); a lane adds
  // constants of its own.  (The transposed diagonal tiles of kind 3 are two 8-byte loads: their pair is 16 doubles apart.)
  constexpr int TPW = NT / 128, NSTG = (NTRI + TPW - 1) / TPW;   // tiles per step of the workgroup; steps per image
  const int tw = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int srr = (tid >> 3) & 15, sc = 2 * (tid & 7);          // row, first column of the pair
  const int s_reg = srr * MP + sc, s_d0 = srr * 16 + sc, s_d3 = sc * 16 + srr;
  const int s_dst0 = (sc >> 3) * 128 + (16 * (sc & 3) + srr) * 2 + ((sc >> 2) & 1);                   // column sc
  const int s_dst1 = ((sc + 1) >> 3) * 128 + (16 * ((sc + 1) & 3) + srr) * 2 + (((sc + 1) >> 2) & 1);   // column sc + 1
  auto ld2 = [](const double* ptr) { return *reinterpret_cast<const d2v*>(ptr); };
  auto issue = [&](int kind, d2v (&st)[NSTG]) {
    const bool up = kind == 0 || kind == 2;   // tiles (tt >= t, t)
#pragma unroll
    for (int u = 0; u < NSTG; ++u) {
      const int idx = tw + TPW * u;
      if (idx < NTRI) {
        int t = 0;
#pragma unroll
        for (int v = 1; v < MT; ++v) t += idx >= (up ? col_off(v) : v * (v + 1) / 2) ? 1 : 0;
        const int tt = up ? t + idx - col_off(t) : idx - t * (t + 1) / 2;
        const int base = 16 * tt * MP + 16 * t;
        d2v val;
        if (kind == 0) val = tt == t ? ld2(nD + t * 256 + s_d0) : ld2(Lm + base + s_reg);
        else if (kind == 1) val = ld2(LqT + base + s_reg);
        else if (kind == 2) val = ld2(Lq + base + s_reg);
        else if (tt == t) { val.x = nD[t * 256 + s_d3]; val.y = nD[t * 256 + s_d3 + 16]; }
        else val = ld2(LTm + base + s_reg);
        st[u] = val;
      }
    }
  };
  auto commit = [&](double* im, const d2v (&st)[NSTG]) {
#pragma unroll
    for (int u = 0; u < NSTG; ++u) {
      const int idx = tw + TPW * u;
      if (idx < NTRI) { im[idx * 256 + s_dst0] = st[u].x; im[idx * 256 + s_dst1] = st[u].y; }
    }
  };
  double* imgA = img;
  double* imgB = img + NIMG;
  d2v stgA[NSTG], stgB[NSTG];
  issue(0, stgA);
  issue(1, stgB);         // (both forward images are in flight from the first instruction: a CU pulls a 57 KB image from the other
                          //  XCDs' L2 in 5-7 us -- its outstanding-miss budget, not bandwidth -- so every image is requested one
                          //  phase before its commit and lands under the phase in between)

  // ---- stage the small shared operands: every load requested before the first LDS store ----
  {
    const bool fl = TRAIN && p.lik == TGP_LIK_FLOW;
    constexpr int NZ = (MP * DP + NT - 1) / NT;
    double zv[NZ];
#pragma unroll
    for (int u = 0; u < NZ; ++u) zv[u] = tid + NT * u < MP * DP ? ws[p.Zs + tid + NT * u] : 0.0;
    const double mv0 = tid < MP ? ws[p.mpad + tid] : 0.0;
    const double il0 = tid < 16 ? ws[p.ils + tid] : 0.0;
    const double tp0 = (fl && tid < P) ? ws[p.tp + tid] : 1.0, tg0 = (fl && tid < P) ? ws[p.tg + tid] : 0.0;
    const double xs0 = (fl && tid < p.S) ? a.xs[tid] : 0.0, wn0 = (fl && tid < p.S) ? a.wn[tid] : 0.0;
#pragma unroll
    for (int u = 0; u < NZ; ++u)
      if (tid + NT * u < MP * DP) zs[tid + NT * u] = zv[u];
    if (tid < MP) mv[tid] = mv0;
    if (tid < 16) ils[tid] = il0;
    if (fl) {
      if (tid < P) { tpL[tid] = tp0; tgL[tid] = tg0; tiL[tid] = rcp_fast(tp0); }
      if (tid < p.S) { xsL[tid] = xs0; wnL[tid] = wn0; }
      for (int i = tid + NT; i < P; i += NT) { const double t0 = ws[p.tp + i]; tpL[i] = t0; tgL[i] = ws[p.tg + i]; tiL[i] = rcp_fast(t0); }
      for (int i = tid + NT; i < p.S; i += NT) { xsL[i] = a.xs[i]; wnL[i] = a.wn[i]; }
      for (int i = tid; i < 4 * p.nblk; i += NT) progL[i] = a.prog.blk[i];
    }
    if (TRAIN) {
      const int nacc = P * NW + RP * NT;
      for (int i = tid; i < nacc; i += NT) acc[i] = 0.0;
    }
  }
  commit(imgA, stgA);
  lds_barrier();
  R4_STAMP(1);

  double x[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) x[d] = d < D ? xraw[d] * ils[d] : 0.0;
  const int mrow = 4 * ((lane >> 2) & 3) + kq;   // 4 b + i: this lane's row inside a group of 16 in the D layout

  // ---- K strip: Kr[t] = K[m = 16 t + 4 b + i][row j] ----
  double Kr[MT];
  {
    double e[MT];
    TGP_EACH(t, MT) {
      const int mm = 16 * t + mrow;
      double d2 = 0.0;
#pragma unroll
      for (int d = 0; d < DP; ++d) {
        const double tt = x[d] - zs[mm * DP + d];
        d2 += tt * tt;
      }
      e[t] = -0.5 * d2;
    }
    exp_fast_n<MT>(e);
    TGP_EACH(t, MT) Kr[t] = (16 * t + mrow < M ? s2 : 0.0) * e[t];
  }
  R4_STAMP(2);

  // operand fetchers.  A operands of tile idx, chunks 2 kp and 2 kp + 1, for this lane (k, b, i_m): one 16-byte read.
  // Broadcast of chunk kc of a register through the scratch: lane (k, b, j) reads what lane (i = k, block kc, j) wrote.
  auto aop2 = [&](const double* im, int idx, int kp) { return *reinterpret_cast<const d2v*>(im + idx * 256 + kp * 128 + lane * 2); };
  auto bcast = [&](double v, double (&bk)[4]) {
    __builtin_amdgcn_wave_barrier();
    bc[lane] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    TGP_EACH(kc, 4) bk[kc] = bc[(lane & 0x30) + 4 * kc + j];
  };
  // out = Dblock * c with Dblock = the diagonal tile idx of the staged image (the close of a substitution group)
  auto close_group = [&](const double* im, int idx, double c) {
    double bk[4];
    bcast(c, bk);
    const d2v d0 = aop2(im, idx, 0), d1 = aop2(im, idx, 1);
    double o = 0.0;
    o = TGP_MFMA4(d0.x, bk[0], o); o = TGP_MFMA4(d0.y, bk[1], o);
    o = TGP_MFMA4(d1.x, bk[2], o); o = TGP_MFMA4(d1.y, bk[3], o);
    return o;
  };
  // out[tt] += tile(tt, t) * src for tt in [lo, hi): idx0 = index of tile (lo, t)
  auto push_group = [&](const double* im, double src, double (&out)[MT], int lo, int hi, int idx0) {
    double bk[4];
    bcast(src, bk);
    TGP_EACH(kp, 2) {
      d2v av[MT];
#pragma unroll
      for (int tt = 0; tt < MT; ++tt)
        if (tt >= lo && tt < hi) av[tt] = aop2(im, idx0 + tt - lo, kp);
#pragma unroll
      for (int tt = 0; tt < MT; ++tt)
        if (tt >= lo && tt < hi) out[tt] = TGP_MFMA4(av[tt].x, bk[2 * kp], out[tt]);
#pragma unroll
      for (int tt = 0; tt < MT; ++tt)
        if (tt >= lo && tt < hi) out[tt] = TGP_MFMA4(av[tt].y, bk[2 * kp + 1], out[tt]);
    }
  };

  // ---- A = L^-1 K, right-looking: group t closes with -Dinv_t (rhs = -K + sum L A), then updates every group below it ----
  double Aa[MT], Ba[MT];
  {
    double rhs[MT];
    TGP_EACH(t, MT) rhs[t] = -Kr[t];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      Aa[t] = close_group(imgA, col_off(t), rhs[t]);
      if (t + 1 < MT) push_group(imgA, Aa[t], rhs, t + 1, MT, col_off(t) + 1);
    }
  }
  R4_STAMP(3);
  commit(imgB, stgB);     // (its own buffer: nobody reads it yet; one barrier makes it visible)
  if constexpr (TRAIN) {  // both backward images: requested now, they land under B = Lq^T A and the quadrature
    issue(2, stgA);
    issue(3, stgB);
  }
  lds_barrier();
  // ---- B = Lq^T A as outer products: source group t updates B[tt], tt <= t ----
  TGP_EACH(t, MT) Ba[t] = 0.0;
#pragma unroll
  for (int t = 0; t < MT; ++t) push_group(imgB, Aa[t], Ba, 0, t + 1, t * (t + 1) / 2);
  R4_STAMP(4);
  // ---- mu, v ----
  double pm = 0.0, pa = 0.0, pb = 0.0;
  TGP_EACH(t, MT) {
    pm += mv[16 * t + mrow] * Aa[t];
    pa += Aa[t] * Aa[t];
    pb += Ba[t] * Ba[t];
  }
  pm = row4_sum(pm); pa = row4_sum(pa); pb = row4_sum(pb);
  const double mu = pm, v = s2 - pa + pb;
  if (a.mu != nullptr && lane < 4 && valid) { a.mu[n] = mu; a.v[n] = v; }
  R4_STAMP(5);
  if constexpr (TRAIN) {
    // ---- expected log-likelihood and its adjoints (the 16 lanes of a data row share its quadrature nodes) ----
    double mub = 0.0, vb = 0.0, ellp = 0.0, etap = 0.0;
    lds_barrier();          // the image of Lq^T is dead: the region becomes the flow stack
    if (p.lik == TGP_LIK_GAUSS) {
      // GaussianLinearMean.expected_log_prob (likelihoods/GaussianLinearMean.py:81-87)
      const double r = y - mu;
      mub = a.scale * einv * r;
      vb = -0.5 * a.scale * einv;
      if (lane < 4) {
        ellp = -0.5 * TGP_LOG_2PI_REF - 0.5 * eta - 0.5 * einv * (r * r + v);
        etap = -0.5 + 0.5 * einv * (r * r + v);
      }
    } else if (p.lik == TGP_LIK_ADJOINT) {
      // tgp_qf_moments_bwd_f64: the adjoints of (mu, v) are the caller's (mu_bar in the Y slot, v_bar in the rowp slot)
      mub = y;
      vb = a.rowp[nc];
    } else {
      // GaussianNonLinearMean.expected_log_prob (likelihoods/GaussianNonLinearMean.py:91-148): lane l16 of a data row takes
      // the quadrature nodes l16, l16 + 16, ...
      FlowDev F{progL, p.nblk, tpL, tgL, tiL};
      const double sq = sqrt(2.0 * v);
      const double* rp = (a.rowp != nullptr && RP > 0) ? a.rowp + (size_t)nc * RP : nullptr;
      double cm = 0.0, cv = 0.0;
      double* accq = acc + wave;
      double* accr = acc + (size_t)P * NW + tid;
      auto sweep = [&](auto nbc) {
        constexpr int NB = decltype(nbc)::value;
        const int ntrip = (p.S + 16 * NB - 1) / (16 * NB);
        for (int it = 0; it < ntrip; ++it) {
          double xn[NB], wq[NB], f[NB], c[NB];
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const int sn = l16 + 16 * (NB * it + u);
            xn[u] = xsL[sn < p.S ? sn : 0];
            wq[u] = (valid && sn < p.S) ? wnL[sn] : 0.0;
            f[u] = mu + sq * xn[u];
          }
          flow_forward_store<NB>(F, f, rp, stack + tid, NT);
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const double r = y - f[u];
            ellp += wq[u] * (-0.5 * TGP_LOG_2PI_REF - 0.5 * eta - 0.5 * einv * r * r);
            etap += wq[u] * (-0.5 + 0.5 * einv * r * r);
            c[u] = a.scale * einv * wq[u] * r;
          }
          flow_backward_store<NB, 2>(F, c, rp, stack + tid, NT, a.prog.nslots, accq, NW, lane == 0, accr, NT);
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            cm += c[u];
            cv += c[u] * xn[u];
          }
        }
      };
      if (L.nb == 2) sweep(std::integral_constant<int, 2>{});
      else sweep(std::integral_constant<int, 1>{});
      mub = row4_sum(cm);
      vb = row4_sum(cv) / sq;
    }
    if (!valid) { mub = 0.0; vb = 0.0; ellp = 0.0; etap = 0.0; }
    R4_STAMP(6);

    // ---- Abar = m mubar^T - 2 A vbar + 2 Lq (B vbar) ;  Kbar = L^-T Abar ----
    lds_barrier();          // every wave is done with the flow stack
    commit(imgA, stgA);     // (requested before the quadrature)
    commit(imgB, stgB);
    lds_barrier();
    double Ca[MT];
    {
      TGP_EACH(t, MT) { Ba[t] *= vb; Ca[t] = 0.0; }
#pragma unroll
      for (int t = 0; t < MT; ++t) push_group(imgA, Ba[t], Ca, t, MT, col_off(t));
      // -Abar: the running right-hand side of the back substitution
      TGP_EACH(t, MT) Ca[t] = -(mv[16 * t + mrow] * mub - 2.0 * Aa[t] * vb + 2.0 * Ca[t]);
    }
    R4_STAMP(7);
    // back substitution, last group first: Kbar_t = (-Dinv_t)^T (-Abar_t + sum_{u > t} L[u, t]^T Kbar_u)
#pragma unroll
    for (int t = MT - 1; t >= 0; --t) {
      Ba[t] = close_group(imgB, t * (t + 1) / 2 + t, Ca[t]);                    // Kbar
      if (t > 0) push_group(imgB, Ba[t], Ca, 0, t, t * (t + 1) / 2);
    }
    R4_STAMP(8);
    lds_barrier();          // image dead: the region becomes the two transposition tiles [m][4 NW data rows]

    // ---- row statistics on v_mfma_f64_16x16x4 (k = the data rows of the workgroup): T = (Kbar o K) [xs, xs^2, 1],
    //      G = A diag(vbar) A^T (lower tiles), s = A mubar ----
    double* slab = a.ws + p.slabs + (size_t)bid * p.slab_len;
    const int col = wave * 4 + j;
    TGP_EACH(t, MT) {
      tileE[(16 * t + mrow) * LD + col] = Ba[t] * Kr[t];
      tileA[(16 * t + mrow) * LD + col] = Aa[t];
    }
    for (int c = l16; c < CT16; c += 16) {
      double val = 0.0;
      if (c < DP) val = x[c < DP ? c : 0];
      else if (c < 2 * DP) val = x[(c - DP) < DP ? (c - DP) : 0] * x[(c - DP) < DP ? (c - DP) : 0];
      else if (c == 2 * DP) val = 1.0;
      xt[col * CT16 + c] = val;
    }
    if (lane < 4) { vbs[col] = vb; mbs[col] = mub; }
    lds_barrier();
    {
      const int nl = lane & 15, q = lane >> 4;
      const int nT = MT * CT, nG = NTRI, nitem = nT + nG + MT;
      for (int it = wave; it < nitem; it += NW) {
        d4 c = {0, 0, 0, 0};
        if (it < nT) {
          const int ti = it / CT, tc = it % CT;
          double af[NW], bf[NW];
#pragma unroll
          for (int ks = 0; ks < NW; ++ks) { af[ks] = tileE[(16 * ti + nl) * LD + 4 * ks + q]; bf[ks] = xt[(4 * ks + q) * CT16 + 16 * tc + nl]; }
#pragma unroll
          for (int ks = 0; ks < NW; ++ks) c = TGP_MFMA(af[ks], bf[ks], c);
#pragma unroll
          for (int r = 0; r < 4; ++r) st_wt(&slab[p.slab_T + (size_t)(16 * ti + q + 4 * r) * CT16 + 16 * tc + nl], c[r]);
        } else if (it < nT + nG) {
          const int t = it - nT;
          int ti = 0;
#pragma unroll
          for (int v = 1; v < MT; ++v) ti += t >= v * (v + 1) / 2 ? 1 : 0;
          const int tj = t - ti * (ti + 1) / 2;
          double af[NW], bf[NW];
#pragma unroll
          for (int ks = 0; ks < NW; ++ks) {
            af[ks] = tileA[(16 * ti + nl) * LD + 4 * ks + q] * vbs[4 * ks + q];
            bf[ks] = tileA[(16 * tj + nl) * LD + 4 * ks + q];
          }
#pragma unroll
          for (int ks = 0; ks < NW; ++ks) c = TGP_MFMA(af[ks], bf[ks], c);
#pragma unroll
          for (int r = 0; r < 4; ++r) st_wt(&slab[p.slab_G + (size_t)t * 256 + (q + 4 * r) * 16 + nl], c[r]);
        } else {
          const int ti = it - nT - nG;
          double af[NW], bf[NW];
#pragma unroll
          for (int ks = 0; ks < NW; ++ks) { af[ks] = tileA[(16 * ti + nl) * LD + 4 * ks + q]; bf[ks] = nl == 0 ? mbs[4 * ks + q] : 0.0; }
#pragma unroll
          for (int ks = 0; ks < NW; ++ks) c = TGP_MFMA(af[ks], bf[ks], c);
          if (nl == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st_wt(&slab[p.slab_S + 16 * ti + q + 4 * r], c[r]);
          }
        }
      }
    }
    R4_STAMP(9);

    // ---- scalars, flow parameter gradients ----
    const double e1 = wave_sum(ellp), e2 = wave_sum(etap), e3 = wave_sum(lane < 4 ? vb : 0.0);
    if (lane == 0) { red[wave * 4] = e1; red[wave * 4 + 1] = e2; red[wave * 4 + 2] = e3; }
    lds_barrier();
    if (tid == 0) {
      double s1 = 0.0, s2_ = 0.0, s3 = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { s1 += red[4 * w]; s2_ += red[4 * w + 1]; s3 += red[4 * w + 2]; }
      st_wt(&slab[p.slab_C + C_ELL], a.scale * s1);
      st_wt(&slab[p.slab_C + C_ETAB], a.scale * s2_);
      st_wt(&slab[p.slab_C + C_SVB], s3);
      st_wt(&slab[p.slab_C + C_PAD], 0.0);
    }
    for (int jp = tid; jp < P; jp += NT) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += acc[jp * NW + w];
      st_wt(&slab[p.slab_C + C_THETA + jp], s);
    }
    for (size_t i = p.slab_C + C_THETA + P + tid; i < p.slab_len; i += NT) st_wt(&slab[i], 0.0);
    if (a.g_rowp != nullptr && RP > 0) {
      const double* rbase = acc + (size_t)P * NW + tid;
      for (int jr = 0; jr < RP; ++jr) {
        const double s = row4_sum(rbase[(size_t)jr * NT]);
        if (lane < 4 && valid) a.g_rowp[(size_t)n * RP + jr] = s;
      }
    }
    R4_STAMP(10);
  }
}

}  // namespace tgp
