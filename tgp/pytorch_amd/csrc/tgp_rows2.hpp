// tgp_rows2.hpp -- the team-split row kernel: the same mathematics as tgp_rows.hpp (forward and backward of everything
// that scales with the number of rows, see there for the reference call sites: models/sparse_MF_SP.py:313-396,
// likelihoods/*.py, models/flow.py and their autograd replay), laid out so that a Power-sized batch (8611 rows)
// occupies ALL 256 CUs instead of 135.
//
// tgp_rows.hpp gives one wave a 16-row group end to end: 539 groups = 135 workgroups of 4 waves, 121 CUs idle, and every
// group is one serial stream of ~600 MFMAs + the flow quadrature.  Here a group belongs to a TEAM of 4 waves:
//   * the output tiles of each of the four triangular GEMMs are dealt to the team's waves (balanced by tile cost: a
//     lower-type tile i costs i+1 panels, an upper-type tile MT-i), each wave keeps its own tiles in registers and
//     publishes them to a [MP][16] LDS region, from which the next product reads its B operands directly (the
//     accumulator layout of v_mfma_f64_16x16x4 IS the B-operand layout, so the region is plain row-major [m][row]);
//   * the A operands (16-column panels of J^T, Lq, Lq^T, J) are used by exactly one wave of a team, so they are read
//     straight from L2 into registers, all k-steps of a tile requested before its first MFMA;
//   * the Gauss-Hermite nodes of a row are dealt to 16 lanes (4 lanes x 4 waves) instead of 4;
//   * a workgroup holds T = ceil(#groups / 256) <= 3 teams (12 waves, 3 per SIMD: the f64 pipe of a SIMD is kept busy
//     by three short dependency chains instead of one long one), the row statistics G, T, s are formed over all 16 T
//     rows of the workgroup and leave as ONE slab per workgroup (<= 256 slabs).
// Every reduction keeps a fixed order: results are bit-reproducible run to run, like the one-wave-per-group kernel's.
#pragma once
#include "tgp_rows.hpp"

namespace tgp {

// stamps inside the wave roles (diagnostic build only): the role functions see the workspace through these
#ifdef TGP_STAMPS
#define TGP_STAMP_R2(i)                                                                                    \
  do {                                                                                                     \
    if (blockIdx.x == 0 && threadIdx.x == 0) c.stamp[(i)] = (double)__builtin_amdgcn_s_memrealtime();     \
  } while (0)
#else
#define TGP_STAMP_R2(i) \
  do {                  \
  } while (0)
#endif

#ifndef TGP_R2_CHAIN_PRIO
#define TGP_R2_CHAIN_PRIO 0
#endif
#ifndef TGP_R2_PREFETCH
#define TGP_R2_PREFETCH 16 /* A fragments requested one product ahead */
#endif

// tile -> (wave, register slot) of a team: longest-processing-time assignment of the lower-type tiles (cost i + 1);
// the upper-type tile i (cost MT - i) goes to the owner of lower tile MT-1-i, so every wave does the same work in
// all four products.  MT = 7: {6}, {5,0}, {4,1}, {3,2}  /  upper {0}, {1,6}, {2,5}, {3,4}.
template <int MT>
struct TeamOwn {
  int owner[MT], slot[MT], nslot;
  int off[MT];  // first k-step of lower tile i in its owner's fragment array (tiles in slot order); upper tile i: off[MT-1-i]
  int nmax;     // k-steps of the busiest wave
};
template <int MT>
__host__ __device__ constexpr TeamOwn<MT> team_own() {
  TeamOwn<MT> o{};
  int load[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
  for (int i = MT - 1; i >= 0; --i) {
    int w = 0;
    for (int c = 1; c < 4; ++c)
      if (load[c] < load[w]) w = c;
    o.owner[i] = w;
    o.slot[i] = cnt[w]++;
    load[w] += i + 1;
  }
  o.nslot = 0;
  o.nmax = 0;
  for (int c = 0; c < 4; ++c) {
    if (cnt[c] > o.nslot) o.nslot = cnt[c];
    if (4 * load[c] > o.nmax) o.nmax = 4 * load[c];
  }
  for (int i = 0; i < MT; ++i) {
    o.off[i] = 0;
    for (int j = 0; j < MT; ++j)
      if (o.owner[j] == o.owner[i] && o.slot[j] < o.slot[i]) o.off[i] += 4 * (j + 1);
  }
  return o;
}
template <int MT>
struct TeamOwnC {
  static constexpr TeamOwn<MT> v = team_own<MT>();
};

// LDS carve-up (offsets in doubles) for T teams
struct Row2Lds {
  size_t zs, ils, mv, tp, tg, ti, xs, wn, prog, red, part, part2, acc, uni, reg0, reg1, stack, tile, xt, vbs, mbs, total;
  int LD;
};
__host__ __device__ inline Row2Lds row2_lds(const Plan& p, int T, int nslots) {
  Row2Lds L;
  size_t o = 0;
  auto take = [&o](size_t n) { size_t r = o; o += (n + 1) / 2 * 2; return r; };
  L.zs = take((size_t)p.MP * p.DP);
  L.ils = take(16);
  L.mv = take(p.MP);
  L.tp = take(p.P + 1);
  L.tg = take(p.P + 1);
  L.ti = take(p.P + 1);
  L.xs = take(p.S + 1);
  L.wn = take(p.S + 1);
  L.prog = take((size_t)2 * p.nblk + 2);
  L.red = take((size_t)16 * T);
  L.part = take((size_t)T * 4 * 3 * 16);
  L.part2 = take((size_t)T * 4 * 2 * 16);
  L.acc = take((size_t)(p.P > 0 ? p.P : 1) * 64 * T + (size_t)p.RP * 256 * T);
  // union: {two [MP][16] exchange regions per team} | {flow stack: one node in flight per lane} | {transposition tile}
  L.uni = o;
  L.LD = 16 * T + 2;  // [m][16 T rows] tile, stride conflict-free for the transposed ds_read_b64
  const size_t gemm = (size_t)2 * T * p.MP * 16;
  const size_t flow = (size_t)(nslots > 0 ? nslots : 1) * 256 * T;
  const size_t stat = (size_t)p.MP * L.LD + (size_t)16 * T * p.CT16 + 2 * 16 * T + 8;
  size_t u = gemm > flow ? gemm : flow;
  if (stat > u) u = stat;
  L.reg0 = L.uni;
  L.reg1 = L.uni + (size_t)T * p.MP * 16;
  L.stack = L.uni;
  L.tile = L.uni;
  L.xt = L.tile + (size_t)p.MP * L.LD + (((size_t)p.MP * L.LD) & 1);
  L.vbs = L.xt + (size_t)16 * T * p.CT16;
  L.mbs = L.vbs + 16 * T;
  o += (u + 1) / 2 * 2;
  L.total = o;
  return L;
}

// One output tile from pre-loaded A fragments: NSTEPS k-steps, A operand k-step s = av[s] (registers, requested from
// L2 one product ahead), B operand = bp[s * 64] (LDS exchange region, 512 contiguous bytes per fragment, read in
// batches of 8 ahead of their MFMAs); one accumulator (forwarded chain).
template <int NSTEPS>
__device__ __forceinline__ d4 chain_r(const double* av, const double* bp) {
  d4 c = {0, 0, 0, 0};
  // Three waves share this SIMD's matrix pipe: their MFMAs interleave and issue 80-100 cycles apart (measured), against
  // 64 for one wave's chain alone.  Raising the wave's priority for the length of its chain (s_setprio 2 here, 0 after
  // it) changed nothing: TGP_R2_CHAIN_PRIO is left at 0 and the instruction is not emitted.
  if (TGP_R2_CHAIN_PRIO) __builtin_amdgcn_s_setprio(TGP_R2_CHAIN_PRIO);
#pragma unroll
  for (int s0 = 0; s0 < NSTEPS; s0 += 8) {
    double bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < NSTEPS) bv[u] = bp[(s0 + u) * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + u < NSTEPS) c = TGP_MFMA(av[s0 + u], bv[u], c);
  }
  if (TGP_R2_CHAIN_PRIO) __builtin_amdgcn_s_setprio(0);
  return c;
}

// ---------------------------------------------------------------------------------------------------
// Wave roles.  The four waves of a team run DIFFERENT straight-line code (role W = wave index within the team), so
// tile ownership is a compile-time fact inside a role: register arrays are indexed by constants and fully defined on
// every path (with run-time `if (w == owner)` tests around each tile the fragment array was live and partially
// defined across all branches and the kernel spilled at 3 waves per SIMD).  The roles are entered through a
// wave-uniform switch; every role executes the same sequence of workgroup barriers.
// ---------------------------------------------------------------------------------------------------
struct R2Ctx {
  const double *JT, *Jm, *Lq, *LqT;  // A-operand matrices (workspace, L2)
  double *reg0, *reg1;               // this team's exchange regions
  const double *zs, *mv;             // scaled inducing points [MP][DP], padded variational mean (LDS)
  int lane, nl, q, M;
  double s2;
  bool act;
  double* stamp;  // workspace header stamp slots (diagnostic build)
};

template <int MT, int DP, int W>
struct TeamRole {
  using OWNC = TeamOwnC<MT>;
  static constexpr int MP = MT * 16;
  static constexpr int NSLOT = OWNC::v.nslot;
  static constexpr int NMAX = OWNC::v.nmax;
  static constexpr int NPRE = NMAX < TGP_R2_PREFETCH ? NMAX : TGP_R2_PREFETCH;

  // A fragments [LO, HI) of this role's array for one product.  LOWER: tile i contracts k-steps 0 .. 4 i + 3 (rows of
  // the transposed operand Mx); upper: k-steps 4 i .. 4 MT - 1.  One 8-byte load per lane and fragment (4 rows x 128 B).
  template <bool LOWER, int LO, int HI, int NA>
  static __device__ __forceinline__ void load(const double* __restrict__ Mx, int q, int nl, double (&av)[NA]) {
    static_for<MT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int li = LOWER ? i : MT - 1 - i;  // the lower tile whose owner / offset this tile shares
      if constexpr (OWNC::v.owner[li] == W) {
        constexpr int off = OWNC::v.off[li], ns = 4 * (li + 1), s0 = LOWER ? 0 : 4 * i;
        const double* __restrict__ ap = Mx + (size_t)(4 * s0 + q) * MP + 16 * i + nl;
#pragma unroll
        for (int s = 0; s < ns; ++s)
          if (off + s >= LO && off + s < HI) av[off + s] = ap[(size_t)s * 4 * MP];
      }
    });
  }
  // av[0 .. NPRE) <- pre[], the rest requested now
  template <bool LOWER>
  static __device__ __forceinline__ void gather(const double* __restrict__ Mx, int q, int nl, const double (&pre)[NPRE],
                                                double (&av)[NMAX]) {
#pragma unroll
    for (int s = 0; s < NPRE; ++s) av[s] = pre[s];
    load<LOWER, NPRE, NMAX>(Mx, q, nl, av);
  }

  // phases 1-3: K tiles -> reg0 | A = J K -> reg1 | B = Lq^T A.  `pre` enters with the first fragments of J^T and leaves
  // with the first fragments of Lq^T (third product), which arrive during the flow phase.
  static __device__ __forceinline__ void forward(const R2Ctx& c, const double (&x)[DP], double (&pre)[NPRE],
                                                 d4 (&Kown)[NSLOT], d4 (&Aown)[NSLOT], d4 (&Bown)[NSLOT], double& pm,
                                                 double& pa, double& pb) {
    const int lane = c.lane, q = c.q, nl = c.nl;
    if (c.act) {
      static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (OWNC::v.owner[MT - 1 - i] == W) {
          constexpr int sl = OWNC::v.slot[MT - 1 - i];
          double e[4];
          TGP_EACH(r, 4) {
            const int mm = 16 * i + 4 * r + q;
            double d2 = 0.0;
#pragma unroll
            for (int d = 0; d < DP; ++d) {
              const double t = x[d] - c.zs[mm * DP + d];
              d2 += t * t;
            }
            e[r] = -0.5 * d2;
          }
          exp_fast_n<4>(e);
          TGP_EACH(r, 4) {
            const double kv = (16 * i + 4 * r + q < c.M ? c.s2 : 0.0) * e[r];
            Kown[sl][r] = kv;
            c.reg0[256 * i + 64 * r + lane] = kv;
          }
        }
      });
    }
    __syncthreads();
    TGP_STAMP_R2(2);
    if (c.act) {
      double av[NMAX];
      gather<true>(c.JT, q, nl, pre, av);
      static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (OWNC::v.owner[i] == W) {
          constexpr int sl = OWNC::v.slot[i];
          const d4 cc = chain_r<4 * (i + 1)>(av + OWNC::v.off[i], c.reg0 + lane);
          Aown[sl] = cc;
          TGP_EACH(r, 4) {
            c.reg1[256 * i + 64 * r + lane] = cc[r];
            pm += c.mv[16 * i + 4 * r + q] * cc[r];
            pa += cc[r] * cc[r];
          }
        }
      });
      load<false, 0, NPRE>(c.Lq, q, nl, pre);  // next product's first fragments: in flight across the barrier
    }
    __syncthreads();
    TGP_STAMP_R2(3);
    if (c.act) {
      double av[NMAX];
      gather<false>(c.Lq, q, nl, pre, av);
      static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (OWNC::v.owner[MT - 1 - i] == W) {
          constexpr int sl = OWNC::v.slot[MT - 1 - i];
          const d4 cc = chain_r<4 * (MT - i)>(av + OWNC::v.off[MT - 1 - i], c.reg1 + 256 * i + lane);
          Bown[sl] = cc;
          TGP_EACH(r, 4) pb += cc[r] * cc[r];
        }
      });
      load<true, 0, NPRE>(c.LqT, q, nl, pre);
    }
  }

  // phases 5-7: B vbar -> reg0 | C = Lq (B vbar), Abar = m mubar^T - 2 A vbar + 2 C -> reg1 | Kbar = J^T Abar,
  // E = Kbar o K left in Bown
  static __device__ __forceinline__ void backward(const R2Ctx& c, double mub, double vb, double (&pre)[NPRE],
                                                  const d4 (&Kown)[NSLOT], const d4 (&Aown)[NSLOT], d4 (&Bown)[NSLOT]) {
    const int lane = c.lane, q = c.q, nl = c.nl;
    if (c.act) {
      static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (OWNC::v.owner[MT - 1 - i] == W) {
          constexpr int sl = OWNC::v.slot[MT - 1 - i];
          TGP_EACH(r, 4) c.reg0[256 * i + 64 * r + lane] = Bown[sl][r] * vb;
        }
      });
    }
    __syncthreads();
    TGP_STAMP_R2(6);
    if (c.act) {
      double av[NMAX];
      gather<true>(c.LqT, q, nl, pre, av);
      static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (OWNC::v.owner[i] == W) {
          constexpr int sl = OWNC::v.slot[i];
          const d4 cc = chain_r<4 * (i + 1)>(av + OWNC::v.off[i], c.reg0 + lane);
          TGP_EACH(r, 4)
            c.reg1[256 * i + 64 * r + lane] = c.mv[16 * i + 4 * r + q] * mub - 2.0 * Aown[sl][r] * vb + 2.0 * cc[r];
        }
      });
      load<false, 0, NPRE>(c.Jm, q, nl, pre);
    }
    __syncthreads();
    TGP_STAMP_R2(7);
    if (c.act) {
      double av[NMAX];
      gather<false>(c.Jm, q, nl, pre, av);
      static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (OWNC::v.owner[MT - 1 - i] == W) {
          constexpr int sl = OWNC::v.slot[MT - 1 - i];
          const d4 cc = chain_r<4 * (MT - i)>(av + OWNC::v.off[MT - 1 - i], c.reg1 + 256 * i + lane);
          TGP_EACH(r, 4) Bown[sl][r] = cc[r] * Kown[sl][r];
        }
      });
    }
  }
};

template <int MT, int DP>
__global__ __launch_bounds__(256 * TGP_R2_MAX_TEAMS) void k_rows2(RowArgs a, int T) {
  constexpr int MP = MT * 16;
  constexpr int CT = (2 * DP + 1 + 15) / 16, CT16 = CT * 16;
  using OWNC = TeamOwnC<MT>;
  constexpr int NSLOT = OWNC::v.nslot;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const Plan& p = a.p;
  const Row2Lds L = row2_lds(p, T, a.prog.nslots);
  const int LD = L.LD;
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
  double* part = sm + L.part;
  double* part2 = sm + L.part2;
  double* acc = sm + L.acc;
  double* stack = sm + L.stack;
  double* tile = sm + L.tile;
  double* xt = sm + L.xt;
  double* vbs = sm + L.vbs;
  double* mbs = sm + L.mbs;

  const int tid = threadIdx.x, nthr = 256 * T, lane = tid & 63, nl = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform BY CONSTRUCTION: keep it in an SGPR, so that
  const int team = wave >> 2, w = wave & 3, nwaves = 4 * T;   // role switches and item loops are scalar branches
  const double* __restrict__ ws = a.ws;
  const int N = p.N, D = p.D, M = p.M, P = p.P, RP = p.RP;

  if ((int)blockIdx.x >= p.nblocks) {
    // ---- passenger blocks (as in k_rows): H'^T = (J^T (S - I))^T tiles and w = J^T m for the backward M x M chain;
    //      they start on the CUs whose workgroup holds fewer groups and finishes early ----
    const int t = blockIdx.x - p.nblocks;
    const double* __restrict__ Jg = ws + p.J;
    if (t == MT * MT) {
      for (int i = tid; i < MP; i += nthr) {
        double s = 0.0;
        for (int k = i; k < MP; ++k) s += Jg[(size_t)k * MP + i] * ws[p.mpad + k];
        a.ws[p.w + i] = s;
      }
      return;
    }
    if (wave != 0) return;
    const int ti = t / MT, tj = t % MT;
    const double* __restrict__ Sg = ws + p.S_;
    d4 hacc = {0, 0, 0, 0};
    hacc = tile_mm_f<TGP_GBATCH>([&](int k) { return Jg[(size_t)(k + q) * MP + 16 * ti + nl]; },
                                 [&](int k) { return Sg[(size_t)(k + q) * MP + 16 * tj + nl]; }, 16 * ti, MP, hacc);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = 16 * ti + q + 4 * rr, col = 16 * tj + nl;
      a.ws[p.HpT + (size_t)col * MP + row] = hacc[rr] - Jg[(size_t)col * MP + row];
    }
    return;
  }

  TGP_STAMP(a.ws, p, 0);
#ifdef TGP_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) a.ws[p.hdr + H_STAMP + 11] = (double)clock64();
#endif
  // group of this team: strided over the workgroups, so that the workgroups holding T groups are the first ones
  const int ngroups = (N + 15) / 16;
  const int grp = blockIdx.x + team * p.nblocks;
  const bool act = grp < ngroups;  // wave-uniform
  const int n = grp * 16 + nl;
  const bool valid = act && n < N;
  const int nc = valid ? n : N - 1;
  double xraw[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) xraw[d] = d < D ? a.X[(size_t)nc * D + d] : 0.0;
  const double y = a.Y[nc];
  const double s2 = ws[p.hdr + H_S2], eta = ws[p.hdr + H_ETA], einv = ws[p.hdr + H_EINV];

  const double* __restrict__ JT = ws + p.JT;
  const double* __restrict__ Jm = ws + p.J;
  const double* __restrict__ Lq = ws + p.Lq;
  const double* __restrict__ LqT = ws + p.LqT;
  // first A fragments of the first product: requested before anything else (they need nothing from LDS)
  constexpr int NPRE = TeamRole<MT, DP, 0>::NPRE;
  double pre[NPRE];
#pragma unroll
  for (int s = 0; s < NPRE; ++s) pre[s] = 0.0;
  if (act) {
    switch (w) {
      case 0: TeamRole<MT, DP, 0>::template load<true, 0, NPRE>(JT, q, nl, pre); break;
      case 1: TeamRole<MT, DP, 1>::template load<true, 0, NPRE>(JT, q, nl, pre); break;
      case 2: TeamRole<MT, DP, 2>::template load<true, 0, NPRE>(JT, q, nl, pre); break;
      default: TeamRole<MT, DP, 3>::template load<true, 0, NPRE>(JT, q, nl, pre); break;
    }
  }

  // ---- stage the small shared operands (the first slice of every array is requested before anything is stored:
  //      one L2 round trip for all of them instead of one per array) ----
  {
    constexpr int NZ = (MP * DP + 255) / 256;
    double zv[NZ];
#pragma unroll
    for (int u = 0; u < NZ; ++u) zv[u] = tid + nthr * u < MP * DP ? ws[p.Zs + tid + nthr * u] : 0.0;
    const double mv0 = tid < MP ? ws[p.mpad + tid] : 0.0;
    const double il0 = tid < 16 ? ws[p.ils + tid] : 0.0;
    const bool fl = p.lik == TGP_LIK_FLOW;
    const double tp0 = (fl && tid < P) ? ws[p.tp + tid] : 0.0, tg0 = (fl && tid < P) ? ws[p.tg + tid] : 0.0;
    const double xs0 = (fl && tid < p.S) ? a.xs[tid] : 0.0, wn0 = (fl && tid < p.S) ? a.wn[tid] : 0.0;
    const int pr0 = (fl && tid < 4 * p.nblk) ? a.prog.blk[tid] : 0;
#pragma unroll
    for (int u = 0; u < NZ; ++u)
      if (tid + nthr * u < MP * DP) zs[tid + nthr * u] = zv[u];
    if (tid < MP) mv[tid] = mv0;
    if (tid < 16) ils[tid] = il0;
    if (fl) {
      if (tid < P) { tpL[tid] = tp0; tgL[tid] = tg0; tiL[tid] = rcp_fast(tp0); }
      if (tid < p.S) { xsL[tid] = xs0; wnL[tid] = wn0; }
      if (tid < 4 * p.nblk) progL[tid] = pr0;
      for (int i = tid + nthr; i < P; i += nthr) { tpL[i] = ws[p.tp + i]; tgL[i] = ws[p.tg + i]; tiL[i] = rcp_fast(tpL[i]); }
      for (int i = tid + nthr; i < p.S; i += nthr) { xsL[i] = a.xs[i]; wnL[i] = a.wn[i]; }
      for (int i = tid + nthr; i < 4 * p.nblk; i += nthr) progL[i] = a.prog.blk[i];
    }
    const int nacc = (P > 0 ? P : 1) * 64 * T + RP * 256 * T;
    for (int i = tid; i < nacc; i += nthr) acc[i] = 0.0;
  }
  __syncthreads();

  TGP_STAMP(a.ws, p, 1);
  double x[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) x[d] = d < D ? xraw[d] * ils[d] : 0.0;

  double* reg0 = sm + L.reg0 + (size_t)team * MP * 16;
  double* reg1 = sm + L.reg1 + (size_t)team * MP * 16;

  d4 Kown[NSLOT], Aown[NSLOT], Bown[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) { Kown[s] = d4{0, 0, 0, 0}; Aown[s] = d4{0, 0, 0, 0}; Bown[s] = d4{0, 0, 0, 0}; }

  // ---- phases 1-3 (wave roles): K tiles -> reg0 | A = J K -> reg1 | B = Lq^T A ----
  R2Ctx cx;
  cx.JT = JT; cx.Jm = Jm; cx.Lq = Lq; cx.LqT = LqT;
  cx.reg0 = reg0; cx.reg1 = reg1; cx.zs = zs; cx.mv = mv;
  cx.lane = lane; cx.nl = nl; cx.q = q; cx.M = M; cx.s2 = s2; cx.act = act;
  cx.stamp = a.ws + p.hdr + H_STAMP;
  double pm = 0.0, pa = 0.0, pb = 0.0;
  switch (w) {
    case 0: TeamRole<MT, DP, 0>::forward(cx, x, pre, Kown, Aown, Bown, pm, pa, pb); break;
    case 1: TeamRole<MT, DP, 1>::forward(cx, x, pre, Kown, Aown, Bown, pm, pa, pb); break;
    case 2: TeamRole<MT, DP, 2>::forward(cx, x, pre, Kown, Aown, Bown, pm, pa, pb); break;
    default: TeamRole<MT, DP, 3>::forward(cx, x, pre, Kown, Aown, Bown, pm, pa, pb); break;
  }
  if (act) {
    pm = quad_sum(pm); pa = quad_sum(pa); pb = quad_sum(pb);
    if (q == 0) {
      double* pp = part + (size_t)(team * 4 + w) * 48 + nl;
      pp[0] = pm; pp[16] = pa; pp[32] = pb;
    }
  }
  __syncthreads();  // partial moments visible; exchange regions dead (the flow stack aliases them)

  TGP_STAMP(a.ws, p, 4);
  // ---- phase 4: mu, v; expected log-likelihood and its adjoints (16 lanes per row: 4 lanes x 4 waves) ----
  double mu = 0.0, v = 1.0, mub = 0.0, vb = 0.0, ellp = 0.0, etap = 0.0;
  if (act) {
    const double* pp = part + (size_t)team * 4 * 48 + nl;
    pm = (pp[0] + pp[48]) + (pp[96] + pp[144]);
    pa = (pp[16] + pp[64]) + (pp[112] + pp[160]);
    pb = (pp[32] + pp[80]) + (pp[128] + pp[176]);
    mu = pm;
    v = s2 - pa + pb;
    if (a.mu != nullptr && w == 0 && q == 0 && valid) { a.mu[n] = mu; a.v[n] = v; }
    if (p.lik == TGP_LIK_GAUSS) {
      // GaussianLinearMean.expected_log_prob (likelihoods/GaussianLinearMean.py:81-87); every wave of the team
      // forms the same mubar, vbar, the scalars are counted once
      const double r = y - mu;
      mub = a.scale * einv * r;
      vb = -0.5 * a.scale * einv;
      if (q == 0 && w == 0) {
        ellp = -0.5 * TGP_LOG_2PI_REF - 0.5 * eta - 0.5 * einv * (r * r + v);
        etap = -0.5 + 0.5 * einv * (r * r + v);
      }
    } else {
      // GaussianNonLinearMean.expected_log_prob (likelihoods/GaussianNonLinearMean.py:91-148): this lane takes the
      // quadrature nodes s = q + 4 w + 16 it of its row
      FlowDev F{progL, p.nblk, tpL, tgL, tiL};
      const double sq = sqrt(2.0 * v);
      const double* rp = a.rowp != nullptr ? a.rowp + (size_t)nc * RP : nullptr;
      double cm = 0.0, cv = 0.0;
      double* accq = acc + (team * 64 + w * 16 + nl);
      double* accr = acc + (size_t)(P > 0 ? P : 1) * 64 * T + tid;
      const int ntrip = (p.S + 15) / 16;
      for (int it = 0; it < ntrip; ++it) {
        double xn[1], wq[1], f[1], c[1];
        const int sn = q + 4 * w + 16 * it;
        xn[0] = xsL[sn < p.S ? sn : 0];
        wq[0] = (valid && sn < p.S) ? wnL[sn] : 0.0;
        f[0] = mu + sq * xn[0];
        flow_forward_store<1>(F, f, rp, stack + tid, nthr);
        const double r = y - f[0];
        ellp += wq[0] * (-0.5 * TGP_LOG_2PI_REF - 0.5 * eta - 0.5 * einv * r * r);
        etap += wq[0] * (-0.5 + 0.5 * einv * r * r);
        c[0] = a.scale * einv * wq[0] * r;
        flow_backward_store<1>(F, c, rp, stack + tid, nthr, a.prog.nslots, accq, 64 * T, q == 0, accr, nthr);
        cm += c[0];
        cv += c[0] * xn[0];
      }
      cm = quad_sum(cm);
      cv = quad_sum(cv);
      if (q == 0) {
        double* p2 = part2 + (size_t)(team * 4 + w) * 32 + nl;
        p2[0] = cm; p2[16] = cv;
      }
    }
  }
  __syncthreads();  // flow stack dead; adjoint partials visible

  TGP_STAMP(a.ws, p, 5);
  // ---- phase 5: mubar, vbar; B vbar -> reg0 ----
  if (act) {
    if (p.lik != TGP_LIK_GAUSS) {
      const double* p2 = part2 + (size_t)team * 4 * 32 + nl;
      mub = (p2[0] + p2[32]) + (p2[64] + p2[96]);
      vb = ((p2[16] + p2[48]) + (p2[80] + p2[112])) / sqrt(2.0 * v);
    }
    if (!valid) { mub = 0.0; vb = 0.0; ellp = 0.0; etap = 0.0; }
  }
  // ---- phases 5-7 (wave roles): B vbar -> reg0 | C = Lq (B vbar), Abar -> reg1 | Kbar = J^T Abar, E = Kbar o K in Bown ----
  switch (w) {
    case 0: TeamRole<MT, DP, 0>::backward(cx, mub, vb, pre, Kown, Aown, Bown); break;
    case 1: TeamRole<MT, DP, 1>::backward(cx, mub, vb, pre, Kown, Aown, Bown); break;
    case 2: TeamRole<MT, DP, 2>::backward(cx, mub, vb, pre, Kown, Aown, Bown); break;
    default: TeamRole<MT, DP, 3>::backward(cx, mub, vb, pre, Kown, Aown, Bown); break;
  }
  __syncthreads();  // exchange regions dead: the union becomes the transposition tile

  TGP_STAMP(a.ws, p, 8);
  double* slab = a.ws + p.slabs + (size_t)blockIdx.x * p.slab_len;
  const int col = team * 16 + nl;

  // ---- phase 8: E through LDS (transposed), T = E [xs, xs^2, 1] over the 16 T rows of the workgroup ----
  static_for<MT>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int ow = OWNC::v.owner[MT - 1 - i], sl = OWNC::v.slot[MT - 1 - i];
    if (w == ow) { TGP_EACH(r, 4) tile[(16 * i + 4 * r + q) * LD + col] = act ? Bown[sl][r] : 0.0; }
  });
  if (w == 0) {
#pragma unroll
    for (int c = 0; c < CT16; ++c) {
      if ((c & 3) == q) {
        double val = 0.0;
        if (c < DP) val = x[c < DP ? c : 0];
        else if (c < 2 * DP) val = x[(c - DP) < DP ? (c - DP) : 0] * x[(c - DP) < DP ? (c - DP) : 0];
        else if (c == 2 * DP) val = 1.0;
        xt[col * CT16 + c] = val;
      }
    }
  }
  __syncthreads();
  for (int t = wave; t < MT * CT; t += nwaves) {
    const int ti = t / CT, tc = t % CT;
    d4 c = {0, 0, 0, 0};
    c = tile_mm_f<4>([&](int k) { return tile[(16 * ti + nl) * LD + k + q]; },
                     [&](int k) { return xt[(k + q) * CT16 + 16 * tc + nl]; }, 0, 16 * T, c);
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[p.slab_T + (size_t)(16 * ti + q + 4 * r) * CT16 + 16 * tc + nl] = c[r];
  }
  __syncthreads();

  TGP_STAMP(a.ws, p, 9);
  // ---- phase 9: A through LDS, G = A diag(vbar) A^T (lower tiles), s = A mubar ----
  static_for<MT>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int ow = OWNC::v.owner[i], sl = OWNC::v.slot[i];
    if (w == ow) { TGP_EACH(r, 4) tile[(16 * i + 4 * r + q) * LD + col] = act ? Aown[sl][r] : 0.0; }
  });
  if (w == 0 && q == 0) { vbs[col] = vb; mbs[col] = mub; }
  __syncthreads();
  TGP_STAMP(a.ws, p, 18);
  {
    // work items: MT (MT + 1) / 2 lower G tiles (row-major over the lower triangle of tiles) then MT tiles of s,
    // dealt round-robin to the 4 T waves; each contracts over the 16 T rows (4 T k-steps)
    constexpr int NG = MT * (MT + 1) / 2;
    for (int it = wave; it < NG + MT; it += nwaves) {
      d4 c = {0, 0, 0, 0};
      if (it < NG) {
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= it) ++ti;
        const int tj = it - ti * (ti + 1) / 2;
        const double* ta = tile + (16 * ti + nl) * LD + q;
        const double* tb = tile + (16 * tj + nl) * LD + q;
        // one team's 16 rows (4 k-steps) per trip: the 12 LDS operands of a trip are requested together (with a
        // run-time trip count inside the k loop every k-step was its own basic block: one LDS round trip each)
        for (int t = 0; t < T; ++t) {
          double af[4], sc[4], bv[4];
          TGP_EACH(k, 4) { af[k] = ta[16 * t + 4 * k]; sc[k] = vbs[16 * t + 4 * k + q]; bv[k] = tb[16 * t + 4 * k]; }
          TGP_EACH(k, 4) af[k] *= sc[k];
          TGP_EACH(k, 4) c = TGP_MFMA(af[k], bv[k], c);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[p.slab_G + (size_t)it * 256 + (q + 4 * r) * 16 + nl] = c[r];
      } else {
        const int ti = it - NG;
        const double* ta = tile + (16 * ti + nl) * LD + q;
        for (int t = 0; t < T; ++t) {
          double af[4], bm[4];
          TGP_EACH(k, 4) { af[k] = ta[16 * t + 4 * k]; bm[k] = nl == 0 ? mbs[16 * t + 4 * k + q] : 0.0; }
          TGP_EACH(k, 4) c = TGP_MFMA(af[k], bm[k], c);
        }
        if (nl == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[p.slab_S + 16 * ti + q + 4 * r] = c[r];
        }
      }
    }
  }

  TGP_STAMP(a.ws, p, 10);
  // ---- scalars, flow parameter gradients ----
  const double e1 = wave_sum(ellp), e2 = wave_sum(etap), e3 = wave_sum((q == 0 && w == 0) ? vb : 0.0);
  if (lane == 0) { red[wave * 4] = e1; red[wave * 4 + 1] = e2; red[wave * 4 + 2] = e3; }
  __syncthreads();
  if (tid == 0) {
    double r0 = 0.0, r1 = 0.0, r2 = 0.0;
    for (int k = 0; k < nwaves; ++k) { r0 += red[4 * k]; r1 += red[4 * k + 1]; r2 += red[4 * k + 2]; }
    slab[p.slab_C + C_ELL] = a.scale * r0;
    slab[p.slab_C + C_ETAB] = a.scale * r1;
    slab[p.slab_C + C_SVB] = r2;
    slab[p.slab_C + C_PAD] = 0.0;
  }
  for (int j = wave; j < P; j += nwaves) {
    double s = 0.0;
    for (int k = 0; k < T; ++k) s += wave_sum(acc[(size_t)j * 64 * T + 64 * k + lane]);
    if (lane == 0) slab[p.slab_C + C_THETA + j] = s;
  }
  for (size_t i = p.slab_C + C_THETA + P + tid; i < p.slab_len; i += nthr) slab[i] = 0.0;
  TGP_STAMP(a.ws, p, 17);
#ifdef TGP_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) a.ws[p.hdr + H_STAMP + 23] = (double)clock64();
#endif
  if (a.g_rowp != nullptr && valid) {
    // per-row parameter gradients: the 16 lanes of a row (4 lanes x 4 waves) each hold a partial
    const double* rbase = acc + (size_t)(P > 0 ? P : 1) * 64 * T;
    if (w == 0 && q == 0) {
      for (int jr = 0; jr < RP; ++jr) {
        const double* ap = rbase + (size_t)jr * nthr + team * 256 + nl;
        double s = 0.0;
        for (int k = 0; k < 16; ++k) s += ap[16 * k];
        a.g_rowp[(size_t)n * RP + jr] = s;
      }
    }
  }
}

}  // namespace tgp
