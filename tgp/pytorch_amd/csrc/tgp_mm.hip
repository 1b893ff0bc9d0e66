// tgp_mm.hip -- the M x M side of the ELBO step (everything that does not scale with the rows).
//
// Forward  (k_prep_a; H' and w ride along the row-kernel launch, tgp_rows.hpp):  lengthscale/outputscale transforms, Zs = Z/l, K_MM, Cholesky L,
//          J = L^-1, masked L_q, S = L_q L_q^T, H' = J^T (S - I), w = J^T m, KL, flow parameter transforms.
//          Replaces models/sparse_MF_SP.py:316,330,344-346,406-431 and dsp/utils.py:222-270 (the retry
//          ladder itself stays on the host, driven by status[]).
// Backward (k_reduce, then k_bwd: one launch, roles handing on inside it): slab reduction of the row statistics, then the hand-derived adjoint
//          (restructured -- see DESIGN.md section 3):
//            Lbar   = -tril(w s^T + 2 H' G)            Lambar = 2 tril(G L_q) - kl (L_q - diag(1/Lam_ii))
//            Q      = Phi(L^T Lbar) + Phi(.)^T         Kbar_MM = 1/2 J^T Q J
//          and the ARD-RBF parameter gradients from Kbar_MM and the row statistics T.
// All GEMM-shaped work is 16x16 output tiles on v_mfma_f64_16x16x4_f64, one wave per tile.
#include "tgp_dev.hpp"
#include "tgp_prep.hpp"
#include "tgp_launch.hpp"

namespace tgp {

// ---------------------------------------------------------------------------------------------------
// k_prep_a  (grid = MT*MT + 3 workgroups of 512 threads; producers of the in-launch hand-off carry the lowest indices)
//   block t < MT^2   : 16x16 tile t of {Lq, Lq^T, K_MM -> HBM} and the S = Lq Lq^T tile (one MFMA chain)
//   block MT^2       : parameter transforms, Zs, padded m, flow parameter transforms, KL, header scalars
//   blocks MT^2 + 1, MT^2 + 2 : the two chain blocks -- K_MM into LDS, blocked right-looking Cholesky (8 waves, MFMA)
// ---------------------------------------------------------------------------------------------------
#define PREP_THREADS 512
#ifndef TGP_W4_WINDOWS
#define TGP_W4_WINDOWS 2 /* wave 4 (the chain wave's SIMD partner) takes tasks only in the first windows, the ones with the most fills */
#endif

__global__ __launch_bounds__(PREP_THREADS) void k_prep_a(Plan p, tgp_model md, FlowProg fp, double* __restrict__ ws,
                                                         int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave number as a SCALAR: every task / tile index derived from it stays in SGPRs and its branches are scalar
  // branches (as a VGPR value hipcc treats them as divergent: exec-mask code, vector address arithmetic, spills)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int M = p.M, D = p.D, MP = p.MP, DP = p.DP, MT = p.MT;

  int32_t* sy = status + 4;   // hand-off words (tgp_prep.hpp): SY_TILES and SY_DONE
  const int nb_total = 3 + MT * MT;
  // The tile blocks -- the PRODUCERS the chain blocks wait for -- come first in the grid: workgroups are dispatched in
  // index order, so they are resident (or done) before a chain block can occupy a CU and spin (with the chain
  // blocks in front, a saturated GPU could seat the waiters and leave their producers queued until the bounded wait
  // expired).
  if ((int)blockIdx.x < MT * MT) {
    // ---------------- tile blocks: Lq, Lq^T, K_MM copies + S = Lq Lq^T (sparse_MF_SP.py:316,344-346); they count
    //                  themselves in SY_TILES once K_MM is out: the chain blocks take K_MM's columns >= 2 from there
    prep_tile_role<PREP_THREADS, true>(p, md, ws, (int)blockIdx.x, sy);
    sync_leave(sy, nb_total);
    return;
  }

  if ((int)blockIdx.x == MT * MT) {
    // ---------------- transforms, padded copies, flow parameter transforms, KL, header ---------------------------
    prep_xform_role<PREP_THREADS, false>(p, md, fp, ws);
    sync_leave(sy, nb_total);
    return;
  }

  // ---------------- blocks 0 and 1: blocked Cholesky (torch.cholesky, dsp/utils.py:239) + inverse -------------------
  // TWO workgroups on two CUs run the same factorisation redundantly (same arithmetic, bit for bit, nothing exchanged)
  // and share what it leaves behind: block b forms and stores the tile COLUMNS c = b (mod 2) of J = L^-1 (a tile of J
  // needs only tiles of its own column) and of L.  The task windows, not the chain, bound this launch (the four SIMDs of
  // one CU saturated by instruction issue): a second CU takes 40 % of a window's work at the price of nothing but a CU
  // that was idle.
  const int cb = (int)blockIdx.x - (MT * MT + 1);   // 0 or 1: this block's column parity
  // Round-3 schedule: RIGHT-looking, one 16-column PANEL factorisation per block column (potrf_panel16: the diagonal
  // tile and every row below it leave one register pass of one wave -- no inverse of the diagonal tile, no triangular
  // solve, no panel product on the critical chain), everything GEMM-shaped on the other waves:
  //   iteration j, phase P : wave 0 (and wave 1 for the rows beyond 64 below the diagonal): column j of L;
  //                          the other waves, off the chain: [j = 0] fill of K_MM columns 1.. ; [j >= 1] the trailing
  //                          update of column j-1 on the tiles (i, k >= j+1), the inverse Dinv_{j-1} of diagonal tile
  //                          j-1 (trtri16), the tiles of row j-2 of J = L^-1, the write-out of finished tile rows
  //                phase U : tiles (i, j+1) -= L(i, j) L(j+1, j)^T, one tile per wave (4 MFMAs): all column j+1 needs
  // Two LDS-only barriers per block column; the chain per block column is potrf_panel16 (3.4 k cycles) + phase U,
  // against potrf+trtri (4.6 k) + panel product + diagonal update + the helpers' skew in the left-looking schedule of
  // rounds 1-2 (10 k cycles per block column measured).
  constexpr int NW = PREP_THREADS / 64;
  const int LD = MP + 1;
  double* A = sm;                            // MP x LD: lower = K_MM -> L
  double* zs = sm + (size_t)MP * LD;         // MP x DP scaled inducing points, present when p.zs_lds (LDS budget allows)
  __shared__ int s_info, s_nan, s_next, s_sync;
  __shared__ double s_ils[16];
  if (tid == 0) { s_info = 0; s_nan = 0; s_next = 0; s_sync = 0; }
  int tbase = 0;  // first task id of the current window (the same in every wave)
  if (tid < 16) s_ils[tid] = tid < D ? 1.0 / softplus_d(md.raw_ls[tid]) : 0.0;
  const double s2 = softplus_d(md.raw_os[0]);
  const bool zl = p.zs_lds != 0;
  if (zl) {
    // every thread transforms the lengthscale it needs itself: the Z and raw_ls loads go out together and the
    // set-up pays one memory round trip and one barrier instead of two of each
    for (int i = tid; i < MP * DP; i += PREP_THREADS) {
      const int mrow = i / DP, d = i % DP;
      zs[i] = (mrow < M && d < D) ? md.Z[(size_t)mrow * D + d] * (1.0 / softplus_d(md.raw_ls[d])) : 0.0;
    }
  }
  __syncthreads();
  // K_MM element (lower triangle only: the factorisation never reads above the diagonal; the strict-upper TILES later
  // receive J^T); identity on the padding keeps L and L^-1 well defined
  bool has_nan = false;
  double jit = md.jitter;  // raised by the on-device retry ladder below
  auto kmm_elem = [&](int rr, int cc) {
    double k;
    if (rr < M) {
      double d2 = 0.0;
      if (zl) {
        // DP is 4, 8 or 16 and the rows are 32-byte aligned: two 16-byte reads per operand and four dimensions per trip
        for (int d = 0; d < DP; d += 4) {
          const double2 a0 = *reinterpret_cast<const double2*>(zs + rr * DP + d), a1 = *reinterpret_cast<const double2*>(zs + rr * DP + d + 2);
          const double2 b0 = *reinterpret_cast<const double2*>(zs + cc * DP + d), b1 = *reinterpret_cast<const double2*>(zs + cc * DP + d + 2);
          const double t0 = a0.x - b0.x, t1 = a0.y - b0.y, t2 = a1.x - b1.x, t3 = a1.y - b1.y;
          d2 += t0 * t0; d2 += t1 * t1; d2 += t2 * t2; d2 += t3 * t3;
        }
      } else {  // M = 128 with D > 8: no LDS left next to A; read Z through L1/L2
        for (int d = 0; d < D; ++d) {
          const double tt = (md.Z[(size_t)rr * D + d] - md.Z[(size_t)cc * D + d]) * s_ils[d];
          d2 += tt * tt;
        }
      }
      k = s2 * exp_fast(-0.5 * d2);
      has_nan |= (k != k);
      if (rr == cc) k += jit;
    } else {
      k = (rr == cc) ? 1.0 : 0.0;
    }
    return k;
  };
  // acc = sum_{kbeg <= k < kend} L[i-tile rows, k] * L[j-tile rows, k]   (both operands in LDS)
  auto ll_sum = [&](int i0, int j0, int kbeg, int kend) {
    d4 acc = {0, 0, 0, 0};
    return tile_mm_f([&](int k) { return A[(i0 + r) * LD + k + q]; }, [&](int k) { return A[(j0 + r) * LD + k + q]; }, kbeg,
                     kend, acc);
  };
  // Write-out of lower tile (ti, tj) of L and of the tile (tj, ti) of L^T, by one wave, 128-byte row segments.  A single
  // CU stores at ~30-50 GB/s, so the chain blocks write only what is non-zero (the structurally-zero tiles of L are
  // written by the otherwise idle tile blocks of this launch) and they do so DURING the factorisation, in the shadow of
  // the chain.  (A retry of the jitter ladder simply stores again.)
  auto write_L = [&](int ti, int tj) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = 16 * ti + q + 4 * u, cc = 16 * tj + r;
      ws[p.L + (size_t)rr * MP + cc] = (ti != tj || cc <= rr) ? A[rr * LD + cc] : 0.0;
      // L^T tile (tj, ti): element [16 tj + q+4u][16 ti + r] = L[16 ti + r][16 tj + q+4u]
      const int rt = 16 * tj + q + 4 * u, ct = 16 * ti + r;
      ws[p.LT + (size_t)rt * MP + ct] = (ti != tj || rt <= ct) ? A[ct * LD + rt] : 0.0;
    }
  };
  // minus the inverse of the finished diagonal tile jt -> workspace (the row kernel closes every substitution step with
  // it, tgp_rows.hpp): one wave, trtri16 on its own copy of L_jj, one window behind the factorisation and off its chain
  auto inv_diag = [&](int jt) {
    const int li = lane & 15;
    double dgv[16], xv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) dgv[c] = A[(16 * jt + li) * LD + 16 * jt + c];
    trtri16(dgv, xv, li);
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // xv[c] = Dinv[c][li]; the four 16-lane rows hold the same values: row q stores columns 4u + q
      const double xv4 = q == 0 ? xv[4 * u] : (q == 1 ? xv[4 * u + 1] : (q == 2 ? xv[4 * u + 2] : xv[4 * u + 3]));
      ws[p.nD + jt * 256 + (4 * u + q) * 16 + li] = -xv4;
    }
  };
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
  // tile (i, c) of K_MM minus the contributions of the block columns [0, ncol) (all final): the tile's first value
  // in LDS (C/D layout: rows q + 4u, column r).  The four exponentials of a lane are written stage by stage
  // (exp_fast_n: independent chains for the scheduler; called element by element they came out back to back)
  const double* __restrict__ Kg = ws + p.Kmm;
  bool tiles_seen = false;
  auto fill_tile = [&](int i, int c, int ncol) {
    double kv[4];
    if (c >= 2) {
      // Round 4: the tile blocks of this launch have K_MM in global memory long before column 2 is due (window 1, ~10 us
      // in; they are done by ~7): four coherent loads per lane instead of four exponential chains -- the 21 tile fills
      // were what the task waves of a window could not finish under the chain's register pass.
      if (!tiles_seen) {
        const int v = sync_wait(sy + SY_TILES, [&](int x) { return (x & 0xffff) >= MT * MT; });
        if (v == (int)0x80000000 && lane == 0) s_sync = 1;
        tiles_seen = true;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) kv[u] = ld_agent(Kg + (size_t)(16 * i + q + 4 * u) * MP + 16 * c + r);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = 16 * i + q + 4 * u, cc = 16 * c + r;
        has_nan |= (kv[u] != kv[u]);
        if (rr == cc) kv[u] = rr < M ? kv[u] + jit : 1.0;   // jitter on the diagonal, identity on the padding
      }
    } else if (zl) {
      const int cc = 16 * c + r;
      double e[4];
      TGP_EACH(u, 4) {
        const int rr = 16 * i + q + 4 * u;
        double d2 = 0.0;
        // DP is 4, 8 or 16 and the rows are 32-byte aligned: two 16-byte reads per operand and four dimensions per trip
        for (int d = 0; d < DP; d += 4) {
          const double2 a0 = *reinterpret_cast<const double2*>(zs + rr * DP + d), a1 = *reinterpret_cast<const double2*>(zs + rr * DP + d + 2);
          const double2 b0 = *reinterpret_cast<const double2*>(zs + cc * DP + d), b1 = *reinterpret_cast<const double2*>(zs + cc * DP + d + 2);
          const double t0 = a0.x - b0.x, t1 = a0.y - b0.y, t2 = a1.x - b1.x, t3 = a1.y - b1.y;
          d2 += t0 * t0; d2 += t1 * t1; d2 += t2 * t2; d2 += t3 * t3;
        }
        e[u] = -0.5 * d2;
      }
      exp_fast_n<4>(e);
      TGP_EACH(u, 4) {
        const int rr = 16 * i + q + 4 * u;
        double k = s2 * e[u];
        has_nan |= (rr < M) && (k != k);
        if (rr == cc) k += jit;
        kv[u] = rr < M ? k : (rr == cc ? 1.0 : 0.0);   // identity on the padding
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) kv[u] = kmm_elem(16 * i + q + 4 * u, 16 * c + r);
    }
    if (ncol > 0) {
      const d4 upd = ll_sum(16 * i, 16 * c, 0, 16 * ncol);
#pragma unroll
      for (int u = 0; u < 4; ++u) kv[u] -= upd[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) A[(16 * i + q + 4 * u) * LD + 16 * c + r] = kv[u];
  };
#define PREP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef TGP_STAMPS
  // diagnostic build: shader-clock sums of the chain's pieces (wave 0) and of one helper's window work (wave 2)
  double tph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  double tcnt[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tl_ = 0;
#define PSTAMP0() do { __builtin_amdgcn_sched_barrier(0); tl_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long tn_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); tph[i] += (double)(tn_ - tl_); tl_ = tn_; } while (0)
#else
#define PSTAMP0() do { } while (0)
#define PSTAMP(i) do { } while (0)
#endif
  // psd_safe_cholesky on the device (dsp/utils.py:256-269): when md.jitter_ladder > 0 a failed factorisation is
  // repeated with jitter_ladder * 10^i, i = 0..2, added to the diagonal, without the host; status[2] reports the
  // level that succeeded (0 = none needed) so that the caller can issue the reference's warning lazily.
  int attempt = 0;
  PSTAMP0();
  for (;; ++attempt) {
    // block column 0 of K_MM heads the chain: one tile per wave (MT <= 8 waves); column j + 1 is filled under potrf(j)
    if (wave < MT) fill_tile(wave, 0, 0);
    PREP_BARRIER();
    PSTAMP(0);
    for (int j = 0; j < MT; ++j) {
      const int j0 = 16 * j;
      const int npan = (MT - 1 - j) * 16;          // rows below the diagonal tile
      const int npw = npan > 64 ? 2 : (npan > 0 ? 1 : 0);  // panel waves: 0 (rows 0..63 below the tile), 1 (the rest)
      double ltile[4] = {0.0, 0.0, 0.0, 0.0};
      bool did_diag = false;
      // ---- phase P ----
      if (wave < npw || (npw == 0 && wave == 0)) {
        // block column j of L: this wave's own copy of the diagonal tile + 64 rows of the panel below it; wave 0 owns L_jj.
        // (Round 4: the tile's inverse is no longer carried in this pass -- nothing on the chain needs it any more, a task
        //  wave forms it one window later (inv_diag) -- so the pass is 3.4 k cycles instead of 4.3 k, and the 48 live
        //  doubles of the three-array form no longer spill inside the chain.)
        __builtin_amdgcn_s_setprio(3);   // the chain: wins instruction issue against its SIMD partner's task work
        const int li = lane & 15, l0 = wave * 64 + lane;
        const bool has = l0 < npan;
        const int prow = j0 + (npan > 0 ? 16 : 0) + (has ? l0 : 0);
        double dg[16], a[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) dg[c] = A[(j0 + li) * LD + j0 + c];
        if (npan > 0) {
#pragma unroll
          for (int c = 0; c < 16; ++c) a[c] = A[prow * LD + j0 + c];
        }
#ifdef TGP_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        PSTAMP(1);
        int bad = 0;
        if (npan > 0) bad = potrf_panel16<true>(dg, a);
        else bad = potrf_panel16<false>(dg, a);
#ifdef TGP_STAMPS
        asm volatile("" ::"v"(dg[15]), "v"(a[15]));
#endif
        PSTAMP(2);
        if (has && npan > 0) {
#pragma unroll
          for (int c = 0; c < 16; ++c) A[prow * LD + j0 + c] = a[c];
        }
        if (wave == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u)   // the four 16-lane rows hold the same dg[]: row q keeps the columns 4u + q
            ltile[u] = q == 0 ? dg[4 * u] : (q == 1 ? dg[4 * u + 1] : (q == 2 ? dg[4 * u + 2] : dg[4 * u + 3]));
          did_diag = true;   // L_jj goes to LDS after the window's barrier: the other panel wave reads the tile's input
          if (lane == 0 && bad != 0 && s_info == 0) s_info = j0 + bad;
        }
#ifdef TGP_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        PSTAMP(3);
        __builtin_amdgcn_s_setprio(0);
#ifdef TGP_STAMPS
        if (wave != 0) { tph[j == 0 ? 7 : 8] += tph[1] + tph[2] + tph[3]; tph[1] = tph[2] = tph[3] = 0.0; }
#endif
      } else if (wave != 4 || j < TGP_W4_WINDOWS) {
        // (wave 4 shares wave 0's SIMD -- a workgroup's waves go to the SIMDs cyclically -- and stays out of the windows:
        //  the chain wave is bound by instruction issue and gets the SIMD to itself)
        // The other waves take this window's tasks from a counter in LDS, heaviest first (all operands are final since
        // the last barrier and the tasks of one window are independent of each other, so any wave may run any of them;
        // dealt round-robin the windows were as long as the unluckiest wave's share):
        //   the inverse of diagonal tile j-1 (its L_jj has been in LDS since the last barrier)
        //   tile (i, j+1)  = K_MM - block columns 0 .. j-1 (its first and only catch-up value: what is left for the
        //                    chain is phase U, - block column j)                        i = j+1 .. MT-1
        //   write-out of L / L^T row j-1
        // (round 4: no tile of J = L^-1 is formed here any more -- the row kernel solves with L itself, and J for the
        //  backward M x M chain comes from its passenger blocks; and a column is filled ONE window before it is due,
        //  6, 5, 4 ... tiles per window, instead of two columns under potrf(0): that first window was twice the chain's)
        const int nt = (j >= 1 && ((j - 1) & 1) == cb) ? 1 : 0;   // tile column j-1 of this block's parity
        const int nf = MT - 1 - j;
        const int nwl = (j + 1 - cb) / 2;                         // (tile columns c = 2 t + cb below j)
        const int ntask = nt + nf + nwl;
        for (;;) {
          int t = 0;
          if (lane == 0) t = atomicAdd(&s_next, 1);
          t = __builtin_amdgcn_readfirstlane(t) - tbase;
          if (t >= ntask) break;
          if (t < nt) { inv_diag(j - 1); continue; }
          t -= nt;
          if (t < nf) { fill_tile(j + 1 + t, j + 1, j); continue; }
          t -= nf;
          write_L(j - 1, 2 * t + cb);
        }
#ifdef TGP_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (j == 0) PSTAMP(7); else PSTAMP(8);
#endif
      }
      tbase += ((j >= 1 && ((j - 1) & 1) == cb) ? 1 : 0) + (MT - 1 - j) + (j + 1 - cb) / 2 +
               (NW - (j < TGP_W4_WINDOWS ? 0 : 1) - (npw > 0 ? npw : 1));   // the tasks + one over-grab per task wave
      PREP_BARRIER();
      PSTAMP(4);
      if (did_diag) {
#pragma unroll
        for (int u = 0; u < 4; ++u) A[(j0 + (lane & 15)) * LD + j0 + 4 * u + q] = ltile[u];
      }
      // ---- phase U: block column j+1 -= L(:, j) L(j+1, j)^T, one tile per wave ----
      if (j + 1 < MT) {
        for (int i = j + 1 + wave; i < MT; i += NW) sub16(i, j + 1, j0);
#ifdef TGP_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        PSTAMP(5);
        PREP_BARRIER();
        PSTAMP(6);
      }
    }
    if (has_nan) s_nan = 1;
    __syncthreads();
    if (s_info == 0 || s_nan != 0 || !(md.jitter_ladder > 0.0) || attempt == 3) break;
    __syncthreads();  // everybody has read s_info
    if (tid == 0) s_info = 0;
    jit = md.jitter + md.jitter_ladder * (attempt == 0 ? 1.0 : (attempt == 1 ? 10.0 : 100.0));
    __syncthreads();
  }
  // ---- tail: the last diagonal tile's inverse, the last tile row of L / L^T (tile columns c = 2 t + cb: of the MT columns
  //      this block owns (MT + 1 - cb) / 2) ----
  if (((MT - 1) & 1) == cb && wave == NW - 1) inv_diag(MT - 1);
  for (int t = wave; t < (MT + 1 - cb) / 2; t += NW) write_L(MT - 1, 2 * t + cb);
  PSTAMP(9);
#ifdef TGP_STAMPS
  if (lane == 0) {
    if (wave == 0)
      for (int i = 0; i < 10; ++i) ws[p.hdr + H_PSTAMP + i] = tph[i];
    if (wave == 3) for (int i = 1; i <= 5; ++i) { ws[p.hdr + H_PSTAMP + 26 + i] = tph[i]; ws[p.hdr + 20 + i] = tcnt[i]; }
    ws[p.hdr + H_PSTAMP + 10 + 2 * wave] = tph[7];      // this wave's window work, j = 0
    ws[p.hdr + H_PSTAMP + 11 + 2 * wave] = tph[8];      // ... summed over j >= 1 (panel / diagonal-tile / task waves alike)
  }
#endif
  if (tid == 0 && cb == 0) {   // (block 1 arrives at the same three words)
    status[0] = s_sync != 0 ? TGP_STATUS_SYNC_TIMEOUT : s_info;
    if (s_sync != 0) __hip_atomic_fetch_add(status + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky: see report_sync_timeout
    status[1] = s_nan;
    status[2] = s_info == 0 ? attempt : 0;
  }
  sync_leave(sy, nb_total);
#undef PREP_BARRIER
}

size_t prep_a_lds_bytes(const Plan& p) {
  return ((size_t)p.MP * (p.MP + 1) + (p.zs_lds ? (size_t)p.MP * p.DP : 0) + 16) * sizeof(double);
}

// ---------------------------------------------------------------------------------------------------
// k_reduce: sum the per-block slabs of the row kernel; the G tiles are expanded to full symmetric matrices.
// A workgroup owns 256 / TGP_RSPLIT slab elements; thread group `sh` of it sums the sh-th contiguous share of the row
// blocks' slabs (the partial sums rounds 1-4 wrote to memory: same shares, same order inside a share), the groups are
// combined through LDS in share order -- bit-identical to the old four-partial form, but the consumers load one value per
// element instead of four (57 -> 14 KB of G per column block of k_bwd, whose CU pulls ~45 GB/s).
// ---------------------------------------------------------------------------------------------------
#define RED_ELEMS (256 / TGP_RSPLIT)
__global__ __launch_bounds__(256) void k_reduce(Plan p, double* __restrict__ ws) {
#ifdef TGP_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) ws[p.dbg + 200 + 16] = (double)__builtin_amdgcn_s_memrealtime();
#endif
  __shared__ double part_s[TGP_RSPLIT][RED_ELEMS];
  const int tid = threadIdx.x, sh = tid / RED_ELEMS, el = tid % RED_ELEMS;
  const size_t e = (size_t)blockIdx.x * RED_ELEMS + el;
  const bool in = e < p.slab_len;
  const int b0 = (int)((long long)p.nblocks * sh / TGP_RSPLIT), b1 = (int)((long long)p.nblocks * (sh + 1) / TGP_RSPLIT);
  const double* sl = ws + p.slabs + (in ? e : 0);
  // RB slabs per batch, all requested before the first add, the ragged end inside the batch (clamped index, zero
  // weight): the slabs come from the Infinity Cache / HBM at 1-2 us per dependent round trip.  Round 5: 28 per batch (was
  // 16) -- the 10-rows-per-wave row kernel writes 216 slabs at Power size, 54 per share: two round trips, not four.
  constexpr int RB = 28;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int b = b0; b < b1; b += RB) {
    double t[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int bu = b + u < b1 ? b + u : b1 - 1;
      t[u] = sl[(size_t)bu * p.slab_len];
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) t[u] = b + u < b1 ? t[u] : 0.0;
#pragma unroll
    for (int u = 0; u < RB; u += 4) { s0 += t[u]; s1 += t[u + 1]; s2 += t[u + 2]; s3 += t[u + 3]; }
  }
  part_s[sh][el] = b1 > b0 ? (s0 + s1) + (s2 + s3) : 0.0;
  __syncthreads();
  if (sh != 0 || !in) return;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < TGP_RSPLIT; ++k) s += part_s[k][el];      // share order, as the consumers of rounds 1-4 added the partials
  if (e < p.slab_T) {
    // tile t = (ti,tj), ti >= tj, row-major over the lower triangle of tiles
    const int t = (int)(e >> 8), in_t = (int)(e & 255), row = in_t >> 4, col = in_t & 15;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    // Diagonal tiles: the MFMA result (a_i vbar) . a_j is not bitwise equal to (a_j vbar) . a_i, so keep the lower
    // half and mirror it -- G is then exactly symmetric and no element has two writers (run-to-run reproducible).
    if (ti == tj && col > row) return;
    double* G = ws + p.Gp;
    st_wt(&G[(size_t)(ti * 16 + row) * p.MP + tj * 16 + col], s);     // (write-through: k_bwd's column blocks start on these)
    st_wt(&G[(size_t)(tj * 16 + col) * p.MP + ti * 16 + row], s);
  } else {
    st_wt(&ws[p.redp + e], s);
  }
}

__device__ __forceinline__ double red_tail(const Plan& p, const double* __restrict__ ws, size_t e) { return ws[p.redp + e]; }

#define BWD_THREADS 512
#define TGP_PF2 32 /* >= MP / 4 k-steps */

// Adam on element i of the flat buffers (torch.optim.Adam; k_adam_dev's arithmetic, tgp_lik.hip)
__device__ __forceinline__ void adam_elem(const AdamDev& A, long i, double g, double pi, double mi, double vi, double bc1,
                                          double bc2s) {
  const double gi = A.sign * g;
  const double m1 = A.b1 * mi + (1.0 - A.b1) * gi;
  const double v1 = A.b2 * vi + (1.0 - A.b2) * gi * gi;
  A.m[i] = m1;
  A.v[i] = v1;
  A.p[i] = pi - (A.lr / bc1) * m1 / (sqrt(v1) / bc2s + A.eps);
}

// ---------------------------------------------------------------------------------------------------
// k_bwd (round 5; rounds 2-4 ran it as three launches k_bwd12 -> k_bwd34 -> k_bwd5): the M x M backward chain behind the slab
// reduction as ONE launch of 4 MT + 1 workgroups (8 waves) that hand their results on through global memory inside the launch
// (the protocol of tgp_prep.hpp: agent-scope stores, every storing wave drains, barrier, one thread moves the word; producers
// carry the lower block indices; all 29 workgroups at Power size are resident from the start):
//   [0, MT)       column block c : G(:, c) -> LDS, Lbar(:, c) -> LDS, Q(i >= c, c) -> global (mirrored);           Q count += 1
//   [MT, 2 MT)    Lam block c    : the L_q-gradient tiles of block row c, Adam on those rows of Lam in the same threads
//   [2 MT, 4 MT)  row block i, half h : J(:, i) -> LDS and its second-phase J fragments -> registers BEFORE it waits for
//                                  Q count == MT;  its half of the column tiles of Y(i, :) = (J^T Q)(i, :) -> LDS, the part of
//                                  Ks = 1/2 Y J that contracts over those columns, PP partials;                    PP count += 1
//   4 MT          waits for PP count == 2 MT; remaining gradients, scalars, Adam on everything but Lam; the last to leave
//                 (it waits for the others' exit count, zeroes the words, advances the Adam step counter)
// Words: status[6] = Q count (bits 16-23) | PP count (bits 24-31), status[7] = workgroups that left.  What the merge buys: two
// launch boundaries and every load a consumer can issue before its producer is done (J, K_MM, Zs, optimiser state).
// The slab reduction stays a launch of its own: as a fourth role (144 producers, the column blocks prefetching H'^T and L
// under it) it ran 11 us instead of 5.8 -- its 15.5 MB want every CU and full occupancy, not 144 eight-wave workgroups at
// this kernel's register count, and its hand-off (drain + count + poll + uncached G loads) costs more than the boundary.
// ---------------------------------------------------------------------------------------------------
enum { SB_PROG = 0, SB_LEFT = 1 };
// diagnostic build (-DTGP_STAMPS): thread 0 of ONE block per role stamps the 100 MHz clock (tools/probes/stamp_bwd.py)
#ifdef TGP_STAMPS
#define BW_STAMP(on, i) do { if ((on) && threadIdx.x == 0) ws[p.dbg + 200 + (i)] = (double)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BW_STAMP(on, i) do { } while (0)
#endif

__device__ __forceinline__ void bwd_leave(int32_t* sb) {
  __syncthreads();
  if (threadIdx.x == 0) sync_add(sb + SB_LEFT, 1);
}

// Bounded wait of a WORKGROUP for a count of this launch: wave 0 polls, the others wait at the barrier (every polling wave is
// one more uncached request stream to the same line, and the producers' traffic shares that channel).  A wait that runs to
// its bound is reported in status[0].
// A timeout is STICKY: status[3] counts the expired waits since the caller last zeroed it and no launch of the
// library ever clears it -- status[0] is rewritten by the next step's prepare launch, so inside a replayed graph of U steps only
// a timeout of the last step would be seen there.  Both words are written as agent-scope atomics: the final role reads
// status[0] past its L1 before the update (bwd_timed_out).
__device__ __forceinline__ void report_sync_timeout(int32_t* __restrict__ status) {
  __hip_atomic_store(status, (int32_t)TGP_STATUS_SYNC_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(status + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class P>
__device__ __forceinline__ void bwd_wait(const int32_t* w, P pred, int32_t* __restrict__ status) {
  if (threadIdx.x < 64) {
    const int v = sync_wait(w, pred);
    if (v == (int)0x80000000 && threadIdx.x == 0) {
      report_sync_timeout(status);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the report has landed before this workgroup's other waves look for it
    }
  }
  __syncthreads();
}
// true when a wait of THIS step has expired: the prepare launch's (it leaves the value in status[0]) or one of this launch's
// whose role has counted itself since (a role drains its stores, the report among them, before it counts)
__device__ __forceinline__ bool bwd_timed_out(const int32_t* __restrict__ status) {
  return __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int32_t)TGP_STATUS_SYNC_TIMEOUT;
}

// G(:, c) -> LDS, s(c-block) -> LDS, Lbar block zeroed (both column roles), in two steps so that
// the caller can put its own operand requests BEHIND these loads and in front of their first use
// (16-byte loads: two adjacent columns per thread -- what a CU pulls from the Infinity Cache is counted in requests)
typedef double bwd_d2 __attribute__((ext_vector_type(2)));
#define BWD_G_NIT ((8 * TGP_MAX_MT * 16 + BWD_THREADS - 1) / BWD_THREADS)
__device__ __forceinline__ void bwd_g_issue(const Plan& p, const double* __restrict__ ws, int c0,
                                            bwd_d2 (&gv)[BWD_G_NIT], double& sv0) {
  const int MP = p.MP, tid = threadIdx.x;
  const double* __restrict__ Gp = ws + p.Gp;
#pragma unroll
  for (int u = 0; u < BWD_G_NIT; ++u) {
    const int i = tid + u * BWD_THREADS;
    const int ic = i < MP * 8 ? i : 0;
    gv[u] = *reinterpret_cast<const bwd_d2*>(Gp + (size_t)(ic >> 3) * MP + c0 + 2 * (ic & 7));
  }
  sv0 = tid < 16 ? red_tail(p, ws, p.slab_S + c0 + tid) : 0.0;
}
__device__ __forceinline__ void bwd_g_commit(const Plan& p, double* Gs, double* LbL, double* svL,
                                             const bwd_d2 (&gv)[BWD_G_NIT], double sv0) {
  const int MP = p.MP, tid = threadIdx.x;
#pragma unroll
  for (int u = 0; u < BWD_G_NIT; ++u) {
    const int i = tid + u * BWD_THREADS;
    if (i < MP * 8) {
      reinterpret_cast<bwd_d2*>(Gs)[i] = gv[u];
      reinterpret_cast<bwd_d2*>(LbL)[i] = bwd_d2{0.0, 0.0};
    }
  }
  if (tid < 16) svL[tid] = sv0;
}

// 16 x 16 tile product on one wave: A fragments `a` (registers, k-step s_ = a[s_]), B fragments from LDS through fb(s_);
// SWAP: the register fragments are the B operand, the LDS ones the A operand
template <int PF, bool SWAP = false, class FB>
__device__ __forceinline__ d4 tile_mm_regA(const double (&a)[PF], FB fb, int n, d4 acc) {
#pragma unroll
  for (int s0 = 0; s0 < PF; s0 += 8) {
    if (s0 < n) {
      double o[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) o[u] = s0 + u < n ? fb(s0 + u) : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + u < n) acc = SWAP ? TGP_MFMA(o[u], a[s0 + u], acc) : TGP_MFMA(a[s0 + u], o[u], acc);
    }
  }
  return acc;
}

// Lam block c: dELBO/dLam(c-rows, :) = 2 tril(G Lq) - kl (Lq - diag(1/Lam_ii)) (strict upper = 0), and Adam on those rows
//   X(j, c) = sum_{k >= j} Lq[k, j]^T G[k, c]  ==  (G Lq)(c, j)^T ; rows of dLam = block c, cols = block j
__device__ __forceinline__ void bwd_lam_role(const Plan& p, const tgp_model& md, const tgp_grads& g, double* __restrict__ ws,
                                             const AdamDev& ad, double* sm, int c, const int32_t* __restrict__ status) {
  static_assert(TGP_MAX_MT <= BWD_THREADS / 64, "one tile per wave");
  const int MP = p.MP, MT = p.MT, M = p.M;
  double* Gs = sm;                     // MP x 16
  double* LbL = Gs + (size_t)MP * 16;  // MP x 16 (unused here)
  double* svL = LbL + (size_t)MP * 16; // 16
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  const int c0 = 16 * c;
  BW_STAMP(c == 0, 4);
  bwd_d2 gv[BWD_G_NIT];
  double sv0;
  bwd_g_issue(p, ws, c0, gv, sv0);
  // ---- behind the G requests: this wave's L_q fragments, its four elements of the factor, their optimiser state ----
  constexpr int PF = TGP_PF2;
  const double* __restrict__ Lq = ws + p.Lq;
  const int j = wave, j0 = 16 * j;
  const bool has = j < MT;
  const int n = has && j <= c ? (MP - j0) / 4 : 0;
  double lq[PF];
#pragma unroll
  for (int s_ = 0; s_ < PF; ++s_) lq[s_] = s_ < n ? Lq[(size_t)(j0 + 4 * s_ + q) * MP + j0 + r] : 0.0;
  const bool upd = ad.p != nullptr && !bwd_timed_out(status);   // (the prepare launch's report; this role waits for nothing itself)
  const double step = upd ? (double)(ad.step_dev[0] + 1) : 1.0;
  double lamv[4], am[4], av[4];
  long ei[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int row = c0 + r, col = j0 + q + 4 * rr;
    const bool in = has && row < M && col < M;
    const size_t e = in ? (size_t)row * M + col : 0;
    ei[rr] = in ? (long)e : -1;
    lamv[rr] = md.Lam[e];
    if (upd) { am[rr] = ad.m[ad.lam_off + e]; av[rr] = ad.v[ad.lam_off + e]; }
  }
  bwd_g_commit(p, Gs, LbL, svL, gv, sv0);
  __syncthreads();
  d4 acc = {0, 0, 0, 0};
  acc = tile_mm_regA<PF>(lq, [&](int s_) { return Gs[(j0 + 4 * s_ + q) * 16 + r]; }, n, acc);
  const double bc1 = 1.0 - exp_fast(step * ad.ln_b1), bc2s = sqrt(1.0 - exp_fast(step * ad.ln_b2));
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    if (ei[rr] < 0) continue;
    const int row = c0 + r, col = j0 + q + 4 * rr;  // transposed store
    double x = 0.0;
    if (col <= row) x = 2.0 * acc[rr] - md.kl_scale * (col == row ? lamv[rr] - 1.0 / lamv[rr] : lamv[rr]);
    g.Lam[ei[rr]] = x;
    if (upd) adam_elem(ad, ad.lam_off + ei[rr], x, ad.p[ad.lam_off + ei[rr]], am[rr], av[rr], bc1, bc2s);
  }
  BW_STAMP(c == 0, 5);
}

// column block c:  Lbar(:, c) = -tril(w s^T + 2 H' G)(:, c) -> LDS;  Q(i, c) = [Phi(L^T Lbar) + Phi(L^T Lbar)^T](i, c), i >= c
// -> global, mirrored
__device__ __forceinline__ void bwd_q_role(const Plan& p, double* __restrict__ ws, double* sm, int c, int32_t* sb) {
  static_assert(TGP_MAX_MT <= BWD_THREADS / 64, "one tile per wave");
  const int MP = p.MP, MT = p.MT;
  double* Gs = sm;                     // MP x 16
  double* LbL = Gs + (size_t)MP * 16;  // MP x 16
  double* svL = LbL + (size_t)MP * 16; // 16
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  const int c0 = 16 * c;
  BW_STAMP(c == 0, 0);
  bwd_d2 gv[BWD_G_NIT];
  double sv0;
  bwd_g_issue(p, ws, c0, gv, sv0);
  // ---- behind the G requests and in front of their first use: the operands of BOTH products of this wave's tile (rows
  //      i0 = 16 (c + wave)) -- 100 KB of H'^T and L fragments per workgroup at ~30 GB/s per CU from the Infinity Cache used to
  //      sit on the chain behind the staging barrier
  constexpr int PF = TGP_PF2;
  const double* __restrict__ HpT = ws + p.HpT;
  const double* __restrict__ Lm = ws + p.L;
  const double* __restrict__ w = ws + p.w;
  // wave t < MT - c owns tile row c + t; the first wave without a tile (there is one unless MT - c = 8) takes the TRANSPOSED
  // product of the diagonal tile off wave 0, which would otherwise run two products back to back on the chain
  const bool has = wave < MT - c;
  const bool spare = MT - c < BWD_THREADS / 64, tw = spare && wave == MT - c;
  const int i0 = tw ? c0 : 16 * (c + wave);
  const int n1 = has ? MP / 4 : 0, n2 = has || tw ? (MP - i0) / 4 : 0;
  double hp[PF], pq[PF], wv[4];
#pragma unroll
  for (int s_ = 0; s_ < PF; ++s_) hp[s_] = s_ < n1 ? HpT[(size_t)(4 * s_ + q) * MP + i0 + r] : 0.0;
#pragma unroll
  for (int s_ = 0; s_ < PF; ++s_) pq[s_] = s_ < n2 ? Lm[(size_t)(i0 + 4 * s_ + q) * MP + i0 + r] : 0.0;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) wv[rr] = has ? w[i0 + q + 4 * rr] : 0.0;
  bwd_g_commit(p, Gs, LbL, svL, gv, sv0);
  __syncthreads();
  BW_STAMP(c == 0, 1);
  if (has) {
    d4 acc = {0, 0, 0, 0};
    acc = tile_mm_regA<PF>(hp, [&](int s_) { return Gs[(4 * s_ + q) * 16 + r]; }, n1, acc);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = i0 + q + 4 * rr, col = c0 + r;
      LbL[row * 16 + r] = (col <= row) ? -(wv[rr] * svL[r] + 2.0 * acc[rr]) : 0.0;
    }
  }
  __syncthreads();
  BW_STAMP(c == 0, 2);
  // ---- Q(i, c) = Phi(M1) + Phi(M1)^T with M1 = L^T Lbar ;  (L^T)[i,k] = L[k,i] = 0 for k < i ----
  double* Q = ws + p.Q;
  if (has) {
    d4 acc = {0, 0, 0, 0};
    acc = tile_mm_regA<PF>(pq, [&](int s_) { return LbL[(i0 + 4 * s_ + q) * 16 + r]; }, n2, acc);
    if (wave != 0) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = i0 + q + 4 * rr, col = c0 + r;
        st_agent(Q + (size_t)row * MP + col, acc[rr]);
        st_agent(Q + (size_t)col * MP + row, acc[rr]);
      }
    } else {
      d4 tr = {0, 0, 0, 0};  // M1^T tile = Lbar^T L: the same L fragments as the B operand
      if (!spare) tr = tile_mm_regA<PF, true>(pq, [&](int s_) { return LbL[(c0 + 4 * s_ + q) * 16 + r]; }, n2, tr);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int rl = q + 4 * rr;
        if (r <= rl) st_agent(Q + (size_t)(c0 + rl) * MP + c0 + r, acc[rr]);
        else if (!spare) st_agent(Q + (size_t)(c0 + rl) * MP + c0 + r, tr[rr]);
      }
    }
  } else if (tw) {
    d4 tr = {0, 0, 0, 0};
    tr = tile_mm_regA<PF, true>(pq, [&](int s_) { return LbL[(c0 + 4 * s_ + q) * 16 + r]; }, n2, tr);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int rl = q + 4 * rr;
      if (r > rl) st_agent(Q + (size_t)(c0 + rl) * MP + c0 + r, tr[rr]);
    }
  }
  handoff_barrier();   // every wave's Q stores have landed before the count moves
  if (tid == 0) sync_add(sb + SB_PROG, 1 << 16);
  BW_STAMP(c == 0, 3);
}

// row block i: Y(i, :) = (J^T Q)(i, :) -> LDS;  Ks(i, j) = 1/2 (Y J)(i, j) = dELL/dK_MM (never stored);
// PP[i][col][d] = sum_{rows in block i} (Ks o K_MM)[row][col] * [Zs[row][d], 1]   (ARD-RBF parameter partials)
__device__ __forceinline__ void bwd_row_role(const Plan& p, double* __restrict__ ws, double* sm, int i, int h, int32_t* sb,
                                             int32_t* __restrict__ status) {
  static_assert(TGP_MAX_MT <= BWD_THREADS / 64, "one Y tile and one Ks tile per wave");
  const int MP = p.MP, MT = p.MT, DP = p.DP, PPW = p.PPW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  // Two workgroups per row block (h = 0, 1): each forms HALF of the column tiles of Y(i, :) -- half of the 100 KB of Q that
  // row block 0 pulls at ~45 GB/s per CU -- and contracts the second product over exactly those columns; the two partial
  // results are two partials more for the final block's sum.
  const int KH = (MT + 1) / 2, kb0 = h ? KH : 0, kb1 = h ? MT : KH, nk = kb1 - kb0;
  double* Yl = sm;                               // nk x 256 (<= MT x 256 reserved)
  double* zsL = Yl + (size_t)MT * 256;           // 16 x DP
  double* Ja = zsL + 16 * DP;                    // (MP - i0) x 16: block column i of J from its diagonal tile down
  const int i0 = 16 * i;
  const double* __restrict__ J = ws + p.J;
  const double* __restrict__ Q = ws + p.Q;
  // ---- everything that does not depend on this launch's Q: requested before the wait ----
  BW_STAMP(i == 0 && h == 0, 6);
  double kmv[4];
  const int jbw = wave < MT ? wave : 0, j0w = 16 * jbw;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) kmv[rr] = (ws + p.Kmm)[(size_t)(i0 + q + 4 * rr) * MP + j0w + r];
  if (tid < 16 * DP) zsL[tid] = (ws + p.Zs)[(size_t)i0 * DP + tid];
  for (int e = tid; e < (MP - i0) * 16; e += BWD_THREADS) Ja[e] = J[(size_t)(i0 + (e >> 4)) * MP + i0 + (e & 15)];
  // second product of wave j: k over the rows of this half's column tiles at or below tile j (J[k, j] = 0 for k < j)
  const int ks0 = 16 * (jbw > kb0 ? jbw : kb0), ks1 = 16 * kb1;
  const int n1 = wave < nk ? (MP - i0) / 4 : 0, n2 = wave < MT && ks1 > ks0 ? (ks1 - ks0) / 4 : 0;
  constexpr int PF = TGP_PF2, PFH = 2 * TGP_MAX_MT;      // k-steps of a full column / of half of the tile rows
  double jb2[PFH];
#pragma unroll
  for (int s_ = 0; s_ < PFH; ++s_) jb2[s_] = s_ < n2 ? J[(size_t)(ks0 + 4 * s_ + q) * MP + j0w + r] : 0.0;
  __syncthreads();
  BW_STAMP(i == 0 && h == 0, 7);
  bwd_wait(sb + SB_PROG, [&](int x) { return ((x >> 16) & 0xff) >= MT; }, status);
  BW_STAMP(i == 0 && h == 0, 8);
  if (wave < nk) {
    // Y tile (i, kb = kb0 + wave): all of the wave's Q fragments in one round trip
    d4 acc = {0, 0, 0, 0};
    double qv[PF];
#pragma unroll
    for (int s_ = 0; s_ < PF; ++s_) qv[s_] = s_ < n1 ? ld_agent(Q + (size_t)(i0 + 4 * s_ + q) * MP + 16 * (kb0 + wave) + r) : 0.0;
    acc = tile_mm_regA<PF, true>(qv, [&](int s_) { return Ja[(4 * s_ + q) * 16 + r]; }, n1, acc);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Yl[wave * 256 + (q + 4 * rr) * 16 + r] = acc[rr];
  }
  __syncthreads();
  BW_STAMP(i == 0 && h == 0, 9);
  if (wave < MT) {
    d4 acc = {0, 0, 0, 0};
    acc = tile_mm_regA<PFH, true>(jb2, [&](int s_) { const int k = ks0 + 4 * s_; return Yl[((k >> 4) - kb0) * 256 + r * 16 + (k & 15) + q]; }, n2, acc);
    double ep[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) ep[rr] = 0.5 * acc[rr] * kmv[rr];
    double cs = quad_sum((ep[0] + ep[1]) + (ep[2] + ep[3]));
    // PP[part = 2 i + h][c][col]: c < DP the Zs-weighted sums, c = DP the plain column sum -- columns contiguous, so that the
    // 16 lanes of a store and the 64 lanes of the final block's loads touch whole lines
    double* out = ws + p.PP + (size_t)(2 * i + h) * PPW * MP + j0w + r;
    if (q == 0) st_agent(out + (size_t)DP * MP, cs);
    for (int d = 0; d < DP; ++d) {
      double s = 0.0;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) s += ep[rr] * zsL[(q + 4 * rr) * DP + d];
      s = quad_sum(s);
      if (q == 0) st_agent(out + (size_t)d * MP, s);
    }
  }
  handoff_barrier();
  if (tid == 0) sync_add(sb + SB_PROG, 1 << 24);
  BW_STAMP(i == 0 && h == 0, 10);
}

// the remaining gradients + the scalars + Adam on everything but the q(u) factor (everything here is O(M D))
#define BWD_MIRROR_MAX 4096 /* doubles of LDS the final role may spend on the gradient mirror (host sizing and device test agree) */
#define BWDF_ADAM_PER_THREAD 2 /* (n - M^2) / BWD_THREADS rounded up: 530 at Power; more falls back to a loop */
__device__ __forceinline__ void bwd_final_role(const Plan& p, const tgp_model& md, const tgp_grads& g, double* __restrict__ out,
                                               double* __restrict__ ws, const AdamDev& ad, double* term, int32_t* sb,
                                               int32_t* __restrict__ status, int nb_total) {
  constexpr int NT = BWD_THREADS;
  const int tid = threadIdx.x;
  BW_STAMP(true, 11);
  const long n_rest = ad.p != nullptr ? ad.n - ad.lam_n : 0;
  double ap[BWDF_ADAM_PER_THREAD], am[BWDF_ADAM_PER_THREAD], av[BWDF_ADAM_PER_THREAD];
  double a_step = 0.0;
  if (ad.p != nullptr) {
    a_step = (double)(ad.step_dev[0] + 1);
#pragma unroll
    for (int u = 0; u < BWDF_ADAM_PER_THREAD; ++u) {
      const long k = tid + (long)NT * u;
      const long i = k < n_rest ? (k < ad.lam_off ? k : k + ad.lam_n) : 0;
      ap[u] = ad.p[i]; am[u] = ad.m[i]; av[u] = ad.v[i];
    }
  }
  const int M = p.M, D = p.D, MP = p.MP, DP = p.DP, CT16 = p.CT16, PPW = p.PPW, MT = p.MT;
  const double* hdr = ws + p.hdr;
  const double s2 = hdr[H_S2];
  const double* Zs = ws + p.Zs;
  // The gradients this workgroup forms are mirrored in LDS at their position in the flat buffer, so that the update below reads
  // them from there instead of draining its stores and fetching them back past the L1 (1.3 us of the chain).  Only when every
  // element of the flat buffer outside Lam is one of them (the engine's layout); any other caller keeps the global round trip.
  double* gl = term + (size_t)M * (D + 1);
  const long nz = (long)M * D, npar = nz + D + 1 + M + 1 + (g.theta != nullptr ? p.P : 0);
  auto inside = [&](const double* q_, long len) { return q_ >= ad.g && q_ + len <= ad.g + ad.n; };
  const bool mirror = ad.p != nullptr && npar == n_rest && n_rest <= BWD_MIRROR_MAX && inside(g.Z, nz) && inside(g.raw_ls, D) && inside(g.raw_os, 1) &&
                      inside(g.m, M) && inside(g.log_var_noise, 1) && (g.theta == nullptr || inside(g.theta, p.P));
  auto put = [&](double* base, long i, double val) {
    base[i] = val;
    if (mirror) {
      const long idx = (base + i) - ad.g;
      gl[idx < ad.lam_off ? idx : idx - ad.lam_n] = val;
    }
  };
  // ---- everything that depends on k_reduce only: before the wait ----
  for (int i = tid; i < M; i += NT) put(g.m, i, red_tail(p, ws, p.slab_S + i) - md.kl_scale * md.m[i]);
  if (g.theta != nullptr)
    for (int i = tid; i < p.P; i += NT) put(g.theta, i, red_tail(p, ws, p.slab_C + C_THETA + i));
  // (the scalars of the last column sum and the lengthscale factors of the others: wave d, lane 0 uses them)
  const int dcol = tid >> 6;
  const double c_svb = red_tail(p, ws, p.slab_C + C_SVB), c_etab = red_tail(p, ws, p.slab_C + C_ETAB),
               c_ell = red_tail(p, ws, p.slab_C + C_ELL), c_kl = hdr[H_KL], c_sig = hdr[H_SIG_OS];
  const double c_ils = ws[p.ils + (dcol < D ? dcol : 0)], c_rls = md.raw_ls[dcol < D ? dcol : 0];
  // item it = d' M + j: column j of the plain sums (d' = 0 -> d = D: they come first, the others need their result) or of the
  // Zs-weighted sums (d = d' - 1)
  const int nitems = M * (D + 1);
  double t0f = 0.0, t1f = 0.0, t2f = 0.0, zjf = 0.0;
  if (tid < nitems) {       // the first item of this thread (the only one up to 512 items)
    const int j = tid % M, d = tid / M == 0 ? D : tid / M - 1, dd = d < D ? d : 0;
    t0f = red_tail(p, ws, p.slab_T + (size_t)j * CT16 + 2 * DP);
    t1f = red_tail(p, ws, p.slab_T + (size_t)j * CT16 + dd);
    t2f = red_tail(p, ws, p.slab_T + (size_t)j * CT16 + DP + dd);
    zjf = Zs[j * DP + dd];
  }
  bwd_wait(sb + SB_PROG, [&](int x) { return (x >> 24) >= 2 * MT; }, status);
  // a hand-off of this step expired (this wait, a row role's, the prepare launch's): the gradients below are built from stale
  // operands -- no update, no step, NaN scalars; status[3] keeps the event for the host
  const bool timed_out = bwd_timed_out(status);
  BW_STAMP(true, 12);
  double* csL = gl + (mirror ? n_rest : 0);      // M: the plain column sums cs_j, formed by the d = D items for the others
  for (int base = 0; base < nitems; base += NT) {
    const int it = base + tid;
    const bool act = it < nitems;
    const int j = act ? it % M : 0, d = act ? (it / M == 0 ? D : it / M - 1) : 0;
    // the 4 MT partials of this item (two per row block), requested before the first one is used: row c = d of the partial
    // sums (c = DP: the plain sums) -- ONE row per item; cs_j reaches the d < D items through LDS, not through 4 MT more loads
    double pv[2 * TGP_MAX_MT];
#pragma unroll
    for (int ib = 0; ib < 2 * TGP_MAX_MT; ++ib)
      pv[ib] = ld_agent(ws + p.PP + ((size_t)(ib < 2 * MT ? ib : 0) * PPW + (d < D ? d : DP)) * MP + j);
    const bool first = base == 0;
    const int dd = d < D ? d : 0;
    const double t0 = first ? t0f : red_tail(p, ws, p.slab_T + (size_t)j * CT16 + 2 * DP);
    const double t1 = first ? t1f : red_tail(p, ws, p.slab_T + (size_t)j * CT16 + dd);
    const double t2 = first ? t2f : red_tail(p, ws, p.slab_T + (size_t)j * CT16 + DP + dd);
    double sv = 0.0;
#pragma unroll
    for (int ib = 0; ib < 2 * TGP_MAX_MT; ++ib) sv += ib < 2 * MT ? pv[ib] : 0.0;
    if (act && d == D) { csL[j] = sv; term[d * M + j] = sv + t0; }
    __syncthreads();
    if (act && d < D) {
      const double cs = csL[j], R = sv;
      const double zj = first ? zjf : Zs[j * DP + d];
      // dELL/dzs_jd = [T1 - zs T0] (rows) + 2 sum_i Ep_ij (zs_id - zs_jd) (K_MM, Ep symmetric)
      put(g.Z, j * D + d, (t1 - zj * t0 + 2.0 * (R - zj * cs)) * ws[p.ils + d]);
      // lengthscale: sum_n E (xs - zs)^2 + sum_ij Ep_ij (zs_id - zs_jd)^2 ; second = 2 sum_j zs_jd (zs_jd cs_j - R_jd)
      term[d * M + j] = (t2 - 2.0 * zj * t1 + zj * zj * t0) + 2.0 * zj * (zj * cs - R);
    }
  }
  __syncthreads();
  BW_STAMP(true, 13);
  // column sums of term[M][D+1]: one wave per column
  for (int d = tid >> 6; d <= D; d += NT / 64) {
    double s = 0.0;
    for (int j = tid & 63; j < M; j += 64) s += term[d * M + j];
    s = wave_sum(s);
    if ((tid & 63) != 0) continue;
    if (d < D) {
      const bool mine = d == dcol;      // (D > 8: a wave takes more than one column; only its first is prefetched)
      put(g.raw_ls, d, s * (mine ? c_ils : ws[p.ils + d]) * sigmoid_d(mine ? c_rls : md.raw_ls[d]));
    } else {
      const double s2b = c_svb + s / s2;
      put(g.raw_os, 0, s2b * c_sig);
      put(g.log_var_noise, 0, c_etab);
      const double ell = c_ell, kl = c_kl, nan_ = __builtin_nan("");
      out[0] = timed_out ? nan_ : ell - kl;
      out[1] = timed_out ? nan_ : ell;
      out[2] = timed_out ? nan_ : kl;
      out[3] = 0.0;
    }
  }
  if (ad.p != nullptr && !timed_out) {
    if (mirror) __syncthreads();
    else handoff_barrier();  // this workgroup's gradient stores have landed
    const double bc1 = 1.0 - exp_fast(a_step * ad.ln_b1), bc2s = sqrt(1.0 - exp_fast(a_step * ad.ln_b2));
    // (without the mirror: the gradient was stored by another thread of this workgroup a moment ago -- read it past the CU's L1)
    auto grad = [&](long k, long i) { return mirror ? gl[k] : __hip_atomic_load(ad.g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
#pragma unroll
    for (int u = 0; u < BWDF_ADAM_PER_THREAD; ++u) {
      const long k = tid + (long)NT * u;
      if (k < n_rest) {
        const long i = k < ad.lam_off ? k : k + ad.lam_n;
        adam_elem(ad, i, grad(k, i), ap[u], am[u], av[u], bc1, bc2s);
      }
    }
    for (long k = tid + (long)NT * BWDF_ADAM_PER_THREAD; k < n_rest; k += NT) {
      const long i = k < ad.lam_off ? k : k + ad.lam_n;
      adam_elem(ad, i, grad(k, i), ad.p[i], ad.m[i], ad.v[i], bc1, bc2s);
    }
  }
  BW_STAMP(true, 14);
  // the last to leave: every other workgroup of the launch has counted itself out (the Lam blocks have read the Adam step
  // counter long ago, nobody polls the words any more) -> zero the words for the next launch, advance the counter
  bwd_wait(sb + SB_LEFT, [&](int x) { return x >= nb_total - 1; }, status);
  __syncthreads();   // EVERY wave of this workgroup has seen the count before it is zeroed (a wave still polling would never see it again)
  if (tid == 0) {
    sync_st(sb + SB_PROG, 0); sync_st(sb + SB_LEFT, 0);
    if (ad.p != nullptr && !timed_out) atomicAdd(&ad.step_dev[0], 1);
  }
  BW_STAMP(true, 15);
}

__global__ __launch_bounds__(BWD_THREADS) void k_bwd(Plan p, tgp_model md, tgp_grads g, double* __restrict__ out,
                                                     double* __restrict__ ws, AdamDev ad, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int MT = p.MT, b = (int)blockIdx.x;
  int32_t* sb = status + 6;
  if (b < MT) {
    bwd_q_role(p, ws, sm, b, sb);
    bwd_leave(sb);
  } else if (b < 2 * MT) {
    bwd_lam_role(p, md, g, ws, ad, sm, b - MT, status);
    bwd_leave(sb);
  } else if (b < 4 * MT) {
    bwd_row_role(p, ws, sm, (b - 2 * MT) >> 1, (b - 2 * MT) & 1, sb, status);
    bwd_leave(sb);
  } else {
    bwd_final_role(p, md, g, out, ws, ad, sm, sb, status, 4 * MT + 1);
  }
}

// ---------------------------------------------------------------------------------------------------
// stand-alone entry points that live on the M x M side
// ---------------------------------------------------------------------------------------------------
// K(X1, X2) behind tgp_kmm_f64 / tgp_knm_f64 / tgp_kernel_matrix_f64 (gpytorch ScaleKernel(RBFKernel | MaternKernel) as
// instance_kernel builds them, models/utils_models.py:188-204; call sites models/sparse_MF_SP.py:313-319).  Both kernels
// below are bound by their output (8 N1 N2 bytes written against (N1 + N2) D read; profiles/r03_pmc_hbm_standalone.csv):
//   * the inverse lengthscales 1/softplus(raw_ls) and the outputscale are formed once per block, not per element;
//   * the block's scaled X1 rows are staged in LDS with coalesced loads (their D values are contiguous in X1);
//   * 16-byte stores (8-byte ones when the row stride N2 is odd); RBF: four exponentials at a time, stage by stage.
//
// k_cov_tile (any N2): block = 64 rows of X1 x 128 columns (rows of X2); thread = one PAIR of adjacent columns x 16
// rows, its two scaled X2 rows in registers, the X1 rows read back from LDS as broadcasts.
#define COV_ROWS 64
#define COV_COLS 128
__global__ __launch_bounds__(256) void k_cov_tile(int kernel, const double* __restrict__ X1, int N1,
                                                   const double* __restrict__ X2, int N2, int D,
                                                   const double* __restrict__ raw_ls, const double* __restrict__ raw_os,
                                                   double jitter, int self, double* __restrict__ K) {
  __shared__ double xl[COV_ROWS * 16];
  __shared__ double ils[17];
  const int tid = threadIdx.x, cp = tid & 63, rg = tid >> 6;
  const long n0 = (long)blockIdx.x * COV_ROWS;
  const int m0 = blockIdx.y * COV_COLS + 2 * cp;
  if (tid < 16) ils[tid] = tid < D ? 1.0 / softplus_d(raw_ls[tid]) : 0.0;
  if (tid == 64) ils[16] = softplus_d(raw_os[0]);
  // the thread's two X2 rows (clamped: out-of-range columns compute a finite value that is never stored)
  double z0[16], z1[16];
  {
    const long ma = m0 < N2 ? m0 : N2 - 1, mb = m0 + 1 < N2 ? m0 + 1 : N2 - 1;
#pragma unroll
    for (int d = 0; d < 16; ++d) {
      z0[d] = d < D ? X2[ma * D + d] : 0.0;
      z1[d] = d < D ? X2[mb * D + d] : 0.0;
    }
  }
  const long nrem = (long)N1 - n0, nr = nrem < COV_ROWS ? nrem : COV_ROWS;
  double xv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {  // COV_ROWS x D values are one contiguous run of X1: coalesced
    const int i = tid + 256 * u;
    xv[u] = i < nr * D ? X1[n0 * D + i] : 0.0;
  }
  __syncthreads();
  const double s2 = ils[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) { z0[d] *= ils[d]; z1[d] *= ils[d]; }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = tid + 256 * u;
    if (i < COV_ROWS * D) xl[(i / D) * 16 + i % D] = xv[u] * ils[i % D];
  }
  __syncthreads();
  const bool pair_ok = (N2 & 1) == 0;
#pragma unroll 2
  for (int u0 = 0; u0 < 16; u0 += 2) {
    double e[4];
    long rown[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int nl = rg * 16 + u0 + u;
      rown[u] = n0 + nl;
      double da = 0.0, db = 0.0;
      for (int d = 0; d < D; ++d) {
        const double x = xl[nl * 16 + d];
        const double ta = x - z0[d], tb = x - z1[d];
        da += ta * ta;
        db += tb * tb;
      }
      e[2 * u] = da;
      e[2 * u + 1] = db;
    }
    if (kernel == TGP_KERNEL_SCALE_RBF) {
      TGP_EACH(u, 4) e[u] *= -0.5;
      exp_fast_n<4>(e);
      TGP_EACH(u, 4) e[u] *= s2;
    } else {
      TGP_EACH(u, 4) e[u] = cov_value(kernel, s2, e[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (rown[u] >= N1) continue;
      double ka = e[2 * u], kb = e[2 * u + 1];
      if (self) {
        if (rown[u] == m0) ka += jitter;
        if (rown[u] == m0 + 1) kb += jitter;
      }
      double* o = K + rown[u] * (long)N2 + m0;
      if (pair_ok && m0 + 1 < N2) {
        *reinterpret_cast<double2*>(o) = make_double2(ka, kb);
      } else {
        if (m0 < N2) o[0] = ka;
        if (m0 + 1 < N2) o[1] = kb;
      }
    }
  }
}

// k_cov_flat (N2 even, N2 * D <= 4096: K_NM with M = 100 is the case that matters -- a 128-column tile would idle 22 %
// of the lanes): block = COVF_ROWS rows x ALL columns, i.e. one contiguous run of the output; thread = element pairs
// e = 2 tid, 2 tid + 512, ... of that run (row r = e / N2, columns c = e % N2 and c + 1), every lane busy whatever N2 is.
// Both operands come from LDS in dimension-major layout: zT[d][c] (a wave reads 64 consecutive pairs: conflict-free
// 16-byte reads), xT[d][r] (one or two rows per wave: broadcasts).
#define COVF_ROWS 128
__global__ __launch_bounds__(256) void k_cov_flat(int kernel, const double* __restrict__ X1, int N1,
                                                   const double* __restrict__ X2, int N2, int D,
                                                   const double* __restrict__ raw_ls, const double* __restrict__ raw_os,
                                                   double jitter, int self, double* __restrict__ K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* zT = reinterpret_cast<double*>(smem_raw);  // D x N2
  double* xT = zT + (size_t)D * N2;                  // D x COVF_ROWS
  __shared__ double ils[17];
  const int tid = threadIdx.x;
  const long n0 = (long)blockIdx.x * COVF_ROWS;
  const long nrem = (long)N1 - n0;
  const int nr = (int)(nrem < COVF_ROWS ? nrem : COVF_ROWS);
  if (tid < 16) ils[tid] = tid < D ? 1.0 / softplus_d(raw_ls[tid]) : 0.0;
  if (tid == 64) ils[16] = softplus_d(raw_os[0]);
  __syncthreads();
  for (int i = tid; i < N2 * D; i += 256) {
    const int c = i / D, d = i - c * D;
    zT[d * N2 + c] = X2[i] * ils[d];
  }
  for (int i = tid; i < nr * D; i += 256) {
    const int r = i / D, d = i - r * D;
    xT[d * COVF_ROWS + r] = X1[n0 * D + i] * ils[d];
  }
  __syncthreads();
  const double s2 = ils[16];
  const int npairs = nr * (N2 / 2);
  // two pairs (four elements) per trip; (r, c) advance incrementally: no division in the loop
  const int sr = 512 / N2, sc = 512 - sr * N2;
  int e = 2 * tid, r = e / N2, c = e - r * N2;
  double* __restrict__ Ko = K + n0 * (long)N2;
  for (int p = tid; p < npairs; p += 512) {
    int r1 = r + sr, c1 = c + sc;
    if (c1 >= N2) { c1 -= N2; ++r1; }
    const bool has1 = p + 256 < npairs;
    const int rb = has1 ? r1 : r, cb = has1 ? c1 : c;
    double ev[4] = {0.0, 0.0, 0.0, 0.0};
    for (int d = 0; d < D; ++d) {
      const double xa = xT[d * COVF_ROWS + r], xb = xT[d * COVF_ROWS + rb];
      const double2 za = *reinterpret_cast<const double2*>(zT + d * N2 + c);
      const double2 zb = *reinterpret_cast<const double2*>(zT + d * N2 + cb);
      const double t0 = xa - za.x, t1 = xa - za.y, t2 = xb - zb.x, t3 = xb - zb.y;
      ev[0] += t0 * t0; ev[1] += t1 * t1; ev[2] += t2 * t2; ev[3] += t3 * t3;
    }
    if (kernel == TGP_KERNEL_SCALE_RBF) {
      TGP_EACH(u, 4) ev[u] *= -0.5;
      exp_fast_n<4>(ev);
      TGP_EACH(u, 4) ev[u] *= s2;
    } else {
      TGP_EACH(u, 4) ev[u] = cov_value(kernel, s2, ev[u]);
    }
    if (self) {
      if (n0 + r == c) ev[0] += jitter;
      if (n0 + r == c + 1) ev[1] += jitter;
      if (n0 + rb == cb) ev[2] += jitter;
      if (n0 + rb == cb + 1) ev[3] += jitter;
    }
    *reinterpret_cast<double2*>(Ko + e) = make_double2(ev[0], ev[1]);
    if (has1) *reinterpret_cast<double2*>(Ko + e + 512) = make_double2(ev[2], ev[3]);
    // advance by 1024 elements = two steps of 512
    r = r1 + sr; c = c1 + sc;
    if (c >= N2) { c -= N2; ++r; }
    e += 1024;
  }
}

// whitened KL + gradients (models/sparse_MF_SP.py:406-431), single block
__global__ __launch_bounds__(256) void k_kl(const double* __restrict__ m, const double* __restrict__ Lam, int M,
                                             double* __restrict__ out, double* __restrict__ g_m,
                                             double* __restrict__ g_Lam) {
  __shared__ double wsum[4];
  const int tid = threadIdx.x;
  double part = 0.0;
  for (int it = tid; it < M * M; it += 256) {
    const int r = it / M, c = it % M;
    double gl = 0.0;
    if (c <= r) {
      const double x = Lam[it];
      part += x * x;
      gl = x;
      if (c == r) { part -= log(x * x); gl = x - 1.0 / x; }
    }
    if (g_Lam) g_Lam[it] = gl;
  }
  for (int i = tid; i < M; i += 256) {
    part += m[i] * m[i];
    if (g_m) g_m[i] = m[i];
  }
  part = wave_sum(part);
  if ((tid & 63) == 0) wsum[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) out[0] = 0.5 * (wsum[0] + wsum[1] + wsum[2] + wsum[3] - (double)M);
}

// copy the Cholesky factor / inverse out of the padded workspace
__global__ void k_unpad(const double* __restrict__ src, int MP, int M, double* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * M) return;
  dst[i] = src[(size_t)(i / M) * MP + i % M];
}

// K_MM given directly (tgp_cholesky_f64): same factorisation code path as k_prep_a, via a model whose
// kernel matrix is supplied.  Implemented by a dedicated small kernel to keep k_prep_a readable.
__global__ __launch_bounds__(PREP_THREADS) void k_chol_only(const double* __restrict__ Ain, int M, int MP,
                                                            double* __restrict__ Lout, double* __restrict__ Jout,
                                                            int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* A = reinterpret_cast<double*>(smem_raw);
  const int LD = MP + 1;
  double* dinv = A + (size_t)MP * LD;
  __shared__ int s_info, s_nan;
  const int tid = threadIdx.x;
  if (tid == 0) { s_info = 0; s_nan = 0; }
  __syncthreads();
  bool has_nan = false;
  for (int i = tid; i < M * M; i += PREP_THREADS) {
    const double x = Ain[i];
    has_nan |= (x != x);
    A[(i / M) * LD + i % M] = x;
  }
  if (has_nan) s_nan = 1;
  for (int j = 0; j < M; ++j) {
    __syncthreads();
    const double d = A[j * LD + j];
    if (!(d > 0.0)) {
      if (tid == 0 && s_info == 0) s_info = j + 1;
    }
    const double dj = sqrt(d), inv = 1.0 / dj;
    __syncthreads();
    for (int i = j + tid; i < M; i += PREP_THREADS) A[i * LD + j] = (i == j) ? dj : A[i * LD + j] * inv;
    __syncthreads();
    const int rem = M - j - 1;
    for (int e = tid; e < rem * rem; e += PREP_THREADS) {
      const int i = j + 1 + e / rem, k = j + 1 + e % rem;
      if (k <= i) A[i * LD + k] -= A[i * LD + j] * A[k * LD + j];
    }
  }
  __syncthreads();
  if (tid < M) dinv[tid] = 1.0 / A[tid * LD + tid];
  __syncthreads();
  if (Jout != nullptr && tid < M) {
    const int c = tid;
    for (int i = c + 1; i < M; ++i) {
      double s = A[i * LD + c] * dinv[c];
      for (int k = c + 1; k < i; ++k) s += A[i * LD + k] * A[c * LD + k];
      A[c * LD + i] = -s * dinv[i];
    }
  }
  __syncthreads();
  for (int i = tid; i < M * M; i += PREP_THREADS) {
    const int r = i / M, c = i % M;
    double l = 0.0, jv = 0.0;
    if (c < r) { l = A[r * LD + c]; jv = A[c * LD + r]; }
    else if (c == r) { l = A[r * LD + r]; jv = dinv[r]; }
    Lout[i] = l;
    if (Jout != nullptr) Jout[i] = jv;
  }
  if (tid == 0) { status[0] = s_info; status[1] = s_nan; }
}

// ---------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------
#define LAUNCH_CHECK()                                   \
  do {                                                   \
    hipError_t e_ = hipGetLastError();                   \
    if (e_ != hipSuccess) return set_error(e_, __FILE__, __LINE__); \
  } while (0)

int launch_prepare(const Plan& p_in, const tgp_model& md, const FlowProg& fp, double* ws, int32_t* status,
                   hipStream_t st) {
  Plan p = p_in;
  p.zs_lds = 1;
  if (prep_a_lds_bytes(p) > 160 * 1024 - 1024) p.zs_lds = 0;
  const size_t lds = prep_a_lds_bytes(p);
  static size_t lds_cur = 48 * 1024;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(k_prep_a), lds, &lds_cur)) return rc;
  hipLaunchKernelGGL(k_prep_a, dim3(3 + p.MT * p.MT), dim3(PREP_THREADS), lds, st, p, md, fp, ws, status);
  LAUNCH_CHECK();
  return 0;
}

int launch_backward_mm(const Plan& p, const tgp_model& md, const tgp_grads& g, double* out, double* ws, int32_t* status,
                       hipStream_t st, const AdamDev* adam) {
  const AdamDev ad = adam != nullptr ? *adam : AdamDev();
  hipLaunchKernelGGL(k_reduce, dim3((unsigned)((p.slab_len + RED_ELEMS - 1) / RED_ELEMS)), dim3(256), 0, st, p, ws);
  LAUNCH_CHECK();
  const size_t lds_col = (size_t)(2 * p.MP * 16 + 16), lds_row = (size_t)p.MT * 256 + 16 * p.DP + (size_t)p.MP * 16,
               lds_fin = (size_t)p.M * (p.D + 1) + (adam != nullptr && ad.n - ad.lam_n <= BWD_MIRROR_MAX ? (size_t)(ad.n - ad.lam_n) : 0) + p.M;   // + the LDS mirror of the gradients + cs
  const size_t lds = sizeof(double) * (lds_col > lds_row ? (lds_col > lds_fin ? lds_col : lds_fin) : (lds_row > lds_fin ? lds_row : lds_fin));
  hipLaunchKernelGGL(k_bwd, dim3(4 * p.MT + 1), dim3(BWD_THREADS), lds, st, p, md, g, out, ws, ad, status);
  LAUNCH_CHECK();
  return 0;
}

static int launch_cov_tile(int kernel, const double* X1, int N1, const double* X2, int N2, int D, const double* raw_ls,
                           const double* raw_os, double jitter, int self, double* K, hipStream_t st) {
  if (D < 1 || D > 16) return TGP_E_UNSUPPORTED;
  if (N1 < 1 || N2 < 1) return 0;
  if ((N2 & 1) == 0 && N2 >= 2 && (long)N2 * D <= 4096 && N2 <= 512 && (reinterpret_cast<uintptr_t>(K) & 15) == 0) {
    const unsigned gx = (unsigned)((N1 + COVF_ROWS - 1) / COVF_ROWS);
    const size_t lds = ((size_t)D * N2 + (size_t)D * COVF_ROWS) * sizeof(double);
    static size_t lds_cur = 48 * 1024;
    if (int rc = ensure_lds(reinterpret_cast<const void*>(k_cov_flat), lds, &lds_cur)) return rc;
    hipLaunchKernelGGL(k_cov_flat, dim3(gx), dim3(256), lds, st, kernel, X1, N1, X2, N2, D, raw_ls, raw_os, jitter, self, K);
    LAUNCH_CHECK();
    return 0;
  }
  const unsigned gx = (unsigned)((N1 + COV_ROWS - 1) / COV_ROWS), gy = (unsigned)((N2 + COV_COLS - 1) / COV_COLS);
  if (gy > 65535u) return TGP_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_cov_tile, dim3(gx, gy), dim3(256), 0, st, kernel, X1, N1, X2, N2, D, raw_ls, raw_os, jitter, self, K);
  LAUNCH_CHECK();
  return 0;
}

int launch_kmm(const double* Z, const double* raw_ls, const double* raw_os, int M, int D, double jitter, double* K,
               hipStream_t st) {
  return launch_cov_tile(TGP_KERNEL_SCALE_RBF, Z, M, Z, M, D, raw_ls, raw_os, jitter, 1, K, st);
}

int launch_knm(const double* X, const double* Z, const double* raw_ls, const double* raw_os, int N, int M, int D,
               double* K, hipStream_t st) {
  return launch_cov_tile(TGP_KERNEL_SCALE_RBF, X, N, Z, M, D, raw_ls, raw_os, 0.0, 0, K, st);
}

int launch_kernel_matrix(int kernel, const double* X1, int N1, const double* X2, int N2, int D, const double* raw_ls,
                         const double* raw_os, double jitter, double* K, hipStream_t st) {
  if (X2 == nullptr) return launch_cov_tile(kernel, X1, N1, X1, N1, D, raw_ls, raw_os, jitter, 1, K, st);
  return launch_cov_tile(kernel, X1, N1, X2, N2, D, raw_ls, raw_os, 0.0, 0, K, st);
}

int launch_kl(const double* m, const double* Lam, int M, double* out, double* g_m, double* g_Lam, hipStream_t st) {
  hipLaunchKernelGGL(k_kl, dim3(1), dim3(256), 0, st, m, Lam, M, out, g_m, g_Lam);
  LAUNCH_CHECK();
  return 0;
}

int launch_cholesky(const double* A, int M, double* L, double* Linv, int32_t* status, hipStream_t st) {
  const int MP = (M + 15) / 16 * 16;
  const size_t lds = ((size_t)MP * (MP + 1) + MP) * sizeof(double);
  static size_t lds_cur = 48 * 1024;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(k_chol_only), lds, &lds_cur)) return rc;
  hipLaunchKernelGGL(k_chol_only, dim3(1), dim3(PREP_THREADS), lds, st, A, M, MP, L, Linv, status);
  LAUNCH_CHECK();
  return 0;
}

}  // namespace tgp
