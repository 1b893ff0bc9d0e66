// tgp_prep.hpp -- the prepare roles of the ELBO step as device functions, so that they can run either as the blocks of a
// launch of their own (k_prep_a, tgp_mm.hip: 512-thread blocks, two chain blocks) or as the leading blocks of the FUSED
// step launch (k_rows<..., FUSED = true>, tgp_rows.hpp: 256-thread blocks, one chain block), where the row blocks start
// beside the factorisation and consume L panel by panel as it is published.
//
// Reference: models/sparse_MF_SP.py:316 (K_MM), :330 + dsp/utils.py:222-270 (psd_safe_cholesky and its jitter ladder),
// :344-346 (masked L_q, S = L_q L_q^T), :406-431 (whitened KL); flow parameter restrictions models/flow.py:1075.
//
// Cross-workgroup hand-off inside ONE launch (fused path only).  Four int32 words behind the caller's status words
// (status[4..7]; the ABI asks for them to be zero before the first call, every launch leaves them zero):
//   SY_TILES bits 0-15: number of tile blocks that have written their tiles of L_q, L_q^T, K_MM; bits 16-23: ... of S;
//            bit 24: the transform block has written Zs, 1/l, the header scalars, the flow parameter transforms, padded m
//   SY_COLS0 / SY_COLS1: a 16-bit field per chain block b (b = 0, 1 in SY_COLS0, b = 2, 3 in SY_COLS1): 16 * attempt + n,
//            every operand panel c < n with c mod TGP_CHAIN_BLOCKS = b is complete in global memory (panel c = tiles
//            (c, k < c) of L and L^T and -Dinv_c); `attempt` = level of the on-device jitter ladder the factorisation is at
//   SY_DONE  number of blocks of the launch that have finished; the last one zeroes the four words
// No fences: an agent-scope release / acquire costs a write-back / invalidate of the whole L2 of the XCD on gfx950 (135
// polling row blocks kept every L2 of the chip empty: the first fused build took 130 us).  Instead every datum that
// crosses workgroups inside the launch is stored and loaded as a RELAXED AGENT-SCOPE ATOMIC (st_agent / ld_agent: plain
// global_store / global_load with the sc1 bit, i.e. written through to / fetched from the level that is coherent for the
// whole device); a producer waits for its stores (s_waitcnt vmcnt(0)), passes a workgroup barrier, and ONE thread sets
// the word; a consumer polls the word (bounded: a waiter that runs out of patience records TGP_STATUS_SYNC_TIMEOUT in
// status[0] and leaves, so a protocol error cannot hang the GPU) and only then issues its loads.
// Deadlock freedom: workgroups are dispatched in index order and the producers carry the lowest indices, so a producer is
// resident before any consumer that waits for it can occupy a CU.
#pragma once
#include "tgp_dev.hpp"

namespace tgp {

enum { SY_COLS0 = 0, SY_TILES = 1, SY_COLS1 = 2, SY_DONE = 3 };
#define TGP_SY_XF_BIT (1 << 24)
#ifndef TGP_CHAIN_BLOCKS
#define TGP_CHAIN_BLOCKS 4 /* redundant factorisation workgroups of the fused launch (1, 2 or 4): they share what leaves the workgroup */
#endif
// chain block b's progress field: 16 bits (16 * ladder attempt + panels complete) in SY_COLS0 (b = 0, 1) / SY_COLS1 (b = 2, 3)
__device__ __forceinline__ int cols_word(int b) { return (b >> 1) ? SY_COLS1 : SY_COLS0; }
__device__ __forceinline__ int cols_field(int x, int b) { return (x >> (16 * (b & 1))) & 0xffff; }
#define TGP_STATUS_SYNC_TIMEOUT (-77)
#define TGP_SYNC_MAX_POLLS (1 << 22)

__device__ __forceinline__ int sync_ld(const int32_t* p) {
  const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");   // (compiler only) nothing that follows is moved above the poll
  return v;
}
// the caller has waited for its data stores (vmcnt(0)) and passed the workgroup's barrier
__device__ __forceinline__ void sync_st(int32_t* p, int v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sync_add(int32_t* p, int v) {
  asm volatile("" ::: "memory");
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// data that crosses workgroups inside one launch (see the header): coherent for the whole device, never through the
// scalar cache
__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// two consecutive doubles (16-byte aligned) in ONE store instruction: a wave's global stores cost ~32 ns each whatever
// their width (tools/probes/sc1_rate.hip), so the write-out of the factor is counted in instructions.  There is no
// 128-bit atomic, so the coherent 16-byte store is a raw buffer store with the sc1 cache-policy bit (aux = 16 on
// gfx940+), `rs` a descriptor over the whole workspace and `off` the element's index in it.
typedef __amdgpu_buffer_rsrc_t ws_rsrc_t;
__device__ __forceinline__ ws_rsrc_t ws_rsrc(double* ws) {
  return __builtin_amdgcn_make_buffer_rsrc(ws, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void st_agent2(ws_rsrc_t rs, size_t off, double a, double b) {
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  const u4v v = {(unsigned)__double2loint(a), (unsigned)__double2hiint(a), (unsigned)__double2loint(b), (unsigned)__double2hiint(b)};
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(off * 8), 0, 16);
}
template <bool SHARED>
__device__ __forceinline__ void st_maybe(double* p, double v) {
  if constexpr (SHARED) st_agent(p, v); else *p = v;
}
template <bool SHARED>
__device__ __forceinline__ double ld_maybe(const double* p) {
  if constexpr (SHARED) return ld_agent(p); else return *p;
}
// One wave-uniform poll loop (every lane loads the same word: one request per wave).  pred(v) -> bool.  Returns the
// value that satisfied pred, or INT_MIN after TGP_SYNC_MAX_POLLS tries.
template <class P>
__device__ __forceinline__ int sync_wait(const int32_t* p, P pred) {
  for (int it = 0; it < TGP_SYNC_MAX_POLLS; ++it) {
    const int v = sync_ld(p);
    if (pred(v)) return v;
    __builtin_amdgcn_s_sleep(4);
  }
  return (int)0x80000000;
}
// Leaving a fused launch: count this block; the last one resets the hand-off words for the next launch.
__device__ __forceinline__ void sync_leave(int32_t* sy, int nblocks_total) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(sy + SY_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nblocks_total - 1) {   // every other block has left: nobody reads the words any more
      sync_st(sy + SY_COLS0, 0); sync_st(sy + SY_TILES, 0); sync_st(sy + SY_COLS1, 0); sync_st(sy + SY_DONE, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// tile role: 16x16 tile t of {Lq, Lq^T, K_MM -> HBM} and the S = Lq Lq^T tile (one MFMA chain)
// (sparse_MF_SP.py:316,344-346).  NT threads (256 or 512); the first 256 copy, one wave forms S.
// ---------------------------------------------------------------------------------------------------
// `sy` (fused launch): SY_TILES += 1 once K_MM / L_q / L_q^T are out -- the chain blocks wait for exactly that -- and
// += 1 << 16 once the S tile is (only the passenger blocks read S, tens of microseconds later).
template <int NT, bool SHARED>
__device__ __forceinline__ void prep_tile_role(const Plan& p, const tgp_model& md, double* __restrict__ ws, int t,
                                               int32_t* sy = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int M = p.M, D = p.D, MP = p.MP, MT = p.MT;
  const int ti = t / MT, tj = t % MT;
  // 1/l_d and sigma^2 once per block (every thread evaluating D + 1 softplus chains of its own took 4 us: nobody waited
  // for the tile blocks while they rode beside a 26 us factorisation, the fused launch's chain block does)
  __shared__ double t_ils[17];
  // (this thread's own operands are requested before the transforms and their barrier: one memory round trip, not two)
  double zr[16], zc[16], lq = 0.0;
  const int rr0 = tid >> 4, cc0 = tid & 15, row0 = ti * 16 + rr0, col0 = tj * 16 + cc0;
  const bool inside = tid < 256 && row0 < M && col0 < M;
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    zr[d] = (inside && d < D) ? md.Z[(size_t)row0 * D + d] : 0.0;
    zc[d] = (inside && d < D) ? md.Z[(size_t)col0 * D + d] : 0.0;
  }
  if (inside && col0 <= row0) lq = md.Lam[(size_t)row0 * M + col0];
  if (tid < 16) t_ils[tid] = tid < D ? 1.0 / softplus_d(md.raw_ls[tid]) : 0.0;
  if (tid == 64) t_ils[16] = softplus_d(md.raw_os[0]);
  __syncthreads();
  if (tid < 256) {
    const int row = row0, col = col0;
    double k = 0.0;
    if (inside) {
      double d2 = 0.0;
#pragma unroll
      for (int d = 0; d < 16; ++d) {
        const double il = t_ils[d];   // 0 beyond D
        const double tt = zr[d] * il - zc[d] * il;
        d2 += tt * tt;
      }
      k = t_ils[16] * exp_fast(-0.5 * d2);
    }
    st_maybe<SHARED>(ws + p.Lq + (size_t)row * MP + col, lq);
    st_maybe<SHARED>(ws + p.LqT + (size_t)col * MP + row, lq);
    st_maybe<SHARED>(ws + p.Kmm + (size_t)row * MP + col, k);
    if (tj > ti) {  // strictly-upper tile: zero in L and in J = L^-1 (formed by the row kernel's passenger blocks)
      ws[p.L + (size_t)row * MP + col] = 0.0;
      ws[p.J + (size_t)row * MP + col] = 0.0;
    }
  }
  if (sy != nullptr) {
    __syncthreads();   // (drains every wave's stores)
    if (tid == 0) sync_add(sy + SY_TILES, 1);
  }
  if (wave == (NT > 256 ? 4 : 3)) {  // S tile (NT = 512: on a wave that did no copy work)
    const int i = ti * 16 + r, j = tj * 16 + r;
    d4 acc = {0, 0, 0, 0};
    const int kend = (ti < tj ? ti : tj) * 16 + 16;  // Lq[i,k] = 0 for k > i
    // (every operand fragment of the tile requested before the first MFMA: the rolled loop paid one L2 round trip per k-step)
    acc = tile_mm_f<TGP_GBATCH>([&](int k) { return (i < M && k + q <= i) ? md.Lam[(size_t)i * M + k + q] : 0.0; },
                                [&](int k) { return (j < M && k + q <= j) ? md.Lam[(size_t)j * M + k + q] : 0.0; }, 0, kend, acc);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) st_maybe<SHARED>(ws + p.S_ + (size_t)(ti * 16 + q + 4 * rr) * MP + tj * 16 + r, acc[rr]);
  }
  if (sy != nullptr) {
    __syncthreads();
    if (tid == 0) sync_add(sy + SY_TILES, 1 << 16);
  }
}

// ---------------------------------------------------------------------------------------------------
// transform role: lengthscale / outputscale / noise transforms, Zs = Z / l, padded m, flow parameter transforms,
// whitened KL, header scalars
// ---------------------------------------------------------------------------------------------------
template <int NT, bool SHARED>
__device__ __forceinline__ void prep_xform_role(const Plan& p, const tgp_model& md, const FlowProg& fp, double* __restrict__ ws,
                                                int32_t* sy = nullptr) {
  __shared__ double red1[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = p.M, D = p.D, MP = p.MP, DP = p.DP;
  double* hdr = ws + p.hdr;
  if (tid < 16) {
    const double l = tid < D ? softplus_d(md.raw_ls[tid]) : 1.0;  // gpytorch Positive constraint = softplus
    st_maybe<SHARED>(ws + p.ls + tid, l);
    st_maybe<SHARED>(ws + p.ils + tid, tid < D ? 1.0 / l : 0.0);
  }
  for (int i = tid; i < MP * DP; i += NT) {
    const int mrow = i / DP, d = i % DP;
    st_maybe<SHARED>(ws + p.Zs + i, (mrow < M && d < D) ? md.Z[(size_t)mrow * D + d] * (1.0 / softplus_d(md.raw_ls[d])) : 0.0);
  }
  for (int i = tid; i < MP; i += NT) st_maybe<SHARED>(ws + p.mpad + i, i < M ? md.m[i] : 0.0);
  // one thread per shared flow parameter (it finds its block; a thread per BLOCK walked up to 4 K softplus / sigmoid
  // chains one after the other -- 10 us of a block the row blocks of the fused launch wait for)
  for (int t = tid; t < p.P; t += NT) {
    for (int b = 0; b < fp.nblk; ++b) {
      const int kind = fp.blk[4 * b], K = fp.blk[4 * b + 1], poff = fp.blk[4 * b + 2], flags = fp.blk[4 * b + 3];
      if (flags & TGP_FLAG_PER_ROW) continue;
      const int np = kind == TGP_FLOW_STEPTANH ? 4 * K : 2;
      if (t < poff || t >= poff + np) continue;
      const int j = t - poff;
      const double x = md.theta[t];
      bool res;
      if (kind == TGP_FLOW_STEPTANH) res = (j & 1);  // b_k, d_k: TanhFlow set_restrictions=True (flow.py:1075)
      else res = (flags & TGP_FLAG_RESTRICT) && j == (kind == TGP_FLOW_AFFINE ? 0 : 1);
      st_maybe<SHARED>(ws + p.tp + t, res ? softplus_d(x) : x);
      st_maybe<SHARED>(ws + p.tg + t, res ? sigmoid_d(x) : 1.0);
      break;
    }
  }
  if (tid == NT - 1) {   // the header scalars the row blocks read
    st_maybe<SHARED>(hdr + H_S2, softplus_d(md.raw_os[0]));
    st_maybe<SHARED>(hdr + H_ETA, md.log_var_noise[0]);
    st_maybe<SHARED>(hdr + H_EINV, exp(-md.log_var_noise[0]));  // 1/positive_transform (dsp/utils.py:39-41, 'exp')
    st_maybe<SHARED>(hdr + H_SIG_OS, sigmoid_d(md.raw_os[0]));
  }
  if (sy != nullptr) {   // fused launch: everything a row block stages is out; the KL below is only read by k_bwd5
    __syncthreads();     // (drains every wave's stores)
    if (tid == 0) sync_add(sy + SY_TILES, TGP_SY_XF_BIT);
  }
  // whitened KL (models/sparse_MF_SP.py:406-431)
  double kl_part = 0.0;
  for (int i = tid; i < M * M; i += NT) {
    const int rr = i / M, cc = i % M;
    if (cc <= rr) {
      const double x = md.Lam[i];
      kl_part += x * x;
      if (cc == rr) kl_part -= log(x * x);
    }
  }
  for (int i = tid; i < M; i += NT) kl_part += md.m[i] * md.m[i];
  kl_part = wave_sum(kl_part);
  if (lane == 0) red1[wave] = kl_part;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < NT / 64; ++i) s += red1[i];
    hdr[H_KL] = 0.5 * (s - (double)M);
  }
}

// ---------------------------------------------------------------------------------------------------
// chain role of the FUSED launch: the blocked right-looking Cholesky of K_MM by a 256-thread workgroup; TWO of them (cb =
// 0, 1) run it redundantly -- same arithmetic, bit for bit, nothing exchanged -- and share what leaves the workgroup
// (write-out of L / L^T, the inverses of the diagonal tiles) by the parity of the panel: four waves alone spent 4-6 us per
// block column on that against the 2.2 us of the chain's register pass.  (one wave per
// SIMD: wave 0 -- for the rows beyond 64 below the diagonal also wave 1 -- runs the register pass of the block column,
// potrf_panel16; the other waves take the window's tasks from a counter in LDS), publishing the operand panels of the
// row blocks as they complete.  Differences from k_prep_a's chain blocks (tgp_mm.hip), all following from the fused
// launch:  * K_MM's columns >= 1 are not generated here: the tile blocks of the same launch have them in global memory
//            long before their column is due (SY_TILES); a task wave fetches column j+2 in window j (columns 1, 2 in
//            window 0), adds the jitter and the identity on the padding;
//          * tile row c of L / L^T (the tiles left of the diagonal: final since pass c-1) is written out in window c,
//            -Dinv_c and the diagonal tile in window c+1, and the panel count goes to SY_COLS right behind that
//            window's barrier; the last diagonal tile carries its inverse in its own pass (no panel rows below it, so
//            the three-array form does not spill) -- the row blocks are waiting for exactly that tile;
//          * a failed pivot ends the attempt at once (the ladder restarts with more jitter; the attempt number is part
//            of SY_COLS, and a row block that has consumed panels of a dead attempt starts its substitution again).
// LDS: MP x (MP+1) doubles for the matrix + MP x DP for the scaled inducing points (when p.zs_lds) at the start of `sm`.
// ---------------------------------------------------------------------------------------------------
template <int MT>
__device__ __forceinline__ void fused_chain_role(const Plan& p, const tgp_model& md, double* __restrict__ ws, int32_t* status,
                                                 double* sm, int cb) {
  constexpr int MP = MT * 16, LD = MP + 1, NW = 4, NT = 256;
  int32_t* sy = status + 4;
  constexpr int R = TGP_CHAIN_BLOCKS;
  int32_t* my_cols = sy + cols_word(cb);
  __shared__ int s_pub, s_drain;   // this block's published field; task waves that have drained their earlier stores (cumulative)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int M = p.M, D = p.D, DP = p.DP;
  double* A = sm;
  double* zs = sm + (size_t)MP * LD;
  __shared__ int s_info, s_nan, s_next, s_sync;
  __shared__ double s_ils[16];
  if (tid == 0) { s_info = 0; s_nan = 0; s_next = 0; s_sync = 0; s_pub = 0; s_drain = 0; }
  int tbase = 0, dbase = 0;
  if (tid < 16) s_ils[tid] = tid < D ? 1.0 / softplus_d(md.raw_ls[tid]) : 0.0;
  const double s2 = softplus_d(md.raw_os[0]);
  const bool zl = p.zs_lds != 0;
  if (zl) {
    for (int i = tid; i < MP * DP; i += NT) {
      const int mrow = i / DP, d = i % DP;
      zs[i] = (mrow < M && d < D) ? md.Z[(size_t)mrow * D + d] * (1.0 / softplus_d(md.raw_ls[d])) : 0.0;
    }
  }
  __syncthreads();
  bool has_nan = false;
  double jit = md.jitter;
  // K_MM tile (i, 0): column 0 heads the chain and is generated here (four staged exponentials per lane)
  auto fill_tile0 = [&](int i) {
    double kv[4];
    const int cc = r;
    double e[4];
    TGP_EACH(u, 4) {
      const int rr = 16 * i + q + 4 * u;
      double d2 = 0.0;
      if (zl) {
        for (int d = 0; d < DP; d += 4) {
          const double2 a0 = *reinterpret_cast<const double2*>(zs + rr * DP + d), a1 = *reinterpret_cast<const double2*>(zs + rr * DP + d + 2);
          const double2 b0 = *reinterpret_cast<const double2*>(zs + cc * DP + d), b1 = *reinterpret_cast<const double2*>(zs + cc * DP + d + 2);
          const double t0 = a0.x - b0.x, t1 = a0.y - b0.y, t2 = a1.x - b1.x, t3 = a1.y - b1.y;
          d2 += t0 * t0; d2 += t1 * t1; d2 += t2 * t2; d2 += t3 * t3;
        }
      } else if (rr < M) {  // M = 128 with D > 8: no LDS left next to the matrix; read Z through L1/L2
        for (int d = 0; d < D; ++d) {
          const double tt = (md.Z[(size_t)rr * D + d] - md.Z[(size_t)cc * D + d]) * s_ils[d];
          d2 += tt * tt;
        }
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
#pragma unroll
    for (int u = 0; u < 4; ++u) A[(16 * i + q + 4 * u) * LD + r] = kv[u];
  };
  // column c of K_MM (tiles (i, c), i >= c) from the tile blocks' copy in global memory -> LDS, with the jitter on the
  // diagonal and the identity on the padding; every load of the column requested before the first use
  const double* __restrict__ Kg = ws + p.Kmm;
  auto fetch_col = [&](int c) {
    double kv[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i < c) continue;
#pragma unroll
      for (int u = 0; u < 4; ++u) kv[i][u] = ld_agent(Kg + (size_t)(16 * i + q + 4 * u) * MP + 16 * c + r);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i < c) continue;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = 16 * i + q + 4 * u, cc = 16 * c + r;
        double k = kv[i][u];
        has_nan |= (k != k);
        if (rr == cc) k = rr < M ? k + jit : 1.0;
        A[rr * LD + cc] = k;
      }
    }
  };
  // block column c, tiles (i, c), i = c .. MT-1, -= L[i rows, 0 .. 16 NC) L[c rows, 0 .. 16 NC)^T (all final): ONE task.
  // NC is a compile-time constant (dispatch below): straight-line code, the B operand (the rows of tile row c) read
  // from LDS once and kept in registers, and two tiles per trip with all their LDS reads ahead of the first MFMA -- as
  // a tile-by-tile loop with a run-time column count (a branch per k-step, read -> wait -> MFMA) a tile with 12 MFMAs
  // took a microsecond.
  auto catch_up_nc = [&](int c, auto ncc) {
    constexpr int NC = decltype(ncc)::value;
    double bf[4 * NC];
#pragma unroll
    for (int s4 = 0; s4 < 4 * NC; ++s4) bf[s4] = A[(16 * c + r) * LD + 4 * s4 + q];
    for (int i = c; i < MT; i += 2) {
      const bool two = i + 1 < MT;
      const int i1 = two ? i + 1 : i;
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
  auto catch_up_col = [&](int c, int ncol) {
    static_for<(MT > 2 ? MT - 2 : 1)>([&](auto k) {
      if (ncol == decltype(k)::value + 1) catch_up_nc(c, std::integral_constant<int, decltype(k)::value + 1>{});
    });
  };
  // lower tile (ti, tj) of L and tile (tj, ti) of L^T -> global memory: four 16-byte stores per lane (lane = row 8 h +
  // (lane >> 3), column pair 2 (lane & 7)), every LDS read of the tile ahead of the first store
  const ws_rsrc_t rs = ws_rsrc(ws);
  auto write_L = [&](int ti, int tj) {
    const int rw = lane >> 3, cp = 2 * (lane & 7);
    double a[2][2], b[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * h + rw;
      a[h][0] = A[(16 * ti + row) * LD + 16 * tj + cp];       // L[16 ti + row][16 tj + cp], [.. + 1]
      a[h][1] = A[(16 * ti + row) * LD + 16 * tj + cp + 1];
      b[h][0] = A[(16 * ti + cp) * LD + 16 * tj + row];       // L^T[16 tj + row][16 ti + cp] = L[16 ti + cp][16 tj + row]
      b[h][1] = A[(16 * ti + cp + 1) * LD + 16 * tj + row];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * h + rw;
      if (ti == tj) {   // diagonal tile: zero above the diagonal (L), below it (L^T)
        if (cp > row) a[h][0] = 0.0;
        if (cp + 1 > row) a[h][1] = 0.0;
        if (row > cp) b[h][0] = 0.0;
        if (row > cp + 1) b[h][1] = 0.0;
      }
      st_agent2(rs, p.L + (size_t)(16 * ti + row) * MP + 16 * tj + cp, a[h][0], a[h][1]);
      st_agent2(rs, p.LT + (size_t)(16 * tj + row) * MP + 16 * ti + cp, b[h][0], b[h][1]);
    }
  };
  auto store_nD = [&](int jt, const double (&xv)[16]) {  // xv[c] = Dinv[c][lane & 15], replicated in the four 16-lane rows
    const int li = lane & 15;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double xv4 = q == 0 ? xv[4 * u] : (q == 1 ? xv[4 * u + 1] : (q == 2 ? xv[4 * u + 2] : xv[4 * u + 3]));
      st_agent(ws + p.nD + jt * 256 + (4 * u + q) * 16 + li, -xv4);
    }
  };
  auto inv_diag = [&](int jt) {
    const int li = lane & 15;
    double dgv[16], xv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) dgv[c] = A[(16 * jt + li) * LD + 16 * jt + c];
    trtri16<false>(dgv, xv, li);
    store_nD(jt, xv);
  };
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
  // LDS traffic AND this wave's global stores have landed, then the workgroup barrier: a window's write-out is complete
  // in global memory when the publishing thread passes it
  auto window_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto lds_only_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // this block's field <- v (monotone; one publisher at a time: thread 0 at a window's barrier, or the inverse's wave
  // inside the window, which joins that barrier afterwards)
  auto publish = [&](int v) {
    const int old = s_pub;
    if (v > old) {
      s_pub = v;
      sync_add(my_cols, (v - old) << (16 * (cb & 1)));
    }
  };

#ifdef TGP_STAMPS
#define CHAIN_STAMP(i) do { if (tid == 0 && cb == 0) ws[p.hdr + H_PSTAMP + (i)] = (double)__builtin_amdgcn_s_memrealtime(); } while (0)
#define WAVE_STAMP(j, k) do { if (lane == 0 && cb == 0) ws[p.dbg + ((j) * 4 + wave) * 4 + (k)] = (double)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CHAIN_STAMP(i) do { } while (0)
#define WAVE_STAMP(j, k) do { } while (0)
#endif
  CHAIN_STAMP(0);
#ifdef TGP_STAMPS
  if (tid == 0 && cb == 0) ws[p.hdr + 61] = (double)p.dbg;
#endif
  const bool ladder = md.jitter_ladder > 0.0;
  int attempt = 0;
  bool tiles_seen = false;
  for (;; ++attempt) {
    for (int i = wave; i < MT; i += NW) fill_tile0(i);
    lds_only_barrier();
    CHAIN_STAMP(1);
    bool failed = false;
    for (int j = 0; j < MT; ++j) {
      const int j0 = 16 * j;
      const int npan = (MT - 1 - j) * 16;
      const int npw = npan > 64 ? 2 : 1;   // panel waves: wave 0 (diagonal tile + rows 0..63 below it), wave 1 (the rest)
      double ltile[4] = {0.0, 0.0, 0.0, 0.0};
      bool did_diag = false;
      WAVE_STAMP(j, 0);
      if (wave < npw) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the tile inverse this wave stored a window ago)
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
        // (the inverse of the LAST diagonal tile rides in its pass -- nothing below the tile, every row block waiting for
        //  it; the others are tasks of the next window: beside a panel the three-array pass took 3.6 us against 2.0)
        int bad = 0;
        if (npan > 0) {
          bad = potrf_panel16<true>(dg, a);
          if (has) {
#pragma unroll
            for (int c = 0; c < 16; ++c) A[prow * LD + j0 + c] = a[c];
          }
        } else if ((j % R) == cb) {
          double xv[16];
          bad = potrf_panel16<false, true>(dg, a, xv, li);
          store_nD(j, xv);
        } else {
          bad = potrf_panel16<false>(dg, a);
        }
        if (wave == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            ltile[u] = q == 0 ? dg[4 * u] : (q == 1 ? dg[4 * u + 1] : (q == 2 ? dg[4 * u + 2] : dg[4 * u + 3]));
          did_diag = true;
          if (lane == 0 && bad != 0 && s_info == 0) s_info = j0 + bad;
        }
      } else {
        // window j, heaviest first:  the inverse of diagonal tile j-1;  column j+2 of K_MM from global memory (columns
        // 1 and 2 in window 0);  block column j+1 -= block columns 0 .. j-1 (one task);  tile row j of L / L^T left of the
        // diagonal (final since pass j-1) and the diagonal tile j-1 -> global memory.
        // The stores are not waited for here: a wave drains its earlier windows' stores when it ENTERS a window (they
        // were issued a window ago: nothing to wait for), so what is known complete at window j's barrier is everything
        // issued up to window j-1, i.e. the operand panels 0 .. j-2 (panel c = tile row c, window c, and -Dinv_c,
        // window c+1) -- published one window late, which costs nothing: only the last panel is ever waited for.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(&s_drain, 1);
        WAVE_STAMP(j, 1);
        // (what leaves the workgroup is shared with the other chain block by the parity of the panel it belongs to:
        //  panel c = tile row c and -Dinv_c, plus the diagonal tile c, is block c & 1's)
        const int nt = (j >= 1 && ((j - 1) % R) == cb) ? 1 : 0;     // -Dinv_{j-1}, and the diagonal tile (j-1, j-1)
        const int nfe = j == 0 ? ((MT > 1 ? 1 : 0) + (MT > 2 ? 1 : 0)) : (j + 2 < MT ? 1 : 0);
        const int ncu = (j >= 1 && j + 1 < MT) ? 1 : 0;
        const int nwo = (j % R) == cb ? j : 0;                       // tiles (j, 0 .. j-1)
        const int ntask = nt + nfe + ncu + nwo + nt;
#ifdef TGP_STAMPS
        int lg = 0;
#endif
        for (;;) {
          int t = 0;
          if (lane == 0) t = atomicAdd(&s_next, 1);
          t = __builtin_amdgcn_readfirstlane(t) - tbase;
#ifdef TGP_STAMPS
          if (j == 3 && cb == 0 && lane == 0 && lg < 10) { ws[p.dbg + 120 + wave * 24 + 2 * lg] = (double)t; ws[p.dbg + 120 + wave * 24 + 2 * lg + 1] = (double)__builtin_amdgcn_s_memrealtime(); }
          ++lg;
#endif
          if (t >= ntask) break;
          if (t < nt) {
            inv_diag(j - 1);
            // panel j-1 is this block's and complete with these stores (its tile row went out a window ago): publish it
            // now rather than at the next window's barrier -- for the last panels that is what the row blocks wait for.
            // (own stores drained; the other task waves' drains of this window's entry counted in LDS)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
              const int need = dbase + (NW - npw);
              for (int it = 0; it < (1 << 20) && __hip_atomic_load(&s_drain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need; ++it)
                __builtin_amdgcn_s_sleep(1);
              if (s_info == 0 && s_sync == 0) publish(16 * attempt + j);
            }
            continue;
          }
          t -= nt;
          if (t < nfe) {
            if (!tiles_seen) {
              const int v = sync_wait(sy + SY_TILES, [&](int x) { return (x & 0xffff) >= MT * MT; });
              if (v == (int)0x80000000 && lane == 0) s_sync = 1;
              tiles_seen = true;
              WAVE_STAMP(j, 2);
            }
            fetch_col(j == 0 ? 1 + t : j + 2);
            continue;
          }
          t -= nfe;
          if (t < ncu) { catch_up_col(j + 1, j); continue; }
          t -= ncu;
          if (t < nwo) write_L(j, t); else write_L(j - 1, j - 1);
        }
      }
      {
        const int nt = (j >= 1 && ((j - 1) % R) == cb) ? 1 : 0;
        const int nfe = j == 0 ? ((MT > 1 ? 1 : 0) + (MT > 2 ? 1 : 0)) : (j + 2 < MT ? 1 : 0);
        tbase += 2 * nt + nfe + ((j >= 1 && j + 1 < MT) ? 1 : 0) + ((j % R) == cb ? j : 0) + (NW - npw);   // tasks + one over-grab per task wave
        dbase += NW - npw;
      }
      WAVE_STAMP(j, 3);
      lds_only_barrier();
      CHAIN_STAMP(2 + 2 * j);
      failed = s_info != 0 || s_sync != 0;
      if (j >= 2 && !failed && tid == 0) publish(16 * attempt + j - 1);
      if (failed) break;
      if (did_diag) {
#pragma unroll
        for (int u = 0; u < 4; ++u) A[(j0 + (lane & 15)) * LD + j0 + 4 * u + q] = ltile[u];
      }
      if (j + 1 < MT) {
        for (int i = j + 1 + wave; i < MT; i += NW) sub16(i, j + 1, j0);
        lds_only_barrier();
      }
      CHAIN_STAMP(3 + 2 * j);
    }
    if (has_nan) s_nan = 1;
    __syncthreads();
    const bool last = !failed || s_nan != 0 || s_sync != 0 || !ladder || attempt == 3;
    if (last) break;
    __syncthreads();  // everybody has read s_info
    if (tid == 0) s_info = 0;
    jit = md.jitter + md.jitter_ladder * (attempt == 0 ? 1.0 : (attempt == 1 ? 10.0 : 100.0));
    __syncthreads();
  }
  // tail: the last diagonal tile of L (k_bwd12 reads it; its inverse left with the pass), every store of the launch
  // drained, then the terminal count.  A final attempt that failed still publishes -- the row blocks must not wait for
  // ever; status[] tells the host.
  if (wave == 1 && ((MT - 1) % R) == cb) write_L(MT - 1, MT - 1);
  window_barrier();
  if (tid == 0) {
    if (cb == 0) {   // (the other chain block arrives at the same three words)
      status[0] = s_sync != 0 ? TGP_STATUS_SYNC_TIMEOUT : s_info;
      status[1] = s_nan;
      status[2] = (s_info == 0 && s_sync == 0) ? attempt : 0;
    }
    publish(16 * attempt + MT);
  }
  CHAIN_STAMP(2 + 2 * MT);
#undef CHAIN_STAMP
#undef WAVE_STAMP
}

}  // namespace tgp
