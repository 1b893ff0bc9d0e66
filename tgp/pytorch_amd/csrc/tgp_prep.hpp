// tgp_prep.hpp -- the tile and transform roles of the prepare launch (k_prep_a, tgp_mm.hip) as device functions, and the
// cross-workgroup hand-off primitives that launch and the M x M backward launch (k_bwd, tgp_mm.hip) use.
//
// Reference: models/sparse_MF_SP.py:316 (K_MM), :344-346 (masked L_q, S = L_q L_q^T), :406-431 (whitened KL); flow
// parameter restrictions models/flow.py:1075.
//
// Cross-workgroup hand-off inside ONE launch.  Two int32 words behind the caller's status words (status[4], status[5]
// of the int32[8] the ABI asks for, zero before the first call -- per caller, so that two engines on one device never
// share them; every launch leaves them zero):
//   SY_TILES number of tile blocks that have written their tile of K_MM (and of L_q, L_q^T)
//   SY_DONE  number of blocks of the launch that have finished; the last one zeroes both words
// No fences: an agent-scope release / acquire costs a write-back / invalidate of the whole L2 of the XCD on gfx950.
// Instead every datum that crosses workgroups inside the launch is stored and loaded as a RELAXED AGENT-SCOPE ATOMIC
// (st_agent / ld_agent: plain global_store / global_load with the sc1 bit, i.e. written through to / fetched from the
// level that is coherent for the whole device); a producer waits for its stores (s_waitcnt vmcnt(0) in EVERY storing
// wave -- handoff_barrier() below: hipcc's __syncthreads() does not emit that wait on gfx950), passes the workgroup
// barrier, and ONE thread sets the word; a consumer polls the word (bounded: a waiter that runs out of patience records
// TGP_STATUS_SYNC_TIMEOUT in status[0] and leaves, so a protocol error cannot hang the GPU) and only then issues its loads.
// Deadlock freedom: workgroups are dispatched in index order and the producers (the tile blocks) carry the LOWEST
// indices of k_prep_a's grid, so every producer is resident before a consumer that waits for it can occupy a CU.
#pragma once
#include "tgp_dev.hpp"

namespace tgp {

enum { SY_TILES = 0, SY_DONE = 1, SY_NWORDS = 2 };
// (TGP_STATUS_SYNC_TIMEOUT, the value a waiter that gives up leaves in status[0], is part of the ABI: include/tgp_hip.h)
// Producer side of a hand-off: EVERY wave drains its global stores (and its LDS traffic), then the workgroup barrier.
// The thread that sets the hand-off word does so behind this -- never behind a bare __syncthreads(), which on gfx950
// compiles to s_barrier with no s_waitcnt vmcnt(0): the word could overtake another wave's sc1 stores.
__device__ __forceinline__ void handoff_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
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
// Leaving the launch: count this block; the last one resets the hand-off words for the next launch.
__device__ __forceinline__ void sync_leave(int32_t* sy, int nblocks_total) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(sy + SY_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nblocks_total - 1) {   // every other block has left: nobody reads the words any more
      sync_st(sy + SY_TILES, 0); sync_st(sy + SY_DONE, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// tile role: 16x16 tile t of {Lq, Lq^T, K_MM -> HBM} and the S = Lq Lq^T tile (one MFMA chain)
// (sparse_MF_SP.py:316,344-346).  NT threads (256 or 512); the first 256 copy, one wave forms S.
// ---------------------------------------------------------------------------------------------------
// `sy`: SY_TILES += 1 once K_MM / L_q / L_q^T are out -- k_prep_a's chain blocks wait for exactly that.
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
    handoff_barrier();   // every wave's sc1 stores have landed before the word moves
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
}

// ---------------------------------------------------------------------------------------------------
// transform role: lengthscale / outputscale / noise transforms, Zs = Z / l, padded m, flow parameter transforms,
// whitened KL, header scalars
// ---------------------------------------------------------------------------------------------------
template <int NT, bool SHARED>
__device__ __forceinline__ void prep_xform_role(const Plan& p, const tgp_model& md, const FlowProg& fp, double* __restrict__ ws) {
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

}  // namespace tgp
