// tgp_mm.hip -- the M x M side of the ELBO step (everything that does not scale with the rows).
//
// Forward  (k_prep_a, k_prep_b):  lengthscale/outputscale transforms, Zs = Z/l, K_MM, Cholesky L,
//          J = L^-1, masked L_q, S = L_q L_q^T, H' = J^T (S - I), w = J^T m, KL, flow parameter transforms.
//          Replaces models/sparse_MF_SP.py:316,330,344-346,406-431 and dsp/utils.py:222-270 (the retry
//          ladder itself stays on the host, driven by status[]).
// Backward (k_reduce, k_bwd1..5): slab reduction of the row statistics, then the hand-derived adjoint
//          (restructured -- see DESIGN.md section 3):
//            Lbar   = -tril(w s^T + 2 H' G)            Lambar = 2 tril(G L_q) - kl (L_q - diag(1/Lam_ii))
//            Q      = Phi(L^T Lbar) + Phi(.)^T         Kbar_MM = 1/2 J^T Q J
//          and the ARD-RBF parameter gradients from Kbar_MM and the row statistics T.
// All GEMM-shaped work is 16x16 output tiles on v_mfma_f64_16x16x4_f64, one wave per tile.
#include "tgp_dev.hpp"
#include "tgp_launch.hpp"

namespace tgp {

// ---------------------------------------------------------------------------------------------------
// k_prep_a: block 0 = transforms + K_MM + Cholesky + inverse (+KL, flow params); blocks 1.. = S tiles
// ---------------------------------------------------------------------------------------------------
#define PREP_THREADS 1024

__global__ __launch_bounds__(PREP_THREADS) void k_prep_a(Plan p, tgp_model md, double* __restrict__ ws,
                                                         int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x;
  const int M = p.M, D = p.D, MP = p.MP, DP = p.DP;

  if (blockIdx.x > 0) {
    // ---- S = Lq Lq^T tile, Lq = tril(Lam) masked on load (models/sparse_MF_SP.py:344-346) ----
    if (tid >= 64) return;
    const int t = blockIdx.x - 1, ti = t / p.MT, tj = t % p.MT;
    const int r = tid & 15, q = tid >> 4;
    const int i = ti * 16 + r, j = tj * 16 + r;
    d4 acc = {0, 0, 0, 0};
    const int kend = (ti < tj ? ti : tj) * 16 + 16;  // Lq[i,k] = 0 for k > i
    for (int k = 0; k < kend; k += 4) {
      const int kk = k + q;
      const double a = (i < M && kk <= i) ? md.Lam[(size_t)i * M + kk] : 0.0;
      const double b = (j < M && kk <= j) ? md.Lam[(size_t)j * M + kk] : 0.0;
      acc = TGP_MFMA(a, b, acc);
    }
    double* S = ws + p.S_;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) S[(size_t)(ti * 16 + q + 4 * rr) * MP + tj * 16 + r] = acc[rr];
    return;
  }

  // ---------------- block 0 ----------------
  const int LD = MP + 1;                 // padded LDS leading dimension
  double* A = sm;                        // MP x LD : K_MM -> L (lower) ; J^T is built in the strict upper part
  double* dinv = sm + (size_t)MP * LD;   // MP : 1 / L_ii
  double* zs = dinv + MP;                // MP x DP
  double* red = zs + (size_t)MP * DP;    // 32 : block reduction scratch
  __shared__ int s_info, s_nan;
  double* hdr = ws + p.hdr;

  if (tid == 0) { s_info = 0; s_nan = 0; }
  // transforms: lengthscale = softplus(raw) (gpytorch Positive constraint), outputscale likewise
  if (tid < 16) {
    double l = 1.0;
    if (tid < D) l = softplus_d(md.raw_ls[tid]);
    ws[p.ls + tid] = l;
    ws[p.ils + tid] = tid < D ? 1.0 / l : 0.0;
    red[tid] = tid < D ? 1.0 / l : 0.0;
  }
  __syncthreads();
  const double s2 = softplus_d(md.raw_os[0]);
  for (int i = tid; i < MP * DP; i += PREP_THREADS) {
    const int mrow = i / DP, d = i % DP;
    const double z = (mrow < M && d < D) ? md.Z[(size_t)mrow * D + d] * red[d] : 0.0;
    zs[i] = z;
    ws[p.Zs + i] = z;
  }
  for (int i = tid; i < MP; i += PREP_THREADS) ws[p.mpad + i] = i < M ? md.m[i] : 0.0;
  // flow parameter transforms (softplus where the reference applies it) and their derivatives
  if (md.program != nullptr) {
    for (int b = tid; b < p.nblk; b += PREP_THREADS) {
      const int kind = md.program[4 * b], K = md.program[4 * b + 1], poff = md.program[4 * b + 2],
                flags = md.program[4 * b + 3];
      if (flags & TGP_FLAG_PER_ROW) continue;
      if (kind == TGP_FLOW_AFFINE || kind == TGP_FLOW_SAL) {
        const int jr = kind == TGP_FLOW_AFFINE ? 0 : 1;  // the restricted parameter: affine.a / SAL.b
        for (int j = 0; j < 2; ++j) {
          const double x = md.theta[poff + j];
          const bool res = (flags & TGP_FLAG_RESTRICT) && j == jr;
          ws[p.tp + poff + j] = res ? softplus_d(x) : x;
          ws[p.tg + poff + j] = res ? sigmoid_d(x) : 1.0;
        }
      } else {
        for (int j = 0; j < 4 * K; ++j) {
          const double x = md.theta[poff + j];
          const bool res = (j & 1);  // b_k, d_k: TanhFlow set_restrictions=True inside StepFlow (flow.py:1075)
          ws[p.tp + poff + j] = res ? softplus_d(x) : x;
          ws[p.tg + poff + j] = res ? sigmoid_d(x) : 1.0;
        }
      }
    }
  }
  // masked Lq / Lq^T copies, KL pieces
  double kl_part = 0.0;
  for (int i = tid; i < MP * MP; i += PREP_THREADS) {
    const int r = i / MP, c = i % MP;
    double x = 0.0;
    if (r < M && c <= r) {
      x = md.Lam[(size_t)r * M + c];
      kl_part += x * x;
      if (c == r) kl_part -= log(x * x);
    }
    ws[p.Lq + i] = x;
    ws[p.LqT + (size_t)c * MP + r] = x;
  }
  for (int i = tid; i < M; i += PREP_THREADS) kl_part += md.m[i] * md.m[i];
  __syncthreads();  // zs visible
  // K_MM (gpytorch ScaleKernel(RBFKernel): sigma^2 exp(-1/2 |zs_i - zs_j|^2)); identity on the padding
  bool has_nan = false;
  for (int i = tid; i < MP * MP; i += PREP_THREADS) {
    const int r = i / MP, c = i % MP;
    double k;
    if (r < M && c < M) {
      double d2 = 0.0;
      for (int d = 0; d < DP; ++d) {
        const double t = zs[r * DP + d] - zs[c * DP + d];
        d2 += t * t;
      }
      k = s2 * exp(-0.5 * d2);
      has_nan |= (k != k);
      ws[p.Kmm + i] = k;
      if (r == c) k += md.jitter;
    } else {
      k = (r == c) ? 1.0 : 0.0;
      ws[p.Kmm + i] = 0.0;
    }
    A[r * LD + c] = k;
  }
  if (has_nan) s_nan = 1;
  // KL block reduction
  kl_part = wave_sum(kl_part);
  if ((tid & 63) == 0) red[16 + (tid >> 6)] = kl_part;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < PREP_THREADS / 64; ++i) s += red[16 + i];
    hdr[H_S2] = s2;
    hdr[H_KL] = 0.5 * (s - (double)M);
    hdr[H_ETA] = md.log_var_noise[0];
    hdr[H_EINV] = exp(-md.log_var_noise[0]);  // 1/positive_transform (dsp/utils.py:39-41, 'exp')
    hdr[H_SIG_OS] = sigmoid_d(md.raw_os[0]);
  }

  // ---- Cholesky, right-looking, in LDS (torch.cholesky at dsp/utils.py:239) ----
  for (int j = 0; j < M; ++j) {
    __syncthreads();
    const double d = A[j * LD + j];
    if (!(d > 0.0)) {  // also catches NaN
      if (tid == 0 && s_info == 0) s_info = j + 1;
    }
    const double dj = sqrt(d);
    const double inv = 1.0 / dj;
    __syncthreads();
    for (int i = j + tid; i < M; i += PREP_THREADS) A[i * LD + j] = (i == j) ? dj : A[i * LD + j] * inv;
    __syncthreads();
    // trailing update of the lower triangle: rows i > j, cols j < k <= i
    const int rem = M - j - 1;
    for (int e = tid; e < rem * rem; e += PREP_THREADS) {
      const int i = j + 1 + e / rem, k = j + 1 + e % rem;
      if (k <= i) A[i * LD + k] -= A[i * LD + j] * A[k * LD + j];
    }
  }
  __syncthreads();
  if (tid < MP) dinv[tid] = 1.0 / A[tid * LD + tid];
  __syncthreads();
  // ---- J = L^-1 by forward substitution, one thread per column c; column c of J is stored as row c of
  //      the strict upper triangle of A (J^T), the diagonal is dinv.  x_i = -(sum_{c<=k<i} L_ik x_k)/L_ii ----
  if (tid < M) {
    const int c = tid;
    for (int i = c + 1; i < M; ++i) {
      double s = A[i * LD + c] * dinv[c];
      for (int k = c + 1; k < i; ++k) s += A[i * LD + k] * A[c * LD + k];
      A[c * LD + i] = -s * dinv[i];
    }
  }
  __syncthreads();
  // ---- write L, J, J^T (padding = identity) ----
  for (int i = tid; i < MP * MP; i += PREP_THREADS) {
    const int r = i / MP, c = i % MP;
    double l = 0.0, jv = 0.0;
    if (c < r) { l = A[r * LD + c]; jv = A[c * LD + r]; }
    else if (c == r) { l = A[r * LD + r]; jv = dinv[r]; }
    ws[p.L + i] = l;
    ws[p.J + i] = jv;
    ws[p.JT + (size_t)c * MP + r] = jv;
  }
  if (tid == 0) {
    status[0] = s_info;
    status[1] = s_nan;
  }
}

size_t prep_a_lds_bytes(const Plan& p) {
  return ((size_t)p.MP * (p.MP + 1) + p.MP + (size_t)p.MP * p.DP + 32) * sizeof(double);
}

// ---------------------------------------------------------------------------------------------------
// k_prep_b: H' = J^T S - J^T (tiles), w = J^T m (last block)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_prep_b(Plan p, double* __restrict__ ws) {
  const int MP = p.MP, MT = p.MT;
  const int l = threadIdx.x, r = l & 15, q = l >> 4;
  const double* J = ws + p.J;
  if ((int)blockIdx.x == MT * MT) {
    const double* mp = ws + p.mpad;
    for (int i = l; i < MP; i += 64) {
      double s = 0.0;
      for (int k = i; k < MP; ++k) s += J[(size_t)k * MP + i] * mp[k];
      ws[p.w + i] = s;
    }
    return;
  }
  const int ti = blockIdx.x / MT, tj = blockIdx.x % MT;
  d4 acc = {0, 0, 0, 0};
  acc = tile_mm<true, false>(J, ws + p.S_, MP, ti * 16, tj * 16, ti * 16, MP, acc);  // (J^T)[i,k] = 0 for k < i
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int row = ti * 16 + q + 4 * rr, col = tj * 16 + r;
    ws[p.Hp + (size_t)row * MP + col] = acc[rr] - J[(size_t)col * MP + row];
  }
}

// ---------------------------------------------------------------------------------------------------
// k_reduce: sum the per-block slabs of the row kernel; G tiles are expanded to a full symmetric matrix
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce(Plan p, double* __restrict__ ws) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.slab_len) return;
  const double* sl = ws + p.slabs + e;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = 0;
  for (; b + 4 <= p.nblocks; b += 4) {
    s0 += sl[(size_t)b * p.slab_len];
    s1 += sl[(size_t)(b + 1) * p.slab_len];
    s2 += sl[(size_t)(b + 2) * p.slab_len];
    s3 += sl[(size_t)(b + 3) * p.slab_len];
  }
  for (; b < p.nblocks; ++b) s0 += sl[(size_t)b * p.slab_len];
  const double s = (s0 + s1) + (s2 + s3);
  if (e < p.slab_T) {
    // tile t = (ti,tj), ti >= tj, row-major over the lower triangle of tiles
    const int t = (int)(e >> 8), in = (int)(e & 255), row = in >> 4, col = in & 15;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    double* G = ws + p.Gf;
    G[(size_t)(ti * 16 + row) * p.MP + tj * 16 + col] = s;
    G[(size_t)(tj * 16 + col) * p.MP + ti * 16 + row] = s;
  } else {
    ws[p.red + e] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// k_bwd1: Lbar = -tril(w s^T + 2 H' G)   and   LamB = 2 tril(G Lq)     (grid: 2 * MT*MT waves)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_bwd1(Plan p, double* __restrict__ ws) {
  const int MP = p.MP, MT = p.MT;
  const int l = threadIdx.x, r = l & 15, q = l >> 4;
  const int op = blockIdx.x / (MT * MT), t = blockIdx.x % (MT * MT);
  const int ti = t / MT, tj = t % MT;
  double* out = ws + (op == 0 ? p.Lb : p.LamB);
  d4 acc = {0, 0, 0, 0};
  if (ti >= tj) {
    if (op == 0) acc = tile_mm<false, false>(ws + p.Hp, ws + p.Gf, MP, ti * 16, tj * 16, 0, MP, acc);
    else acc = tile_mm<true, false>(ws + p.Gf, ws + p.Lq, MP, ti * 16, tj * 16, tj * 16, MP, acc);  // G symmetric; Lq[k,j]=0 for k<j
  }
  const double* w = ws + p.w;
  const double* sv = ws + p.red + p.slab_S;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int row = ti * 16 + q + 4 * rr, col = tj * 16 + r;
    double x = 0.0;
    if (col <= row) x = (op == 0) ? -(w[row] * sv[col] + 2.0 * acc[rr]) : 2.0 * acc[rr];
    out[(size_t)row * MP + col] = x;
  }
}

// ---------------------------------------------------------------------------------------------------
// k_bwd2: Q = Phi(L^T Lbar) + Phi(L^T Lbar)^T   (lower tiles computed, mirrored on store)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_bwd2(Plan p, double* __restrict__ ws) {
  const int MP = p.MP;
  const int l = threadIdx.x, r = l & 15, q = l >> 4;
  int ti = 0;
  const int t = blockIdx.x;
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  const int tj = t - ti * (ti + 1) / 2;
  const double* L = ws + p.L;
  const double* Lb = ws + p.Lb;
  double* Q = ws + p.Q;
  d4 acc = {0, 0, 0, 0};
  acc = tile_mm<true, false>(L, Lb, MP, ti * 16, tj * 16, ti * 16, MP, acc);  // (L^T)[i,k] = L[k,i] = 0 for k < i
  if (ti != tj) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = ti * 16 + q + 4 * rr, col = tj * 16 + r;
      Q[(size_t)row * MP + col] = acc[rr];
      Q[(size_t)col * MP + row] = acc[rr];
    }
  } else {
    d4 tr = {0, 0, 0, 0};  // (L^T Lbar)^T tile = Lbar^T L
    tr = tile_mm<true, false>(Lb, L, MP, ti * 16, tj * 16, ti * 16, MP, tr);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int rl = q + 4 * rr;
      const double x = (r <= rl) ? acc[rr] : tr[rr];  // below/on diagonal: M1[r][c]; above: M1[c][r]
      Q[(size_t)(ti * 16 + rl) * MP + tj * 16 + r] = x;
    }
  }
}

// k_bwd3: Y = J^T Q ;  k_bwd4: Ks = 1/2 Y J
__global__ __launch_bounds__(64) void k_bwd3(Plan p, double* __restrict__ ws) {
  const int MP = p.MP, MT = p.MT;
  const int l = threadIdx.x, r = l & 15, q = l >> 4;
  const int ti = blockIdx.x / MT, tj = blockIdx.x % MT;
  d4 acc = {0, 0, 0, 0};
  acc = tile_mm<true, false>(ws + p.J, ws + p.Q, MP, ti * 16, tj * 16, ti * 16, MP, acc);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) ws[p.Y + (size_t)(ti * 16 + q + 4 * rr) * MP + tj * 16 + r] = acc[rr];
}

__global__ __launch_bounds__(64) void k_bwd4(Plan p, double* __restrict__ ws) {
  const int MP = p.MP, MT = p.MT;
  const int l = threadIdx.x, r = l & 15, q = l >> 4;
  const int ti = blockIdx.x / MT, tj = blockIdx.x % MT;
  d4 acc = {0, 0, 0, 0};
  acc = tile_mm<false, false>(ws + p.Y, ws + p.J, MP, ti * 16, tj * 16, tj * 16, MP, acc);  // J[k,j] = 0 for k < j
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) ws[p.Ks + (size_t)(ti * 16 + q + 4 * rr) * MP + tj * 16 + r] = 0.5 * acc[rr];
}

// ---------------------------------------------------------------------------------------------------
// k_bwd5: assemble every output gradient + the scalars (single block)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bwd5(Plan p, tgp_model md, tgp_grads g, double* __restrict__ out,
                                               double* __restrict__ ws) {
  const int tid = threadIdx.x;
  const int M = p.M, D = p.D, MP = p.MP, DP = p.DP, CT16 = p.CT16;
  __shared__ double lsacc[16];
  __shared__ double wsum[8];
  const double* hdr = ws + p.hdr;
  const double s2 = hdr[H_S2];
  const double* Ks = ws + p.Ks;
  const double* Kmm = ws + p.Kmm;
  const double* Zs = ws + p.Zs;
  const double* T = ws + p.red + p.slab_T;  // [MP][CT16]: cols [0,DP) = T1, [DP,2DP) = T2, 2DP = T0
  const double* C = ws + p.red + p.slab_C;
  if (tid < 16) lsacc[tid] = 0.0;
  __syncthreads();
  double ep_sum = 0.0;  // sum_ij Ks*Kmm + sum_j T0_j
  for (int it = tid; it < M * D; it += 256) {
    const int j = it / D, d = it % D;
    const double zj = Zs[j * DP + d];
    const double t0 = T[j * CT16 + 2 * DP], t1 = T[j * CT16 + d], t2 = T[j * CT16 + DP + d];
    double zsb = t1 - zj * t0;
    double lt = t2 - 2.0 * zj * t1 + zj * zj * t0;
    double es = 0.0;
    for (int i = 0; i < M; ++i) {
      const double ep = Ks[(size_t)i * MP + j] * Kmm[(size_t)i * MP + j];
      const double dz = Zs[i * DP + d] - zj;
      zsb += 2.0 * ep * dz;
      lt += ep * dz * dz;
      es += ep;
    }
    g.Z[it] = zsb * ws[p.ils + d];
    atomicAdd(&lsacc[d], lt);
    if (d == 0) ep_sum += es + t0;
  }
  ep_sum = wave_sum(ep_sum);
  if ((tid & 63) == 0) wsum[tid >> 6] = ep_sum;
  __syncthreads();
  if (tid < D) g.raw_ls[tid] = lsacc[tid] * ws[p.ils + tid] * sigmoid_d(md.raw_ls[tid]);
  if (tid == 0) {
    const double tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const double s2b = C[C_SVB] + tot / s2;
    g.raw_os[0] = s2b * hdr[H_SIG_OS];
    g.log_var_noise[0] = C[C_ETAB];
    const double ell = C[C_ELL], kl = hdr[H_KL];
    out[0] = ell - kl;
    out[1] = ell;
    out[2] = kl;
    out[3] = 0.0;
  }
  // m: sbar = A mubar summed over rows, minus the KL part (KL' = m)
  const double* sv = ws + p.red + p.slab_S;
  for (int i = tid; i < M; i += 256) g.m[i] = sv[i] - md.kl_scale * md.m[i];
  // Lam: tril only; strict upper triangle receives exactly zero (mask at use, sparse_MF_SP.py:344-345)
  const double* LamB = ws + p.LamB;
  for (int it = tid; it < M * M; it += 256) {
    const int r = it / M, c = it % M;
    double x = 0.0;
    if (c <= r) {
      const double lam = md.Lam[it];
      x = LamB[(size_t)r * MP + c] - md.kl_scale * (c == r ? lam - 1.0 / lam : lam);
    }
    g.Lam[it] = x;
  }
  if (g.theta != nullptr)
    for (int i = tid; i < p.P; i += 256) g.theta[i] = C[C_THETA + i];
}

// ---------------------------------------------------------------------------------------------------
// stand-alone entry points that live on the M x M side
// ---------------------------------------------------------------------------------------------------
__global__ void k_kmm(const double* __restrict__ Z, const double* __restrict__ raw_ls, const double* __restrict__ raw_os,
                      int M, int D, double jitter, double* __restrict__ K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * M) return;
  const int r = i / M, c = i % M;
  double d2 = 0.0;
  for (int d = 0; d < D; ++d) {
    const double il = 1.0 / softplus_d(raw_ls[d]);
    const double t = Z[r * D + d] * il - Z[c * D + d] * il;
    d2 += t * t;
  }
  K[i] = softplus_d(raw_os[0]) * exp(-0.5 * d2) + (r == c ? jitter : 0.0);
}

__global__ void k_knm(const double* __restrict__ X, const double* __restrict__ Z, const double* __restrict__ raw_ls,
                      const double* __restrict__ raw_os, int N, int M, int D, double* __restrict__ K) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * M) return;
  const size_t r = i / M;
  const int c = (int)(i % M);
  double d2 = 0.0;
  for (int d = 0; d < D; ++d) {
    const double il = 1.0 / softplus_d(raw_ls[d]);
    const double t = X[r * D + d] * il - Z[c * D + d] * il;
    d2 += t * t;
  }
  K[i] = softplus_d(raw_os[0]) * exp(-0.5 * d2);
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

int launch_prepare(const Plan& p, const tgp_model& md, double* ws, int32_t* status, hipStream_t st) {
  const size_t lds = prep_a_lds_bytes(p);
  static size_t lds_cur = 48 * 1024;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(k_prep_a), lds, &lds_cur)) return rc;
  hipLaunchKernelGGL(k_prep_a, dim3(1 + p.MT * p.MT), dim3(PREP_THREADS), lds, st, p, md, ws, status);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_prep_b, dim3(p.MT * p.MT + 1), dim3(64), 0, st, p, ws);
  LAUNCH_CHECK();
  return 0;
}

int launch_backward_mm(const Plan& p, const tgp_model& md, const tgp_grads& g, double* out, double* ws, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce, dim3((unsigned)((p.slab_len + 255) / 256)), dim3(256), 0, st, p, ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bwd1, dim3(2 * p.MT * p.MT), dim3(64), 0, st, p, ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bwd2, dim3(p.ntri), dim3(64), 0, st, p, ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bwd3, dim3(p.MT * p.MT), dim3(64), 0, st, p, ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bwd4, dim3(p.MT * p.MT), dim3(64), 0, st, p, ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bwd5, dim3(1), dim3(256), 0, st, p, md, g, out, ws);
  LAUNCH_CHECK();
  return 0;
}

int launch_kmm(const double* Z, const double* raw_ls, const double* raw_os, int M, int D, double jitter, double* K,
               hipStream_t st) {
  hipLaunchKernelGGL(k_kmm, dim3((M * M + 255) / 256), dim3(256), 0, st, Z, raw_ls, raw_os, M, D, jitter, K);
  LAUNCH_CHECK();
  return 0;
}

int launch_knm(const double* X, const double* Z, const double* raw_ls, const double* raw_os, int N, int M, int D,
               double* K, hipStream_t st) {
  const size_t tot = (size_t)N * M;
  hipLaunchKernelGGL(k_knm, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, X, Z, raw_ls, raw_os, N, M, D, K);
  LAUNCH_CHECK();
  return 0;
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
