// tgp_lik.hip -- stand-alone likelihood, flow, prediction and optimiser kernels.
//
// These serve the evaluation path, the operator-level API (K10/K11 of SURVEY 2.2) and
// the trainer; the training step itself uses the fused row kernel (tgp_rows.hpp), which shares the
// flow code in tgp_dev.hpp.  All of them are HBM/VALU-bound streaming kernels: one thread per row,
// coalesced row-major reads, lane-private LDS accumulators, two-pass deterministic reductions.
#include "tgp_dev.hpp"
#include "tgp_launch.hpp"

namespace tgp {

#define LAUNCH_CHECK()                                              \
  do {                                                              \
    hipError_t e_ = hipGetLastError();                              \
    if (e_ != hipSuccess) return set_error(e_, __FILE__, __LINE__); \
  } while (0)

// k_ell_flow: lanes per data row.  4 by default (64 rows per workgroup); 16 for small problems (16 rows per workgroup): a
// rank's 1 250 rows of an 8-GPU minibatch were 20 workgroups, each lane walking 8 quadrature nodes through 30 tanh steps
// and back -- 91 us of one dependent chain, whatever N; with 16 lanes per row a lane has 2 nodes and 79 workgroups run.
// Round 6: 32 lanes per row (8 rows per workgroup, one node per lane at S = 32) below 2 560 rows -- the 1 250-row share again: 79
// workgroups leave three quarters of the SIMDs without a wave, and a lane's chain is as long as its nodes.
#define ELL_LPR16_MAXN 12288
#define ELL_LPR32_MAXN 2560
static int ell_flow_lpr(int N) { return N <= ELL_LPR32_MAXN ? 32 : (N <= ELL_LPR16_MAXN ? 16 : 4); }

// Sized for ANY call with at most N rows: a caller that sizes once for its largest chunk (the general-M path) may
// launch a ragged last chunk that falls into the 16-lanes-per-row mode and then has MORE workgroups than the largest one.
size_t lik_workspace_doubles(int N, int P, int RP) {
  const size_t nb16 = (size_t)((N < ELL_LPR16_MAXN ? N : ELL_LPR16_MAXN) + 15) / 16, nb64 = (size_t)(N + 63) / 64;
  const size_t nb32 = (size_t)((N < ELL_LPR32_MAXN ? N : ELL_LPR32_MAXN) + 7) / 8;
  const size_t nbm = nb16 > nb64 ? nb16 : nb64;
  const size_t nb = (nbm > nb32 ? nbm : nb32) + 1;  // k_ell_flow: one partial per workgroup
  return nb * (size_t)(2 + P) + 2 * (size_t)P + 64;
}

// ---------------------------------------------------------------------------------------------------
// SVGP closed form: likelihoods/GaussianLinearMean.py:60-87 + dsp/utils.py:164-195
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ell_gauss(const double* __restrict__ Y, const double* __restrict__ mu,
                                                    const double* __restrict__ v, int N,
                                                    const double* __restrict__ lvn, double scale,
                                                    double* __restrict__ part, double* __restrict__ g_mu,
                                                    double* __restrict__ g_v) {
  __shared__ double red[8];
  const int n = blockIdx.x * 256 + threadIdx.x;
  const double eta = lvn[0], einv = exp(-eta);
  double e = 0.0, et = 0.0;
  if (n < N) {
    const double r = Y[n] - mu[n];
    e = -0.5 * TGP_LOG_2PI_REF - 0.5 * eta - 0.5 * einv * (r * r + v[n]);
    et = -0.5 + 0.5 * einv * (r * r + v[n]);
    if (g_mu) g_mu[n] = scale * einv * r;
    if (g_v) g_v[n] = -0.5 * scale * einv;
  }
  e = wave_sum(e); et = wave_sum(et);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = e; red[4 + (threadIdx.x >> 6)] = et; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = scale * (red[0] + red[1] + red[2] + red[3]);
    part[2 * blockIdx.x + 1] = scale * (red[4] + red[5] + red[6] + red[7]);
  }
}

// sum `nb` partial vectors of length `len` (stride len) into out[0..split) / out2; block = 32 columns x 8 row groups
__global__ __launch_bounds__(256) void k_sum_parts(const double* __restrict__ part, int nb, int len,
                                                    double* __restrict__ out, double* __restrict__ out2, int split) {
  __shared__ double red[8][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5, j = blockIdx.x * 32 + c;
  double s0 = 0.0, s1 = 0.0;
  if (j < len) {
    int b = g;
    // (eight partials requested before the first add, added in the order of the two-at-a-time loop that follows: with two
    //  loads in flight per thread the 625 workgroup partials of a 10 000-row likelihood were 39 dependent round trips, 15 us)
    for (; b + 56 < nb; b += 64) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(b + 8 * u) * len + j];
#pragma unroll
      for (int u = 0; u < 8; u += 2) { s0 += t[u]; s1 += t[u + 1]; }
    }
    for (; b + 8 < nb; b += 16) {
      s0 += part[(size_t)b * len + j];
      s1 += part[(size_t)(b + 8) * len + j];
    }
    if (b < nb) s0 += part[(size_t)b * len + j];
  }
  red[g][c] = s0 + s1;
  __syncthreads();
  if (g == 0 && j < len) {
    const double s = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
    if (j < split) out[j] = s;
    else if (out2) out2[j - split] = s;
  }
}

// transformed shared flow parameters into LDS (same rule as k_prep_a); whole block, ends with a barrier
// `ti` (optional): 1 / tp[i] for every shared parameter, so that a step-tanh step takes 1 / softplus(d_k) from the table instead of
// running a reciprocal chain per step and sweep
__device__ inline void flow_params_lds(const tgp_model& md, const FlowProg& fp, double* tp, double* tg, double* ti = nullptr) {
  for (int b = threadIdx.x; b < fp.nblk; b += blockDim.x) {
    const int kind = fp.blk[4 * b], K = fp.blk[4 * b + 1], poff = fp.blk[4 * b + 2], flags = fp.blk[4 * b + 3];
    if (flags & TGP_FLAG_PER_ROW) continue;
    const int np = kind == TGP_FLOW_STEPTANH ? 4 * K : 2;
    for (int j = 0; j < np; ++j) {
      const double x = md.theta[poff + j];
      bool res;
      if (kind == TGP_FLOW_STEPTANH) res = (j & 1);
      else res = (flags & TGP_FLAG_RESTRICT) && j == (kind == TGP_FLOW_AFFINE ? 0 : 1);
      const double tv = res ? softplus_d(x) : x;
      tp[poff + j] = tv;
      tg[poff + j] = res ? sigmoid_d(x) : 1.0;
      if (ti != nullptr) ti[poff + j] = rcp_fast(tv);
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// TGP quadrature likelihood with gradients (likelihoods/GaussianNonLinearMean.py:64-150), four lanes per row
// ---------------------------------------------------------------------------------------------------

// LPR lanes share a data row (64 / LPR rows per wave: row = lane % RW, node group = lane / RW), NB nodes in flight per
// lane.  The launcher picks LPR from N so that a chunk of ~16k rows still yields ~1000 workgroups.
template <int LPR, int NB>
__global__ __launch_bounds__(256) void k_ell_flow(tgp_model md, FlowProg fp, const double* __restrict__ Y,
                                                   const double* __restrict__ mu, const double* __restrict__ v,
                                                   const double* __restrict__ rowp, double* __restrict__ part,
                                                   double* __restrict__ g_mu, double* __restrict__ g_v,
                                                   double* __restrict__ g_rowp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, P = md.P, RP = md.RP;
  const int nb = fp.nblk > 0 ? fp.nblk : 1;
  constexpr int RW = 64 / LPR;                              // rows per wave
  double* stack = sm;                                       // nblk * NB * 256: block inputs (checkpoint mode)
  double* accw = stack + (size_t)nb * NB * 256;             // 4 waves x P: per-wave shared-parameter accumulators
  double* accr = accw + (size_t)4 * (P > 0 ? P : 1);        // RP * 256 (per-row parameters, lane-private)
  double* red = accr + (size_t)RP * 256;                    // 16
  double* tp = red + 16;                                    // P+2
  double* tg = tp + (P + 2) / 2 * 2;                        // P+2
  double* ti = tg + (P + 2) / 2 * 2;                        // P+2: reciprocals (flow_rcp_param)
  for (int i = tid; i < 4 * (P > 0 ? P : 1) + RP * 256; i += 256) accw[i] = 0.0;
  flow_params_lds(md, fp, tp, tg, ti);
  // lane group q takes the quadrature nodes s = q + LPR (NB j + u).  Every lane runs the same trip count (wave-wide
  // sums inside the reverse sweep); nodes past S and padding rows carry weight 0.
  const int qn = lane / RW;
  const int n = blockIdx.x * (4 * RW) + wave * RW + (lane % RW);
  auto group_sum = [&](double x) {   // over the LPR lanes of a row: lanes differing in the bits above log2(RW)
#pragma unroll
    for (int o = RW; o < 64; o <<= 1) x += __shfl_xor(x, o);
    return x;
  };
  const bool valid = n < md.N;
  const int nc = valid ? n : md.N - 1;
  const double eta = md.log_var_noise[0], einv = exp(-eta);
  FlowDev F{fp.blk, fp.nblk, tp, tg, ti};
  double ellp = 0.0, etap = 0.0, cm = 0.0, cv = 0.0;
  const double m_ = mu[nc], sq = sqrt(2.0 * v[nc]), y = Y[nc];
  const double* rp = rowp ? rowp + (size_t)nc * RP : nullptr;
  double* aw = accw + (size_t)wave * (P > 0 ? P : 1);
  for (int s0 = 0; s0 < md.S; s0 += LPR * NB) {
    double f[NB], c[NB], xsn[NB], wsn[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int s = s0 + LPR * u + qn, sc = s < md.S ? s : md.S - 1;
      xsn[u] = md.xs[sc];
      wsn[u] = (valid && s < md.S) ? md.wn[sc] : 0.0;
      f[u] = m_ + sq * xsn[u];
    }
    flow_forward_ckpt<NB>(F, f, rp, stack + tid, 256);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const double r = y - f[u];
      ellp += wsn[u] * (-0.5 * TGP_LOG_2PI_REF - 0.5 * eta - 0.5 * einv * r * r);
      etap += wsn[u] * (-0.5 + 0.5 * einv * r * r);
      c[u] = md.scale * einv * wsn[u] * r;
    }
    flow_backward_ckpt<NB>(F, c, rp, stack + tid, 256, aw, lane, accr + tid, 256);
#pragma unroll
    for (int u = 0; u < NB; ++u) { cm += c[u]; cv += c[u] * xsn[u]; }
  }
  cm = group_sum(cm);
  cv = group_sum(cv);
  for (int j = 0; j < RP; ++j) {
    const double a = group_sum(accr[(size_t)j * 256 + tid]);
    if (valid && qn == 0 && g_rowp) g_rowp[(size_t)n * RP + j] = a;
  }
  if (valid && qn == 0) {
    if (g_mu) g_mu[n] = cm;
    if (g_v) g_v[n] = cv / sq;
  }
  ellp = wave_sum(ellp); etap = wave_sum(etap);
  if (lane == 0) { red[wave] = ellp; red[4 + wave] = etap; }
  __syncthreads();
  double* pb = part + (size_t)blockIdx.x * (2 + P);
  if (tid == 0) {
    pb[0] = md.scale * (red[0] + red[1] + red[2] + red[3]);
    pb[1] = md.scale * (red[4] + red[5] + red[6] + red[7]);
  }
  for (int j = tid; j < P; j += 256) pb[2 + j] = (accw[j] + accw[P + j]) + (accw[2 * P + j] + accw[3 * P + j]);
}

// ---------------------------------------------------------------------------------------------------
// flow evaluation: G, dG/df, log dG/df over an (S,N) array (row n = idx % N)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_eval(tgp_model md, FlowProg fp, const double* __restrict__ f, size_t total, int N,
                                                    const double* __restrict__ rowp, double* __restrict__ G,
                                                    double* __restrict__ dG, double* __restrict__ logdG,
                                                    double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* tp = reinterpret_cast<double*>(smem_raw);
  double* tg = tp + (md.P + 2) / 2 * 2;
  double* ti = tg + (md.P + 2) / 2 * 2;
  __shared__ double redl[4];
  flow_params_lds(md, fp, tp, tg, ti);
  FlowDev F{fp.blk, fp.nblk, tp, tg, ti};
  // four elements per thread, a grid stride apart (coalesced), evaluated stage by stage (flow_forward_n)
  constexpr int NB = 4;
  const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  double fv[NB], der[NB];
  const double* rp[NB];
  TGP_EACH(u, NB) {
    const size_t i = i0 + u * stride;
    const size_t ic = i < total ? i : 0;
    fv[u] = f[ic];
    rp[u] = rowp ? rowp + (ic % N) * md.RP : nullptr;
  }
  if (dG || logdG || part) flow_forward_n<NB, true>(F, fv, rp, der);
  else flow_forward_n<NB, false>(F, fv, rp, der);
  if (part) {
    // fused log-Jacobian accumulation: sum of log dG/df over this workgroup's elements, fixed order (lane partials ->
    // butterfly over the wave -> the four waves in LDS); the launcher's second kernel adds the workgroups' partials
    double sl = 0.0;
    TGP_EACH(u, NB) sl += (i0 + u * stride < total) ? log(der[u]) : 0.0;
    sl = wave_sum(sl);
    if ((threadIdx.x & 63) == 0) redl[threadIdx.x >> 6] = sl;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (redl[0] + redl[1]) + (redl[2] + redl[3]);
  }
  if (logdG) {
    double lg[NB];
    TGP_EACH(u, NB) lg[u] = der[u];
    TGP_EACH(u, NB) {
      const size_t i = i0 + u * stride;
      if (i < total) logdG[i] = log(lg[u]);
    }
  }
  TGP_EACH(u, NB) {
    const size_t i = i0 + u * stride;
    if (i < total) {
      if (G) G[i] = fv[u];
      if (dG) dG[i] = der[u];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// prediction given q(f) moments: m1, m2 and per-row test log-likelihood (without the -0.5 log(pi) constant)
//   flow : GaussianNonLinearMean.marginal_moments (:176-203) ; sparse_MF_SP.test_log_likelihood (:705-776)
//   gauss: GaussianLinearMean.marginal_moments (:89-118)     ; sparse_MF_SP.py:786-799
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_predict(tgp_model md, FlowProg fp, const double* __restrict__ mu,
                                                  const double* __restrict__ v, const double* __restrict__ rowp,
                                                  const double* __restrict__ Y, double Y_std, double* __restrict__ m1o,
                                                  double* __restrict__ m2o, double* __restrict__ logp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* tp = reinterpret_cast<double*>(smem_raw);
  double* tg = tp + (md.P + 2) / 2 * 2;
  if (md.lik == TGP_LIK_FLOW) flow_params_lds(md, fp, tp, tg);
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= md.N) return;
  const double noise = exp(md.log_var_noise[0]);
  if (md.lik == TGP_LIK_GAUSS) {
    const double m1 = mu[n], m2 = noise + v[n];
    if (m1o) m1o[n] = m1;
    if (m2o) m2o[n] = m2;
    if (logp && Y) {
      const double sd = Y_std * sqrt(m2), var = sd * sd, r = Y_std * Y[n] - Y_std * m1;
      logp[n] = -0.5 * (TGP_LOG_2PI_REF + log(var) + r * r / var);
    }
    return;
  }
  FlowDev F{fp.blk, fp.nblk, tp, tg};
  const double* rp = rowp ? rowp + (size_t)n * md.RP : nullptr;
  const double m_ = mu[n], sq = sqrt(2.0 * v[n]);
  const double sdy = Y_std * sqrt(noise), var = sdy * sdy;
  const double yy = Y ? Y_std * Y[n] : 0.0;
  double m1 = 0.0, e2 = 0.0, mx = -INFINITY, se = 0.0;
  const double lvar = log(var), ivar = 1.0 / var;
  constexpr int NB = 4;  // quadrature nodes in flight (stage-by-stage evaluation, flow_forward_n)
  for (int s0 = 0; s0 < md.S; s0 += NB) {
    double g[NB], der[NB], wsn[NB];
    const double* rpn[NB];
    TGP_EACH(u, NB) {
      const int s = s0 + u < md.S ? s0 + u : md.S - 1;
      g[u] = m_ + sq * md.xs[s];
      wsn[u] = s0 + u < md.S ? md.wn[s] : 0.0;
      rpn[u] = rp;
    }
    flow_forward_n<NB, false>(F, g, rpn, der);
    TGP_EACH(u, NB) {
      if (s0 + u < md.S) {
        m1 += wsn[u] * g[u];
        e2 += wsn[u] * g[u] * g[u];
        if (logp && Y) {
          // log w_s = log(wn_s) + 0.5 log(pi); the caller adds the reference's constants
          const double r = yy - Y_std * g[u];
          const double t = log(wsn[u]) - 0.5 * (TGP_LOG_2PI_REF + lvar + r * r * ivar);
          if (t > mx) { se = se * exp(mx - t) + 1.0; mx = t; }
          else se += exp(t - mx);
        }
      }
    }
  }
  if (m1o) m1o[n] = m1;
  if (m2o) m2o[n] = noise + e2 - m1 * m1;
  if (logp && Y) logp[n] = mx + log(se);
}

// ---------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics; dsp/trainers/optimizers.py:12)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam(double* __restrict__ p, const double* __restrict__ g,
                                               double* __restrict__ m, double* __restrict__ v, int64_t n, double lr,
                                               double b1, double b2, double eps, double wd, double bc1, double bc2s,
                                               double sign) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double gi = sign * g[i];
  if (wd != 0.0) gi += wd * p[i];
  const double mi = b1 * m[i] + (1.0 - b1) * gi;
  const double vi = b2 * v[i] + (1.0 - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] -= (lr / bc1) * mi / (sqrt(vi) / bc2s + eps);
}

// device-side step counter variant (hipGraph replay): step = *step_dev + 1, counter bumped by k_step_inc afterwards
__global__ __launch_bounds__(256) void k_adam_dev(double* __restrict__ p, const double* __restrict__ g,
                                                   double* __restrict__ m, double* __restrict__ v, int64_t n, double lr,
                                                   double b1, double b2, double eps, double wd,
                                                   int32_t* __restrict__ step_dev, double sign, int64_t n_plain,
                                                   double ln_b1, double ln_b2, int64_t skip_off, int64_t skip_n) {
  const double step = (double)(step_dev[0] + 1);
  // beta^step = exp(step ln beta), ln beta from the host: the library pow() is ~300 dependent f64 instructions per
  // call and every thread made two of them -- 3 of this kernel's 4.3 us (relative error of the power <= 1e-14 at
  // step 1e5, far inside the 1e-8 the parity tests hold the optimiser trajectory to)
  const double bc1 = 1.0 - exp_fast(step * ln_b1), bc2s = sqrt(1.0 - exp_fast(step * ln_b2));
  // grid-stride: the launcher caps the grid (the ticket below is one atomic per workgroup on ONE word, ~12 ns each:
  // the 4 096 workgroups of a 1 M-parameter buffer spent 50 us queueing for it)
  // [skip_off, skip_off + skip_n): entries another launch of this step has updated already (general-M path: Lam, in k_big_glam)
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n - skip_n; j += (int64_t)gridDim.x * 256) {
    const int64_t i = j < skip_off ? j : j + skip_n;
    double gi = sign * g[i];
    if (wd != 0.0 && i >= n_plain) gi += wd * p[i];  // weight decay only on the tail group
    const double mi = b1 * m[i] + (1.0 - b1) * gi;
    const double vi = b2 * v[i] + (1.0 - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrt(vi) / bc2s + eps);
  }
  // Every block has read step_dev[0] before it takes a ticket; the block that draws the last ticket bumps the
  // counter for the next launch (no second kernel, valid under hipGraph replay).
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = atomicAdd(&step_dev[1], 1);
    if (t == (int)gridDim.x - 1) {
      step_dev[1] = 0;
      atomicAdd(&step_dev[0], 1);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Minibatch rows of a data set that stays resident in HBM (the reference's DataLoader re-collates the rows on the host
// and copies them every step, dsp/data/data.py:86-88, trainers/trainer_base.py:330):
//   Xb[r] = X[index[cursor + offset + r]],  Yb[r] = Y[...]     r < nrows   (index == NULL: the stored order)
// `cursor` = int32[2] {position in the epoch, ticket} on the device: every workgroup reads the position before it
// takes a ticket, the last one advances it (wrapping to 0 at `wrap`), so a captured launch walks through the epoch
// replay after replay without the host touching it.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_rows(const double* __restrict__ X, const double* __restrict__ Y, int N, int D,
                                                      const int32_t* __restrict__ index, int32_t* __restrict__ cursor,
                                                      int offset, int nrows, int advance, int wrap,
                                                      double* __restrict__ Xb, double* __restrict__ Yb) {
  const int c0 = cursor[0];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < (long long)nrows * (D + 1)) {
    const int r = (int)(i / (D + 1)), d = (int)(i % (D + 1));
    int src = c0 + offset + r;
    src = src < N ? src : N - 1;
    if (index != nullptr) src = index[src];
    src = src < 0 ? 0 : (src < N ? src : N - 1);
    if (d < D) Xb[(size_t)r * D + d] = X[(size_t)src * D + d];
    else Yb[r] = Y[src];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = atomicAdd(&cursor[1], 1);
    if (t == (int)gridDim.x - 1) {
      cursor[1] = 0;
      const int nx = c0 + advance;
      cursor[0] = nx >= wrap ? 0 : nx;
    }
  }
}

int launch_gather_rows(const double* X, const double* Y, int N, int D, const int32_t* index, int32_t* cursor, int offset,
                       int nrows, int advance, int wrap, double* Xb, double* Yb, hipStream_t st) {
  const long long n = (long long)nrows * (D + 1);
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, Y, N, D, index, cursor, offset,
                     nrows, advance, wrap, Xb, Yb);
  LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------
int launch_ell_gauss(const double* Y, const double* mu, const double* v, int N, const double* lvn, double scale,
                     double* out, double* g_mu, double* g_v, double* ws, hipStream_t st) {
  const int nb = (N + 255) / 256;
  hipLaunchKernelGGL(k_ell_gauss, dim3(nb), dim3(256), 0, st, Y, mu, v, N, lvn, scale, ws, g_mu, g_v);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(256), 0, st, ws, nb, 2, out, (double*)nullptr, 2);
  LAUNCH_CHECK();
  return 0;
}

static int flow_lds(const tgp_model& md, int nblk, int NB, size_t* bytes) {
  const size_t d = (size_t)(nblk > 0 ? nblk : 1) * NB * 256 + (size_t)(md.P > 0 ? md.P : 1) * 4 + (size_t)md.RP * 256 + 16 +
                   3 * (size_t)(md.P + 2);
  *bytes = d * sizeof(double);
  return *bytes > 160 * 1024 - 1024 ? TGP_E_LDS : 0;
}

int launch_ell_flow(const tgp_model& md, const FlowProg& fp, const double* Y, const double* mu, const double* v, const double* rowp,
                    double* out, double* g_mu, double* g_v, double* g_theta, double* g_rowp, double* ws,
                    hipStream_t st) {
  // nodes in flight per lane: all of the lane's nodes when the checkpoint stack fits (the per-step wave reductions of
  // the shared-parameter partials are then paid once), else fewer
  size_t lds = 0;
  const int LPR = ell_flow_lpr(md.N);
  int NB = LPR == 4 ? (md.S > 16 ? 8 : 4) : (LPR == 16 ? (md.S > 32 ? 4 : (md.S > 16 ? 2 : 1)) : (md.S > 64 ? 4 : (md.S > 32 ? 2 : 1)));
  while (NB > 1 && (flow_lds(md, fp.nblk, NB, &lds) != 0 || lds > 120 * 1024)) NB >>= 1;
  if (int rc = flow_lds(md, fp.nblk, NB, &lds)) return rc;
  const int rows = 256 / LPR;
  const int nb = (md.N + rows - 1) / rows;
  static size_t cur[10] = {48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024};
#define ELLF_LAUNCH(lpr, nbv, slot)                                                                                         \
  do {                                                                                                                      \
    if (int rc = ensure_lds(reinterpret_cast<const void*>(k_ell_flow<lpr, nbv>), lds, &cur[slot])) return rc;               \
    hipLaunchKernelGGL((k_ell_flow<lpr, nbv>), dim3(nb), dim3(256), lds, st, md, fp, Y, mu, v, rowp, ws, g_mu, g_v, g_rowp); \
  } while (0)
  if (LPR == 4) {
    if (NB == 8) ELLF_LAUNCH(4, 8, 0);
    else if (NB == 4) ELLF_LAUNCH(4, 4, 1);
    else if (NB == 2) ELLF_LAUNCH(4, 2, 2);
    else ELLF_LAUNCH(4, 1, 3);
  } else if (LPR == 16) {
    if (NB == 4) ELLF_LAUNCH(16, 4, 4);
    else if (NB == 2) ELLF_LAUNCH(16, 2, 5);
    else ELLF_LAUNCH(16, 1, 6);
  } else {
    if (NB == 4) ELLF_LAUNCH(32, 4, 7);
    else if (NB == 2) ELLF_LAUNCH(32, 2, 8);
    else ELLF_LAUNCH(32, 1, 9);
  }
#undef ELLF_LAUNCH
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_parts, dim3((2 + md.P + 31) / 32), dim3(256), 0, st, ws, nb, 2 + md.P, out, g_theta, 2);
  LAUNCH_CHECK();
  return 0;
}

int launch_flow_eval(const tgp_model& md, const FlowProg& fp, const double* f, int S, int N, const double* rowp, double* G, double* dG,
                     double* logdG, hipStream_t st, double* sum_out, double* ws) {
  const size_t total = (size_t)S * N;
  const size_t lds = 3 * (size_t)(md.P + 2) * sizeof(double);
  const unsigned nb = (unsigned)((total + 1023) / 1024);
  hipLaunchKernelGGL(k_flow_eval, dim3(nb), dim3(256), lds, st, md, fp, f, total, N, rowp, G, dG, logdG,
                     sum_out != nullptr ? ws : (double*)nullptr);
  LAUNCH_CHECK();
  if (sum_out != nullptr) {
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(256), 0, st, ws, (int)nb, 1, sum_out, (double*)nullptr, 1);
    LAUNCH_CHECK();
  }
  return 0;
}

int launch_predict(const tgp_model& md, const FlowProg& fp, const double* mu, const double* v, const double* rowp, const double* Y,
                   double Y_std, double* m1, double* m2, double* logp, hipStream_t st) {
  const size_t lds = 2 * (size_t)(md.P + 2) * sizeof(double);
  hipLaunchKernelGGL(k_predict, dim3((md.N + 255) / 256), dim3(256), lds, st, md, fp, mu, v, rowp, Y, Y_std, m1, m2, logp);
  LAUNCH_CHECK();
  return 0;
}

int launch_adam(double* params, const double* grads, double* exp_avg, double* exp_avg_sq, int64_t n, double lr,
                double beta1, double beta2, double eps, double weight_decay, int step, int maximize, hipStream_t st) {
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2s = sqrt(1.0 - pow(beta2, (double)step));
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n,
                     lr, beta1, beta2, eps, weight_decay, bc1, bc2s, maximize ? -1.0 : 1.0);
  LAUNCH_CHECK();
  return 0;
}

int launch_adam_dev(double* params, const double* grads, double* exp_avg, double* exp_avg_sq, int64_t n, double lr,
                    double beta1, double beta2, double eps, double weight_decay, int32_t* step_dev, int maximize,
                    hipStream_t st, int64_t n_plain, int64_t skip_off, int64_t skip_n) {
  const int64_t nb = (n - skip_n + 255) / 256 > 0 ? (n - skip_n + 255) / 256 : 1;
  hipLaunchKernelGGL(k_adam_dev, dim3((unsigned)(nb < 512 ? nb : 512)), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq,
                     n, lr, beta1, beta2, eps, weight_decay, step_dev, maximize ? -1.0 : 1.0, n_plain, log(beta1), log(beta2), skip_off, skip_n);
  LAUNCH_CHECK();
  return 0;
}

}  // namespace tgp
