// tgp_mlp.hip -- the per-row parameter networks of the input-dependent flows:
// Sinh_ArcsinhFlow builds a_n = NN_a(x_n), b_n = NN_b(x_n) from two MLPs per block (models/flow.py:836-897,949-965;
// layers = pytorchlib apply_linear: Linear -> activation -> Dropout, flow.py:853-871), 6 nets of 4 -> 50 -> 50 -> 1
// at BASELINE config C4, with dropout active in training (MC dropout, sparse_MF_SP.py:133-134).
//
// All nets of a flow have one architecture, so they run as ONE launch: grid = (row blocks, nets), 128 rows per
// block = 4 waves x 32 rows.  A net's weights (zero-padded to 64 units x pad4(inputs)) and the block's activation
// strips [unit][row] live in LDS; every layer product runs on v_mfma_f64_16x16x4_f64 with the weight tile as the A
// operand and four 16-row groups of the strip as B operands (the strip layout is k-major, conflict-free).
//   forward : out[n][net]
//   backward: recomputes the forward (cheaper than storing N x H x L activations in HBM), back-propagates d out with
//             W^T tiles as A operands, and forms the weight gradients as [units x 128 rows] x [128 rows x units]
//             MFMA contractions from the strips; per-block partials are summed in a second kernel in a fixed order.
// Dropout is a counter-based hash of (seed, step, net, layer, row, unit): the same mask in the forward, in the
// backward recomputation and under hipGraph replay (step is read from device memory), different every step.
#include "tgp_dev.hpp"
#include "tgp_launch.hpp"

namespace tgp {

#define LAUNCH_CHECK()                                              \
  do {                                                              \
    hipError_t e_ = hipGetLastError();                              \
    if (e_ != hipSuccess) return set_error(e_, __FILE__, __LINE__); \
  } while (0)

#define MLP_T 64       /* rows per block */
#define MLP_NT 256     /* threads per block = 4 waves x 16 rows; the LDS image (81 KB at C4) lets TWO blocks share a CU */
#define MLP_RT 1      /* 16-row MFMA column tiles per wave */
#define MLP_ST 65      /* LDS stride of an activation strip [unit][row] (k-major for the layer products) */
#define MLP_HP 64      /* units padded to four 16-wide MFMA tiles */
#define MLP_MAXH 64
#define MLP_MAXL 3

__host__ __device__ inline int mlp_weights_per_net(int D, int H, int L) {
  return D * H + H + (L - 1) * (H * H + H) + H + 1;
}
__host__ __device__ inline int mlp_pad4(int x) { return (x + 3) & ~3; }

// Dropout keep test.  One splitmix64 finaliser per (seed, step, net, layer, row, group) yields four 16-bit lanes, one
// per unit of the group; unit j belongs to group (j >> 4) * 4 + (j & 3), lane (j >> 2) & 3 -- the four accumulator
// registers of one MFMA lane.  keep iff lane >= round(p * 65536)  (p quantised to 1/65536; 0.25 is exact).
__device__ __forceinline__ uint64_t mlp_hash4(uint64_t seed, int step, int net, int layer, int row, int group) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(unsigned)step;
  z ^= ((uint64_t)(unsigned)net << 56) ^ ((uint64_t)(unsigned)layer << 48) ^ ((uint64_t)(unsigned)group << 32) ^ (uint64_t)(unsigned)row;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned mlp_thresh(double p) { return (unsigned)(p * 65536.0 + 0.5); }

__device__ __forceinline__ double mlp_act(int act, double z) { return act == 0 ? fmax(z, 0.0) : tanh(z); }

// global -> LDS copy of one hidden layer's weights into the zero-padded [HP][KP] image (+ bias [HP])
// global -> LDS copy of one hidden layer's weights into the zero-padded [HR][KP] image (+ bias [HR]).
// (the image has HR = pad4(H) rows, not 64: the MFMA tiles read rows >= HR as zeros through a predicate -- 5 KB
//  that decide whether two blocks fit a CU.  Requesting all arrays before the first store was tried: the copy is bound
//  by its index arithmetic, not by the round trips, and it got slower.)
__device__ __forceinline__ void mlp_stage_layer(const double* __restrict__ src, int H, int nin, int KP, double* Wp, double* bp,
                                                int tid) {
  const int HR = mlp_pad4(H);
  // (row, column) of element e = tid, tid + NT, ...: one division per thread, then increments (an e / KP, e % KP per
  // element with a runtime KP was most of the copy's instructions)
  const int dj = MLP_NT / KP, di = MLP_NT % KP;
  int j = tid / KP, i = tid % KP;
  for (int base = 0; base < HR * KP; base += 8 * MLP_NT) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = base + u * MLP_NT + tid;
      v[u] = (e < HR * KP && j < H && i < nin) ? src[j * nin + i] : 0.0;
      j += dj; i += di;
      if (i >= KP) { i -= KP; ++j; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = base + u * MLP_NT + tid;
      if (e < HR * KP) Wp[e] = v[u];
    }
  }
  if (tid < HR) bp[tid] = tid < H ? src[H * nin + tid] : 0.0;
}

#ifdef TGP_STAMPS
#define MSTAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) mst[i] = (double)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif
struct MlpArgs {
  int N, D, H, L, nnets, act, training;
  double p;
  uint64_t seed;
  const double* X;
  const double* W;
  const int32_t* step_dev;  // may be nullptr (step 0)
};

// LDS image shared by both kernels:
//   per hidden layer l: Wp_l [KPH][KP_l] (zero padded), b_l [KPH] ; output layer wo [KPH], bo [2]
//   a0 [KP_0][ST] ; a_1 .. a_L [KPH][ST]  (KP_0 = pad4(D), KPH = pad4(H); padded units are exact zeros)
// (offsets are closed-form: an array indexed by the layer would live in scratch memory)
struct MlpLds {
  int KP0, KPH, L, wo, act0, actl, gos, total;
  __host__ __device__ int wp(int l) const { return l == 0 ? 0 : KPH * KP0 + KPH + (l - 1) * (KPH * KPH + KPH); }
  __host__ __device__ int bp(int l) const { return wp(l) + KPH * (l == 0 ? KP0 : KPH); }
};
__host__ __device__ inline MlpLds mlp_lds(int D, int H, int L, bool bwd) {
  MlpLds o;
  o.KP0 = mlp_pad4(D); o.KPH = mlp_pad4(H); o.L = L;
  int p = o.bp(L - 1) + o.KPH;
  o.wo = p; p += o.KPH + 2;
  o.act0 = p; p += o.KP0 * MLP_ST;
  o.actl = p; p += (bwd ? L : (L > 1 ? 2 : 1)) * o.KPH * MLP_ST;
  p = (p + 1) & ~1;
  o.gos = p; p += bwd ? MLP_T : 0;
  o.total = p;
  return o;
}
// offset of layer l inside the packed weight vector of a net (l == L: the output layer)
__host__ __device__ inline int mlp_woff(int D, int H, int l) { return l == 0 ? 0 : D * H + H + (l - 1) * (H * H + H); }

// (Both kernels declare two waves per SIMD: with the default 512-register budget hipcc splits the file into VGPRs and
// AGPRs and shuttles all 32 accumulator registers through v_accvgpr_write/read around every k step.)
//
// One hidden layer on the matrix cores, this wave's 64 rows:  out[j][r] = act(b_j + sum_i W[j][i] in[i][r]) (+ dropout).
//   A operand = W tile (lane: unit j = l&15, k = i = l>>4), B operand = activation strip rows (k-major), C/D: lane holds
//   units q, q+4, q+8, q+12 of row r = l&15 -- exactly one dropout hash group.
__device__ __forceinline__ void mlp_layer_mfma(const double* Wp, const double* bp, int KP, const double* ain, double* aout,
                                               int KPout, int lane, int wave, int act, bool drop, double p, double scale,
                                               uint64_t seed, int step, int net, int layer, int row0) {
  const int n = lane & 15, q = lane >> 4;
  const unsigned th = mlp_thresh(p);
  for (int jt = 0; jt * 16 < KPout; ++jt) {
    d4 acc[MLP_RT];
#pragma unroll
    for (int rt = 0; rt < MLP_RT; ++rt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) acc[rt][rr] = 16 * jt + q + 4 * rr < KPout ? bp[16 * jt + q + 4 * rr] : 0.0;
    // operands of eight k-steps are requested before their MFMAs (tile_mm_f): a one-deep prefetch made every k-step
    // wait for an LDS round trip (about 150 cycles per MFMA instead of 64)
    const bool vj = 16 * jt + n < KPout;   // rows of the weight image beyond pad4(H) do not exist: zeros
    const double* wrow = Wp + (vj ? 16 * jt + n : 0) * KP + q;
#pragma unroll
    for (int rt = 0; rt < MLP_RT; ++rt) {
      const double* brow = ain + q * MLP_ST + 16 * MLP_RT * wave + 16 * rt + n;
      acc[rt] = tile_mm_f<8>([&](int k) { return vj ? wrow[k] : 0.0; }, [&](int k) { return brow[k * MLP_ST]; }, 0, KP, acc[rt]);
    }
#pragma unroll
    for (int rt = 0; rt < MLP_RT; ++rt) {
      const int rl = 16 * MLP_RT * wave + 16 * rt + n;
      const uint64_t h4 = drop ? mlp_hash4(seed, step, net, layer, row0 + rl, 4 * jt + q) : 0;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int j = 16 * jt + q + 4 * rr;
        double v = mlp_act(act, acc[rt][rr]);
        if (drop) v = ((unsigned)((h4 >> (16 * rr)) & 0xFFFFu) >= th) ? v * scale : 0.0;
        if (j < KPout) aout[j * MLP_ST + rl] = v;
      }
    }
  }
}

__device__ __forceinline__ void mlp_stage_all(const MlpArgs& m, const MlpLds& Lo, double* sm, int net, int tid) {
  const int PW = mlp_weights_per_net(m.D, m.H, m.L);
  const double* src = m.W + (size_t)net * PW;
  int nin = m.D;
  for (int l = 0; l < m.L; ++l) {
    mlp_stage_layer(src, m.H, nin, l == 0 ? Lo.KP0 : Lo.KPH, sm + Lo.wp(l), sm + Lo.bp(l), tid);
    src += m.H * nin + m.H;
    nin = m.H;
  }
  if (tid < Lo.KPH) sm[Lo.wo + tid] = tid < m.H ? src[tid] : 0.0;
  if (tid == 0) sm[Lo.wo + Lo.KPH] = src[m.H];
  // inputs: a0 [KP0][ST], this block's rows (padding rows repeat the last row; their d out is zero)
  if (tid < MLP_T) {
    const int row = blockIdx.x * MLP_T + tid, rc = row < m.N ? row : m.N - 1;
    for (int d = 0; d < Lo.KP0; ++d) sm[Lo.act0 + d * MLP_ST + tid] = d < m.D ? m.X[(size_t)rc * m.D + d] : 0.0;
  }
}

__global__ __launch_bounds__(MLP_NT, 2) void k_mlp_fwd(MlpArgs m, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, net = blockIdx.y, H = m.H, L = m.L;
  const MlpLds Lo = mlp_lds(m.D, H, L, false);
  mlp_stage_all(m, Lo, sm, net, tid);
  __syncthreads();
  const int step = m.step_dev ? m.step_dev[0] : 0;
  const bool drop = m.training && m.p > 0.0;
  const double scale = drop ? 1.0 / (1.0 - m.p) : 1.0;
  const double* ain = sm + Lo.act0;
  for (int l = 0; l < L; ++l) {
    double* aout = sm + Lo.actl + (l & 1) * Lo.KPH * MLP_ST;
    mlp_layer_mfma(sm + Lo.wp(l), sm + Lo.bp(l), l == 0 ? Lo.KP0 : Lo.KPH, ain, aout, Lo.KPH, lane, wave, m.act, drop, m.p, scale,
                   m.seed, step, net, l, blockIdx.x * MLP_T);
    ain = aout;
    __builtin_amdgcn_wave_barrier();  // a wave reads back only its own 32 columns
  }
  __syncthreads();  // row r's strip column was written by wave r / 32, the output dot runs on threads 0..127
  if (tid < MLP_T) {
    const int row = blockIdx.x * MLP_T + tid;
    const double* wo = sm + Lo.wo;
    double s0 = wo[Lo.KPH], s1 = 0.0;
    for (int i = 0; i + 2 <= Lo.KPH; i += 2) {
      s0 = fma(wo[i], ain[i * MLP_ST + tid], s0);
      s1 = fma(wo[i + 1], ain[(i + 1) * MLP_ST + tid], s1);
    }
    if (row < m.N) out[(size_t)row * m.nnets + net] = s0 + s1;
  }
}

// backward; partial weight gradients of this (row block, net) into part[(blockIdx.x * nnets + net) * PW ...].
// The delta of a layer overwrites that layer's activation strip in place (own element only).
__global__ __launch_bounds__(MLP_NT, 2) void k_mlp_bwd(MlpArgs m, const double* __restrict__ g_out, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
  const int net = blockIdx.y, D = m.D, H = m.H, L = m.L;
  const int PW = mlp_weights_per_net(D, H, L);
  const MlpLds Lo = mlp_lds(D, H, L, true);
  const int KPH = Lo.KPH;
#ifdef TGP_STAMPS
  double* mst = part + (size_t)gridDim.x * m.nnets * PW;  // the 16 spare doubles behind the partials
#endif
  MSTAMP(0);
  mlp_stage_all(m, Lo, sm, net, tid);
  const int row = blockIdx.x * MLP_T + tid;
  double* gos = sm + Lo.gos;
  if (tid < MLP_T) gos[tid] = row < m.N ? g_out[(size_t)row * m.nnets + net] : 0.0;
  __syncthreads();
  MSTAMP(1);
  const int step = m.step_dev ? m.step_dev[0] : 0;
  const bool drop = m.training && m.p > 0.0;
  const double scale = drop ? 1.0 / (1.0 - m.p) : 1.0;
  const unsigned th = mlp_thresh(m.p);
  // ---- forward recomputation, every layer's output kept ----
  {
    const double* ain = sm + Lo.act0;
    for (int l = 0; l < L; ++l) {
      double* aout = sm + Lo.actl + (size_t)l * KPH * MLP_ST;
      mlp_layer_mfma(sm + Lo.wp(l), sm + Lo.bp(l), l == 0 ? Lo.KP0 : KPH, ain, aout, KPH, lane, wave, m.act, drop, m.p, scale,
                     m.seed, step, net, l, blockIdx.x * MLP_T);
      ain = aout;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  MSTAMP(2);
  double* gp = part + ((size_t)blockIdx.x * m.nnets + net) * PW;
  // ---- output layer: out = wo . aL + bo :  dwo[i] = sum_rows go * aL[i][row], dbo = sum_rows go ----
  double* aL = sm + Lo.actl + (size_t)(L - 1) * KPH * MLP_ST;
  for (int i = tid; i <= H; i += MLP_NT) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 2
    for (int r = 0; r < MLP_T; r += 4) {
      s0 = fma(gos[r], i < H ? aL[i * MLP_ST + r] : 1.0, s0);
      s1 = fma(gos[r + 1], i < H ? aL[i * MLP_ST + r + 1] : 1.0, s1);
      s2 = fma(gos[r + 2], i < H ? aL[i * MLP_ST + r + 2] : 1.0, s2);
      s3 = fma(gos[r + 3], i < H ? aL[i * MLP_ST + r + 3] : 1.0, s3);
    }
    gp[mlp_woff(D, H, L) + i] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  MSTAMP(3);
  // derivative of (activation -> dropout) through the stored value a; `kept` only matters for tanh
  auto dfac = [&](double a, bool kept) {
    if (m.act == 0) return a > 0.0 ? scale : 0.0;
    const double t = a / scale;
    return kept ? scale * (1.0 - t * t) : 0.0;
  };
  auto kept_flag = [&](int layer, int rl, int j) {
    if (!drop || m.act == 0) return true;
    const uint64_t h4 = mlp_hash4(m.seed, step, net, layer, blockIdx.x * MLP_T + rl, (j >> 4) * 4 + (j & 3));
    return (unsigned)((h4 >> (16 * ((j >> 2) & 3))) & 0xFFFFu) >= th;
  };
  {
    // delta_L in place (zero on padded units): 256 threads over the MLP_T x KPH strip, MLP_NT / MLP_T ranges of units
    const double* wo = sm + Lo.wo;
    constexpr int parts = MLP_NT / MLP_T;
    const int rl = tid % MLP_T, half = tid / MLP_T, hk = (KPH + parts - 1) / parts;
    const double go = gos[rl];
    for (int i = half * hk; i < min(KPH, (half + 1) * hk); ++i) {
      const double a = aL[i * MLP_ST + rl];
      aL[i * MLP_ST + rl] = wo[i] * go * dfac(a, kept_flag(L - 1, rl, i));
    }
  }
  __syncthreads();
  MSTAMP(4);
  // ---- hidden layers, last to first: the layer's strip now holds its delta ----
  for (int l = L - 1; l >= 0; --l) {
    const int nin = l == 0 ? D : H, KPin = l == 0 ? Lo.KP0 : KPH;
    const double* dl = sm + Lo.actl + (size_t)l * KPH * MLP_ST;
    double* ain = l == 0 ? sm + Lo.act0 : sm + Lo.actl + (size_t)(l - 1) * KPH * MLP_ST;
    // dW[j][i] = sum_rows delta[j][row] ain[i][row]: 16 x 16 tiles over the block's 128 rows, tiles dealt to the two waves
    {
      const int njt = (H + 15) / 16, nit = (nin + 15) / 16;
      for (int t = wave; t < njt * nit; t += MLP_NT / 64) {
        const int jt = t / nit, it = t % nit;
        const int ja = 16 * jt + n, ib = 16 * it + n;
        const bool va = ja < KPH, vb = ib < KPin;
        const double* pa = dl + (va ? ja : 0) * MLP_ST + q;
        const double* pb = ain + (vb ? ib : 0) * MLP_ST + q;
        d4 acc = {0, 0, 0, 0};
        acc = tile_mm_f<8>([&](int k) { return va ? pa[k] : 0.0; }, [&](int k) { return vb ? pb[k] : 0.0; }, 0, MLP_T, acc);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int j = 16 * jt + q + 4 * rr, i = 16 * it + n;
          if (j < H && i < nin) gp[mlp_woff(D, H, l) + j * nin + i] = acc[rr];
        }
      }
      // db[j] = sum_rows delta[j][row]
      for (int j = tid; j < H; j += MLP_NT) {
        const double* dj = dl + j * MLP_ST;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 2
        for (int r = 0; r < MLP_T; r += 4) { s0 += dj[r]; s1 += dj[r + 1]; s2 += dj[r + 2]; s3 += dj[r + 3]; }
        gp[mlp_woff(D, H, l) + H * nin + j] = (s0 + s1) + (s2 + s3);
      }
    }
    MSTAMP(5 + 2 * (L - 1 - l));
    if (l == 0) break;
    __syncthreads();
    // delta_{l-1}[i][r] = (sum_j W_l[j][i] delta_l[j][r]) * d(act, dropout)(a_{l-1}[i][r]), in place over a_{l-1}.
    //   A operand = W^T tile (lane: i = l&15, k = j = l>>4), B operand = delta strip rows, this wave's 64 rows.
    {
      const double* Wp = sm + Lo.wp(l);
      for (int it = 0; it * 16 < KPH; ++it) {
        d4 acc[MLP_RT];
#pragma unroll
        for (int rt = 0; rt < MLP_RT; ++rt) acc[rt] = {0, 0, 0, 0};
        const bool vi = 16 * it + n < KPH;
        const double* wcol = Wp + q * KPH + (vi ? 16 * it + n : 0);
#pragma unroll
        for (int rt = 0; rt < MLP_RT; ++rt) {
          const double* brow = dl + q * MLP_ST + 16 * MLP_RT * wave + 16 * rt + n;
          acc[rt] = tile_mm_f<8>([&](int k) { return vi ? wcol[k * KPH] : 0.0; }, [&](int k) { return brow[k * MLP_ST]; }, 0, KPH,
                                 acc[rt]);
        }
#pragma unroll
        for (int rt = 0; rt < MLP_RT; ++rt) {
          const int rl = 16 * MLP_RT * wave + 16 * rt + n;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int i = 16 * it + q + 4 * rr;
            if (i < KPH) {
              const double a = ain[i * MLP_ST + rl];
              ain[i * MLP_ST + rl] = acc[rt][rr] * dfac(a, kept_flag(l - 1, rl, i));
            }
          }
        }
      }
    }
    __syncthreads();
    MSTAMP(6 + 2 * (L - 1 - l));
  }
}

// g_W[net][k] = sum over row blocks of the partials (fixed order).  One workgroup = 64 consecutive elements x 4 quarters
// of the row blocks (a wave per quarter); 16 partials are requested before the first add -- with two loads in flight per
// thread the 135 partials of a Power-sized step were 67 dependent L2 round trips (21 us on the critical path of the
// ID_TGP step); the four quarter sums meet in LDS and are added in a fixed order.
__global__ __launch_bounds__(256) void k_mlp_reduce(const double* __restrict__ part, int nblk, size_t len, double* __restrict__ g_W) {
  __shared__ double q4[4][64];
  const int el = threadIdx.x & 63, qt = threadIdx.x >> 6;
  const size_t e = (size_t)blockIdx.x * 64 + el;
  const int b0 = (int)((long long)nblk * qt / 4), b1 = (int)((long long)nblk * (qt + 1) / 4);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (e < len) {
    const double* pe = part + e;
    for (int b = b0; b < b1; b += 16) {
      double t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int bu = b + u < b1 ? b + u : b1 - 1;
        t[u] = pe[(size_t)bu * len];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = b + u < b1 ? t[u] : 0.0;
      s0 += (t[0] + t[4]) + (t[8] + t[12]); s1 += (t[1] + t[5]) + (t[9] + t[13]);
      s2 += (t[2] + t[6]) + (t[10] + t[14]); s3 += (t[3] + t[7]) + (t[11] + t[15]);
    }
  }
  q4[qt][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (qt == 0 && e < len) g_W[e] = (q4[0][el] + q4[1][el]) + (q4[2][el] + q4[3][el]);
}

static size_t mlp_lds_bytes(int D, int H, int L, bool bwd) { return (size_t)mlp_lds(D, H, L, bwd).total * sizeof(double); }

size_t mlp_workspace_doubles(int N, int D, int H, int L, int nnets) {
  const size_t nblk = (size_t)(N + MLP_T - 1) / MLP_T;
  return nblk * nnets * (size_t)mlp_weights_per_net(D, H, L) + 16;
}

static int mlp_check(const tgp_mlp& d) {
  if (d.N < 1 || d.D < 1 || d.D > 64 || d.H < 1 || d.H > MLP_MAXH || d.L < 1 || d.L > MLP_MAXL || d.nnets < 1) return -1;
  if (d.act != 0 && d.act != 1) return -1;
  if (!(d.drop_p >= 0.0 && d.drop_p < 1.0)) return -1;
  return 0;
}

static MlpArgs mlp_args(const tgp_mlp& d, const double* X, const double* W, const int32_t* step_dev) {
  MlpArgs a;
  a.N = d.N; a.D = d.D; a.H = d.H; a.L = d.L; a.nnets = d.nnets; a.act = d.act; a.training = d.training;
  a.p = d.drop_p; a.seed = d.seed; a.X = X; a.W = W; a.step_dev = step_dev;
  return a;
}

int launch_mlp_forward(const tgp_mlp& d, const double* X, const double* W, const int32_t* step_dev, double* out,
                       hipStream_t st) {
  if (int rc = mlp_check(d)) return rc;
  const size_t lds = mlp_lds_bytes(d.D, d.H, d.L, false);
  static size_t lds_cur = 48 * 1024;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(k_mlp_fwd), lds, &lds_cur)) return rc;
  hipLaunchKernelGGL(k_mlp_fwd, dim3((d.N + MLP_T - 1) / MLP_T, d.nnets), dim3(MLP_NT), lds, st, mlp_args(d, X, W, step_dev), out);
  LAUNCH_CHECK();
  return 0;
}

int launch_mlp_backward(const tgp_mlp& d, const double* X, const double* W, const int32_t* step_dev, const double* g_out,
                        double* g_W, double* ws, size_t ws_doubles, hipStream_t st) {
  if (int rc = mlp_check(d)) return rc;
  if (ws_doubles < mlp_workspace_doubles(d.N, d.D, d.H, d.L, d.nnets)) return TGP_E_WORKSPACE;
  const size_t lds = mlp_lds_bytes(d.D, d.H, d.L, true);
  static size_t lds_cur = 48 * 1024;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(k_mlp_bwd), lds, &lds_cur)) return rc;
  const int nblk = (d.N + MLP_T - 1) / MLP_T;
  hipLaunchKernelGGL(k_mlp_bwd, dim3(nblk, d.nnets), dim3(MLP_NT), lds, st, mlp_args(d, X, W, step_dev), g_out, ws);
  LAUNCH_CHECK();
  const size_t len = (size_t)d.nnets * mlp_weights_per_net(d.D, d.H, d.L);
  hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)((len + 63) / 64)), dim3(256), 0, st, ws, nblk, len, g_W);
  LAUNCH_CHECK();
  return 0;
}

}  // namespace tgp
