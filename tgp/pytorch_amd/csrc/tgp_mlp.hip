// tgp_mlp.hip -- the per-row parameter networks of the input-dependent flows:
// Sinh_ArcsinhFlow builds a_n = NN_a(x_n), b_n = NN_b(x_n) from two MLPs per block (models/flow.py:836-897,949-965;
// layers = pytorchlib apply_linear: Linear -> activation -> Dropout, flow.py:853-871), 6 nets of 4 -> 50 -> 50 -> 1
// at BASELINE config C4, with dropout active in training (MC dropout, sparse_MF_SP.py:133-134).
//
// All nets of a flow have one architecture, so they run as ONE launch: grid = (row blocks, nets), 128 rows per
// block, one thread per row.  A net's weights (2 851 doubles at C4) and the block's activations ([H][128] per
// layer: a thread's column is its private, conflict-free strip) live in LDS; weights are read as wave-uniform
// broadcasts, four output units per activation read.
//   forward : out[n][net]
//   backward: recomputes the forward (cheaper than storing N x H x L activations in HBM), back-propagates
//             d out, and forms the weight gradients block-cooperatively as [units x 128 rows] x [128 rows x units]
//             contractions from LDS; per-block partials are summed in a second kernel in a fixed order.
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

#define MLP_T 128      /* rows per block */
#define MLP_ST 129     /* LDS stride of an activation strip [unit][row]: a thread's own column is conflict-free and the
                          block-cooperative (lanes across units) reads of the weight-gradient pass are 2-way at worst */
#define MLP_MAXH 64
#define MLP_MAXL 3

__host__ __device__ inline int mlp_weights_per_net(int D, int H, int L) {
  return D * H + H + (L - 1) * (H * H + H) + H + 1;
}

// Dropout keep test.  One splitmix64 finaliser per (seed, step, net, layer, row, unit / 4) yields four 16-bit lanes,
// one per unit of the group: keep iff lane >= round(p * 65536)  (p is quantised to 1/65536; 0.25 is exact).
__device__ __forceinline__ uint64_t mlp_hash4(uint64_t seed, int step, int net, int layer, int row, int ugroup) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(unsigned)step;
  z ^= ((uint64_t)(unsigned)net << 56) ^ ((uint64_t)(unsigned)layer << 48) ^ ((uint64_t)(unsigned)ugroup << 32) ^ (uint64_t)(unsigned)row;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned mlp_thresh(double p) { return (unsigned)(p * 65536.0 + 0.5); }
__device__ __forceinline__ bool mlp_keep(uint64_t seed, int step, int net, int layer, int row, int unit, double p) {
  const uint64_t h = mlp_hash4(seed, step, net, layer, row, unit >> 2);
  return (unsigned)((h >> (16 * (unit & 3))) & 0xFFFFu) >= mlp_thresh(p);
}

__device__ __forceinline__ double mlp_act(int act, double z) { return act == 0 ? fmax(z, 0.0) : tanh(z); }

// one hidden layer for this thread's row: ain [nin][ST] -> aout [nout][ST]  (post-activation, post-dropout)
__device__ __forceinline__ void mlp_layer(const double* __restrict__ W, const double* __restrict__ b, int nin, int nout,
                                          const double* ain, double* aout, int tid, int act, bool drop, double p, double scale,
                                          uint64_t seed, int step, int net, int layer, int row) {
  int j = 0;
  for (; j + 4 <= nout; j += 4) {
    double s0 = b[j], s1 = b[j + 1], s2 = b[j + 2], s3 = b[j + 3];
    const double* w0 = W + (size_t)j * nin;
    int i = 0;
    // 4 x 4 register block: 4 activation reads + 16 broadcast weight reads in flight per 16 FMAs (the loop is LDS-latency
    // bound at two waves per CU unless several reads are outstanding)
    for (; i + 4 <= nin; i += 4) {
      double a[4], wv[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = ain[(i + u) * MLP_ST + tid];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) wv[r][u] = w0[r * nin + i + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s0 = fma(wv[0][u], a[u], s0);
        s1 = fma(wv[1][u], a[u], s1);
        s2 = fma(wv[2][u], a[u], s2);
        s3 = fma(wv[3][u], a[u], s3);
      }
    }
    for (; i < nin; ++i) {
      const double a = ain[i * MLP_ST + tid];
      s0 = fma(w0[i], a, s0);
      s1 = fma(w0[nin + i], a, s1);
      s2 = fma(w0[2 * nin + i], a, s2);
      s3 = fma(w0[3 * nin + i], a, s3);
    }
    double o[4] = {s0, s1, s2, s3};
    const uint64_t h4 = drop ? mlp_hash4(seed, step, net, layer, row, j >> 2) : 0;
    const unsigned th = mlp_thresh(p);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      double a = mlp_act(act, o[u]);
      if (drop) a = ((unsigned)((h4 >> (16 * u)) & 0xFFFFu) >= th) ? a * scale : 0.0;
      aout[(j + u) * MLP_ST + tid] = a;
    }
  }
  for (; j < nout; ++j) {
    double s = b[j];
    for (int i = 0; i < nin; ++i) s = fma(W[(size_t)j * nin + i], ain[i * MLP_ST + tid], s);
    double a = mlp_act(act, s);
    if (drop) a = mlp_keep(seed, step, net, layer, row, j, p) ? a * scale : 0.0;
    aout[j * MLP_ST + tid] = a;
  }
}

// global -> LDS copy of one net's weights with 8 loads in flight per thread (a plain loop pays one memory round
// trip per element: 22 of them at C4)
__device__ __forceinline__ void mlp_stage_weights(const double* __restrict__ src, int n, double* dst, int tid) {
  for (int base = 0; base < n; base += 8 * MLP_T) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * MLP_T + tid;
      v[u] = i < n ? src[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * MLP_T + tid;
      if (i < n) dst[i] = v[u];
    }
  }
}

struct MlpArgs {
  int N, D, H, L, nnets, act, training;
  double p;
  uint64_t seed;
  const double* X;
  const double* W;
  const int32_t* step_dev;  // may be nullptr (step 0)
};

// LDS: weights (PW) | a0 [D][T] | a1 [H][T] | ... | aL [H][T]
__global__ __launch_bounds__(MLP_T) void k_mlp_fwd(MlpArgs m, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, net = blockIdx.y, D = m.D, H = m.H, L = m.L;
  const int PW = mlp_weights_per_net(D, H, L);
  double* Wl = sm;
  double* a0 = Wl + ((PW + 1) & ~1);
  double* a1 = a0 + D * MLP_ST;
  double* a2 = a1 + H * MLP_ST;
  mlp_stage_weights(m.W + (size_t)net * PW, PW, Wl, tid);
  const int row = blockIdx.x * MLP_T + tid, rc = row < m.N ? row : m.N - 1;
  for (int d = 0; d < D; ++d) a0[d * MLP_ST + tid] = m.X[(size_t)rc * D + d];
  __syncthreads();
  const int step = m.step_dev ? m.step_dev[0] : 0;
  const bool drop = m.training && m.p > 0.0;
  const double scale = drop ? 1.0 / (1.0 - m.p) : 1.0;
  const double* w = Wl;
  const double* ain = a0;
  int nin = D;
  for (int l = 0; l < L; ++l) {
    double* aout = (l & 1) ? a2 : a1;
    mlp_layer(w, w + (size_t)H * nin, nin, H, ain, aout, tid, m.act, drop, m.p, scale, m.seed, step, net, l, row);
    w += (size_t)H * nin + H;
    ain = aout;
    nin = H;
  }
  double s = w[H];
  for (int i = 0; i < H; ++i) s = fma(w[i], ain[i * MLP_ST + tid], s);
  if (row < m.N) out[(size_t)row * m.nnets + net] = s;
}

// backward; partial weight gradients of this (row block, net) into part[(blockIdx.x * nnets + net) * PW ...]
// LDS: weights | a0 [D][ST] | a1 .. aL [H][ST] each.  The delta of a layer overwrites that layer's activations in place
// (it depends on the thread's own element only, once the weight gradients that read the activations are done).
__global__ __launch_bounds__(MLP_T) void k_mlp_bwd(MlpArgs m, const double* __restrict__ g_out, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, net = blockIdx.y, D = m.D, H = m.H, L = m.L;
  const int PW = mlp_weights_per_net(D, H, L);
  double* Wl = sm;
  double* act0 = Wl + ((PW + 1) & ~1);
  double* actl = act0 + D * MLP_ST;
  double* gos = actl + (size_t)L * H * MLP_ST;  // d out of the block's rows [T]
  mlp_stage_weights(m.W + (size_t)net * PW, PW, Wl, tid);
  const int row = blockIdx.x * MLP_T + tid;
  const bool valid = row < m.N;
  const int rc = valid ? row : m.N - 1;
  for (int d = 0; d < D; ++d) act0[d * MLP_ST + tid] = m.X[(size_t)rc * D + d];
  const double go = valid ? g_out[(size_t)row * m.nnets + net] : 0.0;
  gos[tid] = go;
  __syncthreads();
  const int step = m.step_dev ? m.step_dev[0] : 0;
  const bool drop = m.training && m.p > 0.0;
  const double scale = drop ? 1.0 / (1.0 - m.p) : 1.0;
  int woff[MLP_MAXL + 1];
  {
    int o = 0, nin = D;
    for (int l = 0; l < L; ++l) { woff[l] = o; o += H * nin + H; nin = H; }
    woff[L] = o;
  }
  // ---- forward recomputation, every layer's output kept ----
  {
    const double* ain = act0;
    int nin = D;
    for (int l = 0; l < L; ++l) {
      double* aout = actl + (size_t)l * H * MLP_ST;
      const double* w = Wl + woff[l];
      mlp_layer(w, w + (size_t)H * nin, nin, H, ain, aout, tid, m.act, drop, m.p, scale, m.seed, step, net, l, row);
      ain = aout;
      nin = H;
    }
  }
  __syncthreads();
  double* gp = part + ((size_t)blockIdx.x * m.nnets + net) * PW;
  // derivative of (activation -> dropout) through the stored value: relu' from its sign; tanh' needs the kept flag
  auto dfac = [&](double a, int layer, int unit) {
    if (m.act == 0) return a > 0.0 ? scale : 0.0;
    const bool kept = !drop || mlp_keep(m.seed, step, net, layer, row, unit, m.p);
    const double t = a / scale;
    return kept ? scale * (1.0 - t * t) : 0.0;
  };
  // ---- output layer: out = Wo . aL + bo :  dWo[i] = sum_rows go * aL[i][row], dbo = sum_rows go ----
  double* aL = actl + (size_t)(L - 1) * H * MLP_ST;
  for (int i = tid; i <= H; i += MLP_T) {
    double s0 = 0.0, s1 = 0.0;
    for (int r = 0; r < MLP_T; r += 2) {
      s0 = fma(gos[r], i < H ? aL[i * MLP_ST + r] : 1.0, s0);
      s1 = fma(gos[r + 1], i < H ? aL[i * MLP_ST + r + 1] : 1.0, s1);
    }
    gp[woff[L] + i] = s0 + s1;
  }
  __syncthreads();
  {
    const double* wo = Wl + woff[L];
    for (int i = 0; i < H; ++i) {
      const double a = aL[i * MLP_ST + tid];
      aL[i * MLP_ST + tid] = wo[i] * go * dfac(a, L - 1, i);   // delta_L in place
    }
  }
  __syncthreads();
  // ---- hidden layers, last to first: the layer's buffer now holds its delta ----
  for (int l = L - 1; l >= 0; --l) {
    const int nin = l == 0 ? D : H;
    const double* dl = actl + (size_t)l * H * MLP_ST;
    double* ain = l == 0 ? act0 : actl + (size_t)(l - 1) * H * MLP_ST;
    const double* w = Wl + woff[l];
    // dW[j][i] = sum_rows delta[j][row] * ain[i][row] ; db[j] = sum_rows delta[j][row]   (block-cooperative)
    for (int e = tid; e < H * (nin + 1); e += MLP_T) {
      const int j = e / (nin + 1), i = e % (nin + 1);
      const double* dj = dl + j * MLP_ST;
      double s0 = 0.0, s1 = 0.0;
      if (i < nin) {
        const double* ai = ain + i * MLP_ST;
        double s2 = 0.0, s3 = 0.0;
        for (int r = 0; r < MLP_T; r += 8) {
          double dv[8], av[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { dv[u] = dj[r + u]; av[u] = ai[r + u]; }
#pragma unroll
          for (int u = 0; u < 8; u += 4) {
            s0 = fma(dv[u], av[u], s0);
            s1 = fma(dv[u + 1], av[u + 1], s1);
            s2 = fma(dv[u + 2], av[u + 2], s2);
            s3 = fma(dv[u + 3], av[u + 3], s3);
          }
        }
        gp[woff[l] + j * nin + i] = (s0 + s1) + (s2 + s3);
      } else {
        for (int r = 0; r < MLP_T; r += 2) { s0 += dj[r]; s1 += dj[r + 1]; }
        gp[woff[l] + H * nin + j] = s0 + s1;
      }
    }
    if (l == 0) break;
    __syncthreads();
    // delta_{l-1}[i] = (sum_j W[j][i] delta_l[j]) * d(act, dropout)(a_{l-1}[i]) -- own column only, in place
    for (int i0 = 0; i0 < H; i0 += 4) {
      double s[4] = {0, 0, 0, 0};
      const bool full = i0 + 4 <= H;
      int j = 0;
      if (full) {
        for (; j + 4 <= H; j += 4) {
          double dv[4], wv[4][4];
#pragma unroll
          for (int t = 0; t < 4; ++t) dv[t] = dl[(j + t) * MLP_ST + tid];
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) wv[t][u] = w[(size_t)(j + t) * H + i0 + u];
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] = fma(wv[t][u], dv[t], s[u]);
        }
      }
      for (; j < H; ++j) {
        const double dj = dl[j * MLP_ST + tid];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (i0 + u < H) s[u] = fma(w[(size_t)j * H + i0 + u], dj, s[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u;
        if (i < H) {
          const double a = ain[i * MLP_ST + tid];
          ain[i * MLP_ST + tid] = s[u] * dfac(a, l - 1, i);
        }
      }
    }
    __syncthreads();
  }
}

// g_W[net][k] = sum over row blocks of the partials (fixed order)
__global__ __launch_bounds__(256) void k_mlp_reduce(const double* __restrict__ part, int nblk, size_t len, double* __restrict__ g_W) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= len) return;
  double s0 = 0.0, s1 = 0.0;
  int b = 0;
  for (; b + 2 <= nblk; b += 2) {
    s0 += part[(size_t)b * len + e];
    s1 += part[(size_t)(b + 1) * len + e];
  }
  if (b < nblk) s0 += part[(size_t)b * len + e];
  g_W[e] = s0 + s1;
}

static size_t mlp_lds_bytes(int D, int H, int L, bool bwd) {
  const size_t pw = (size_t)((mlp_weights_per_net(D, H, L) + 1) & ~1);
  const size_t acts = bwd ? (size_t)D + (size_t)L * H : (size_t)D + 2 * (size_t)H;
  return (pw + acts * MLP_ST + (bwd ? MLP_T : 0)) * sizeof(double);
}

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
  hipLaunchKernelGGL(k_mlp_fwd, dim3((d.N + MLP_T - 1) / MLP_T, d.nnets), dim3(MLP_T), lds, st, mlp_args(d, X, W, step_dev), out);
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
  hipLaunchKernelGGL(k_mlp_bwd, dim3(nblk, d.nnets), dim3(MLP_T), lds, st, mlp_args(d, X, W, step_dev), g_out, ws);
  LAUNCH_CHECK();
  const size_t len = (size_t)d.nnets * mlp_weights_per_net(d.D, d.H, d.L);
  hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, ws, nblk, len, g_W);
  LAUNCH_CHECK();
  return 0;
}

}  // namespace tgp
