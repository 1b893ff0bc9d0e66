// tgp_mlp.hip -- the per-row parameter networks of the input-dependent flows:
// Sinh_ArcsinhFlow builds a_n = NN_a(x_n), b_n = NN_b(x_n) from two MLPs per block (models/flow.py:836-897,949-965;
// layers = pytorchlib apply_linear: Linear -> activation -> Dropout, flow.py:853-871), 6 nets of 4 -> 50 -> 50 -> 1
// at BASELINE config C4, with dropout active in training (MC dropout, sparse_MF_SP.py:133-134).
//
// All nets of a flow have one architecture, so they run as ONE launch: grid = (row groups, nets); a workgroup = 4 waves
// x 16 rows works through one or more 64-row chunks of one net with that net's weights staged once, as zero-padded
// images in LDS (every fragment address = a per-tile base + a constant); every product runs on v_mfma_f64_16x16x4_f64.
//   chain   : a wave carries ITS 16 rows through all layers in registers.  The accumulator layout of the 16x16x4
//             instruction (lane (n, q) holds units q + 4 rr of row n) IS its B-operand layout of k-step rr, so the
//             four output tiles of a layer are the sixteen k-steps of the next one without leaving the registers:
//             forward, delta_L and the back-propagation delta_{l-1} = (W_l^T delta_l) . act' touch the LDS only for
//             weight fragments.  (The first version kept every activation strip in LDS and read it back per k-step:
//             a block spent three times its MFMA time on LDS round trips and workgroup barriers.)
//   forward : out[n][net] = bo + wo . a_L, the dot product finished across the four lanes of a row.
//   backward: recomputes the forward (cheaper than N x H x L activations through HBM) and forms the weight gradients
//             dW_l = delta_l a_{l-1}^T as [units x 64 rows] x [64 rows x units] contractions: the chain leaves a_l^T,
//             then delta_l^T, in LDS strips [unit][row]; wave w owns unit tile w of delta_l and walks the tiles of
//             a_{l-1}; bias gradients are lane-local sums of the same fragments; the output layer's sums over rows use
//             the DPP row rotations.  A workgroup adds its chunks' gradients in its own slot; the slots are summed in a
//             second kernel in a fixed order.
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

#define MLP_T 64       /* rows per chunk = 4 waves x 16 rows */
#define MLP_NT 256
#define MLP_ST 66      /* row stride of a strip [unit][row]: the transposed fragment reads (lane = unit, q = row) of the
                          contractions fall on 32 distinct 8-byte banks per half wave */
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

template <int ACT>
__device__ __forceinline__ double mlp_act(double z) {
  if constexpr (ACT == 0) return fmax(z, 0.0);
  else return tanh(z);
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

// LDS image (doubles), R = 4 NKS >= H the padded unit count of the kernel variant:
//   per hidden layer l: W_l [R][KP_l] zero padded (KP_0 = pad4(D), KP_l = R), b_l [R] ; output layer wo [R], bo [2] ;
//   x0 [pad4(D)][ST] the block's inputs, transposed ;
//   backward only: dwo [4][H + 1] per-wave sums of the output layer's gradient ; strips S_0 .. S_{L-1} [H][ST]
//   (S_l: a_{l+1}^T for l < L - 1, then delta_{l+1}^T).
// With the padding IN the image every fragment address is a per-tile base plus a constant: the first register-chained
// version read the packed vector through a clamp and two selects per fragment, and the kernels were bound by their VALU
// instructions, not by the matrix pipe.  Lanes of units >= R read past their image (whatever follows it in LDS): a row
// of A only reaches the same row of the product, and those rows are never stored or used as a k index.
struct MlpLds {
  int R, KP0, x0, dwo, strip, sstride, total;
  __host__ __device__ int img(int l) const { return l == 0 ? 0 : R * KP0 + R + (l - 1) * (R * R + R); }
  __host__ __device__ int bias(int l) const { return img(l) + R * (l == 0 ? KP0 : R); }
};
__host__ __device__ inline int mlp_nks(int H) { return H <= 32 ? 8 : H <= 52 ? 13 : 16; }
__host__ __device__ inline MlpLds mlp_lds(int D, int H, int L, bool bwd) {
  MlpLds o;
  o.R = 4 * mlp_nks(H); o.KP0 = mlp_pad4(D);
  int p = o.img(L) + o.R + 2;          // img(L) = the output layer's wo
  o.x0 = p; p += o.KP0 * MLP_ST;
  o.dwo = p; p += bwd ? 4 * (H + 1) : 0;
  p = (p + 1) & ~1;
  o.strip = p; o.sstride = H * MLP_ST;
  p += bwd ? L * o.sstride : 0;
  o.total = p;
  return o;
}
// offset of layer l inside the packed weight vector of a net (l == L: the output layer)
__host__ __device__ inline int mlp_woff(int D, int H, int l) { return l == 0 ? 0 : D * H + H + (l - 1) * (H * H + H); }

// one layer, global -> LDS: image [R][KP] <- src [H][nin] zero padded, bias [R] <- src + H nin (four independent loads in
// flight per thread; the index is clamped, not predicated, so the loads of a trip issue back to back)
template <int R>
__device__ __forceinline__ void mlp_stage_image(const double* __restrict__ src, int H, int nin, int KP, double* img, int tid) {
  const int tot = R * KP;
  for (int base = 0; base < tot; base += 4 * MLP_NT) {
    double v[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = base + u * MLP_NT + tid, j = e / KP, i = e - j * KP;
      ok[u] = j < H && i < nin;
      v[u] = src[ok[u] ? j * nin + i : 0];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = base + u * MLP_NT + tid;
      if (e < tot) img[e] = ok[u] ? v[u] : 0.0;
    }
  }
  if (tid < R) img[tot + tid] = tid < H ? src[H * nin + tid] : 0.0;
}
template <int R, int L>
__device__ __forceinline__ void mlp_stage_weights(const MlpArgs& m, const MlpLds& Lo, double* sm, int net, int tid, int PW) {
  const double* __restrict__ src = m.W + (size_t)net * PW;
  mlp_stage_image<R>(src, m.H, m.D, Lo.KP0, sm, tid);
#pragma unroll
  for (int l = 1; l < L; ++l) mlp_stage_image<R>(src + mlp_woff(m.D, m.H, l), m.H, m.H, R, sm + Lo.img(l), tid);
  const double* wo = src + mlp_woff(m.D, m.H, L);
  if (tid < R) sm[Lo.img(L) + tid] = tid < m.H ? wo[tid] : 0.0;
  if (tid == 0) sm[Lo.img(L) + R] = wo[m.H];
}
// the 64 rows from row0 on, transposed into the x0 strip (rows beyond N repeat the last row; their d out is zero)
__device__ __forceinline__ void mlp_stage_inputs(const MlpArgs& m, double* x0, int row0, int tid) {
  const int KP0 = mlp_pad4(m.D);
  for (int e = tid; e < MLP_T * KP0; e += MLP_NT) {
    const int r = e / KP0, d = e - r * KP0;
    const int row = row0 + r < m.N ? row0 + r : m.N - 1;
    x0[d * MLP_ST + r] = m.X[(size_t)row * m.D + (d < m.D ? d : 0)] * (d < m.D ? 1.0 : 0.0);
  }
}

// A workgroup owns `per` consecutive 64-row chunks of one net (the weights are staged once, the partial gradients of its
// chunks add up in its own slot): per = the smallest count that lets the whole grid be resident at once
// (slots = resident workgroups on 256 CUs), so no workgroup waits for a slot behind a full first round.
__host__ __device__ inline int mlp_chunks(int N) { return (N + MLP_T - 1) / MLP_T; }
__host__ __device__ inline int mlp_per_block(int N, int nnets, int slots) {
  const long long work = (long long)mlp_chunks(N) * nnets;
  return (int)((work + slots - 1) / slots);
}
__host__ __device__ inline int mlp_groups(int N, int nnets, int slots) {
  const int per = mlp_per_block(N, nnets, slots);
  return (mlp_chunks(N) + per - 1) / per;
}
#define MLP_SLOTS_FWD 1024 /* 4 workgroups per CU */
#define MLP_SLOTS_BWD 512  /* 2 workgroups per CU (LDS) */

// activation and dropout of one accumulator tile (lane: units 16 jt + q + 4 rr of its row).  Straight-line: the
// activation is a template parameter and dropout a select (scale = 1, threshold = 0 when it is off) -- a branch per
// element ended the scheduling region after every tile, and a wave then paid its LDS and MFMA latencies one by one.
template <int ACT>
__device__ __forceinline__ d4 mlp_tile_act(d4 acc, unsigned th, double scale, uint64_t seed, int step, int net, int layer, int row,
                                           int jt, int q) {
  const uint64_t h4 = mlp_hash4(seed, step, net, layer, row, 4 * jt + q);
  d4 v;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const double x = mlp_act<ACT>(acc[rr]) * scale;
    v[rr] = ((unsigned)((h4 >> (16 * rr)) & 0xFFFFu) >= th) ? x : 0.0;
  }
  return v;
}

// first hidden layer, all NJT output tiles of this wave's 16 rows: A = W_0 tiles (lane: unit 16 jt + n, k = input
// 4 s + q), B = the x0 strip (lane: input 4 s + q of row n), shared by the tiles.  pad4(D) / 4 k-steps (1 at Power).
template <int NJT>
__device__ __forceinline__ void mlp_first_layer(const double* W0, const double* b0, int KP0, const double* x0w, d4 (&acc)[NJT], int n,
                                                int q) {
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[jt][rr] = b0[16 * jt + q + 4 * rr];
  const double* wrow = W0 + n * KP0 + q;
  for (int k = 0; k < KP0; k += 4) {
    const double b = x0w[(k + q) * MLP_ST + n];
    double af[NJT];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) af[jt] = wrow[16 * jt * KP0 + k];
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) acc[jt] = TGP_MFMA(af[jt], b, acc[jt]);
  }
}

// hidden layer l >= 1, output tile jt: A = W_l tile (lane: unit 16 jt + n, k = 4 s + q), B = the previous layer's
// accumulator registers (k-step s = register s & 3 of tile s >> 2).  All NKS fragments are requested before the MFMAs.
template <int NKS, int NJT>
__device__ __forceinline__ d4 mlp_hidden_tile(const double* Wl, const d4 (&in)[NJT], int jt, int n, int q) {
  constexpr int R = 4 * NKS;
  const double* bl = Wl + R * R;
  d4 acc;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) acc[rr] = bl[16 * jt + q + 4 * rr];
  const double* wrow = Wl + (16 * jt + n) * R + q;
  double af[NKS];
#pragma unroll
  for (int s = 0; s < NKS; ++s) af[s] = wrow[4 * s];
#pragma unroll
  for (int s = 0; s < NKS; ++s) acc = TGP_MFMA(af[s], in[s >> 2][s & 3], acc);
  return acc;
}

// back-propagation through layer l >= 1, tile it of the layer below: A = W_l^T tile (lane: unit i = 16 it + n of the
// layer below, k = unit j = 4 s + q of layer l), B = delta_l in registers.
template <int NKS, int NJT>
__device__ __forceinline__ d4 mlp_back_tile(const double* Wl, const d4 (&dl)[NJT], int it, int n, int q) {
  constexpr int R = 4 * NKS;
  const double* wcol = Wl + q * R + 16 * it + n;
  double af[NKS];
#pragma unroll
  for (int s = 0; s < NKS; ++s) af[s] = wcol[4 * s * R];
  d4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < NKS; ++s) acc = TGP_MFMA(af[s], dl[s >> 2][s & 3], acc);
  return acc;
}

template <int L, int NKS, int ACT>
__global__ __launch_bounds__(MLP_NT, L == 3 ? 2 : 4) void k_mlp_fwd(MlpArgs m, double* __restrict__ out) {
  constexpr int NJT = (4 * NKS + 15) / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int net = blockIdx.y, D = m.D, H = m.H;
  const int PW = mlp_weights_per_net(D, H, L);
  const MlpLds Lo = mlp_lds(D, H, L, false);
  mlp_stage_weights<4 * NKS, L>(m, Lo, sm, net, tid, PW);
  const int step = m.step_dev ? m.step_dev[0] : 0;
  const bool drop = m.training && m.p > 0.0;
  const double scale = drop ? 1.0 / (1.0 - m.p) : 1.0;
  const unsigned th = drop ? mlp_thresh(m.p) : 0u;   // 0: every unit kept
  const int per = mlp_per_block(m.N, m.nnets, MLP_SLOTS_FWD), nch = mlp_chunks(m.N);
  const int c0 = blockIdx.x * per, c1 = min(c0 + per, nch);
  const int n0 = lane & 15, q0 = lane >> 4;
  for (int c = c0; c < c1; ++c) {
    if (c > c0) __syncthreads();   // the previous chunk's inputs have been read
    mlp_stage_inputs(m, sm + Lo.x0, c * MLP_T, tid);
    __syncthreads();
    // lane coordinates the optimiser cannot see through: with loop-invariant n, q it hoists the ~60 clamped fragment
    // addresses of a chunk out of the chunk loop and spills them
    int n = n0, q = q0;
    asm volatile("" : "+v"(n), "+v"(q));
    const int row = c * MLP_T + 16 * wave + n;
    d4 a[2][NJT];
    mlp_first_layer<NJT>(sm, sm + Lo.bias(0), Lo.KP0, sm + Lo.x0 + 16 * wave, a[0], n, q);
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) a[0][jt] = mlp_tile_act<ACT>(a[0][jt], th, scale, m.seed, step, net, 0, row, jt, q);
#pragma unroll
    for (int l = 1; l < L; ++l) {
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt)
        a[l & 1][jt] = mlp_tile_act<ACT>(mlp_hidden_tile<NKS, NJT>(sm + Lo.img(l), a[(l - 1) & 1], jt, n, q), th, scale,
                                         m.seed, step, net, l, row, jt, q);
    }
    const double* wo = sm + Lo.img(L);   // zero beyond H, like the activations
    double s = 0.0;
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        if (16 * jt + 4 * rr < 4 * NKS) s = fma(wo[16 * jt + q + 4 * rr], a[(L - 1) & 1][jt][rr], s);
    s = quad_sum(s) + wo[4 * NKS];
    if (q == 0 && row < m.N) out[(size_t)row * m.nnets + net] = s;
  }
}

// dW_l = delta_l a_{l-1}^T over the block's 64 rows, unit tile `wave` of delta_l:  A = delta_l^T fragments from strip SA
// (lane: unit 16 wave + n, row 4 t + q), B = a_{l-1}^T fragments from strip SB (lane: unit 16 it + n, row 4 t + q);
// db_l = lane-local sum of the A fragments, finished across the four q.  `first`: store, else add to the workgroup's own
// partial (its earlier chunks).  NIT > 0: that many tiles of a_{l-1}, straight
// line (hidden layers); NIT == 0: ceil(nin / 16) tiles in a loop (the inputs).
template <int NIT>
__device__ __forceinline__ void mlp_contract(const double* SA, const double* SB, int H, int nin, double* __restrict__ gW, bool first,
                                             int wave, int n, int q) {
  if (16 * wave >= H) return;
  // units beyond H / nin read the last row of their strip: a row of A (a column of B) only reaches the same row (column)
  // of the product, and those are not stored
  const int ja = 16 * wave + n;
  const double* pa = SA + min(ja, H - 1) * MLP_ST + q;
  double af[16], bs = 0.0;
#pragma unroll
  for (int t = 0; t < 16; ++t) af[t] = pa[4 * t];
#pragma unroll
  for (int t = 0; t < 16; ++t) bs += af[t];
  bs = quad_sum(bs);
  if (q == 0 && ja < H) {
    double* o = gW + H * nin + ja;
    const double old = *o;
    *o = first ? bs : old + bs;
  }
  auto tile = [&](int it) {
    const int ib = 16 * it + n;
    const double* pb = SB + min(ib, nin - 1) * MLP_ST + q;
    double bf[16], old[4];
#pragma unroll
    for (int t = 0; t < 16; ++t) bf[t] = pb[4 * t];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = 16 * wave + q + 4 * rr;
      old[rr] = gW[min(j, H - 1) * nin + min(ib, nin - 1)];   // requested before the MFMAs; unused on the first chunk
    }
    d4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = TGP_MFMA(af[t], bf[t], acc);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = 16 * wave + q + 4 * rr;
      if (j < H && ib < nin) gW[j * nin + ib] = first ? acc[rr] : old[rr] + acc[rr];
    }
  };
  if constexpr (NIT > 0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) tile(it);
  } else {
    for (int it = 0; 16 * it < nin; ++it) tile(it);
  }
}

// this wave's 16 rows of a layer, transposed into a strip [unit][row]
template <int NJT>
__device__ __forceinline__ void mlp_put_strip(double* S, int H, const d4 (&v)[NJT], int rl, int q) {
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = 16 * jt + q + 4 * rr;
      if (j < H) S[j * MLP_ST + rl] = v[jt][rr];
    }
}

// backward; partial weight gradients of this (row block, net) into part[(blockIdx.x * nnets + net) * PW ...]
template <int L, int NKS, int ACT>
__global__ __launch_bounds__(MLP_NT, 2) void k_mlp_bwd(MlpArgs m, const double* __restrict__ g_out, double* __restrict__ part) {
  constexpr int NJT = (4 * NKS + 15) / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* sm = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int net = blockIdx.y, D = m.D, H = m.H;
  const int PW = mlp_weights_per_net(D, H, L);
  const MlpLds Lo = mlp_lds(D, H, L, true);
#ifdef TGP_STAMPS
  double* mst = part + (size_t)gridDim.x * m.nnets * PW;  // the 16 spare doubles behind the partials
#endif
  MSTAMP(0);
  mlp_stage_weights<4 * NKS, L>(m, Lo, sm, net, tid, PW);
  const int step = m.step_dev ? m.step_dev[0] : 0;
  const bool drop = m.training && m.p > 0.0;
  const double scale = drop ? 1.0 / (1.0 - m.p) : 1.0;
  const unsigned th = drop ? mlp_thresh(m.p) : 0u;   // 0: every unit kept
  auto strip = [&](int l) { return sm + Lo.strip + l * Lo.sstride; };
  double* dwo = sm + Lo.dwo;
  double* gp = part + ((size_t)blockIdx.x * m.nnets + net) * PW;
  const int per = mlp_per_block(m.N, m.nnets, MLP_SLOTS_BWD), nch = mlp_chunks(m.N);
  const int c0 = blockIdx.x * per, c1 = min(c0 + per, nch);
  const int n0 = lane & 15, q0 = lane >> 4;
  for (int c = c0; c < c1; ++c) {
    const bool first = c == c0;
    if (!first) __syncthreads();   // the previous chunk's strips and inputs have been read
    mlp_stage_inputs(m, sm + Lo.x0, c * MLP_T, tid);
    int n = n0, q = q0;            // opaque per chunk (see k_mlp_fwd)
    asm volatile("" : "+v"(n), "+v"(q));
    const int rl = 16 * wave + n, row = c * MLP_T + rl;
    const double go = row < m.N ? g_out[(size_t)row * m.nnets + net] : 0.0;
    __syncthreads();
    MSTAMP(1);
    // ---- forward recomputation in registers; a_1 .. a_{L-1} also go to their strips (the contractions' B operands) ----
    d4 a[L][NJT];
    mlp_first_layer<NJT>(sm, sm + Lo.bias(0), Lo.KP0, sm + Lo.x0 + 16 * wave, a[0], n, q);
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) a[0][jt] = mlp_tile_act<ACT>(a[0][jt], th, scale, m.seed, step, net, 0, row, jt, q);
    if (L > 1) mlp_put_strip<NJT>(strip(0), H, a[0], rl, q);
#pragma unroll
    for (int l = 1; l < L; ++l) {
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt)
        a[l][jt] = mlp_tile_act<ACT>(mlp_hidden_tile<NKS, NJT>(sm + Lo.img(l), a[l - 1], jt, n, q), th, scale, m.seed,
                                     step, net, l, row, jt, q);
      if (l < L - 1) mlp_put_strip<NJT>(strip(l), H, a[l], rl, q);
    }
    MSTAMP(2);
    // derivative of (activation -> dropout) through the stored value; the keep flag only matters for tanh
    auto dfac4 = [&](const d4& av, int layer, int jt) {
      d4 f;
      if constexpr (ACT == 0) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) f[rr] = av[rr] > 0.0 ? scale : 0.0;
      } else {
        const uint64_t h4 = mlp_hash4(m.seed, step, net, layer, row, 4 * jt + q);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const double t = av[rr] / scale;
          f[rr] = ((unsigned)((h4 >> (16 * rr)) & 0xFFFFu) >= th) ? scale * (1.0 - t * t) : 0.0;
        }
      }
      return f;
    };
    {
      // output layer out = wo . a_L + bo:  dwo[j] = sum_rows go a_L[j][row], dbo = sum_rows go -- this wave's 16 rows are
      // the 16 lanes of a DPP row (four rotate-and-add steps per value), the four waves meet in LDS after the barrier;
      // then delta_L = wo go act'(a_L), in the registers of a_L
      const double* wo = sm + Lo.img(L);
      auto row16_sum = [](double x) { return ror_sum<1>(ror_sum<2>(ror_sum<4>(ror_sum<8>(x)))); };
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt) {
        const d4 f = dfac4(a[L - 1][jt], L - 1, jt);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          if (16 * jt + 4 * rr < 4 * NKS) {   // units beyond R: never stored, never a k index
            const int j = 16 * jt + q + 4 * rr;
            const double sj = row16_sum(go * a[L - 1][jt][rr]);
            if (n == 0 && j < H) dwo[wave * (H + 1) + j] = sj;
            a[L - 1][jt][rr] = wo[j] * go * f[rr];
          }
        }
      }
      const double sg = row16_sum(go);
      if (lane == 0) dwo[wave * (H + 1) + H] = sg;
    }
    mlp_put_strip<NJT>(strip(L - 1), H, a[L - 1], rl, q);
    __syncthreads();
    MSTAMP(3);
    if (tid <= H) {
      const double sj = (dwo[tid] + dwo[H + 1 + tid]) + (dwo[2 * (H + 1) + tid] + dwo[3 * (H + 1) + tid]);
      double* o = gp + mlp_woff(D, H, L) + tid;
      const double old = *o;             // (garbage before the workgroup's first chunk: selected away)
      *o = first ? sj : old + sj;
    }
    MSTAMP(4);
    // ---- hidden layers, last to first: strip l holds delta_{l+1}^T, strip l - 1 (or x0) the layer's input ----
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
      if (l > 0) {
        // delta of the layer below, in the registers of its activation (needs delta_{l+1} of ALL units: registers only)
#pragma unroll
        for (int it = 0; it < NJT; ++it) {
          const d4 acc = mlp_back_tile<NKS, NJT>(sm + Lo.img(l), a[l], it, n, q);
          const d4 f = dfac4(a[l - 1][it], l - 1, it);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) a[l - 1][it][rr] = acc[rr] * f[rr];
        }
        mlp_contract<NJT>(strip(l), strip(l - 1), H, H, gp + mlp_woff(D, H, l), first, wave, n, q);
      } else {
        mlp_contract<0>(strip(0), sm + Lo.x0, H, D, gp, first, wave, n, q);
      }
      MSTAMP(5 + 2 * (L - 1 - l));
      if (l == 0) break;
      __syncthreads();   // a_l^T has been read by every wave
      mlp_put_strip<NJT>(strip(l - 1), H, a[l - 1], rl, q);
      __syncthreads();
      MSTAMP(6 + 2 * (L - 1 - l));
    }
  }
}

// g_W[net][k] = sum over row blocks of the partials (fixed order).  One workgroup = 64 consecutive elements x 4 quarters
// of the row blocks (a wave per quarter); 16 partials are requested before the first add -- with two loads in flight per
// thread the 135 partials of a Power-sized step were 67 dependent L2 round trips (21 us on the critical path of the
// ID_TGP step); the four quarter sums meet in LDS and are added in a fixed order.
// `ad` (tgp_mlp_backward_adam_f64): the thread that forms a weight's gradient applies torch.optim.Adam to it (k_adam_dev's
// arithmetic, weight decay `wd` on every entry: these are the reference's 'NNets' group, main.py:276-288) and the last
// workgroup bumps the group's device step counter -- the update no longer costs the ID_TGP side chain a launch of its own.
__global__ __launch_bounds__(256) void k_mlp_reduce(const double* __restrict__ part, int nblk, size_t len, double* __restrict__ g_W,
                                                     AdamDev ad, double wd) {
  __shared__ double q4[4][64];
  const double step = ad.p != nullptr ? (double)(ad.step_dev[0] + 1) : 1.0;   // (read before any workgroup's ticket)
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
  if (qt == 0 && e < len) {
    const double g = (q4[0][el] + q4[1][el]) + (q4[2][el] + q4[3][el]);
    g_W[e] = g;
    if (ad.p != nullptr) {
      const double bc1 = 1.0 - exp_fast(step * ad.ln_b1), bc2s = sqrt(1.0 - exp_fast(step * ad.ln_b2));
      double gi = ad.sign * g;
      if (wd != 0.0) gi += wd * ad.p[e];
      const double mi = ad.b1 * ad.m[e] + (1.0 - ad.b1) * gi;
      const double vi = ad.b2 * ad.v[e] + (1.0 - ad.b2) * gi * gi;
      ad.m[e] = mi;
      ad.v[e] = vi;
      ad.p[e] -= (ad.lr / bc1) * mi / (sqrt(vi) / bc2s + ad.eps);
    }
  }
  if (ad.p != nullptr) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const int t = atomicAdd(&ad.step_dev[1], 1);
      if (t == (int)gridDim.x - 1) {
        ad.step_dev[1] = 0;
        atomicAdd(&ad.step_dev[0], 1);
      }
    }
  }
}

static size_t mlp_lds_bytes(int D, int H, int L, bool bwd) { return (size_t)mlp_lds(D, H, L, bwd).total * sizeof(double); }

size_t mlp_workspace_doubles(int N, int D, int H, int L, int nnets) {
  const size_t nblk = (size_t)mlp_groups(N, nnets, MLP_SLOTS_BWD);
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

// kernel variants: L hidden layers x NKS k-steps per hidden product (4 NKS >= H; 13 is the reference's 50 units) x activation
using FwdKernel = void (*)(MlpArgs, double*);
using BwdKernel = void (*)(MlpArgs, const double*, double*);
#define MLP_NVAR 18
static int mlp_variant(int H, int L, int act) { return ((L - 1) * 3 + (H <= 32 ? 0 : H <= 52 ? 1 : 2)) * 2 + act; }
#define MLP_ROW(K, L_) K<L_, 8, 0>, K<L_, 8, 1>, K<L_, 13, 0>, K<L_, 13, 1>, K<L_, 16, 0>, K<L_, 16, 1>
static FwdKernel mlp_fwd_kernel(int v) {
  static const FwdKernel t[MLP_NVAR] = {MLP_ROW(k_mlp_fwd, 1), MLP_ROW(k_mlp_fwd, 2), MLP_ROW(k_mlp_fwd, 3)};
  return t[v];
}
static BwdKernel mlp_bwd_kernel(int v) {
  static const BwdKernel t[MLP_NVAR] = {MLP_ROW(k_mlp_bwd, 1), MLP_ROW(k_mlp_bwd, 2), MLP_ROW(k_mlp_bwd, 3)};
  return t[v];
}

int launch_mlp_forward(const tgp_mlp& d, const double* X, const double* W, const int32_t* step_dev, double* out,
                       hipStream_t st) {
  if (int rc = mlp_check(d)) return rc;
  const size_t lds = mlp_lds_bytes(d.D, d.H, d.L, false);
  static size_t lds_cur[MLP_NVAR];   // 0: below the 64 KiB default until a launch says otherwise
  const int v = mlp_variant(d.H, d.L, d.act);
  const FwdKernel k = mlp_fwd_kernel(v);
  if (int rc = ensure_lds(reinterpret_cast<const void*>(k), lds, &lds_cur[v])) return rc;
  hipLaunchKernelGGL(k, dim3(mlp_groups(d.N, d.nnets, MLP_SLOTS_FWD), d.nnets), dim3(MLP_NT), lds, st, mlp_args(d, X, W, step_dev), out);
  LAUNCH_CHECK();
  return 0;
}

int launch_mlp_backward(const tgp_mlp& d, const double* X, const double* W, const int32_t* step_dev, const double* g_out,
                        double* g_W, double* ws, size_t ws_doubles, hipStream_t st, const AdamDev* adam, double weight_decay) {
  if (int rc = mlp_check(d)) return rc;
  if (ws_doubles < mlp_workspace_doubles(d.N, d.D, d.H, d.L, d.nnets)) return TGP_E_WORKSPACE;
  const size_t lds = mlp_lds_bytes(d.D, d.H, d.L, true);
  static size_t lds_cur[MLP_NVAR];   // 0: below the 64 KiB default until a launch says otherwise
  const int v = mlp_variant(d.H, d.L, d.act);
  const BwdKernel k = mlp_bwd_kernel(v);
  if (int rc = ensure_lds(reinterpret_cast<const void*>(k), lds, &lds_cur[v])) return rc;
  const int nblk = mlp_groups(d.N, d.nnets, MLP_SLOTS_BWD);
  hipLaunchKernelGGL(k, dim3(nblk, d.nnets), dim3(MLP_NT), lds, st, mlp_args(d, X, W, step_dev), g_out, ws);
  LAUNCH_CHECK();
  const size_t len = (size_t)d.nnets * mlp_weights_per_net(d.D, d.H, d.L);
  AdamDev ad;
  if (adam != nullptr) {
    if ((size_t)adam->n != len) return -9;   // the update covers exactly the weights whose gradients this call forms
    ad = *adam;
  }
  hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)((len + 63) / 64)), dim3(256), 0, st, ws, nblk, len, g_W, ad, weight_decay);
  LAUNCH_CHECK();
  return 0;
}

}  // namespace tgp
