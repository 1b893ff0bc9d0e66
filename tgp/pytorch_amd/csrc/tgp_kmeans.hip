// tgp_kmeans.hip -- inducing-point initialisation on the GPU: the numerical kernels of
// k-means++ seeding and Lloyd iterations behind utils.KMEANS (dsp/utils.py:143-159, which calls
// sklearn.cluster.KMeans(init='k-means++')).  The random draws and the control flow (sklearn's stopping rule,
// n_init restarts) stay on the host in utils.py; everything of size N x K or N x trials is here.
// All sums are fixed-order (no float atomics): results are reproducible run to run.
#include "tgp_dev.hpp"
#include "tgp_launch.hpp"

namespace tgp {

#define LAUNCH_CHECK()                                              \
  do {                                                              \
    hipError_t e_ = hipGetLastError();                              \
    if (e_ != hipSuccess) return set_error(e_, __FILE__, __LINE__); \
  } while (0)

// E-step: label_n = argmin_k |x_n - c_k|^2 (first minimum wins, like np.argmin), mind2_n = that distance.
// One thread per row; centres stream through LDS in chunks of KC.
#define KM_KC 256
__global__ __launch_bounds__(256) void k_kmeans_assign(const double* __restrict__ X, int N, int D, const double* __restrict__ C,
                                                        int K, int32_t* __restrict__ labels, double* __restrict__ mind2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* cl = reinterpret_cast<double*>(smem_raw);  // KM_KC x D
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int nc = n < N ? n : N - 1;
  double x[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) x[d] = d < D ? X[(size_t)nc * D + d] : 0.0;
  double best = INFINITY;
  int bk = 0;
  for (int k0 = 0; k0 < K; k0 += KM_KC) {
    const int kc = min(KM_KC, K - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < kc * D; i += 256) cl[i] = C[(size_t)k0 * D + i];
    __syncthreads();
    for (int k = 0; k < kc; ++k) {
      double d2 = 0.0;
      for (int d = 0; d < D; ++d) {
        const double t = x[d] - cl[k * D + d];
        d2 = fma(t, t, d2);
      }
      if (d2 < best) { best = d2; bk = k0 + k; }
    }
  }
  if (n < N) {
    labels[n] = bk;
    if (mind2) mind2[n] = best;
  }
}

// M-step sums: rows sorted by label (order[], segment offsets offs[k]..offs[k+1]); one wave per cluster, lanes stride
// over the segment in a fixed order, then a fixed-tree wave reduction.  sums (K, D).
__global__ __launch_bounds__(256) void k_kmeans_segsum(const double* __restrict__ X, int D, const int64_t* __restrict__ order,
                                                        const int64_t* __restrict__ offs, int K, double* __restrict__ sums) {
  const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  const int64_t b = offs[k], e = offs[k + 1];
  double acc[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) acc[d] = 0.0;
  for (int64_t i = b + lane; i < e; i += 64) {
    const double* xr = X + (size_t)order[i] * D;
#pragma unroll
    for (int d = 0; d < 16; ++d)
      if (d < D) acc[d] += xr[d];
  }
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    if (d < D) {
      const double s = wave_sum(acc[d]);
      if (lane == 0) sums[(size_t)k * D + d] = s;
    }
  }
}

// k-means++ trial step: out[t][n] = min(closest[n], |x_n - x_cand[t]|^2)   (sklearn _kmeans_plusplus, the
// np.minimum(closest_dist_sq, distance_to_candidates) line); T trials <= 16
__global__ __launch_bounds__(256) void k_kmeans_pp(const double* __restrict__ X, int N, int D, const int64_t* __restrict__ cand,
                                                    int T, const double* __restrict__ closest, double* __restrict__ out) {
  __shared__ double cx[16 * 16];
  for (int i = threadIdx.x; i < T * D; i += 256) cx[i] = X[(size_t)cand[i / D] * D + i % D];
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  double x[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) x[d] = d < D ? X[(size_t)n * D + d] : 0.0;
  const double c0 = closest ? closest[n] : INFINITY;
  for (int t = 0; t < T; ++t) {
    double d2 = 0.0;
    for (int d = 0; d < D; ++d) {
      const double u = x[d] - cx[t * D + d];
      d2 = fma(u, u, d2);
    }
    out[(size_t)t * N + n] = fmin(c0, d2);
  }
}

int launch_kmeans_assign(const double* X, int N, int D, const double* C, int K, int32_t* labels, double* mind2, hipStream_t st) {
  hipLaunchKernelGGL(k_kmeans_assign, dim3((N + 255) / 256), dim3(256), (size_t)KM_KC * D * sizeof(double), st, X, N, D, C, K,
                     labels, mind2);
  LAUNCH_CHECK();
  return 0;
}

int launch_kmeans_segsum(const double* X, int D, const int64_t* order, const int64_t* offs, int K, double* sums, hipStream_t st) {
  hipLaunchKernelGGL(k_kmeans_segsum, dim3((K + 3) / 4), dim3(256), 0, st, X, D, order, offs, K, sums);
  LAUNCH_CHECK();
  return 0;
}

int launch_kmeans_pp(const double* X, int N, int D, const int64_t* cand, int T, const double* closest, double* out,
                     hipStream_t st) {
  hipLaunchKernelGGL(k_kmeans_pp, dim3((N + 255) / 256), dim3(256), 0, st, X, N, D, cand, T, closest, out);
  LAUNCH_CHECK();
  return 0;
}

}  // namespace tgp
