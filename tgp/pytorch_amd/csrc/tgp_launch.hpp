// tgp_launch.hpp -- host-side launcher declarations shared by the translation units of libtgp_hip.so
#pragma once
#include "tgp_dev.hpp"

namespace tgp {

int set_error(hipError_t e, const char* file, int line);  // records tgp_last_error(), returns TGP_E_LAUNCH
void set_error_text(const char* fmt, ...);                // records tgp_last_error()

// tgp_comm.hip (RCCL bound at run time)
int comm_load(const char* path);
int comm_unique_id(void* id128);
int comm_init(const void* id128, int nranks, int rank, void** comm);
int comm_allreduce(void* comm, double* buf, int64_t n, hipStream_t st);
int comm_destroy(void* comm);

// Raise a kernel's dynamic-LDS ceiling when a launch needs more than the 64 KiB default (gfx950: 160 KiB/CU).
// `cur` is the caller's per-kernel high-water mark.  Returns 0 or TGP_E_LDS.
inline int ensure_lds(const void* func, size_t bytes, size_t* cur) {
  if (bytes <= *cur) return 0;
  if (bytes > 160 * 1024 - 1024) return TGP_E_LDS;  // leave room for the static __shared__ words
  const size_t want = bytes > 63 * 1024 ? 160 * 1024 - 1024 : 63 * 1024;
  hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
  if (e != hipSuccess) { (void)hipGetLastError(); return set_error(e, __FILE__, __LINE__); }
  *cur = want;
  return 0;
}

// tgp_mm.hip
int launch_prepare(const Plan& p, const tgp_model& md, const FlowProg& fp, double* ws, int32_t* status, hipStream_t st);
// Adam folded into the backward launches of the fused path (tgp_elbo_step_adam_f64); p == nullptr: none
struct AdamDev {
  double* p = nullptr;
  const double* g = nullptr;
  double* m = nullptr;
  double* v = nullptr;
  long n = 0, lam_off = 0, lam_n = 0;  // [lam_off, lam_off + lam_n): the q(u) factor's entries, updated beside the K_MM adjoint
  double lr = 0, b1 = 0, b2 = 0, eps = 0, ln_b1 = 0, ln_b2 = 0, sign = 1;
  int32_t* step_dev = nullptr;
};
int launch_backward_mm(const Plan& p, const tgp_model& md, const tgp_grads& g, double* out, double* ws, int32_t* status,
                       hipStream_t st, const AdamDev* adam = nullptr);
int launch_kmm(const double* Z, const double* raw_ls, const double* raw_os, int M, int D, double jitter, double* K,
               hipStream_t st);
int launch_knm(const double* X, const double* Z, const double* raw_ls, const double* raw_os, int N, int M, int D,
               double* K, hipStream_t st);
int launch_kernel_matrix(int kernel, const double* X1, int N1, const double* X2, int N2, int D, const double* raw_ls,
                         const double* raw_os, double jitter, double* K, hipStream_t st);
int launch_kl(const double* m, const double* Lam, int M, double* out, double* g_m, double* g_Lam, hipStream_t st);
int launch_cholesky(const double* A, int M, double* L, double* Linv, int32_t* status, hipStream_t st);

// tgp_rows.hip
// which row kernel a plan gets: 0 = k_rows (16 rows per wave), else the waves per workgroup of k_rows4 (4 rows per wave)
int choose_rows4(const Plan& p, bool train, int sel = 0);
// data rows per wave of k_rows for this plan (16, or 10: training with the flow likelihood at Power-like sizes)
int rows_per_wave(const Plan& p, const FlowProg& fp, bool train, int sel = 0);
// does the training launch of the row kernel fit a CU's LDS with `nslots` flow-stack slots (its leanest form: one node in flight)?
bool rows_train_lds_fits(const Plan& p, int nslots);
int launch_rows(const Plan& p, const tgp_model& md, const FlowProg& fp, const double* X, const double* Y,
                const double* rowp, double* g_rowp, double* mu, double* v, double* ws, bool train, hipStream_t st);

// tgp_big.hip (general-M path, 128 < M <= TGP_BIG_MAX_M)
size_t big_workspace_doubles(int N, int D, int M, int S, int nblk, int P, int RP, int kernel, int plan = 0);
int launch_big_step(const tgp_model& md, const FlowProg& fp, const double* X, const double* Y, const double* rowp,
                    double* out, const tgp_grads& g, double* mu, double* v, int32_t* status, double* ws, size_t ws_doubles,
                    uint32_t phases, hipStream_t st, const AdamDev* adam = nullptr);
int launch_big_moments(const tgp_model& md, const double* X, double* mu, double* v, int32_t* status, double* ws,
                       size_t ws_doubles, hipStream_t st);
size_t big_cholesky_workspace_doubles(int M);
int launch_big_cholesky(const double* A, int M, double* Lo, double* Jo, int32_t* status, double* ws, size_t ws_doubles,
                        hipStream_t st);
int launch_big_cholesky_bwd(const double* L, const double* Linv, const double* Lbar, int M, double* Abar, double* ws,
                            size_t ws_doubles, hipStream_t st);
int launch_gemm_plain(bool ta, bool tb, int tri, int m, int n, int k, double alpha, const double* A, int lda, const double* B,
                      int ldb, double beta, double* C, int ldc, hipStream_t st);

// tgp_kmeans.hip
int launch_kmeans_assign(const double* X, int N, int D, const double* C, int K, int32_t* labels, double* mind2, hipStream_t st);
int launch_kmeans_segsum(const double* X, int D, const int64_t* order, const int64_t* offs, int K, double* sums, hipStream_t st);
int launch_kmeans_pp(const double* X, int N, int D, const int64_t* cand, int T, const double* closest, double* out,
                     hipStream_t st);

// tgp_mlp.hip
size_t mlp_workspace_doubles(int N, int D, int H, int L, int nnets);
int launch_mlp_forward(const tgp_mlp& d, const double* X, const double* W, const int32_t* step_dev, double* out, hipStream_t st);
int launch_mlp_backward(const tgp_mlp& d, const double* X, const double* W, const int32_t* step_dev, const double* g_out,
                        double* g_W, double* ws, size_t ws_doubles, hipStream_t st, const AdamDev* adam = nullptr,
                        double weight_decay = 0.0);

// tgp_lik.hip
int launch_ell_gauss(const double* Y, const double* mu, const double* v, int N, const double* log_var_noise,
                     double scale, double* out, double* g_mu, double* g_v, double* ws, hipStream_t st);
int launch_ell_flow(const tgp_model& md, const FlowProg& fp, const double* Y, const double* mu, const double* v, const double* rowp,
                    double* out, double* g_mu, double* g_v, double* g_theta, double* g_rowp, double* ws,
                    hipStream_t st);
int launch_flow_eval(const tgp_model& md, const FlowProg& fp, const double* f, int S, int N, const double* rowp, double* G, double* dG,
                     double* logdG, hipStream_t st, double* sum_out = nullptr, double* ws = nullptr);
int launch_predict(const tgp_model& md, const FlowProg& fp, const double* mu, const double* v, const double* rowp, const double* Y,
                   double Y_std, double* m1, double* m2, double* logp, hipStream_t st);
int launch_adam(double* params, const double* grads, double* exp_avg, double* exp_avg_sq, int64_t n, double lr,
                double beta1, double beta2, double eps, double weight_decay, int step, int maximize, hipStream_t st);
int launch_adam_dev(double* params, const double* grads, double* exp_avg, double* exp_avg_sq, int64_t n, double lr,
                    double beta1, double beta2, double eps, double weight_decay, int32_t* step_dev, int maximize,
                    hipStream_t st, int64_t n_plain = 0, int64_t skip_off = 0, int64_t skip_n = 0);
size_t lik_workspace_doubles(int N, int P, int RP);
int launch_gather_rows(const double* X, const double* Y, int N, int D, const int32_t* index, int32_t* cursor, int offset,
                       int nrows, int advance, int wrap, double* Xb, double* Yb, hipStream_t st);

}  // namespace tgp
