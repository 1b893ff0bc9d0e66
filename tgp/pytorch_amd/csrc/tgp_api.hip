// tgp_api.hip -- the extern "C" surface of libtgp_hip.so (declared in include/tgp_hip.h).
// Argument checking, plan/workspace bookkeeping and the launch sequence of one ELBO step:
//   M <= 128 : k_prep_a -> row kernel (k_rows / k_rows4) -> k_reduce -> k_bwd   (4 launches, no host sync)
//   M  > 128 : the chunked GEMM pipeline of tgp_big.hip (same entry points, chosen by M / kernel)
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "tgp_dev.hpp"
#include "tgp_launch.hpp"

namespace tgp {

static thread_local char g_err[256] = "";

int set_error(hipError_t e, const char* file, int line) {
  snprintf(g_err, sizeof(g_err), "%s:%d: %s", file, line, hipGetErrorString(e));
  return TGP_E_LAUNCH;
}

void set_error_text(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int check_model(const tgp_model* m, bool need_lik) {
  if (m == nullptr) return -1;
  if (m->N < 1 || m->D < 1 || m->D > 16) return -1;
  if (m->M < 1) return -1;
  if (m->M > TGP_BIG_MAX_M) return TGP_E_UNSUPPORTED;
  if (m->kernel != TGP_KERNEL_SCALE_RBF && m->kernel != TGP_KERNEL_SCALE_MATERN32) return -1;
  if (!m->Z || !m->raw_ls || !m->raw_os || !m->m || !m->Lam || !m->log_var_noise) return -1;
  if (need_lik && m->lik == TGP_LIK_FLOW) {
    if (m->S < 1 || m->nblk < 0 || !m->xs || !m->wn) return -1;
    if (m->nblk > 0 && !m->program) return -1;
    if (m->P > 0 && !m->theta) return -1;
  }
  return 0;
}

// host array (nblk x 4 int32) -> by-value kernel argument; validates kinds and parameter offsets
static int make_prog(const tgp_model* m, bool flow, FlowProg& fp) {
  fp.nblk = 0;
  fp.nslots = 0;
  if (!flow || m->nblk <= 0) return 0;
  if (m->nblk > TGP_MAX_BLOCKS) return TGP_E_UNSUPPORTED;
  if (!m->program) return -1;
  fp.nblk = m->nblk;
  for (int b = 0; b < m->nblk; ++b) {
    const int kind = m->program[4 * b], K = m->program[4 * b + 1], poff = m->program[4 * b + 2],
              flags = m->program[4 * b + 3];
    if (kind < TGP_FLOW_AFFINE || kind > TGP_FLOW_STEPTANH) return -1;
    const int np = kind == TGP_FLOW_STEPTANH ? 4 * K : 2;
    if (kind == TGP_FLOW_STEPTANH && (K < 1 || (flags & TGP_FLAG_PER_ROW))) return -1;
    if (poff < 0 || poff + np > ((flags & TGP_FLAG_PER_ROW) ? m->RP : m->P)) return -1;
    for (int j = 0; j < 4; ++j) fp.blk[4 * b + j] = m->program[4 * b + j];
  }
  fp.nslots = flow_slots(fp.blk, fp.nblk);
  return 0;
}

}  // namespace tgp

using namespace tgp;

extern "C" {

int tgp_version(void) { return TGP_VERSION; }

const char* tgp_last_error(void) { return g_err; }

#ifndef TGP_SRC_HASH
#define TGP_SRC_HASH "unstamped"
#endif
const char* tgp_source_hash(void) { return TGP_SRC_HASH; }

size_t tgp_workspace_bytes(int32_t N, int32_t D, int32_t M, int32_t S, int32_t nblk, int32_t P, int32_t RP) {
  return tgp_workspace_bytes_kernel(N, D, M, S, nblk, P, RP, TGP_KERNEL_SCALE_RBF);
}

size_t tgp_workspace_bytes_kernel(int32_t N, int32_t D, int32_t M, int32_t S, int32_t nblk, int32_t P, int32_t RP,
                                  int32_t kernel) {
  return tgp_workspace_bytes_plan(N, D, M, S, nblk, P, RP, kernel, 0);
}

size_t tgp_workspace_bytes_plan(int32_t N, int32_t D, int32_t M, int32_t S, int32_t nblk, int32_t P, int32_t RP,
                                int32_t kernel, int32_t plan) {
  if (M > TGP_FUSED_MAX_M || kernel != TGP_KERNEL_SCALE_RBF)
    return big_workspace_doubles(N, D, M, S, nblk, P, RP, kernel, plan) * sizeof(double);
  Plan p;
  if (make_plan(p, N, D, M, S, nblk, P, RP, TGP_LIK_FLOW) != 0) return 0;
  size_t d = p.total + (size_t)(plan_alloc_blocks(N) - p.nblocks) * p.slab_len;   // slabs for whichever row kernel runs
  const size_t lik = lik_workspace_doubles(N, P, RP);
  if (lik > d) d = lik;
  // A training step whose flow stack does not fit a CU's LDS beside the row kernel's tiles (M > 112 with a 5 x 6 tanh flow)
  // runs on the general-M path (elbo_step_impl): room for that path too.  The program is not known here; an upper bound
  // of its stack slots decides: a block has at most 3 slots (SAL) or 1 + K (step-tanh, whose 4 K parameters are counted in P, or in
  // RP when they are per-row), so 3 nblk + (P + RP) / 4 bounds every mix of blocks (max(nblk + P/4, 3 nblk) did not --
  // SAL + step-tanh(K = 10) has 14 slots against a "bound" of 12)
  const int slots_ub = 3 * nblk + (P + RP) / 4;
  if (nblk > 0 && !rows_train_lds_fits(p, slots_ub)) {
    const size_t big = big_workspace_doubles(N, D, M, S, nblk, P, RP, kernel, plan);
    if (big > d) d = big;
  }
  return d * sizeof(double);
}

int tgp_elbo_step_f64(const tgp_model* model, const double* X, const double* Y, const double* rowp, double* out,
                      const tgp_grads* grads, double* mu, double* v, int32_t* status, void* workspace,
                      size_t workspace_bytes, void* stream) {
  return tgp_elbo_step_phases_f64(model, X, Y, rowp, out, grads, mu, v, status, workspace, workspace_bytes, 7u, stream);
}

static int elbo_step_impl(const tgp_model* model, const double* X, const double* Y, const double* rowp, double* out,
                          const tgp_grads* grads, double* mu, double* v, int32_t* status, void* workspace,
                          size_t workspace_bytes, uint32_t phases, const tgp_adam_args* adam, void* stream) {
  if (int rc = check_model(model, true)) return rc;
  if (!X) return -2;
  if (!Y) return -3;
  if (model->RP > 0 && !rowp) return -4;
  if (!out) return -5;
  if (!grads || !grads->Z || !grads->raw_ls || !grads->raw_os || !grads->m || !grads->Lam || !grads->log_var_noise)
    return -6;
  if (model->P > 0 && model->lik == TGP_LIK_FLOW && !grads->theta) return -6;
  if (model->RP > 0 && !grads->rowp) return -6;
  if ((mu == nullptr) != (v == nullptr)) return -7;
  if (!status) return -9;
  if (!workspace) return -10;
  const int nblk = model->lik == TGP_LIK_FLOW ? model->nblk : 0;
  const int P = model->lik == TGP_LIK_FLOW ? model->P : 0;
  const int RP = model->lik == TGP_LIK_FLOW ? model->RP : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* ws = static_cast<double*>(workspace);
  tgp_model md = *model;
  md.nblk = nblk; md.P = P; md.RP = RP;
  FlowProg fp;
  if (int rc = make_prog(&md, model->lik == TGP_LIK_FLOW, fp)) return rc;
  md.program = nullptr;  // kernels use the by-value copy
  AdamDev ad;
  if (adam != nullptr) {
    if (!adam->params || !adam->grads || !adam->exp_avg || !adam->exp_avg_sq || !adam->step_dev || adam->n < 1) return -12;
    // every gradient of the call must be a view of adam->grads
    const double* lo = adam->grads;
    const double* hi = adam->grads + adam->n;
    // ... START AND EXTENT: the update kernels index params / exp_avg / exp_avg_sq with the gradients' offsets, so a block
    // that ran past adam->n would be written past the caller's buffers
    const double* gp[7] = {grads->Z, grads->raw_ls, grads->raw_os, grads->m, grads->Lam, grads->log_var_noise, grads->theta};
    const size_t gn[7] = {(size_t)model->M * model->D, (size_t)model->D, 1, (size_t)model->M, (size_t)model->M * model->M, 1,
                          (size_t)P};
    for (int k = 0; k < 7; ++k)
      if (gp[k] != nullptr && gn[k] > 0 && (gp[k] < lo || gp[k] + gn[k] > hi)) return -12;
    ad.p = adam->params; ad.g = adam->grads; ad.m = adam->exp_avg; ad.v = adam->exp_avg_sq; ad.n = (long)adam->n;
    ad.lam_off = (long)(grads->Lam - adam->grads); ad.lam_n = (long)model->M * model->M;
    ad.lr = adam->lr; ad.b1 = adam->beta1; ad.b2 = adam->beta2; ad.eps = adam->eps;
    ad.ln_b1 = log(adam->beta1); ad.ln_b2 = log(adam->beta2); ad.sign = adam->maximize ? -1.0 : 1.0;
    ad.step_dev = adam->step_dev;
  }
  bool general = model->M > TGP_FUSED_MAX_M || model->kernel != TGP_KERNEL_SCALE_RBF;
  if (!general && fp.nslots > 0) {
    // the fused path keeps the flow stack of a row block in LDS beside its operand tiles; a program that does not fit even
    // with one node in flight (TGP_E_LDS until round 5: M > 112 with the 5 x 6 tanh flow) takes the general-M path
    Plan pl;
    if (int rc = make_plan(pl, model->N, model->D, model->M, model->S, nblk, P, RP, model->lik)) return rc;
    general = !rows_train_lds_fits(pl, fp.nslots);
  }
  if (general) {
    const bool lam_early = adam != nullptr && (phases & TGP_PHASE_BACKWARD) && ad.lam_n > 0;
    if (int rc = launch_big_step(md, fp, X, Y, rowp, out, *grads, mu, v, status, ws, workspace_bytes / sizeof(double), phases, st,
                                 lam_early ? &ad : nullptr))
      return rc;
    if (adam != nullptr)   // general-M path: the update is a launch of its own -- except Lam's entries (the bulk: M^2 of the
                           // M^2 + M (D + 1) + ... parameters), which the kernel that forms their gradient updated beside the K_MM chain
      return launch_adam_dev(adam->params, adam->grads, adam->exp_avg, adam->exp_avg_sq, adam->n, adam->lr, adam->beta1,
                             adam->beta2, adam->eps, 0.0, adam->step_dev, adam->maximize, st, 0, lam_early ? ad.lam_off : 0,
                             lam_early ? ad.lam_n : 0);
    return 0;
  }
  Plan p;
  if (int rc = make_plan(p, model->N, model->D, model->M, model->S, nblk, P, RP, model->lik)) return rc;
  p.nslots = fp.nslots;
  if (const int nw4 = choose_rows4(p, true, model->plan)) {
    if (int rc = make_plan(p, model->N, model->D, model->M, model->S, nblk, P, RP, model->lik, nw4)) return rc;
    p.nslots = fp.nslots;
  } else if (const int rw = rows_per_wave(p, fp, true, model->plan); rw != 16) {
    if (int rc = make_plan(p, model->N, model->D, model->M, model->S, nblk, P, RP, model->lik, 0, rw)) return rc;
    p.nslots = fp.nslots;
  }
  if (workspace_bytes < p.total * sizeof(double)) return TGP_E_WORKSPACE;
  if (phases & TGP_PHASE_PREPARE)
    if (int rc = launch_prepare(p, md, fp, ws, status, st)) return rc;
  if (phases & TGP_PHASE_ROWS)
    if (int rc = launch_rows(p, md, fp, X, Y, rowp, grads->rowp, mu, v, ws, true, st)) return rc;
  if (phases & TGP_PHASE_BACKWARD)
    if (int rc = launch_backward_mm(p, md, *grads, out, ws, status, st, adam != nullptr ? &ad : nullptr)) return rc;
  return 0;
}

int tgp_elbo_step_phases_f64(const tgp_model* model, const double* X, const double* Y, const double* rowp,
                             double* out, const tgp_grads* grads, double* mu, double* v, int32_t* status,
                             void* workspace, size_t workspace_bytes, uint32_t phases, void* stream) {
  return elbo_step_impl(model, X, Y, rowp, out, grads, mu, v, status, workspace, workspace_bytes, phases, nullptr, stream);
}

int tgp_elbo_step_adam_f64(const tgp_model* model, const double* X, const double* Y, const double* rowp, double* out,
                           const tgp_grads* grads, double* mu, double* v, int32_t* status, void* workspace,
                           size_t workspace_bytes, const tgp_adam_args* adam, void* stream) {
  if (adam == nullptr) return -12;
  const uint32_t all = TGP_PHASE_PREPARE | TGP_PHASE_ROWS | TGP_PHASE_BACKWARD;
  const uint32_t phases = adam->phases != 0 ? (adam->phases & all) : all;
  return elbo_step_impl(model, X, Y, rowp, out, grads, mu, v, status, workspace, workspace_bytes, phases,
                        (phases & TGP_PHASE_BACKWARD) ? adam : nullptr, stream);
}

int tgp_qf_moments_f64(const tgp_model* model, const double* X, double* mu, double* v, int32_t* status,
                       void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_model(model, false)) return rc;
  if (!X) return -2;
  if (!mu) return -3;
  if (!v) return -4;
  if (!status) return -5;
  if (!workspace) return -6;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* ws = static_cast<double*>(workspace);
  tgp_model md = *model;
  md.nblk = 0; md.P = 0; md.RP = 0; md.lik = TGP_LIK_GAUSS; md.program = nullptr;
  if (model->M > TGP_FUSED_MAX_M || model->kernel != TGP_KERNEL_SCALE_RBF) return launch_big_moments(md, X, mu, v, status, ws, workspace_bytes / sizeof(double), st);
  Plan p;
  if (int rc = make_plan(p, model->N, model->D, model->M, 1, 0, 0, 0, TGP_LIK_GAUSS)) return rc;
  if (const int nw4 = choose_rows4(p, false, model->plan))
    if (int rc = make_plan(p, model->N, model->D, model->M, 1, 0, 0, 0, TGP_LIK_GAUSS, nw4)) return rc;
  if (workspace_bytes < p.total * sizeof(double)) return TGP_E_WORKSPACE;
  FlowProg fp;
  fp.nblk = 0; fp.nslots = 0;
  if (int rc = launch_prepare(p, md, fp, ws, status, st)) return rc;
  return launch_rows(p, md, fp, X, nullptr, nullptr, nullptr, mu, v, ws, false, st);
}

int tgp_qf_moments_bwd_f64(const tgp_model* model, const double* X, const double* mu_bar, const double* v_bar,
                           const tgp_grads* grads, int32_t* status, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (int rc = check_model(model, false)) return rc;
  if (!X) return -2;
  if (!mu_bar) return -3;
  if (!v_bar) return -4;
  if (!grads || !grads->Z || !grads->raw_ls || !grads->raw_os || !grads->m || !grads->Lam || !grads->log_var_noise) return -5;
  if (!status) return -6;
  if (!workspace) return -7;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* ws = static_cast<double*>(workspace);
  tgp_model md = *model;
  md.nblk = 0; md.P = 0; md.RP = 0; md.S = 1; md.lik = TGP_LIK_ADJOINT; md.program = nullptr;
  md.scale = 1.0; md.kl_scale = 0.0;
  tgp_grads g = *grads;
  g.theta = nullptr; g.rowp = nullptr;
  FlowProg fp;
  fp.nblk = 0; fp.nslots = 0;
  const uint32_t all = TGP_PHASE_PREPARE | TGP_PHASE_ROWS | TGP_PHASE_BACKWARD;
  if (model->M > TGP_FUSED_MAX_M || model->kernel != TGP_KERNEL_SCALE_RBF) {
    // (the scalars of the step go to the header words the general-M plan reserves as well: its hdr is the first block)
    return launch_big_step(md, fp, X, mu_bar, v_bar, ws + H_OUT, g, nullptr, nullptr, status, ws, workspace_bytes / sizeof(double),
                           all, st);
  }
  Plan p;
  if (int rc = make_plan(p, model->N, model->D, model->M, 1, 0, 0, 0, TGP_LIK_ADJOINT)) return rc;
  if (const int nw4 = choose_rows4(p, true, model->plan))
    if (int rc = make_plan(p, model->N, model->D, model->M, 1, 0, 0, 0, TGP_LIK_ADJOINT, nw4)) return rc;
  if (workspace_bytes < p.total * sizeof(double)) return TGP_E_WORKSPACE;
  if (int rc = launch_prepare(p, md, fp, ws, status, st)) return rc;
  if (int rc = launch_rows(p, md, fp, X, mu_bar, v_bar, nullptr, nullptr, nullptr, ws, true, st)) return rc;
  return launch_backward_mm(p, md, g, ws + p.hdr + H_OUT, ws, status, st);
}

int tgp_kmm_f64(const double* Z, const double* raw_ls, const double* raw_os, int32_t M, int32_t D, double jitter,
                double* K, void* stream) {
  if (!Z) return -1;
  if (!raw_ls) return -2;
  if (!raw_os) return -3;
  if (M < 1) return -4;
  if (D < 1) return -5;
  if (!K) return -7;
  return launch_kmm(Z, raw_ls, raw_os, M, D, jitter, K, static_cast<hipStream_t>(stream));
}

int tgp_kernel_matrix_f64(int32_t kernel, const double* X1, int32_t N1, const double* X2, int32_t N2, int32_t D,
                          const double* raw_ls, const double* raw_os, double jitter, double* K, void* stream) {
  if (kernel != TGP_KERNEL_SCALE_RBF && kernel != TGP_KERNEL_SCALE_MATERN32) return -1;
  if (!X1) return -2;
  if (N1 < 1) return -3;
  if (X2 && N2 < 1) return -5;
  if (D < 1) return -6;
  if (!raw_ls) return -7;
  if (!raw_os) return -8;
  if (!K) return -10;
  return launch_kernel_matrix(kernel, X1, N1, X2, N2, D, raw_ls, raw_os, jitter, K, static_cast<hipStream_t>(stream));
}

int tgp_knm_f64(const double* X, const double* Z, const double* raw_ls, const double* raw_os, int32_t N, int32_t M,
                int32_t D, double* K, void* stream) {
  if (!X) return -1;
  if (!Z) return -2;
  if (!raw_ls) return -3;
  if (!raw_os) return -4;
  if (N < 1) return -5;
  if (M < 1) return -6;
  if (D < 1) return -7;
  if (!K) return -8;
  return launch_knm(X, Z, raw_ls, raw_os, N, M, D, K, static_cast<hipStream_t>(stream));
}

size_t tgp_ell_workspace_bytes(int32_t N, int32_t P, int32_t RP) {
  if (N < 1 || P < 0 || RP < 0) return 0;
  return tgp::lik_workspace_doubles(N, P, RP) * sizeof(double);
}

size_t tgp_cholesky_workspace_bytes(int32_t M) {
  if (M <= TGP_FUSED_MAX_M) return 0;
  return big_cholesky_workspace_doubles(M) * sizeof(double);
}

int tgp_cholesky_f64(const double* A, int32_t M, double* L, double* Linv, int32_t* status, void* workspace,
                     size_t workspace_bytes, void* stream) {
  if (!A) return -1;
  if (M < 1) return -2;
  if (M > TGP_BIG_MAX_M) return TGP_E_UNSUPPORTED;
  if (!L) return -3;
  if (!status) return -5;
  if (M > TGP_FUSED_MAX_M) {  // blocked multi-kernel factorisation of the general-M path; needs tgp_cholesky_workspace_bytes(M)
    if (!workspace) return -6;
    return launch_big_cholesky(A, M, L, Linv, status, static_cast<double*>(workspace), workspace_bytes / sizeof(double),
                               static_cast<hipStream_t>(stream));
  }
  return launch_cholesky(A, M, L, Linv, status, static_cast<hipStream_t>(stream));
}

int tgp_cholesky_bwd_f64(const double* L, const double* Linv, const double* L_bar, int32_t M, double* A_bar, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (!L) return -1;
  if (!Linv) return -2;
  if (!L_bar) return -3;
  if (M < 1) return -4;
  if (M > TGP_BIG_MAX_M) return TGP_E_UNSUPPORTED;
  if (!A_bar) return -5;
  if (!workspace) return -6;
  return launch_big_cholesky_bwd(L, Linv, L_bar, M, A_bar, static_cast<double*>(workspace), workspace_bytes / sizeof(double),
                                 static_cast<hipStream_t>(stream));
}

size_t tgp_cholesky_bwd_workspace_bytes(int32_t M) { return big_cholesky_workspace_doubles(M) * sizeof(double); }

int tgp_gemm_f64(int32_t trans_a, int32_t trans_b, int32_t tri, int32_t m, int32_t n, int32_t k, double alpha,
                 const double* A, int32_t lda, const double* B, int32_t ldb, double beta, double* C, int32_t ldc,
                 void* stream) {
  if (tri < 0 || tri > 31) return -3;
  if (m < 1 || m % 128) return -4;
  if (n < 1 || n % 128) return -5;
  if (k < 1 || k % 16) return -6;
  if (!A) return -8;
  if (!B) return -10;
  if (!C) return -13;
  return launch_gemm_plain(trans_a != 0, trans_b != 0, tri, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc,
                           static_cast<hipStream_t>(stream));
}

int tgp_kl_whitened_f64(const double* m, const double* Lam, int32_t M, double* out, double* g_m, double* g_Lam,
                        void* stream) {
  if (!m) return -1;
  if (!Lam) return -2;
  if (M < 1) return -3;
  if (!out) return -4;
  return launch_kl(m, Lam, M, out, g_m, g_Lam, static_cast<hipStream_t>(stream));
}

int tgp_ell_gauss_f64(const double* Y, const double* mu, const double* v, int32_t N, const double* log_var_noise,
                      double scale, double* out, double* g_mu, double* g_v, void* workspace, size_t workspace_bytes,
                      void* stream) {
  if (!Y) return -1;
  if (!mu) return -2;
  if (!v) return -3;
  if (N < 1) return -4;
  if (!log_var_noise) return -5;
  if (!out) return -7;
  if (!workspace) return -10;
  if (workspace_bytes < lik_workspace_doubles(N, 0, 0) * sizeof(double)) return TGP_E_WORKSPACE;
  return launch_ell_gauss(Y, mu, v, N, log_var_noise, scale, out, g_mu, g_v, static_cast<double*>(workspace),
                          static_cast<hipStream_t>(stream));
}

int tgp_ell_flow_f64(const tgp_model* model, const double* Y, const double* mu, const double* v, const double* rowp,
                     double* out, double* g_mu, double* g_v, double* g_theta, double* g_rowp, void* workspace,
                     size_t workspace_bytes, void* stream) {
  if (!model || model->N < 1 || model->S < 1 || !model->xs || !model->wn || !model->log_var_noise) return -1;
  if (model->nblk > 0 && !model->program) return -1;
  if (model->P > 0 && !model->theta) return -1;
  if (!Y) return -2;
  if (!mu) return -3;
  if (!v) return -4;
  if (model->RP > 0 && !rowp) return -5;
  if (!out) return -6;
  if (!workspace) return -11;
  if (workspace_bytes < lik_workspace_doubles(model->N, model->P, model->RP) * sizeof(double)) return TGP_E_WORKSPACE;
  FlowProg fp;
  if (int rc = make_prog(model, true, fp)) return rc;
  tgp_model md = *model;
  md.program = nullptr;
  return launch_ell_flow(md, fp, Y, mu, v, rowp, out, g_mu, g_v, g_theta, g_rowp, static_cast<double*>(workspace),
                         static_cast<hipStream_t>(stream));
}

int tgp_flow_eval_f64(const tgp_model* model, const double* f, int32_t S, int32_t N, const double* rowp, double* G,
                      double* dG, double* logdG, void* stream) {
  if (!model) return -1;
  if (model->nblk > 0 && !model->program) return -1;
  if (model->P > 0 && !model->theta) return -1;
  if (!f) return -2;
  if (S < 1) return -3;
  if (N < 1) return -4;
  if (model->RP > 0 && !rowp) return -5;
  FlowProg fp;
  if (int rc = make_prog(model, true, fp)) return rc;
  tgp_model md = *model;
  md.program = nullptr;
  return launch_flow_eval(md, fp, f, S, N, rowp, G, dG, logdG, static_cast<hipStream_t>(stream));
}

size_t tgp_flow_logdet_workspace_bytes(int32_t S, int32_t N) {
  return (((size_t)S * (size_t)N + 1023) / 1024 + 16) * sizeof(double);
}

int tgp_flow_logdet_f64(const tgp_model* model, const double* f, int32_t S, int32_t N, const double* rowp, double* G,
                        double* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!model) return -1;
  if (model->nblk > 0 && !model->program) return -1;
  if (model->P > 0 && !model->theta) return -1;
  if (!f) return -2;
  if (S < 1) return -3;
  if (N < 1) return -4;
  if (model->RP > 0 && !rowp) return -5;
  if (!out) return -7;
  if (!workspace) return -8;
  if (workspace_bytes < tgp_flow_logdet_workspace_bytes(S, N)) return TGP_E_WORKSPACE;
  FlowProg fp;
  if (int rc = make_prog(model, true, fp)) return rc;
  tgp_model md = *model;
  md.program = nullptr;
  return launch_flow_eval(md, fp, f, S, N, rowp, G, nullptr, nullptr, static_cast<hipStream_t>(stream), out,
                          static_cast<double*>(workspace));
}

int tgp_predict_f64(const tgp_model* model, const double* mu, const double* v, const double* rowp, const double* Y,
                    double Y_std, double* m1, double* m2, double* logp, void* stream) {
  if (!model || model->N < 1 || !model->log_var_noise) return -1;
  if (model->lik == TGP_LIK_FLOW && (model->S < 1 || !model->xs || !model->wn)) return -1;
  if (!mu) return -2;
  if (!v) return -3;
  if (model->lik == TGP_LIK_FLOW && model->RP > 0 && !rowp) return -4;
  FlowProg fp;
  if (int rc = make_prog(model, model->lik == TGP_LIK_FLOW, fp)) return rc;
  tgp_model md = *model;
  md.program = nullptr;
  return launch_predict(md, fp, mu, v, rowp, Y, Y_std, m1, m2, logp, static_cast<hipStream_t>(stream));
}

int tgp_kmeans_assign_f64(const double* X, int32_t N, int32_t D, const double* C, int32_t K, int32_t* labels, double* mind2,
                          void* stream) {
  if (!X) return -1;
  if (N < 1) return -2;
  if (D < 1 || D > 16) return -3;
  if (!C) return -4;
  if (K < 1) return -5;
  if (!labels) return -6;
  return launch_kmeans_assign(X, N, D, C, K, labels, mind2, static_cast<hipStream_t>(stream));
}

int tgp_kmeans_segsum_f64(const double* X, int32_t D, const int64_t* order, const int64_t* offs, int32_t K, double* sums,
                          void* stream) {
  if (!X) return -1;
  if (D < 1 || D > 16) return -2;
  if (!order) return -3;
  if (!offs) return -4;
  if (K < 1) return -5;
  if (!sums) return -6;
  return launch_kmeans_segsum(X, D, order, offs, K, sums, static_cast<hipStream_t>(stream));
}

int tgp_kmeans_pp_f64(const double* X, int32_t N, int32_t D, const int64_t* cand, int32_t T, const double* closest,
                      double* out, void* stream) {
  if (!X) return -1;
  if (N < 1) return -2;
  if (D < 1 || D > 16) return -3;
  if (!cand) return -4;
  if (T < 1 || T > 16) return -5;
  if (!out) return -7;
  return launch_kmeans_pp(X, N, D, cand, T, closest, out, static_cast<hipStream_t>(stream));
}

int tgp_adam_f64(double* params, const double* grads, double* exp_avg, double* exp_avg_sq, int64_t n, double lr,
                 double beta1, double beta2, double eps, double weight_decay, int32_t step, int32_t maximize,
                 void* stream) {
  if (!params) return -1;
  if (!grads) return -2;
  if (!exp_avg) return -3;
  if (!exp_avg_sq) return -4;
  if (n < 1) return -5;
  if (step < 1) return -11;
  return launch_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, maximize,
                     static_cast<hipStream_t>(stream));
}

int tgp_adam_dev_f64(double* params, const double* grads, double* exp_avg, double* exp_avg_sq, int64_t n, double lr,
                     double beta1, double beta2, double eps, double weight_decay, int32_t* step_dev, int32_t maximize,
                     void* stream) {
  if (!params) return -1;
  if (!grads) return -2;
  if (!exp_avg) return -3;
  if (!exp_avg_sq) return -4;
  if (n < 1) return -5;
  if (!step_dev) return -11;
  return launch_adam_dev(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_dev, maximize,
                         static_cast<hipStream_t>(stream));
}

int tgp_gather_rows_f64(const double* X, const double* Y, int32_t N, int32_t D, const int32_t* index, int32_t* cursor_dev,
                        int32_t offset, int32_t nrows, int32_t advance, int32_t wrap, double* Xb, double* Yb, void* stream) {
  if (!X) return -1;
  if (!Y) return -2;
  if (N < 1 || D < 1) return -3;
  if (!cursor_dev) return -6;
  if (offset < 0 || nrows < 1 || nrows > N) return -8;
  if (advance < 0 || wrap < 1) return -9;
  if (!Xb) return -11;
  if (!Yb) return -12;
  return launch_gather_rows(X, Y, N, D, index, cursor_dev, offset, nrows, advance, wrap, Xb, Yb,
                            static_cast<hipStream_t>(stream));
}

int tgp_adam_dev_groups_f64(double* params, const double* grads, double* exp_avg, double* exp_avg_sq, int64_t n,
                            double lr, double beta1, double beta2, double eps, int64_t n_plain, double weight_decay_tail,
                            int32_t* step_dev, int32_t maximize, void* stream) {
  if (!params) return -1;
  if (!grads) return -2;
  if (!exp_avg) return -3;
  if (!exp_avg_sq) return -4;
  if (n < 1) return -5;
  if (n_plain < 0 || n_plain > n) return -10;
  if (!step_dev) return -12;
  return launch_adam_dev(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay_tail, step_dev, maximize,
                         static_cast<hipStream_t>(stream), n_plain);
}

size_t tgp_mlp_workspace_bytes(const tgp_mlp* mlp) {
  if (!mlp) return 0;
  return mlp_workspace_doubles(mlp->N, mlp->D, mlp->H, mlp->L, mlp->nnets) * sizeof(double);
}

int tgp_mlp_forward_f64(const tgp_mlp* mlp, const double* X, const double* W, const int32_t* step_dev, double* out,
                        void* stream) {
  if (!mlp) return -1;
  if (!X) return -2;
  if (!W) return -3;
  if (!out) return -5;
  return launch_mlp_forward(*mlp, X, W, step_dev, out, static_cast<hipStream_t>(stream));
}

int tgp_mlp_backward_f64(const tgp_mlp* mlp, const double* X, const double* W, const int32_t* step_dev,
                         const double* g_out, double* g_W, void* workspace, size_t workspace_bytes, void* stream) {
  if (!mlp) return -1;
  if (!X) return -2;
  if (!W) return -3;
  if (!g_out) return -5;
  if (!g_W) return -6;
  if (!workspace) return -7;
  return launch_mlp_backward(*mlp, X, W, step_dev, g_out, g_W, static_cast<double*>(workspace),
                             workspace_bytes / sizeof(double), static_cast<hipStream_t>(stream));
}

int tgp_mlp_backward_adam_f64(const tgp_mlp* mlp, const double* X, double* W, const int32_t* step_dev, const double* g_out,
                              double* g_W, void* workspace, size_t workspace_bytes, const tgp_adam_args* adam,
                              double weight_decay, void* stream) {
  if (!mlp) return -1;
  if (!X) return -2;
  if (!W) return -3;
  if (!g_out) return -5;
  if (!g_W) return -6;
  if (!workspace) return -7;
  if (!adam || adam->params != W || adam->grads != g_W || !adam->exp_avg || !adam->exp_avg_sq || !adam->step_dev) return -9;
  // the optimiser state must cover every weight the kernel updates (nnets x weights per net)
  {
    size_t per = (size_t)mlp->H * mlp->D + mlp->H;
    for (int l = 1; l < mlp->L; ++l) per += (size_t)mlp->H * mlp->H + mlp->H;
    per += (size_t)mlp->H + 1;
    if (adam->n < 1 || (size_t)adam->n != per * (size_t)mlp->nnets) return -9;
  }
  AdamDev ad;
  ad.p = adam->params; ad.g = adam->grads; ad.m = adam->exp_avg; ad.v = adam->exp_avg_sq; ad.n = (long)adam->n;
  ad.lr = adam->lr; ad.b1 = adam->beta1; ad.b2 = adam->beta2; ad.eps = adam->eps;
  ad.ln_b1 = log(adam->beta1); ad.ln_b2 = log(adam->beta2); ad.sign = adam->maximize ? -1.0 : 1.0;
  ad.step_dev = adam->step_dev;
  return launch_mlp_backward(*mlp, X, W, step_dev, g_out, g_W, static_cast<double*>(workspace),
                             workspace_bytes / sizeof(double), static_cast<hipStream_t>(stream), &ad, weight_decay);
}

int tgp_comm_load(const char* rccl_path) { return comm_load(rccl_path); }

int tgp_comm_unique_id(void* id128) {
  if (!id128) return -1;
  return comm_unique_id(id128);
}

int tgp_comm_init(const void* id128, int32_t nranks, int32_t rank, void** comm) {
  if (!id128) return -1;
  if (nranks < 1) return -2;
  if (rank < 0 || rank >= nranks) return -3;
  if (!comm) return -4;
  return comm_init(id128, nranks, rank, comm);
}

int tgp_allreduce_f64(void* comm, double* buf, int64_t n, void* stream) {
  if (!comm) return -1;
  if (!buf) return -2;
  if (n < 1) return -3;
  return comm_allreduce(comm, buf, n, static_cast<hipStream_t>(stream));
}

int tgp_comm_destroy(void* comm) {
  if (!comm) return -1;
  return comm_destroy(comm);
}

}  // extern "C"
