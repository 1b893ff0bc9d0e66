/*
 * tgp_hip.h -- C ABI of libtgp_hip.so: the MI355X (gfx950) implementation of the TGP
 * sparse-variational ELBO hot path (jmaronas/TGP.pytorch, `sparse_MF_SP.ELBO` and below).
 *
 * The reference has no FFI of its own (it is pure Python on ATen); the boundary it offers is the
 * model-class API.  This header is the plain-C surface that sits directly underneath that API:
 * every entry point names the reference call site(s) it replaces (file:line relative to
 * /root/reference/code).  The Python mirror of the reference classes (tgp/pytorch_amd) binds these
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - all matrices row-major, contiguous, float64 (`_f64`; the reference's main.py runs in float64,
 *     dsp/config.py:37-46);
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensor.data_ptr()), with ONE exception:
 *     tgp_model.program (a few int32 describing the flow) is a HOST array, read during the call;
 *     outputs and the workspace are pre-allocated by the caller, nothing is allocated inside;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous and stream-ordered, safe
 *     to capture into a hipGraph, and re-entrant when callers use distinct streams + workspaces + status buffers
 *     (status[4..7] are hand-off words of the launches in flight: two concurrent calls that shared them would pass each
 *     other's waits early);
 *     the general-M path (M > 128 or a non-RBF kernel) forks independent phases onto helper streams that the
 *     library creates at a host thread's first such call and joins before the call returns (event record / wait:
 *     valid under capture of `stream`, where they become parallel branches of the graph) -- make a thread's first
 *     call outside a capture; the helper streams and events are per host thread, so threads never share them;
 *   - return value: 0 = launched; <0 = -(index of the offending argument) or TGP_E_*;
 *     numerical failure is reported ASYNCHRONOUSLY through `status` (device int32[8], ZERO before its first use):
 *       status[0] = LAPACK-style info of the Cholesky of K_MM (0 ok, j>0 = pivot j not positive;
 *                   TGP_STATUS_SYNC_TIMEOUT: a workgroup of the prepare or of the M x M backward launch gave up waiting for a hand-off word --
 *                   the hand-off words were not zero, or its producers never became resident; results invalid),
 *       status[1] = 1 if K_MM contained a NaN (the reference raises NanError, dsp/utils.py:241-254),
 *       status[2] = level of the on-device jitter ladder that succeeded (tgp_model.jitter_ladder),
 *       status[3] = STICKY count of expired hand-off waits: incremented by the launch whose wait expired, never cleared by the
 *                   library (status[0] is rewritten by the next call's prepare launch, so inside a replayed graph of several
 *                   steps a timeout shows only here); a fused-update call (tgp_elbo_step_adam_f64) whose backward launch sees
 *                   the timeout skips the update outside Lam and the step counter and returns NaN scalars; the caller
 *                   zeroes the word after it has handled the event,
 *       status[4..7] = hand-off words between workgroups of ONE launch (M <= 128: the tile blocks of the prepare
 *                   launch count themselves in status[4] once their tile of K_MM is in global memory, status[5] counts
 *                   the blocks that have left; the M x M backward launch counts its finished column blocks in bits 16-23
 *                   and its finished row-block halves in bits 24-31 of status[6], and the blocks that have left in status[7];
 *                   M > 128: the workgroups that produce the diagonal block a factorising workgroup of the same launch
 *                   waits for -- the first block of K_MM, then the trailing update's tiles -- count themselves in status[4]);
 *                   the library leaves them zero at the end of every
 *                   call, the caller must not touch them while a call is in flight.  A caller built against the
 *                   int32[4] status of ABI versions <= 100 must grow the buffer: check tgp_version() >= 101,
 *     so the host can replay with the reference's jitter ladder (dsp/utils.py:256-269) without a
 *     device sync per step.  No exception crosses the ABI.
 */
#ifndef TGP_HIP_H
#define TGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TGP_VERSION 103
#define TGP_FUSED_MAX_M 128 /* up to here the whole step is 4 fused kernel launches (operators resident in LDS/registers) */
#define TGP_BIG_MAX_M 4096  /* above: chunked path built on a tiled float64 MFMA GEMM                             */

/* error codes (negative return values below -64 are generic) */
#define TGP_E_UNSUPPORTED (-100) /* shape outside this build's limits (M > 4096, D > 16, ...) */
#define TGP_E_WORKSPACE (-101)   /* workspace too small                                        */
#define TGP_E_LDS (-102)         /* flow program too large for one CU's LDS                    */
#define TGP_E_LAUNCH (-103)      /* hipLaunchKernel failed; see tgp_last_error()               */
#define TGP_E_COMM (-104)        /* RCCL not loaded / an RCCL call failed; see tgp_last_error() */
/* value of status[0] (not a return value): a hand-off wait inside a launch expired (see Conventions) */
#define TGP_STATUS_SYNC_TIMEOUT (-77)

/* ---- flow program (models/flow.py CompositeFlow.forward :155-158) -------------------------------
 * A flow is a sequence of `nblk` blocks; block b is four int32: {kind, K, poff, flags}.
 *   kind   TGP_FLOW_AFFINE   g = a f + b                                   (flow.py:330-340)
 *          TGP_FLOW_SAL      g = sinh(b asinh(f) - a), asinh = log(f+sqrt(f^2+1)) (flow.py:904-905,936-977)
 *          TGP_FLOW_STEPTANH g = [f +] sum_k a_k + sp(b_k) tanh((f-c_k)/sp(d_k)) (flow.py:1096-1103,760-771)
 *   K      number of tanh steps (STEPTANH only)
 *   poff   offset of the block's parameters in `theta` (shared scalars: AFFINE {a,b}, SAL {a,b},
 *          STEPTANH {a_k,b_k,c_k,d_k}_k) or first column in `rowp` when TGP_FLAG_PER_ROW is set
 *   flags  TGP_FLAG_RESTRICT  set_restrictions=True: softplus on AFFINE.a / SAL.b
 *          TGP_FLAG_ADD_F0    add_init_f0=True
 *          TGP_FLAG_PER_ROW   input-dependent parameters, one value per data row (flow.py:949-965)
 */
#define TGP_FLOW_AFFINE 0
#define TGP_FLOW_SAL 1
#define TGP_FLOW_STEPTANH 2
#define TGP_FLAG_RESTRICT 1
#define TGP_FLAG_ADD_F0 2
#define TGP_FLAG_PER_ROW 4

/* likelihood selector */
#define TGP_LIK_GAUSS 0 /* GaussianLinearMean.expected_log_prob, likelihoods/GaussianLinearMean.py:60-87       */
#define TGP_LIK_FLOW 1  /* GaussianNonLinearMean.expected_log_prob, likelihoods/GaussianNonLinearMean.py:64-150 */
#define TGP_LIK_ADJOINT 2 /* internal to tgp_qf_moments_bwd_f64: no likelihood, the adjoints of mu, v are inputs */

/* covariance function: instance_kernel(name, ...) of models/utils_models.py:145-204 (gpytorch kernels, ARD, softplus
 * parameters).  RBF: s2 exp(-r^2/2);  MATERN32: s2 (1 + sqrt3 r) exp(-sqrt3 r), r = sqrt(max(r^2, 1e-30)) as gpytorch's
 * covar_dist clamps it.  MATERN32 always runs on the general (tiled-GEMM) path. */
#define TGP_KERNEL_SCALE_RBF 0      /* 'scale_rbf'      utils_models.py:188-193 (what main.py:229 uses) */
#define TGP_KERNEL_SCALE_MATERN32 1 /* 'scale_matern32' utils_models.py:199-204                          */

/* tgp_model.plan: which of the library's equivalent implementations a call runs (results agree to rounding; speed differs).
 * The library's own choice (0) depends on the shape only.  A workspace sized by tgp_workspace_bytes[_kernel] holds every
 * row-kernel variant; a forced chunk size needs tgp_workspace_bytes_plan.
 *   bits 0-3  row kernel of the fused path (M <= 128):
 *     TGP_PLAN_ROWS_AUTO  library's choice        TGP_PLAN_ROWS_K16  k_rows, 16 data rows per wave
 *     TGP_PLAN_ROWS_K     k_rows, rows per wave by the library's rule (16 or 10): never k_rows4
 *     TGP_PLAN_ROWS4_NW4 / TGP_PLAN_ROWS4_NW8  k_rows4 with 4 / 8 waves per workgroup (ignored where its LDS plan or the
 *                          workspace's slab count does not allow it)
 *   bit 4     TGP_PLAN_NO_CHUNK_OVERLAP  general-M path: one set of chunk buffers, forward and backward of the chunks in line
 *   bits 8-23 TGP_PLAN_CHUNK_ROWS(n)     general-M path: row chunks of at most n rows (rounded up to 128; 0 = 16 384) */
#define TGP_PLAN_ROWS_AUTO 0
#define TGP_PLAN_ROWS_K16 1
#define TGP_PLAN_ROWS_K 2
#define TGP_PLAN_ROWS4_NW4 3
#define TGP_PLAN_ROWS4_NW8 4
#define TGP_PLAN_ROWS_MASK 15
#define TGP_PLAN_NO_CHUNK_OVERLAP 16
#define TGP_PLAN_CHUNK_ROWS(n) (((((n) + 127) / 128) & 0xffff) << 8)
#define TGP_PLAN_CHUNK_OF(plan) ((((plan) >> 8) & 0xffff) * 128)

typedef struct tgp_model {
  int32_t N;       /* rows in this call (this rank's shard of the minibatch)            */
  int32_t D;       /* input dimension (<= 16)                                           */
  int32_t M;       /* inducing points: <= 128 fused path, <= TGP_BIG_MAX_M tiled-GEMM path */
  int32_t S;       /* quadrature nodes (TGP_LIK_FLOW)                                   */
  int32_t nblk;    /* flow blocks                                                       */
  int32_t P;       /* shared flow scalars in theta                                      */
  int32_t RP;      /* per-row flow parameter columns in rowp                            */
  int32_t lik;     /* TGP_LIK_*                                                         */
  int32_t kernel;  /* TGP_KERNEL_*                                                      */
  int32_t plan;    /* kernel-selection overrides of THIS call, TGP_PLAN_* below; 0 = automatic (what every caller but
                      an A/B measurement or a test of a specific kernel passes).  Was `reserved0` up to ABI 102.  */
  double scale;    /* N_total / MB_global: sparse_MF_SP.ELL, models/sparse_MF_SP.py:623-626 */
  double jitter;   /* added to diag(K_MM) before the Cholesky (0 unless retrying)       */
  double kl_scale; /* weight of the KL gradient in this call: 1/world_size so that an
                      all-reduce(sum) over row shards counts it once                    */
  double jitter_ladder; /* > 0: psd_safe_cholesky's retry ladder (dsp/utils.py:256-269) runs ON THE DEVICE -- a failed
                      factorisation is repeated with jitter_ladder * 10^i, i = 0..2, added to diag(K_MM); status[2] = the
                      level that succeeded (1..3) or 0.  0: no retry, status[0] reports the pivot (host ladder:
                      ops.elbo_step_safe).  Fused path: inside k_prep_a; general-M path: one extra launch that returns
                      at once unless the blocked factorisation failed (then a single-workgroup refactorisation).  */
  /* parameters (reference nn.Parameter names in brackets) */
  const double* Z;             /* (M,D)   [Z]                                               */
  const double* raw_ls;        /* (D)     [covariance_function.base_kernel.raw_lengthscale] */
  const double* raw_os;        /* (1)     [covariance_function.raw_outputscale]             */
  const double* m;             /* (M)     [q_U.variational_mean]                            */
  const double* Lam;           /* (M,M)   [q_U.chol_variational_covar] dense, tril at use   */
  const double* log_var_noise; /* (1)     [likelihood.log_var_noise]                        */
  const double* theta;         /* (P)     [G_matrix.0.flow_arr.*] or NULL                   */
  const int32_t* program;      /* (nblk,4) HOST array (copied into the kernel arguments), NULL if none; nblk <= 64 */
  const double* xs;            /* (S) Gauss-Hermite nodes                                   */
  const double* wn;            /* (S) weights / sqrt(pi)                                    */
} tgp_model;

typedef struct tgp_grads {
  double* Z;
  double* raw_ls;
  double* raw_os;
  double* m;
  double* Lam;
  double* log_var_noise;
  double* theta; /* (P) or NULL     */
  double* rowp;  /* (N,RP) or NULL  */
} tgp_grads;

int tgp_version(void);
const char* tgp_last_error(void);
/* sha256 (first 16 hex digits) of the kernel sources this library was compiled from (csrc Makefile SRC_HASH): the
   Python binding recomputes it from the tree and refuses a stale binary. */
const char* tgp_source_hash(void);

/* Bytes of workspace the calls below need for a problem of this shape (training is the maximum). */
size_t tgp_workspace_bytes(int32_t N, int32_t D, int32_t M, int32_t S, int32_t nblk, int32_t P, int32_t RP);
/* Same for a given covariance function (tgp_workspace_bytes == TGP_KERNEL_SCALE_RBF). */
size_t tgp_workspace_bytes_kernel(int32_t N, int32_t D, int32_t M, int32_t S, int32_t nblk, int32_t P, int32_t RP,
                                  int32_t kernel);
/* Same for a call that passes tgp_model.plan = `plan` (a forced chunk size changes the general-M path's buffers). */
size_t tgp_workspace_bytes_plan(int32_t N, int32_t D, int32_t M, int32_t S, int32_t nblk, int32_t P, int32_t RP,
                                int32_t kernel, int32_t plan);

/* One fused ELBO evaluation with gradients: replaces sparse_MF_SP.ELBO (models/sparse_MF_SP.py:552-598)
 * + loss.backward() (trainers/trainer_base.py:341) for one minibatch shard.
 *   X (N,D), Y (N), rowp (N,RP) or NULL
 *   out[0..3] = {ELBO_shard = ELL_shard - KL, ELL_shard, KL, 0}; gradients are d(ELL_shard - kl_scale*KL).
 *   mu, v: optional per-row q(f) moments (NULL to skip). */
int tgp_elbo_step_f64(const tgp_model* model, const double* X, const double* Y, const double* rowp, double* out,
                      const tgp_grads* grads, double* mu, double* v, int32_t* status, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Same call restricted to some of its phases (profiling / roofline measurement: bench.py brackets one phase with
 * HIP events).  phases: bit0 = M x M prepare (K_MM, Cholesky, L^-1, KL), bit1 = fused row kernel,
 * bit2 = slab reduction + M x M adjoint + gradient assembly.  Phases must have run in order at least once on the
 * same workspace; tgp_elbo_step_f64 == phases 7. */
#define TGP_PHASE_PREPARE 1u
#define TGP_PHASE_ROWS 2u
#define TGP_PHASE_BACKWARD 4u
int tgp_elbo_step_phases_f64(const tgp_model* model, const double* X, const double* Y, const double* rowp,
                             double* out, const tgp_grads* grads, double* mu, double* v, int32_t* status,
                             void* workspace, size_t workspace_bytes, uint32_t phases, void* stream);

/* The whole training step of one rank in one call: tgp_elbo_step_f64 followed by tgp_adam_dev_f64 over a flat parameter
 * buffer (trainers/trainer_base.py:337-342: ELBO, backward, optimizer.step()).  `adam` describes the flat buffers; the
 * pointers of `grads` must point INTO adam->grads (the engine's layout: every gradient a view of one buffer), no weight
 * decay.  On the fused path (M <= 128) the update is applied inside the last two backward launches (the q(u) factor's
 * 10^4 entries by passenger workgroups beside the K_MM adjoint, the rest where the gradients are assembled): one launch
 * and one dependent pass over the buffers less than the two calls.  Same arithmetic, same results.  Single rank only: a
 * data-parallel step has its all-reduce between the gradients and the update and keeps the two calls. */
typedef struct tgp_adam_args {
  double* params;      /* (n) */
  double* grads;       /* (n) written by this call */
  double* exp_avg;     /* (n) */
  double* exp_avg_sq;  /* (n) */
  int64_t n;
  double lr, beta1, beta2, eps;
  int32_t* step_dev;   /* device int32[2] {step, ticket}, as tgp_adam_dev_f64 */
  int32_t maximize;    /* != 0: ascend (the gradients are of +ELBO) */
  uint32_t phases;     /* 0: the whole step; else a TGP_PHASE_* mask as in tgp_elbo_step_phases_f64 -- the update is applied
                          with TGP_PHASE_BACKWARD (an engine that runs the phases apart, e.g. the two-stream ID_TGP step) */
} tgp_adam_args;
int tgp_elbo_step_adam_f64(const tgp_model* model, const double* X, const double* Y, const double* rowp, double* out,
                           const tgp_grads* grads, double* mu, double* v, int32_t* status, void* workspace,
                           size_t workspace_bytes, const tgp_adam_args* adam, void* stream);

/* q(f) marginals only: sparse_MF_SP.marginal_variational_qf_parameters (models/sparse_MF_SP.py:274-396,
 * whitened, diagonal=True).  mu, v: (N). */
int tgp_qf_moments_f64(const tgp_model* model, const double* X, double* mu, double* v, int32_t* status,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Adjoint of tgp_cholesky_f64 (what autograd replays for torch.cholesky inside psd_safe_cholesky, dsp/utils.py:239, when a
 * caller differentiates through the factor outside ELBO()): given L, Linv = L^-1 (both as tgp_cholesky_f64 returns them)
 * and L_bar (M x M; its part on and below the diagonal counts) it writes the SYMMETRIC
 *   A_bar = 1/2 L^-T (Phi(L^T L_bar) + Phi(L^T L_bar)^T) L^-1,   Phi = lower triangle with the diagonal halved
 * -- torch's cholesky backward.  Any M <= TGP_BIG_MAX_M: operands are padded to a multiple of 128 and run through the
 * three products of the general-M backward chain; workspace tgp_cholesky_bwd_workspace_bytes(M). */
size_t tgp_cholesky_bwd_workspace_bytes(int32_t M);
int tgp_cholesky_bwd_f64(const double* L, const double* Linv, const double* L_bar, int32_t M, double* A_bar, void* workspace,
                         size_t workspace_bytes, void* stream);

/* Adjoint of tgp_qf_moments_f64 (what autograd replays for models/sparse_MF_SP.py:274-396 when a caller differentiates
 * the q(f) marginals outside ELBO(): predictive moments with respect to the inducing points, hyper-parameters or q(u)).
 * Given mu_bar, v_bar (N) it writes d(sum_n mu_bar_n mu_n + v_bar_n v_n)/d{Z, raw_ls, raw_os, m, Lam} into `grads`
 * (grads->log_var_noise receives 0; theta / rowp are not touched).  Same kernels as the training step -- the row
 * kernel takes the adjoints instead of forming them from a likelihood, the M x M chain runs with KL weight 0 -- on
 * either path (fused for M <= 128, general-M above); workspace as tgp_workspace_bytes for S = 1, no flow. */
int tgp_qf_moments_bwd_f64(const tgp_model* model, const double* X, const double* mu_bar, const double* v_bar,
                           const tgp_grads* grads, int32_t* status, void* workspace, size_t workspace_bytes,
                           void* stream);

/* K_MM assembly: gpytorch ScaleKernel(RBFKernel(ard)) as built by instance_kernel('scale_rbf'),
 * models/utils_models.py:188-193, called at models/sparse_MF_SP.py:316.  K (M,M). */
int tgp_kmm_f64(const double* Z, const double* raw_ls, const double* raw_os, int32_t M, int32_t D, double jitter,
                double* K, void* stream);
/* Same with the covariance function selected (TGP_KERNEL_*); X2 == NULL: K(X1, X1) + jitter I (N2 ignored),
 * else K(X1, X2) of shape (N1, N2). */
int tgp_kernel_matrix_f64(int32_t kernel, const double* X1, int32_t N1, const double* X2, int32_t N2, int32_t D,
                          const double* raw_ls, const double* raw_os, double jitter, double* K, void* stream);

/* K_NM assembly (models/sparse_MF_SP.py:319); K (N,M).  Diagnostic/next-row use: the training path
 * never materialises K_NM. */
int tgp_knm_f64(const double* X, const double* Z, const double* raw_ls, const double* raw_os, int32_t N, int32_t M,
                int32_t D, double* K, void* stream);

/* Lower Cholesky with LAPACK-style info: torch.cholesky inside psd_safe_cholesky, dsp/utils.py:239.
 * A (M,M) symmetric, L (M,M) lower (strict upper zeroed); Linv (M,M) = L^-1 or NULL. */
int tgp_cholesky_f64(const double* A, int32_t M, double* L, double* Linv, int32_t* status, void* workspace,
                     size_t workspace_bytes, void* stream);
/* Workspace of tgp_cholesky_f64: 0 for M <= 128 (factorised in one CU's LDS), the blocked multi-kernel path above. */
size_t tgp_cholesky_workspace_bytes(int32_t M);

/* Dense float64 contraction on the matrix cores, the building block of the M > 128 path (the reference's
 * torch.bmm / triangular_solve calls at models/sparse_MF_SP.py:354,376-382 on (M,M)x(M,N) operands):
 *   C = alpha * op(A) op(B) + beta * C, row-major, op = transpose when trans_* != 0.
 * m, n multiples of 128, k a multiple of 16, lda/ldb even, A and B 16-byte aligned (callers pad).  `tri` declares triangular operands so that the k range
 * is trimmed per output tile: 1 op(A) lower, 2 op(A) upper, 4 op(B) lower, 8 op(B) upper, 16 compute only the
 * lower block triangle of C (flags OR-ed; 0 = general). */
int tgp_gemm_f64(int32_t trans_a, int32_t trans_b, int32_t tri, int32_t m, int32_t n, int32_t k, double alpha,
                 const double* A, int32_t lda, const double* B, int32_t ldb, double beta, double* C, int32_t ldc,
                 void* stream);

/* Whitened KL and its gradients: sparse_MF_SP.KLD, models/sparse_MF_SP.py:406-431. out[0] = KL. */
int tgp_kl_whitened_f64(const double* m, const double* Lam, int32_t M, double* out, double* g_m, double* g_Lam,
                        void* stream);

/* Workspace of the two stand-alone likelihood entries below (per-workgroup partial sums; the workgroup count depends on
 * N: the flow kernel gives small problems 16 lanes per row instead of 4). */
size_t tgp_ell_workspace_bytes(int32_t N, int32_t P, int32_t RP);

/* SVGP closed-form expected log-likelihood (likelihoods/GaussianLinearMean.py:60-87 + dsp/utils.py:164-195),
 * summed over rows and multiplied by `scale`.  out[0] = ELL, out[1] = dELL/dlog_var_noise; g_mu, g_v: (N). */
int tgp_ell_gauss_f64(const double* Y, const double* mu, const double* v, int32_t N, const double* log_var_noise,
                      double scale, double* out, double* g_mu, double* g_v, void* workspace, size_t workspace_bytes,
                      void* stream);

/* TGP Gauss-Hermite expected log-likelihood through the flow (likelihoods/GaussianNonLinearMean.py:64-150),
 * with gradients w.r.t. mu, v, theta, rowp, log_var_noise.  out[0] = ELL, out[1] = dELL/dlog_var_noise. */
int tgp_ell_flow_f64(const tgp_model* model, const double* Y, const double* mu, const double* v, const double* rowp,
                     double* out, double* g_mu, double* g_v, double* g_theta, double* g_rowp, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Flow evaluation G(f), dG/df, log dG/df for f of shape (S,N) (row n uses rowp[n,:]):
 * CompositeFlow.forward (models/flow.py:155-158) and Flow.forward_grad (:101-104); serves prediction and the
 * WGP-style log-Jacobian (likelihoods/WarpedGaussianLinearMean.py:65-85).  Any output may be NULL. */
int tgp_flow_eval_f64(const tgp_model* model, const double* f, int32_t S, int32_t N, const double* rowp, double* G,
                      double* dG, double* logdG, void* stream);

/* Fused flow + log-Jacobian accumulation: out[0] = sum over the (S,N) array of log dG/df (fixed summation order,
 * bit-reproducible), G (optional) = the warped values -- the two quantities of a warped-GP likelihood term
 * (likelihoods/WarpedGaussianLinearMean.py:65-85: log p(y) = log N(G(y) | ...) + sum log G'(y)) in one pass over f,
 * nothing of size S x N written unless G is asked for. */
size_t tgp_flow_logdet_workspace_bytes(int32_t S, int32_t N);
int tgp_flow_logdet_f64(const tgp_model* model, const double* f, int32_t S, int32_t N, const double* rowp, double* G,
                        double* out, void* workspace, size_t workspace_bytes, void* stream);

/* Evaluation path given q(f) moments: predictive moments m1, m2
 * (GaussianNonLinearMean.marginal_moments :152-203 / GaussianLinearMean.marginal_moments :89-118) and the
 * per-row test log-likelihood WITHOUT the -0.5*log(pi) constant (models/sparse_MF_SP.py:705-776, 786-799).
 * Y may be NULL (then logp is not written). */
int tgp_predict_f64(const tgp_model* model, const double* mu, const double* v, const double* rowp, const double* Y,
                    double Y_std, double* m1, double* m2, double* logp, void* stream);

/* Inducing-point initialisation: the numerical kernels behind utils.KMEANS, which replaces
 * sklearn.cluster.KMeans(init='k-means++', n_init, random_state) of dsp/utils.py:143-159.  Random draws, the stopping
 * rule and the restarts stay on the host (tgp/pytorch_amd/utils.py); D <= 16.
 *   assign : labels[n] = argmin_k |x_n - c_k|^2 (first minimum), mind2[n] = that distance (mind2 may be NULL)
 *   segsum : sums[k,:] = sum of the rows X[order[i]], offs[k] <= i < offs[k+1] (rows sorted by label; fixed order)
 *   pp     : out[t,n] = min(closest[n], |x_n - x_cand[t]|^2) for T <= 16 k-means++ trial candidates (closest NULL: +inf) */
int tgp_kmeans_assign_f64(const double* X, int32_t N, int32_t D, const double* C, int32_t K, int32_t* labels, double* mind2,
                          void* stream);
int tgp_kmeans_segsum_f64(const double* X, int32_t D, const int64_t* order, const int64_t* offs, int32_t K, double* sums,
                          void* stream);
int tgp_kmeans_pp_f64(const double* X, int32_t N, int32_t D, const int64_t* cand, int32_t T, const double* closest,
                      double* out, void* stream);

/* The per-row parameter networks of the input-dependent flows: Sinh_ArcsinhFlow's NNets_a / NNets_b,
 * models/flow.py:836-897,949-965 -- `nnets` MLPs of one architecture D -> H (x L hidden layers) -> 1, each layer
 * Linear -> activation -> Dropout(p) (pytorchlib apply_linear, flow.py:853-871; layer order unpinned).
 * Packed weights per net, torch parameter order: [W1 (H,D) | b1 (H) | W2 (H,H) | b2 (H) | ... | Wout (1,H) | bout (1)].
 * Dropout masks are a counter-based hash of (seed, step, net, layer, row, unit); `step_dev` (int32 on the device,
 * e.g. the Adam step counter of tgp_adam_dev_f64; NULL = 0) makes the mask change every step and stay valid under
 * hipGraph replay; forward and backward of one step must see the same value.  H <= 64, L <= 3. */
typedef struct tgp_mlp {
  int32_t N, D, H, L, nnets;
  int32_t act;      /* 0 relu, 1 tanh */
  int32_t training; /* != 0: dropout active (set_is_training, models/sparse_MF_SP.py:133-134) */
  int32_t reserved0;
  double drop_p;
  uint64_t seed;
} tgp_mlp;
size_t tgp_mlp_workspace_bytes(const tgp_mlp* mlp);
/* out (N, nnets): column k = output of net k (feeds `rowp` of tgp_elbo_step_f64). */
int tgp_mlp_forward_f64(const tgp_mlp* mlp, const double* X, const double* W, const int32_t* step_dev, double* out,
                        void* stream);
/* g_W (nnets * weights_per_net) = d(objective)/dW given g_out (N, nnets) (= g_rowp of tgp_elbo_step_f64). */
int tgp_mlp_backward_f64(const tgp_mlp* mlp, const double* X, const double* W, const int32_t* step_dev,
                         const double* g_out, double* g_W, void* workspace, size_t workspace_bytes, void* stream);
/* The same followed by the Adam update of the network weights (the reference's second parameter group: names containing
 * 'NNets', weight decay 1e-5, main.py:276-288; torch.optim.Adam, dsp/trainers/optimizers.py:12) in the launch that
 * reduces the weight gradients: adam->params must be W, adam->grads g_W, adam->n the number of weights; adam->step_dev is
 * the group's device step counter (bumped by this call; adam->phases is ignored).  `step_dev` (the dropout masks' step
 * word) may be the same counter: it is read before the bump.  Same arithmetic as tgp_mlp_backward_f64 +
 * tgp_adam_dev_f64(weight_decay). */
int tgp_mlp_backward_adam_f64(const tgp_mlp* mlp, const double* X, double* W, const int32_t* step_dev, const double* g_out,
                              double* g_W, void* workspace, size_t workspace_bytes, const tgp_adam_args* adam,
                              double weight_decay, void* stream);

/* Minibatch rows of a data set resident in HBM -- replaces the per-step host collation + H2D copy of the reference's
 * DataLoader (dsp/data/data.py:86-88, trainers/trainer_base.py:330): Xb[r] = X[index[cursor + offset + r]] and Yb
 * likewise for r < nrows (index NULL: the stored order; entries are row numbers < N).  `cursor_dev` = int32[2]
 * {position in the epoch, ticket}, both 0 at the start of an epoch; the launch advances the position by `advance`
 * and wraps it to 0 when it reaches `wrap`, so a captured launch serves batch after batch under hipGraph replay.
 * `offset` selects a rank's shard inside the batch. */
int tgp_gather_rows_f64(const double* X, const double* Y, int32_t N, int32_t D, const int32_t* index, int32_t* cursor_dev,
                        int32_t offset, int32_t nrows, int32_t advance, int32_t wrap, double* Xb, double* Yb, void* stream);

/* Adam on a flat parameter buffer (torch.optim.Adam semantics, dsp/trainers/optimizers.py:12; L2 weight decay
 * added to the gradient as torch does).  `maximize` != 0 ascends (gradients here are of +ELBO). */
int tgp_adam_f64(double* params, const double* grads, double* exp_avg, double* exp_avg_sq, int64_t n, double lr,
                 double beta1, double beta2, double eps, double weight_decay, int32_t step, int32_t maximize,
                 void* stream);

/* Graph-capturable Adam: the step count lives on the device (`step_dev`, int32[2] = {step, ticket}, both start at
 * 0) so that a captured launch stays valid across replays; the call uses step = step_dev[0] + 1 and the last
 * workgroup to finish increments the counter. */
int tgp_adam_dev_f64(double* params, const double* grads, double* exp_avg, double* exp_avg_sq, int64_t n, double lr,
                     double beta1, double beta2, double eps, double weight_decay, int32_t* step_dev, int32_t maximize,
                     void* stream);

/* tgp_adam_dev_f64 with two parameter groups in one buffer: elements [0, n_plain) without weight decay, elements
 * [n_plain, n) with `weight_decay_tail` -- the reference's optimiser groups (main.py:276-288: weight decay 1e-5 on
 * the parameters whose name contains 'NNets'). */
int tgp_adam_dev_groups_f64(double* params, const double* grads, double* exp_avg, double* exp_avg_sq, int64_t n,
                            double lr, double beta1, double beta2, double eps, int64_t n_plain, double weight_decay_tail,
                            int32_t* step_dev, int32_t maximize, void* stream);

/* ---- the collective of the data-parallel step (ONE sum all-reduce of the flat [gradients | ELBO, ELL, KL]
 * buffer per step over RCCL / xGMI; the reference itself is single-device, trainers/trainer_base.py:329-349) ----
 * One process per GPU, one communicator per process.  RCCL is bound at run time: tgp_comm_load(path) dlopens it
 * (NULL = "librccl.so" by the loader's rules; a process that imported torch already holds torch/lib/librccl.so), every
 * other entry point of this library works without it.  Rank 0 draws a 128-byte id (tgp_comm_unique_id), the host
 * distributes it (any channel: a file, MPI, torch.distributed's store), every rank calls tgp_comm_init on its device;
 * tgp_allreduce_f64 sums `buf` (n doubles, device memory) in place across the ranks ON `stream` -- stream-ordered like
 * the kernels, valid under stream capture, so the step [kernels -> all-reduce -> Adam] can be ONE captured graph with
 * no side stream.  The C ABI's counterpart of engine.allreduce_flat (torch.distributed, backend nccl = RCCL), which
 * stays the default of the Python engine.  Errors: TGP_E_COMM + tgp_last_error(). */
int tgp_comm_load(const char* rccl_path);
int tgp_comm_unique_id(void* id128);
int tgp_comm_init(const void* id128, int32_t nranks, int32_t rank, void** comm);
int tgp_allreduce_f64(void* comm, double* buf, int64_t n, void* stream);
int tgp_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* TGP_HIP_H */
