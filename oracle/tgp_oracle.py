"""CPU oracle for the TGP sparse-variational ELBO hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this module; the product path (tgp/pytorch_amd) never
does and fails loudly when its HIP library is missing.

What it is: a plain eager PyTorch-CPU (float64 by default) restatement of the reference's
algorithm for the path named by BASELINE.json (jmaronas/TGP.pytorch, `sparse_MF_SP.ELBO` and
everything below it).  Every function cites the reference file:line it follows and keeps the
reference's *op sequence* (the repeated triangular solves, the `.repeat` of Y/noise over the
quadrature nodes, the three-product log-Gaussian ...), so that timing it is a fair stand-in for
the reference's CPU path ("cpu_baseline.kind" = "port").  Gradients come from torch.autograd,
exactly as in the reference (it has no hand-written backward).

Parity status: PINNED for all reference-owned arithmetic -- tests/golden/*.npz were produced by
executing the reference's own files (oracle/gen_golden.py, shim import of /root/reference in the
build container) and tests/test_oracle_golden.py checks this module against them to 1e-10.
UNPINNED for the arithmetic that lives in third-party packages absent from /root/reference
(gpytorch 1.1.1: ScaleKernel(RBFKernel) formula, softplus constraints, GaussHermiteQuadrature1D,
inv_softplus; jmaronas/pytorch_library@version-1.5.0: apply_linear layer order): those are
restated from their published formulas, here and in the shims alike.

Flow "program" (shared vocabulary with include/tgp_hip.h): a list of blocks, each a tuple
(kind, K, poff, flags); parameters live in one flat vector `theta` (shared scalars) or in
per-row columns `rowp[:, poff + j]` when the PER_ROW flag is set (input-dependent flows).
"""
import math

import numpy as np
import torch
from torch.nn.functional import softplus

# ---- flow program vocabulary (mirrors include/tgp_hip.h) -----------------------------------------
FLOW_AFFINE = 0      # g = a*f + b                                (flow.py:330-340)
FLOW_SAL = 1         # g = sinh(b*asinh(f) - a)                   (flow.py:936-977)
FLOW_STEPTANH = 2    # g = f + sum_k a_k + sp(b_k) tanh((f-c_k)/sp(d_k))   (flow.py:1096-1103,730-773)
FLAG_RESTRICT = 1    # set_restrictions=True  (softplus on affine.a / SAL.b)
FLAG_ADD_F0 = 2      # add_init_f0=True
FLAG_PER_ROW = 4     # parameters are per-row columns (input dependent flow, flow.py:949-965)


# The reference builds `cg.pi = torch.tensor(math.pi)` while the default dtype is still float32
# (dsp/config.py:49-50,71 run before set_maximum_precission), so every log-Gaussian constant uses the
# float32-rounded pi, and test_log_likelihood's "-0.5 log pi" is even evaluated in float32
# (sparse_MF_SP.py:768-776).  Reproduced here on purpose: results must equal the reference's.
PI_REF = float(np.float32(math.pi))
LOG_2PI_REF = math.log(2.0 * PI_REF)                       # dsp/utils.py:181
LOG_PI_REF_F32 = float(np.log(np.float32(math.pi)))        # sparse_MF_SP.py:776


def nparams(kind, K):
    return 4 * K if kind == FLOW_STEPTANH else 2


def inv_softplus(x):
    """gpytorch.utils.transforms.inv_softplus (third party, closed form)."""
    x = torch.as_tensor(x, dtype=torch.float64)
    return x + torch.log(-torch.expm1(-x))


def hermgauss(S, dtype=torch.float64):
    """Nodes/weights as gpytorch GaussHermiteQuadrature1D builds them (numpy hermgauss)."""
    x, w = np.polynomial.hermite.hermgauss(S)
    return torch.tensor(x, dtype=dtype), torch.tensor(w, dtype=dtype)


# ---- kernel: gpytorch ScaleKernel(RBFKernel(ard)) -- utils_models.py:188-193 ----------------------
def scale_rbf(X1, X2, raw_lengthscale, raw_outputscale):
    """sigma^2 exp(-0.5 ||(x-z)/l||^2); gpytorch's centred expansion + clamp (third party)."""
    ls = softplus(raw_lengthscale).reshape(1, -1)
    a = X1 / ls
    b = X2 / ls
    shift = a.mean(-2, keepdim=True)
    a = a - shift
    b = b - shift
    sq = a.pow(2).sum(-1, keepdim=True) - 2.0 * a.matmul(b.transpose(-2, -1)) \
        + b.pow(2).sum(-1, keepdim=True).transpose(-2, -1)
    return softplus(raw_outputscale).reshape(()) * torch.exp(-0.5 * sq.clamp_min(0.0))


def scale_matern32(X1, X2, raw_lengthscale, raw_outputscale):
    """sigma^2 (1 + sqrt3 r) exp(-sqrt3 r): gpytorch 1.1.1 ScaleKernel(MaternKernel(nu=1.5, ard)) as built by
    instance_kernel('scale_matern32'), utils_models.py:199-204 (third party: mean-centred inputs, the distance is
    sqrt(clamp(sq_dist, 1e-30)) with the same centred expansion as the RBF)."""
    ls = softplus(raw_lengthscale).reshape(1, -1)
    mean = X1.reshape(-1, X1.shape[-1]).mean(0, keepdim=True)
    a = (X1 - mean) / ls
    b = (X2 - mean) / ls
    shift = a.mean(-2, keepdim=True)
    a = a - shift
    b = b - shift
    sq = a.pow(2).sum(-1, keepdim=True) - 2.0 * a.matmul(b.transpose(-2, -1)) \
        + b.pow(2).sum(-1, keepdim=True).transpose(-2, -1)
    r = sq.clamp_min(1e-30).sqrt()
    return softplus(raw_outputscale).reshape(()) * (1.0 + math.sqrt(3.0) * r) * torch.exp(-math.sqrt(3.0) * r)


KERNELS = {"scale_rbf": scale_rbf, "scale_matern32": scale_matern32}


def scale_rbf_diag(X, raw_outputscale):
    """kernel(X, diag=True) == outputscale for a stationary kernel (sparse_MF_SP.py:313)."""
    return softplus(raw_outputscale).reshape(()) * torch.ones(X.shape[:-1], dtype=X.dtype)


# ---- psd_safe_cholesky -- dsp/utils.py:222-270 ----------------------------------------------------
class NanError(RuntimeError):
    pass


def psd_safe_cholesky(A, jitter=None, constant_jitter=None):
    """Cholesky with the reference's retry ladder: on failure add jitter*10^i, i=0..2, to the
    diagonal (jitter 1e-6 fp32 / 1e-8 fp64 unless given); NaN input -> NanError.
    Returns (L, A_used, jitter_used)."""
    if constant_jitter is not None:
        A = A + constant_jitter * torch.eye(A.shape[-1], dtype=A.dtype)
    if torch.isnan(A).any():
        raise NanError("cholesky: %d of %d elements are NaN" % (int(torch.isnan(A).sum()), A.numel()))
    L, info = torch.linalg.cholesky_ex(A)
    if int(info.max()) == 0:
        return L, A, 0.0
    if jitter is None:
        jitter = 1e-6 if A.dtype == torch.float32 else 1e-8
    Ap = A.clone()
    prev = 0.0
    for i in range(3):
        new = jitter * (10 ** i)
        Ap.diagonal(dim1=-2, dim2=-1).add_(new - prev)
        prev = new
        L, info = torch.linalg.cholesky_ex(Ap)
        if int(info.max()) == 0:
            return L, Ap, new
    raise RuntimeError("cholesky: matrix not positive definite even with jitter %g" % prev)


# ---- q(f) marginals -- sparse_MF_SP.py:274-396 (whitened, diagonal) -------------------------------
def qf_moments(X, Z, raw_lengthscale, raw_outputscale, m, Lam, jitter=None, kernel="scale_rbf"):
    """mu_n, v_n of q(f_n).  X (N,D), Z (M,D), m (M,), Lam (M,M) dense (tril applied at use).
    Keeps the reference's sequence: K_xx diag, K_zz, K_xz, chol, tril mask, S = Lq Lq^T,
    triangular_solve(m, L^T), cholesky_solve(K_zx, L), triangular_solve(K_zx, L)."""
    kfun = KERNELS[kernel]
    K_xx = scale_rbf_diag(X, raw_outputscale)                                   # :313 (k(x,x) = sigma^2 for both kernels)
    K_zz = kfun(Z, Z, raw_lengthscale, raw_outputscale)                         # :316
    K_xz = kfun(X, Z, raw_lengthscale, raw_outputscale)                         # :319
    K_zx = K_xz.transpose(-2, -1)                                               # :327
    L, _, _ = psd_safe_cholesky(K_zz, jitter=jitter)                            # :330
    mask = torch.ones(Lam.shape[-2:], dtype=Lam.dtype).tril(0)                  # :344
    Lq = Lam * mask                                                             # :345
    S = Lq @ Lq.transpose(-2, -1)                                               # :346
    sol_m = torch.linalg.solve_triangular(L.transpose(-2, -1), m.reshape(-1, 1), upper=True)  # :354
    mu = (K_xz @ sol_m).reshape(-1)                                             # :355
    sol = torch.cholesky_solve(K_zx, L, upper=False)                            # :376
    rhs = torch.linalg.solve_triangular(L, K_zx, upper=False)                   # :380
    v = K_xx - (K_zx * sol).sum(0) + (rhs * (S @ rhs)).sum(0)                   # :382
    return mu, v


# ---- whitened KL -- sparse_MF_SP.py:406-431 -------------------------------------------------------
def kld_whitened(m, Lam):
    M = m.shape[-1]
    mask = torch.ones(Lam.shape[-2:], dtype=Lam.dtype).tril(0)
    Lq = Lam * mask
    S = Lq @ Lq.transpose(-2, -1)
    dot_mean = (m * m).sum()
    log_det = torch.log(torch.diagonal(Lq) ** 2).sum()
    trace = torch.diagonal(S).sum()
    return 0.5 * (-log_det + dot_mean + trace - float(M))


# ---- flows -- models/flow.py ----------------------------------------------------------------------
def _asinh_ref(f):
    """Sinh_ArcsinhFlow.asinh, flow.py:904-905 (the naive log form, reproduced on purpose)."""
    return torch.log(f + (f ** 2 + 1) ** 0.5)


def flow_forward(f, program, theta, rowp=None):
    """CompositeFlow.forward (flow.py:155-158) over the block program.  f: (S,N) or (N,)."""
    for kind, K, poff, flags in program:
        per_row = bool(flags & FLAG_PER_ROW)

        def P(j):
            return rowp[:, poff + j] if per_row else theta[poff + j]

        if kind == FLOW_AFFINE:                                     # flow.py:330-340
            a = P(0)
            if flags & FLAG_RESTRICT:
                a = softplus(a)
            f = a * f + P(1)
        elif kind == FLOW_SAL:                                      # flow.py:936-977
            a, b = P(0), P(1)
            if flags & FLAG_RESTRICT:
                b = softplus(b)
            g = torch.sinh(b * _asinh_ref(f) - a)
            f = g + f if flags & FLAG_ADD_F0 else g
        elif kind == FLOW_STEPTANH:                                 # flow.py:1096-1103 + 760-771
            acc = 0.0
            for k in range(K):
                a, b, c, d = P(4 * k), P(4 * k + 1), P(4 * k + 2), P(4 * k + 3)
                acc = acc + (a + softplus(b) * torch.tanh((f - c) / softplus(d)))   # switch_off = (1, 0)
            f = acc + f if flags & FLAG_ADD_F0 else acc
        else:
            raise ValueError("unknown flow kind %r" % (kind,))
    return f


# ---- likelihoods ----------------------------------------------------------------------------------
def batched_log_gaussian(obs, mean, cov, cov_is_inverse):
    """dsp/utils.py:164-195 with diagonal=True; reduces the last dim (size 1 on this path)."""
    n = mean.shape[-1]
    cte = n * LOG_2PI_REF
    log_det = torch.log(cov).sum(-1)
    inv_c = cov
    if not cov_is_inverse:
        inv_c = 1.0 / cov
    else:
        log_det = -log_det
    arg = (obs * inv_c * obs).sum(-1) - 2.0 * (obs * inv_c * mean).sum(-1) + (mean * inv_c * mean).sum(-1)
    return -0.5 * (cte + log_det + arg)


def ell_gauss(Y, mu, v, log_var_noise):
    """GaussianLinearMean.expected_log_prob, likelihoods/GaussianLinearMean.py:60-87. Y,mu,v: (N,)."""
    n = Y.shape[0]
    c_inv = (1.0 / torch.exp(log_var_noise)).reshape(1).expand(n)
    # one Gaussian of dimension N (Dy=1): obs/mean/cov of shape (1,N)
    log_p = batched_log_gaussian(Y.reshape(1, n), mu.reshape(1, n), c_inv.reshape(1, n), cov_is_inverse=True)
    trace = -0.5 * (c_inv * v).sum()
    return (log_p + trace).reshape(())


def ell_flow(Y, mu, v, log_var_noise, program, theta, xs, ws, rowp=None):
    """GaussianNonLinearMean.expected_log_prob + log_non_linear (GaussianNonLinearMean.py:64-150)
    through gpytorch's GaussHermiteQuadrature1D.  Y,mu,v: (N,). Returns scalar (summed over N)."""
    S = xs.shape[0]
    n = Y.shape[0]
    noise = torch.exp(log_var_noise).reshape(1).expand(n)
    f = torch.sqrt(2.0 * v).reshape(1, n) * xs.reshape(S, 1) + mu.reshape(1, n)     # quadrature.py
    Yr = Y.reshape(1, n, 1).repeat(S, 1, 1)                                         # :91
    Cr = noise.reshape(1, n, 1).repeat(S, 1, 1)                                     # :92
    fK = f.clone()                                                                  # :94
    fK = flow_forward(fK, program, theta, rowp)                                     # :101-103
    fK = fK.reshape(S, n, 1)
    log_p = batched_log_gaussian(Yr, fK, Cr, cov_is_inverse=False)                  # :108  (S,N)
    ell_n = ((1.0 / math.sqrt(math.pi)) * (log_p * ws.reshape(S, 1))).sum(0)        # quadrature.py
    return ell_n.sum()                                                              # :148


# ---- ELBO -- sparse_MF_SP.py:552-626 --------------------------------------------------------------
def elbo(X, Y, Z, raw_lengthscale, raw_outputscale, m, Lam, log_var_noise, N_total,
         program=None, theta=None, xs=None, ws=None, rowp=None, jitter=None, kernel="scale_rbf"):
    """Returns (ELBO, ELL, KLD) as 0-d tensors.  program=None -> SVGP closed form (sparse_MF_GP)."""
    Xr = X.repeat(1, 1, 1)[0]                                                       # :565 (Dy = 1)
    kl = kld_whitened(m, Lam)                                                       # :571
    mu, v = qf_moments(Xr, Z, raw_lengthscale, raw_outputscale, m, Lam, jitter, kernel)   # :580
    y = Y.reshape(-1)
    if program is None:
        ell = ell_gauss(y, mu, v, log_var_noise)
    else:
        ell = ell_flow(y, mu, v, log_var_noise, program, theta, xs, ws, rowp)
    ell = (float(N_total) / X.shape[0]) * ell                                       # :626
    return ell - kl, ell, kl                                                        # :590-598


def elbo_and_grads(X, Y, params, N_total, program=None, xs=None, ws=None, rowp=None, kernel="scale_rbf"):
    """params: dict with Z, raw_lengthscale, raw_outputscale, m, Lam, log_var_noise[, theta].
    Returns ((ELBO, ELL, KLD), grads dict) -- autograd, like the reference (trainer_base.py:341)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    rp = None
    if rowp is not None:
        rp = rowp.detach().clone().requires_grad_(True)
    out = elbo(X, Y, leaves["Z"], leaves["raw_lengthscale"], leaves["raw_outputscale"], leaves["m"],
               leaves["Lam"], leaves["log_var_noise"], N_total, program, leaves.get("theta"), xs, ws, rp, kernel=kernel)
    out[0].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    if rp is not None:
        grads["rowp"] = rp.grad
    return tuple(o.detach() for o in out), grads


# ---- evaluation path ---------------------------------------------------------------
def marginal_moments_flow(mu, v, log_var_noise, program, theta, xs, ws, rowp=None):
    """GaussianNonLinearMean.marginal_moments, GaussianNonLinearMean.py:152-203."""
    S = xs.shape[0]
    n = mu.shape[0]
    f = torch.sqrt(2.0 * v).reshape(1, n) * xs.reshape(S, 1) + mu.reshape(1, n)
    g = flow_forward(f, program, theta, rowp)
    c = 1.0 / math.sqrt(math.pi)
    m1 = c * (g * ws.reshape(S, 1)).sum(0)
    e2 = c * (g * g * ws.reshape(S, 1)).sum(0)
    return m1, torch.exp(log_var_noise).reshape(()) + e2 - m1 ** 2


def marginal_moments_gauss(mu, v, log_var_noise):
    """GaussianLinearMean.marginal_moments, GaussianLinearMean.py:89-118 (diagonal)."""
    return mu.clone(), torch.exp(log_var_noise).reshape(()) + v


def test_log_lik_flow(Y, mu, v, log_var_noise, program, theta, xs, ws, Y_std, rowp=None):
    """Per-row log p(y_n) of sparse_MF_SP.test_log_likelihood, sparse_MF_SP.py:705-776
    (non-Bayesian branch): logsumexp_s[log w_s + logN(Ystd y | Ystd G(f_s), Ystd^2 s2y)] - 0.5 log pi."""
    S = xs.shape[0]
    n = mu.shape[0]
    f = torch.sqrt(2.0 * v).reshape(1, n) * xs.reshape(S, 1) + mu.reshape(1, n)
    g = flow_forward(f, program, theta, rowp) * Y_std
    var = torch.exp(log_var_noise).reshape(()) * Y_std ** 2
    y = Y.reshape(1, n) * Y_std
    logn = -0.5 * (LOG_2PI_REF + torch.log(var) + (y - g) ** 2 / var)
    return torch.logsumexp(torch.log(ws).reshape(S, 1) + logn, 0) - 0.5 * LOG_PI_REF_F32


def test_log_lik_sum_ref(per_row_logp):
    """Sum over the batch exactly as sparse_MF_SP.py:776 forms it: the constant 0.5*MB*log(pi) is a
    float32 product there (cg.pi is a float32 tensor), so rebuild it that way."""
    mb = per_row_logp.shape[0]
    lse_sum = (per_row_logp + 0.5 * LOG_PI_REF_F32).sum()
    return lse_sum - float(np.float32(0.5 * mb) * np.log(np.float32(math.pi)))


# ---- synthetic, seeded problem instances shared by tests and bench ---------------------
def sal_program(num_blocks, per_row=False):
    """dsp/flows.py:115-136 SAL: [sinh_arcsinh, affine] x num_blocks; identity init a=0,b=1 / a=1,b=0."""
    prog, theta, poff, rcol = [], [], 0, 0
    for _ in range(num_blocks):
        if per_row:
            prog.append((FLOW_SAL, 0, rcol, FLAG_PER_ROW))
            rcol += 2
        else:
            prog.append((FLOW_SAL, 0, poff, 0))
            theta += [0.0, 1.0]
            poff += 2
        prog.append((FLOW_AFFINE, 0, poff, 0))
        theta += [1.0, 0.0]
        poff += 2
    return prog, torch.tensor(theta, dtype=torch.float64)


def steptanh_program(num_blocks, num_steps, rng):
    """dsp/flows.py:239-277 StepTanhL (add_f0=True as exp_utils.py:31 passes): [step_flow(K tanh), affine]."""
    prog, theta, poff = [], [], 0
    for _ in range(num_blocks):
        prog.append((FLOW_STEPTANH, num_steps, poff, FLAG_ADD_F0))
        for _k in range(num_steps):
            e1, e2, e3, e4 = rng.standard_normal(4)
            e2 = float(inv_softplus(abs((e2 + 1.0) / num_steps)))
            e4 = float(inv_softplus(abs((e4 + 1.0) / num_steps)))
            theta += [e1, e2, e3, e4]
        poff += 4 * num_steps
        prog.append((FLOW_AFFINE, 0, poff, 0))
        theta += [1.0, 0.0]
        poff += 2
    return prog, torch.tensor(theta, dtype=torch.float64)


def mlp_rowp(X, W, D, H, L, nnets, act="relu", masks=None, drop_p=0.0):
    """Per-row flow parameters of the input-dependent SAL blocks (models/flow.py:853-871, 949-965): `nnets` MLPs
    D -> H x L -> 1, each hidden layer Linear -> activation -> Dropout (the assumed pytorchlib.apply_linear order,
    unpinned), evaluated from the packed weights W (per net, per layer: weight (out, in) then bias).  Eval mode unless
    `masks[k][l]` (N, H) keep-masks are given.  Returns (N, nnets): columns a_0, b_0, a_1, b_1, ..."""
    actf = {"relu": torch.relu, "tanh": torch.tanh}[act]
    pw = D * H + H + (L - 1) * (H * H + H) + H + 1
    outs = []
    for k in range(nnets):
        w = W[k * pw:(k + 1) * pw]
        o, h, nin = 0, X, D
        for l in range(L):
            Wl = w[o:o + H * nin].reshape(H, nin)
            o += H * nin
            bl = w[o:o + H]
            o += H
            h = actf(h @ Wl.T + bl)
            if masks is not None:
                h = h * masks[k][l] / (1.0 - drop_p)
            nin = H
        outs.append(h @ w[o:o + H] + w[o + H])
    return torch.stack(outs, 1)


def synthetic_problem(N, D, M, seed=0, flow="sal2", S=32, perturb=True, dtype=torch.float64):
    """Seeded inputs per SURVEY.md 8(d): X~N(0,1); Y = zscore(sin(Xw)+0.1 x0^2+0.05 eps);
    Z = first M rows of a seeded permutation; l=2, s2=2; m ~ 0.5 N(0,1); Lq = sqrt(1e-5) I +
    0.05 tril(N(0,1)); noise 0.05; flow params = identity init + 0.3 N(0,1) (SAL) or StepTanhL init."""
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, D, generator=g, dtype=torch.float64)
    w = torch.randn(D, generator=g, dtype=torch.float64)
    Y = torch.sin(X @ w) + 0.1 * X[:, 0] ** 2 + 0.05 * torch.randn(N, generator=g, dtype=torch.float64)
    Y = ((Y - Y.mean()) / Y.std()).reshape(N, 1)
    perm = torch.randperm(N, generator=g)
    Z = X[perm[:M]].clone()
    g1 = torch.Generator().manual_seed(seed + 1)
    p = {
        "Z": Z,
        "raw_lengthscale": inv_softplus(torch.full((D,), 2.0)),
        "raw_outputscale": inv_softplus(torch.tensor(2.0)).reshape(1),
        "m": torch.zeros(M, dtype=torch.float64),
        "Lam": math.sqrt(1e-5) * torch.eye(M, dtype=torch.float64),
        "log_var_noise": torch.log(torch.tensor([0.05], dtype=torch.float64)),
    }
    if perturb:
        p["m"] = 0.5 * torch.randn(M, generator=g1, dtype=torch.float64)
        p["Lam"] = p["Lam"] + 0.05 * torch.randn(M, M, generator=g1, dtype=torch.float64)  # dense: upper part must be ignored
        p["raw_lengthscale"] = p["raw_lengthscale"] + 0.1 * torch.randn(D, generator=g1, dtype=torch.float64)
    program = None
    rowp = None
    if flow is not None and flow.startswith("sal"):
        program, theta = sal_program(int(flow[3:]))
        if perturb:
            theta = theta + 0.3 * torch.randn(theta.shape, generator=g1, dtype=torch.float64)
        p["theta"] = theta
    elif flow is not None and flow.startswith("idsal"):
        nb = int(flow[5:])
        program, theta = sal_program(nb, per_row=True)
        if perturb:
            theta = theta + 0.3 * torch.randn(theta.shape, generator=g1, dtype=torch.float64)
        p["theta"] = theta
        base = torch.tensor([0.0, 1.0] * nb, dtype=torch.float64)
        rowp = base.reshape(1, -1) + 0.2 * torch.randn(N, 2 * nb, generator=g1, dtype=torch.float64)
    elif flow is not None and flow.startswith("tanh"):
        nb, ns = (int(t) for t in flow[4:].split("x"))
        program, theta = steptanh_program(nb, ns, np.random.default_rng(seed))
        p["theta"] = theta
    xs, ws = hermgauss(S)
    out = {"X": X, "Y": Y, "params": p, "program": program, "xs": xs, "ws": ws, "rowp": rowp, "N_total": float(N)}
    if dtype != torch.float64:
        out = _cast(out, dtype)
    return out


def _cast(o, dtype):
    if torch.is_tensor(o):
        return o.to(dtype) if o.is_floating_point() else o
    if isinstance(o, dict):
        return {k: _cast(v, dtype) for k, v in o.items()}
    return o
