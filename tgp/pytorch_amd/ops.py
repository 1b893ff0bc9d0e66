"""Operator layer: torch tensors in, HIP kernels underneath (through the C ABI in lib.py).

`elbo_step` is the fused replacement of `sparse_MF_SP.ELBO` + `loss.backward()` for one minibatch
(reference: code/dsp/models/sparse_MF_SP.py:552-598, code/dsp/trainers/trainer_base.py:337-341);
`ElboFunction` wraps it as a torch.autograd.Function so the reference's trainer idiom
(`loss = -ELBO; loss.backward(); optimizer.step()`) keeps working on the drop-in model classes.
The other functions expose the stand-alone operators of the path.
"""
import math
import warnings

import numpy as np
import torch

from . import lib as L


class NanError(RuntimeError):
    """Same role as gpytorch.utils.errors.NanError raised by dsp/utils.py:241-254."""


STATUS_SYNC_TIMEOUT = -77   # include/tgp_hip.h TGP_STATUS_SYNC_TIMEOUT: status[0] of a launch whose hand-off wait expired


class HandoffTimeoutError(RuntimeError):
    """A workgroup of the prepare or of the M x M backward launch gave up waiting for a hand-off word (status[0] == TGP_STATUS_SYNC_TIMEOUT):
    the status buffer's hand-off words (status[4..7]) were not zero at the call, or the launch's producer workgroups
    never became resident.  The results of that call are invalid; this is NOT a Cholesky failure."""


class NotPSDError(RuntimeError):
    pass


class NumericalWarning(RuntimeWarning):
    pass


# ---------------------------------------------------------------------------------------------------
# quadrature nodes (gpytorch GaussHermiteQuadrature1D: numpy hermgauss) and workspace cache
# ---------------------------------------------------------------------------------------------------
_quad_cache = {}


def gauss_hermite(S, device):
    """(xs, wn = w/sqrt(pi), logw) as float64 device tensors."""
    key = (int(S), str(device))
    if key not in _quad_cache:
        x, w = np.polynomial.hermite.hermgauss(int(S))
        xs = torch.tensor(x, dtype=torch.float64, device=device)
        wn = torch.tensor(w / math.sqrt(math.pi), dtype=torch.float64, device=device)
        _quad_cache[key] = (xs, wn)
    return _quad_cache[key]


_ws_cache = {}


def kernel_id(kernel):
    """'scale_rbf' / 'scale_matern32' (instance_kernel names, models/utils_models.py:188-204) or a TGP_KERNEL_* id."""
    if isinstance(kernel, str):
        if kernel not in L.KERNELS:
            raise L.TgpError("kernel '%s' has no HIP implementation (have: %s)" % (kernel, ", ".join(L.KERNELS)))
        return L.KERNELS[kernel]
    return int(kernel)


def workspace(N, D, M, S, nblk, P, RP, device, kernel=0, plan=0):
    key = (N, D, M, S, nblk, P, RP, str(device), torch.cuda.current_stream().cuda_stream, kernel, int(plan))
    buf = _ws_cache.get(key)
    if buf is None:
        nbytes = L.load().tgp_workspace_bytes_plan(N, D, M, max(S, 1), nblk, P, RP, kernel, int(plan))
        if nbytes == 0:
            raise L.TgpError("unsupported problem shape N=%d D=%d M=%d (this build: D<=16, M<=4096)" % (N, D, M))
        buf = torch.empty(nbytes // 8 + 16, dtype=torch.float64, device=device)
        _ws_cache[key] = buf
    return buf


def _c(t, name):
    if t is None:
        return None
    if t.dtype != torch.float64:
        raise L.TgpError("%s must be float64 (the reference's main.py runs in float64), got %s" % (name, t.dtype))
    return t.contiguous()


class FlowSpec:
    """Device-side description of a flow: program (nblk,4) int32, P shared scalars, RP per-row columns."""

    def __init__(self, program, P, RP, device):
        self.blocks = [tuple(int(v) for v in b) for b in program]
        self.nblk = len(self.blocks)
        self.P, self.RP = int(P), int(RP)
        # host array: the C ABI copies the program into the kernel arguments
        self.program = np.ascontiguousarray(np.array(self.blocks if self.blocks else [(0, 0, 0, 0)], dtype=np.int32))

    @property
    def program_ptr(self):
        import ctypes
        return ctypes.c_void_p(self.program.ctypes.data)

    def to(self, device):
        return FlowSpec(self.blocks, self.P, self.RP, device)


def _model_struct(X, Z, raw_ls, raw_os, m, Lam, lvn, scale, jitter, kl_scale, flow, theta, S, kernel=0, plan=0):
    md = L.TgpModel()
    md.kernel = kernel_id(kernel)
    md.plan = int(plan)          # lib.PLAN_*: which of the equivalent kernels this call runs; 0 = the library's choice
    md.N, md.D = X.shape[0], X.shape[1]
    md.M = m.numel()
    md.scale, md.jitter, md.kl_scale = float(scale), float(jitter), float(kl_scale)
    md.Z, md.raw_ls, md.raw_os = L.ptr(Z), L.ptr(raw_ls), L.ptr(raw_os)
    md.m, md.Lam, md.log_var_noise = L.ptr(m), L.ptr(Lam), L.ptr(lvn)
    keep = []
    if flow is None:
        md.lik, md.S, md.nblk, md.P, md.RP = L.LIK_GAUSS, 1, 0, 0, 0
    else:
        xs, wn = gauss_hermite(S, X.device)
        keep += [xs, wn]
        md.lik, md.S, md.nblk, md.P, md.RP = L.LIK_FLOW, int(S), flow.nblk, flow.P, flow.RP
        md.program, md.xs, md.wn = flow.program_ptr, L.ptr(xs), L.ptr(wn)
        md.theta = L.ptr(theta) if flow.P > 0 else None
    return md, keep


def elbo_step(X, Y, Z, raw_ls, raw_os, m, Lam, lvn, N_total, flow=None, theta=None, rowp=None, S=None, jitter=0.0,
              kl_scale=1.0, mb_global=None, want_moments=False, kernel="scale_rbf", plan=0):
    """One fused ELBO evaluation + all gradients on the GPU.  Returns (out[4], grads dict, status[8], (mu, v)); status[0..2] are the
    Cholesky words of include/tgp_hip.h, status[4..7] the in-launch hand-off words (zero before and after every call).

    out = [ELL_shard - KL, ELL_shard, KL, 0]; grads are d(ELL_shard - kl_scale*KL)/d(param).
    `mb_global` = global minibatch size when X is a row shard (defaults to X.shape[0])."""
    lib = L.load()
    X, Y = _c(X, "X"), _c(Y.reshape(-1), "Y")
    Z, raw_ls, raw_os, m, Lam, lvn = (_c(t, n) for t, n in ((Z, "Z"), (raw_ls, "raw_ls"), (raw_os, "raw_os"), (m, "m"),
                                                               (Lam, "Lam"), (lvn, "log_var_noise")))
    theta, rowp = _c(theta, "theta"), _c(rowp, "rowp")
    dev = X.device
    N, D = X.shape
    M = m.numel()
    scale = float(N_total) / float(mb_global if mb_global is not None else N)
    md, keep = _model_struct(X, Z, raw_ls, raw_os, m, Lam, lvn, scale, jitter, kl_scale, flow, theta, S, kernel, plan)
    ws = workspace(N, D, M, md.S, md.nblk, md.P, md.RP, dev, md.kernel, plan)
    out = torch.empty(4, dtype=torch.float64, device=dev)
    status = torch.zeros(8, dtype=torch.int32, device=dev)
    g = {"Z": torch.empty_like(Z), "raw_ls": torch.empty_like(raw_ls), "raw_os": torch.empty_like(raw_os),
         "m": torch.empty_like(m), "Lam": torch.empty_like(Lam), "lvn": torch.empty_like(lvn)}
    gs = L.TgpGrads()
    gs.Z, gs.raw_ls, gs.raw_os, gs.m, gs.Lam, gs.log_var_noise = (L.ptr(g[k]) for k in ("Z", "raw_ls", "raw_os", "m",
                                                                                          "Lam", "lvn"))
    if md.P > 0:
        g["theta"] = torch.empty_like(theta)
        gs.theta = L.ptr(g["theta"])
    if md.RP > 0:
        g["rowp"] = torch.empty_like(rowp)
        gs.rowp = L.ptr(g["rowp"])
    mu = v = None
    if want_moments:
        mu = torch.empty(N, dtype=torch.float64, device=dev)
        v = torch.empty(N, dtype=torch.float64, device=dev)
    rc = lib.tgp_elbo_step_f64(md, L.ptr(X), L.ptr(Y), L.ptr(rowp) if md.RP > 0 else None, L.ptr(out), gs, L.ptr(mu),
                               L.ptr(v), L.ptr(status), L.ptr(ws), ws.numel() * 8, L.stream_ptr())
    L.check(rc, "tgp_elbo_step_f64")
    return out, g, status, (mu, v)


def jitter_ladder(dtype=torch.float64, jitter=None):
    """The retry values of psd_safe_cholesky (dsp/utils.py:256-269): jitter * 10^i, i = 0..2."""
    if jitter is None:
        jitter = 1e-6 if dtype == torch.float32 else 1e-8
    return [jitter * (10 ** i) for i in range(3)]


def raise_for_status(status, retrying=False):
    """Translate the device status words into the reference's exceptions; returns True if a retry with more
    jitter is needed."""
    info, nan = int(status[0]), int(status[1])
    sticky = int(status[3]) if len(status) > 3 else 0
    if info == STATUS_SYNC_TIMEOUT or sticky:
        # status[3] counts the expired waits since the caller last zeroed it: status[0] is rewritten by every prepare launch, so a
        # timeout inside a replayed multi-step graph is only visible there
        raise HandoffTimeoutError("a hand-off wait inside the prepare / backward launch expired (status[0] = %d, %d expired wait(s) "
                                  "counted in status[3]): status[4..7] must be zero before the first call and untouched while a "
                                  "call is in flight; the step's results are invalid and the fused update was skipped where the "
                                  "launch could tell (Lam rows may have moved): restore the parameters" % (info, sticky))
    if nan:
        raise NanError("cholesky: K_MM contains NaN")
    return info != 0


def elbo_step_safe(*args, global_jitter=None, **kw):
    """elbo_step + the reference's psd_safe_cholesky protocol (one device sync to read the status)."""
    res = elbo_step(*args, **kw)
    if not raise_for_status(res[2].cpu()):
        return res
    for jit in jitter_ladder(jitter=global_jitter):
        kw["jitter"] = jit
        res = elbo_step(*args, **kw)
        if not raise_for_status(res[2].cpu()):
            warnings.warn("A not p.d., added jitter of %g to the diagonal" % jit, NumericalWarning)
            return res
    raise NotPSDError("K_MM not positive definite even with jitter %g (pivot %d)" % (jit, int(res[2][0])))


class ElboFunction(torch.autograd.Function):
    """(ELBO, ELL, KLD) = f(Z, raw_ls, raw_os, m, Lam, log_var_noise, theta, rowp); gradients flow through ELBO
    (the reference trainer differentiates `-ELBO`, trainers/trainers_regression.py:85-86); ELL and KLD are
    returned for logging and marked non-differentiable."""

    @staticmethod
    def forward(ctx, X, Y, Z, raw_ls, raw_os, m, Lam, lvn, theta, rowp, cfg):
        out, g, status, _ = elbo_step_safe(X, Y, Z, raw_ls, raw_os, m, Lam, lvn, cfg["N_total"], flow=cfg.get("flow"),
                                           theta=theta, rowp=rowp, S=cfg.get("S"),
                                           kl_scale=cfg.get("kl_scale", 1.0), mb_global=cfg.get("mb_global"),
                                           global_jitter=cfg.get("global_jitter"), kernel=cfg.get("kernel", "scale_rbf")) \
            if cfg.get("check_status", True) else \
            elbo_step(X, Y, Z, raw_ls, raw_os, m, Lam, lvn, cfg["N_total"], flow=cfg.get("flow"), theta=theta,
                      rowp=rowp, S=cfg.get("S"), kl_scale=cfg.get("kl_scale", 1.0), mb_global=cfg.get("mb_global"),
                      kernel=cfg.get("kernel", "scale_rbf"))
        ctx.grads = g
        ctx.shapes = tuple(None if t is None else t.shape for t in (Z, raw_ls, raw_os, m, Lam, lvn, theta, rowp))
        cfg["last_status"] = status
        elbo, ell, kld = out[0].clone(), out[1].clone(), out[2].clone()
        ctx.mark_non_differentiable(ell, kld)
        return elbo, ell, kld

    @staticmethod
    def backward(ctx, g_elbo, g_ell, g_kld):
        g = ctx.grads
        keys = ("Z", "raw_ls", "raw_os", "m", "Lam", "lvn", "theta", "rowp")
        res = []
        for k, shp in zip(keys, ctx.shapes):
            if shp is None or k not in g:
                res.append(None)
            else:
                res.append((g[k] * g_elbo).reshape(shp))
        return (None, None) + tuple(res) + (None,)


# ---------------------------------------------------------------------------------------------------
# stand-alone operators
# ---------------------------------------------------------------------------------------------------
def qf_moments(X, Z, raw_ls, raw_os, m, Lam, jitter=0.0, check=True, kernel="scale_rbf", info=None, plan=0):
    """q(f) marginals (models/sparse_MF_SP.py:274-396): returns mu, v of shape (N,).  `info` (a dict) receives the
    jitter the factorisation ended with (info["jitter"]: the ladder of psd_safe_cholesky may have raised it)."""
    lib = L.load()
    X = _c(X, "X")
    Z, raw_ls, raw_os, m, Lam = (_c(t, "param") for t in (Z, raw_ls, raw_os, m, Lam))
    dev = X.device
    lvn = torch.zeros(1, dtype=torch.float64, device=dev)
    md, _ = _model_struct(X, Z, raw_ls, raw_os, m, Lam, lvn, 1.0, jitter, 1.0, None, None, None, kernel, plan)
    ws = workspace(X.shape[0], X.shape[1], m.numel(), 1, 0, 0, 0, dev, md.kernel, plan)
    mu = torch.empty(X.shape[0], dtype=torch.float64, device=dev)
    v = torch.empty_like(mu)
    status = torch.zeros(8, dtype=torch.int32, device=dev)
    rc = lib.tgp_qf_moments_f64(md, L.ptr(X), L.ptr(mu), L.ptr(v), L.ptr(status), L.ptr(ws), ws.numel() * 8,
                                L.stream_ptr())
    L.check(rc, "tgp_qf_moments_f64")
    if info is not None:
        info["jitter"] = float(jitter)
    if check and raise_for_status(status.cpu()):
        for jit in jitter_ladder():
            md.jitter = jit
            L.check(lib.tgp_qf_moments_f64(md, L.ptr(X), L.ptr(mu), L.ptr(v), L.ptr(status), L.ptr(ws),
                                           ws.numel() * 8, L.stream_ptr()), "tgp_qf_moments_f64")
            if not raise_for_status(status.cpu()):
                warnings.warn("A not p.d., added jitter of %g to the diagonal" % jit, NumericalWarning)
                if info is not None:
                    info["jitter"] = float(jit)
                return mu, v
        raise NotPSDError("K_MM not positive definite")
    return mu, v


def qf_moments_bwd(X, Z, raw_ls, raw_os, m, Lam, mu_bar, v_bar, jitter=0.0, kernel="scale_rbf", plan=0):
    """Adjoint of qf_moments (tgp_qf_moments_bwd_f64): d(sum mu_bar*mu + v_bar*v)/d{Z, raw_ls, raw_os, m, Lam} as a dict."""
    lib = L.load()
    X = _c(X, "X")
    Z, raw_ls, raw_os, m, Lam = (_c(t, "param") for t in (Z, raw_ls, raw_os, m, Lam))
    mu_bar, v_bar = _c(mu_bar.reshape(-1), "mu_bar"), _c(v_bar.reshape(-1), "v_bar")
    if mu_bar.numel() != X.shape[0] or v_bar.numel() != X.shape[0]:
        raise ValueError("mu_bar / v_bar must have one entry per row of X")
    dev = X.device
    lvn = torch.zeros(1, dtype=torch.float64, device=dev)
    md, _ = _model_struct(X, Z, raw_ls, raw_os, m, Lam, lvn, 1.0, jitter, 0.0, None, None, None, kernel, plan)
    ws = workspace(X.shape[0], X.shape[1], m.numel(), 1, 0, 0, 0, dev, md.kernel, plan)
    g = {"Z": torch.empty_like(Z), "raw_ls": torch.empty_like(raw_ls), "raw_os": torch.empty_like(raw_os),
         "m": torch.empty_like(m), "Lam": torch.empty_like(Lam)}
    glvn = torch.empty_like(lvn)
    gs = L.TgpGrads()
    gs.Z, gs.raw_ls, gs.raw_os, gs.m, gs.Lam = (L.ptr(g[k]) for k in ("Z", "raw_ls", "raw_os", "m", "Lam"))
    gs.log_var_noise = L.ptr(glvn)
    status = torch.zeros(8, dtype=torch.int32, device=dev)
    rc = lib.tgp_qf_moments_bwd_f64(md, L.ptr(X), L.ptr(mu_bar), L.ptr(v_bar), gs, L.ptr(status), L.ptr(ws), ws.numel() * 8,
                                    L.stream_ptr())
    L.check(rc, "tgp_qf_moments_bwd_f64")
    return g


class QfMomentsFunction(torch.autograd.Function):
    """(mu, v) = q(f) marginals with autograd in Z, raw_ls, raw_os, m, Lam (what differentiating the reference's
    marginal_variational_qf_parameters, models/sparse_MF_SP.py:274-396, gives outside ELBO()); X gets no gradient.
    Forward tgp_qf_moments_f64 (with psd_safe_cholesky's jitter ladder), backward tgp_qf_moments_bwd_f64 at the jitter
    the forward ended with."""

    @staticmethod
    def forward(ctx, X, Z, raw_ls, raw_os, m, Lam, kernel):
        info = {}
        mu, v = qf_moments(X, Z, raw_ls, raw_os, m, Lam, kernel=kernel, info=info)
        ctx.save_for_backward(X, Z, raw_ls, raw_os, m, Lam)
        ctx.kernel = kernel
        ctx.jitter = info["jitter"]
        return mu, v

    @staticmethod
    def backward(ctx, g_mu, g_v):
        X, Z, raw_ls, raw_os, m, Lam = ctx.saved_tensors
        if g_mu is None:
            g_mu = torch.zeros(X.shape[0], dtype=torch.float64, device=X.device)
        if g_v is None:
            g_v = torch.zeros(X.shape[0], dtype=torch.float64, device=X.device)
        g = qf_moments_bwd(X, Z, raw_ls, raw_os, m, Lam, g_mu, g_v, jitter=ctx.jitter, kernel=ctx.kernel)
        return (None, g["Z"].reshape(Z.shape), g["raw_ls"].reshape(raw_ls.shape), g["raw_os"].reshape(raw_os.shape),
                g["m"].reshape(m.shape), g["Lam"].reshape(Lam.shape), None)


class KlFunction(torch.autograd.Function):
    """Whitened KL with autograd in (m, Lam): tgp_kl_whitened_f64 returns the value and both gradients in one launch."""

    @staticmethod
    def forward(ctx, m, Lam):
        kl, gm, gL = kl_whitened(m, Lam)
        ctx.save_for_backward(gm, gL)
        ctx.shapes = (m.shape, Lam.shape)
        return kl.clone()

    @staticmethod
    def backward(ctx, g):
        gm, gL = ctx.saved_tensors
        return (g * gm).reshape(ctx.shapes[0]), (g * gL).reshape(ctx.shapes[1])


def kernel_matrix(X1, X2, raw_ls, raw_os, kernel="scale_rbf", jitter=0.0):
    """K(X1, X2) (X2 None: K(X1, X1) + jitter I) for 'scale_rbf' / 'scale_matern32' (tgp_kernel_matrix_f64)."""
    lib = L.load()
    X1, raw_ls, raw_os = _c(X1, "X1"), _c(raw_ls, "raw_ls"), _c(raw_os, "raw_os")
    X2 = _c(X2, "X2")
    N1, D = X1.shape
    N2 = X2.shape[0] if X2 is not None else N1
    K = torch.empty(N1, N2, dtype=torch.float64, device=X1.device)
    L.check(lib.tgp_kernel_matrix_f64(kernel_id(kernel), L.ptr(X1), N1, L.ptr(X2), N2, D, L.ptr(raw_ls), L.ptr(raw_os),
                                      float(jitter), L.ptr(K), L.stream_ptr()), "tgp_kernel_matrix_f64")
    return K


def kmm(Z, raw_ls, raw_os, jitter=0.0):
    lib = L.load()
    Z, raw_ls, raw_os = _c(Z, "Z"), _c(raw_ls, "raw_ls"), _c(raw_os, "raw_os")
    M, D = Z.shape
    K = torch.empty(M, M, dtype=torch.float64, device=Z.device)
    L.check(lib.tgp_kmm_f64(L.ptr(Z), L.ptr(raw_ls), L.ptr(raw_os), M, D, float(jitter), L.ptr(K), L.stream_ptr()),
            "tgp_kmm_f64")
    return K


def knm(X, Z, raw_ls, raw_os):
    lib = L.load()
    X, Z, raw_ls, raw_os = _c(X, "X"), _c(Z, "Z"), _c(raw_ls, "raw_ls"), _c(raw_os, "raw_os")
    N, D = X.shape
    M = Z.shape[0]
    K = torch.empty(N, M, dtype=torch.float64, device=X.device)
    L.check(lib.tgp_knm_f64(L.ptr(X), L.ptr(Z), L.ptr(raw_ls), L.ptr(raw_os), N, M, D, L.ptr(K), L.stream_ptr()),
            "tgp_knm_f64")
    return K


def cholesky(A, want_inverse=False):
    """Lower Cholesky with LAPACK-style info (no jitter ladder here): returns (L, Linv or None, status)."""
    lib = L.load()
    A = _c(A, "A")
    M = A.shape[0]
    Lo = torch.empty_like(A)
    Li = torch.empty_like(A) if want_inverse else None
    status = torch.zeros(8, dtype=torch.int32, device=A.device)
    nws = lib.tgp_cholesky_workspace_bytes(M)
    ws = torch.empty(nws // 8 + 16, dtype=torch.float64, device=A.device) if nws else None
    L.check(lib.tgp_cholesky_f64(L.ptr(A), M, L.ptr(Lo), L.ptr(Li), L.ptr(status), L.ptr(ws),
                                 ws.numel() * 8 if ws is not None else 0, L.stream_ptr()), "tgp_cholesky_f64")
    return Lo, Li, status


def cholesky_bwd(Lo, Li, L_bar):
    """Adjoint of `cholesky` (tgp_cholesky_bwd_f64): the symmetric A_bar for a factor adjoint L_bar, given L and L^-1."""
    lib = L.load()
    Lo, Li, L_bar = _c(Lo, "L"), _c(Li, "Linv"), _c(L_bar, "L_bar")
    M = Lo.shape[0]
    A_bar = torch.empty_like(Lo)
    ws = torch.empty(lib.tgp_cholesky_bwd_workspace_bytes(M) // 8 + 16, dtype=torch.float64, device=Lo.device)
    L.check(lib.tgp_cholesky_bwd_f64(L.ptr(Lo), L.ptr(Li), L.ptr(L_bar), M, L.ptr(A_bar), L.ptr(ws), ws.numel() * 8,
                                     L.stream_ptr()), "tgp_cholesky_bwd_f64")
    return A_bar


class CholeskyFunction(torch.autograd.Function):
    """L = chol(A) with autograd (torch.cholesky inside psd_safe_cholesky, dsp/utils.py:239, is differentiable in the
    reference): forward tgp_cholesky_f64 (L and L^-1 in one call), backward tgp_cholesky_bwd_f64.  Raises NotPSDError on a
    non-positive pivot (the jitter ladder is psd_safe_cholesky's)."""

    @staticmethod
    def forward(ctx, A):
        Lo, Li, status = cholesky(A, want_inverse=True)
        if raise_for_status(status.cpu()):
            raise NotPSDError("matrix not positive definite (pivot %d)" % int(status[0]))
        ctx.save_for_backward(Lo, Li)
        return Lo

    @staticmethod
    def backward(ctx, g):
        Lo, Li = ctx.saved_tensors
        return cholesky_bwd(Lo, Li, g)


def psd_safe_cholesky(A, jitter=None):
    """dsp/utils.py:222-270 on the GPU: returns (L, A_used).  With autograd on and A requiring grad the factor carries a
    gradient to A (CholeskyFunction), as the reference's torch.cholesky does."""
    diff = torch.is_grad_enabled() and A.requires_grad

    def attempt(Ax):
        if diff:
            try:
                return CholeskyFunction.apply(Ax), False
            except NotPSDError:
                return None, True
        Lo, _, status = cholesky(Ax)
        return Lo, raise_for_status(status.cpu())

    Lo, bad = attempt(A)
    if not bad:
        return Lo, A
    Ap = A.clone()          # (a non-leaf copy: the in-place diagonal updates below are autograd-safe)
    prev = 0.0
    for jit in jitter_ladder(A.dtype, jitter):
        Ap.diagonal().add_(jit - prev)
        prev = jit
        Lo, bad = attempt(Ap)
        if not bad:
            warnings.warn("A not p.d., added jitter of %g to the diagonal" % jit, NumericalWarning)
            return Lo, Ap
    raise NotPSDError("matrix not positive definite even with jitter %g" % prev)


TRI_A_LOWER, TRI_A_UPPER, TRI_B_LOWER, TRI_B_UPPER, TRI_C_LOWER = 1, 2, 4, 8, 16


def gemm(A, B, trans_a=False, trans_b=False, alpha=1.0, beta=0.0, C=None, tri=0):
    """C = alpha op(A) op(B) + beta C on the float64 matrix cores (tgp_gemm_f64); m, n multiples of 128, k of 16.

    The building block of the M > 128 path: stands in for the reference's torch.bmm / triangular_solve on
    (M,M)x(M,N) operands (models/sparse_MF_SP.py:354,376-382).  `tri` declares triangular operands (TRI_*)."""
    A, B = _c(A, "A"), _c(B, "B")
    m, k = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    k2, n = (B.shape[1], B.shape[0]) if trans_b else (B.shape[0], B.shape[1])
    if k != k2:
        raise L.TgpError("gemm: inner dimensions differ (%d vs %d)" % (k, k2))
    if C is None:
        C = torch.zeros(m, n, dtype=torch.float64, device=A.device)
    L.check(L.load().tgp_gemm_f64(int(trans_a), int(trans_b), int(tri), m, n, k, float(alpha), L.ptr(A), A.shape[1],
                                  L.ptr(B), B.shape[1], float(beta), L.ptr(C), C.shape[1], L.stream_ptr()),
            "tgp_gemm_f64")
    return C


def kl_whitened(m, Lam):
    """Whitened KL and gradients (models/sparse_MF_SP.py:406-431): returns (KL 0-d, g_m, g_Lam)."""
    lib = L.load()
    m, Lam = _c(m, "m"), _c(Lam, "Lam")
    out = torch.empty(1, dtype=torch.float64, device=m.device)
    gm, gL = torch.empty_like(m), torch.empty_like(Lam)
    L.check(lib.tgp_kl_whitened_f64(L.ptr(m), L.ptr(Lam), m.numel(), L.ptr(out), L.ptr(gm), L.ptr(gL),
                                    L.stream_ptr()), "tgp_kl_whitened_f64")
    return out[0], gm, gL


def ell_gauss(Y, mu, v, lvn, scale=1.0):
    """SVGP expected log-likelihood (likelihoods/GaussianLinearMean.py:60-87): (ELL, dELL/dlvn, g_mu, g_v)."""
    lib = L.load()
    Y, mu, v, lvn = _c(Y.reshape(-1), "Y"), _c(mu, "mu"), _c(v, "v"), _c(lvn, "lvn")
    N = Y.numel()
    ws = torch.empty(lib.tgp_ell_workspace_bytes(N, 0, 0) // 8 + 16, dtype=torch.float64, device=Y.device)
    out = torch.empty(2, dtype=torch.float64, device=Y.device)
    gmu, gv = torch.empty_like(mu), torch.empty_like(v)
    L.check(lib.tgp_ell_gauss_f64(L.ptr(Y), L.ptr(mu), L.ptr(v), N, L.ptr(lvn), float(scale), L.ptr(out), L.ptr(gmu),
                                  L.ptr(gv), L.ptr(ws), ws.numel() * 8, L.stream_ptr()), "tgp_ell_gauss_f64")
    return out[0], out[1], gmu, gv


def _flow_model(N, S, flow, theta, lvn, dev, scale=1.0, lik=L.LIK_FLOW):
    md = L.TgpModel()
    md.N, md.D, md.M, md.S = N, 1, 1, int(S)
    md.nblk, md.P, md.RP, md.lik = flow.nblk, flow.P, flow.RP, lik
    md.scale, md.jitter, md.kl_scale = float(scale), 0.0, 1.0
    xs, wn = gauss_hermite(S, dev)
    md.program, md.xs, md.wn = flow.program_ptr, L.ptr(xs), L.ptr(wn)
    md.theta = L.ptr(theta) if flow.P > 0 else None
    md.log_var_noise = L.ptr(lvn)
    return md, (xs, wn)


def ell_flow(Y, mu, v, lvn, flow, theta, S, rowp=None, scale=1.0):
    """TGP quadrature ELL with gradients (likelihoods/GaussianNonLinearMean.py:64-150).
    Returns dict(ell, g_lvn, g_mu, g_v, g_theta, g_rowp)."""
    lib = L.load()
    Y, mu, v, lvn = _c(Y.reshape(-1), "Y"), _c(mu, "mu"), _c(v, "v"), _c(lvn, "lvn")
    theta, rowp = _c(theta, "theta"), _c(rowp, "rowp")
    dev, N = Y.device, Y.numel()
    md, keep = _flow_model(N, S, flow, theta, lvn, dev, scale)
    ws = torch.empty(lib.tgp_ell_workspace_bytes(N, flow.P, md.RP) // 8 + 16, dtype=torch.float64, device=dev)
    out = torch.empty(2, dtype=torch.float64, device=dev)
    gmu, gv = torch.empty_like(mu), torch.empty_like(v)
    gth = torch.empty(max(flow.P, 1), dtype=torch.float64, device=dev)
    grp = torch.empty_like(rowp) if rowp is not None else None
    L.check(lib.tgp_ell_flow_f64(md, L.ptr(Y), L.ptr(mu), L.ptr(v), L.ptr(rowp), L.ptr(out), L.ptr(gmu), L.ptr(gv),
                                 L.ptr(gth), L.ptr(grp), L.ptr(ws), ws.numel() * 8, L.stream_ptr()),
            "tgp_ell_flow_f64")
    return {"ell": out[0], "g_lvn": out[1], "g_mu": gmu, "g_v": gv, "g_theta": gth[:flow.P], "g_rowp": grp}


class EllGaussFunction(torch.autograd.Function):
    """ELL = GaussianLinearMean.expected_log_prob (likelihoods/GaussianLinearMean.py:60-87) with autograd in (mu, v,
    log_var_noise) -- the reference's method is plain torch code, differentiable wherever it is called; tgp_ell_gauss_f64
    returns the value and every gradient in one launch."""

    @staticmethod
    def forward(ctx, Y, mu, v, lvn):
        ell, g_lvn, gmu, gv = ell_gauss(Y, mu.detach(), v.detach(), lvn.detach())
        ctx.save_for_backward(g_lvn, gmu, gv)
        ctx.lvn_shape = lvn.shape
        return ell.reshape(1)

    @staticmethod
    def backward(ctx, g):
        g_lvn, gmu, gv = ctx.saved_tensors
        g = g.reshape(())
        return None, g * gmu, g * gv, (g * g_lvn).reshape(ctx.lvn_shape)


class EllFlowFunction(torch.autograd.Function):
    """ELL = GaussianNonLinearMean.expected_log_prob (likelihoods/GaussianNonLinearMean.py:64-150) with autograd in (mu, v,
    log_var_noise, theta, rowp): tgp_ell_flow_f64 returns all of them."""

    @staticmethod
    def forward(ctx, Y, mu, v, lvn, theta, rowp, flow, S):
        res = ell_flow(Y, mu.detach(), v.detach(), lvn.detach(), flow, theta.detach() if theta is not None else None, S,
                       rowp.detach() if rowp is not None else None)
        ctx.save_for_backward(res["g_lvn"], res["g_mu"], res["g_v"], res["g_theta"],
                              res["g_rowp"] if res["g_rowp"] is not None else res["g_lvn"])
        ctx.has = (theta is not None, rowp is not None)
        ctx.lvn_shape = lvn.shape
        return res["ell"].reshape(1)

    @staticmethod
    def backward(ctx, g):
        g_lvn, gmu, gv, gth, grp = ctx.saved_tensors
        g = g.reshape(())
        return (None, g * gmu, g * gv, (g * g_lvn).reshape(ctx.lvn_shape), g * gth if ctx.has[0] else None,
                g * grp if ctx.has[1] else None, None, None)


def flow_eval(f, flow, theta, rowp=None, want=("G", "dG", "logdG")):
    """G(f), dG/df, log dG/df for f of shape (S,N) or (N,) (CompositeFlow.forward / forward_grad)."""
    lib = L.load()
    f = _c(f, "f")
    theta, rowp = _c(theta, "theta"), _c(rowp, "rowp")
    f2 = f.reshape(1, -1) if f.dim() == 1 else f
    S, N = f2.shape
    lvn = torch.zeros(1, dtype=torch.float64, device=f.device)
    md, keep = _flow_model(N, 1, flow, theta, lvn, f.device)
    outs = {k: (torch.empty_like(f) if k in want else None) for k in ("G", "dG", "logdG")}
    L.check(lib.tgp_flow_eval_f64(md, L.ptr(f2), S, N, L.ptr(rowp), L.ptr(outs["G"]), L.ptr(outs["dG"]),
                                  L.ptr(outs["logdG"]), L.stream_ptr()), "tgp_flow_eval_f64")
    return outs


def flow_logdet(f, flow, theta, rowp=None, want_G=False):
    """sum log dG/df over f (S,N) or (N,) in one fused pass (tgp_flow_logdet_f64); returns (sum 0-d, G or None)."""
    lib = L.load()
    f = _c(f, "f")
    theta, rowp = _c(theta, "theta"), _c(rowp, "rowp")
    f2 = f.reshape(1, -1) if f.dim() == 1 else f
    S, N = f2.shape
    lvn = torch.zeros(1, dtype=torch.float64, device=f.device)
    md, keep = _flow_model(N, 1, flow, theta, lvn, f.device)
    G = torch.empty_like(f) if want_G else None
    out = torch.empty(1, dtype=torch.float64, device=f.device)
    ws = torch.empty(lib.tgp_flow_logdet_workspace_bytes(S, N) // 8, dtype=torch.float64, device=f.device)
    L.check(lib.tgp_flow_logdet_f64(md, L.ptr(f2), S, N, L.ptr(rowp), L.ptr(G), L.ptr(out), L.ptr(ws), ws.numel() * 8,
                                    L.stream_ptr()), "tgp_flow_logdet_f64")
    return out[0], G


def predict(mu, v, lvn, flow=None, theta=None, S=None, rowp=None, Y=None, Y_std=1.0):
    """Predictive moments m1, m2 and per-row test log-likelihood kernel (see tgp_predict_f64)."""
    lib = L.load()
    mu, v, lvn = _c(mu, "mu"), _c(v, "v"), _c(lvn, "lvn")
    theta, rowp = _c(theta, "theta"), _c(rowp, "rowp")
    dev, N = mu.device, mu.numel()
    if flow is None:
        md = L.TgpModel()
        md.N, md.D, md.M, md.S, md.lik = N, 1, 1, 1, L.LIK_GAUSS
        md.log_var_noise = L.ptr(lvn)
        keep = None
    else:
        md, keep = _flow_model(N, S, flow, theta, lvn, dev)
    m1, m2 = torch.empty_like(mu), torch.empty_like(mu)
    logp = torch.empty_like(mu) if Y is not None else None
    Yc = _c(Y.reshape(-1), "Y") if Y is not None else None
    L.check(lib.tgp_predict_f64(md, L.ptr(mu), L.ptr(v), L.ptr(rowp), L.ptr(Yc), float(Y_std), L.ptr(m1), L.ptr(m2),
                                L.ptr(logp), L.stream_ptr()), "tgp_predict_f64")
    return m1, m2, logp


# ---------------------------------------------------------------------------------------------------
# per-row parameter networks of the input-dependent flows (models/flow.py:836-897)
# ---------------------------------------------------------------------------------------------------
class MlpSpec:
    """nnets MLPs of one architecture D -> H x L -> 1 (Linear -> act -> Dropout per hidden layer)."""

    def __init__(self, D, H, L, nnets, act="relu", drop_p=0.0, seed=0):
        self.D, self.H, self.L, self.nnets = int(D), int(H), int(L), int(nnets)
        self.act = {"relu": 0, "tanh": 1}[act]
        self.drop_p, self.seed = float(drop_p), int(seed)

    def salted(self, salt):
        """The same networks with the dropout-mask stream of another call site (seed ^ salt): masks are a hash of (seed,
        step, net, layer, row, unit) and every call site counts its own steps from 0, so without a salt the masks of
        evaluation call k would be those of training step k."""
        return MlpSpec(self.D, self.H, self.L, self.nnets, act={0: "relu", 1: "tanh"}[self.act], drop_p=self.drop_p,
                       seed=self.seed ^ int(salt))

    @property
    def weights_per_net(self):
        return self.D * self.H + self.H + (self.L - 1) * (self.H * self.H + self.H) + self.H + 1

    def lds_bytes(self):
        """LDS image of the backward kernel (tgp_mlp.hip mlp_lds): padded weights + activation strips."""
        kp0, kph = (self.D + 3) // 4 * 4, (self.H + 3) // 4 * 4
        w = kph * kp0 + kph + (self.L - 1) * (kph * kph + kph) + kph + 2
        n = w + (kp0 + self.L * kph) * 65           # strips [unit][64 rows + 1]
        return ((n + 1) // 2 * 2 + 64) * 8           # + d out of the block's 64 rows

    def struct(self, N, training):
        d = L.TgpMlp()
        d.N, d.D, d.H, d.L, d.nnets, d.act = int(N), self.D, self.H, self.L, self.nnets, self.act
        d.training, d.drop_p, d.seed = int(bool(training)), self.drop_p, self.seed
        return d


def mlp_forward(spec, X, W, training=False, step_dev=None):
    """out (N, nnets) = the nets' outputs for every row (tgp_mlp_forward_f64)."""
    X, W = _c(X, "X"), _c(W, "W")
    d = spec.struct(X.shape[0], training)
    out = torch.empty(X.shape[0], spec.nnets, dtype=torch.float64, device=X.device)
    L.check(L.load().tgp_mlp_forward_f64(d, L.ptr(X), L.ptr(W), L.ptr(step_dev), L.ptr(out), L.stream_ptr()),
            "tgp_mlp_forward_f64")
    return out


_mlp_ws = {}


def mlp_backward(spec, X, W, g_out, training=False, step_dev=None, g_W=None):
    """d(objective)/dW (packed like W) from g_out (N, nnets) (tgp_mlp_backward_f64; recomputes the forward)."""
    X, W, g_out = _c(X, "X"), _c(W, "W"), _c(g_out, "g_out")
    lib = L.load()
    d = spec.struct(X.shape[0], training)
    key = (X.shape[0], spec.D, spec.H, spec.L, spec.nnets, str(X.device), torch.cuda.current_stream().cuda_stream)
    ws = _mlp_ws.get(key)
    if ws is None:
        ws = torch.empty(lib.tgp_mlp_workspace_bytes(d) // 8 + 16, dtype=torch.float64, device=X.device)
        _mlp_ws[key] = ws
    if g_W is None:
        g_W = torch.empty_like(W)
    L.check(lib.tgp_mlp_backward_f64(d, L.ptr(X), L.ptr(W), L.ptr(step_dev), L.ptr(g_out), L.ptr(g_W), L.ptr(ws),
                                     ws.numel() * 8, L.stream_ptr()), "tgp_mlp_backward_f64")
    return g_W


class MlpFunction(torch.autograd.Function):
    """rowp = MLPs(X; W) with the HIP forward/backward; W is the packed weight vector (torch.cat of the nets'
    parameters, so autograd scatters g_W back onto the individual nn.Parameters)."""

    @staticmethod
    def forward(ctx, X, W, spec, training, step_dev):
        ctx.spec, ctx.training, ctx.step_dev = spec, training, step_dev
        ctx.save_for_backward(X, W)
        return mlp_forward(spec, X, W.detach(), training, step_dev)

    @staticmethod
    def backward(ctx, g_out):
        X, W = ctx.saved_tensors
        return None, mlp_backward(ctx.spec, X, W.detach(), g_out.contiguous(), ctx.training, ctx.step_dev), None, None, None


MASK_SALT_EVAL = 0x45564131     # model-class evaluation (test_log_likelihood, predictive moments)
MASK_SALT_NETS = 0x4E455453     # flow.nets_rowp (CompositeFlow.forward, sampling)


def mlp_keep_mask(seed, step, net, layer, rows, units, p):
    """The dropout keep mask of tgp_mlp.hip (rows x units, bool) restated in numpy: test infrastructure and the
    reference for anyone who needs to reproduce a training-mode forward elsewhere."""
    import numpy as np
    M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
    r = np.arange(rows, dtype=np.uint64).reshape(-1, 1)
    j = np.arange(units, dtype=np.uint64)
    ug = ((j >> np.uint64(4)) * np.uint64(4) + (j & np.uint64(3))).reshape(1, -1)    # group: the 4 accumulator regs of a lane
    lane = ((j >> np.uint64(2)) & np.uint64(3)).reshape(1, -1)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.uint64(step)) & M64
        z = z ^ (np.uint64(net) << np.uint64(56)) ^ (np.uint64(layer) << np.uint64(48)) ^ (ug << np.uint64(32)) ^ r
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        z = z ^ (z >> np.uint64(31))
    bits = (z >> (np.uint64(16) * lane)) & np.uint64(0xFFFF)
    return bits >= np.uint64(int(p * 65536.0 + 0.5))


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr=0.01, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
              maximize=False):
    """In-place Adam on flat float64 buffers (torch.optim.Adam semantics)."""
    lib = L.load()
    L.check(lib.tgp_adam_f64(L.ptr(params), L.ptr(grads), L.ptr(exp_avg), L.ptr(exp_avg_sq), params.numel(), lr,
                             betas[0], betas[1], eps, weight_decay, int(step), int(bool(maximize)), L.stream_ptr()),
            "tgp_adam_f64")
