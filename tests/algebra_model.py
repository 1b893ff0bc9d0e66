"""Stage-by-stage numpy/torch model of the algebra the HIP kernels implement.

Test infrastructure: documents and checks the hand-derived forward/backward restructuring
(no Cholesky-solve per row; a single symmetric operator W and a 3-GEMM Cholesky adjoint) against
the oracle's autograd.  Stage names match the kernels in tgp/pytorch_amd/csrc.
"""
import math

import torch

from oracle import tgp_oracle as orc


def softplus(x):
    return torch.nn.functional.softplus(x)


def prepare(p, jitter=0.0):
    """Stages A-C: everything M x M that the row kernel needs."""
    ls = softplus(p["raw_lengthscale"])
    s2 = softplus(p["raw_outputscale"]).reshape(())
    Zs = p["Z"] / ls
    d2 = ((Zs[:, None, :] - Zs[None, :, :]) ** 2).sum(-1)
    Kmm = s2 * torch.exp(-0.5 * d2)
    M = Kmm.shape[0]
    L = torch.linalg.cholesky(Kmm + jitter * torch.eye(M, dtype=Kmm.dtype))
    J = torch.linalg.solve_triangular(L, torch.eye(M, dtype=Kmm.dtype), upper=False)
    Lq = torch.tril(p["Lam"])
    S = Lq @ Lq.T
    Hp = J.T @ S - J.T                # J^T (S - I): only the backward needs it
    w = J.T @ p["m"]
    kl = 0.5 * (-(torch.log(torch.diagonal(Lq) ** 2)).sum() + p["m"] @ p["m"] + (Lq * Lq).sum() - M)
    return dict(ls=ls, s2=s2, Zs=Zs, Kmm=Kmm, L=L, J=J, Lq=Lq, S=S, Hp=Hp, w=w, kl=kl)


def flow_fwd_bwd(f, program, theta, rowp):
    """G(f), dG/df, and per-parameter partials dG/dtheta_j (dict j -> tensor like f) for one node set."""
    chain_in = []
    local = []           # per block: (g', {param_index: dg/dparam})
    for kind, K, poff, flags in program:
        per_row = bool(flags & orc.FLAG_PER_ROW)

        def P(j):
            return rowp[:, poff + j] if per_row else theta[poff + j]
        chain_in.append(f)
        if kind == orc.FLOW_AFFINE:
            a = P(0)
            fa = torch.ones_like(a)
            if flags & orc.FLAG_RESTRICT:
                fa = torch.sigmoid(a)
                a = softplus(a)
            g = a * f + P(1)
            local.append((a * torch.ones_like(f), {0: f * fa, 1: torch.ones_like(f)}))
        elif kind == orc.FLOW_SAL:
            a, b = P(0), P(1)
            fb = torch.ones_like(b)
            if flags & orc.FLAG_RESTRICT:
                fb = torch.sigmoid(b)
                b = softplus(b)
            u = torch.log(f + torch.sqrt(f * f + 1))
            t = b * u - a
            ch = torch.cosh(t)
            g = torch.sinh(t)
            gp = b * ch / torch.sqrt(1 + f * f)
            if flags & orc.FLAG_ADD_F0:
                g = g + f
                gp = gp + 1
            local.append((gp, {0: -ch, 1: u * ch * fb}))
        else:
            g = f if flags & orc.FLAG_ADD_F0 else torch.zeros_like(f)
            gp = torch.ones_like(f) if flags & orc.FLAG_ADD_F0 else torch.zeros_like(f)
            part = {}
            for k in range(K):
                a, b, c, d = P(4 * k), P(4 * k + 1), P(4 * k + 2), P(4 * k + 3)
                bt, dt = softplus(b), softplus(d)
                t = (f - c) / dt
                th = torch.tanh(t)
                se = 1 - th * th
                g = g + a + bt * th
                gp = gp + bt * se / dt
                part[4 * k] = torch.ones_like(f)
                part[4 * k + 1] = th * torch.sigmoid(b)
                part[4 * k + 2] = -bt * se / dt
                part[4 * k + 3] = -bt * se * t / dt * torch.sigmoid(d)
            local.append((gp, part))
        f = g
    return f, local


def rows(X, Y, st, p, N_total, program, xs, ws, rowp=None):
    """Stage R (+ the slab reduce D): per-row forward/backward and the row-summed statistics."""
    N = X.shape[0]
    c = N_total / N
    Xs = X / st["ls"]
    d2 = ((Xs[:, None, :] - st["Zs"][None, :, :]) ** 2).sum(-1)
    K = st["s2"] * torch.exp(-0.5 * d2)                # (N, M)
    A = st["J"] @ K.T                                  # (M,N)  tri-GEMM 1: A = L^-1 K_MN
    B = st["Lq"].T @ A                                 # (M,N)  tri-GEMM 2
    mu = A.T @ p["m"]
    v = st["s2"] - (A * A).sum(0) + (B * B).sum(0)
    eta = p["log_var_noise"].reshape(())
    e = torch.exp(-eta)
    y = Y.reshape(-1)
    out = {"mu": mu, "v": v}
    if program is None:
        ell = (-0.5 * orc.LOG_2PI_REF - 0.5 * eta - 0.5 * e * ((y - mu) ** 2 + v)).sum()
        mub = c * e * (y - mu)
        vb = -0.5 * c * e * torch.ones_like(v)
        etab = c * (-0.5 + 0.5 * e * ((y - mu) ** 2 + v)).sum()
        thetab = None
    else:
        wn = ws / math.sqrt(math.pi)
        sq = torch.sqrt(2 * v)
        f0 = mu[None, :] + sq[None, :] * xs[:, None]                       # (S,N)
        g, local = flow_fwd_bwd(f0, program, p["theta"], rowp)
        r = y[None, :] - g
        ell = (wn[:, None] * (-0.5 * orc.LOG_2PI_REF - 0.5 * eta - 0.5 * e * r * r)).sum()
        etab = c * (wn[:, None] * (-0.5 + 0.5 * e * r * r)).sum()
        chain = c * e * wn[:, None] * r                                     # d(c*ell)/dG
        thetab = torch.zeros_like(p["theta"])
        rowpb = torch.zeros_like(rowp) if rowp is not None else None
        for (kind, Kk, poff, flags), (gp, part) in zip(reversed(program), reversed(local)):
            for j, dg in part.items():
                if flags & orc.FLAG_PER_ROW:
                    rowpb[:, poff + j] += (chain * dg).sum(0)
                else:
                    thetab[poff + j] += (chain * dg).sum()
            chain = chain * gp
        mub = chain.sum(0)
        vb = (chain * xs[:, None]).sum(0) / sq
        out["rowpb"] = rowpb
    Ab = p["m"][:, None] * mub[None, :] - 2 * A * vb[None, :] + 2 * st["Lq"] @ (B * vb[None, :])   # tri-GEMM 3
    kb = (st["J"].T @ Ab).T                                                 # tri-GEMM 4: dELL/dK_NM (N,M)
    E = kb * K
    out.update(ell=c * ell, etab=etab, thetab=thetab,
               G=(A * vb[None, :]) @ A.T, wb=A @ mub, s2b_direct=vb.sum(),      # G1 = A V A^T, s = A mu_bar
               T0=E.sum(0), T1=E.T @ Xs, T2=E.T @ (Xs * Xs))
    return out


def backward_mm(st, rs, p, kl_scale=1.0):
    """Stages E1-E5: from the row statistics to the parameter gradients of ELBO = ELL - KL."""
    J, Hp, Lq, L, Kmm, Zs, ls, s2 = (st[k] for k in ("J", "Hp", "Lq", "L", "Kmm", "Zs", "ls", "s2"))
    G, sv = rs["G"], rs["wb"]
    Lb = -torch.tril(st["w"][:, None] * sv[None, :] + 2 * Hp @ G)           # E1: dELL/dL
    Lam_b = 2 * torch.tril(G @ Lq)                                           # E1
    m_b = sv
    Pp = torch.tril(L.T @ Lb)                                                # E2
    Pp = Pp - 0.5 * torch.diag(torch.diagonal(Pp))
    Ks = 0.5 * J.T @ (Pp + Pp.T) @ J                                         # E3, E4: symmetric dELL/dKmm
    Ep = Ks * Kmm
    dz = Zs[:, None, :] - Zs[None, :, :]                                     # [i,j,d] = zs_i - zs_j
    zsb = rs["T1"] - Zs * rs["T0"][:, None] + 2 * (Ep[:, :, None] * dz).sum(0)
    lsb_times_ls = (rs["T2"] - 2 * Zs * rs["T1"] + Zs * Zs * rs["T0"][:, None]).sum(0) + (Ep[:, :, None] * dz * dz).sum((0, 1))
    s2b = rs["s2b_direct"] + rs["T0"].sum() / s2 + Ep.sum() / s2
    grads = {
        "Z": zsb / ls,
        "raw_lengthscale": lsb_times_ls / ls * torch.sigmoid(p["raw_lengthscale"]),
        "raw_outputscale": (s2b * torch.sigmoid(p["raw_outputscale"])).reshape(1),
        "m": m_b - kl_scale * p["m"],
        "Lam": Lam_b - kl_scale * (Lq - torch.diag(1.0 / torch.diagonal(p["Lam"]))),
        "log_var_noise": rs["etab"].reshape(1),
    }
    if rs["thetab"] is not None:
        grads["theta"] = rs["thetab"]
    if rs.get("rowpb") is not None:
        grads["rowp"] = rs["rowpb"]
    return grads


def elbo_and_grads(X, Y, p, N_total, program=None, xs=None, ws=None, rowp=None, jitter=0.0):
    st = prepare(p, jitter)
    rs = rows(X, Y, st, p, N_total, program, xs, ws, rowp)
    grads = backward_mm(st, rs, p)
    return (rs["ell"] - st["kl"], rs["ell"], st["kl"]), grads, rs
