"""Seeded synthetic workloads of the dataset shapes main.py trains on: what bench.py feeds the
engine and what a user without the UCI files can train on.  Product code, self-contained: the parity tests'
own generator is held to the same draws by tests/test_host_logic.py."""
import math

import numpy as np
import torch

from . import lib as L


def _isp(x):
    """Inverse softplus in float64 whatever the default dtype is (gpytorch.utils.transforms.inv_softplus)."""
    x = torch.as_tensor(x, dtype=torch.float64)
    return x + torch.log(-torch.expm1(-x))


class FlowProgram:
    """Accumulates (kind, K, offset, flags) blocks and the shared parameter vector they index."""

    def __init__(self):
        self.blocks, self.theta, self.row_cols = [], [], 0

    def affine(self):                       # dsp/flows.py: every block ends in an identity-initialised affine
        self.blocks.append((L.FLOW_AFFINE, 0, len(self.theta), 0))
        self.theta += [1.0, 0.0]

    def sal(self, per_row):                 # dsp/flows.py:115-136 (a=0, b=1 is the identity)
        if per_row:
            self.blocks.append((L.FLOW_SAL, 0, self.row_cols, L.FLAG_PER_ROW))
            self.row_cols += 2
        else:
            self.blocks.append((L.FLOW_SAL, 0, len(self.theta), 0))
            self.theta += [0.0, 1.0]

    def steptanh(self, K, rng):             # dsp/flows.py:239-277 StepTanhL initialisation, add_f0=True
        self.blocks.append((L.FLOW_STEPTANH, K, len(self.theta), L.FLAG_ADD_F0))
        for _ in range(K):
            e = rng.standard_normal(4)
            self.theta += [e[0], float(_isp(abs((e[1] + 1.0) / K))), e[2], float(_isp(abs((e[3] + 1.0) / K)))]

    def vector(self):
        return torch.tensor(self.theta, dtype=torch.float64)


def flow_program(flow, seed=0):
    """'sal<B>' | 'idsal<B>' | 'tanh<B>x<K>' | None -> (blocks, theta, per-row columns)."""
    if flow is None:
        return None, None, 0
    fp = FlowProgram()
    if flow.startswith("tanh"):
        nb, K = (int(t) for t in flow[4:].split("x"))
        rng = np.random.default_rng(seed)
        for _ in range(nb):
            fp.steptanh(K, rng)
            fp.affine()
    else:
        per_row = flow.startswith("idsal")
        for _ in range(int(flow[5 if per_row else 3:])):
            fp.sal(per_row)
            fp.affine()
    return fp.blocks, fp.vector(), fp.row_cols


def synthetic_problem(N, D, M, seed=0, flow="sal2", S=32, perturb=True):
    """X ~ N(0,1); Y = zscore(sin(Xw) + 0.1 x0^2 + 0.05 eps); Z = M rows of a seeded permutation; lengthscale 2,
    outputscale 2, noise 0.05; q(u): m ~ 0.5 N(0,1), Lq = sqrt(1e-5) I + 0.05 N(0,1) (dense -- the strict upper
    part must be ignored by the consumer); SAL parameters = identity + 0.3 N(0,1), StepTanhL at its own init."""
    f64 = torch.float64
    gd = torch.Generator().manual_seed(seed)            # data stream
    X = torch.randn(N, D, generator=gd, dtype=f64)
    w = torch.randn(D, generator=gd, dtype=f64)
    Y = torch.sin(X @ w) + 0.1 * X[:, 0] ** 2 + 0.05 * torch.randn(N, generator=gd, dtype=f64)
    Y = ((Y - Y.mean()) / Y.std()).reshape(N, 1)
    Z = X[torch.randperm(N, generator=gd)[:M]].clone()

    gp = torch.Generator().manual_seed(seed + 1)        # parameter stream (draw order matters)
    m = torch.zeros(M, dtype=f64)
    Lam = math.sqrt(1e-5) * torch.eye(M, dtype=f64)
    rls = _isp(torch.full((D,), 2.0))
    if perturb:
        m = 0.5 * torch.randn(M, generator=gp, dtype=f64)
        Lam = Lam + 0.05 * torch.randn(M, M, generator=gp, dtype=f64)
        rls = rls + 0.1 * torch.randn(D, generator=gp, dtype=f64)
    params = {"Z": Z, "raw_lengthscale": rls, "raw_outputscale": _isp(2.0).reshape(1), "m": m, "Lam": Lam,
              "log_var_noise": torch.log(torch.tensor([0.05], dtype=f64))}

    program, theta, row_cols = flow_program(flow, seed)
    rowp = None
    if program is not None:
        if perturb and not flow.startswith("tanh"):
            theta = theta + 0.3 * torch.randn(theta.shape, generator=gp, dtype=f64)
        params["theta"] = theta
        if row_cols:
            ident = torch.tensor([0.0, 1.0] * (row_cols // 2), dtype=f64)
            rowp = ident.reshape(1, -1) + 0.2 * torch.randn(N, row_cols, generator=gp, dtype=f64)
    x, wq = np.polynomial.hermite.hermgauss(int(S))
    return {"X": X, "Y": Y, "params": params, "program": program, "xs": torch.tensor(x, dtype=f64),
            "ws": torch.tensor(wq, dtype=f64), "rowp": rowp, "N_total": float(N)}
