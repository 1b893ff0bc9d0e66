"""The 4-rows-per-wave row kernel (csrc/tgp_rows4.hpp, v_mfma_f64_4x4x4_4b) against the 16-rows-per-wave one on the same
inputs: which of the two a training launch gets is decided inside the library (row blocks + passenger blocks at most one per
CU -> k_rows4); `tgp_model.plan` (lib.PLAN_ROWS_K: k_rows with its own rows-per-wave rule, PLAN_ROWS_K16: k_rows at 16 rows per
wave) overrides that choice for ONE call, so both sides run in this process.  Also: bit reproducibility of k_rows4, the oracle
at a size where k_rows4 runs with 8-wave workgroups, and k_rows<.., 10> against k_rows<.., 16> on the same inputs."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

CASES = [(455, 13, 5, "tanh3x2", 32), (1077, 4, 100, "tanh3x2", 32), (2153, 4, 100, "sal2", 32), (4306, 4, 100, "tanh3x2", 32),
         (1000, 8, 37, "idsal3", 20), (3001, 16, 128, "sal2", 32), (37, 3, 16, "tanh3x2", 7),
         (3968, 4, 100, "tanh3x2", 32),      # the last size with 4-wave workgroups at MT = 7 (248 row blocks + 7 passengers)
         (7936, 4, 100, "sal2", 32),         # ... with 8-wave workgroups
         (3984, 4, 100, "tanh3x2", 32),      # one block more: 8-wave workgroups
         (2153, 4, 100, None, 32), (6000, 8, 64, None, 32)]      # closed-form likelihood (SVGP)


def _run(case, plan=0):
    from tgp.pytorch_amd import ops, synthetic
    N, D, M, flow, S = case
    dev = torch.device("cuda:0")
    prob = synthetic.synthetic_problem(N, D, M, seed=11, flow=flow, S=S)
    p = {k: v.to(dev) for k, v in prob["params"].items()}
    rowp = prob["rowp"].to(dev) if prob["rowp"] is not None else None
    fs = ops.FlowSpec(prob["program"], p["theta"].numel(), 0 if rowp is None else rowp.shape[1], dev) if flow else None
    out, g, status, (mu, v) = ops.elbo_step(prob["X"].to(dev), prob["Y"].to(dev), p["Z"], p["raw_lengthscale"],
                                            p["raw_outputscale"], p["m"], p["Lam"], p["log_var_noise"], prob["N_total"],
                                            flow=fs, theta=p.get("theta"), rowp=rowp, S=S, want_moments=True, plan=plan)
    torch.cuda.synchronize()
    assert int(status[0]) == 0 and int(status[1]) == 0
    res = {"out": out.cpu(), "mu": mu.cpu(), "v": v.cpu()}
    res.update({"g_" + k: t.cpu() for k, t in g.items() if t is not None})
    return res


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "N%d_D%d_M%d_%s_S%d" % c)
def test_rows4_matches_rows16_and_is_reproducible(case):
    from tgp.pytorch_amd import lib
    r16 = _run(case, plan=lib.PLAN_ROWS_K)          # k_rows (rows per wave by its own rule), same process, same inputs
    a, b = _run(case), _run(case)                   # the library's choice at these sizes: k_rows4
    for k in a:
        assert torch.equal(a[k], b[k]), ("k_rows4 is not bit-reproducible", case, k)
        tol = 1e-10 if k in ("out", "mu", "v") else 1e-8
        assert _rel(a[k], r16[k]) < tol, (case, k, _rel(a[k], r16[k]))


# k_rows<.., RW = 10> (selected for 7 936 < N <= 10 240 with a flow likelihood whose 10 S (row, node) pairs fill the lanes in one
# trip) against k_rows<.., 16> forced by plan: the two tilings share every tile chain and differ in the quadrature's lane
# assignment and in the statistics' contraction length (the selection window and its edges)
RW_CASES = [(8611, 4, 100, "tanh3x2", 32), (7937, 4, 100, "tanh3x2", 32), (10240, 8, 64, "sal2", 32), (9000, 4, 100, "idsal3", 20),
            (8000, 3, 16, "tanh1x1", 8), (8200, 16, 37, "sal2", 32),
            (9000, 13, 128, "idsal3", 20)]     # MT = 8, DP = 16 with a 9-slot stack: the 10-row LDS plan does not fit -> 16 rows either way
RW_FALLS_BACK = {(9000, 13, 128, "idsal3", 20)}


@pytest.mark.parametrize("case", RW_CASES, ids=lambda c: "N%d_D%d_M%d_%s_S%d" % c)
def test_rows10_matches_rows16_and_is_reproducible(case):
    from tgp.pytorch_amd import lib
    r16 = _run(case, plan=lib.PLAN_ROWS_K16)
    a, b = _run(case), _run(case)
    for k in a:
        assert torch.equal(a[k], b[k]), ("k_rows<.., 10> is not bit-reproducible", case, k)
        tol = 1e-10 if k in ("out", "mu", "v") else 1e-8
        assert _rel(a[k], r16[k]) < tol, (case, k, _rel(a[k], r16[k]))
    differs = any(not torch.equal(a[k], r16[k]) for k in a)
    assert differs != (case in RW_FALLS_BACK), "plan=PLAN_ROWS_K16 and the automatic choice ran %s kernel" % ("the same" if not differs else "different")


def test_rows4_against_the_oracle_at_a_two_gpu_shard_size():
    """N = 4306 (half of Power: 8-wave workgroups of k_rows4) against the CPU oracle: values 1e-9, gradients 1e-7."""
    from oracle import tgp_oracle as orc       # checker only
    from tgp.pytorch_amd import ops
    dev = torch.device("cuda:0")
    prob = orc.synthetic_problem(4306, 4, 100, seed=5, flow="tanh3x2", S=32)
    (elbo, ell, kld), og = orc.elbo_and_grads(prob["X"], prob["Y"], prob["params"], prob["N_total"], prob["program"],
                                              prob["xs"], prob["ws"])
    p = {k: v.to(dev) for k, v in prob["params"].items()}
    flow = ops.FlowSpec(prob["program"], p["theta"].numel(), 0, dev)
    out, g, status, _ = ops.elbo_step(prob["X"].to(dev), prob["Y"].to(dev), p["Z"], p["raw_lengthscale"],
                                      p["raw_outputscale"], p["m"], p["Lam"], p["log_var_noise"], prob["N_total"],
                                      flow=flow, theta=p["theta"], S=32)
    torch.cuda.synchronize()
    assert int(status[0]) == 0
    assert _rel(out[0].cpu(), elbo) < 1e-9 and _rel(out[1].cpu(), ell) < 1e-9
    for k_hip, k_or in (("Z", "Z"), ("raw_ls", "raw_lengthscale"), ("raw_os", "raw_outputscale"), ("m", "m"),
                        ("Lam", "Lam"), ("lvn", "log_var_noise"), ("theta", "theta")):
        assert _rel(g[k_hip].cpu(), og[k_or]) < 1e-7, (k_hip, _rel(g[k_hip].cpu(), og[k_or]))
