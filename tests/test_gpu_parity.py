"""GPU (-m gpu): the HIP path, called through the C ABI, against (a) the golden fixtures produced by
executing the reference and (b) the oracle on fresh seeded problems.  Tolerances are relative to the
largest entry of each tensor; float64 end to end (the reference's main.py mode); north_star asks 1e-5."""
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL_VAL = 1e-9
TOL_GRAD = 1e-7   # cond(K_MM) ~ 1e7 amplifies rounding in the reference and here alike

FIXTURES = ["tiny_svgp", "tiny_sal2", "tiny_tanh3x2", "tiny_idsal3", "ragged_sal2", "boston_like_svgp", "med_svgp",
            "med_sal2", "med_tanh3x2", "init_sal2_identity", "init_svgp"]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def run_hip(g, want_moments=True, **kw):
    from tgp.pytorch_amd import ops
    dev = _dev()
    p = {k: v.to(dev) for k, v in g["params"].items()}
    flow = None
    theta = rowp = None
    S = None
    if g["program"] is not None:
        RP = g["rowp"].shape[1] if g.get("rowp") is not None else 0
        flow = ops.FlowSpec(g["program"], p["theta"].numel(), RP, dev)
        theta = p["theta"]
        rowp = g["rowp"].to(dev) if g.get("rowp") is not None else None
        S = g["xs"].numel()
    out, grads, status, (mu, v) = ops.elbo_step(g["X"].to(dev), g["Y"].to(dev), p["Z"], p["raw_lengthscale"],
                                                 p["raw_outputscale"], p["m"], p["Lam"], p["log_var_noise"],
                                                 float(g["N_total"]), flow=flow, theta=theta, rowp=rowp, S=S,
                                                 want_moments=want_moments, kernel=g.get("kernel", "scale_rbf"), **kw)
    torch.cuda.synchronize()
    return out.cpu(), {k: t.cpu() for k, t in grads.items()}, status.cpu(), (mu.cpu(), v.cpu())


def compare(out, grads, g, tol_val=TOL_VAL, tol_grad=TOL_GRAD):
    assert rel_err(out[0], g["ELBO"]) < tol_val
    assert rel_err(out[1], g["ELL"]) < tol_val
    assert rel_err(out[2], g["KLD"]) < tol_val
    names = {"Z": "g_Z", "raw_ls": "g_raw_lengthscale", "raw_os": "g_raw_outputscale", "m": "g_m", "Lam": "g_Lam",
             "lvn": "g_log_var_noise", "theta": "g_theta", "rowp": "g_rowp"}
    for k, t in grads.items():
        assert rel_err(t, g[names[k]]) < tol_grad, (k, rel_err(t, g[names[k]]))


@pytest.mark.parametrize("name", FIXTURES)
def test_elbo_step_matches_reference_fixture(name):
    g = load_golden(name)
    out, grads, status, (mu, v) = run_hip(g)
    assert int(status[0]) == 0 and int(status[1]) == 0
    compare(out, grads, g)
    assert rel_err(mu, g["mu"]) < 1e-9
    assert float(((v - g["v"]).abs() / g["v"].abs()).max()) < 1e-6     # v itself carries a 1e5-fold cancellation at init
    # strict upper triangle of Lam gets exactly zero gradient (tril mask at use)
    assert float(torch.triu(grads["Lam"], 1).abs().max()) == 0.0


@pytest.mark.parametrize("N,D,M,flow,S", [(300, 4, 100, "sal2", 32), (1000, 8, 37, "tanh3x2", 32), (129, 13, 5, None, 8),
                                           (513, 6, 128, "sal1", 20), (64, 16, 16, "idsal3", 32), (9, 3, 8, "sal2", 8), (65, 3, 17, "tanh1x1", 5),
                                           (200, 13, 5, "tanh10x2", 20), (300, 8, 100, "tanh5x6", 32), (150, 4, 30, "sal3", 100),
                                           # every tile count of the factorisation's panel schedule (MT = 4, 5, 6) and
                                           # M = 128 with D > 8 (no room for Zs in the factorisation block's LDS)
                                           (100, 3, 60, None, 8), (250, 5, 70, "sal1", 8), (250, 4, 90, None, 8),
                                           (200, 13, 128, "sal2", 16),
                                           # a flow stack that does not fit a CU's LDS beside the fused row kernel's tiles
                                           # (35 slots, 120 parameters at MT = 8): the step runs on the general-M path
                                           (767, 5, 127, "tanh5x6", 8),
                                           # the selection window of k_rows<.., RW = 10> (7 936 < N <= 10 240 at 10 S <= 320) and its
                                           # edges: first size above k_rows4's range; the window's last size and the
                                           # first beyond it (-> 16 rows per wave); per-row SAL at MT = 8, D = 13; one tile, S = 8;
                                           # S = 33 (10 S > 320 -> 16 rows per wave)
                                           (7937, 4, 100, "tanh3x2", 32), (10240, 8, 64, "sal2", 32), (10241, 8, 64, "sal2", 32),
                                           (9000, 13, 128, "idsal3", 20), (8000, 3, 16, "tanh1x1", 8), (8611, 4, 100, "sal2", 33)])
def test_elbo_step_matches_oracle(N, D, M, flow, S):
    from oracle import tgp_oracle as orc
    prob = orc.synthetic_problem(N, D, M, seed=3, flow=flow, S=S)
    (elbo, ell, kld), og = orc.elbo_and_grads(prob["X"], prob["Y"], prob["params"], prob["N_total"], prob["program"],
                                              prob["xs"], prob["ws"], prob["rowp"])
    g = dict(prob)
    g.update(ELBO=elbo, ELL=ell, KLD=kld, g_Z=og["Z"], g_raw_lengthscale=og["raw_lengthscale"],
             g_raw_outputscale=og["raw_outputscale"], g_m=og["m"], g_Lam=og["Lam"], g_log_var_noise=og["log_var_noise"])
    if "theta" in og:
        g["g_theta"] = og["theta"]
    if "rowp" in og:
        g["g_rowp"] = og["rowp"]
    out, grads, status, _ = run_hip(g)
    assert int(status[0]) == 0
    compare(out, grads, g)
