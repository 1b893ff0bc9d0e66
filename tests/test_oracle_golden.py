"""CPU: the oracle (oracle/tgp_oracle.py) against fixtures produced by executing the reference
(oracle/gen_golden.py).  This is what pins the oracle; the GPU parity tests then compare the HIP
path with the oracle and with the same fixtures."""
import math

import pytest
import torch

from conftest import load_golden, rel_err
from oracle import tgp_oracle as orc

STEP0 = ["tiny_svgp", "tiny_sal2", "tiny_tanh3x2", "tiny_idsal3", "ragged_sal2", "boston_like_svgp",
         "med_svgp", "med_sal2", "med_tanh3x2", "init_sal2_identity", "init_svgp"]
TOL = 1e-10


@pytest.mark.parametrize("name", STEP0)
def test_step0_matches_reference(name):
    g = load_golden(name)
    (elbo, ell, kld), grads = orc.elbo_and_grads(g["X"], g["Y"], g["params"], float(g["N_total"]), g["program"],
                                                 g["xs"], g["ws"], g.get("rowp"))
    assert rel_err(elbo, g["ELBO"]) < TOL
    assert rel_err(ell, g["ELL"]) < TOL
    assert rel_err(kld, g["KLD"]) < TOL
    p = g["params"]
    mu, v = orc.qf_moments(g["X"], p["Z"], p["raw_lengthscale"], p["raw_outputscale"], p["m"], p["Lam"])
    assert rel_err(mu, g["mu"]) < 1e-9
    assert rel_err(v, g["v"]) < 1e-8          # v carries cond(K_MM) ~ 1e7 amplification in BOTH codes
    for key in ("Z", "m", "Lam", "raw_outputscale", "raw_lengthscale", "log_var_noise"):
        assert rel_err(grads[key], g["g_" + key]) < 1e-8, key
    if g["program"] is not None:
        assert rel_err(grads["theta"], g["g_theta"]) < 1e-8
    if "rowp" in g:
        assert rel_err(grads["rowp"], g["g_rowp"]) < 1e-8
    # strict upper triangle of Lam never receives gradient (tril mask at use, sparse_MF_SP.py:344-345)
    assert float(torch.triu(grads["Lam"], 1).abs().max()) == 0.0


@pytest.mark.parametrize("name", ["tiny_sal2", "tiny_tanh3x2", "med_sal2", "ragged_sal2"])
def test_evaluation_path_matches_reference(name):
    g = load_golden(name)
    p = g["params"]
    m1, m2 = orc.marginal_moments_flow(g["mu"], g["v"], p["log_var_noise"], g["program"], p["theta"], g["xs"], g["ws"])
    assert rel_err(m1, g["pred_m1"]) < TOL
    assert rel_err(m2, g["pred_m2"]) < 1e-9
    lp = orc.test_log_lik_flow(g["Y"].reshape(-1), g["mu"], g["v"], p["log_var_noise"], g["program"], p["theta"],
                               g["xs"], g["ws"], float(g["Y_std"]))
    assert rel_err(orc.test_log_lik_sum_ref(lp), g["test_logp_sum"]) < TOL


@pytest.mark.parametrize("name", ["tiny_matern_svgp", "med_matern_sal2", "med_matern_tanh2x2"])
def test_matern32_step0_matches_reference(name):
    """'scale_matern32' (utils_models.py:199-204): the oracle's restatement against the reference's model classes
    executed on the gpytorch stand-in (third-party kernel arithmetic: unpinned, see oracle/gen_golden.py)."""
    g = load_golden(name)
    assert g["kernel"] == "scale_matern32"
    (elbo, ell, kld), grads = orc.elbo_and_grads(g["X"], g["Y"], g["params"], float(g["N_total"]), g["program"],
                                                 g["xs"], g["ws"], None, kernel=g["kernel"])
    assert rel_err(elbo, g["ELBO"]) < TOL and rel_err(ell, g["ELL"]) < TOL and rel_err(kld, g["KLD"]) < TOL
    p = g["params"]
    mu, v = orc.qf_moments(g["X"], p["Z"], p["raw_lengthscale"], p["raw_outputscale"], p["m"], p["Lam"], kernel=g["kernel"])
    assert rel_err(mu, g["mu"]) < 1e-9 and rel_err(v, g["v"]) < 1e-8
    for key in ("Z", "m", "Lam", "raw_outputscale", "raw_lengthscale", "log_var_noise"):
        assert rel_err(grads[key], g["g_" + key]) < 1e-8, key
    if g["program"] is not None:
        assert rel_err(grads["theta"], g["g_theta"]) < 1e-8


def test_known_answers_at_init():
    """identity-initialised SAL TGP == SVGP ELBO; KL at init = 0.5(-M ln 1e-5 + M 1e-5 - M)."""
    a, b = load_golden("init_sal2_identity"), load_golden("init_svgp")
    assert rel_err(a["ELBO"], b["ELBO"]) < 1e-12
    M = a["params"]["m"].shape[0]
    kl = 0.5 * (-M * math.log(1e-5) + M * 1e-5 - M)
    assert abs(float(a["KLD"]) - kl) < 1e-9 * kl
    assert abs(kl - 525.646773) < 1e-5       # the value observed on real Power, M=100


def test_cholesky_ladder_matches_reference():
    g = load_golden("chol_ladder")
    L, A_used, jit = orc.psd_safe_cholesky(g["A"])
    assert abs(jit - float(g["jitter_used"])) < 1e-12
    assert rel_err(L, g["L"]) < 1e-9
    with pytest.raises(orc.NanError):
        bad = g["A"].clone()
        bad[0, 0] = float("nan")
        orc.psd_safe_cholesky(bad)


@pytest.mark.parametrize("name,flow", [("adam5_svgp", None), ("adam5_sal2", "sal2")])
def test_first_adam_steps_match_reference(name, flow):
    """Trainer sequence ELBO -> backward -> Adam(lr=0.01) (trainer_base.py:337-342)."""
    g = load_golden(name)
    leaves = {k: v.clone().requires_grad_(True) for k, v in g["params"].items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=0.01)
    hist = []
    for _ in range(g["history"].shape[0]):
        elbo, ell, kld = orc.elbo(g["X"], g["Y"], leaves["Z"], leaves["raw_lengthscale"], leaves["raw_outputscale"],
                                  leaves["m"], leaves["Lam"], leaves["log_var_noise"], float(g["N_total"]),
                                  g["program"], leaves.get("theta"), g["xs"], g["ws"])
        opt.zero_grad()
        (-elbo).backward()
        opt.step()
        hist.append([elbo.item(), ell.item(), kld.item()])
    assert rel_err(torch.tensor(hist, dtype=torch.float64), g["history"]) < 1e-9
    assert rel_err(leaves["Z"].detach(), g["final_Z"]) < 1e-9


def test_quadrature_exact_on_polynomials():
    """GH rule with S nodes integrates polynomials of degree < 2S exactly (known-answer, no oracle)."""
    xs, ws = orc.hermgauss(8)
    mu, v = torch.tensor([0.3], dtype=torch.float64), torch.tensor([1.7], dtype=torch.float64)
    f = torch.sqrt(2 * v) * xs + mu
    for k, want in ((1, 0.3), (2, 0.3 ** 2 + 1.7), (4, 0.3 ** 4 + 6 * 0.09 * 1.7 + 3 * 1.7 ** 2)):
        got = float((ws * f ** k).sum() / math.sqrt(math.pi))
        assert abs(got - want) < 1e-12 * max(1.0, abs(want))
