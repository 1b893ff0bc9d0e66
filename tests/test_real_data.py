"""CPU: the real-data path ("exact for index work") and the oracle at the full BASELINE size.

* the product loader (tgp/pytorch_amd/data.py) against tests/golden/{power,boston}_seed1.npz, which
  oracle/gen_golden.py produced by calling the reference's own `return_dataset` (code/dsp/data/datasets.py:81,
  uci_datasets.py:62-107, data.py:260-299): split indices bit-exact, z-scored rows and Y_std to 1e-15.
  Needs the reference's data files (CSV + split pickle: data, not source) -- found through $TGP_DATA_ROOT or
  /root/reference in the build container; skipped where they do not exist (the GPU box).
* the oracle against the reference's step 0 / evaluation path / first Adam steps at N=8611, M=100 (Power split 1)
  and N=455, M=5 (Boston): the known answers of SURVEY.md 8(c) included.
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import tgp_oracle as orc

UCI = os.environ.get("TGP_DATA_ROOT") or "/root/reference/code/datasets/regression/uci"
needs_data = pytest.mark.skipif(not os.path.exists(os.path.join(UCI, "power.csv")),
                                reason="the reference's UCI data files are not on this machine")


@pytest.fixture()
def f64():
    from tgp.pytorch_amd import config as cg
    old, old_dev = torch.get_default_dtype(), cg.device
    cg.set_maximum_precission()
    cg.device = "cpu"            # the loader only places tensors; nothing is computed here
    yield cg
    torch.set_default_dtype(old)
    cg.device = old_dev


@needs_data
@pytest.mark.parametrize("name", ["power", "boston"])
def test_product_loader_equals_reference_loader(name, f64):
    from tgp.pytorch_amd.data import return_dataset
    g = load_golden(name + "_seed1")
    loaders, dc = return_dataset(name, 10000, use_validation=None, seed=1,
                                 options={"shuffle_train": True, "split_from_disk": True, "root": UCI})
    assert len(loaders) == 2                                   # [train, test] (datasets.py:139-142)
    assert np.array_equal(dc["train_idx"], g["train_idx"].numpy()) and np.array_equal(dc["test_idx"], g["test_idx"].numpy())
    assert dc["N_tr"] == int(g["N_tr"]) and dc["N_te"] == int(g["N_te"]) and dc["Dx"] == g["X_tr"].shape[1] and dc["Dy"] == 1
    for k in ("X_tr", "Y_tr", "X_te", "Y_te"):
        assert dc[k].dtype == torch.float64 and dc[k].shape == g[k].shape
        assert float((dc[k] - g[k]).abs().max()) <= 1e-15, k
    assert np.asarray(dc["Y_std"]).shape == (1,) and abs(float(dc["Y_std"][0]) - float(g["Y_std"][0])) <= 1e-15
    # population standard deviation (numpy.std, ddof=0), not torch's unbiased default (data.py:262-268)
    assert abs(float(dc["X_tr"].std(0, unbiased=False).mean()) - 1.0) < 1e-12
    (x, y), = list(loaders[0])                                 # batch_size 10000 >= N: one full batch per epoch (main.py:74)
    assert x.shape == g["X_tr"].shape and y.shape == (g["X_tr"].shape[0], 1)
    # the main.py idiom that turns Y_std into the trainer's argument (main.py:292)
    ystd = torch.ones((1,)) * dc["Y_std"]
    assert ystd.shape == (1,) and abs(float(ystd[0]) - float(g["Y_std"][0])) <= 1e-15


@needs_data
@pytest.mark.parametrize("name", ["concrete", "kin8nm", "energy", "wine_red", "wine_white", "naval"])
def test_other_uci_sets_equal_the_reference_loader(name, f64):
    """The other regression sets the reference ships with split pickles (uci_datasets.py:186-283: separators ',', ';' and
    three blanks; energy's target is the second-to-last column and the last is dropped) against a compact fixture of the
    reference's own return_dataset: indices bit-exact, shapes, Y_std, every 97th z-scored row of each split to 1e-15."""
    from tgp.pytorch_amd.data import return_dataset
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "uci_loaders_seed1.npz"))
    loaders, dc = return_dataset(name, 10000, use_validation=None, seed=1,
                                 options={"shuffle_train": True, "split_from_disk": True, "root": UCI})
    assert np.array_equal(dc["train_idx"], g[name + ".train_idx"]) and np.array_equal(dc["test_idx"], g[name + ".test_idx"])
    n_tr, n_te, dx = (int(v) for v in g[name + ".shape"])
    assert (dc["N_tr"], dc["N_te"], dc["Dx"], dc["Dy"]) == (n_tr, n_te, dx, 1)
    assert abs(float(dc["Y_std"][0]) - float(g[name + ".Y_std"][0])) <= 1e-15 * max(1.0, float(g[name + ".Y_std"][0]))
    for k in ("X_tr", "Y_tr", "X_te", "Y_te"):
        ref = g[name + "." + k + "_every97"]
        got = dc[k].numpy()[::97]
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-15 * max(1.0, np.abs(ref).max()), k


def test_unknown_dataset_is_rejected(f64):
    from tgp.pytorch_amd.data import return_dataset
    with pytest.raises(ValueError):
        return_dataset("protein", 100, seed=1, options={"root": UCI})      # no split pickle ships for it
    with pytest.raises(ValueError):
        return_dataset("synthetic_concrete", 100, seed=1)


@needs_data
def test_validation_split_follows_the_reference(f64):
    """use_validation = [seed, N_val] (uci_datasets.py:54-56, data.py:216-234): numpy.random.seed permutation, the
    statistics come from the reduced train split."""
    from tgp.pytorch_amd.data import return_dataset
    g = load_golden("power_seed1")
    loaders, dc = return_dataset("power", 10000, use_validation=[3, 50], seed=1, options={"root": UCI})
    assert len(loaders) == 3 and dc["N_va"] == 50 and dc["N_tr"] == 8611 - 50
    assert float((dc["X_va"][:8] - g["val_X_va_head"]).abs().max()) <= 1e-15
    assert abs(float(dc["Y_std"][0]) - float(g["val_Y_std"][0])) <= 1e-15


@needs_data
def test_loader_errors(f64):
    from tgp.pytorch_amd.data import return_dataset
    with pytest.raises(KeyError):
        return_dataset("power", 10000, seed=99, options={"root": UCI})
    with pytest.raises(ValueError):
        return_dataset("protein", 10000, seed=1, options={"root": UCI})
    with pytest.raises(FileNotFoundError):
        return_dataset("power", 10000, seed=1, options={"root": "/nonexistent"})


def test_device_loader_minibatch_protocol(f64):
    """DataLoader(shuffle=True, drop_last=False) semantics: one permutation per epoch, consecutive slices, ragged tail."""
    from tgp.pytorch_amd.data import DeviceLoader
    X = torch.arange(23, dtype=torch.float64).reshape(23, 1)
    ld = DeviceLoader(X, X.clone(), 5, shuffle=True, device="cpu", seed=0)
    assert len(ld) == 5
    ref = torch.Generator().manual_seed(0)
    for _ in range(2):
        perm = torch.randperm(23, generator=ref)
        got = torch.cat([x for x, _ in ld]).reshape(-1).long()
        assert torch.equal(got, perm)
    sizes = [x.shape[0] for x, _ in DeviceLoader(X, X.clone(), 5, shuffle=False, device="cpu")]
    assert sizes == [5, 5, 5, 5, 3]


# ---------------------------------------------------------------------------------------------------
# the oracle at full size
# ---------------------------------------------------------------------------------------------------
FULL = ["power_init_svgp", "power_init_sal2", "power_svgp", "power_sal2", "power_tanh3x2", "power_idsal3",
        "boston_init_svgp", "boston_svgp", "med_idsal3"]
MLP = dict(D=4, H=50, L=2, nnets=6, act="relu")


def oracle_rowp(g, W=None):
    if "nn_W" not in g:
        return None
    return orc.mlp_rowp(g["X"], g["nn_W"] if W is None else W, **MLP)


@pytest.mark.parametrize("name", FULL)
def test_oracle_matches_reference_at_full_size(name):
    g = load_golden(name)
    W = g["nn_W"].clone().requires_grad_(True) if "nn_W" in g else None
    rowp = oracle_rowp(g, W)
    if rowp is not None:
        assert rel_err(rowp[:256].detach(), g["rowp_head"]) < 1e-13
    p = {k: v.clone().requires_grad_(True) for k, v in g["params"].items()}
    elbo, ell, kld = orc.elbo(g["X"], g["Y"], p["Z"], p["raw_lengthscale"], p["raw_outputscale"], p["m"], p["Lam"],
                              p["log_var_noise"], float(g["N_total"]), g["program"], p.get("theta"), g["xs"], g["ws"], rowp)
    elbo.backward()
    assert rel_err(elbo.detach(), g["ELBO"]) < 1e-10 and rel_err(ell.detach(), g["ELL"]) < 1e-10
    assert rel_err(kld.detach(), g["KLD"]) < 1e-10
    for key in ("Z", "m", "Lam", "raw_outputscale", "raw_lengthscale", "log_var_noise"):
        assert rel_err(p[key].grad, g["g_" + key]) < 1e-8, key
    if "g_theta" in g:
        assert rel_err(p["theta"].grad, g["g_theta"]) < 1e-8
    if W is not None:
        assert rel_err(W.grad, g["g_nn_W"]) < 1e-8
    if "mu" in g:
        pp = g["params"]
        mu, v = orc.qf_moments(g["X"], pp["Z"], pp["raw_lengthscale"], pp["raw_outputscale"], pp["m"], pp["Lam"])
        assert rel_err(mu, g["mu"]) < 1e-9 and rel_err(v, g["v"]) < 1e-8


def test_known_answers_on_real_power_and_boston():
    """step-0 ELBO -81723.694286 / ELL -81198.047513 / KLD 525.646773 for SVGP *and* the identity-
    initialised TGP on Power split 1 (M=100, KMEANS(n_init=1, seed=0)); Boston M=5 SVGP -10154.927639 / 26.28234."""
    a, b, c = load_golden("power_init_svgp"), load_golden("power_init_sal2"), load_golden("boston_init_svgp")
    assert abs(float(a["ELBO"]) + 81723.694286) < 1e-6 and abs(float(a["ELL"]) + 81198.047513) < 1e-6
    assert abs(float(a["KLD"]) - 525.646773) < 1e-6
    assert rel_err(b["ELBO"], a["ELBO"]) < 1e-12
    assert abs(float(c["ELBO"]) + 10154.927639) < 1e-6 and abs(float(c["KLD"]) - 26.28234) < 1e-5


@pytest.mark.parametrize("name", ["power_sal2", "power_idsal3", "boston_svgp"])
def test_oracle_evaluation_and_adam_steps_at_full_size(name):
    """Evaluation on the TEST split (what Trainer.compute_metrics reports) and the first Adam steps."""
    g = load_golden(name)
    pp = g["params"]
    mu, v = orc.qf_moments(g["X_te"], pp["Z"], pp["raw_lengthscale"], pp["raw_outputscale"], pp["m"], pp["Lam"])
    ystd = float(g["Y_std"][0])
    if g["program"] is None:
        m1, m2 = orc.marginal_moments_gauss(mu, v, pp["log_var_noise"])
    else:
        rowp_te = orc.mlp_rowp(g["X_te"], g["nn_W"], **MLP) if "nn_W" in g else None
        m1, m2 = orc.marginal_moments_flow(mu, v, pp["log_var_noise"], g["program"], pp.get("theta"), g["xs"], g["ws"], rowp_te)
        lp = orc.test_log_lik_flow(g["Y_te"].reshape(-1), mu, v, pp["log_var_noise"], g["program"], pp.get("theta"),
                                   g["xs"], g["ws"], ystd, rowp_te)
        assert rel_err(orc.test_log_lik_sum_ref(lp), g["test_logp_sum"]) < 1e-10
    assert rel_err(m1, g["pred_m1"]) < 1e-10 and rel_err(m2, g["pred_m2"]) < 1e-9
    # Adam history (trainer_base.py:337-342; two groups with weight decay 1e-5 on the nets for ID_TGP, main.py:276-288)
    leaves = {k: t.clone().requires_grad_(True) for k, t in pp.items()}
    groups = [{"params": list(leaves.values())}]
    W = None
    if "nn_W" in g:
        W = g["nn_W"].clone().requires_grad_(True)
        groups.append({"params": [W], "weight_decay": 1e-5})
    opt = torch.optim.Adam(groups, lr=0.01)
    hist = []
    for _ in range(g["history"].shape[0]):
        rowp = orc.mlp_rowp(g["X"], W, **MLP) if W is not None else None
        e, l, k = orc.elbo(g["X"], g["Y"], leaves["Z"], leaves["raw_lengthscale"], leaves["raw_outputscale"], leaves["m"],
                           leaves["Lam"], leaves["log_var_noise"], float(g["N_total"]), g["program"], leaves.get("theta"),
                           g["xs"], g["ws"], rowp)
        opt.zero_grad()
        (-e).backward()
        opt.step()
        hist.append([e.item(), l.item(), k.item()])
    assert rel_err(torch.tensor(hist, dtype=torch.float64), g["history"]) < 1e-9
    assert rel_err(leaves["Z"].detach(), g["final_Z"]) < 1e-9
    if W is not None:
        assert rel_err(W.detach()[:512], g["final_nn_W_head"]) < 1e-9
