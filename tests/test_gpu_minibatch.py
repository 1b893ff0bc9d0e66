"""GPU (-m gpu): minibatch training that stays resident (SURVEY.md 8f N2 + the Airline recipe, code/main.py:74,
trainers/trainer_base.py:322-349): rows gathered on the device by an index buffer, every step replayed from a HIP graph,
the ragged last batch on its own graph, through engine.MinibatchEngine and through Trainer_SP_regression.train."""
import pytest
import torch

from conftest import rel_err
from oracle import tgp_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def f64():
    from tgp.pytorch_amd import config as cg
    old = torch.get_default_dtype()
    cg.set_maximum_precission()
    cg.device = DEV
    yield
    torch.set_default_dtype(old)


def test_gather_rows_walks_through_the_epoch():
    """tgp_gather_rows_f64: batch r of the epoch = rows index[cursor : cursor + B]; the launch advances the device cursor
    and wraps it; NULL index = stored order."""
    from tgp.pytorch_amd import lib as L
    h = L.load()
    g = torch.Generator().manual_seed(0)
    N, D, B = 1000, 5, 300
    X = torch.randn(N, D, generator=g, dtype=torch.float64).to(DEV)
    Y = torch.randn(N, generator=g, dtype=torch.float64).to(DEV)
    perm = torch.randperm(N, generator=g).to(torch.int32).to(DEV)
    cur = torch.zeros(2, dtype=torch.int32, device=DEV)
    Xb, Yb = torch.zeros(B, D, dtype=torch.float64, device=DEV), torch.zeros(B, dtype=torch.float64, device=DEV)
    for step in range(5):                      # 3 full batches, the ragged one (100 rows), then the next epoch's first
        c0 = int(cur[0])
        n = min(B, N - c0)
        L.check(h.tgp_gather_rows_f64(L.ptr(X), L.ptr(Y), N, D, L.ptr(perm), L.ptr(cur), 0, n, n, N, L.ptr(Xb), L.ptr(Yb),
                                      L.stream_ptr()), "gather")
        idx = perm[c0:c0 + n].long()
        assert torch.equal(Xb[:n], X[idx]) and torch.equal(Yb[:n], Y[idx])
        assert int(cur[0]) == (0 if c0 + n >= N else c0 + n) and int(cur[1]) == 0
    L.check(h.tgp_gather_rows_f64(L.ptr(X), L.ptr(Y), N, D, None, L.ptr(cur), 7, 50, 0, N, L.ptr(Xb), L.ptr(Yb),
                                  L.stream_ptr()), "gather")
    c0 = int(cur[0])
    assert torch.equal(Xb[:50], X[c0 + 7:c0 + 57])          # shard offset, identity order, cursor left alone


@pytest.mark.parametrize("flow,M,graph", [("sal2", 20, True), ("tanh2x2", 20, False), (None, 150, True)])
def test_minibatch_engine_matches_oracle_adam_history(flow, M, graph):
    """2 epochs x (3 full batches of 256 + a ragged one of 132), stored order, against the oracle stepped batch by batch
    with torch.optim.Adam (ELL scale N_total / MB per batch, sparse_MF_SP.py:623-626).  M = 150 takes the general path."""
    from tgp.pytorch_amd.engine import MinibatchEngine
    N, B = 900, 256
    prob = orc.synthetic_problem(N, 4, M, seed=5, flow=flow, S=12)
    leaves = {k: t.clone().requires_grad_(True) for k, t in prob["params"].items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=0.01)
    ref = []
    for _ in range(2):
        for lo in range(0, N, B):
            xb, yb = prob["X"][lo:lo + B], prob["Y"][lo:lo + B]
            e, l, k = orc.elbo(xb, yb, leaves["Z"], leaves["raw_lengthscale"], leaves["raw_outputscale"], leaves["m"],
                               leaves["Lam"], leaves["log_var_noise"], float(N), prob["program"], leaves.get("theta"),
                               prob["xs"], prob["ws"])
            ref.append([float(e.detach()), float(l.detach()), float(k.detach())])
            opt.zero_grad()
            (-e).backward()
            opt.step()
    eng = MinibatchEngine(prob["X"], prob["Y"], prob["params"], float(N), B, device=DEV, flow_blocks=prob["program"],
                          S=12 if flow else None)
    assert eng.steps_per_epoch == 4 and eng.rest == 132
    hist = torch.zeros(8, 3, dtype=torch.float64, device=DEV)
    if graph:
        eng.capture()
    for ep in range(2):
        eng.set_order(None)
        eng.run_epoch(hist, 4 * ep, replay=graph)
    eng.check_status()
    assert rel_err(hist.cpu(), torch.tensor(ref, dtype=torch.float64)) < 1e-8
    assert rel_err(eng.fp.view("Z").cpu(), leaves["Z"].detach()) < 1e-8
    assert rel_err(eng.fp.view("Lam").cpu().tril(), leaves["Lam"].detach().tril()) < 1e-8


def test_trainer_minibatches_stay_on_the_resident_engine():
    """Trainer_SP_regression.train with a shuffling DeviceLoader of several batches: the resident engine (graph replay,
    device-side gather by the loader's own permutation stream) against the eager loop (autograd Function + torch Adam)
    fed by the same loader state -- same per-step losses, same final parameters."""
    from test_gpu_models import build_model
    from tgp.pytorch_amd import config as cg
    from tgp.pytorch_amd.data import DeviceLoader
    from tgp.pytorch_amd.engine import MinibatchEngine
    from tgp.pytorch_amd.trainers import Trainer_SP_regression
    prob = orc.synthetic_problem(1000, 4, 30, seed=2, flow="sal2", S=12)
    g = {"X": prob["X"], "Y": prob["Y"], "params": prob["params"], "xs": prob["xs"]}
    runs = {}
    for resident in (True, False):
        model = build_model(g, "sal2")
        loader = DeviceLoader(prob["X"], prob["Y"], 300, shuffle=True, device=DEV, seed=123)
        tr = Trainer_SP_regression(model, [loader], 1e20, False, False, torch.ones(1, device=DEV), -1, 100, True)
        cg.use_step_engine = resident
        try:
            tr.train(epochs=3, lr_ALL=0.01, opt="adam", keep_parameter_groups=True)
        finally:
            cg.use_step_engine = True
        assert isinstance(tr._engine, MinibatchEngine) == resident
        assert len(tr.loss_arr) == 3 * 4                  # 3 epochs x (3 full + 1 ragged) optimiser steps
        runs[resident] = (torch.tensor(tr.loss_arr, dtype=torch.float64), model.Z.detach().cpu().clone(),
                          model.q_U.variational_mean.detach().cpu().clone())
    assert rel_err(runs[True][0], runs[False][0]) < 1e-8
    assert rel_err(runs[True][1], runs[False][1]) < 1e-8 and rel_err(runs[True][2], runs[False][2]) < 1e-8
