"""In-launch hand-offs (include/tgp_hip.h status[4..7]; csrc/tgp_prep.hpp): the prepare launch and the M x M backward launch pass
results between their workgroups through counted words with BOUNDED waits.  A wait that runs to its bound costs ~0.9 s and, when
the giving-up wave is not the one that reports, leaves no trace but the time -- so this test holds the eager path (no Adam in the
launch), the fused-Adam path and two engines on two streams to a time budget and to zero words afterwards."""
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _engine(N, D, M, flow, seed, stream=None):
    from tgp.pytorch_amd import ops, synthetic
    from tgp.pytorch_amd.engine import ElboEngine
    prob = synthetic.synthetic_problem(N, D, M, seed=seed, flow=flow, S=32)
    ops._ws_cache.clear()
    return ElboEngine(prob["X"], prob["Y"], prob["params"], N_total=float(N), flow_blocks=prob["program"], S=32,
                      rowp=prob["rowp"], device=torch.device("cuda:0"))


@pytest.mark.parametrize("flow,M,D", [("tanh3x2", 100, 4), ("idsal3", 100, 4), (None, 100, 4),
                                      ("sal2", 128, 13),      # MT = 8: no spare wave in the column blocks, 8 + 8 + 16 + 1 blocks
                                      ("tanh3x2", 5, 3)])     # MT = 1: the second half of the only row block has no tile
def test_no_hand_off_wait_runs_to_its_bound(flow, M, D):
    eng = _engine(2153, D, M, flow, seed=3)
    eng.elbo()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(20):
        eng.elbo()                      # eager, gradients only
    for ph in (1, 2, 4):
        eng.elbo(ph)
    if eng.fused_adam:
        for _ in range(20):
            eng.step_adam()             # the update inside the backward launch
    torch.cuda.synchronize()
    dt = time.time() - t
    st = eng.status.cpu().tolist()
    assert st[0] == 0 and st[4:] == [0, 0, 0, 0], st
    assert dt < 0.5, "43 launches took %.2f s: a hand-off wait ran to its bound" % dt


def test_two_engines_on_two_streams():
    engs, sts = [], []
    for k in range(2):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            e = _engine(8611, 4, 100, "tanh3x2", seed=k)
            e.step()
        engs.append(e)
        sts.append(st)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(30):
        for e, st in zip(engs, sts):
            with torch.cuda.stream(st):
                e.step()
    torch.cuda.synchronize()
    dt = time.time() - t
    for e in engs:
        st = e.status.cpu().tolist()
        assert st[0] == 0 and st[4:] == [0, 0, 0, 0], st
    assert dt < 0.5, "2 x 30 concurrent steps took %.2f s" % dt


@pytest.mark.parametrize("N,D,M,flow", [(2153, 13, 128, "sal2"),      # 1 792 assembly items, 1 815 parameters outside Lam: the loops
                                        (2153, 4, 100, "tanh3x2"), (700, 16, 60, None), (455, 13, 5, None)])
def test_update_inside_the_backward_launch_equals_the_separate_update(N, D, M, flow):
    """tgp_elbo_step_adam_f64 (Adam inside k_bwd: Lam in the threads that form its gradient, the rest from the LDS mirror of the
    final block) against the same steps as gradients + tgp_adam_dev: parameters, moments and scalars after three steps."""
    a = _engine(N, D, M, flow, seed=5)
    assert a.fused_adam
    for _ in range(3):
        a.step_adam()
    torch.cuda.synchronize()
    pa, ma, va, oa = a.fp.data.clone(), a.fp.exp_avg.clone(), a.fp.exp_avg_sq.clone(), a.fp.out[:3].clone()
    b = _engine(N, D, M, flow, seed=5)
    for _ in range(3):
        b.forward_backward()
        b.adam()
    torch.cuda.synchronize()
    for x, y, what in ((pa, b.fp.data, "parameters"), (ma, b.fp.exp_avg, "exp_avg"), (va, b.fp.exp_avg_sq, "exp_avg_sq"),
                       (oa, b.fp.out[:3], "scalars")):
        err = float((x - y).abs().max() / (y.abs().max() + 1e-300))
        assert err < 1e-13, (what, err)
    assert a.status.cpu().tolist()[4:] == [0, 0, 0, 0]
