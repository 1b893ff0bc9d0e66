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


def test_a_hand_off_timeout_is_sticky_and_skips_the_update():
    """a wait of the backward launch that runs to its bound must not be lost when the next step's prepare launch rewrites
    status[0], and must not step the optimiser on gradients built from stale operands.  The PP count of status[6] is poisoned
    (sign bit: the count can never reach its target), one fused-update step is launched, then two healthy ones:
    status[3] keeps the event, the first step changed nothing outside Lam and returned NaN scalars, the Adam step counter did
    not advance, check_status() raises HandoffTimeoutError although status[0] is 0 again."""
    from tgp.pytorch_amd import ops
    eng = _engine(2153, 4, 100, "tanh3x2", seed=7)
    assert eng.fused_adam
    eng.step_adam()
    torch.cuda.synchronize()
    assert eng.status.cpu().tolist()[3] == 0
    before, step0 = eng.fp.data.clone(), int(eng.step_dev[0])
    lam_lo = eng.fp.offsets["Lam"]
    lam_hi = lam_lo + eng.fp.sizes["Lam"]
    eng.status[6] = -2 ** 31                       # bits 24-31 are the PP count: never >= 2 MT with the sign bit set
    t = time.time()
    eng.step_adam()
    torch.cuda.synchronize()
    dt = time.time() - t
    st = eng.status.cpu().tolist()
    assert st[0] == ops.STATUS_SYNC_TIMEOUT and st[3] >= 1, st
    assert st[4:] == [0, 0, 0, 0], st             # the final role still zeroes the words: the next launch starts clean
    assert dt < 5.0, "a bounded wait took %.1f s" % dt
    assert int(eng.step_dev[0]) == step0, "the optimiser stepped on a timed-out launch"
    after = eng.fp.data
    assert torch.equal(after[:lam_lo], before[:lam_lo]) and torch.equal(after[lam_hi:], before[lam_hi:])
    assert bool(torch.isnan(eng.fp.out[:3]).all())
    sticky = st[3]
    for _ in range(2):
        eng.step_adam()
    torch.cuda.synchronize()
    st = eng.status.cpu().tolist()
    assert st[0] == 0 and st[3] == sticky, st     # status[0] is rewritten by every prepare launch; status[3] is not
    assert int(eng.step_dev[0]) == step0 + 2 and bool(torch.isfinite(eng.fp.out[:3]).all())
    with pytest.raises(ops.HandoffTimeoutError):
        eng.check_status()
    eng.status[3] = 0                             # handled: the caller clears the sticky word
    eng.check_status()


def test_general_path_hand_offs_leave_the_words_zero_and_a_timeout_is_sticky():
    """Round 6: the general-M factorisation hands diagonal blocks over INSIDE launches (k_big_kmm_potrf, k_fac_potrf: status[4]).
    A healthy step leaves status[4..7] zero and status[3] untouched; with status[4] poisoned (sign bit: the count can never reach
    its target) the factorising workgroup's wait expires within seconds, the event is counted in status[3] (sticky),
    check_status() raises HandoffTimeoutError, the word is zero again, and the next steps are healthy."""
    from tgp.pytorch_amd import ops
    eng = _engine(700, 5, 300, "sal2", seed=5)      # M = 300: three 128-column steps, every merged launch of the chain
    eng.elbo()
    torch.cuda.synchronize()
    st = eng.status.cpu().tolist()
    assert st[0] == 0 and st[3] == 0 and st[4:] == [0, 0, 0, 0], st
    ref = eng.fp.out[:3].clone()
    eng.status[4] = -2 ** 31
    t = time.time()
    eng.elbo()
    torch.cuda.synchronize()
    dt = time.time() - t
    st = eng.status.cpu().tolist()
    assert st[3] >= 1 and st[4:] == [0, 0, 0, 0], st
    assert dt < 20.0, "a bounded wait took %.1f s" % dt
    with pytest.raises(ops.HandoffTimeoutError):
        eng.check_status()
    sticky = st[3]
    eng.elbo()
    torch.cuda.synchronize()
    st = eng.status.cpu().tolist()
    assert st[0] == 0 and st[3] == sticky and st[4:] == [0, 0, 0, 0], st
    assert torch.equal(eng.fp.out[:3], ref)          # same parameters: the healthy step is the first one bit for bit
    eng.status[3] = 0
    eng.check_status()
