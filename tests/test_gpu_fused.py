"""GPU (-m gpu): the fused prepare + rows launch (k_rows<..., FUSED>, csrc/tgp_prep.hpp; opt-in through
TGP_FUSED_LAUNCH=1, read once per process) against the same fixtures as the two-launch path: the parity and full-size
suites, the device jitter ladder (a failed attempt restarts the row blocks' substitution), bit reproducibility."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fused_launch_passes_the_parity_suites():
    env = dict(os.environ, TGP_FUSED_LAUNCH="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_fullsize.py"),
                        os.path.join(ROOT, "tests", "test_gpu_models.py") + "::test_device_jitter_ladder_inside_the_captured_step",
                        os.path.join(ROOT, "tests", "test_gpu_fused.py") + "::test_fused_is_on_and_reproducible"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


def test_fused_is_on_and_reproducible():
    """Runs in the child process of the test above (skipped in the parent, where the launch is not fused): the status
    array's hand-off words are back to zero after every call, and two runs of a Power-sized step agree bit for bit."""
    if os.environ.get("TGP_FUSED_LAUNCH") != "1":
        pytest.skip("only meaningful with TGP_FUSED_LAUNCH=1 (test_fused_launch_passes_the_parity_suites starts it)")
    import torch
    from tgp.pytorch_amd import ops, synthetic
    prob = synthetic.synthetic_problem(8611, 4, 100, seed=0, flow="tanh3x2", S=32)
    dev = torch.device("cuda:0")
    p = {k: v.to(dev) for k, v in prob["params"].items()}
    flow = ops.FlowSpec(prob["program"], p["theta"].numel(), 0, dev)
    res = []
    for _ in range(3):
        out, g, status, _ = ops.elbo_step(prob["X"].to(dev), prob["Y"].to(dev), p["Z"], p["raw_lengthscale"], p["raw_outputscale"],
                                          p["m"], p["Lam"], p["log_var_noise"], 8611.0, flow=flow, theta=p["theta"], S=32)
        torch.cuda.synchronize()
        assert status.tolist() == [0] * 8, status.tolist()
        res.append((out.clone().cpu(), {k: v.clone().cpu() for k, v in g.items()}))
    for o, gg in res[1:]:
        assert torch.equal(o, res[0][0])
        for k in gg:
            assert torch.equal(gg[k], res[0][1][k]), k
