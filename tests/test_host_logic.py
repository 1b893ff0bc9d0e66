"""CPU: host-side mirror of the reference API -- parameter names, flow compilation, optimiser groups, data path."""
import numpy as np
import pytest
import torch

import os

from oracle import tgp_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def f64():
    from tgp.pytorch_amd import config as cg
    old = torch.get_default_dtype()
    cg.set_maximum_precission()
    cg.device = "cpu"
    yield
    torch.set_default_dtype(old)


def build(flow_specs=None, M=10, D=4):
    from tgp.pytorch_amd.kernels import instance_kernel
    from tgp.pytorch_amd.likelihoods import GaussianLinearMean, GaussianNonLinearMean
    from tgp.pytorch_amd.models import sparse_MF_GP, sparse_MF_SP
    X = torch.randn(50, D)
    K = instance_kernel("scale_rbf", ard_num_dim=D, num_multioutput=1, kernel_is_shared=False,
                        init_params={"length_scale": 2.0, "kernel_scale": 2.0})
    ip = {"variational_distribution": {"variance_scale": 1e-5, "mean_scale": 0.0}}
    if flow_specs is None:
        return sparse_MF_GP(["zero", K], X, X[:M].clone(), 50, GaussianLinearMean(1, 0.05, False), 1, True, False, False,
                            False, False, 0.0, init_params=ip)
    return sparse_MF_SP(["zero", K], X, X[:M].clone(), 50, GaussianNonLinearMean(1, 0.05, False, 32), 1, True, False,
                        False, False, False, [flow_specs], "single", 0.0, init_params=ip)


def test_parameter_names_match_reference_appendix_b():
    from tgp.pytorch_amd.flows import SAL
    names = dict(build(SAL(2)).named_parameters())
    for n, shape in (("Z", (1, 10, 4)), ("likelihood.log_var_noise", (1, 1)), ("q_U.variational_mean", (1, 10)),
                     ("q_U.chol_variational_covar", (1, 10, 10)), ("covariance_function.raw_outputscale", (1,)),
                     ("covariance_function.base_kernel.raw_lengthscale", (1, 1, 4)), ("G_matrix.0.flow_arr.0.a", ()),
                     ("G_matrix.0.flow_arr.3.b", ())):
        assert tuple(names[n].shape) == shape, n
    svgp = dict(build(None).named_parameters())
    assert not any("G_matrix" in n for n in svgp)
    # initial values (main.py:95-110): l = 2, s2 = 2 through softplus, noise 0.05 through exp, Lq = sqrt(1e-5) I
    sp = torch.nn.functional.softplus
    assert torch.allclose(sp(svgp["covariance_function.base_kernel.raw_lengthscale"]), torch.tensor(2.0))
    assert torch.allclose(sp(svgp["covariance_function.raw_outputscale"]), torch.tensor(2.0))
    assert torch.allclose(svgp["likelihood.log_var_noise"].exp(), torch.tensor(0.05))
    assert torch.allclose(torch.diagonal(svgp["q_U.chol_variational_covar"][0]), torch.tensor(1e-5).sqrt())


def test_flow_compilation_matches_oracle_program():
    from tgp.pytorch_amd.flow import compile_flow, instance_flow
    from tgp.pytorch_amd.flows import SAL, StepTanhL
    spec, theta, nets = compile_flow(instance_flow(SAL(2)))
    prog, th = orc.sal_program(2)
    assert spec.blocks == prog and [float(p) for p in theta] == th.tolist() and not nets
    np.random.seed(0)
    spec, theta, _ = compile_flow(instance_flow(StepTanhL(3, 2, add_f0=True)))
    prog, th = orc.steptanh_program(3, 2, np.random.default_rng(0))
    assert spec.blocks == prog and len(theta) == th.numel() == 30
    idf = instance_flow(SAL(3, input_dependent=True, input_dim=4, num_hidden_layers=2, batch_norm=0, dropout=0.25,
                            hidden_dim=50, hidden_activation="relu", inference="MC_dropout"))
    with pytest.raises(AssertionError):
        compile_flow(idf)                      # the reference insists on turn_off_initializer_parameters() first
    idf.turn_off_initializer_parameters()
    spec, theta, nets = compile_flow(idf)
    assert spec.blocks == orc.sal_program(3, per_row=True)[0] and len(nets) == 6 and len(theta) == 6
    names = [n for n, _ in idf.named_parameters()]
    assert all(("NNets" in n) or n.endswith((".a", ".b")) for n in names) and sum("NNets" in n for n in names) == 36


def test_optimizer_groups_follow_main_py():
    from tgp.pytorch_amd.flow import instance_flow
    from tgp.pytorch_amd.flows import SAL
    from tgp.pytorch_amd.trainers import Trainer_SP_regression
    idf = instance_flow(SAL(3, input_dependent=True, input_dim=4, num_hidden_layers=2, dropout=0.25, hidden_dim=50,
                            hidden_activation="relu", inference="MC_dropout"))
    idf.turn_off_initializer_parameters()
    model = build(idf)
    tr = Trainer_SP_regression(model, [[], None, None], 1e20, False, False, torch.ones(1), -1, 100, True)
    sched = [[0.01, n] for n, _ in model.named_parameters() if "G_matrix" in n and "NNets" not in n]
    sched.append([0.01, 1e-5, "NNets"])
    groups, placed = tr._param_groups(sched, 0.01)
    wd = {g["weight_decay"]: len(g["params"]) for g in groups}
    assert wd[1e-5] == 36                                   # 6 MLPs x 3 layers x (weight, bias)
    assert sum(len(g["params"]) for g in groups) == len(list(model.parameters())) == len(placed)
    with pytest.raises(ValueError):
        tr._param_groups([[0.01, "NNets"], [0.01, 1e-5, "NNets"]], 0.01)
    # lr = 0.0 freezes (trainer_base.py:155-179): the parameter joins no group ...
    groups, placed = tr._param_groups([[0.0, "Z"], [0.02, "variational_mean"]], 0.01)
    ids = {id(q) for g in groups for q in g["params"]}
    assert id(model.Z) not in ids and "Z" not in placed
    assert [g["lr"] for g in groups] == [0.02, 0.01]
    # ... and a kept optimiser's parameters are skipped by later stages, naming one again is an error
    groups, placed = tr._param_groups([[0.01, "Z"]], 0.01, already_added=[n for n in placed])
    assert placed == ["Z"] and len(groups) == 1 and groups[0]["params"][0] is model.Z
    with pytest.raises(ValueError):
        tr._param_groups([[0.01, "Z"]], 0.01, already_added=["Z"])


def test_frozen_parameters_keep_the_run_off_the_resident_engine(monkeypatch):
    """optimisation_schedule entries with lr = 0.0 freeze parameters; the resident engine updates its whole
    flat buffer, so such a stage must not use it (no GPU needed: _engine_for is asked, never built)."""
    from tgp.pytorch_amd import trainers
    model = build(None)
    tr = trainers.Trainer_SP_regression(model, [[], None, None], 1e20, False, False, torch.ones(1), -1, 100, True)
    asked = []
    monkeypatch.setattr(tr, "_engine_for", lambda groups, lr, opt: asked.append(groups) or None)
    monkeypatch.setattr(tr, "_train_eager", lambda n, tot: None)
    tr.train(epochs=2, lr_ALL=0.01, opt="adam", keep_parameter_groups=True, optimisation_schedule=([0.5, 0.5], [[[0.0, "Z"]], [[0.01, "Z"]]]))
    assert asked == []                                      # two stages: never eligible
    assert len(tr.optimizer.param_groups) == 2              # stage 2 ADDED Z's group to the kept optimiser
    assert tr.optimizer.param_groups[1]["params"][0] is model.Z


def test_device_loader_protocol_and_normalisation():
    from tgp.pytorch_amd.data import return_dataset
    loaders, dc = return_dataset("synthetic_power", 10000, seed=1, options={"shuffle_train": True})
    assert dc["N_tr"] == 8611 and dc["Dx"] == 4 and dc["X_te"].shape[0] == 957
    (x, y), = list(loaders[0])                              # full batch: one step per epoch (main.py:74)
    assert x.shape == (8611, 4) and y.shape == (8611, 1) and len(loaders[0]) == 1
    assert abs(float(x.mean())) < 1e-12 and abs(float(y.std(unbiased=False)) - 1.0) < 1e-9     # numpy.std, ddof=0 (data.py:262-268)
    small, _ = return_dataset("synthetic_boston", 100, seed=2)
    assert sum(b[0].shape[0] for b in small[0]) == 455 and len(small[0]) == 5


def test_cpu_tensors_are_rejected_loudly():
    from tgp.pytorch_amd import lib as L
    m = build(None)
    with pytest.raises(L.TgpError):
        m.ELBO(torch.randn(5, 4), torch.randn(5, 1))


def _same(a, b):
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, torch.Tensor):
        return a.dtype == b.dtype and torch.equal(a, b)
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return a == b


@pytest.mark.parametrize("flow", [None, "sal2", "idsal3", "tanh3x2", "tanh5x6"])
@pytest.mark.parametrize("perturb", [True, False])
def test_product_generator_draws_what_the_oracle_generator_draws(flow, perturb):
    """bench.py builds its inputs with tgp.pytorch_amd.synthetic (product code, no oracle import); the parity tests
    build theirs with the oracle's generator.  Same seeds -> the same bits, also under a float32 default dtype."""
    from oracle import tgp_oracle as orc
    from tgp.pytorch_amd.synthetic import synthetic_problem
    old = torch.get_default_dtype()
    try:
        torch.set_default_dtype(torch.float32)
        mine = synthetic_problem(257, 4, 19, seed=5, flow=flow, S=20, perturb=perturb)
    finally:
        torch.set_default_dtype(old)
    ref = orc.synthetic_problem(257, 4, 19, seed=5, flow=flow, S=20, perturb=perturb)
    assert _same(ref, mine)


def test_bench_gpus_n_starts_its_own_ranks_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus N` without RANK in the environment (how the driver runs it) hands over to a fresh
    torch.distributed.run child with one rank per GPU and relays its exit code; with RANK set (a rank started by a
    launcher) it does not spawn again.  No GPU here: the spawn is observed, not executed."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = list(cmd), dict(env)

        class R:
            returncode = 7
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("GPU touched by the parent")))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert bench.main() == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_metric_label_follows_what_is_stepped():
    import bench
    w = bench.WORKLOADS["tgp_power_tanh3x2"]
    assert bench.metric_name("tgp_power_tanh3x2", w, 8611) == bench.BASELINE_METRIC          # 1 GPU, or strong scaling
    weak = bench.metric_name("tgp_power_tanh3x2", w, 8 * 8611)
    assert weak != bench.BASELINE_METRIC and "8 shards of 8611 rows" in weak
    a = bench.WORKLOADS["tgp_airline_tanh5x6"]
    assert "2000000-row" in bench.metric_name("tgp_airline_tanh5x6", a, 8 * 250000)


def test_timeline_tool_reads_a_kernel_trace(tmp_path):
    """tools/probes/timeline.py (how the general-M step was read off a rocprofv3 --kernel-trace CSV): one steady-state step,
    gaps per queue, union of busy intervals."""
    import os
    import subprocess
    import sys
    rows = ["Kind,Agent_Id,Queue_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp"]
    t = 1000
    for step in range(5):
        for name, q, dur, gap in (("tgp::k_prep_a(tgp::Plan)", 1, 3000, 0), ("void tgp::k_rows<7, 4, 1>(tgp::RowArgs)", 1, 5000, 0),
                                  ("tgp::k_mlp_fwd(x)", 3, 2000, -4000), ("tgp::k_reduce(tgp::Plan, double*)", 1, 500, 1000)):
            s = t + gap
            rows.append(f"KERNEL_DISPATCH,1,{q},1,\"{name}\",1,{s},{s + dur}")
            t = max(t, s + dur)
    f = tmp_path / "trace.csv"
    f.write_text("\n".join(rows) + "\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "probes", "timeline.py"), str(f)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "step = 4 kernels" in out.stdout and "k_rows<7, 4, 1>" in out.stdout and "union of kernel intervals" in out.stdout


def test_collective_decision_function():
    """engine.choose_collective: which all-reduce a data-parallel engine starts with."""
    from tgp.pytorch_amd.engine import choose_collective as c
    assert c(None, 1, None) == ("torch", False)            # one rank, no process group: nothing to reduce
    assert c(None, 2, "gloo") == ("torch", False)          # rehearsal backend: no RCCL under it
    assert c(None, 8, "nccl") == ("abi", True)             # the 8-GPU node: ABI collective, after its self-check
    assert c("auto", 1, "nccl") == ("abi", True)           # asked for explicitly: one rank is enough to run the check
    assert c("torch", 8, "nccl") == ("torch", False)       # the caller's choice is taken as is
    assert c("abi", 2, "gloo") == ("abi", False)
    import pytest
    with pytest.raises(ValueError):
        c("rccl", 2, "nccl")


def test_bench_rank_failure_is_a_clean_exit_3(tmp_path):
    """a rank of bench.py that fails behind init_process_group (communicator bootstrap timeout, RCCL missing, a
    hand-off timeout) reports, tears its process group down in a bounded way and leaves with code 3 through a fresh exit --
    here with a live 1-rank gloo group and a TimeoutError raised where the engine would raise it."""
    import subprocess
    import sys
    script = (
        "import os, sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')\n"
        "torch.distributed.init_process_group('gloo', rank=0, world_size=1)\n"
        "def boom(*a, **k):\n"
        "    raise TimeoutError('tgp_comm_init did not return within 1 s on rank 0')\n"
        "bench.run_bench = boom\n"
        "try:\n"
        "    bench.run_bench(None, 1, 0, None, 'gloo')\n"
        "except BaseException as e:\n"
        "    bench.fail_all_ranks(e, 0, 1)\n"
        "print('NOT REACHED')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert "NOT REACHED" not in r.stdout and "TimeoutError" in r.stderr and "rank 0 of 1 failed" in r.stderr


def test_rccl_comm_without_a_carrier_for_the_id_raises_instead_of_hanging():
    """RcclComm(world_size > 1) with torch.distributed not initialised has no way to carry rank 0's id to the other
    ranks; ranks != 0 used to call ncclCommInitRank with an all-zero id (a hang).  It raises RcclUnavailable now."""
    from tgp.pytorch_amd.engine import RcclComm, RcclUnavailable
    assert not torch.distributed.is_initialized()
    with pytest.raises(RcclUnavailable):
        RcclComm(world_size=2, rank=1)


def test_bench_expected_line_carries_what_it_is_built_from():
    import bench
    w = bench.WORKLOADS["tgp_power_tanh3x2"]
    e1 = bench.expected_line("tgp_power_tanh3x2", w, 1, "strong", 10540)
    e8 = bench.expected_line("tgp_power_tanh3x2", w, 8, "strong", 10540)
    assert e1["value"] > e8["value"] > 0 and "ESTIMATE" in e8["basis"]
    assert bench.allreduce_estimate_us(1, 10540) == 0.0 and bench.allreduce_estimate_us(8, 10540) > bench.allreduce_estimate_us(2, 10540)
