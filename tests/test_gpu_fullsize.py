"""GPU (-m gpu): the HIP path against the REFERENCE at the full BASELINE sizes -- Power split 1 (N=8611, D=4,
M=100, S=32; SVGP, TGP SAL x 2, TGP StepTanhL 3 x 2, ID_TGP SAL x 3 with the six 4->50->50->1 nets) and Boston
(N=455, D=13, M=5).  Fixtures: tests/golden/power_*.npz / boston_*.npz / med_idsal3.npz, written by
oracle/gen_golden.py from the reference's own loader and model classes (real rows, the reference's KMeans centres).
Covered: step 0 (values, every gradient, the MLP weight gradients), the known answers of SURVEY.md 8(c), the
resident engine's first Adam steps (eager and HIP-graph replay), and the evaluation path on the test split."""
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_VAL, TOL_GRAD = 1e-9, 1e-7
MLP = dict(D=4, H=50, L=2, nnets=6)


@pytest.fixture(scope="module", autouse=True)
def f64():
    from tgp.pytorch_amd import config as cg
    old = torch.get_default_dtype()
    cg.set_maximum_precission()
    cg.device = DEV
    yield
    torch.set_default_dtype(old)


def mlp_spec():
    from tgp.pytorch_amd import ops
    return ops.MlpSpec(MLP["D"], MLP["H"], MLP["L"], MLP["nnets"], act="relu", drop_p=0.25, seed=0)


def hip_step0(g):
    """tgp_elbo_step_f64 (+ tgp_mlp_forward/backward_f64 for the input-dependent flow) on the fixture's inputs."""
    from tgp.pytorch_amd import ops
    p = {k: v.to(DEV) for k, v in g["params"].items()}
    X, Y = g["X"].to(DEV), g["Y"].to(DEV)
    flow = theta = rowp = S = None
    spec = None
    if g["program"] is not None:
        S = g["xs"].numel()
        theta = p.get("theta")
        RP = 0
        if "nn_W" in g:
            spec = mlp_spec()
            rowp = ops.mlp_forward(spec, X, g["nn_W"].to(DEV), training=False)
            RP = rowp.shape[1]
        flow = ops.FlowSpec(g["program"], theta.numel() if theta is not None else 0, RP, DEV)
    out, grads, status, (mu, v) = ops.elbo_step(X, Y, p["Z"], p["raw_lengthscale"], p["raw_outputscale"], p["m"], p["Lam"],
                                                 p["log_var_noise"], float(g["N_total"]), flow=flow, theta=theta, rowp=rowp,
                                                 S=S, want_moments=True)
    if spec is not None:
        grads["nn_W"] = ops.mlp_backward(spec, X, g["nn_W"].to(DEV), grads.pop("rowp"), training=False)
    torch.cuda.synchronize()
    return out.cpu(), {k: t.cpu() for k, t in grads.items()}, status.cpu(), mu.cpu(), v.cpu(), (rowp.cpu() if rowp is not None else None)


NAMES = {"Z": "g_Z", "raw_ls": "g_raw_lengthscale", "raw_os": "g_raw_outputscale", "m": "g_m", "Lam": "g_Lam",
         "lvn": "g_log_var_noise", "theta": "g_theta", "nn_W": "g_nn_W"}
FULL = ["power_init_svgp", "power_init_sal2", "power_svgp", "power_sal2", "power_tanh3x2", "power_idsal3",
        "boston_init_svgp", "boston_svgp", "med_idsal3"]


@pytest.mark.parametrize("name", FULL)
def test_step0_matches_reference_at_full_size(name):
    g = load_golden(name)
    out, grads, status, mu, v, rowp = hip_step0(g)
    assert int(status[0]) == 0 and int(status[1]) == 0
    assert rel_err(out[0], g["ELBO"]) < TOL_VAL and rel_err(out[1], g["ELL"]) < TOL_VAL and rel_err(out[2], g["KLD"]) < TOL_VAL
    for k, t in grads.items():
        assert rel_err(t, g[NAMES[k]]) < TOL_GRAD, (k, rel_err(t, g[NAMES[k]]))
    assert float(torch.triu(grads["Lam"], 1).abs().max()) == 0.0
    if rowp is not None:
        assert rel_err(rowp[:256], g["rowp_head"]) < 1e-12
    if "mu" in g:
        assert rel_err(mu, g["mu"]) < 1e-9
        assert float(((v - g["v"]).abs() / g["v"].abs()).max()) < 1e-6


def test_known_answers_on_real_power():
    """ELBO -81723.694286, ELL -81198.047513, KLD 525.646773 at initialisation, SVGP == identity TGP."""
    a = hip_step0(load_golden("power_init_svgp"))[0]
    b = hip_step0(load_golden("power_init_sal2"))[0]
    assert abs(float(a[0]) + 81723.694286) < 1e-5 and abs(float(a[1]) + 81198.047513) < 1e-5
    assert abs(float(a[2]) - 525.646773) < 1e-6
    assert rel_err(b[0], a[0]) < 1e-11


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("name", ["power_svgp", "power_sal2", "power_tanh3x2", "power_idsal3", "boston_svgp"])
def test_engine_adam_history_matches_reference_at_full_size(name, graph):
    """ELBO -> backward -> Adam(lr=0.01) (trainer_base.py:337-342; weight decay 1e-5 on the nets, main.py:276-288)
    through the resident engine, 5 steps from the fixture's state, against the reference's own history."""
    from tgp.pytorch_amd.engine import ElboEngine
    g = load_golden(name)
    kw = {}
    if "nn_W" in g:
        kw = dict(mlp=mlp_spec(), mlp_weights=g["nn_W"], mlp_training=False, nn_weight_decay=1e-5)
    eng = ElboEngine(g["X"], g["Y"], g["params"], float(g["N_total"]), flow_blocks=g["program"],
                     S=g["xs"].numel() if g["program"] is not None else None, device=DEV, **kw)
    hist = []
    if graph:
        eng.capture()
    for _ in range(g["history"].shape[0]):
        (eng.replay if graph else eng.step)()
        hist.append(list(eng.scalars()))
    eng.check_status()
    assert rel_err(torch.tensor(hist, dtype=torch.float64), g["history"]) < 1e-8
    assert rel_err(eng.fp.view("Z").cpu(), g["final_Z"]) < 1e-8
    assert rel_err(eng.fp.view("m").cpu(), g["final_m"]) < 1e-8
    assert rel_err(eng.fp.view("lvn").cpu(), g["final_log_var_noise"]) < 1e-8
    if "final_theta" in g:
        assert rel_err(eng.fp.view("theta").cpu(), g["final_theta"]) < 1e-8
    if "nn_W" in g:
        assert rel_err(eng.fp.view("nn").cpu()[:512], g["final_nn_W_head"]) < 1e-8


def build_model(g, flow_name):
    """The drop-in classes, built the way main.py builds them, carrying the fixture's parameters."""
    from tgp.pytorch_amd.flow import compile_flow, instance_flow
    from tgp.pytorch_amd.flows import SAL, StepTanhL
    from tgp.pytorch_amd.kernels import instance_kernel
    from tgp.pytorch_amd.likelihoods import GaussianLinearMean, GaussianNonLinearMean
    from tgp.pytorch_amd.models import sparse_MF_GP, sparse_MF_SP
    p = g["params"]
    N, D = g["X"].shape
    M = p["m"].numel()
    K = instance_kernel("scale_rbf", ard_num_dim=D, num_multioutput=1, kernel_is_shared=False,
                        init_params={"length_scale": 2.0, "kernel_scale": 2.0, "noisy_variance": 1e-6})
    ip = {"variational_distribution": {"variance_scale": 1e-5, "mean_scale": 0.0}}
    if flow_name is None:
        model = sparse_MF_GP(["zero", K], g["X"], p["Z"].clone(), N, GaussianLinearMean(1, 0.05, False), 1, True, False,
                             False, False, False, 0.0, init_params=ip)
    else:
        lik = GaussianNonLinearMean(1, 0.05, False, quadrature_points=g["xs"].numel())
        if flow_name.startswith("idsal"):
            specs = instance_flow(SAL(int(flow_name[5:]), input_dependent=True, input_dim=D, num_hidden_layers=2,
                                      batch_norm=0, dropout=0.25, hidden_dim=50, hidden_activation="relu",
                                      inference="MC_dropout"))
            specs.turn_off_initializer_parameters()
        elif flow_name.startswith("sal"):
            specs = SAL(int(flow_name[3:]))
        else:
            nb, ns = (int(t) for t in flow_name[4:].split("x"))
            specs = instance_flow(StepTanhL(nb, ns, add_f0=True))
        model = sparse_MF_SP(["zero", K], g["X"], p["Z"].clone(), N, lik, 1, True, False, False, False, False, [specs],
                             "single", 0.0, init_params=ip)
    with torch.no_grad():
        model.Z.data = p["Z"].reshape(1, M, D).clone()
        model.q_U.variational_mean.data = p["m"].reshape(1, M).clone()
        model.q_U.chol_variational_covar.data = p["Lam"].reshape(1, M, M).clone()
        model.covariance_function.raw_outputscale.data = p["raw_outputscale"].reshape(1).clone()
        model.covariance_function.base_kernel.raw_lengthscale.data = p["raw_lengthscale"].reshape(1, 1, D).clone()
        model.likelihood.log_var_noise.data = p["log_var_noise"].reshape(1, 1).clone()
        if flow_name is not None:
            _, theta_list, nets = compile_flow(model.G_matrix[0])
            for prm, val in zip(theta_list, p.get("theta", [])):
                prm.data = val.clone().reshape(())
            o = 0
            for net in nets:
                for q in net.parameters():
                    q.data = g["nn_W"][o:o + q.numel()].reshape(q.shape).clone()
                    o += q.numel()
    return model.to(DEV)


@pytest.mark.parametrize("name,flow", [("power_svgp", None), ("power_sal2", "sal2"), ("power_tanh3x2", "tanh3x2"),
                                       ("power_idsal3", "idsal3"), ("boston_svgp", None)])
def test_evaluation_on_the_test_split_matches_reference(name, flow):
    """test_log_likelihood + predictive moments on the held-out rows (sparse_MF_SP.py:637-825), i.e. the numbers
    Trainer.compute_metrics turns into the README's NLL / RMSE."""
    g = load_golden(name)
    model = build_model(g, flow)
    model.set_is_training(False)
    logp, (m1, m2) = model.test_log_likelihood(g["X_te"].to(DEV), g["Y_te"].to(DEV), return_moments=True,
                                               Y_std=g["Y_std"].to(DEV))
    assert rel_err(logp.cpu(), g["test_logp_sum"]) < 1e-9
    assert rel_err(m1.cpu().reshape(-1), g["pred_m1"]) < 1e-9
    assert rel_err(m2.cpu().reshape(-1), g["pred_m2"]) < 1e-8


@pytest.mark.parametrize("name,flow", [("power_sal2", "sal2"), ("power_idsal3", "idsal3")])
def test_model_classes_match_reference_at_full_size(name, flow):
    """ELBO() + (-ELBO).backward() on the drop-in classes (the eager trainer's idiom) at N=8611."""
    from tgp.pytorch_amd.flow import compile_flow
    g = load_golden(name)
    model = build_model(g, flow)
    model.set_is_training(True)
    model.eval()                           # dropout off, as in the fixture
    elbo, ell, kld = model.ELBO(g["X"].to(DEV), g["Y"].to(DEV))
    (-elbo).backward()
    assert rel_err(elbo.detach().cpu(), g["ELBO"]) < TOL_VAL and rel_err(ell.cpu(), g["ELL"]) < TOL_VAL
    assert rel_err(-model.Z.grad.cpu()[0], g["g_Z"]) < TOL_GRAD
    assert rel_err(-model.q_U.chol_variational_covar.grad.cpu()[0], g["g_Lam"]) < TOL_GRAD
    _, theta_list, nets = compile_flow(model.G_matrix[0])
    if theta_list:
        assert rel_err(torch.stack([-q.grad.reshape(()) for q in theta_list]).cpu(), g["g_theta"]) < TOL_GRAD
    if nets:
        gw = torch.cat([-q.grad.reshape(-1) for net in nets for q in net.parameters()]).cpu()
        assert rel_err(gw, g["g_nn_W"]) < TOL_GRAD
