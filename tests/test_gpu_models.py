"""GPU (-m gpu): the drop-in model classes, the trainer/engine sequence, the stand-alone operators, the
Cholesky failure protocol and size-independent properties at the BASELINE sizes -- all through the C ABI."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
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


def build_model(g, flow_name):
    """Reference-style construction (code/main.py:217-262) + the fixture's parameter values."""
    from tgp.pytorch_amd.flow import instance_flow
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
        if flow_name.startswith("sal"):
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
            from tgp.pytorch_amd.flow import compile_flow
            for prm, val in zip(compile_flow(model.G_matrix[0])[1], p["theta"]):
                prm.data = val.clone().reshape(())
    return model.to(DEV)


CASES = [("tiny_svgp", None), ("tiny_sal2", "sal2"), ("tiny_tanh3x2", "tanh3x2"), ("med_svgp", None), ("med_sal2", "sal2"),
         ("med_tanh3x2", "tanh3x2"), ("boston_like_svgp", None), ("ragged_sal2", "sal2")]


@pytest.mark.parametrize("name,flow", CASES)
def test_model_elbo_and_backward_match_reference(name, flow):
    g = load_golden(name)
    model = build_model(g, flow)
    model.set_is_training(True)
    elbo, ell, kld = model.ELBO(g["X"].to(DEV), g["Y"].to(DEV))
    (-elbo).backward()                                   # the trainer's idiom (trainers_regression.py:85-86)
    assert rel_err(elbo.detach().cpu(), g["ELBO"]) < 1e-9
    assert rel_err(ell.cpu(), g["ELL"]) < 1e-9 and rel_err(kld.cpu(), g["KLD"]) < 1e-9
    named = dict(model.named_parameters())
    checks = [("Z", "g_Z"), ("q_U.variational_mean", "g_m"), ("q_U.chol_variational_covar", "g_Lam"),
              ("covariance_function.raw_outputscale", "g_raw_outputscale"),
              ("covariance_function.base_kernel.raw_lengthscale", "g_raw_lengthscale"),
              ("likelihood.log_var_noise", "g_log_var_noise")]
    for pn, gn in checks:
        assert rel_err(-named[pn].grad.cpu().reshape(-1), g[gn].reshape(-1)) < 1e-7, pn
    if flow is not None:
        from tgp.pytorch_amd.flow import compile_flow
        gt = torch.stack([-q.grad.reshape(()) for q in compile_flow(model.G_matrix[0])[1]]).cpu()
        assert rel_err(gt, g["g_theta"]) < 1e-7


@pytest.mark.parametrize("name,flow", [("tiny_sal2", "sal2"), ("med_sal2", "sal2"), ("tiny_tanh3x2", "tanh3x2"),
                                       ("ragged_sal2", "sal2"), ("tiny_svgp", None), ("med_svgp", None)])
def test_evaluation_path_matches_reference(name, flow):
    """test_log_likelihood / predictive_distribution / marginal q(f) (sparse_MF_SP.py:457-540, 637-825)."""
    g = load_golden(name)
    model = build_model(g, flow)
    model.set_is_training(False)
    Y_std = g["Y_std"].to(DEV) if "Y_std" in g else torch.tensor([1.7], device=DEV)
    logp, (m1, m2) = model.test_log_likelihood(g["X"].to(DEV), g["Y"].to(DEV), return_moments=True, Y_std=Y_std)
    assert rel_err(logp.cpu(), g["test_logp_sum"]) < 1e-9
    assert rel_err(m1.cpu().reshape(-1), g["pred_m1"]) < 1e-9
    assert rel_err(m2.cpu().reshape(-1), g["pred_m2"]) < 1e-8
    mu, v = model.marginal_variational_qf_parameters(g["X"].to(DEV), diagonal=True, is_duvenaud=False)
    assert mu.shape == (1, g["X"].shape[0], 1) and rel_err(mu.cpu().reshape(-1), g["mu"]) < 1e-9
    assert rel_err(model.KLD().cpu(), g["KLD"]) < 1e-12


@pytest.mark.parametrize("resident", [True, False])
@pytest.mark.parametrize("name,flow", [("adam5_svgp", None), ("adam5_sal2", "sal2")])
def test_trainer_first_steps_match_reference(name, flow, resident):
    """Trainer sequence ELBO -> backward -> Adam(lr=0.01) on the drop-in classes vs the reference's history: through the
    resident graph-captured engine the trainer switches to for full-batch Adam runs, and through the eager loop
    (autograd Function + torch.optim.Adam) it keeps for everything else."""
    from tgp.pytorch_amd import config as cg
    from tgp.pytorch_amd.data import DeviceLoader
    from tgp.pytorch_amd.trainers import Trainer_SP_regression
    g = load_golden(name)
    model = build_model(g, flow)
    loader = DeviceLoader(g["X"], g["Y"], 10000, shuffle=False, device=DEV)
    tr = Trainer_SP_regression(model, [loader, None, None], 1e20, False, False, torch.ones(1, device=DEV), -1, 100, True)
    cg.use_step_engine = resident
    try:
        tr.train(epochs=g["history"].shape[0], lr_ALL=0.01, opt="adam", keep_parameter_groups=True)
    finally:
        cg.use_step_engine = True
    assert (tr._engine is not None) == resident
    hist = torch.tensor([[-l, e, k] for l, e, k in zip(tr.loss_arr, tr.ELL_arr, tr.KLD_arr)], dtype=torch.float64)
    assert rel_err(hist, g["history"]) < 1e-8
    assert rel_err(model.Z.detach().cpu()[0], g["final_Z"]) < 1e-8


def test_frozen_parameters_are_not_trained():
    """optimisation_schedule entries with lr = 0.0 freeze parameters (trainer_base.py:155-179).  The resident engine
    updates its whole flat buffer, so the trainer must fall back to the torch optimiser for such a run."""
    from tgp.pytorch_amd.data import DeviceLoader
    from tgp.pytorch_amd.trainers import Trainer_SP_regression
    g = load_golden("adam5_sal2")
    model = build_model(g, "sal2")
    Z0 = model.Z.detach().clone()
    m0 = model.q_U.variational_mean.detach().clone()
    loader = DeviceLoader(g["X"], g["Y"], 10000, shuffle=False, device=DEV)
    tr = Trainer_SP_regression(model, [loader], 1e20, False, False, torch.ones(1, device=DEV), -1, 100, True)
    tr.train(epochs=3, lr_ALL=0.01, opt="adam", keep_parameter_groups=True, optimisation_schedule=([1.0], [[[0.0, "Z"]]]))
    assert tr._engine is None and tr.optimizer is not None
    assert torch.equal(model.Z.detach(), Z0) and not torch.equal(model.q_U.variational_mean.detach(), m0)


@pytest.mark.parametrize("graph", [False, True])
def test_engine_hip_adam_matches_reference_history(graph):
    """The resident step engine (fused ELBO + tgp_adam_dev_f64, optionally replayed from a HIP graph)."""
    from tgp.pytorch_amd.engine import ElboEngine
    g = load_golden("adam5_sal2")
    eng = ElboEngine(g["X"], g["Y"], g["params"], float(g["N_total"]), flow_blocks=g["program"], S=g["xs"].numel(),
                     device=DEV)
    hist = []
    if graph:
        eng.capture()
    for _ in range(g["history"].shape[0]):
        (eng.replay if graph else eng.step)()
        hist.append(list(eng.scalars()))
    eng.check_status()
    assert rel_err(torch.tensor(hist, dtype=torch.float64), g["history"]) < 1e-8
    assert rel_err(eng.fp.view("Z").cpu(), g["final_Z"]) < 1e-8
    assert rel_err(eng.fp.view("theta").cpu(), g["final_theta"]) < 1e-8


def test_input_dependent_flow_gradients_reach_the_mlps():
    """ID_TGP: per-row a_n, b_n from MLPs (flow.py:949-965); gradient w.r.t. the MLP weights = autograd of the
    MLPs driven by the kernel's d ELBO / d rowp, checked against the oracle's d ELBO / d rowp."""
    from tgp.pytorch_amd.flow import compile_flow, instance_flow
    from tgp.pytorch_amd.flows import SAL
    from tgp.pytorch_amd.kernels import instance_kernel
    from tgp.pytorch_amd.likelihoods import GaussianNonLinearMean
    from tgp.pytorch_amd.models import sparse_MF_SP
    torch.manual_seed(0)
    prob = orc.synthetic_problem(200, 4, 20, seed=2, flow="idsal3", S=16)
    p = prob["params"]
    idf = instance_flow(SAL(3, input_dependent=True, input_dim=4, num_hidden_layers=2, batch_norm=0, dropout=0.25,
                            hidden_dim=50, hidden_activation="relu", inference="MC_dropout"))
    idf.turn_off_initializer_parameters()
    K = instance_kernel("scale_rbf", ard_num_dim=4, num_multioutput=1, kernel_is_shared=False,
                        init_params={"length_scale": 2.0, "kernel_scale": 2.0})
    model = sparse_MF_SP(["zero", K], prob["X"], p["Z"].clone(), 200, GaussianNonLinearMean(1, 0.05, False, 16), 1, True,
                         False, False, False, False, [idf], "single", 0.0,
                         init_params={"variational_distribution": {"variance_scale": 1e-5, "mean_scale": 0.0}}).to(DEV)
    with torch.no_grad():
        model.q_U.variational_mean.data = p["m"].reshape(1, -1).to(DEV)
        model.q_U.chol_variational_covar.data = p["Lam"].reshape(1, 20, 20).to(DEV)
        for prm, val in zip(compile_flow(model.G_matrix[0])[1], p["theta"]):
            prm.data = val.clone().reshape(()).to(DEV)
    model.eval()                                    # dropout off: deterministic per-row parameters
    X, Y = prob["X"].to(DEV), prob["Y"].to(DEV)
    nets = compile_flow(model.G_matrix[0])[2]
    with torch.no_grad():
        rowp = torch.cat([n(X) for n in nets], -1).cpu()
    p2 = dict(p)
    p2["Z"], p2["raw_lengthscale"], p2["raw_outputscale"] = (model.Z.detach().cpu()[0],
                                                             model.covariance_function.base_kernel.raw_lengthscale.detach().cpu().reshape(-1),
                                                             model.covariance_function.raw_outputscale.detach().cpu())
    p2["log_var_noise"] = model.likelihood.log_var_noise.detach().cpu().reshape(-1)
    (elbo_o, _, _), og = orc.elbo_and_grads(prob["X"], prob["Y"], p2, 200.0, prob["program"], prob["xs"], prob["ws"], rowp)
    elbo, _, _ = model.ELBO(X, Y)
    elbo.backward()
    assert rel_err(elbo.detach().cpu(), elbo_o) < 1e-9
    assert rel_err(model.Z.grad.cpu()[0], og["Z"]) < 1e-7
    # expected MLP gradients: push the oracle's d ELBO / d rowp through the same MLPs with torch autograd
    want = torch.autograd.grad(torch.cat([n(X) for n in nets], -1), [q for n in nets for q in n.parameters()],
                               grad_outputs=og["rowp"].to(DEV))
    got = [q.grad for n in nets for q in n.parameters()]
    for a, b in zip(got, want):
        assert rel_err(a.cpu(), b.cpu()) < 1e-7


def test_fully_bayesian_log_likelihood_applies_mc_dropout():
    """Fully Bayesian ID_TGP evaluation (sparse_MF_SP.py:753-776): enable_eval_dropout() re-enables only the Dropout
    layers after eval(), and every MC sample draws a fresh mask.  The HIP MLP path must follow the LAYERS' state (the
    container's .training is False there): S_MC samples -> logsumexp - log S, each sample's mask = the hash mask of
    the step counter (ops.mlp_keep_mask), restated here on the host for the first sample."""
    from tgp.pytorch_amd import ops
    from tgp.pytorch_amd.flow import compile_flow, instance_flow
    from tgp.pytorch_amd.flows import SAL
    from tgp.pytorch_amd.kernels import instance_kernel
    from tgp.pytorch_amd.likelihoods import GaussianNonLinearMean
    from tgp.pytorch_amd.models import sparse_MF_SP
    torch.manual_seed(0)
    prob = orc.synthetic_problem(200, 4, 20, seed=2, flow="idsal3", S=16)
    p = prob["params"]
    idf = instance_flow(SAL(3, input_dependent=True, input_dim=4, num_hidden_layers=2, batch_norm=0, dropout=0.25,
                            hidden_dim=50, hidden_activation="relu", inference="MC_dropout"))
    idf.turn_off_initializer_parameters()
    K = instance_kernel("scale_rbf", ard_num_dim=4, num_multioutput=1, kernel_is_shared=False,
                        init_params={"length_scale": 2.0, "kernel_scale": 2.0})
    model = sparse_MF_SP(["zero", K], prob["X"], p["Z"].clone(), 200, GaussianNonLinearMean(1, 0.05, False, 16), 1, True,
                         False, False, False, False, [idf], "single", 0.0,
                         init_params={"variational_distribution": {"variance_scale": 1e-5, "mean_scale": 0.0}}).to(DEV)
    with torch.no_grad():
        model.q_U.variational_mean.data = p["m"].reshape(1, -1).to(DEV)
        model.q_U.chol_variational_covar.data = p["Lam"].reshape(1, 20, 20).to(DEV)
    X, Y = prob["X"].to(DEV), prob["Y"].to(DEV)
    Ystd = torch.ones(1, device=DEV)
    model.set_is_training(False)
    model.be_fully_bayesian(False)
    plain, _ = model.test_log_likelihood(X, Y, False, Ystd)
    plain2, _ = model.test_log_likelihood(X, Y, False, Ystd)
    assert torch.equal(plain, plain2)                       # no dropout: deterministic
    model.be_fully_bayesian(True)
    b1, _ = model.test_log_likelihood(X, Y, False, Ystd, S_MC_NNet=8)
    b2, _ = model.test_log_likelihood(X, Y, False, Ystd, S_MC_NNet=8)
    assert bool(torch.isfinite(b1).all()) and bool(torch.isfinite(b2).all())
    assert float((b1 - plain).abs()) > 1e-6                 # the masks are applied ...
    assert float((b1 - b2).abs()) > 1e-9                    # ... and redrawn for every call
    # one sample, restated: per-row parameters from the nets with the host mask of the NEXT step counter value
    _, _, nets = compile_flow(model.G_matrix[0])
    mspec = model._cfg["mlp"].salted(ops.MASK_SALT_EVAL)     # evaluation's own mask stream
    step = int(model._cfg["mlp_step"][0]) + 1
    W = torch.cat([q.reshape(-1) for net in nets for q in net.parameters()]).detach()
    masks = [[torch.from_numpy(ops.mlp_keep_mask(mspec.seed, step, k, l, 200, mspec.H, mspec.drop_p)).to(torch.float64)
              for l in range(mspec.L)] for k in range(mspec.nnets)]
    rowp_ref = _torch_mlps(prob["X"], W.cpu(), mspec, masks)
    one, _ = model.test_log_likelihood(X, Y, False, Ystd, S_MC_NNet=1)
    model._eval_mode()
    with torch.no_grad():
        mq, cq = model.marginal_variational_qf_parameters(X.repeat(1, 1, 1), diagonal=True, is_duvenaud=False)
    spec, theta, _ = model._flow_inputs(X, with_grad=False)      # (advances the counter once more: not used below)
    lvn = model.likelihood.log_var_noise.detach().reshape(-1)[:1].contiguous()
    _, _, lp = ops.predict(mq.reshape(-1).contiguous(), cq.reshape(-1).contiguous(), lvn, spec,
                           theta.detach() if theta is not None else None, model.quad_points, rowp_ref.to(DEV),
                           Y=Y, Y_std=1.0)
    import numpy
    want = (lp + 0.5 * float(numpy.log(numpy.pi))).sum() - 200 * float(numpy.float32(0.5) * numpy.log(numpy.float32(numpy.pi)))
    assert rel_err(one.cpu(), want.reshape(1).cpu()) < 1e-9
    # S_MC = 3 samples in ONE pass (the reference's X.repeat to (S_MC, N, Dx)): the nets run over 3 * 200 rows in one
    # launch, sample s of row n carries the mask of row s * 200 + n at the next counter value; predictive moments of
    # the same pass = the mixture over the samples (sparse_MF_SP.py:516-531)
    S = 3
    step = int(model._cfg["mlp_step"][0]) + 1
    masks = [[torch.from_numpy(ops.mlp_keep_mask(mspec.seed, step, k, l, S * 200, mspec.H, mspec.drop_p)).to(torch.float64)
              for l in range(mspec.L)] for k in range(mspec.nnets)]
    rowp_s = _torch_mlps(prob["X"].repeat(S, 1), W.cpu(), mspec, masks).to(DEV)
    three, _ = model.test_log_likelihood(X, Y, False, Ystd, S_MC_NNet=S)
    mu_q, v_q = mq.reshape(-1).contiguous(), cq.reshape(-1).contiguous()
    m1s, m2s, lps = ops.predict(mu_q.repeat(S), v_q.repeat(S), lvn, spec, theta.detach() if theta is not None else None,
                                model.quad_points, rowp_s, Y=Y.reshape(-1).repeat(S), Y_std=1.0)
    stack = lps.reshape(S, 200) + 0.5 * float(numpy.log(numpy.pi)) - float(numpy.float32(0.5) * numpy.log(numpy.float32(numpy.pi)))
    want3 = torch.logsumexp(stack, 0).sum() - 200 * numpy.log(S)
    assert rel_err(three.cpu(), want3.reshape(1).cpu()) < 1e-9
    step = int(model._cfg["mlp_step"][0]) + 1
    masks = [[torch.from_numpy(ops.mlp_keep_mask(mspec.seed, step, k, l, S * 200, mspec.H, mspec.drop_p)).to(torch.float64)
              for l in range(mspec.L)] for k in range(mspec.nnets)]
    rowp_s = _torch_mlps(prob["X"].repeat(S, 1), W.cpu(), mspec, masks).to(DEV)
    m1, m2, _, _ = model.predictive_distribution(X, diagonal=True, S_MC_NNet=S)
    mY, cY, _ = ops.predict(mu_q.repeat(S), v_q.repeat(S), lvn, spec, theta.detach() if theta is not None else None,
                            model.quad_points, rowp_s)
    mY, cY = mY.reshape(S, 200), cY.reshape(S, 200)
    assert rel_err(m1.reshape(-1).cpu(), mY.mean(0).cpu()) < 1e-10
    assert rel_err(m2.reshape(-1).cpu(), ((cY + mY ** 2).mean(0) - mY.mean(0) ** 2).cpu()) < 1e-9


def test_cholesky_failure_protocol():
    """status word -> the reference's jitter ladder (dsp/utils.py:256-269) and NanError (:241-254)."""
    from tgp.pytorch_amd import ops
    g = load_golden("chol_ladder")
    with pytest.warns(ops.NumericalWarning):
        L, A_used = ops.psd_safe_cholesky(g["A"].to(DEV))
    assert abs(float((A_used.cpu() - g["A"]).diagonal().mean()) - float(g["jitter_used"])) < 1e-12
    assert rel_err(L.cpu(), g["L"]) < 1e-6              # rank-3 + 1e-8 I: cond ~ 1e9, factor agrees to ~1e-7
    Lo, Li, status = ops.cholesky(g["A"].to(DEV), want_inverse=True)
    assert int(status[0]) > 0                           # LAPACK-style info: first non-positive pivot
    bad = g["A"].clone()
    bad[0, 0] = float("nan")
    with pytest.raises(ops.NanError):
        ops.psd_safe_cholesky(bad.to(DEV))
    # the same protocol inside the fused step: duplicated inducing points make K_MM singular
    prob = orc.synthetic_problem(128, 3, 16, seed=1, flow=None, S=8)
    p = {k: v.to(DEV) for k, v in prob["params"].items()}
    p["Z"][1] = p["Z"][0]
    with pytest.warns(ops.NumericalWarning):
        out, grads, status, _ = ops.elbo_step_safe(prob["X"].to(DEV), prob["Y"].to(DEV), p["Z"], p["raw_lengthscale"],
                                                    p["raw_outputscale"], p["m"], p["Lam"], p["log_var_noise"], 128.0)
    assert int(status[0]) == 0 and bool(torch.isfinite(out).all())


def test_standalone_operators_match_torch():
    from tgp.pytorch_amd import ops
    prob = orc.synthetic_problem(300, 5, 40, seed=4, flow="tanh2x3", S=12)
    p = {k: v.to(DEV) for k, v in prob["params"].items()}
    X, Y = prob["X"].to(DEV), prob["Y"].to(DEV)
    # K1/K2: kernel matrices (gpytorch formula restated in the oracle)
    Kmm = ops.kmm(p["Z"], p["raw_lengthscale"], p["raw_outputscale"]).cpu()
    Knm = ops.knm(X, p["Z"], p["raw_lengthscale"], p["raw_outputscale"]).cpu()
    pc = prob["params"]
    assert rel_err(Kmm, orc.scale_rbf(pc["Z"], pc["Z"], pc["raw_lengthscale"], pc["raw_outputscale"])) < 1e-13
    assert rel_err(Knm, orc.scale_rbf(prob["X"], pc["Z"], pc["raw_lengthscale"], pc["raw_outputscale"])) < 1e-13
    # K4: blocked Cholesky and inverse
    A = Kmm + 1e-6 * torch.eye(40, dtype=torch.float64)
    L, Li, st = ops.cholesky(A.to(DEV), want_inverse=True)
    assert int(st[0]) == 0
    assert rel_err(L.cpu() @ L.cpu().T, A) < 1e-13 and rel_err(Li.cpu() @ L.cpu(), torch.eye(40, dtype=torch.float64)) < 1e-9
    assert float(torch.triu(L.cpu(), 1).abs().max()) == 0.0
    # K5-K8: q(f) moments
    mu, v = ops.qf_moments(X, p["Z"], p["raw_lengthscale"], p["raw_outputscale"], p["m"], p["Lam"])
    mo, vo = orc.qf_moments(prob["X"], pc["Z"], pc["raw_lengthscale"], pc["raw_outputscale"], pc["m"], pc["Lam"])
    assert rel_err(mu.cpu(), mo) < 1e-9 and rel_err(v.cpu(), vo) < 1e-9
    # K9: whitened KL + gradients
    kl, gm, gL = ops.kl_whitened(p["m"], p["Lam"])
    m_, Lam_ = pc["m"].clone().requires_grad_(True), pc["Lam"].clone().requires_grad_(True)
    klo = orc.kld_whitened(m_, Lam_)
    klo.backward()
    assert rel_err(kl.cpu(), klo.detach()) < 1e-13 and rel_err(gm.cpu(), m_.grad) < 1e-13 and rel_err(gL.cpu(), Lam_.grad) < 1e-13
    # K10: SVGP ELL
    ell, g_eta, gmu, gv = ops.ell_gauss(Y, mu, v, p["log_var_noise"], scale=2.5)
    leaves = [t.clone().requires_grad_(True) for t in (mo, vo, pc["log_var_noise"])]
    e0 = 2.5 * orc.ell_gauss(prob["Y"].reshape(-1), *leaves)
    e0.backward()
    assert rel_err(ell.cpu(), e0.detach()) < 1e-12 and rel_err(gmu.cpu(), leaves[0].grad) < 1e-12
    assert rel_err(gv.cpu(), leaves[1].grad) < 1e-12 and rel_err(g_eta.cpu(), leaves[2].grad) < 1e-12
    # K11: quadrature ELL through the flow, all gradients
    flow = ops.FlowSpec(prob["program"], p["theta"].numel(), 0, DEV)
    res = ops.ell_flow(Y, mu, v, p["log_var_noise"], flow, p["theta"], 12, scale=2.5)
    leaves = [t.clone().requires_grad_(True) for t in (mo, vo, pc["log_var_noise"], pc["theta"])]
    e1 = 2.5 * orc.ell_flow(prob["Y"].reshape(-1), leaves[0], leaves[1], leaves[2], prob["program"], leaves[3],
                            prob["xs"], prob["ws"])
    e1.backward()
    assert rel_err(res["ell"].cpu(), e1.detach()) < 1e-11
    for got, want in ((res["g_mu"], leaves[0].grad), (res["g_v"], leaves[1].grad), (res["g_lvn"], leaves[2].grad),
                      (res["g_theta"], leaves[3].grad)):
        assert rel_err(got.cpu(), want) < 1e-9
    # flow evaluation: G, dG/df, log dG/df (forward_grad, flow.py:101-104)
    f = torch.linspace(-3, 3, 500, dtype=torch.float64)
    out = ops.flow_eval(f.to(DEV), flow, p["theta"])
    fr = f.clone().requires_grad_(True)
    G = orc.flow_forward(fr, prob["program"], pc["theta"])
    dG, = torch.autograd.grad(G.sum(), fr)
    assert rel_err(out["G"].cpu(), G.detach()) < 1e-12 and rel_err(out["dG"].cpu(), dG) < 1e-11
    assert rel_err(out["logdG"].cpu(), torch.log(dG)) < 1e-10


def test_adam_kernel_matches_torch_adam():
    from tgp.pytorch_amd import ops
    torch.manual_seed(1)
    p0 = torch.randn(1000, dtype=torch.float64)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=0.01, weight_decay=1e-5)
    p, m, v = p0.clone().to(DEV), torch.zeros(1000, dtype=torch.float64, device=DEV), torch.zeros(1000, dtype=torch.float64, device=DEV)
    for step in range(1, 6):
        g = torch.randn(1000, dtype=torch.float64)
        ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.to(DEV), m, v, step, lr=0.01, weight_decay=1e-5)
    assert rel_err(p.cpu(), ref.detach()) < 1e-14


@pytest.mark.parametrize("flow", ["tanh3x2", "sal2", None, "idsal3"])
def test_full_size_properties(flow):
    """At the BASELINE size (8611 x 4, M = 100, S = 32) the oracle is too slow to be the checker for every run, so
    check size-independent properties: row-shard additivity (the multi-GPU contract), run-to-run bit
    reproducibility (two-pass reductions, no float atomics), TGP == SVGP at the identity flow, KL at init."""
    from tgp.pytorch_amd import ops
    prob = orc.synthetic_problem(8611, 4, 100, seed=0, flow=flow, S=32)
    p = {k: v.to(DEV) for k, v in prob["params"].items()}
    X, Y = prob["X"].to(DEV), prob["Y"].to(DEV)
    rowp = prob["rowp"].to(DEV) if prob.get("rowp") is not None else None      # ID flow: per-row (a_n, b_n) columns
    RP = rowp.shape[1] if rowp is not None else 0
    spec = ops.FlowSpec(prob["program"], p["theta"].numel(), RP, DEV) if flow else None
    th = p.get("theta")

    def run(lo, hi, kl_scale=1.0, mbg=None):
        out, g, st, _ = ops.elbo_step(X[lo:hi], Y[lo:hi], p["Z"], p["raw_lengthscale"], p["raw_outputscale"], p["m"], p["Lam"],
                                      p["log_var_noise"], 8611.0, flow=spec, theta=th, S=32, kl_scale=kl_scale,
                                      mb_global=mbg, rowp=rowp[lo:hi].contiguous() if rowp is not None else None)
        assert int(st[0]) == 0
        return out.clone(), {k: t.clone() for k, t in g.items()}
    full, gfull = run(0, 8611)
    again, gagain = run(0, 8611)
    assert torch.equal(full, again) and all(torch.equal(gfull[k], gagain[k]) for k in gfull)      # bit reproducible
    spans = ((0, 2000), (2000, 4311), (4311, 6000), (6000, 8611))
    parts = [run(lo, hi, kl_scale=0.25, mbg=8611) for lo, hi in spans]
    assert rel_err(sum(o[1] for o, _ in parts).cpu(), full[1].cpu()) < 1e-12                      # ELL additive
    for k in gfull:
        if k == "rowp":       # per-row gradients: each shard produces its own rows
            assert rel_err(torch.cat([g[k] for _, g in parts]).cpu(), gfull[k].cpu()) < 1e-9
            continue
        assert rel_err(sum(g[k] for _, g in parts).cpu(), gfull[k].cpu()) < 1e-9, k                # gradients additive
    assert abs(float(full[0] - (full[1] - full[2]))) < 1e-9 * abs(float(full[0]))
    if flow == "sal2":
        ident = orc.synthetic_problem(8611, 4, 100, seed=0, flow="sal2", S=32, perturb=False)
        pi = {k: v.to(DEV) for k, v in ident["params"].items()}
        a, _, _, _ = ops.elbo_step(X, Y, pi["Z"], pi["raw_lengthscale"], pi["raw_outputscale"], pi["m"], pi["Lam"],
                                   pi["log_var_noise"], 8611.0, flow=spec, theta=pi["theta"], S=32)
        b, _, _, _ = ops.elbo_step(X, Y, pi["Z"], pi["raw_lengthscale"], pi["raw_outputscale"], pi["m"], pi["Lam"],
                                   pi["log_var_noise"], 8611.0)
        assert rel_err(a[0].cpu(), b[0].cpu()) < 1e-12                    # identity-initialised TGP == SVGP
        assert abs(float(a[2]) - 0.5 * (-100 * math.log(1e-5) + 100 * 1e-5 - 100)) < 1e-9      # KL at init = 525.6468


def test_predictive_sampling_shapes_and_moments():
    g = load_golden("med_sal2")
    model = build_model(g, "sal2")
    model.set_is_training(False)
    X = g["X"][:256].to(DEV)
    torch.manual_seed(0)
    samples, f_k, f_0 = model.sample_from_predictive_distribution(X, S=400)
    assert samples.shape == (1, 400, 256, 1) and f_k.shape == (1, 400 * 256) and f_0.shape == f_k.shape
    m1, m2, _, _ = model.predictive_distribution(X)
    err = (samples.mean(1).reshape(-1) - m1.reshape(-1)).abs() / m2.reshape(-1).sqrt()
    assert float(err.max()) < 0.35                    # |mean of 400 draws - m1| within ~7 standard errors


def test_flow_initialiser_runs_on_the_hip_kernels():
    """initializers.find_forward_params (code/dsp/initializers/initializers.py:29-109): MSE and its gradient come from
    the 1-node quadrature kernel; check them against autograd of the oracle flow, and that Adam drives the
    StepTanhL flow towards the identity map."""
    from tgp.pytorch_amd.flow import compile_flow, instance_flow
    from tgp.pytorch_amd.flows import StepTanhL
    from tgp.pytorch_amd.initializers import find_forward_params, flow_mse_and_grads
    np.random.seed(0)
    flow = instance_flow(StepTanhL(3, 2, add_f0=True))
    x = torch.linspace(-3, 3, 500, dtype=torch.float64)
    mse, grads, theta_list = flow_mse_and_grads(flow, x.to(DEV), x.to(DEV))
    spec = compile_flow(flow)[0]
    th = torch.stack([p.detach().reshape(()) for p in theta_list]).clone().requires_grad_(True)
    ref = ((orc.flow_forward(x, spec.blocks, th) - x) ** 2).mean()
    ref.backward()
    assert rel_err(mse.cpu(), ref.detach()) < 1e-10
    assert rel_err(torch.stack([g.reshape(()) for g in grads]).cpu(), th.grad) < 1e-9

    def fn():
        np.random.seed(1)
        return instance_flow(StepTanhL(3, 2, add_f0=True))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fitted, curve = find_forward_params(x.numpy(), x.numpy().copy(), fn, num_restarts=1, num_epochs=300)
    assert curve[-1] < 0.05 * curve[0]


@pytest.mark.parametrize("N,D,K,n_init", [(3000, 4, 40, 3), (8611, 4, 100, 1), (1200, 8, 17, 2)])
def test_kmeans_hip_matches_sklearn(N, D, K, n_init):
    """utils.KMEANS(backend='hip') (k-means++ seeding and Lloyd iterations on the GPU, sklearn's draws and stopping
    rule restated) against the reference's own call, sklearn.cluster.KMeans(..., random_state=seed) on the host."""
    from tgp.pytorch_amd import config as cg
    from tgp.pytorch_amd.utils import KMEANS
    cg.set_maximum_precission()
    g = torch.Generator().manual_seed(12)
    X = torch.randn(N, D, generator=g, dtype=torch.float64)
    X[:, 0] += 2.0 * (torch.arange(N) % 3)            # some structure
    Zs = KMEANS(X, K, n_init=n_init, seed=0, backend="sklearn").cpu()
    Zh, info = KMEANS(X.to(DEV), K, n_init=n_init, seed=0, backend="hip", return_info=True)
    Zh = Zh.cpu()
    assert Zh.shape == (K, D)
    assert rel_err(Zh, Zs) < 1e-8, (rel_err(Zh, Zs), info["n_iter"])


def _torch_mlps(X, W, spec, masks=None):
    """Reference: the nets as plain torch ops on the packed weights (Linear -> act -> dropout mask / (1-p))."""
    outs = []
    pw = spec.weights_per_net
    actf = torch.relu if spec.act == 0 else torch.tanh
    for k in range(spec.nnets):
        w = W[k * pw:(k + 1) * pw]
        o, h, nin = 0, X, spec.D
        for l in range(spec.L):
            Wl = w[o:o + spec.H * nin].reshape(spec.H, nin); o += spec.H * nin
            bl = w[o:o + spec.H]; o += spec.H
            h = actf(h @ Wl.T + bl)
            if masks is not None:
                h = h * masks[k][l] / (1.0 - spec.drop_p)
            nin = spec.H
        outs.append(h @ w[o:o + spec.H] + w[o + spec.H])
    return torch.stack(outs, 1)


@pytest.mark.parametrize("N,D,H,Lh,nnets,act,p", [(1000, 4, 50, 2, 6, "relu", 0.25), (333, 7, 17, 1, 2, "tanh", 0.1),
                                                   (129, 3, 32, 3, 3, "relu", 0.0),
                                                   # several 64-row chunks per workgroup, inputs and units of more than one tile
                                                   (40000, 20, 64, 2, 6, "relu", 0.1),
                                                   # H a multiple of 4 but not of 16: the last unit tile reads past the image
                                                   (500, 5, 52, 2, 2, "tanh", 0.2), (700, 33, 40, 3, 2, "tanh", 0.0)])
def test_mlp_kernels_match_torch(N, D, H, Lh, nnets, act, p):
    """tgp_mlp_forward/backward_f64 (the NNets of the input-dependent flows, flow.py:836-897) against torch autograd:
    eval mode, and training mode with the dropout mask restated on the host (ops.mlp_keep_mask)."""
    from tgp.pytorch_amd import ops
    g = torch.Generator().manual_seed(3)
    spec = ops.MlpSpec(D, H, Lh, nnets, act=act, drop_p=p, seed=1234)
    X = torch.randn(N, D, generator=g, dtype=torch.float64)
    W = (0.4 * torch.randn(nnets * spec.weights_per_net, generator=g, dtype=torch.float64)).requires_grad_(True)
    G = torch.randn(N, nnets, generator=g, dtype=torch.float64)
    step = torch.tensor([7, 0], dtype=torch.int32, device=DEV)
    for training in (False, True):
        masks = None
        if training and p > 0:
            masks = [[torch.from_numpy(ops.mlp_keep_mask(1234, 7, k, l, N, H, p)).to(torch.float64) for l in range(Lh)]
                     for k in range(nnets)]
        ref = _torch_mlps(X, W, spec, masks)
        gW_ref, = torch.autograd.grad((ref * G).sum(), W)
        out = ops.mlp_forward(spec, X.to(DEV), W.detach().to(DEV), training, step)
        gW = ops.mlp_backward(spec, X.to(DEV), W.detach().to(DEV), G.to(DEV), training, step)
        torch.cuda.synchronize()
        assert rel_err(out.cpu(), ref.detach()) < 1e-12, (training, rel_err(out.cpu(), ref.detach()))
        assert rel_err(gW.cpu(), gW_ref) < 1e-11, (training, rel_err(gW.cpu(), gW_ref))
    if p > 0:
        keep = ops.mlp_keep_mask(1234, 7, 0, 0, 4000, H, p)
        assert abs(keep.mean() - (1 - p)) < 0.01          # the mask keeps 1 - p of the units
        assert (ops.mlp_keep_mask(1234, 8, 0, 0, 4000, H, p) != keep).mean() > 0.05   # and changes with the step


@pytest.mark.parametrize("graph", [False, True])
def test_engine_id_tgp_hip_mlps_match_oracle_adam_history(graph):
    """ID_TGP with everything resident: MLPs (tgp_mlp_*_f64) -> fused ELBO step -> MLP backward -> grouped Adam
    (weight decay 1e-5 on the network weights, main.py:276-288), 4 steps, against the oracle + the same nets as torch
    ops + torch.optim.Adam with two parameter groups.  Dropout off (deterministic comparison)."""
    import bench
    from tgp.pytorch_amd.engine import ElboEngine
    w = dict(N=300, D=4, M=20, S=12, flow="idsal3", B=3, c=20, mlp=dict(H=50, L=2, p=0.25))
    prob = orc.synthetic_problem(w["N"], w["D"], w["M"], seed=4, flow="idsal3", S=w["S"])
    spec, W0 = bench.make_mlp(w, seed=0)
    leaves = {k: t.clone().requires_grad_(True) for k, t in prob["params"].items()}
    Wn = W0.clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": list(leaves.values())}, {"params": [Wn], "weight_decay": 1e-5}], lr=0.01)
    ref = []
    for _ in range(4):
        rowp = bench.torch_mlps(prob["X"], Wn, spec)
        elbo, ell, kl = orc.elbo(prob["X"], prob["Y"], leaves["Z"], leaves["raw_lengthscale"], leaves["raw_outputscale"],
                                 leaves["m"], leaves["Lam"], leaves["log_var_noise"], prob["N_total"], prob["program"],
                                 leaves.get("theta"), prob["xs"], prob["ws"], rowp)
        ref.append([float(elbo.detach()), float(ell.detach()), float(kl.detach())])
        opt.zero_grad()
        (-elbo).backward()
        opt.step()
    eng = ElboEngine(prob["X"], prob["Y"], prob["params"], float(prob["N_total"]), flow_blocks=prob["program"], S=w["S"],
                     device=DEV, mlp=spec, mlp_weights=W0, mlp_training=False)
    hist = []
    if graph:
        eng.capture()
    for _ in range(4):
        (eng.replay if graph else eng.step)()
        hist.append(list(eng.scalars()))
    eng.check_status()
    assert rel_err(torch.tensor(hist, dtype=torch.float64), torch.tensor(ref, dtype=torch.float64)) < 1e-8
    assert rel_err(eng.fp.view("nn").cpu(), Wn.detach()) < 1e-7
    assert rel_err(eng.fp.view("Z").cpu(), leaves["Z"].detach()) < 1e-7


@pytest.mark.parametrize("flow", ["tanh3x2", "idsal3"])
def test_unrolled_graph_equals_single_step_replays(flow):
    """engine.replay_many: U steps per graph launch (the second captured graph) + the remainder one by one, against the
    same number of single-step replays: scalars of EVERY step (the unrolled steps write theirs to engine.hist_u through
    the step's redirected `out` argument) and the final parameters, bit for bit."""
    import bench
    from tgp.pytorch_amd.engine import ElboEngine
    w = dict(N=700, D=4, M=30, S=12, flow=flow, B=3, c=20, mlp=dict(H=50, L=2, p=0.25))
    prob = orc.synthetic_problem(w["N"], w["D"], w["M"], seed=6, flow=flow, S=w["S"])
    kw = {}
    if flow == "idsal3":
        spec, W0 = bench.make_mlp(w, seed=0)
        kw = dict(mlp=spec, mlp_weights=W0, mlp_training=True)      # dropout on: the masks follow the device step counter

    def make():
        return ElboEngine(prob["X"], prob["Y"], prob["params"], float(prob["N_total"]), flow_blocks=prob["program"], S=w["S"],
                          device=DEV, **kw)
    n = 11                                       # 2 x U(4) + 3 single steps
    e1 = make()
    e1.capture(unroll=1)
    assert e1.gU is None
    h1 = torch.zeros(n, 3, dtype=torch.float64, device=DEV)
    e1.replay_many(n, h1)
    eu = make()
    eu.capture(unroll=4)
    assert eu.unroll == 4 and eu.gU is not None
    hu = torch.zeros(n, 3, dtype=torch.float64, device=DEV)
    eu.replay_many(n, hu)
    torch.cuda.synchronize()
    e1.check_status(); eu.check_status()
    assert torch.equal(h1, hu), (h1 - hu).abs().max()
    assert torch.equal(e1.fp.data, eu.fp.data)
    assert torch.equal(e1.fp.out[:3], eu.fp.out[:3]) and torch.equal(hu[-1], eu.fp.out[:3])
    assert float((hu[1:, 0] - hu[:-1, 0]).abs().min()) > 0.0          # eleven different steps, not one step eleven times
    # the default capture: a U-step graph and a 4 U-step graph for long runs -- 57 steps = 40 + 10 + 7 single ones
    # (one engine after the other, as above: engines of one shape share the cached workspace, and the rotated ID_TGP unit
    #  keeps the prepared step in it between capture and replay)
    n2 = 57
    ed = make()
    ed.capture()
    assert ed.unroll == 10 and ed.unroll_long == 40 and ed.gL is not None
    hd = torch.zeros(n2, 3, dtype=torch.float64, device=DEV)
    ed.replay_many(n2, hd)
    e2 = make()
    e2.capture(unroll=1)
    h2 = torch.zeros(n2, 3, dtype=torch.float64, device=DEV)
    e2.replay_many(n2, h2)
    torch.cuda.synchronize()
    assert torch.equal(hd, h2) and torch.equal(ed.fp.data, e2.fp.data)


def test_device_jitter_ladder_inside_the_captured_step():
    """The resident engine cannot ask the host to retry a failed Cholesky: psd_safe_cholesky's ladder (dsp/utils.py:256-269)
    runs inside k_prep_a.  Duplicated inducing points: without the ladder the status word reports the pivot; with it the
    step succeeds with jitter 1e-8, status[2] names the level, the engine warns once, and the result equals the step
    launched with that jitter up front."""
    from tgp.pytorch_amd import ops
    from tgp.pytorch_amd.engine import ElboEngine
    prob = orc.synthetic_problem(128, 3, 16, seed=1, flow=None, S=8)
    prob["params"]["Z"][1] = prob["params"]["Z"][0]
    e0 = ElboEngine(prob["X"], prob["Y"], prob["params"], 128.0, device=DEV, jitter_ladder=0.0)
    e0.elbo()
    torch.cuda.synchronize()
    assert int(e0.status[0]) > 0
    e1 = ElboEngine(prob["X"], prob["Y"], prob["params"], 128.0, device=DEV, jitter_ladder=1e-8)
    e1.elbo()
    with pytest.warns(ops.NumericalWarning):
        e1.check_status()
    assert int(e1.status[0]) == 0 and int(e1.status[2]) == 1
    p = {k: v.to(DEV) for k, v in prob["params"].items()}
    out, grads, status, _ = ops.elbo_step(prob["X"].to(DEV), prob["Y"].to(DEV), p["Z"], p["raw_lengthscale"],
                                           p["raw_outputscale"], p["m"], p["Lam"], p["log_var_noise"], 128.0, jitter=1e-8)
    assert torch.equal(e1.fp.out[:3].cpu(), out[:3].cpu())
    assert torch.equal(e1.fp.gview("Z").cpu(), grads["Z"].cpu())


def test_input_dependent_initialiser_runs_on_the_hip_mlp_kernels():
    """initializers.find_forward_params_input_dependent_flow (code/dsp/initializers/initializers.py:111-182): the epoch
    (MLP forward -> d loss -> MLP backward -> Adam) on the HIP kernels.  One epoch with dropout off against torch autograd
    of FLOW.forward_initializer + torch.optim.Adam; then the captured loop drives the nets to the scalars (dropout on)."""
    from tgp.pytorch_amd.data import DeviceLoader
    from tgp.pytorch_amd.flow import instance_flow, mlp_spec
    from tgp.pytorch_amd.flows import SAL
    from tgp.pytorch_amd.initializers import IdInitEngine, find_forward_params_input_dependent_flow, id_nets_and_targets

    def make():
        torch.manual_seed(3)
        f = instance_flow(SAL(3, input_dependent=True, input_dim=4, num_hidden_layers=2, batch_norm=0, dropout=0.25,
                              hidden_dim=50, hidden_activation="relu", inference="MC_dropout"))
        with torch.no_grad():                      # distinct targets per block so that a mix-up of columns shows
            for k, fl in enumerate(f.flow_arr):
                if hasattr(fl, "NNets_a"):
                    fl.a.fill_(0.1 * k)
                    fl.b.fill_(1.0 + 0.05 * k)
        return f.to(DEV)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(2000, 4, generator=g, dtype=torch.float64).to(DEV)
    # (a) one epoch, dropout off: loss, and the weights after one Adam step, against torch
    ref = make()
    ref.eval()
    opt = torch.optim.Adam([p for p in ref.parameters()], lr=0.01)
    loss = ref.forward_initializer(X)
    loss.backward()
    opt.step()
    hipf = make()
    nets, targets = id_nets_and_targets(hipf)
    assert len(nets) == 6
    spec = mlp_spec(nets, seed=0)
    eng = IdInitEngine(X, nets, targets, spec, lr=0.01)
    eng.d = spec.struct(X.shape[0], False)          # dropout off for the comparison
    got = eng.run(1)
    assert abs(got - float(loss)) < 1e-10 * abs(float(loss))
    eng.write_back()
    for a, b in zip(hipf.parameters(), ref.parameters()):
        assert rel_err(a.detach().cpu(), b.detach().cpu()) < 1e-9
    # (b) the entry point: 400 epochs with dropout, captured; the nets end near their targets, the scalars are switched off
    f2 = make()
    loader = DeviceLoader(X, torch.zeros(2000, 1, dtype=torch.float64), 10000, shuffle=True, device=DEV)
    f2, last = find_forward_params_input_dependent_flow(loader, FLOW=f2, num_epochs=400, noise_var=0.0)
    assert all(fl.parameters_are_turn_off for fl in f2.flow_arr if hasattr(fl, "NNets_a"))
    f2.eval()
    with torch.no_grad():
        for k, fl in enumerate(f2.flow_arr):
            if hasattr(fl, "NNets_a"):
                assert float((fl.NNets_a(X) - 0.1 * k).abs().mean()) < 0.05
                assert float((fl.NNets_b(X) - (1.0 + 0.05 * k)).abs().mean()) < 0.05
    assert last < 0.2


@pytest.mark.parametrize("graph", [False, True])
def test_engine_id_tgp_training_mode_dropout_matches_host_restated_masks(graph):
    """ID_TGP in TRAINING mode (dropout on, set_is_training(True): models/sparse_MF_SP.py:133-134, flow.py:949-965) through the
    resident engine: the keep masks are a counter-based hash of (seed, Adam step, net, layer, row, unit group), so the run
    is reproducible on the host -- oracle + the same nets as torch ops with ops.mlp_keep_mask(step) + torch Adam with
    the two parameter groups, 3 steps.  `graph`: the rotated captured unit (rows -> adjoint -> Adam -> prepare | MLP
    backward -> Adam(nets) -> MLP forward), whose mask counter is the network group's own."""
    from tgp.pytorch_amd import ops
    from tgp.pytorch_amd.engine import ElboEngine
    N, D, M, S = 320, 4, 24, 12
    prob = orc.synthetic_problem(N, D, M, seed=6, flow="idsal3", S=S)
    spec = ops.MlpSpec(D, 50, 2, 6, act="relu", drop_p=0.25, seed=77)
    g = torch.Generator().manual_seed(5)
    W0 = 0.25 * (2 * torch.rand(6 * spec.weights_per_net, generator=g, dtype=torch.float64) - 1)
    pw = spec.weights_per_net
    for k in range(6):                                   # a nets near 0, b nets near 1: a well-conditioned flow
        W0[(k + 1) * pw - 1] = float(k % 2)
    leaves = {k: t.clone().requires_grad_(True) for k, t in prob["params"].items()}
    Wn = W0.clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": list(leaves.values())}, {"params": [Wn], "weight_decay": 1e-5}], lr=0.01)
    ref = []
    for step in range(3):
        masks = [[torch.from_numpy(ops.mlp_keep_mask(spec.seed, step, k, l, N, spec.H, spec.drop_p)).to(torch.float64)
                  for l in range(spec.L)] for k in range(spec.nnets)]
        rowp = orc.mlp_rowp(prob["X"], Wn, D, 50, 2, 6, "relu", masks=masks, drop_p=0.25)
        e, l, k = orc.elbo(prob["X"], prob["Y"], leaves["Z"], leaves["raw_lengthscale"], leaves["raw_outputscale"], leaves["m"],
                           leaves["Lam"], leaves["log_var_noise"], prob["N_total"], prob["program"], leaves.get("theta"),
                           prob["xs"], prob["ws"], rowp)
        ref.append([float(e.detach()), float(l.detach()), float(k.detach())])
        opt.zero_grad()
        (-e).backward()
        opt.step()
    eng = ElboEngine(prob["X"], prob["Y"], prob["params"], float(prob["N_total"]), flow_blocks=prob["program"], S=S, device=DEV,
                     mlp=spec, mlp_weights=W0, mlp_training=True)
    hist = []
    if graph:
        eng.capture()
        assert eng.graph == "rotated"
    for _ in range(3):
        (eng.replay if graph else eng.step)()
        hist.append(list(eng.scalars()))
    eng.check_status()
    assert rel_err(torch.tensor(hist, dtype=torch.float64), torch.tensor(ref, dtype=torch.float64)) < 1e-8
    assert rel_err(eng.fp.view("nn").cpu(), Wn.detach()) < 1e-7
    assert rel_err(eng.fp.view("Z").cpu(), leaves["Z"].detach()) < 1e-7


@pytest.mark.parametrize("N", [1600, 2048, 4096, 4097, 8611])
def test_ell_gauss_stand_alone_at_every_size(N):
    """ops.ell_gauss / GaussianLinearMean.expected_log_prob size their workspace with tgp_ell_workspace_bytes: the old
    host formula was too small for 1600 < N <= 4096 and N > ~8100 (TGP_E_WORKSPACE)."""
    from tgp.pytorch_amd import ops
    g = torch.Generator().manual_seed(N)
    Y, mu = torch.randn(N, generator=g, dtype=torch.float64), torch.randn(N, generator=g, dtype=torch.float64)
    v = torch.rand(N, generator=g, dtype=torch.float64) + 0.1
    lvn = torch.tensor([-1.3], dtype=torch.float64)
    ell, g_eta, gmu, gv = ops.ell_gauss(Y.to(DEV), mu.to(DEV), v.to(DEV), lvn.to(DEV), scale=1.7)
    leaves = [t.clone().requires_grad_(True) for t in (mu, v, lvn)]
    e0 = 1.7 * orc.ell_gauss(Y, *leaves)
    e0.backward()
    assert rel_err(ell.cpu(), e0.detach()) < 1e-12 and rel_err(gmu.cpu(), leaves[0].grad) < 1e-12
    assert rel_err(gv.cpu(), leaves[1].grad) < 1e-12 and rel_err(g_eta.cpu(), leaves[2].grad) < 1e-12
    from tgp.pytorch_amd.likelihoods import GaussianLinearMean
    lik = GaussianLinearMean(1, 0.05, True).to(DEV)
    out = lik.expected_log_prob(Y.to(DEV).reshape(1, -1), mu.to(DEV).reshape(1, -1), v.to(DEV).reshape(1, -1))
    assert out.shape == (1,) and bool(torch.isfinite(out).all())


@pytest.mark.parametrize("N,M,D", [(1, 1, 1), (33, 5, 13), (1000, 101, 4), (517, 300, 16), (4099, 100, 8), (777, 512, 8),
                                   (130, 2, 16), (129, 256, 16), (5000, 1000, 8)])
def test_stand_alone_distance_kernels_every_tiling(N, M, D):
    """tgp_knm_f64 / tgp_kmm_f64 / tgp_kernel_matrix_f64 (the tiled kernel k_cov_tile: 32 x 128 blocks, column pairs,
    16-byte stores or the 8-byte path when the row stride is odd) against the oracle's gpytorch restatement: ragged row
    and column tiles, several column tiles, odd strides, D up to 16, both covariance functions, jitter on the diagonal."""
    from tgp.pytorch_amd import ops
    g = torch.Generator().manual_seed(1000 * N + M)
    X, Z = torch.randn(N, D, generator=g, dtype=torch.float64), torch.randn(M, D, generator=g, dtype=torch.float64)
    rl = torch.randn(D, generator=g, dtype=torch.float64) * 0.3 + 1.5
    ro = torch.tensor([0.7], dtype=torch.float64)
    Xd, Zd, rld, rod = X.to(DEV), Z.to(DEV), rl.to(DEV), ro.to(DEV)
    assert rel_err(ops.knm(Xd, Zd, rld, rod).cpu(), orc.scale_rbf(X, Z, rl, ro)) < 1e-13
    Kmm = ops.kmm(Zd, rld, rod, jitter=1e-3).cpu()
    assert rel_err(Kmm, orc.scale_rbf(Z, Z, rl, ro) + 1e-3 * torch.eye(M, dtype=torch.float64)) < 1e-13
    for name, fn in (("scale_rbf", orc.scale_rbf), ("scale_matern32", orc.scale_matern32)):
        assert rel_err(ops.kernel_matrix(Xd, Zd, rld, rod, kernel=name).cpu(), fn(X, Z, rl, ro)) < 1e-12, name
        Kxx = ops.kernel_matrix(Zd, None, rld, rod, kernel=name, jitter=0.25).cpu()
        assert rel_err(Kxx, fn(Z, Z, rl, ro) + 0.25 * torch.eye(M, dtype=torch.float64)) < 1e-12, name


def test_flow_networks_outside_the_mlp_kernel_raise():
    """One MLP implementation on the whole path: a network the HIP kernel does not cover (H = 100 > 64) raises instead of
    silently evaluating through torch.nn."""
    from tgp.pytorch_amd import lib
    from tgp.pytorch_amd.flow import compile_flow, instance_flow, nets_rowp
    from tgp.pytorch_amd.flows import SAL
    idf = instance_flow(SAL(1, input_dependent=True, input_dim=4, num_hidden_layers=2, batch_norm=0, dropout=0.25,
                            hidden_dim=100, hidden_activation="relu", inference="MC_dropout")).to(DEV)
    idf.turn_off_initializer_parameters()
    nets = compile_flow(idf)[2]
    with pytest.raises(lib.TgpError):
        nets_rowp(nets, torch.zeros(10, 4, dtype=torch.float64, device=DEV))
    ok = instance_flow(SAL(1, input_dependent=True, input_dim=4, num_hidden_layers=2, batch_norm=0, dropout=0.25,
                           hidden_dim=50, hidden_activation="relu", inference="MC_dropout")).to(DEV).eval()
    ok.turn_off_initializer_parameters()
    nets = compile_flow(ok)[2]
    Xr = torch.randn(77, 4, dtype=torch.float64, device=DEV)
    got = nets_rowp(nets, Xr)
    with torch.no_grad():
        want = torch.cat([n(Xr) for n in nets], -1)
    assert rel_err(got.cpu(), want.cpu()) < 1e-12
    f = torch.randn(5, 77, dtype=torch.float64, device=DEV)
    assert ok(f, Xr).shape == f.shape


@pytest.mark.parametrize("flowname,S,N", [("tanh3x2", 1, 5000), ("sal2", 7, 1333), ("idsal3", 3, 700)])
def test_fused_flow_logdet_matches_elementwise_and_autograd(flowname, S, N):
    """tgp_flow_logdet_f64: G and the log-Jacobian sum in one pass == the sum of tgp_flow_eval_f64's per-element log dG/df
    == torch autograd through the oracle's flow; bit-reproducible."""
    from tgp.pytorch_amd import ops
    prob = orc.synthetic_problem(N, 4, 8, seed=11, flow=flowname, S=4)
    theta = prob["params"]["theta"]
    rowp = prob.get("rowp")
    RP = rowp.shape[1] if rowp is not None else 0
    flow = ops.FlowSpec(prob["program"], theta.numel(), RP, DEV)
    g = torch.Generator().manual_seed(3)
    f = torch.randn(S, N, generator=g, dtype=torch.float64)
    rd = rowp.to(DEV) if rowp is not None else None
    tot, G = ops.flow_logdet(f.to(DEV), flow, theta.to(DEV), rd, want_G=True)
    tot2, _ = ops.flow_logdet(f.to(DEV), flow, theta.to(DEV), rd)
    assert torch.equal(tot, tot2)
    ev = ops.flow_eval(f.to(DEV), flow, theta.to(DEV), rd)
    assert rel_err(tot.cpu(), ev["logdG"].sum().cpu()) < 1e-13 and torch.equal(G, ev["G"])
    fa = f.clone().requires_grad_(True)
    Ga = orc.flow_forward(fa, prob["program"], theta, rowp)
    (dGa,) = torch.autograd.grad(Ga.sum(), fa)
    assert rel_err(G.cpu(), Ga.detach()) < 1e-12 and rel_err(tot.cpu(), torch.log(dGa).sum()) < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,M,kernel", [(300, 5, 40, "scale_rbf"), (777, 4, 100, "scale_rbf"), (64, 13, 5, "scale_rbf"),
                                           (700, 6, 200, "scale_rbf"), (500, 3, 150, "scale_matern32")])
def test_qf_moments_adjoint_matches_oracle_autograd(N, D, M, kernel):
    """tgp_qf_moments_bwd_f64 (fused path for M <= 128, general-M above) against autograd through the oracle's restatement of
    models/sparse_MF_SP.py:274-396, for random adjoints of (mu, v); then the same through the autograd Function and through
    the model's (now differentiable) marginal_variational_qf_parameters / KLD."""
    from tgp.pytorch_amd import ops
    prob = orc.synthetic_problem(N, D, M, seed=11, flow=None)
    pc = {k: v.clone().requires_grad_(True) for k, v in prob["params"].items() if k in ("Z", "raw_lengthscale", "raw_outputscale", "m", "Lam")}
    gen = torch.Generator().manual_seed(5)
    mub, vb = torch.randn(N, generator=gen, dtype=torch.float64), torch.randn(N, generator=gen, dtype=torch.float64)
    mo, vo = orc.qf_moments(prob["X"], pc["Z"], pc["raw_lengthscale"], pc["raw_outputscale"], pc["m"], pc["Lam"], kernel=kernel)
    ((mo * mub).sum() + (vo * vb).sum()).backward()
    p = {k: v.detach().to(DEV) for k, v in pc.items()}
    X = prob["X"].to(DEV)
    g = ops.qf_moments_bwd(X, p["Z"], p["raw_lengthscale"], p["raw_outputscale"], p["m"], p["Lam"], mub.to(DEV), vb.to(DEV),
                           kernel=kernel)
    names = {"Z": "Z", "raw_ls": "raw_lengthscale", "raw_os": "raw_outputscale", "m": "m", "Lam": "Lam"}
    for k, kk in names.items():
        assert rel_err(g[k].cpu().reshape(pc[kk].shape), pc[kk].grad) < 1e-7, k       # tolerance of the gradient fixtures
    assert float(torch.triu(g["Lam"].cpu().reshape(M, M), 1).abs().max()) == 0.0     # tril mask applied at use (:344-345)
    # the autograd Function: same numbers through torch's engine, X without gradient
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    mu, v = ops.QfMomentsFunction.apply(X, q["Z"], q["raw_lengthscale"], q["raw_outputscale"], q["m"], q["Lam"], kernel)
    assert rel_err(mu.detach().cpu(), mo.detach()) < 1e-9 and rel_err(v.detach().cpu(), vo.detach()) < 1e-8
    ((mu * mub.to(DEV)).sum() + (v * vb.to(DEV)).sum()).backward()
    for k, kk in names.items():
        assert torch.equal(q[kk].grad.reshape(-1), g[k].reshape(-1)), k


@pytest.mark.gpu
def test_model_moments_and_kl_are_differentiable_like_the_references():
    """marginal_variational_qf_parameters / KLD called under autograd outside ELBO() (the reference's are plain torch code):
    gradients reach Z, the kernel hyper-parameters and q(u), and equal the oracle's."""
    from tgp.pytorch_amd import ops
    g = load_golden("power_sal2")
    model = build_model(g, "sal2")
    X = g["X"].to(DEV)
    X3 = X[:512].unsqueeze(0)
    mu, cov = model.marginal_variational_qf_parameters(X3, diagonal=True, is_duvenaud=False)
    assert mu.requires_grad and cov.requires_grad and mu.shape == (1, 512, 1)
    loss = (mu ** 2).sum() + cov.sum() + 0.5 * model.KLD().sum()
    loss.backward()
    Z, rl, ro, m, Lam, _ = model._gp_params()
    pc = [t.detach().cpu().clone().requires_grad_(True) for t in (Z, rl, ro, m, Lam)]
    mo, vo = orc.qf_moments(X[:512].cpu(), *pc)
    ((mo ** 2).sum() + vo.sum() + 0.5 * orc.kld_whitened(pc[3], pc[4])).backward()
    got = [model.Z.grad[0], model.covariance_function.base_kernel.raw_lengthscale.grad.reshape(-1),
           model.covariance_function.raw_outputscale.grad.reshape(-1), model.q_U.variational_mean.grad[0],
           model.q_U.chol_variational_covar.grad[0]]
    for a, b in zip(got, pc):
        assert rel_err(a.cpu().reshape(b.shape), b.grad) < 1e-7
    with torch.no_grad():
        mu2, _ = model.marginal_variational_qf_parameters(X3, diagonal=True, is_duvenaud=False)
    assert not mu2.requires_grad and torch.equal(mu2, mu.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("M", [5, 40, 128, 300, 1000])
def test_cholesky_adjoint_matches_torch(M):
    """tgp_cholesky_bwd_f64 against torch.linalg.cholesky's backward (what the reference's psd_safe_cholesky replays,
    dsp/utils.py:239), for a dense factor adjoint whose strictly-upper part must be ignored; then through the autograd Function
    and through ops.psd_safe_cholesky."""
    from tgp.pytorch_amd import ops
    gen = torch.Generator().manual_seed(M)
    B = torch.randn(M, M + 3, generator=gen, dtype=torch.float64)
    A = (B @ B.T / M + 0.5 * torch.eye(M, dtype=torch.float64)).requires_grad_(True)
    Lbar = torch.randn(M, M, generator=gen, dtype=torch.float64)
    Lc = torch.linalg.cholesky(A)
    (Lc * Lbar.tril()).sum().backward()
    Lo, Li, st = ops.cholesky(A.detach().to(DEV), want_inverse=True)
    assert int(st[0]) == 0
    Ab = ops.cholesky_bwd(Lo, Li, Lbar.to(DEV)).cpu()
    assert rel_err(Ab, A.grad) < 1e-9
    assert float((Ab - Ab.T).abs().max()) <= 1e-12 * float(Ab.abs().max())
    Ad = A.detach().to(DEV).requires_grad_(True)
    L2 = ops.CholeskyFunction.apply(Ad)
    (L2 * Lbar.to(DEV).tril()).sum().backward()
    assert torch.equal(Ad.grad.cpu(), Ab)
    Ad.grad = None
    L3, used = ops.psd_safe_cholesky(Ad)
    assert L3.requires_grad and used is Ad
    (L3 * Lbar.to(DEV).tril()).sum().backward()
    assert torch.equal(Ad.grad.cpu(), Ab)


def test_likelihood_expected_log_prob_is_differentiable():
    """GaussianLinearMean / GaussianNonLinearMean.expected_log_prob carry autograd (the reference's are plain torch code,
    likelihoods/GaussianLinearMean.py:60-87, GaussianNonLinearMean.py:64-150): gradients w.r.t. the moments, the noise and
    the flow's parameters against autograd through the oracle."""
    from tgp.pytorch_amd.likelihoods import GaussianLinearMean, GaussianNonLinearMean
    from tgp.pytorch_amd.flow import instance_flow
    from tgp.pytorch_amd.flows import SAL
    torch.manual_seed(0)
    N = 300
    Y = torch.randn(N, 1, dtype=torch.float64)
    mu = torch.randn(1, N, 1, dtype=torch.float64) * 0.5
    v = torch.rand(1, N, 1, dtype=torch.float64) * 0.3 + 0.05
    # ---- closed form
    lik = GaussianLinearMean(1, 0.05, False).to(DEV)
    mu_d, v_d = mu.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)
    ell = lik.expected_log_prob(Y.to(DEV).t(), mu_d, v_d)
    ell.sum().backward()
    mu_o, v_o = mu.reshape(-1).clone().requires_grad_(True), v.reshape(-1).clone().requires_grad_(True)
    lvn_o = lik.log_var_noise.detach().cpu().reshape(-1).clone().requires_grad_(True)
    want = orc.ell_gauss(Y.reshape(-1), mu_o, v_o, lvn_o)
    want.backward()
    assert rel_err(ell.detach().cpu(), want.detach().reshape(1)) < 1e-10
    assert rel_err(mu_d.grad.reshape(-1).cpu(), mu_o.grad) < 1e-9 and rel_err(v_d.grad.reshape(-1).cpu(), v_o.grad) < 1e-9
    assert rel_err(lik.log_var_noise.grad.reshape(-1).cpu(), lvn_o.grad) < 1e-9
    # ---- quadrature through a SAL x 2 flow
    G = instance_flow(SAL(2)).to(DEV)
    with torch.no_grad():
        for q in G.parameters():
            q.add_(0.2 * torch.randn_like(q))
    lik2 = GaussianNonLinearMean(1, 0.05, False, 16).to(DEV)
    mu_d, v_d = mu.to(DEV).requires_grad_(True), v.to(DEV).requires_grad_(True)
    ell2 = lik2.expected_log_prob(Y.to(DEV).t(), mu_d, v_d, flow=[G], X=torch.zeros(1, N, 2, dtype=torch.float64, device=DEV))
    ell2.sum().backward()
    from tgp.pytorch_amd.flow import compile_flow
    spec, theta_list, _ = compile_flow(G)
    th_o = torch.stack([q.detach().cpu().reshape(()) for q in theta_list]).clone().requires_grad_(True)
    mu_o, v_o = mu.reshape(-1).clone().requires_grad_(True), v.reshape(-1).clone().requires_grad_(True)
    lvn_o = lik2.log_var_noise.detach().cpu().reshape(-1).clone().requires_grad_(True)
    xs, ws = orc.hermgauss(16)
    want2 = orc.ell_flow(Y.reshape(-1), mu_o, v_o, lvn_o, orc.sal_program(2)[0], th_o, xs, ws)
    want2.backward()
    assert rel_err(ell2.detach().cpu(), want2.detach().reshape(1)) < 1e-10
    assert rel_err(mu_d.grad.reshape(-1).cpu(), mu_o.grad) < 1e-8 and rel_err(v_d.grad.reshape(-1).cpu(), v_o.grad) < 1e-8
    assert rel_err(lik2.log_var_noise.grad.reshape(-1).cpu(), lvn_o.grad) < 1e-8
    got_th = torch.stack([q.grad.reshape(()) for q in theta_list]).cpu()
    assert rel_err(got_th, th_o.grad) < 1e-8
