"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN FILES (build container only).

Run:  python oracle/gen_golden.py            (needs /root/reference; never runs on the GPU box)

The reference (jmaronas/TGP.pytorch, pure Python) is imported unmodified from /root/reference/code
with the builder-written third-party stand-ins in oracle/shims on sys.path (gpytorch, pytorchlib,
torchvision, ... are absent from this image).  Inputs come from
oracle/tgp_oracle.synthetic_problem (seeded), are pushed into the reference's model objects, and
the reference's outputs (ELBO/ELL/KLD, q(f) moments, all gradients, evaluation quantities, first
Adam steps) are stored next to the inputs.  Fixtures are data only: no reference source travels.
"""
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.dont_write_bytecode = True            # /root/reference is read-only
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, "/root/reference/code")
sys.path.insert(0, REPO)
warnings.simplefilter("ignore")

import numpy as np                         # noqa: E402
import torch                               # noqa: E402
import scipy.integrate as _si              # noqa: E402

_si.cumtrapz = _si.cumulative_trapezoid    # dsp/utils.py:26 imports the removed name
_real_version = torch.__version__
torch.__version__ = "1.7.0"                # dsp/config.py:18-25,74 hard version gate
import dsp.config as cg                    # noqa: E402

torch.__version__ = _real_version
cg.device = "cpu"
cg.set_maximum_precission()                # float64 like code/main.py:124

from dsp.models import instance_kernel, sparse_MF_SP, sparse_MF_GP     # noqa: E402
from dsp.models.flow import instance_flow                              # noqa: E402
from dsp.likelihoods import GaussianNonLinearMean, GaussianLinearMean  # noqa: E402
from dsp.flows import SAL, StepTanhL                                   # noqa: E402

from oracle import tgp_oracle as orc                                   # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")
# `--out DIR`: write the fixtures somewhere else (a reproducibility check regenerates into a scratch directory and compares)
OUT = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else GOLDEN
IP = {"variational_distribution": {"variance_scale": 1e-5, "mean_scale": 0.0}}
KINIT = {"length_scale": 2.0, "kernel_scale": 2.0, "noisy_variance": 1e-6}


def build_reference_model(prob, flow, kernel="scale_rbf"):
    """Instantiate the reference's model classes and overwrite their parameters with prob['params']."""
    X, p = prob["X"], prob["params"]
    N, D = X.shape
    M = p["Z"].shape[0]
    S = prob["xs"].shape[0]
    K = instance_kernel(kernel, ard_num_dim=D, num_multioutput=1, kernel_is_shared=False, init_params=KINIT)
    if flow is None:
        lik = GaussianLinearMean(out_dim=1, noise_init=0.05, noise_is_shared=False)
        model = sparse_MF_GP(["zero", K], X, p["Z"].clone(), N, lik, 1, True, False, False, False, False, 0.0,
                             init_params=IP)
    else:
        lik = GaussianNonLinearMean(out_dim=1, noise_init=0.05, noise_is_shared=False, quadrature_points=S)
        if flow.startswith("sal"):
            specs = SAL(int(flow[3:]))
        elif flow.startswith("idsal"):
            specs = SAL(int(flow[5:]), input_dependent=True, input_dim=D, num_hidden_layers=2, batch_norm=0,
                        dropout=0.25, hidden_dim=50, hidden_activation="relu", inference="MC_dropout")
            specs = instance_flow(specs)
            specs.turn_off_initializer_parameters()
        else:
            nb, ns = (int(t) for t in flow[4:].split("x"))
            np.random.seed(0)
            specs = instance_flow(StepTanhL(nb, ns, add_f0=True))
        model = sparse_MF_SP(["zero", K], X, p["Z"].clone(), N, lik, 1, True, False, False, False, False,
                             [specs], "single", 0.0, init_params=IP)
    with torch.no_grad():
        model.Z.data = p["Z"].reshape(1, M, D).clone()
        model.q_U.variational_mean.data = p["m"].reshape(1, M).clone()
        model.q_U.chol_variational_covar.data = p["Lam"].reshape(1, M, M).clone()
        model.covariance_function.raw_outputscale.data = p["raw_outputscale"].reshape(1).clone()
        model.covariance_function.base_kernel.raw_lengthscale.data = p["raw_lengthscale"].reshape(1, 1, D).clone()
        model.likelihood.log_var_noise.data = p["log_var_noise"].reshape(1, 1).clone()
        if flow is not None:
            load_theta(model, prob["program"], p["theta"])
    return model


def flow_scalar_params(model, program):
    """Reference nn.Parameters in the order of the flat theta vector."""
    arr = model.G_matrix[0].flow_arr
    out = []
    for blk, (kind, K, poff, flags) in zip(arr, program):
        if flags & orc.FLAG_PER_ROW:
            continue
        if kind == orc.FLOW_STEPTANH:
            for t in blk.flow_arr:
                out += [t.a, t.b, t.c, t.d]
        else:
            out += [blk.a, blk.b]
    return out


def load_theta(model, program, theta):
    for prm, val in zip(flow_scalar_params(model, program), theta):
        prm.data = val.clone().reshape(())


def reference_step0(prob, flow, name, kernel="scale_rbf"):
    """ELBO/ELL/KLD + q(f) moments + every gradient from the reference model."""
    torch.manual_seed(0)
    model = build_reference_model(prob, flow, kernel)
    model.set_is_training(True)
    X, Y, p = prob["X"], prob["Y"], prob["params"]
    captured = {}
    if flow is not None and flow.startswith("idsal"):
        model.eval()                         # dropout off => deterministic per-row parameters
        hooks = []
        for bi, blk in enumerate(model.G_matrix[0].flow_arr):
            if hasattr(blk, "NNets_a"):
                for nm in ("a", "b"):
                    def hook(mod, inp, out, key=(bi, nm)):
                        if key not in captured and out.requires_grad:
                            out.retain_grad()
                            captured[key] = out
                    hooks.append(getattr(blk, "NNets_" + nm).register_forward_hook(hook))
    elbo, ell, kld = model.ELBO(X, Y)
    (elbo).backward()
    out = {
        "X": X, "Y": Y, "xs": prob["xs"], "ws": prob["ws"], "N_total": np.float64(prob["N_total"]),
        "ELBO": elbo.detach(), "ELL": ell.detach(), "KLD": kld.detach(),
        "g_Z": model.Z.grad[0], "g_m": model.q_U.variational_mean.grad[0],
        "g_Lam": model.q_U.chol_variational_covar.grad[0],
        "g_raw_outputscale": model.covariance_function.raw_outputscale.grad,
        "g_raw_lengthscale": model.covariance_function.base_kernel.raw_lengthscale.grad.reshape(-1),
        "g_log_var_noise": model.likelihood.log_var_noise.grad.reshape(-1),
    }
    for k, v in p.items():
        out["p_" + k] = v
    if kernel != "scale_rbf":
        out["kernel"] = np.array(kernel)
    if flow is not None:
        out["program"] = np.array(prob["program"], dtype=np.int32)
        out["g_theta"] = torch.stack([q.grad.reshape(()) for q in flow_scalar_params(model, prob["program"])])
    if captured:
        keys = sorted(captured.keys())
        out["rowp"] = torch.stack([captured[k].detach().reshape(-1) for k in keys], 1)
        out["g_rowp"] = torch.stack([captured[k].grad.reshape(-1) for k in keys], 1)
    with torch.no_grad():
        mu, v = model.marginal_variational_qf_parameters(X, diagonal=True, is_duvenaud=False, init_Z=None)
        out["mu"], out["v"] = mu.reshape(-1), v.reshape(-1)
        if not captured:
            # evaluation path (sparse_MF_SP.py:457-540, 637-825); skipped for the ID fixture whose
            # per-row parameters are only defined through the captured tensors above
            model.set_is_training(False)
            Y_std = torch.tensor([1.7])
            logp, (m1, m2) = model.test_log_likelihood(X, Y, return_moments=True, Y_std=Y_std, S_MC_NNet=None)
            out["test_logp_sum"], out["pred_m1"], out["pred_m2"] = logp.reshape(-1), m1.reshape(-1), m2.reshape(-1)
            out["Y_std"] = Y_std
    save(name, out)
    return model


def reference_adam_steps(prob, flow, name, steps=5):
    """First `steps` of Trainer_base.train's inner loop: ELBO -> (-ELBO).backward() -> Adam(lr=0.01).step()
    (trainers/trainer_base.py:337-342, optimizers.py:12)."""
    model = build_reference_model(prob, flow)
    model.set_is_training(True)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    hist = []
    for _ in range(steps):
        elbo, ell, kld = model.ELBO(prob["X"], prob["Y"])
        loss = -elbo
        opt.zero_grad()
        loss.backward()
        opt.step()
        hist.append([elbo.item(), ell.item(), kld.item()])
    out = {"X": prob["X"], "Y": prob["Y"], "xs": prob["xs"], "ws": prob["ws"],
           "N_total": np.float64(prob["N_total"]), "history": np.array(hist)}
    for k, v in prob["params"].items():
        out["p_" + k] = v
    if flow is not None:
        out["program"] = np.array(prob["program"], dtype=np.int32)
        out["final_theta"] = torch.stack([q.detach().reshape(()) for q in flow_scalar_params(model, prob["program"])])
    out["final_Z"] = model.Z.detach()[0]
    out["final_m"] = model.q_U.variational_mean.detach()[0]
    save(name, out)


def cholesky_ladder_fixture():
    """dsp/utils.py:222-270 on a numerically singular K: which jitter the reference's ladder lands on."""
    from dsp.utils import psd_safe_cholesky
    g = torch.Generator().manual_seed(3)
    B = torch.randn(12, 3, generator=g)
    A = B @ B.t()                                  # rank 3 => plain Cholesky fails
    L, Ap = psd_safe_cholesky(A.clone(), upper=False, jitter=None)
    save("chol_ladder", {"A": A, "L": L, "A_used": Ap, "jitter_used": (Ap - A).diagonal().mean()})


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    arrs = {}
    for k, v in d.items():
        arrs[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print("wrote", name, {k: a.shape for k, a in list(arrs.items())[:4]}, "...")


def matern_fixtures():
    """'scale_matern32' (utils_models.py:199-204; main.py never selects it): the reference's model classes run
    unmodified on top of the builder-written gpytorch stand-in for MaternKernel (third-party arithmetic: unpinned)."""
    for name, kw in (("tiny_matern_svgp", dict(N=64, D=3, M=8, S=8, flow=None)),
                     ("med_matern_sal2", dict(N=512, D=4, M=60, S=16, flow="sal2")),
                     ("med_matern_tanh2x2", dict(N=300, D=6, M=40, S=12, flow="tanh2x2"))):
        prob = orc.synthetic_problem(kw["N"], kw["D"], kw["M"], seed=0, flow=kw["flow"], S=kw["S"])
        reference_step0(prob, kw["flow"], name, kernel="scale_matern32")


# ---------------------------------------------------------------------------------------------------
# real data (the reference's own CSV + split pickle) and full-size (N=8611, M=100) fixtures
# ---------------------------------------------------------------------------------------------------
def real_dataset_fixture(name):
    """The reference's loader, called as code/main.py:134 does -> tests/golden/<name>_seed1.npz
    (split indices, z-scored tensors, Y_std, KMeans centres of the reference's KMEANS).  Data, not source."""
    import pickle
    from dsp.data import return_dataset
    from dsp.utils import KMEANS
    opts = {"shuffle_train": True, "split_from_disk": True, "use_generator": True, "n_workers": 0}
    loaders, dc = return_dataset(name, 10000, use_validation=None, seed=1, options=opts)
    with open("/root/reference/code/datasets/regression/uci/splits_idx_%s.pkl" % name, "rb") as fh:
        sp = pickle.load(fh)["seed_1"]
    M = {"power": 100, "boston": 5}[name]
    # The inducing points are the COMMITTED ones when the fixture exists: the reference's KMEANS is sklearn's,
    # whose centres differ by one ulp from run to run, and every full-size fixture is built on them -- re-running sklearn made
    # each regeneration drift by 1e-11 (3e-3 absolute in the Adam history that starts at m = 0).  `--rekmeans` draws them anew
    # and checks them against the committed ones at 1e-12.
    have = os.path.join(GOLDEN, name + "_seed1.npz")
    if os.path.exists(have) and "--rekmeans" not in sys.argv:
        Z1 = torch.tensor(np.load(have)["Z_kmeans_n1_seed0"])
    else:
        Z1 = KMEANS(dc["X_tr"], M, n_init=1, seed=0)
        if os.path.exists(have):
            Zc = np.load(have)["Z_kmeans_n1_seed0"]
            assert np.abs(np.asarray(Z1) - Zc).max() <= 1e-12 * np.abs(Zc).max(), "KMEANS moved by more than rounding"
    out = {"train_idx": np.asarray(sp["train"]), "test_idx": np.asarray(sp["test"]), "X_tr": dc["X_tr"], "Y_tr": dc["Y_tr"],
           "X_te": dc["X_te"], "Y_te": dc["Y_te"], "Y_std": np.asarray(dc["Y_std"], dtype=np.float64).reshape(-1),
           "Z_kmeans_n1_seed0": Z1, "N_tr": np.int64(dc["N_tr"]), "N_te": np.int64(dc["N_te"])}
    # validation split as the reference draws it (uci_datasets.py:54-56): only the indices, to pin random_split_validation
    # (datasets.py:179 raises NameError('toy_list') for use_validation != None, so the dataset class is called directly)
    from dsp.data.uci_datasets import Power, Boston
    ds = {"power": Power, "boston": Boston}[name](1, use_validation=[3, 50], split_from_disk=True)
    out["val_X_va_head"], out["val_Y_std"] = ds.X_va[:8], np.asarray(ds.Y_std).reshape(-1)
    save(name + "_seed1", out)
    return dc, Z1


def uci_loader_fixtures():
    """The other regression sets whose CSV + split pickle ship with the reference (uci_datasets.py:186-283), through the
    reference's own return_dataset -> ONE compact fixture tests/golden/uci_loaders_seed1.npz: split indices (bit-exact
    check), shapes, Y_std and every 97th z-scored row of each split.  Data, not source."""
    from dsp.data import return_dataset
    opts = {"shuffle_train": True, "split_from_disk": True, "use_generator": True, "n_workers": 0}
    out = {}
    for name in ("concrete", "kin8nm", "energy", "wine_red", "wine_white", "naval"):
        loaders, dc = return_dataset(name, 10000, use_validation=None, seed=1, options=opts)
        ds = loaders[0].dataset
        stem = {"wine_red": "wine-red", "wine_white": "wine-white"}.get(name, name)
        import pickle
        with open("/root/reference/code/datasets/regression/uci/splits_idx_%s.pkl" % stem, "rb") as fh:
            sp = pickle.load(fh)["seed_1"]
        out[name + ".train_idx"] = np.asarray(sp["train"]).astype(np.int32)
        out[name + ".test_idx"] = np.asarray(sp["test"]).astype(np.int32)
        out[name + ".shape"] = np.asarray([dc["N_tr"], dc["N_te"], dc["X_tr"].shape[1]], dtype=np.int64)
        out[name + ".Y_std"] = np.asarray(dc["Y_std"], dtype=np.float64).reshape(-1)
        for k in ("X_tr", "Y_tr", "X_te", "Y_te"):
            out[name + "." + k + "_every97"] = np.asarray(dc[k])[::97]
    save("uci_loaders_seed1", out)


def problem_on(dc, Z, flow, S, perturb, seed=0):
    """orc.synthetic_problem's parameter recipe on real rows: same generators, X/Y/Z replaced."""
    N, D = dc["X_tr"].shape
    prob = orc.synthetic_problem(N, D, Z.shape[0], seed=seed, flow=flow, S=S, perturb=perturb)
    prob["X"], prob["Y"] = dc["X_tr"].clone(), dc["Y_tr"].clone()
    prob["params"]["Z"] = Z.clone()
    prob["N_total"] = float(N)
    return prob


def id_model_with_weights(prob, seed=0):
    """ID_TGP (SAL x 3, MLPs 4->50->50->1, code/exp_config.py:31-55) with seeded network weights whose output
    layers are shrunk around the identity flow (a_n ~ 0, b_n ~ 1) -- the state find_forward_params_input_dependent_flow
    leaves them in -- so the flow is well conditioned.  Eval mode: dropout off, deterministic."""
    torch.manual_seed(seed)
    model = build_reference_model(prob, "idsal3")
    with torch.no_grad():
        for blk in model.G_matrix[0].flow_arr:
            if hasattr(blk, "NNets_a"):
                for nm, bias in (("a", 0.0), ("b", 1.0)):
                    last = list(getattr(blk, "NNets_" + nm))[-1].w
                    last.weight.mul_(0.3)
                    last.bias.fill_(bias)
    model.set_is_training(True)
    model.eval()
    return model


def nn_params(model):
    """The MLP parameters in the packed order of the product (tgp/pytorch_amd/flow.py compile_flow: block by block,
    NNets_a then NNets_b; per layer weight then bias)."""
    out = []
    for blk in model.G_matrix[0].flow_arr:
        if hasattr(blk, "NNets_a"):
            out += list(blk.NNets_a.parameters()) + list(blk.NNets_b.parameters())
    return out


def full_size_fixture(dc, dte, Z, flow, name, perturb=True, steps=5):
    """Reference step 0 (values, every gradient), the evaluation path on the TEST split, and `steps` Adam steps, at
    the full training-split size.  X/Y are not stored again: 'data' names the <dataset>_seed1 fixture."""
    S = 32
    prob = problem_on(dc, Z, flow, S, perturb)
    is_id = flow is not None and flow.startswith("idsal")
    model = id_model_with_weights(prob) if is_id else build_reference_model(prob, flow)
    model.set_is_training(True)
    X, Y = prob["X"], prob["Y"]
    elbo, ell, kld = model.ELBO(X, Y)
    elbo.backward()
    out = {"data": np.array(dte), "xs": prob["xs"], "ws": prob["ws"], "N_total": np.float64(prob["N_total"]),
           "ELBO": elbo.detach(), "ELL": ell.detach(), "KLD": kld.detach(),
           "g_Z": model.Z.grad[0], "g_m": model.q_U.variational_mean.grad[0],
           "g_Lam": model.q_U.chol_variational_covar.grad[0],
           "g_raw_outputscale": model.covariance_function.raw_outputscale.grad,
           "g_raw_lengthscale": model.covariance_function.base_kernel.raw_lengthscale.grad.reshape(-1),
           "g_log_var_noise": model.likelihood.log_var_noise.grad.reshape(-1)}
    for k, v in prob["params"].items():
        out["p_" + k] = v
    if flow is not None:
        out["program"] = np.array(prob["program"], dtype=np.int32)
        th = flow_scalar_params(model, prob["program"])
        if th:
            out["g_theta"] = torch.stack([q.grad.reshape(()) for q in th])
    if is_id:
        out["nn_W"] = torch.cat([q.detach().reshape(-1) for q in nn_params(model)])
        out["g_nn_W"] = torch.cat([q.grad.reshape(-1) for q in nn_params(model)])
        with torch.no_grad():
            cols = []
            for blk in model.G_matrix[0].flow_arr:
                if hasattr(blk, "NNets_a"):
                    cols += [blk.NNets_a(X[:256]).reshape(-1), blk.NNets_b(X[:256]).reshape(-1)]
            out["rowp_head"] = torch.stack(cols, 1)
    with torch.no_grad():
        if name.endswith("svgp"):
            mu, v = model.marginal_variational_qf_parameters(X, diagonal=True, is_duvenaud=False, init_Z=None)
            out["mu"], out["v"] = mu.reshape(-1), v.reshape(-1)
        # evaluation path on the test split (sparse_MF_SP.py:637-825), as Trainer.compute_metrics calls it
        model.set_is_training(False)
        Y_std = torch.tensor(np.asarray(dc["Y_std"], dtype=np.float64).reshape(-1))
        logp, (m1, m2) = model.test_log_likelihood(dc["X_te"], dc["Y_te"], return_moments=True, Y_std=Y_std, S_MC_NNet=None)
        out["test_logp_sum"], out["pred_m1"], out["pred_m2"] = logp.reshape(-1), m1.reshape(-1), m2.reshape(-1)
        model.set_is_training(True)
        if is_id:
            model.eval()
    # Adam steps continue from the same state (gradients are recomputed; trainer_base.py:337-342)
    if is_id:
        nn_ids = {id(q) for q in nn_params(model)}
        groups = [{"params": [q for q in model.parameters() if id(q) not in nn_ids], "lr": 0.01},
                  {"params": nn_params(model), "lr": 0.01, "weight_decay": 1e-5}]       # main.py:276-288
        opt = torch.optim.Adam(groups, lr=0.01)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=0.01)
    hist = []
    for _ in range(steps):
        e, l, k = model.ELBO(X, Y)
        opt.zero_grad()
        (-e).backward()
        opt.step()
        hist.append([e.item(), l.item(), k.item()])
    out["history"] = np.array(hist)
    out["final_Z"], out["final_m"] = model.Z.detach()[0], model.q_U.variational_mean.detach()[0]
    out["final_log_var_noise"] = model.likelihood.log_var_noise.detach().reshape(-1)
    if flow is not None and flow_scalar_params(model, prob["program"]):
        out["final_theta"] = torch.stack([q.detach().reshape(()) for q in flow_scalar_params(model, prob["program"])])
    if is_id:
        out["final_nn_W_head"] = torch.cat([q.detach().reshape(-1) for q in nn_params(model)])[:512]
    save(name, out)


def real_data_fixtures():
    dc, Z = real_dataset_fixture("power")
    # known answers at initialisation (ELBO -81723.694286, KLD 525.646773 for SVGP and identity TGP)
    full_size_fixture(dc, "power_seed1", Z, None, "power_init_svgp", perturb=False)
    full_size_fixture(dc, "power_seed1", Z, "sal2", "power_init_sal2", perturb=False, steps=2)
    for flow, name in ((None, "power_svgp"), ("sal2", "power_sal2"), ("tanh3x2", "power_tanh3x2"), ("idsal3", "power_idsal3")):
        full_size_fixture(dc, "power_seed1", Z, flow, name)
    dcb, Zb = real_dataset_fixture("boston")
    full_size_fixture(dcb, "boston_seed1", Zb, None, "boston_init_svgp", perturb=False)
    full_size_fixture(dcb, "boston_seed1", Zb, None, "boston_svgp")
    # medium ID fixture (N=1024, M=100, D=4, S=32, ID-SAL x 3) on the first 1024 Power rows
    dcm = dict(dc)
    dcm["X_tr"], dcm["Y_tr"] = dc["X_tr"][:1024].clone(), dc["Y_tr"][:1024].clone()
    full_size_fixture(dcm, "power_seed1[:1024]", Z, "idsal3", "med_idsal3")


def main():
    if "--uci-only" in sys.argv:
        uci_loader_fixtures()
        return
    if "--real-only" in sys.argv:
        real_data_fixtures()
        return
    if "--matern-only" in sys.argv:
        matern_fixtures()
        return
    matern_fixtures()
    cases = [
        ("tiny_svgp", dict(N=64, D=3, M=8, S=8, flow=None)),
        ("tiny_sal2", dict(N=64, D=3, M=8, S=8, flow="sal2")),
        ("tiny_tanh3x2", dict(N=64, D=3, M=8, S=8, flow="tanh3x2")),
        ("tiny_idsal3", dict(N=64, D=3, M=8, S=8, flow="idsal3")),
        ("ragged_sal2", dict(N=77, D=5, M=19, S=12, flow="sal2")),      # N, M not multiples of 16
        ("boston_like_svgp", dict(N=455, D=13, M=5, S=20, flow=None)),  # C1 shape
        ("med_svgp", dict(N=1024, D=4, M=100, S=32, flow=None)),        # C2 shape, fewer rows
        ("med_sal2", dict(N=1024, D=4, M=100, S=32, flow="sal2")),      # C3 shape, fewer rows
        ("med_tanh3x2", dict(N=1024, D=4, M=100, S=32, flow="tanh3x2")),
    ]
    for name, kw in cases:
        prob = orc.synthetic_problem(kw["N"], kw["D"], kw["M"], seed=0, flow=kw["flow"], S=kw["S"])
        reference_step0(prob, kw["flow"], name)
    # identity-initialised TGP == SVGP known answer and KL-at-init (ii)
    prob = orc.synthetic_problem(256, 4, 100, seed=0, flow="sal2", S=32, perturb=False)
    reference_step0(prob, "sal2", "init_sal2_identity")
    prob = orc.synthetic_problem(256, 4, 100, seed=0, flow=None, S=32, perturb=False)
    reference_step0(prob, None, "init_svgp")
    # trainer sequence
    for name, flow in (("adam5_svgp", None), ("adam5_sal2", "sal2")):
        prob = orc.synthetic_problem(64, 3, 8, seed=0, flow=flow, S=8)
        reference_adam_steps(prob, flow, name)
    cholesky_ladder_fixture()
    real_data_fixtures()


if __name__ == "__main__":
    main()
