"""Drop-in counterpart of the reference's code/main.py (same four flags, same hyper-parameters and call
sequence), running on the MI355X kernels:

    python -m tgp.pytorch_amd.main --model TGP --dataset power --train_test_seed_split 1 --num_inducing 100

`--dataset synthetic_power|synthetic_boston` uses seeded data of the same shape (the CSVs are the
reference's data files; point $TGP_DATA_ROOT at them for the real runs).  `--epochs` shortens the 15000-epoch
recipe for smoke runs.
"""
import argparse

import numpy
import torch

from . import config as cg
from .data import return_dataset
from .flow import instance_flow
from .flows import SAL, StepTanhL
from .initializers import find_forward_params, find_forward_params_input_dependent_flow
from .kernels import instance_kernel
from .likelihoods import GaussianLinearMean, GaussianNonLinearMean
from .models import sparse_MF_GP, sparse_MF_SP
from .trainers import Trainer_SP_regression
from .utils import KMEANS

# code/exp_config.py:4-86
HYPER = {
    ("ID_TGP", "boston"): dict(arch="SAL", blocks=1, steps=None, act="tanh", layers=1, DR=0.5, BN=0, H=25),
    ("ID_TGP", "power"): dict(arch="SAL", blocks=3, steps=None, act="relu", layers=2, DR=0.25, BN=0, H=50),
    ("TGP", "boston"): dict(arch="StepTanhL", blocks=10, steps=2),
    ("TGP", "power"): dict(arch="SAL", blocks=2, steps=None),
}


def main(argv=None):
    ap = argparse.ArgumentParser(description="TGP on MI355X")
    ap.add_argument("--model", required=True, help="ID_TGP, TGP or SVGP")
    ap.add_argument("--dataset", required=True, choices=["boston", "power", "synthetic_boston", "synthetic_power"])
    ap.add_argument("--train_test_seed_split", required=True, type=int)
    ap.add_argument("--num_inducing", required=True, type=int)
    ap.add_argument("--epochs", type=int, default=15000)
    args = ap.parse_args(argv)
    base = args.dataset.replace("synthetic_", "")

    cg.device = "cuda:0"
    cg.set_maximum_precission()
    loaders, dc = return_dataset(args.dataset, 10000, use_validation=None, seed=args.train_test_seed_split,
                                 options={"shuffle_train": True})
    Dx, Dy = dc["Dx"], dc["Dy"]
    init_Z = KMEANS(dc["X_tr"], args.num_inducing, n_init=10, seed=cg.config_seed)

    flow_specs = None
    if args.model != "SVGP":
        hp = HYPER[(args.model, base)]
        rest = {"input_dependent": args.model == "ID_TGP", "input_dim": Dx, "num_hidden_layers": hp.get("layers"),
                "batch_norm": hp.get("BN"), "dropout": hp.get("DR"), "hidden_dim": hp.get("H"),
                "hidden_activation": hp.get("act"), "inference": "MC_dropout"}
        rest = {k: v for k, v in rest.items() if v is not None}
        if hp["arch"] == "SAL":
            flow_specs = SAL(hp["blocks"], **rest)
        else:
            def random_flow_fn():
                return instance_flow(StepTanhL(hp["blocks"], hp["steps"], add_f0=True))
            Ytr = dc["Y_tr"]
            x_in = numpy.linspace(float(Ytr.min()) - 1, float(Ytr.max()) + 1, 5000)
            flow_specs, mse = find_forward_params(x_in, x_in.copy(), random_flow_fn, num_restarts=1, num_epochs=2000)
            if numpy.any(numpy.isnan(numpy.array(mse))):
                raise RuntimeError("Got MSE loss to Nan on the flow initializer.")
        if args.model == "ID_TGP":
            T_flow = instance_flow(flow_specs) if isinstance(flow_specs, list) else flow_specs
            flow_specs, _ = find_forward_params_input_dependent_flow(loaders[0], FLOW=T_flow, num_epochs=2000, noise_var=0.0)

    if args.model == "SVGP":
        lik = GaussianLinearMean(out_dim=Dy, noise_init=0.05, noise_is_shared=False)
    else:
        lik = GaussianNonLinearMean(out_dim=Dy, noise_init=0.05, noise_is_shared=False, quadrature_points=cg.quad_points)
    K = instance_kernel("scale_rbf", ard_num_dim=Dx, num_multioutput=Dy, kernel_is_shared=False,
                        init_params={"length_scale": 2.0, "kernel_scale": 2.0, "noisy_variance": 1e-6})
    ip = {"variational_distribution": {"variance_scale": 1e-5, "mean_scale": 0.0}}
    common = dict(model_specs=["zero", K], X=dc["X_tr"], init_Z=init_Z, N=dc["N_tr"], likelihood=lik, num_outputs=Dy,
                  is_whiten=True, K_is_shared=False, mean_is_shared=False, Z_is_shared=False, q_U_is_shared=False,
                  add_noise_inducing=0.0, init_params=ip)
    if args.model == "SVGP":
        model = sparse_MF_GP(**common)
    else:
        model = sparse_MF_SP(flow_specs=[flow_specs], flow_connection="single", be_fully_bayesian=False, **common)
    model.to(cg.device)

    lr = 0.01
    specs = [[]]
    if args.model == "ID_TGP":
        sched = [[lr, n] for n, _ in model.named_parameters() if "G_matrix" in n and "NNets" not in n]
        sched.append([lr, 1e-5, "NNets"])
        specs[0].extend(sched)
    Y_std = (torch.ones((Dy,)) * dc["Y_std"]).to(cg.device)
    trainer = Trainer_SP_regression(model=model, data_loaders=loaders, validate_each=max(args.epochs // 10, 1), plot=False,
                                    track=False, Y_std=Y_std, plot_each=-1, S_test=100, inference_in_cpu=True)
    trainer.train(epochs=args.epochs, lr_ALL=lr, opt="adam", keep_parameter_groups=True,
                  optimisation_schedule=([1.0], specs), lr_groups=None)
    res = trainer.compute_metrics()
    if args.model == "ID_TGP":       # the reference's result lines (main.py:309-324)
        print("Dataset {}, num inducing points {}, POINT ESTIMATE FLOW , Test Negative LOGL {:.3f}, Test RMSE {:.3f}".format(
            args.dataset, args.num_inducing, -res[6], res[7]))
    else:
        print("Dataset {}, num inducing points {}, model {}, Test Negative LOGL {:.3f}, Test RMSE {:.3f}".format(
            args.dataset, args.num_inducing, args.model, -res[6], res[7]))
    if args.model == "ID_TGP":
        model.be_fully_bayesian(True)
        res = trainer.compute_metrics()
        print("Dataset {}, num inducing points {}, BAYESIAN FLOW , Test Negative LOGL {:.3f}, Test RMSE {:.3f}".format(
            args.dataset, args.num_inducing, -res[6], res[7]))
    return res


if __name__ == "__main__":
    main()
