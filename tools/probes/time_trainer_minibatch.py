"""Minibatch training THROUGH Trainer_SP_regression.train (the Airline recipe: batch_size 10000, M = 1000, StepTanhL 5 x 6;
code/main.py:74, bash_scripts/launch_test_uci_large_regression.sh:27-33) on a resident synthetic data set: steps per second
of the resident minibatch engine against bench.py's fixed-batch `tgp_airline_mb10k` figure."""
import os, sys, time
import numpy as np
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from tgp.pytorch_amd import config as cg
cg.set_maximum_precission()
cg.device = "cuda:0"
from tgp.pytorch_amd import synthetic
from tgp.pytorch_amd.data import DeviceLoader
from tgp.pytorch_amd.flow import instance_flow, compile_flow
from tgp.pytorch_amd.flows import StepTanhL
from tgp.pytorch_amd.kernels import instance_kernel
from tgp.pytorch_amd.likelihoods import GaussianNonLinearMean
from tgp.pytorch_amd.models import sparse_MF_SP
from tgp.pytorch_amd.trainers import Trainer_SP_regression

N, D, M, B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 8, 1000, 10000
prob = synthetic.synthetic_problem(N, D, M, seed=0, flow="tanh5x6", S=32)
K = instance_kernel("scale_rbf", ard_num_dim=D, num_multioutput=1, kernel_is_shared=False,
                    init_params={"length_scale": 2.0, "kernel_scale": 2.0, "noisy_variance": 1e-6})
np.random.seed(0)
flow = instance_flow(StepTanhL(5, 6, add_f0=True))
model = sparse_MF_SP(["zero", K], prob["X"], prob["params"]["Z"].clone(), N, GaussianNonLinearMean(1, 0.05, False, 32), 1, True,
                     False, False, False, False, [flow], "single", 0.0,
                     init_params={"variational_distribution": {"variance_scale": 1e-5, "mean_scale": 0.0}}).to("cuda:0")
with torch.no_grad():
    for prm, val in zip(compile_flow(model.G_matrix[0])[1], prob["params"]["theta"]):
        prm.data = val.clone().reshape(()).to("cuda:0")
loader = DeviceLoader(prob["X"], prob["Y"], B, shuffle=True, device="cuda:0", seed=0)
for resident in (True, False):
    cg.use_step_engine = resident
    tr = Trainer_SP_regression(model, [loader], 1e20, False, False, torch.ones(1, device="cuda:0"), -1, 100, True)
    tr.train(epochs=1, lr_ALL=0.01, opt="adam", keep_parameter_groups=True)       # warm-up + capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ep = 3 if resident else 1
    tr.train(epochs=ep, lr_ALL=0.01, opt="adam", keep_parameter_groups=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = ep * len(loader)
    print("Trainer.train minibatch N=%d B=%d M=%d tanh5x6 %s: %d steps in %.2f s = %.1f steps/s (%.2f ms/step), last ELBO %.1f"
          % (N, B, M, "RESIDENT (MinibatchEngine)" if resident else "eager loop", steps, dt, steps / dt, 1e3 * dt / steps, -tr.loss_arr[-1]))
