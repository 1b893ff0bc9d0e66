"""Fully Bayesian ID_TGP evaluation of the Power test split (957 rows, S_MC = 100 dropout samples, 100 quadrature points):
Trainer-style test_log_likelihood(return_moments=True) -- the slowest thing in the reference (12.6 s on its CPU
path in the build container).  Here: ONE MLP launch over 100 x 957 rows + ONE tgp_predict_f64 launch per quantity."""
import os
import sys
import time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tgp.pytorch_amd import config as cg
cg.set_maximum_precission()
from tgp.pytorch_amd.synthetic import synthetic_problem
from tgp.pytorch_amd.kernels import instance_kernel
from tgp.pytorch_amd.flow import instance_flow
from tgp.pytorch_amd.flows import SAL
from tgp.pytorch_amd.likelihoods import GaussianNonLinearMean
from tgp.pytorch_amd.models import sparse_MF_SP
dev = torch.device("cuda:0")
cg.device = dev
prob = synthetic_problem(8611, 4, 100, seed=0, flow="sal2", S=32)
te = synthetic_problem(957, 4, 100, seed=1, flow="sal2", S=32)
X, Xte, Yte = prob["X"].to(dev), te["X"].to(dev), te["Y"].to(dev).reshape(-1, 1)
K = instance_kernel("scale_rbf", ard_num_dim=4, num_multioutput=1, kernel_is_shared=False,
                    init_params={"length_scale": 2.0, "kernel_scale": 2.0}).to(dev)
lik = GaussianNonLinearMean(out_dim=1, noise_init=0.05, noise_is_shared=False, quadrature_points=cg.quad_points)
fl = instance_flow(SAL(3, input_dependent=True, input_dim=4, num_hidden_layers=2, batch_norm=0, dropout=0.25, hidden_dim=50,
                       hidden_activation="relu", inference="MC_dropout"))
fl.turn_off_initializer_parameters()
model = sparse_MF_SP(["zero", K], X, prob["params"]["Z"].clone().to(dev), 8611, lik, 1, True, False, False, False, False,
                     [fl], "single", 0.0, init_params={"variational_distribution": {"variance_scale": 1e-5, "mean_scale": 0.0}}).to(dev)
model.set_is_training(False)
Ystd = torch.ones(1, dtype=torch.float64, device=dev)
for bayes, S in ((False, None), (True, 100)):
    model.be_fully_bayesian(bayes)
    for _ in range(3):
        lp, mom = model.test_log_likelihood(Xte, Yte, True, Ystd, S_MC_NNet=S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        lp, mom = model.test_log_likelihood(Xte, Yte, True, Ystd, S_MC_NNet=S)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("ID_TGP test_log_likelihood(return_moments=True), 957 rows, %d quadrature points, %s: %.2f ms per call  (logp %.4f)"
          % (cg.quad_points, "fully Bayesian S_MC=100" if bayes else "point estimate", dt * 1e3, float(lp)), flush=True)
print("reference CPU path, same evaluation, build container (SURVEY N1): 12.6 s")
