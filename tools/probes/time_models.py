import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tgp.pytorch_amd import config as cg
cg.set_maximum_precission()
from oracle import tgp_oracle as orc
from tgp.pytorch_amd.kernels import instance_kernel
from tgp.pytorch_amd.flow import instance_flow
from tgp.pytorch_amd.flows import SAL, StepTanhL
from tgp.pytorch_amd.likelihoods import GaussianNonLinearMean, GaussianLinearMean
from tgp.pytorch_amd.models import sparse_MF_SP, sparse_MF_GP
dev = torch.device("cuda:0")
prob = orc.synthetic_problem(8611, 4, 100, seed=0, flow="sal2", S=32)
X, Y = prob["X"].to(dev), prob["Y"].to(dev)
IP = {"variational_distribution": {"variance_scale": 1e-5, "mean_scale": 0.0}}
for name in ("TGP", "ID_TGP"):
    K = instance_kernel("scale_rbf", ard_num_dim=4, num_multioutput=1, kernel_is_shared=False,
                        init_params={"length_scale": 2.0, "kernel_scale": 2.0}).to(dev)
    lik = GaussianNonLinearMean(out_dim=1, noise_init=0.05, noise_is_shared=False, quadrature_points=32)
    if name == "TGP":
        specs = instance_flow(SAL(2))
    else:
        specs = SAL(3, input_dependent=True, input_dim=4, num_hidden_layers=2, batch_norm=0, dropout=0.25, hidden_dim=50,
                    hidden_activation="relu", inference="MC_dropout")
        specs = instance_flow(specs)
        specs.turn_off_initializer_parameters()
    model = sparse_MF_SP(["zero", K], X, prob["params"]["Z"].clone().to(dev), 8611, lik, 1, True, False, False, False, False,
                         [specs], "single", 0.0, init_params=IP).to(dev)
    model.set_is_training(True)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    def step():
        opt.zero_grad()
        elbo, ell, kld = model.ELBO(X, Y)
        (-elbo).backward()
        opt.step()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%s model-class path (autograd wrapper + torch Adam): %.3f ms/step = %.0f steps/s" % (name, dt * 1e3, 1 / dt))
