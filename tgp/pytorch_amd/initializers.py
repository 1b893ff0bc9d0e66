"""Flow initialisers with the reference's signatures (code/dsp/initializers/initializers.py:29-182).

* `find_forward_params`: fits a randomly initialised flow to a target map (main.py: the identity on the range of
  Y) by Adam on the mean squared error.  The MSE and its gradient w.r.t. the flow parameters come from the HIP
  quadrature-likelihood kernel run with ONE node (x_0 = 0, w_0/sqrt(pi) = 1, unit noise):
      ELL = sum_n [-1/2 log 2pi - 1/2 (y_n - G(x_n))^2]   =>   MSE = -2 (ELL/N + 1/2 log 2pi),
  so no second flow implementation exists for the initialiser.
* `find_forward_params_input_dependent_flow`: regresses the per-row MLPs onto the scalar flow parameters
  (plain PyTorch modules on the GPU, as in the reference), then switches the scalars off.
"""
import warnings

import numpy
import torch
from torch import optim

from . import config as cg
from . import ops
from .flow import compile_flow


def flow_mse_and_grads(flow, x, y):
    """(MSE, [d MSE / d theta_i]) of G(x) vs y on the GPU; x, y: (N,) float64 device tensors."""
    spec, theta_list, nets = compile_flow(flow)
    assert not nets, "the identity initialiser works on flows with shared parameters"
    dev = x.device
    theta = torch.stack([p.detach().reshape(()) for p in theta_list]).to(dev)
    lvn = torch.zeros(1, dtype=torch.float64, device=dev)
    res = ops.ell_flow(y, x, torch.ones_like(x), lvn, spec, theta, 1, None, 1.0)
    n = x.numel()
    mse = -2.0 * (res["ell"] / n + 0.5 * ops_log2pi())
    return mse, [(-2.0 / n) * g for g in res["g_theta"]], theta_list


def ops_log2pi():
    import math
    return math.log(2.0 * float(numpy.float32(math.pi)))     # the kernels use the reference's float32-rounded pi


def find_forward_params(x_input, y_ouput, random_flow_fn=None, num_restarts=1, optimizer_fn=None, num_epochs=None,
                        seed=0, verbose=0, verbose_level=0):
    if random_flow_fn is None:
        raise RuntimeError("random_flow_fn must be specified")
    if optimizer_fn is None:
        warnings.warn("Using default optimizer (optim.Adam(trainable_params, lr=0.01))", Warning)
        optimizer_fn = lambda trainable_params: optim.Adam(trainable_params, lr=0.01)   # noqa: E731
    if num_epochs is None:
        warnings.warn("Using default number of epochs (100)", Warning)
        num_epochs = 100
    numpy.random.seed(seed)
    dev = torch.device(cg.device)
    x = torch.as_tensor(numpy.asarray(x_input), dtype=torch.float64).reshape(-1).to(dev)
    y = torch.as_tensor(numpy.asarray(y_ouput), dtype=torch.float64).reshape(-1).to(dev)
    flows, finals, curves = [], [], []
    for r in range(num_restarts):
        flow = random_flow_fn()
        params = [p for _, p in flow.named_parameters()]
        optimizer = optimizer_fn(params)
        curve = []
        for e in range(num_epochs):
            optimizer.zero_grad()
            mse, grads, theta_list = flow_mse_and_grads(flow, x, y)
            for p, g in zip(theta_list, grads):
                p.grad = g.reshape(p.shape).to(p.device, p.dtype)
            optimizer.step()
            optimizer.zero_grad()
            curve.append(float(mse))
            if verbose and verbose_level != 1:
                print("Restart {} Step {} - {}".format(r, e, curve[-1]))
        flows.append(flow)
        finals.append(curve[-1])
        curves.append(curve)
    ok = [i for i, v in enumerate(finals) if not numpy.isnan(v)]
    best = min(ok, key=lambda i: finals[i])
    return flows[best], curves[best]


def find_forward_params_input_dependent_flow(x_loader, FLOW, optimizer_fn=None, num_epochs=None, seed=0, verbose=0,
                                             verbose_level=0, noise_var=0.0):
    if optimizer_fn is None:
        optimizer_fn = lambda trainable_params: optim.Adam(trainable_params, lr=0.01)   # noqa: E731
    if num_epochs is None:
        num_epochs = 100
    numpy.random.seed(seed)
    params = [p for _, p in FLOW.named_parameters()]
    optimizer = optimizer_fn(params)
    state = FLOW.training
    FLOW.train()
    FLOW.to(cg.device)
    loss_acc = 0.0
    for e in range(num_epochs):
        loss_acc = 0.0
        for x, y in x_loader:
            x = x.to(cg.device)
            if isinstance(noise_var, float):
                if noise_var > 0.0:
                    x = x + torch.zeros_like(x).normal_() * numpy.sqrt(noise_var)
            elif isinstance(noise_var, list):
                x = x + torch.zeros_like(x).normal_() * numpy.sqrt(noise_var[numpy.random.randint(len(noise_var))])
            else:
                raise NotImplementedError()
            optimizer.zero_grad()
            loss = FLOW.forward_initializer(x)
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            loss_acc += loss.item()
    FLOW.train(state)
    FLOW.turn_off_initializer_parameters()
    return FLOW, loss_acc
