"""Flow initialisers with the reference's signatures (code/dsp/initializers/initializers.py:29-182).

* `find_forward_params`: fits a randomly initialised flow to a target map (main.py: the identity on the range of
  Y) by Adam on the mean squared error.  The MSE and its gradient w.r.t. the flow parameters come from the HIP
  quadrature-likelihood kernel run with ONE node (x_0 = 0, w_0/sqrt(pi) = 1, unit noise):
      ELL = sum_n [-1/2 log 2pi - 1/2 (y_n - G(x_n))^2]   =>   MSE = -2 (ELL/N + 1/2 log 2pi),
  so no second flow implementation exists for the initialiser.
* `find_forward_params_input_dependent_flow`: regresses the per-row MLPs onto the scalar flow parameters, then
  switches the scalars off.  For the run main.py sets up (default Adam, no input noise, one resident batch, nets the
  HIP MLP kernels cover) the whole epoch -- tgp_mlp_forward_f64 (dropout on) -> d loss / d outputs ->
  tgp_mlp_backward_f64 -> tgp_adam_dev_f64 -- is captured in one HIP graph and replayed `num_epochs` times with no host
  synchronisation (the reference: 6 ATen MLPs + loss.item() per batch, 2000 epochs before training starts,
  main.py:194-208); anything else keeps the reference's PyTorch loop.
"""
import warnings

import numpy
import torch
from .engine import CAPTURE_MODE
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


def id_nets_and_targets(FLOW):
    """[NNets_a, NNets_b, ...] of the input-dependent blocks in column order and the scalars they are regressed onto
    (Sinh_ArcsinhFlow.forward_initializer, models/flow.py:907-918)."""
    nets, targets = [], []
    for fl in FLOW.flow_arr:
        if getattr(fl, "input_dependent", False) and hasattr(fl, "NNets_a") and not fl.parameters_are_turn_off:
            nets += [fl.NNets_a, fl.NNets_b]
            targets += [fl.a.detach().reshape(()), fl.b.detach().reshape(())]
    return nets, targets


class IdInitEngine:
    """The identity initialiser of the input-dependent flows on the HIP MLP kernels:
        loss = sum_k mean_n (NN_k(x_n) - target_k)^2     (k over the 2 B nets; dropout active, FLOW.train())
    one Adam(lr) step per epoch on the packed weights, the epoch captured in a HIP graph."""

    def __init__(self, X, nets, targets, spec, lr=0.01):
        from . import lib as L
        self.L, self.lib = L, L.load()
        self.spec, self.nets = spec, nets
        dev = X.device
        self.X = X.contiguous()
        self.N = X.shape[0]
        self.W = torch.cat([p.detach().reshape(-1) for net in nets for p in net.parameters()]).to(dev, torch.float64).contiguous()
        self.gW = torch.zeros_like(self.W)
        self.m, self.v = torch.zeros_like(self.W), torch.zeros_like(self.W)
        self.tgt = torch.stack(targets).to(dev, torch.float64).reshape(1, -1)
        self.out = torch.zeros(self.N, spec.nnets, dtype=torch.float64, device=dev)
        self.g_out = torch.zeros_like(self.out)
        self.loss = torch.zeros((), dtype=torch.float64, device=dev)
        self.step_dev = torch.zeros(2, dtype=torch.int32, device=dev)
        self.d = spec.struct(self.N, True)
        self.ws = torch.empty(self.lib.tgp_mlp_workspace_bytes(self.d) // 8 + 16, dtype=torch.float64, device=dev)
        self.lr = float(lr)
        self.graph = None

    def epoch(self):
        L, lib, sp = self.L, self.lib, L_stream()
        L.check(lib.tgp_mlp_forward_f64(self.d, L.ptr(self.X), L.ptr(self.W), L.ptr(self.step_dev), L.ptr(self.out), sp),
                "tgp_mlp_forward_f64")
        torch.sub(self.out, self.tgt, out=self.g_out)
        self.loss.copy_((self.g_out * self.g_out).sum())         # sum over rows and nets; / N below
        self.g_out.mul_(2.0 / self.N)
        L.check(lib.tgp_mlp_backward_f64(self.d, L.ptr(self.X), L.ptr(self.W), L.ptr(self.step_dev), L.ptr(self.g_out),
                                         L.ptr(self.gW), L.ptr(self.ws), self.ws.numel() * 8, sp), "tgp_mlp_backward_f64")
        L.check(lib.tgp_adam_dev_f64(L.ptr(self.W), L.ptr(self.gW), L.ptr(self.m), L.ptr(self.v), self.W.numel(), self.lr,
                                     0.9, 0.999, 1e-8, 0.0, L.ptr(self.step_dev), 0, sp), "tgp_adam_dev_f64")

    def run(self, num_epochs):
        if num_epochs < 1:
            return 0.0                                # the reference's loop body never runs: loss_acc stays 0.0
        self.epoch()                                  # eager once: kernel attributes, allocator warm-up
        torch.cuda.synchronize()
        if num_epochs > 1:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
                self.epoch()
            for _ in range(num_epochs - 1):
                self.graph.replay()
        return float(self.loss) / self.N                # the only synchronisation: the last epoch's loss

    def write_back(self):
        o = 0
        with torch.no_grad():
            for net in self.nets:
                for p in net.parameters():
                    p.copy_(self.W[o:o + p.numel()].reshape(p.shape).to(p.dtype))
                    o += p.numel()


def L_stream():
    from . import lib as L
    return L.stream_ptr()


def find_forward_params_input_dependent_flow(x_loader, FLOW, optimizer_fn=None, num_epochs=None, seed=0, verbose=0,
                                             verbose_level=0, noise_var=0.0):
    from .data import DeviceLoader
    from .flow import mlp_spec
    if (optimizer_fn is None and isinstance(noise_var, float) and noise_var == 0.0 and isinstance(x_loader, DeviceLoader)
            and len(x_loader) == 1 and x_loader.X.is_cuda and x_loader.X.dtype == torch.float64
            and getattr(cg, "use_step_engine", True)):
        nets, targets = id_nets_and_targets(FLOW)
        spec = mlp_spec(nets, seed=cg.config_seed) if nets else None
        if spec is not None:
            warnings.warn("Using default optimizer (optim.Adam(trainable_params, lr=0.01))", Warning)
            if num_epochs is None:
                warnings.warn("Using default number of epochs (100)", Warning)
            numpy.random.seed(seed)                   # as the reference does (initializers.py:126): later numpy draws agree
            FLOW.to(cg.device)
            eng = IdInitEngine(x_loader.X, nets, targets, spec, lr=0.01)
            loss_acc = eng.run(100 if num_epochs is None else num_epochs)
            eng.write_back()
            FLOW.turn_off_initializer_parameters()
            return FLOW, loss_acc
    if optimizer_fn is None:
        warnings.warn("Using default optimizer (optim.Adam(trainable_params, lr=0.01))", Warning)
        optimizer_fn = lambda trainable_params: optim.Adam(trainable_params, lr=0.01)   # noqa: E731
    if num_epochs is None:
        warnings.warn("Using default number of epochs (100)", Warning)
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
