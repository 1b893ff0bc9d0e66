"""Flow modules with the reference's class names, constructor arguments and parameter names
(code/dsp/models/flow.py), so `named_parameters()` reads `G_matrix.0.flow_arr.{k}.a` ... and
`...NNets_a...` exactly like the reference (code/main.py:281-286 selects on those substrings).

The modules are parameter holders + a compiler to the flow *program* the HIP kernels execute
(`compile_flow`); `forward()` evaluates the flow on the GPU through tgp_flow_eval_f64 (no autograd: the
training path differentiates inside the fused kernel, ops.ElboFunction).  Only the flows main.py can
reach are provided: affine, sinh_arcsinh, tanh (inside step_flow), step_flow, identity.
"""
import torch
import torch.nn as nn

from . import config as cg
from . import lib as L
from . import ops


def instance_flow(flow_list, is_composite=True):
    """code/dsp/models/flow.py:39-85."""
    FL = []
    for name, init_values in flow_list:
        if name == "affine":
            fl = AffineFlow(**init_values)
        elif name == "sinh_arcsinh":
            fl = Sinh_ArcsinhFlow(**init_values)
        elif name == "identity":
            fl = IdentityFlow()
        elif name == "tanh":
            fl = TanhFlow(**init_values)
        elif name == "step_flow":
            fl = StepFlow(**init_values)
        else:
            raise ValueError("Unkown flow identifier {} (this build provides affine, sinh_arcsinh, tanh, step_flow, "
                             "identity: the flows reachable from main.py)".format(name))
        FL.append(fl)
    return CompositeFlow(FL) if is_composite else FL


class apply_linear(nn.Module):
    """One MLP layer of the input-dependent flows.  Stand-in for jmaronas/pytorch_library `apply_linear`
    (absent; layer order unpinned): Linear -> activation -> Dropout(p)."""

    def __init__(self, inp, out, act, shape=None, std=0.0, drop=0.0, bn=0):
        super().__init__()
        assert bn == 0 and std == 0.0, "batch-norm / noisy layers are not used by any reference configuration"
        self.linear = nn.Linear(inp, out)
        self.act = {"relu": nn.ReLU(), "tanh": nn.Tanh(), "linear": nn.Identity(), "sigmoid": nn.Sigmoid()}[act]
        self.drop = nn.Dropout(drop) if drop > 0 else None

    def forward(self, x):
        x = self.act(self.linear(x))
        return self.drop(x) if self.drop is not None else x


def _mlp(input_dim, cfg):
    H = cfg.get("hidden_dim", input_dim)
    act = cfg.get("hidden_activation", "relu")
    num_H = cfg.get("num_hidden_layers", 1)
    DR = cfg.get("dropout", 0.0)
    BN = cfg.get("batch_norm", 0)
    if cfg.get("inference", "MC_dropout") != "MC_dropout":
        raise NotImplementedError("only MC_dropout inference is provided (the one exp_config.py uses)")
    layers, inp = [], input_dim
    for _ in range(num_H):
        layers.append(apply_linear(inp, H, act, drop=DR, bn=BN))
        inp = H
    layers.append(apply_linear(H, 1, "linear"))
    return nn.Sequential(*layers)


class Flow(nn.Module):
    input_dependent = False

    def KLD(self):
        return 0.0

    def forward_initializer(self, X):
        return 0.0

    def turn_off_initializer_parameters(self):
        return None

    def forward(self, f0, X=None):
        return CompositeFlow([self]).forward(f0, X)


class IdentityFlow(Flow):
    def forward(self, f0, X=None):
        return f0

    def inverse(self, f):
        return f


class AffineFlow(Flow):
    """fk = a*f0 + b (flow.py:310-361)."""

    def __init__(self, init_a, init_b, set_restrictions, input_dependent=False, input_dim=-1,
                 input_dependent_config={}):
        super().__init__()
        if input_dependent:
            raise NotImplementedError("input dependent affine flows raise NotImplementedError in the reference too")
        self.a = nn.Parameter(torch.tensor(init_a, dtype=cg.dtype))
        self.b = nn.Parameter(torch.tensor(init_b, dtype=cg.dtype))
        self.set_restrictions = set_restrictions
        self.input_dependent = False


class TanhFlow(Flow):
    """fk = a + b*tanh((f0-c)/d) [+ f0] (flow.py:619-815); only used inside StepFlow, shared parameters."""

    def __init__(self, init_a, init_b, init_c, init_d, add_init_f0, set_restrictions, input_dependent=False,
                 input_dim=-1, input_dependent_config={}):
        super().__init__()
        if input_dependent:
            raise NotImplementedError("input dependent tanh/step flows are not reachable from main.py")
        for n, v in (("a", init_a), ("b", init_b), ("c", init_c), ("d", init_d)):
            setattr(self, n, nn.Parameter(torch.tensor(v, dtype=cg.dtype)))
        self.set_restrictions = True if add_init_f0 else set_restrictions
        self.add_init_f0 = add_init_f0
        self.input_dependent = False


class Sinh_ArcsinhFlow(Flow):
    """fk = sinh(b*asinh(f0) - a) [+ f0] (flow.py:817-996); input dependent variant: a, b = MLPs of x."""

    def __init__(self, init_a, init_b, add_init_f0, set_restrictions, input_dependent=False, input_dim=-1,
                 input_dependent_config={}):
        super().__init__()
        self.a = nn.Parameter(torch.tensor(init_a, dtype=cg.dtype))
        self.b = nn.Parameter(torch.tensor(init_b, dtype=cg.dtype))
        if input_dependent:
            assert input_dim > 0, "Set input dimension if input_dependent = True"
            self.NNets_a = _mlp(input_dim, input_dependent_config)
            self.NNets_b = _mlp(input_dim, input_dependent_config)
            self.inference = input_dependent_config.get("inference", "MC_dropout")
            self.parameters_are_turn_off = False
        self.set_restrictions = True if add_init_f0 else set_restrictions
        self.add_init_f0 = add_init_f0
        self.input_dependent = input_dependent

    def forward_initializer(self, X):
        if not self.input_dependent:
            return 0.0
        a, b = self.NNets_a(X), self.NNets_b(X)
        return ((a - self.a.detach()) ** 2).mean() + ((b - self.b.detach()) ** 2).mean()

    def turn_off_initializer_parameters(self):
        """flow.py:920-934: the scalar a, b only served the initialiser; drop them from the trainable set."""
        if self.input_dependent and not self.parameters_are_turn_off:
            self.a_untracked = self.a.data.detach()
            self.b_untracked = self.b.data.detach()
            self.a = None
            self.b = None
            self.parameters_are_turn_off = True


class StepFlow(Flow):
    """fk = [f0 +] sum_i flow_i(f0) (flow.py:1039-1149); tanh steps need no switch-off scale (a=1, b=0)."""

    def __init__(self, flow_arr, add_init_f0):
        super().__init__()
        assert isinstance(add_init_f0, bool)
        self.add_init_f0 = add_init_f0
        if isinstance(flow_arr[0], (list, tuple)):
            for name, prm in flow_arr:
                assert name == "tanh", "this build provides tanh step flows (StepTanhL) only"
                assert prm["set_restrictions"], "set_restrictions must be True. Got false for flow {}".format(name)
            self.flow_arr = nn.ModuleList(instance_flow(flow_arr, is_composite=False))
        else:
            self.flow_arr = nn.ModuleList(flow_arr)

    @property
    def input_dependent(self):
        return False

    @input_dependent.setter
    def input_dependent(self, value):
        pass


class CompositeFlow(Flow):
    """flow.py:146-191."""

    def __init__(self, flow_arr):
        super().__init__()
        self.flow_arr = nn.ModuleList(flow_arr)

    def forward(self, f, X=None):
        """G(f) on the GPU (no autograd).  f: (..., N); X: (N, Dx) for input dependent flows."""
        spec, theta_list, nets = compile_flow(self)
        if spec.nblk == 0:
            return f
        dev = f.device
        theta = torch.stack([p.detach().reshape(()) for p in theta_list]).to(dev) if theta_list else None
        rowp = nets_rowp(nets, X.reshape(-1, X.shape[-1])) if nets else None     # HIP MLP kernel, one launch for all nets
        flat = f.detach().reshape(-1, f.shape[-1]).contiguous()
        return ops.flow_eval(flat, spec, theta, rowp, want=("G",))["G"].reshape(f.shape)

    def forward_initializer(self, X):
        loss = 0.0
        for flow in self.flow_arr:
            loss = loss + flow.forward_initializer(X)
        return loss

    def KLD(self):
        return 0.0

    @property
    def input_dependent(self):
        return None

    @input_dependent.setter
    def input_dependent(self, value):
        for flow in self.flow_arr:
            if isinstance(flow, Sinh_ArcsinhFlow) and hasattr(flow, "NNets_a"):
                flow.input_dependent = value

    def turn_off_initializer_parameters(self):
        for flow in self.flow_arr:
            flow.turn_off_initializer_parameters()


def mlp_spec(nets, seed=0):
    """ops.MlpSpec of a list of per-row parameter networks if they share one architecture the HIP kernels cover
    (Sequential of apply_linear: D -> H x L -> 1, relu/tanh, dropout), else None (the caller evaluates them in PyTorch)."""
    try:
        first = nets[0]
        layers = list(first)
        D, H, Lh = layers[0].linear.in_features, layers[0].linear.out_features, len(layers) - 1
        act = {nn.ReLU: "relu", nn.Tanh: "tanh"}[type(layers[0].act)]
        p = layers[0].drop.p if layers[0].drop is not None else 0.0
        if not (1 <= Lh <= 3 and H <= 64):
            return None
        for net in nets:
            ls = list(net)
            if len(ls) != Lh + 1 or ls[-1].linear.out_features != 1 or not isinstance(ls[-1].act, nn.Identity):
                return None
            nin = D
            for l in ls[:-1]:
                if (l.linear.in_features, l.linear.out_features) != (nin, H) or type(l.act) is not type(layers[0].act):
                    return None
                if (l.drop.p if l.drop is not None else 0.0) != p:
                    return None
                nin = H
            if ls[-1].linear.in_features != H:
                return None
        spec = ops.MlpSpec(D, H, Lh, len(nets), act=act, drop_p=p, seed=seed)
        if spec.lds_bytes() > 158 * 1024:      # padded weights + activation strips must fit one CU's LDS (tgp_mlp.hip)
            return None
        return spec
    except (KeyError, AttributeError, IndexError, TypeError):
        return None


_mask_step = {}     # device -> int32[2] counter behind the dropout masks drawn outside the training step


def nets_rowp(nets, X2d, samples=1, spec=None, with_grad=False):
    """Per-row flow parameters (rows, len(nets)) of an input-dependent flow through tgp_mlp_forward_f64 -- the ONE MLP
    implementation of this package (training, evaluation, moments, sampling).  `with_grad`: the result carries autograd
    to the networks' weights (ops.MlpFunction: forward AND backward on the HIP MLP kernels) -- same checks, same mask
    counter, same salt as the plain call (ONE copy of the mask-stream logic).  Dropout follows the nets' Dropout LAYERS (train mode in training and in the
    fully Bayesian evaluation, where enable_eval_dropout() re-enables only them, models/utils_models.py:358-364).
    `samples` > 1 evaluates the rows `samples` times in the same launch -- what the reference does by expanding X to
    (S_MC, N, Dx) (models/sparse_MF_SP.py:753-758): output row s * N + n, every (sample, row) with its own mask.
    Architectures the kernel does not cover (H > 64, L > 3, mixed nets, batch norm) raise: nothing falls back to torch.nn."""
    spec = mlp_spec(nets, seed=cg.config_seed) if spec is None else spec
    if spec is None:
        raise L.TgpError("input-dependent flow networks outside the HIP MLP kernel's coverage (one architecture "
                         "D -> H x L -> 1 with H <= 64, 1 <= L <= 3, relu/tanh, dropout): no torch.nn fallback in this package")
    X2d = X2d.detach()
    if not X2d.is_cuda or X2d.dtype != torch.float64:
        raise L.TgpError("input-dependent flows run on the GPU in float64 (got %s %s)" % (X2d.device, X2d.dtype))
    drop_on = any(mod.training for mod in nets[0].modules() if "Dropout" in type(mod).__name__)
    step = _mask_step.get(str(X2d.device))
    if step is None:
        step = _mask_step[str(X2d.device)] = torch.zeros(2, dtype=torch.int32, device=X2d.device)
    if drop_on:
        step[0] += 1                                    # a fresh mask per call
    Xs = X2d.contiguous() if samples == 1 else X2d.repeat(samples, 1)
    # (this call site's own mask stream: its counter starts at 0 like the training step's and the evaluation's)
    if with_grad and torch.is_grad_enabled():
        W = torch.cat([p.reshape(-1) for net in nets for p in net.parameters()])
        # (the backward recomputes the masks of THIS call: it gets its own copy of the counter's value)
        return ops.MlpFunction.apply(Xs, W, spec.salted(ops.MASK_SALT_NETS), bool(drop_on), step.clone())
    W = torch.cat([p.detach().reshape(-1) for net in nets for p in net.parameters()])
    return ops.mlp_forward(spec.salted(ops.MASK_SALT_NETS), Xs, W, bool(drop_on), step)


def compile_flow(flow):
    """CompositeFlow -> (ops.FlowSpec, [shared scalar nn.Parameters in theta order], [per-row MLPs in column order])."""
    blocks, theta, nets = [], [], []
    for fl in flow.flow_arr:
        if isinstance(fl, IdentityFlow):
            continue
        if isinstance(fl, AffineFlow):
            blocks.append((L.FLOW_AFFINE, 0, len(theta), L.FLAG_RESTRICT if fl.set_restrictions else 0))
            theta += [fl.a, fl.b]
        elif isinstance(fl, Sinh_ArcsinhFlow):
            flags = (L.FLAG_RESTRICT if fl.set_restrictions else 0) | (L.FLAG_ADD_F0 if fl.add_init_f0 else 0)
            if fl.input_dependent:
                assert fl.parameters_are_turn_off, ("Call the method turn_off_initializer_parameters before using the "
                                                    "flow in an optimization loop.")
                blocks.append((L.FLOW_SAL, 0, len(nets), flags | L.FLAG_PER_ROW))
                nets += [fl.NNets_a, fl.NNets_b]
            else:
                blocks.append((L.FLOW_SAL, 0, len(theta), flags))
                theta += [fl.a, fl.b]
        elif isinstance(fl, StepFlow):
            K = len(fl.flow_arr)
            blocks.append((L.FLOW_STEPTANH, K, len(theta), L.FLAG_ADD_F0 if fl.add_init_f0 else 0))
            for t in fl.flow_arr:
                assert isinstance(t, TanhFlow) and not t.add_init_f0 and t.set_restrictions
                theta += [t.a, t.b, t.c, t.d]
        else:
            raise NotImplementedError("flow %s has no HIP implementation" % type(fl).__name__)
    spec = ops.FlowSpec(blocks, len(theta), len(nets), None)
    return spec, theta, nets
