"""Kernel stand-ins: ScaleKernel(RBFKernel(ard)) as used at utils_models.py:188-193 of the reference."""
import torch
from torch.nn.functional import softplus


class _Dense:
    def __init__(self, t):
        self._t = t

    def evaluate(self):
        return self._t


class Kernel(torch.nn.Module):
    def __init__(self, ard_num_dims=None, batch_shape=torch.Size([]), **kwargs):
        super().__init__()
        self.ard_num_dims = ard_num_dims
        self.batch_shape = batch_shape

    def __call__(self, x1, x2=None, diag=False, **params):
        if x2 is None:
            x2 = x1
        out = self.forward(x1, x2, diag=diag, **params)
        return out if diag else _Dense(out)


class RBFKernel(Kernel):
    def __init__(self, ard_num_dims=None, batch_shape=torch.Size([]), **kwargs):
        super().__init__(ard_num_dims, batch_shape)
        d = 1 if ard_num_dims is None else ard_num_dims
        self.raw_lengthscale = torch.nn.Parameter(torch.zeros(*batch_shape, 1, d))

    @property
    def lengthscale(self):
        return softplus(self.raw_lengthscale)

    def forward(self, x1, x2, diag=False, **params):
        a = x1 / self.lengthscale
        b = x2 / self.lengthscale
        if diag:
            if a.shape == b.shape and torch.equal(a, b):
                return torch.ones(a.shape[:-1], dtype=a.dtype, device=a.device)
            return torch.exp(-0.5 * (a - b).pow(2).sum(-1))
        shift = a.mean(-2, keepdim=True)
        a = a - shift
        b = b - shift
        sq = a.pow(2).sum(-1, keepdim=True) - 2.0 * a.matmul(b.transpose(-2, -1)) \
            + b.pow(2).sum(-1, keepdim=True).transpose(-2, -1)
        return torch.exp(-0.5 * sq.clamp_min(0.0))


class MaternKernel(RBFKernel):
    """gpytorch 1.1.1 MaternKernel.forward (autograd branch): inputs centred on the mean of x1, divided by the
    lengthscale, distance = sqrt(clamp(sq_dist, 1e-30)), exp(-sqrt(2 nu) d) times the nu-dependent polynomial."""

    def __init__(self, nu=1.5, **kwargs):
        super().__init__(**kwargs)
        if nu not in (0.5, 1.5, 2.5):
            raise RuntimeError("nu expected to be 0.5, 1.5, or 2.5")
        self.nu = nu

    def forward(self, x1, x2, diag=False, **params):
        import math
        mean = x1.reshape(-1, x1.size(-1)).mean(0)[(None,) * (x1.dim() - 1)]
        a = (x1 - mean) / self.lengthscale
        b = (x2 - mean) / self.lengthscale
        if diag:
            d = (a - b).pow(2).sum(-1).clamp_min(1e-30).sqrt()
        else:
            shift = a.mean(-2, keepdim=True)
            a = a - shift
            b = b - shift
            sq = a.pow(2).sum(-1, keepdim=True) - 2.0 * a.matmul(b.transpose(-2, -1)) \
                + b.pow(2).sum(-1, keepdim=True).transpose(-2, -1)
            d = sq.clamp_min(1e-30).sqrt()
        e = torch.exp(-math.sqrt(self.nu * 2) * d)
        if self.nu == 0.5:
            c = 1
        elif self.nu == 1.5:
            c = (math.sqrt(3) * d).add(1)
        else:
            c = (math.sqrt(5) * d).add(1).add(5.0 / 3.0 * d ** 2)
        return c * e


class ScaleKernel(Kernel):
    def __init__(self, base_kernel, batch_shape=torch.Size([]), **kwargs):
        super().__init__(None, batch_shape)
        self.base_kernel = base_kernel
        self.raw_outputscale = torch.nn.Parameter(torch.zeros(*batch_shape))

    @property
    def outputscale(self):
        return softplus(self.raw_outputscale)

    def forward(self, x1, x2, diag=False, **params):
        base = self.base_kernel.forward(x1, x2, diag=diag, **params)
        s = self.outputscale
        s = s.view(*s.shape, *([1] * (1 if diag else 2)))
        return base * s


class AdditiveKernel(Kernel):
    pass


class ProductKernel(Kernel):
    pass


class PeriodicKernel(Kernel):
    pass


class CosineKernel(Kernel):
    pass
