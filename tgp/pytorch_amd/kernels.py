"""Kernel factory with the reference's signature (code/dsp/models/utils_models.py:145-280).  main.py only builds
'scale_rbf' = gpytorch ScaleKernel(RBFKernel(ard)); the module below keeps gpytorch's parameter names
(`raw_outputscale`, `base_kernel.raw_lengthscale`) and softplus constraints; evaluation runs on the GPU."""
import torch
import torch.nn as nn
from torch.nn.functional import softplus

from . import config as cg
from . import ops
from .utils import inv_softplus


class RBFKernel(nn.Module):
    def __init__(self, ard_num_dims=None, batch_shape=torch.Size([])):
        super().__init__()
        self.ard_num_dims = ard_num_dims
        self.batch_shape = batch_shape
        d = 1 if ard_num_dims is None else ard_num_dims
        self.raw_lengthscale = nn.Parameter(torch.zeros(*batch_shape, 1, d, dtype=cg.dtype))

    @property
    def lengthscale(self):
        return softplus(self.raw_lengthscale)


class MaternKernel(RBFKernel):
    """gpytorch MaternKernel parameter holder; nu = 1.5 is the only smoothness with a HIP implementation
    ('scale_matern32', models/utils_models.py:199-204)."""

    def __init__(self, nu=1.5, ard_num_dims=None, batch_shape=torch.Size([])):
        super().__init__(ard_num_dims, batch_shape)
        if nu != 1.5:
            raise NotImplementedError("MaternKernel: only nu=1.5 has a HIP implementation in this build")
        self.nu = nu


class ScaleKernel(nn.Module):
    """sigma^2 * k(|(x - z)/l|) with sigma^2 = softplus(raw_outputscale), l = softplus(raw_lengthscale);
    k = exp(-r^2/2) (RBFKernel) or (1 + sqrt3 r) exp(-sqrt3 r) (MaternKernel, nu = 1.5)."""

    def __init__(self, base_kernel, batch_shape=torch.Size([])):
        super().__init__()
        self.base_kernel = base_kernel
        self.batch_shape = batch_shape
        self.raw_outputscale = nn.Parameter(torch.zeros(*batch_shape, dtype=cg.dtype))

    @property
    def outputscale(self):
        return softplus(self.raw_outputscale)

    @property
    def hip_kernel(self):
        """The instance_kernel name the HIP library knows this covariance function by."""
        return "scale_matern32" if isinstance(self.base_kernel, MaternKernel) else "scale_rbf"

    def _params(self, idx=0):
        D = self.base_kernel.raw_lengthscale.shape[-1]
        return (self.base_kernel.raw_lengthscale.detach().reshape(-1, D)[idx].contiguous(),
                self.raw_outputscale.detach().reshape(-1)[idx:idx + 1].contiguous())

    def forward(self, x1, x2=None, diag=False, **params):
        """Dense K(x1, x2) (or its diagonal) for output 0, evaluated by the HIP K_NM / K_MM kernels (no autograd;
        the training path never materialises K_NM)."""
        x1_ = x1[0] if x1.dim() == 3 else x1
        if diag:
            return self.outputscale.reshape(-1, 1).detach() * torch.ones(x1.shape[:-1], dtype=x1.dtype, device=x1.device)
        raw_ls, raw_os = self._params()
        x2_ = None if x2 is None else (x2[0] if x2.dim() == 3 else x2).contiguous()
        K = ops.kernel_matrix(x1_.contiguous(), x2_, raw_ls, raw_os, kernel=self.hip_kernel)
        return _Dense(K.unsqueeze(0) if x1.dim() == 3 else K)


class _Dense:
    def __init__(self, t):
        self._t = t

    def evaluate(self):
        return self._t


def instance_kernel(name, ard_num_dim, num_multioutput, kernel_is_shared, init_params={}, kernels=None):
    if ard_num_dim is not None and not isinstance(ard_num_dim, int):
        raise ValueError("ard_num_dim must be None or int, got {}".format(type(ard_num_dim)))
    ls = init_params.get("length_scale", 1.0)
    ks = init_params.get("kernel_scale", 1.0)
    if kernel_is_shared:
        num_multioutput = 1
    if name not in ("scale_rbf", "scale_matern32"):
        raise NotImplementedError("kernel '%s': 'scale_rbf' (the kernel main.py builds, code/main.py:229) and "
                                  "'scale_matern32' have a HIP implementation in this build" % name)
    if name == "scale_matern32":
        rbf = MaternKernel(nu=1.5, ard_num_dims=ard_num_dim, batch_shape=torch.Size([num_multioutput]))
    else:
        rbf = RBFKernel(ard_num_dims=ard_num_dim, batch_shape=torch.Size([num_multioutput]))
    rbf.raw_lengthscale.data = inv_softplus(torch.ones(num_multioutput, 1, rbf.raw_lengthscale.size(-1), dtype=cg.dtype) * ls)
    K = ScaleKernel(rbf, batch_shape=torch.Size([num_multioutput]))
    K.raw_outputscale.data = inv_softplus(torch.ones(num_multioutput, dtype=cg.dtype) * ks)
    return K
