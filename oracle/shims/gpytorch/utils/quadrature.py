import math
import numpy as np
import torch
from .broadcasting import _pad_with_singletons


class GaussHermiteQuadrature1D(torch.nn.Module):
    """(1/sqrt(pi)) * sum_s w_s f(sqrt(2 var) x_s + mean); nodes from numpy hermgauss."""

    def __init__(self, num_locs=20):
        super().__init__()
        self.num_locs = num_locs
        x, w = np.polynomial.hermite.hermgauss(num_locs)
        self.locations = torch.Tensor(x)
        self.weights = torch.Tensor(w)

    def forward(self, func, gaussian_dists):
        mean = gaussian_dists.mean
        var = gaussian_dists.variance
        x = _pad_with_singletons(self.locations, 0, mean.dim())
        vals = func(torch.sqrt(2.0 * var) * x + mean)
        w = _pad_with_singletons(self.weights, 0, vals.dim() - 1)
        return ((1.0 / math.sqrt(math.pi)) * (vals * w)).sum(0)
