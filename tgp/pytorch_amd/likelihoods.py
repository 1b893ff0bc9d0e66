"""Likelihood classes with the reference's constructor signatures and attribute names
(code/dsp/likelihoods/GaussianLinearMean.py, GaussianNonLinearMean.py).  They hold `log_var_noise`
(observation noise, positive transform = exp) and evaluate their moments on the GPU; the expected
log-likelihood used in training lives inside the fused ELBO kernel (ops.ElboFunction)."""
import torch
import torch.nn as nn
import torch.distributions as td

from . import config as cg
from . import ops
from .flow import compile_flow
from .utils import inverse_positive_transform, positive_transform


class _GaussianBase(nn.Module):
    def __init__(self, out_dim, noise_init, noise_is_shared):
        super().__init__()
        self.out_dim = out_dim
        self.noise_is_shared = noise_is_shared
        n = 1 if noise_is_shared else out_dim
        init = inverse_positive_transform(torch.tensor(noise_init, dtype=cg.dtype))
        self.log_var_noise = nn.Parameter(torch.ones(n, 1, dtype=cg.dtype) * init)

    def _lvn(self):
        return self.log_var_noise.expand(self.out_dim, 1) if self.noise_is_shared else self.log_var_noise

    def sample_from_output(self, f, i, **kwargs):
        var = positive_transform(self._lvn()[i])
        return td.Normal(f, torch.ones_like(f) * torch.sqrt(var)).sample()


class GaussianLinearMean(_GaussianBase):
    """p(y|f) = N(y|f, s2): closed-form ELL (GaussianLinearMean.py:60-87) and moments (:89-118)."""

    def expected_log_prob(self, Y, gauss_mean, gauss_cov, **kwargs):
        # differentiable in the moments and the noise like the reference's torch code (gradients from the same launch)
        return ops.EllGaussFunction.apply(Y.reshape(-1), gauss_mean.reshape(-1).contiguous(), gauss_cov.reshape(-1).contiguous(),
                                          self._lvn().reshape(-1)[:1].contiguous())

    def marginal_moments(self, gauss_mean, gauss_cov, diagonal=True, **kwargs):
        assert diagonal, "only diagonal covariances on this path"
        C_Y = positive_transform(self._lvn()).detach().expand(-1, gauss_mean.size(1)) + gauss_cov
        return gauss_mean.clone(), C_Y


class GaussianNonLinearMean(_GaussianBase):
    """p(y|G(f)) with Gauss-Hermite integration over q(f) (GaussianNonLinearMean.py:64-203)."""

    def __init__(self, out_dim, noise_init, noise_is_shared, quadrature_points):
        super().__init__(out_dim, noise_init, noise_is_shared)
        self.quad_points = quadrature_points

    def _flow_inputs(self, flow, X, dev, with_grad=False):
        spec, theta_list, nets = compile_flow(flow[0])
        if with_grad:     # keep the graph to the flow's parameters (expected_log_prob)
            theta = torch.stack([p.reshape(()) for p in theta_list]).to(dev) if theta_list else None
        else:
            theta = torch.stack([p.detach().reshape(()) for p in theta_list]).to(dev) if theta_list else None
        rowp = None
        if nets:
            from .flow import nets_rowp
            X2d = X[0] if X.dim() == 3 else X
            rowp = nets_rowp(nets, X2d, with_grad=with_grad)      # the HIP MLP kernel (dropout follows the layers)
        return spec, theta, rowp

    def expected_log_prob(self, Y, gauss_mean, gauss_cov, flow, X, **kwargs):
        assert len(flow) == self.out_dim == 1, "one flow per output; Dy = 1 on this path"
        # differentiable in the moments, the noise and the flow's parameters like the reference's torch code
        spec, theta, rowp = self._flow_inputs(flow, X, gauss_mean.device, with_grad=True)
        return ops.EllFlowFunction.apply(Y.reshape(-1), gauss_mean.reshape(-1).contiguous(), gauss_cov.reshape(-1).contiguous(),
                                         self._lvn().reshape(-1)[:1].contiguous(), theta, rowp, spec, self.quad_points)

    def marginal_moments(self, gauss_mean, gauss_cov, flow, X, **kwargs):
        assert len(flow) == self.out_dim == 1
        spec, theta, rowp = self._flow_inputs(flow, X, gauss_mean.device)
        m1, m2, _ = ops.predict(gauss_mean.reshape(-1).contiguous(), gauss_cov.reshape(-1).contiguous(),
                                self._lvn().detach().reshape(-1)[:1].contiguous(), spec, theta, self.quad_points, rowp)
        return m1.reshape(gauss_mean.shape), m2.reshape(gauss_mean.shape)
