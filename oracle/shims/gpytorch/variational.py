import torch


class CholeskyVariationalDistribution(torch.nn.Module):
    """Parameter holder only (sparse_MF_SP.py:164-174 overwrites both tensors)."""

    def __init__(self, num_inducing_points, batch_shape=torch.Size([])):
        super().__init__()
        self.variational_mean = torch.nn.Parameter(torch.zeros(*batch_shape, num_inducing_points))
        eye = torch.eye(num_inducing_points).repeat(*batch_shape, 1, 1)
        self.chol_variational_covar = torch.nn.Parameter(eye)
