"""Small helpers with the reference's names (code/dsp/utils.py)."""
import torch

from . import config as cg


def inv_softplus(x):
    """gpytorch.utils.transforms.inv_softplus."""
    x = torch.as_tensor(x, dtype=torch.get_default_dtype())
    return x + torch.log(-torch.expm1(-x))


def positive_transform(x):
    """dsp/utils.py:39-46 ('exp' is the only transform main.py configures)."""
    if cg.positive_transform == "exp":
        return torch.exp(x)
    if cg.positive_transform == "softplus":
        return torch.log(torch.exp(x) + 1)
    raise NotImplementedError("positive_transform function %s is not implemented." % cg.positive_transform)


def inverse_positive_transform(x):
    if cg.positive_transform == "exp":
        return torch.log(x)
    if cg.positive_transform == "softplus":
        return torch.log(torch.exp(x) - 1.0)
    raise NotImplementedError("inverse_positive_transform for %s is not implemented." % cg.positive_transform)


def KMEANS(X, num_Z, n_init=1, seed=None):
    """Inducing-point initialisation (dsp/utils.py:143-159): sklearn KMeans on the host, as in the reference."""
    from sklearn.cluster import KMeans
    if seed is None:
        seed = cg.config_seed
    km = KMeans(n_clusters=num_Z, init="k-means++", n_init=n_init, random_state=seed).fit(X.to("cpu").numpy())
    return torch.tensor(km.cluster_centers_, dtype=cg.dtype).to(cg.device)


def psd_safe_cholesky(A, upper=False, out=None, jitter=None):
    """dsp/utils.py:222-270 on the GPU (HIP blocked Cholesky + the reference's jitter ladder): (L, A_used)."""
    from . import ops
    if cg.constant_jitter is not None:
        A.diagonal(dim1=-2, dim2=-1).add_(cg.constant_jitter)
    squeeze = A.dim() == 3
    A2 = A[0] if squeeze else A
    L, Ap = ops.psd_safe_cholesky(A2, jitter)
    if upper:
        L = L.transpose(-1, -2)
    return (L.unsqueeze(0), Ap.unsqueeze(0)) if squeeze else (L, Ap)
