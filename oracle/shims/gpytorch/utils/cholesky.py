import torch


def psd_safe_cholesky(A, upper=False, out=None, jitter=None):
    L = torch.linalg.cholesky(A)
    return L.transpose(-1, -2) if upper else L
