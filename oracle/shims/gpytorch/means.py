import torch


class Mean(torch.nn.Module):
    pass


class ZeroMean(Mean):
    def forward(self, x):
        return torch.zeros(x.shape[:-1], dtype=x.dtype, device=x.device)


class MultitaskMean(Mean):
    pass
