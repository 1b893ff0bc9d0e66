"""Stand-in for jmaronas/pytorch_library@version-1.5.0 (absent here, layer order UNPINNED).

Assumed order: Linear -> activation -> Dropout(p) when p > 0 (bn = 0 and std = 0.0 in every config
the reference's main.py reaches).  Test infrastructure only.
"""
import torch.nn as nn


def return_activation(name):
    table = {"relu": nn.ReLU, "tanh": nn.Tanh, "linear": nn.Identity, "sigmoid": nn.Sigmoid}
    return table[name]()


class apply_linear(nn.Module):
    def __init__(self, inp, out, act, shape=None, std=0.0, drop=0.0, bn=0):
        super().__init__()
        self.w = nn.Linear(inp, out)
        self.act = return_activation(act)
        self.drop = nn.Dropout(drop) if drop > 0 else None

    def forward(self, x):
        x = self.act(self.w(x))
        if self.drop is not None:
            x = self.drop(x)
        return x


def compute_calibration_measures(*args, **kwargs):
    raise NotImplementedError("classification-only helper; not on the regression path")
