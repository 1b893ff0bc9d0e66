"""Stand-in for jmaronas/pytorch_library@version-1.5.0 (absent here, layer order UNPINNED).

Assumed order: Linear -> activation -> Dropout(p) when p > 0 (bn = 0 and std = 0.0 in every config
the reference's main.py reaches).  Test infrastructure only.
"""
import os

import torch.nn as nn

# "post" (default): Linear -> activation -> Dropout.  "pre": Dropout -> Linear -> activation (dropout on the layer's
# INPUT) -- the other order a library could implement; Linear -> Dropout -> activation equals "post" for relu (both are
# non-negative elementwise scalings) and is not a separate case for the reference's Power recipe.  Switch used by the
# round-3 sweep that tried to explain the ID_TGP offset against the README table (profiles/r03_readme_table.txt).
ORDER = os.environ.get("TGP_SHIM_APPLY_LINEAR_ORDER", "post")


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
        if ORDER == "pre":
            if self.drop is not None:
                x = self.drop(x)
            return self.act(self.w(x))
        x = self.act(self.w(x))
        if self.drop is not None:
            x = self.drop(x)
        return x


def compute_calibration_measures(*args, **kwargs):
    raise NotImplementedError("classification-only helper; not on the regression path")
