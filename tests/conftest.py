import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Golden fixture -> dict of float64 torch tensors (+ 'program' as a list of int tuples)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        if k == "program":
            out[k] = [tuple(int(t) for t in row) for row in a]
        elif k in ("kernel", "data"):
            out[k] = str(a)
        elif a.dtype.kind == "f":
            out[k] = torch.from_numpy(np.array(a, dtype=np.float64))
        else:
            out[k] = torch.from_numpy(np.array(a))
    out["params"] = {k[2:]: out[k] for k in out if k.startswith("p_")}
    if "data" in out and "X" not in out:
        # full-size fixtures name the dataset fixture that holds their rows: "power_seed1" or "power_seed1[:1024]"
        base, _, cut = out["data"].partition("[:")
        d = load_golden(base)
        n = int(cut[:-1]) if cut else d["X_tr"].shape[0]
        out["X"], out["Y"] = d["X_tr"][:n], d["Y_tr"][:n]
        out["X_te"], out["Y_te"], out["Y_std"] = d["X_te"], d["Y_te"], d["Y_std"]
    out.setdefault("program", None)
    out.setdefault("kernel", "scale_rbf")
    return out


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


@pytest.fixture(scope="session")
def golden():
    return load_golden
