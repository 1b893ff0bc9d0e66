"""Data path: UCI-style regression splits, z-scored with the train statistics
(code/dsp/data/data.py:260-299), served by a loader with the torch DataLoader iteration protocol whose
tensors are resident on the GPU -- the reference re-collates 8611 rows item by item and copies them H2D
every step (code/dsp/data/data.py:86-88, trainers/trainer_base.py:330).

`return_dataset(name, batch_size, use_validation, seed, options)` keeps the reference's call signature
(code/dsp/data/datasets.py:81-221).  'power' / 'boston' read <root>/<name>.csv + splits_idx_<name>.pkl
(the reference's own data files, located through options['root'] or $TGP_DATA_ROOT); 'synthetic_power' /
'synthetic_boston' generate seeded data of the same shape (no network in the build environment).
"""
import os
import pickle

import numpy
import torch

from . import config as cg

SHAPES = {"power": (9568, 4, 8611), "boston": (506, 13, 455)}


class DeviceLoader:
    """Iterates (x, y) minibatches already resident on `device`; len() = number of batches."""

    def __init__(self, X, Y, batch_size, shuffle=False, device=None, seed=0):
        dev = device or cg.device
        self.X, self.Y = X.to(dev), Y.to(dev)
        self.batch_size, self.shuffle = int(batch_size), shuffle
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        self.dataset = self

    def __len__(self):
        return (self.X.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.X.shape[0]
        if self.shuffle and self.batch_size < n:
            perm = torch.randperm(n, generator=self.gen).to(self.X.device)
            for i in range(0, n, self.batch_size):
                idx = perm[i:i + self.batch_size]
                yield self.X[idx], self.Y[idx]
        else:
            for i in range(0, n, self.batch_size):
                yield self.X[i:i + self.batch_size], self.Y[i:i + self.batch_size]


def standard_normalization(X_tr, Y_tr, X_te, Y_te):
    """z-score by train statistics (+1e-15), code/dsp/data/data.py:262-268."""
    mx, sx = X_tr.mean(0), X_tr.std(0) + 1e-15
    my, sy = Y_tr.mean(0), Y_tr.std(0) + 1e-15
    return (X_tr - mx) / sx, (Y_tr - my) / sy, (X_te - mx) / sx, (Y_te - my) / sy, sy


def _synthetic(name, seed):
    n, d, _ = SHAPES[name]
    rng = numpy.random.default_rng(1234)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal(d)
    Y = numpy.sin(X @ w) + 0.1 * X[:, 0] ** 2 + 0.05 * rng.standard_normal(n)
    return X, Y.reshape(-1, 1)


def return_dataset(dataset_name, batch_size, use_validation=None, seed=1, options=None):
    options = options or {}
    synth = dataset_name.startswith("synthetic_")
    base = dataset_name.replace("synthetic_", "")
    if base not in SHAPES:
        raise ValueError("dataset must be power, boston, synthetic_power or synthetic_boston")
    n, d, n_tr = SHAPES[base]
    if synth:
        X, Y = _synthetic(base, seed)
        perm = numpy.random.default_rng(seed).permutation(n)
        tr, te = perm[:n_tr], perm[n_tr:]
    else:
        root = options.get("root", os.environ.get("TGP_DATA_ROOT", ""))
        csv = os.path.join(root, base + ".csv")
        if not os.path.exists(csv):
            raise FileNotFoundError("%s not found: point options['root'] / $TGP_DATA_ROOT at the reference's "
                                    "code/datasets/regression/uci directory" % csv)
        arr = numpy.loadtxt(csv, delimiter=",", skiprows=1) if base == "power" else numpy.genfromtxt(csv, delimiter=",", skip_header=1)
        X, Y = arr[:, :d], arr[:, d:d + 1]
        with open(os.path.join(root, "splits_idx_%s.pkl" % base), "rb") as fh:
            splits = pickle.load(fh)
        sp = splits["seed_%d" % seed] if isinstance(splits, dict) else splits[seed]
        tr, te = numpy.asarray(sp[0]), numpy.asarray(sp[1])
    t = lambda a: torch.tensor(a, dtype=cg.dtype)
    X_tr, Y_tr, X_te, Y_te, y_std = standard_normalization(t(X[tr]), t(Y[tr]), t(X[te]), t(Y[te]))
    loaders = [DeviceLoader(X_tr, Y_tr, batch_size, shuffle=options.get("shuffle_train", True), seed=cg.config_seed),
               None, DeviceLoader(X_te, Y_te, batch_size)]
    data_config = {"Dx": d, "Dy": 1, "X_tr": X_tr, "Y_tr": Y_tr, "N_tr": X_tr.shape[0], "Y_std": float(y_std[0]),
                   "X_te": X_te, "Y_te": Y_te}
    return loaders, data_config
