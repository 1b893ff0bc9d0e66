"""Data path: UCI regression splits, z-scored with the train statistics, served by a loader with
the torch DataLoader iteration protocol whose tensors are resident on the GPU -- the reference re-collates 8611
rows item by item and copies them H2D every step (code/dsp/data/data.py:86-88, trainers/trainer_base.py:330).

`return_dataset(name, batch_size, use_validation, seed, options)` keeps the reference's call signature and return
value (code/dsp/data/datasets.py:81-221): `data_loaders` = [train, test] or [train, valid, test], `data_config` with
the reference's keys.  'power' / 'boston' read <root>/<name>.csv (no header row, target = last column) and the row
indices `splits_idx_<name>.pkl['seed_k']['train'|'test']` exactly as code/dsp/data/uci_datasets.py:62-107 does --
these are the reference's data files, found through options['root'] or $TGP_DATA_ROOT.  'synthetic_power' /
'synthetic_boston' generate seeded data of the same shape (no network in the build environment).
"""
import os
import pickle

import numpy
import torch

from . import config as cg

SHAPES = {"power": (9568, 4, 8611), "boston": (506, 13, 455)}


class DeviceLoader:
    """Iterates (x, y) minibatches already resident on `device`; len() = number of batches.

    Shuffling follows torch's RandomSampler (what the reference's DataLoader(shuffle=True, generator=gen) uses,
    code/dsp/data/data.py:42-60): one `torch.randperm(n, generator)` per epoch from a generator seeded with
    cg.config_seed, consecutive slices of it are the minibatches, the last one ragged (drop_last=False).
    `epoch_permutation()` hands the same permutation to the resident minibatch engine."""

    def __init__(self, X, Y, batch_size, shuffle=False, device=None, seed=0):
        dev = device or cg.device
        self.X, self.Y = X.to(dev), Y.to(dev)
        self.batch_size, self.shuffle = int(batch_size), shuffle
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        self.dataset = self

    def __len__(self):
        return (self.X.shape[0] + self.batch_size - 1) // self.batch_size

    def epoch_permutation(self):
        """Row order of the next epoch (int64, host), or None when the order is the stored one."""
        n = self.X.shape[0]
        if not self.shuffle:
            return None
        return torch.randperm(n, generator=self.gen)

    def __iter__(self):
        n = self.X.shape[0]
        perm = self.epoch_permutation()
        if perm is not None and self.batch_size < n:
            perm = perm.to(self.X.device)
            for i in range(0, n, self.batch_size):
                idx = perm[i:i + self.batch_size]
                yield self.X[idx], self.Y[idx]
        elif perm is not None:
            # one full batch: the ELBO is a sum over rows, the order only changes the summation order
            yield self.X, self.Y
        else:
            for i in range(0, n, self.batch_size):
                yield self.X[i:i + self.batch_size], self.Y[i:i + self.batch_size]


def standard_normalization(X_tr, Y_tr, X_va, Y_va, X_te, Y_te):
    """z-score by the train split's mean and *population* standard deviation (numpy.std, ddof=0) + 1e-15,
    code/dsp/data/data.py:260-299.  numpy float64 in, numpy out; returns (..., Y_std) with Y_std of shape (Dy,)."""
    eps = 1e-15
    X_mean, X_std = numpy.mean(X_tr, 0), numpy.std(X_tr, 0) + eps
    Y_mean, Y_std = numpy.mean(Y_tr, 0), numpy.std(Y_tr, 0) + eps
    zx = lambda a: None if a is None else (a - X_mean) / X_std
    zy = lambda a: None if a is None else (a - Y_mean) / Y_std
    return zx(X_tr), zy(Y_tr), zx(X_va), zy(Y_va), zx(X_te), zy(Y_te), Y_std


def random_split_validation(X, Y, seed, N_val):
    """code/dsp/data/data.py:216-234: numpy.random.seed(seed) permutation, the last N_val rows validate."""
    n = X.shape[0]
    assert N_val <= n, "Got more validation points {} than total training size {}".format(N_val, n)
    numpy.random.seed(seed)
    aux = numpy.random.permutation(n)
    tr, va = aux[0:n - N_val], aux[n - N_val:]
    return X[tr, :], Y[tr], X[va, :], Y[va]


# the regression sets of code/dsp/data/uci_datasets.py whose CSV and split pickle ship with the reference
# (datasets/regression/uci): file, separator, target column (uci_datasets.py:173-283; `index` splits X | Y as
# data[:, :index], data[:, index] -- energy keeps its second-to-last column as the target and drops the last).
# SCOPE: the reference's main.py accepts only 'boston' and 'power' (code/main.py:50); those two are the supported
# surface, pinned end to end (loader == reference loader, README table).  The other six entries are OUTSIDE SURVEY 8:
# their loaders agree with the reference's loader (tests/golden/uci_loaders_seed1.npz) but no model run on them is
# tested against the reference, and main.py here does not offer them either.
UCI = {
    "boston": ("boston.csv", ",", -1),
    "concrete": ("concrete.csv", ",", -1),
    "kin8nm": ("kin8nm.csv", ",", -1),
    "energy": ("energy.csv", ",", -2),
    "power": ("power.csv", ",", -1),
    "wine_red": ("wine-red.csv", ",", -1),
    "wine_white": ("wine-white.csv", ";", -1),
    "naval": ("naval.tsv", "   ", -1),
}


def load_uci_split(base, seed, root):
    """(X_tr, Y_tr, X_te, Y_te, tr_idx, te_idx) of the split stored on disk (code/dsp/data/uci_datasets.py:73-97)."""
    import pandas                   # the reference parses the CSV with pandas (data.py:186); numpy.loadtxt rounds
    fname, sep, index = UCI[base]                   # a few Boston entries differently (1 ulp)
    csv = os.path.join(root, fname)
    if not os.path.exists(csv):
        raise FileNotFoundError("%s not found: point options['root'] / $TGP_DATA_ROOT at the reference's "
                                "code/datasets/regression/uci directory" % csv)
    data = pandas.read_csv(csv, sep=sep, header=None, engine="python" if len(sep) > 1 else "c").to_numpy()
    stem = fname.split(".")[0]
    with open(os.path.join(root, "splits_idx_%s.pkl" % stem), "rb") as fh:
        split_dict = pickle.load(fh)
    key = "seed_" + str(seed)
    if key not in split_dict:
        raise KeyError("split %s not in splits_idx_%s.pkl (has %d splits)" % (key, stem, len(split_dict)))
    tr_idx, te_idx = split_dict[key]["train"], split_dict[key]["test"]
    data_tr, data_te = data[tr_idx], data[te_idx]
    return (data_tr[:, :index], data_tr[:, index].reshape(-1, 1), data_te[:, :index], data_te[:, index].reshape(-1, 1),
            tr_idx, te_idx)


def _synthetic(name, seed):
    n, d, _ = SHAPES[name]
    rng = numpy.random.default_rng(1234)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal(d)
    Y = numpy.sin(X @ w) + 0.1 * X[:, 0] ** 2 + 0.05 * rng.standard_normal(n)
    return X, Y.reshape(-1, 1)


def return_dataset(dataset_name, batch_size, use_validation=None, seed=None, options=None):
    options = options or {}
    synth = dataset_name.startswith("synthetic_")
    base = dataset_name.replace("synthetic_", "")
    if (synth and base not in SHAPES) or (not synth and base not in UCI):
        raise ValueError("Unkown dataset provided {}".format(dataset_name))
    if not options.get("split_from_disk", True):
        raise ValueError("only the splits stored on disk are supported (split_from_disk=True, as code/main.py sets it)")
    if synth:
        n, d, n_tr = SHAPES[base]
        X, Y = _synthetic(base, seed)
        perm = numpy.random.default_rng(seed).permutation(n)
        tr_idx, te_idx = perm[:n_tr], perm[n_tr:]
        X_tr, Y_tr, X_te, Y_te = X[tr_idx], Y[tr_idx], X[te_idx], Y[te_idx]
    else:
        root = options.get("root", os.environ.get("TGP_DATA_ROOT", ""))
        X_tr, Y_tr, X_te, Y_te, tr_idx, te_idx = load_uci_split(base, seed, root)
    X_va = Y_va = None
    if use_validation is not None:          # [seed, N_val], uci_datasets.py:54-56
        X_tr, Y_tr, X_va, Y_va = random_split_validation(X_tr, Y_tr, use_validation[0], use_validation[1])
    X_tr, Y_tr, X_va, Y_va, X_te, Y_te, Y_std = standard_normalization(X_tr, Y_tr, X_va, Y_va, X_te, Y_te)
    t = lambda a: None if a is None else torch.tensor(a, dtype=cg.dtype)
    X_tr, Y_tr, X_va, Y_va, X_te, Y_te = (t(a) for a in (X_tr, Y_tr, X_va, Y_va, X_te, Y_te))
    shuffle = options.get("shuffle_train", True)
    train = DeviceLoader(X_tr, Y_tr, batch_size, shuffle=shuffle, seed=cg.config_seed)
    test = DeviceLoader(X_te, Y_te, batch_size)
    loaders = [train, test]
    if use_validation is not None:
        loaders = [train, DeviceLoader(X_va, Y_va, batch_size, shuffle=shuffle, seed=cg.config_seed), test]
    data_config = {"X_tr": X_tr, "Y_tr": Y_tr, "X_va": X_va, "Y_va": Y_va, "X_te": X_te, "Y_te": Y_te,
                   "N_tr": X_tr.shape[0], "N_va": 0 if X_va is None else X_va.shape[0], "N_te": X_te.shape[0],
                   "Dx": X_tr.shape[1], "Dy": 1, "Y_std": Y_std, "X_all": None, "Y_all": None,
                   "train_idx": numpy.asarray(tr_idx), "test_idx": numpy.asarray(te_idx)}
    return loaders, data_config
