"""Trainer with the reference's constructor/`train`/`compute_metrics` surface
(code/dsp/trainers/trainer_base.py:250-391, trainers_regression.py:30-338).

Per minibatch: model.set_is_training(True) -> loss = -ELBO -> zero_grad -> backward -> optimizer.step()
(trainer_base.py:334-345), Adam lr as given, `optimisation_schedule` = (percentages, specifications) with
specifications entries [lr, name_substring] or [lr, weight_decay, name_substring] (main.py:274-298 puts the
'NNets' weights in a weight-decay group).  Data stay wherever the loader yields them; use data.DeviceLoader
to keep the split resident in HBM (removes the reference's per-step H2D copy and per-row collation).
`inference_in_cpu` is accepted for signature compatibility and ignored: metrics are computed on the GPU.
"""
import time
from collections import OrderedDict

import numpy
import torch

from . import config as cg


def return_optimizer(opt, parameters, lr):
    """dsp/trainers/optimizers.py:10-22."""
    if opt == "adam":
        return torch.optim.Adam(parameters, lr, amsgrad=False)
    if opt == "adam_W":
        return torch.optim.AdamW(parameters, lr, amsgrad=False)
    if opt == "sgd":
        return torch.optim.SGD(parameters, lr)
    raise ValueError("opt must be adam, adam_W or sgd")


class Trainer_SP_regression:
    def __init__(self, model, data_loaders, validate_each, plot, track, Y_std, plot_each, S_test,
                 inference_in_cpu=False):
        self.model = model
        # [train], [train, test] or [train, valid, test] (trainer_base.py:50-59)
        dl = list(data_loaders)
        self.train_loader, self.valid_loader, self.test_loader = dl[0], None, None
        if len(dl) == 3:
            self.valid_loader, self.test_loader = dl[1], dl[2]
        elif len(dl) == 2:
            self.test_loader = dl[1]
        self.is_valid, self.is_test = self.valid_loader is not None, self.test_loader is not None
        self.num_outputs = model.out_dim
        self.validate_each = validate_each
        self.S_test = S_test
        self.inference_in_cpu = inference_in_cpu
        if inference_in_cpu:
            # the reference moves the model to the CPU for compute_metrics (trainers_regression.py:321-338); the product has
            # no CPU implementation, so the flag is accepted and metrics stay on the GPU -- said once, not silently
            print("[tgp.pytorch_amd] inference_in_cpu=True ignored: metrics are computed on the GPU (no CPU path in this package)")
        assert len(Y_std.shape) == 1 and Y_std.shape[0] == self.num_outputs
        self.Y_std = Y_std
        self.optimizer = None
        self._engine = None
        self._names_added = []
        self.loss_arr, self.ELL_arr, self.KLD_arr = [], [], []
        self.total_trainer_epochs = 0

    # ---- optimiser groups (trainer_base.py:106-248) ---------------------------------------------------
    def _param_groups(self, specs, lr_all, already_added=()):
        """Parameter groups of one schedule stage.  `specs` entries are [lr, name_substring] or [lr, weight_decay,
        name_substring]; parameters no entry names take (lr_all, 0).  lr == 0.0 freezes: the parameter joins no group
        (and, with a kept optimiser, stays free to be added by a later stage).  `already_added`: names a kept optimiser
        holds already (keep_parameter_groups=True) -- skipped, and naming one again is an error like in the reference.
        Returns (groups, names placed in them)."""
        named = OrderedDict(self.model.named_parameters())
        taken, frozen, groups, placed = set(), set(), [], []
        seen_keys = set()
        for sp in (specs or []):
            if len(sp) == 3:
                lr, wd, key = sp
            elif len(sp) == 2:
                (lr, key), wd = sp, 0.0
            else:
                raise ValueError("Unvalid argument optimisation_schedule. Parameters should be specified as lr, param or "
                                 "lr, weight_decay, param")
            if key in seen_keys:
                raise ValueError("Parameter {} already added to the parameter list".format(key))
            seen_keys.add(key)
            names = [n for n in named if key in n]
            for n in names:
                if n in taken and lr != 0.0:
                    raise ValueError("Got repeated parameter {}".format(n))
                if n in already_added:
                    raise ValueError("Got repeated parameter {} with root {}. This parameter already belongs to a "
                                     "parameter group".format(n, key))
            if lr == 0.0:
                frozen.update(names)
                continue
            taken.update(names)
            if names:
                for g in groups:
                    if g["lr"] == lr and g["weight_decay"] == wd:
                        g["params"] += [named[n] for n in names]
                        break
                else:
                    groups.append({"params": [named[n] for n in names], "lr": lr, "weight_decay": wd})
                placed += names
        rest = [n for n in named if n not in taken and n not in frozen and n not in already_added]
        if rest:
            groups.append({"params": [named[n] for n in rest], "lr": lr_all, "weight_decay": 0.0})
            placed += rest
        return groups, placed

    # ---- resident fast path ---------------------------------------------------------------------------
    def _engine_for(self, groups, lr_ALL, opt):
        """The resident step engine (engine.ElboEngine: the same ELBO -> backward -> Adam sequence as the loop below, every
        launch captured in a HIP graph, no host synchronisation per step) when the run is what main.py sets up: Adam with
        one learning rate, weight decay on the 'NNets' group only, one full batch per epoch from a data.DeviceLoader.
        The model's nn.Parameters are re-pointed at the engine's flat buffer, so the modules stay the parameter holders
        and everything downstream (metrics, prediction, state_dict) sees the trained values.  Returns None otherwise."""
        from .data import DeviceLoader
        from .engine import ElboEngine, MinibatchEngine
        from .flow import compile_flow, mlp_spec
        from .likelihoods import GaussianLinearMean
        if not getattr(cg, "use_step_engine", True) or opt != "adam":
            return None
        ld = self.train_loader
        if not isinstance(ld, DeviceLoader) or not ld.X.is_cuda or ld.Y.shape[1] != 1:
            return None
        model = self.model
        if not hasattr(model, "_gp_params") or any(g["lr"] != lr_ALL for g in groups):
            return None
        # the engine updates EVERY parameter of its flat buffer: a stage that freezes some (lr = 0.0 entries,
        # trainer_base.py:155-179) or leaves some to a later stage must take the torch optimiser
        if {id(q) for g in groups for q in g["params"]} != {id(q) for q in model.parameters()}:
            return None
        nets, theta_list, blocks = [], [], None
        if not isinstance(model.likelihood, GaussianLinearMean):
            spec, theta_list, nets = compile_flow(model.G_matrix[0])
            blocks = spec.blocks
        nn_params = [p for net in nets for p in net.parameters()]
        nn_ids = {id(p) for p in nn_params}
        wd = 0.0
        for g in groups:
            if g["weight_decay"] != 0.0:
                if {id(p) for p in g["params"]} != nn_ids or wd != 0.0:
                    return None
                wd = g["weight_decay"]
        mspec = mlp_spec(nets, seed=cg.config_seed) if nets else None
        if nets and mspec is None:
            return None
        Z, rl, ro, m, Lam, lvn = model._gp_params()
        params = {"Z": Z.detach(), "raw_lengthscale": rl.detach(), "raw_outputscale": ro.detach(), "m": m.detach(),
                  "Lam": Lam.detach(), "log_var_noise": lvn.detach()}
        if theta_list:
            params["theta"] = torch.stack([p.detach().reshape(()) for p in theta_list])
        W = torch.cat([p.detach().reshape(-1) for p in nn_params]) if nn_params else None
        kw = dict(flow_blocks=blocks, S=getattr(model, "quad_points", None), lr=lr_ALL,
                  kernel=model.covariance_function.hip_kernel, mlp=mspec, mlp_weights=W, nn_weight_decay=wd, mlp_training=True,
                  jitter_ladder=cg.global_jitter if cg.global_jitter is not None else 1e-8)
        if len(ld) == 1:
            eng = ElboEngine(ld.X, ld.Y, params, float(model.N), device=ld.X.device, **kw)
        else:
            # several minibatches per epoch (the Airline recipe, main.py:74): rows gathered on the device by index,
            # one captured step per batch size (full batches + the ragged last one), see engine.MinibatchEngine
            eng = MinibatchEngine(ld.X, ld.Y, params, float(model.N), ld.batch_size, device=ld.X.device, **kw)
            eng.has_order = bool(ld.shuffle)
        # the modules' parameters become views of the engine's flat buffer
        fp, k = eng.fp, model.covariance_function
        with torch.no_grad():
            model.Z.data = fp.view("Z").view_as(model.Z)
            k.base_kernel.raw_lengthscale.data = fp.view("raw_ls").view_as(k.base_kernel.raw_lengthscale)
            k.raw_outputscale.data = fp.view("raw_os").view_as(k.raw_outputscale)
            model.q_U.variational_mean.data = fp.view("m").view_as(model.q_U.variational_mean)
            model.q_U.chol_variational_covar.data = fp.view("Lam").view_as(model.q_U.chol_variational_covar)
            model.likelihood.log_var_noise.data = fp.view("lvn").view_as(model.likelihood.log_var_noise)
            for i, p in enumerate(theta_list):
                p.data = fp.view("theta")[i:i + 1].view_as(p)
            o = 0
            for p in nn_params:
                p.data = fp.view("nn")[o:o + p.numel()].view_as(p)
                o += p.numel()
        eng.capture()
        return eng

    def _train_resident(self, eng, n_epochs, epochs_total):
        from .engine import MinibatchEngine
        mb = isinstance(eng, MinibatchEngine)
        spe = eng.steps_per_epoch if mb else 1          # optimiser steps per epoch
        hist = torch.empty(max(n_epochs * spe, 1), 3, dtype=torch.float64, device=eng.device)
        t0 = time.time()
        last = 0
        ep_done = 0
        for ep in range(n_epochs):
            if mb:
                # the loader's own permutation stream (torch RandomSampler semantics), written to the device once per
                # epoch; the batches are then gathered by index inside the captured steps
                eng.set_order(self.train_loader.epoch_permutation())
                eng.run_epoch(hist, ep * spe)
            elif ep >= ep_done:
                # one full-batch step per epoch: all epochs up to the next report in as few graph launches as the engine's
                # unrolled graph allows (engine.replay_many); the scalars of every step land in `hist` on the device
                nxt = n_epochs if self.validate_each <= 0 else min(n_epochs, (ep // self.validate_each + 1) * self.validate_each)
                eng.replay_many(nxt - ep, hist, ep)
                ep_done = nxt
            self.total_trainer_epochs += 1
            if self.validate_each > 0 and (ep + 1) % self.validate_each == 0:
                h = hist[last * spe:(ep + 1) * spe].mean(0).cpu()     # the only host sync: once per `validate_each` epochs
                eng.check_status()                      # a failed Cholesky surfaces here, not thousands of epochs later
                print("| Epoch [{}/{}] ELBO {:.5f} ELL {:.5f} KLD {:.5f} ({:.3f}s)".format(
                    ep + 1, epochs_total, float(h[0]), float(h[1]), float(h[2]), time.time() - t0))
                t0, last = time.time(), ep + 1
        eng.check_status()
        h = hist[:n_epochs * spe].cpu()
        self.loss_arr += (-h[:, 0]).tolist()
        self.ELL_arr += h[:, 1].tolist()
        self.KLD_arr += h[:, 2].tolist()

    def ELBO_call(self, x, y):
        loss, elogl, kld = self.model.ELBO(x, y)
        loss = -loss
        self.param_elbo = [loss, elogl, kld]
        return loss

    def train(self, epochs, lr_ALL, opt, keep_parameter_groups, lr_groups=None, optimisation_schedule=None):
        """trainer_base.py:250-361.  keep_parameter_groups=True keeps the optimiser (its groups and moments) across
        stages and calls: later stages ADD the groups of parameters not yet held, earlier groups take `lr_groups` (or
        lr_ALL).  keep_parameter_groups=False starts from a fresh optimiser and re-creates it at every stage."""
        if optimisation_schedule is None:
            optimisation_schedule = ([1.0], [None])
        percentages, specifications = optimisation_schedule
        if abs(sum(percentages) - 1.0) > 1e-12:
            raise ValueError("percentages must sum 1, got {}".format(sum(percentages)))
        if len(percentages) != len(specifications):
            raise ValueError("Percentages and specifications must have same length")
        if keep_parameter_groups:
            if self.optimizer is None and self._engine is None:
                self._names_added = []
            elif self.optimizer is not None:
                if lr_groups is not None and len(lr_groups) != len(self.optimizer.param_groups):
                    raise ValueError("The provided `lr_groups` {} does not match the number of parameter groups in the "
                                     "optimizer {}".format(lr_groups, len(self.optimizer.param_groups)))
                for g, lr_i in zip(self.optimizer.param_groups, lr_groups or [lr_ALL] * len(self.optimizer.param_groups)):
                    g["lr"] = lr_i
            elif lr_groups is not None and any(v != lr_ALL for v in lr_groups):
                raise ValueError("the resident step engine holds one learning rate: pass lr_groups=None (or all equal to "
                                 "lr_ALL), or set config.use_step_engine = False")
        else:
            self.optimizer, self._engine, self._names_added = None, None, []
        single = len(percentages) == 1
        for per, specs in zip(percentages, specifications):
            fresh = not keep_parameter_groups or (self.optimizer is None and self._engine is None)
            groups, placed = self._param_groups(specs, lr_ALL, () if fresh else self._names_added)
            n_ep = int(epochs * per)
            if self._engine is not None and not fresh:
                if groups:
                    raise ValueError("the resident step engine already trains every parameter; new groups cannot be added")
                if self._engine.lr != float(lr_ALL):       # the learning rate is a launch argument of the captured step
                    self._engine.set_lr(float(lr_ALL))
                    self._engine.capture()
                self._train_resident(self._engine, n_ep, epochs)
                continue
            if fresh:
                self._engine = self._engine_for(groups, lr_ALL, opt) if single else None
                self._names_added = list(placed) if keep_parameter_groups else []
                if self._engine is not None:
                    self._train_resident(self._engine, n_ep, epochs)
                    continue
                self.optimizer = return_optimizer(opt, groups, lr_ALL)
            else:
                for g in groups:
                    self.optimizer.add_param_group(g)
                self._names_added += placed
            self._train_eager(n_ep, epochs)
        if not keep_parameter_groups:
            self.optimizer = None
            self._engine = None

    def _train_eager(self, n_epochs, epochs_total):
        """The reference's loop as is: per minibatch ELBO -> zero_grad -> backward -> optimizer.step
        (trainer_base.py:329-349), autograd through ops.ElboFunction."""
        for ep in range(n_epochs):
            t0 = time.time()
            acc = [0.0, 0.0, 0.0]
            nb = 0
            for x, y in self.train_loader:
                x, y = x.to(cg.device), y.to(cg.device)
                assert x.dim() == 2 and y.dim() == 2, "x and y must be (MB,D) and (MB,D')"
                self.model.set_is_training(True)
                loss = self.ELBO_call(x, y)
                self.optimizer.zero_grad()
                loss.backward()
                self.optimizer.step()
                self.model.set_is_training(False)
                lv = loss.item()
                self.loss_arr.append(lv)
                self.ELL_arr.append(self.param_elbo[1].item())
                self.KLD_arr.append(self.param_elbo[2].item())
                acc[0] += -lv
                acc[1] += self.ELL_arr[-1]
                acc[2] += self.KLD_arr[-1]
                nb += 1
            self.total_trainer_epochs += 1
            if self.validate_each > 0 and (ep + 1) % self.validate_each == 0:
                print("| Epoch [{}/{}] ELBO {:.5f} ELL {:.5f} KLD {:.5f} ({:.3f}s)".format(
                    ep + 1, epochs_total, acc[0] / nb, acc[1] / nb, acc[2] / nb, time.time() - t0))

    # ---- metrics (trainers_regression.py:108-224, 317-338) ----------------------------------------------
    def performance_metrics(self, X, Y):
        self.model.set_is_training(False)
        logp, (m1, _m2) = self.model.test_log_likelihood(X, Y, return_moments=True, Y_std=self.Y_std.to(X.device),
                                                         S_MC_NNet=self.S_test if self.model.fully_bayesian else None)
        samples, _, _ = self.model.sample_from_predictive_distribution(X, S=self.S_test)      # (Dy,S,N,1)
        q = numpy.quantile(samples.to("cpu").numpy(), [0.025, 0.975], axis=1)                  # (2,Dy,N,1)
        y = Y[:, 0].to("cpu")
        cover = ((y >= torch.tensor(q[0, 0, :, 0])) & (y <= torch.tensor(q[1, 0, :, 0]))).float().sum().item()
        se = ((m1.reshape(-1) - Y[:, 0]) ** 2).sum().item()
        return float(logp.reshape(-1)[0]), se, cover

    def _loader_metrics(self, loader):
        tot, lp, se, cov = 0.0, 0.0, 0.0, 0.0
        for x, y in loader:
            x, y = x.to(cg.device), y.to(cg.device)
            a, b, c = self.performance_metrics(x, y)
            lp, se, cov, tot = lp + a, se + b, cov + c, tot + x.size(0)
        ystd = float(self.Y_std.reshape(-1)[0])
        return lp / tot, ystd * numpy.sqrt(se / tot), cov / tot

    def compute_metrics(self):
        """(logL, rmse, coverage) for train, valid, test -- nine floats like the reference."""
        out = list(self._loader_metrics(self.train_loader))
        out += list(self._loader_metrics(self.valid_loader)) if self.is_valid else [0.0, 0.0, 0.0]
        out += list(self._loader_metrics(self.test_loader)) if self.is_test else [0.0, 0.0, 0.0]
        return tuple(out)
