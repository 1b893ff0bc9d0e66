"""Drop-in model classes: `sparse_MF_SP` (TGP / ID_TGP) and `sparse_MF_GP` (SVGP).

Same constructor signatures, attribute and nn.Parameter names, method names, argument meaning and return
shapes as the reference (code/dsp/models/sparse_MF_SP.py:47, sparse_MF_GP.py:40), so the reference's
main.py / trainer idiom works unchanged:

    ELBO, ELL, KLD = model.ELBO(x, y);  (-ELBO).backward();  optimizer.step()

What differs is underneath: every number is produced by the HIP kernels of libtgp_hip.so (fused row kernel,
blocked Cholesky, hand-derived adjoints, the MLP kernel for the input-dependent flows' networks in training AND in
every evaluation method).  There is NO CPU fallback: calling these methods with CPU tensors
raises (the oracle in oracle/ is the CPU restatement, and it is test infrastructure only).
Restrictions of this build (asserted): one output GP (Dy = 1, all BASELINE configs), whitened q(u),
zero mean, 'scale_rbf' kernel, float64.
"""
from typing import List

import numpy
import torch
import torch.nn as nn

from . import config as cg
from . import ops
from .flow import CompositeFlow, IdentityFlow, compile_flow, instance_flow
from .likelihoods import GaussianLinearMean, GaussianNonLinearMean
from .utils import positive_transform

DEFAULT_INIT = {"variational_distribution": {"variance_scale": 1.0, "mean_scale": 0.0}}


class CholeskyVariationalDistribution(nn.Module):
    """Parameter holder with gpytorch's names (the reference uses it as such, sparse_MF_SP.py:158-177)."""

    def __init__(self, num_inducing_points, batch_shape=torch.Size([])):
        super().__init__()
        self.variational_mean = nn.Parameter(torch.zeros(*batch_shape, num_inducing_points, dtype=cg.dtype))
        eye = torch.eye(num_inducing_points, dtype=cg.dtype).repeat(*batch_shape, 1, 1)
        self.chol_variational_covar = nn.Parameter(eye)


class ZeroMean(nn.Module):
    def forward(self, x):
        return torch.zeros(x.shape[:-1], dtype=x.dtype, device=x.device)


def enable_eval_dropout(modules):
    """code/dsp/models/utils_models.py:358-364."""
    found = False
    for module in modules:
        if "Dropout" in type(module).__name__:
            module.train()
            found = True
    return found


class sparse_MF_SP(nn.Module):
    def __init__(self, model_specs: list, X: torch.tensor, init_Z: torch.tensor, N: float, likelihood: nn.Module,
                 num_outputs: int, is_whiten: bool, K_is_shared: bool, mean_is_shared: bool, Z_is_shared: bool,
                 q_U_is_shared: bool, flow_specs: list, flow_connection: str, add_noise_inducing: float,
                 be_fully_bayesian: bool = False, init_params: dict = {}) -> None:
        super().__init__()
        assert len(model_specs) == 2, "Parameter model_specs should be len 2: mean name and kernel instance"
        assert int(num_outputs) == 1, "this build implements the single-output path (Dy = 1, every BASELINE config)"
        assert is_whiten, "only the whitened representation (main.py: whiten = True) has a HIP implementation"
        assert model_specs[0] == "zero", "only the 'zero' mean function (main.py) is provided"
        assert not (K_is_shared or mean_is_shared or Z_is_shared or q_U_is_shared), "sharing flags are False in main.py"
        self.out_dim = int(num_outputs)
        self.inp_dim = int(init_Z.size(1))
        self.kernel_is_shared, self.mean_is_shared = K_is_shared, mean_is_shared
        self.Z_is_shared, self.q_U_is_shared = Z_is_shared, q_U_is_shared
        self.N = float(N)
        self.M = init_Z.size(0)
        self.likelihood = likelihood
        self.fully_bayesian = be_fully_bayesian
        ip = dict(DEFAULT_INIT)
        ip.update(init_params)
        self.init_params = ip
        self.standard_sampler = None        # the reference re-creates a td.MultivariateNormal here; sampling uses torch.randn
        self.is_training = True
        self.quad_points = likelihood.quad_points if isinstance(likelihood, GaussianNonLinearMean) else cg.quad_points
        self.is_whiten = is_whiten

        # inducing points (sparse_MF_SP.py:140-156)
        Z = torch.zeros(self.out_dim, self.M, self.inp_dim, dtype=cg.dtype)
        for l in range(self.out_dim):
            aux = init_Z.clone().to(cg.dtype)
            if add_noise_inducing > 0.0:
                aux = init_Z * torch.tensor(add_noise_inducing * numpy.random.randn(self.M, self.inp_dim), dtype=cg.dtype)
            Z[l, :] = aux
        self.Z = nn.Parameter(Z)
        # q(u) (sparse_MF_SP.py:158-177)
        q_U = CholeskyVariationalDistribution(self.M, batch_shape=torch.Size([self.out_dim]))
        vs = ip["variational_distribution"]["variance_scale"]
        ms = ip["variational_distribution"]["mean_scale"]
        q_U.chol_variational_covar.data = torch.eye(self.M, dtype=cg.dtype).view(1, self.M, self.M).repeat(self.out_dim, 1, 1) * numpy.sqrt(vs)
        q_U.variational_mean.data = torch.ones(self.out_dim, self.M, dtype=cg.dtype) * ms
        self.q_U = q_U
        self.mean_function = ZeroMean()
        self.covariance_function = model_specs[1]
        # flows (sparse_MF_SP.py:232-266)
        assert flow_connection == "single", "flow_connection must be 'single'"
        assert len(flow_specs) == self.out_dim
        G = []
        for fl in flow_specs:
            G.append(instance_flow(fl) if isinstance(fl, list) else fl)
        self.G_matrix = nn.ModuleList(G)
        self.G_flow_connection = flow_connection
        self.l2_regularize = False
        self._cfg = {}

    # ---- configuration -----------------------------------------------------------------------------
    def be_fully_bayesian(self, mode):
        self.fully_bayesian = mode

    def set_is_training(self, mode):
        self.is_training = mode

    # ---- helpers ---------------------------------------------------------------------------------
    def _require_gpu(self, t):
        if not t.is_cuda:
            raise ops.L.TgpError("tgp.pytorch_amd models run on the GPU only (got a %s tensor); there is no CPU "
                                 "fallback in the product path" % t.device)
        if t.dtype != torch.float64:
            raise ops.L.TgpError("float64 only: call config.set_maximum_precission() before building the model "
                                 "(code/main.py:124)")

    def _gp_params(self):
        k = self.covariance_function
        return (self.Z[0], k.base_kernel.raw_lengthscale.reshape(-1), k.raw_outputscale.reshape(-1),
                self.q_U.variational_mean[0], self.q_U.chol_variational_covar[0],
                self.likelihood.log_var_noise.reshape(-1)[:1])

    def _flow_inputs(self, X2d, with_grad, samples=1):
        """(FlowSpec or None, theta, rowp): shared scalars stacked into one vector, per-row parameters from the MLPs on the
        HIP kernel (dropout follows the nets' Dropout layers, as in the reference).  `samples` > 1 (fully Bayesian
        evaluation): the rows are evaluated `samples` times in ONE launch, rowp row s * N + n, every (sample, row) with
        its own dropout mask -- the reference's X.repeat to (S_MC, N, Dx), models/sparse_MF_SP.py:753-758."""
        if isinstance(self.likelihood, GaussianLinearMean):
            return None, None, None
        spec, theta_list, nets = compile_flow(self.G_matrix[0])
        ctx = torch.enable_grad() if with_grad else torch.no_grad()
        with ctx:
            theta = torch.stack([p.reshape(()) for p in theta_list]) if theta_list else None
            rowp = None
            if nets:
                if "mlp" not in self._cfg:
                    from .flow import mlp_spec
                    self._cfg["mlp"] = mlp_spec(nets, seed=cg.config_seed)
                    self._cfg["mlp_step"] = torch.zeros(2, dtype=torch.int32, device=X2d.device)
                mspec = self._cfg["mlp"]
                if mspec is None:
                    raise ops.L.TgpError("the flow's parameter networks are outside the HIP MLP kernel's coverage (one "
                                         "architecture D -> H x L -> 1, H <= 64, 1 <= L <= 3, relu/tanh, dropout); this "
                                         "package has no torch.nn fallback for them")
                # all nets in one HIP launch (tgp_mlp_forward/backward_f64); a fresh dropout mask per call, the same
                # one for this call's backward (the counter moves before the forward, not after it)
                self._cfg["mlp_step"][0] += 1
                W = torch.cat([p.reshape(-1) for net in nets for p in net.parameters()])
                # dropout is on when the nets' Dropout layers are in train mode: in training, and in the fully
                # Bayesian evaluation, where enable_eval_dropout() re-enables ONLY those layers after eval()
                # (models/utils_models.py:358-364) -- the container's own .training flag is False there
                drop_on = any(mod.training for mod in nets[0].modules() if "Dropout" in type(mod).__name__)
                # the counter is snapshotted per call: a second forward before this call's backward (loss accumulated
                # over minibatches, an evaluation between ELBO() and backward()) must not change the mask the
                # backward recomputes
                Xs = X2d.contiguous() if samples == 1 else X2d.repeat(samples, 1)
                # evaluation draws from a mask stream of its own: the resident engine counts its training steps from 0
                # too, and MC-dropout samples at test time must not replay the masks of training step k
                mcall = mspec if with_grad else mspec.salted(ops.MASK_SALT_EVAL)
                rowp = ops.MlpFunction.apply(Xs, W, mcall, bool(drop_on), self._cfg["mlp_step"].clone())
        return spec, theta, rowp

    # ---- model computations ------------------------------------------------------------------------
    def marginal_variational_qf_parameters(self, X, diagonal: bool, is_duvenaud: bool, init_Z=None):
        """q(f) marginals (sparse_MF_SP.py:274-396): returns mu, cov of shape (Dy, MB, 1)."""
        assert diagonal and not is_duvenaud, "diagonal=True, is_duvenaud=False on this path"
        X2 = X[0] if X.dim() == 3 else X
        self._require_gpu(X2)
        Z, rl, ro, m, Lam, _ = self._gp_params()
        if torch.is_grad_enabled() and any(t.requires_grad for t in (Z, rl, ro, m, Lam)):
            # differentiable like the reference's (autograd through :274-396): tgp_qf_moments_bwd_f64 in the backward
            mu, v = ops.QfMomentsFunction.apply(X2.detach(), Z, rl, ro, m, Lam, self.covariance_function.hip_kernel)
        else:
            mu, v = ops.qf_moments(X2, *(t.detach() for t in (Z, rl, ro, m, Lam)), kernel=self.covariance_function.hip_kernel)
        return mu.reshape(1, -1, 1), v.reshape(1, -1, 1)

    def KLD(self):
        """Whitened KL (sparse_MF_SP.py:406-431), shape (Dy,); differentiable in (m, L_q) like the reference's."""
        m, Lam = self.q_U.variational_mean[0], self.q_U.chol_variational_covar[0]
        if torch.is_grad_enabled() and (m.requires_grad or Lam.requires_grad):
            return ops.KlFunction.apply(m, Lam).reshape(1)
        kl, _, _ = ops.kl_whitened(m.detach(), Lam.detach())
        return kl.reshape(1)

    def ELBO(self, X, Y):
        """Returns (ELBO, ELL, KLD): positive ELBO with autograd, the trainer negates (sparse_MF_SP.py:552-598)."""
        X2 = X[0] if X.dim() == 3 else X
        self._require_gpu(X2)
        assert Y.dim() == 2 and Y.shape[1] == 1, "Y must be (MB, 1)"
        Z, rl, ro, m, Lam, lvn = self._gp_params()
        spec, theta, rowp = self._flow_inputs(X2, with_grad=True)
        cfg = self._cfg
        cfg.update(N_total=self.N, flow=spec, S=self.quad_points, check_status=(cg.status_check == "always"),
                   global_jitter=cg.global_jitter, kernel=self.covariance_function.hip_kernel)
        elbo, ell, kld = ops.ElboFunction.apply(X2, Y, Z, rl, ro, m, Lam, lvn, theta, rowp, cfg)
        return elbo, ell, kld

    def check_status(self):
        """Lazy Cholesky status check (cg.status_check = 'lazy'): raises like psd_safe_cholesky would have."""
        st = self._cfg.get("last_status")
        if st is not None and ops.raise_for_status(st.cpu()):
            raise ops.NotPSDError("K_MM was not positive definite in the last ELBO call (pivot %d)" % int(st[0]))

    def ELL(self, X, Y, mean, cov):
        """N/MB * E_q(f)[log p(y|G(f))] from given moments (sparse_MF_SP.py:601-626); no autograd."""
        MB = Y.size(0)
        ell = self.likelihood.expected_log_prob(Y.t(), mean.squeeze(dim=2), cov.squeeze(dim=2), flow=self.G_matrix, X=X)
        return self.N / MB * ell

    def _eval_mode(self):
        self.eval()
        if self.fully_bayesian:
            assert enable_eval_dropout(self.modules()), "fully bayesian mode needs dropout layers in the flow"

    def predictive_distribution(self, X, diagonal: bool = True, S_MC_NNet: int = None):
        """m1, m2 (Dy, MB) + q(f) moments (sparse_MF_SP.py:457-540)."""
        assert not self.is_training, "This method only works in eval mode"
        assert diagonal
        X3 = X.repeat(self.out_dim, 1, 1) if X.dim() == 2 else X
        self._eval_mode()
        with torch.no_grad():
            mean_q_f, cov_q_f = self.marginal_variational_qf_parameters(X3, diagonal=True, is_duvenaud=False)
            if self.fully_bayesian:
                assert S_MC_NNet is not None
                # all S_MC dropout samples in ONE pass, as the reference does by expanding X to (S_MC, N, Dx)
                # (sparse_MF_SP.py:753-758): one MLP launch over S_MC * N rows (a mask per sample and row), one
                # tgp_predict_f64 launch, then the mixture moments (:516-531)
                S, MB = int(S_MC_NNet), X3.shape[1]
                spec, theta, rowp = self._flow_inputs(X3[0], with_grad=False, samples=S)
                lvn = self.likelihood.log_var_noise.detach().reshape(-1)[:1].contiguous()
                mY, cY, _ = ops.predict(mean_q_f.reshape(-1).repeat(S), cov_q_f.reshape(-1).repeat(S), lvn, spec,
                                        theta.detach() if theta is not None else None, self.quad_points, rowp)
                mY, cY = mY.reshape(1, S, MB), cY.reshape(1, S, MB)       # (Dy, S, MB)
                m1 = mY.mean(1)
                m2 = (cY + mY ** 2).mean(1) - m1 ** 2
            else:
                m1, m2 = self.likelihood.marginal_moments(mean_q_f.squeeze(2), cov_q_f.squeeze(2), diagonal=True,
                                                          flow=self.G_matrix, X=X3)
        self.train()
        return m1, m2, mean_q_f, cov_q_f

    def test_log_likelihood(self, X, Y, return_moments: bool, Y_std, S_MC_NNet: int = None):
        """log p(Y*|X*) summed over the batch, shape (Dy,), and optionally [m1, m2] (sparse_MF_SP.py:637-825)."""
        assert not self.is_training, "This method only works in eval mode"
        MB = X.size(0)
        X3 = X.repeat(self.out_dim, 1, 1) if X.dim() == 2 else X
        self._require_gpu(X3)
        predictive_params = None
        if return_moments:
            m1, m2, mean_q_f, cov_q_f = self.predictive_distribution(X3, diagonal=True, S_MC_NNet=S_MC_NNet)
            predictive_params = [m1, m2]
        else:
            self._eval_mode()
            with torch.no_grad():
                mean_q_f, cov_q_f = self.marginal_variational_qf_parameters(X3, diagonal=True, is_duvenaud=False)
        self._eval_mode()
        ystd = float(Y_std.reshape(-1)[0])
        mu, v = mean_q_f.reshape(-1).contiguous(), cov_q_f.reshape(-1).contiguous()
        lvn = self.likelihood.log_var_noise.detach().reshape(-1)[:1].contiguous()
        with torch.no_grad():
            if isinstance(self.likelihood, GaussianLinearMean):
                _, _, lp = ops.predict(mu, v, lvn, Y=Y, Y_std=ystd)
                log_p_y = lp.sum().reshape(1)
            else:
                # S_MC dropout samples (fully Bayesian) in ONE pass: the nets over S_MC * N rows, one tgp_predict_f64
                # launch over the same rows (sparse_MF_SP.py:753-768 expands X, Y the same way)
                n_mc = int(S_MC_NNet) if self.fully_bayesian else 1
                spec, theta, rowp = self._flow_inputs(X3[0], with_grad=False, samples=n_mc)
                rep = (lambda t: t.repeat(n_mc)) if n_mc > 1 else (lambda t: t)
                _, _, lp = ops.predict(rep(mu), rep(v), lvn, spec, theta.detach() if theta is not None else None,
                                       self.quad_points, rowp, Y=rep(Y.reshape(-1)), Y_std=ystd)
                # kernel: logsumexp_s[log(w_s/sqrt(pi)) + logN]; the reference sums log w_s + logN and subtracts
                # 0.5*log(pi) where cg.pi is a float32 tensor (sparse_MF_SP.py:768-776): rebuild exactly that
                lp = lp.reshape(n_mc, MB) + 0.5 * float(numpy.log(numpy.pi))
                # float32 arithmetic of the reference's constant, with the correctly rounded float32 log(pi) (a host
                # torch.log in float32 differs by 1 ulp between CPU ISAs, which would make the result host-dependent)
                log_pi32 = numpy.log(numpy.float32(numpy.pi))
                if self.fully_bayesian:
                    stack = lp - float(numpy.float32(0.5) * log_pi32)
                    log_p_y = (torch.logsumexp(stack, 0).sum() - MB * numpy.log(n_mc)).reshape(1)
                else:
                    log_p_y = (lp[0].sum() - float(numpy.float32(0.5 * MB) * log_pi32)).reshape(1)
        self.train()
        return log_p_y, predictive_params

    # ---- sampling (sparse_MF_SP.py:837-992) --------------------------------------------------------
    def sample_from_variational_marginal_base(self, X, diagonal: bool, is_duvenaud: bool, init_Z=None):
        if not diagonal:
            raise NotImplementedError("This function only works with diagonal=True")
        X3 = X.repeat(self.out_dim, 1, 1) if X.dim() == 2 else X
        Dy, SMB, _ = X3.shape
        mean_q_f, cov_q_f = self.marginal_variational_qf_parameters(X3, diagonal=True, is_duvenaud=is_duvenaud)
        e = torch.randn(Dy, SMB, 1, dtype=mean_q_f.dtype, device=mean_q_f.device)
        f = (e * cov_q_f.sqrt() + mean_q_f).squeeze(dim=2)
        return f, mean_q_f, cov_q_f

    def sample_from_variational_marginal(self, X, S: int, diagonal: bool, is_duvenaud: bool, init_Z=None):
        X3 = X.repeat(self.out_dim, 1, 1) if X.dim() == 2 else X
        X3 = X3.repeat(1, S, 1)
        if self.is_training:
            self.train()
        else:
            self._eval_mode()
        with torch.no_grad():
            f0, mean_q_f0, cov_q_f0 = self.sample_from_variational_marginal_base(X3, diagonal, is_duvenaud, init_Z)
            f = torch.stack([g(f0[i], X3[i]) for i, g in enumerate(self.G_matrix)], 0)
        self.train()
        return f, mean_q_f0, cov_q_f0, f0

    def sample_from_predictive_distribution(self, X, S: int) -> List[torch.tensor]:
        assert not self.is_training, "This method only works in eval mode"
        assert X.dim() == 2, "Invalid input X.shape"
        N, _ = X.shape
        with torch.no_grad():
            f_k, _, _, f_0 = self.sample_from_variational_marginal(X, S, diagonal=True, is_duvenaud=False)
            samples = [self.likelihood.sample_from_output(f_k, i).view(S, N, 1) for i in range(self.out_dim)]
        self.train()
        return torch.stack(samples, dim=0), f_k, f_0


class sparse_MF_GP(sparse_MF_SP):
    """SVGP (Hensman et al.): the same class with identity flows (sparse_MF_GP.py:40-64)."""

    def __init__(self, model_specs: list, X, init_Z, N: float, likelihood: nn.Module, num_outputs: int,
                 is_whiten: bool, K_is_shared: bool, mean_is_shared: bool, Z_is_shared: bool, q_U_is_shared: bool,
                 add_noise_inducing: float, init_params: dict = {}) -> None:
        flow_specs = [[("identity", [])] for _ in range(num_outputs)]
        super().__init__(model_specs, X, init_Z, N, likelihood, num_outputs, is_whiten, K_is_shared, mean_is_shared,
                         Z_is_shared, q_U_is_shared, flow_specs, "single", add_noise_inducing,
                         be_fully_bayesian=False, init_params=init_params)

    def sample_from_variational_marginal(self, X, S: int, diagonal: bool, is_duvenaud: bool, init_Z=None):
        X3 = X.repeat(self.out_dim, 1, 1) if X.dim() == 2 else X
        X3 = X3.repeat(1, S, 1)
        with torch.no_grad():
            f, mean_q_f, cov_q_f = self.sample_from_variational_marginal_base(X3, diagonal, is_duvenaud, init_Z)
        return f, mean_q_f, cov_q_f, f
