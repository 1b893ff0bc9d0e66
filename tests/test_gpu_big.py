"""GPU (-m gpu): the general-M path (M > 128: tiled float64 MFMA GEMM, multi-kernel blocked Cholesky, chunked rows)
through the C ABI against the oracle, plus the GEMM building block against torch.matmul."""
import os
import sys

import pytest
import torch

from conftest import rel_err
from test_gpu_parity import compare, run_hip, TOL_GRAD

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_matches_torch(ta, tb):
    from tgp.pytorch_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    m, n, k = 256, 384, 208
    A = torch.randn((k, m) if ta else (m, k), generator=g, dtype=torch.float64).to(dev)
    B = torch.randn((n, k) if tb else (k, n), generator=g, dtype=torch.float64).to(dev)
    C0 = torch.randn(m, n, generator=g, dtype=torch.float64).to(dev)
    ref = 0.7 * (A.T if ta else A) @ (B.T if tb else B) - 1.3 * C0
    C = ops.gemm(A, B, trans_a=ta, trans_b=tb, alpha=0.7, beta=-1.3, C=C0.clone())
    torch.cuda.synchronize()
    assert rel_err(C.cpu(), ref.cpu()) < 1e-13


def test_gemm_triangular_trimming():
    from tgp.pytorch_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(6)
    n = 384
    Lo = torch.tril(torch.randn(n, n, generator=g, dtype=torch.float64)).to(dev)
    X = torch.randn(n, n, generator=g, dtype=torch.float64).to(dev)
    cases = [
        (dict(trans_a=False, trans_b=False, tri=ops.TRI_A_LOWER), Lo, X, Lo @ X),
        (dict(trans_a=True, trans_b=False, tri=ops.TRI_A_UPPER), Lo, X, Lo.T @ X),
        (dict(trans_a=False, trans_b=False, tri=ops.TRI_B_LOWER), X, Lo, X @ Lo),
        (dict(trans_a=False, trans_b=True, tri=ops.TRI_B_UPPER), X, Lo, X @ Lo.T),
        (dict(trans_a=False, trans_b=True, tri=ops.TRI_A_LOWER | ops.TRI_B_UPPER), Lo, Lo, Lo @ Lo.T),
        (dict(trans_a=True, trans_b=False, tri=ops.TRI_A_UPPER | ops.TRI_B_LOWER), Lo, Lo, Lo.T @ Lo),
    ]
    for kw, A, B, ref in cases:
        C = ops.gemm(A, B, **kw)
        torch.cuda.synchronize()
        assert rel_err(C.cpu(), ref.cpu()) < 1e-13, kw
    # lower block triangle only: tiles above the diagonal stay untouched
    C = ops.gemm(X, X, trans_b=True, tri=ops.TRI_C_LOWER, C=torch.full((n, n), 7.0, dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    ref = (X @ X.T).cpu()
    C = C.cpu()
    for bi in range(3):
        for bj in range(3):
            blk = C[128 * bi:128 * bi + 128, 128 * bj:128 * bj + 128]
            if bj <= bi:
                assert rel_err(blk, ref[128 * bi:128 * bi + 128, 128 * bj:128 * bj + 128]) < 1e-13
            else:
                assert float((blk - 7.0).abs().max()) == 0.0


def _oracle_case(N, D, M, flow, S, seed=3, kernel="scale_rbf", lengthscale=None):
    from oracle import tgp_oracle as orc
    prob = orc.synthetic_problem(N, D, M, seed=seed, flow=flow, S=S)
    if lengthscale is not None:       # softplus^-1 of the wanted lengthscale, every dimension
        ls = torch.full((D,), float(lengthscale), dtype=torch.float64)
        prob["params"]["raw_lengthscale"] = torch.log(torch.expm1(ls))
    (elbo, ell, kld), og = orc.elbo_and_grads(prob["X"], prob["Y"], prob["params"], prob["N_total"], prob["program"],
                                              prob["xs"], prob["ws"], prob["rowp"], kernel=kernel)
    g = dict(prob)
    g["kernel"] = kernel
    g.update(ELBO=elbo, ELL=ell, KLD=kld, g_Z=og["Z"], g_raw_lengthscale=og["raw_lengthscale"],
             g_raw_outputscale=og["raw_outputscale"], g_m=og["m"], g_Lam=og["Lam"], g_log_var_noise=og["log_var_noise"])
    if "theta" in og:
        g["g_theta"] = og["theta"]
    if "rowp" in og:
        g["g_rowp"] = og["rowp"]
    return g


@pytest.mark.parametrize("N,D,M,flow,S", [(300, 8, 200, "tanh5x6", 32), (1000, 4, 129, None, 8), (700, 8, 300, "sal2", 32),
                                           (400, 5, 150, "idsal3", 16), (2000, 8, 1000, "tanh5x6", 32)])
def test_big_elbo_step_matches_oracle(N, D, M, flow, S):
    g = _oracle_case(N, D, M, flow, S)
    out, grads, status, (mu, v) = run_hip(g)
    assert int(status[0]) == 0 and int(status[1]) == 0
    compare(out, grads, g)
    assert float(torch.triu(grads["Lam"], 1).abs().max()) == 0.0


@pytest.mark.parametrize("name", ["tiny_matern_svgp", "med_matern_sal2", "med_matern_tanh2x2"])
def test_matern32_matches_reference_fixture(name):
    """'scale_matern32' runs on the general path whatever M is; fixtures from the reference's model classes."""
    from conftest import load_golden
    g = load_golden(name)
    out, grads, status, (mu, v) = run_hip(g)
    assert int(status[0]) == 0 and int(status[1]) == 0
    compare(out, grads, g)
    assert rel_err(mu, g["mu"]) < 1e-9
    assert float(((v - g["v"]).abs() / g["v"].abs()).max()) < 1e-6


@pytest.mark.parametrize("N,D,M,flow,S", [(400, 5, 150, "sal2", 16), (300, 8, 20, "tanh3x2", 8), (500, 3, 260, None, 8)])
def test_matern32_matches_oracle(N, D, M, flow, S):
    g = _oracle_case(N, D, M, flow, S, seed=6, kernel="scale_matern32")
    out, grads, status, _ = run_hip(g)
    assert int(status[0]) == 0
    compare(out, grads, g)


def test_matern32_kernel_matrix_and_model_class():
    """instance_kernel('scale_matern32') -> ScaleKernel(MaternKernel): dense K through tgp_kernel_matrix_f64 and one
    ELBO + backward through the model class, against the oracle."""
    from oracle import tgp_oracle as orc
    from tgp.pytorch_amd import config as cg
    from tgp.pytorch_amd.kernels import instance_kernel
    from tgp.pytorch_amd.likelihoods import GaussianLinearMean
    from tgp.pytorch_amd.models import sparse_MF_GP
    dev = _dev()
    cg.set_maximum_precission()
    prob = orc.synthetic_problem(200, 4, 30, seed=8, flow=None, S=8)
    p = prob["params"]
    K = instance_kernel("scale_matern32", ard_num_dim=4, num_multioutput=1, kernel_is_shared=False,
                        init_params={"length_scale": 2.0, "kernel_scale": 2.0}).to(dev)
    with torch.no_grad():
        K.raw_outputscale.data = p["raw_outputscale"].reshape(1).to(dev)
        K.base_kernel.raw_lengthscale.data = p["raw_lengthscale"].reshape(1, 1, 4).to(dev)
    Kd = K(prob["X"].to(dev), p["Z"].to(dev)).evaluate()
    ref = orc.scale_matern32(prob["X"], p["Z"], p["raw_lengthscale"], p["raw_outputscale"])
    assert rel_err(Kd.cpu(), ref) < 1e-12
    lik = GaussianLinearMean(out_dim=1, noise_init=0.05, noise_is_shared=False)
    model = sparse_MF_GP(["zero", K], prob["X"].to(dev), p["Z"].clone().to(dev), 200, lik, 1, True, False, False, False,
                         False, 0.0, init_params={"variational_distribution": {"variance_scale": 1e-5, "mean_scale": 0.0}})
    model = model.to(dev)
    with torch.no_grad():
        model.q_U.variational_mean.data = p["m"].reshape(1, -1).to(dev)
        model.q_U.chol_variational_covar.data = p["Lam"].reshape(1, 30, 30).to(dev)
        model.likelihood.log_var_noise.data = p["log_var_noise"].reshape(1, 1).to(dev)
    elbo, ell, kld = model.ELBO(prob["X"].to(dev), prob["Y"].to(dev))
    elbo.backward()
    (e_o, l_o, k_o), og = orc.elbo_and_grads(prob["X"], prob["Y"], p, prob["N_total"], None, prob["xs"], prob["ws"], None,
                                             kernel="scale_matern32")
    assert rel_err(elbo.detach().cpu(), e_o) < 1e-9
    assert rel_err(model.Z.grad[0].cpu(), og["Z"]) < 1e-7
    assert rel_err(K.base_kernel.raw_lengthscale.grad.reshape(-1).cpu(), og["raw_lengthscale"]) < 1e-7
    assert rel_err(K.raw_outputscale.grad.reshape(-1).cpu(), og["raw_outputscale"]) < 1e-7


def test_big_qf_moments_matches_oracle():
    from oracle import tgp_oracle as orc
    from tgp.pytorch_amd import ops
    dev = _dev()
    prob = orc.synthetic_problem(900, 6, 260, seed=4, flow=None, S=8)
    p = prob["params"]
    mu_o, v_o = orc.qf_moments(prob["X"], p["Z"], p["raw_lengthscale"], p["raw_outputscale"], p["m"], p["Lam"])
    mu, v = ops.qf_moments(prob["X"].to(dev), p["Z"].to(dev), p["raw_lengthscale"].to(dev), p["raw_outputscale"].to(dev),
                           p["m"].to(dev), p["Lam"].to(dev))
    torch.cuda.synchronize()
    assert rel_err(mu.cpu(), mu_o.reshape(-1)) < 1e-9
    assert rel_err(v.cpu(), v_o.reshape(-1)) < 1e-7


def _chunked_cases(cases, chunk_rows):
    """The cases through the general-M path with row chunks of at most `chunk_rows` rows (tgp_model.plan =
    TGP_PLAN_CHUNK_ROWS: the chunk size is a property of the call -- it was a process-global environment switch, and these
    tests child processes, until round 6)."""
    from tgp.pytorch_amd import lib
    for N, D, M, flow, S in cases:
        g = _oracle_case(N, D, M, flow, S, seed=5)
        out, grads, status, (mu, v) = run_hip(g, plan=lib.plan_chunk_rows(chunk_rows))
        assert int(status[0]) == 0
        compare(out, grads, g)


def test_big_path_several_row_chunks():
    """Chunks of 256 rows: 4 row chunks, the last one ragged -- accumulation across chunks, the two-buffer chunk pipeline."""
    _chunked_cases([(1000, 6, 160, "sal2", 16), (700, 5, 140, "idsal2", 8), (901, 4, 130, None, 8)], 256)


def test_big_path_chunks_in_line_equal_the_overlapped_pipeline():
    """TGP_PLAN_NO_CHUNK_OVERLAP (one set of chunk buffers, forward and backward of the chunks in line) against the
    two-buffer pipeline: the same sums in the same order -- bit-identical."""
    from tgp.pytorch_amd import lib
    g = _oracle_case(1000, 6, 160, "sal2", 16, seed=5)
    o1, g1, st1, _ = run_hip(g, plan=lib.plan_chunk_rows(256))
    o2, g2, st2, _ = run_hip(g, plan=lib.plan_chunk_rows(256) | lib.PLAN_NO_CHUNK_OVERLAP)
    assert int(st1[0]) == 0 and int(st2[0]) == 0
    assert torch.equal(o1, o2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


@pytest.mark.parametrize("graph", [False, True])
def test_big_engine_adam_history_matches_oracle(graph):
    """Resident engine on the general-M path: 4 Adam steps (tgp_adam_dev_f64), eager and replayed from a HIP graph,
    against the oracle stepped with torch.optim.Adam on the CPU."""
    from oracle import tgp_oracle as orc
    from tgp.pytorch_amd.engine import ElboEngine
    prob = orc.synthetic_problem(600, 5, 150, seed=9, flow="sal2", S=16)
    leaves = {k: t.clone().requires_grad_(True) for k, t in prob["params"].items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=0.01)
    ref = []
    for _ in range(4):
        elbo, ell, kl = orc.elbo(prob["X"], prob["Y"], leaves["Z"], leaves["raw_lengthscale"], leaves["raw_outputscale"],
                                 leaves["m"], leaves["Lam"], leaves["log_var_noise"], prob["N_total"], prob["program"],
                                 leaves.get("theta"), prob["xs"], prob["ws"])
        ref.append([float(elbo.detach()), float(ell.detach()), float(kl.detach())])
        opt.zero_grad()
        (-elbo).backward()
        opt.step()
    eng = ElboEngine(prob["X"], prob["Y"], prob["params"], float(prob["N_total"]), flow_blocks=prob["program"], S=16,
                     device=_dev())
    hist = []
    if graph:
        eng.capture()
    for _ in range(4):
        (eng.replay if graph else eng.step)()
        hist.append(list(eng.scalars()))
    eng.check_status()
    assert rel_err(torch.tensor(hist, dtype=torch.float64), torch.tensor(ref, dtype=torch.float64)) < 1e-8
    assert rel_err(eng.fp.view("Z").cpu(), leaves["Z"].detach()) < 1e-7


def test_big_step_is_bit_reproducible():
    g = _oracle_case(500, 6, 200, "tanh3x2", 16, seed=11)
    o1, g1, _, _ = run_hip(g)
    o2, g2, _, _ = run_hip(g)
    assert torch.equal(o1, o2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


def test_big_non_psd_reports_pivot():
    """Duplicate inducing points make K_MM singular: status[0] is the first failing pivot (LAPACK info), as for M <= 128.
    (Whether a singular pivot comes out <= 0 is a matter of rounding; the last row duplicating the first one does for
    every seed tried: d = K_00 - fl(sqrt(K_00))^2 - sum of squares, and fl(sqrt 2)^2 > 2.)"""
    from oracle import tgp_oracle as orc
    prob = orc.synthetic_problem(400, 4, 200, seed=7, flow=None, S=8)
    prob["params"]["Z"][199] = prob["params"]["Z"][0]
    g = dict(prob)
    g.update(program=None)
    from tgp.pytorch_amd import ops
    dev = _dev()
    p = {k: t.to(dev) for k, t in prob["params"].items()}
    out, grads, status, _ = ops.elbo_step(prob["X"].to(dev), prob["Y"].to(dev), p["Z"], p["raw_lengthscale"],
                                           p["raw_outputscale"], p["m"], p["Lam"], p["log_var_noise"], 400.0)
    torch.cuda.synchronize()
    assert int(status[0]) == 200


@pytest.mark.parametrize("M", [130, 300, 1000])
def test_big_standalone_cholesky_matches_torch(M):
    """tgp_cholesky_f64 above 128 (blocked multi-kernel factorisation + block-row inverse) against torch.linalg."""
    from tgp.pytorch_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M)
    B = torch.randn(M, M, generator=g, dtype=torch.float64)
    A = B @ B.T / M + torch.eye(M, dtype=torch.float64)
    Lo, Li, status = ops.cholesky(A.to(dev), want_inverse=True)
    torch.cuda.synchronize()
    assert int(status[0]) == 0 and int(status[1]) == 0
    Lref = torch.linalg.cholesky(A)
    assert rel_err(Lo.cpu(), Lref) < 1e-12
    assert rel_err(Li.cpu(), torch.linalg.inv(Lref)) < 1e-10
    assert float(torch.triu(Lo, 1).abs().max()) == 0.0
    A[M // 2, M // 2] = -1.0                      # not positive definite: LAPACK-style info = first failing pivot
    _, _, status = ops.cholesky(A.to(dev))
    torch.cuda.synchronize()
    assert int(status[0]) == M // 2 + 1


def test_big_jitter_ladder_recovers():
    """The reference's psd_safe_cholesky protocol (dsp/utils.py:256-269) on the general-M path: the duplicated inducing
    point of test_big_non_psd_reports_pivot -> status -> retry with jitter 1e-8 -> NumericalWarning, finite results,
    identical to the step launched with that jitter up front (the retry is a plain re-launch)."""
    from oracle import tgp_oracle as orc
    from tgp.pytorch_amd import ops
    dev = _dev()
    prob = orc.synthetic_problem(400, 4, 200, seed=7, flow=None, S=8)
    prob["params"]["Z"][199] = prob["params"]["Z"][0]
    p = {k: t.to(dev) for k, t in prob["params"].items()}
    args = (prob["X"].to(dev), prob["Y"].to(dev), p["Z"], p["raw_lengthscale"], p["raw_outputscale"], p["m"], p["Lam"],
            p["log_var_noise"], 400.0)
    with pytest.warns(ops.NumericalWarning):
        out, grads, status, _ = ops.elbo_step_safe(*args)
    assert int(status[0]) == 0
    assert bool(torch.isfinite(out).all()) and all(bool(torch.isfinite(t).all()) for t in grads.values())
    out2, grads2, status2, _ = ops.elbo_step(*args, jitter=1e-8)
    assert int(status2[0]) == 0 and torch.equal(out.cpu(), out2.cpu())


def test_device_jitter_ladder_on_the_general_path():
    """M = 200 with duplicated inducing points: the blocked factorisation reports the pivot; with md.jitter_ladder > 0
    (what the resident engine sets) the step repeats the factorisation on the device with 1e-8 * 10^i on the diagonal
    (dsp/utils.py:256-269), status[2] names the level, the engine warns lazily, and the result agrees with the step
    launched with that jitter up front -- eagerly and replayed from a HIP graph."""
    from oracle import tgp_oracle as orc
    from tgp.pytorch_amd import ops
    from tgp.pytorch_amd.engine import ElboEngine
    dev = _dev()
    prob = orc.synthetic_problem(700, 4, 200, seed=1, flow="sal2", S=8)
    prob["params"]["Z"][1] = prob["params"]["Z"][0]
    e0 = ElboEngine(prob["X"], prob["Y"], prob["params"], 700.0, flow_blocks=prob["program"], S=8, device=dev, jitter_ladder=0.0)
    e0.elbo()
    torch.cuda.synchronize()
    assert int(e0.status[0]) > 0                              # no ladder: LAPACK-style info of the failing pivot
    p = {k: v.to(dev) for k, v in prob["params"].items()}
    flow = ops.FlowSpec(prob["program"], p["theta"].numel(), 0, dev)
    ref, _, st, _ = ops.elbo_step(prob["X"].to(dev), prob["Y"].to(dev), p["Z"], p["raw_lengthscale"], p["raw_outputscale"],
                                  p["m"], p["Lam"], p["log_var_noise"], 700.0, flow=flow, theta=p["theta"], S=8, jitter=1e-8)
    assert int(st[0]) == 0
    for graph in (False, True):
        e1 = ElboEngine(prob["X"], prob["Y"], prob["params"], 700.0, flow_blocks=prob["program"], S=8, device=dev,
                        jitter_ladder=1e-8)
        if graph:
            e1.capture()
            e1.fp.data.copy_(e0.fp.data)                      # capture() ran one forward/backward; parameters unchanged
        e1.elbo()
        with pytest.warns(ops.NumericalWarning):
            e1.check_status()
        assert int(e1.status[0]) == 0 and int(e1.status[2]) == 1
        assert bool(torch.isfinite(e1.fp.grad).all())
        # K_MM + 1e-8 I has condition ~1e8: two correct factorisations agree to ~1e-8 on the ELBO, not bitwise
        assert rel_err(e1.fp.out[:3].cpu(), ref[:3].cpu()) < 1e-6


def test_big_path_ragged_last_chunk_in_the_small_problem_likelihood_mode():
    """Chunks of 4224 rows at N = 8224: 4224 rows (4 lanes per row in k_ell_flow, 67 partials) and a ragged last chunk
    of 4000 rows (<= 4096: 16 lanes per row, 250 partials) -- the likelihood workspace is sized for either mode."""
    _chunked_cases([(8224, 4, 130, "sal2", 16)], 4224)


@pytest.mark.parametrize("N", [10000, 20000])
def test_big_airline_recipe_sizes_match_oracle(N):
    """BASELINE configs[4]'s model (D=8, M=1000, StepTanhL 5x6, S=32) against the CPU oracle at the reference's Airline
    minibatch size (10 000 rows, main.py:74) and at 20 000 rows -- more than one natural 16 384-row chunk (no test hook)."""
    g = _oracle_case(N, 8, 1000, "tanh5x6", 32, seed=21)
    out, grads, status, _ = run_hip(g)
    assert int(status[0]) == 0 and int(status[1]) == 0
    compare(out, grads, g)
    assert float(torch.triu(grads["Lam"], 1).abs().max()) == 0.0


def _shard_step(N, lo=0, hi=None, mb_global=None, plan=0):
    """One ELBO step of configs[4]'s per-GPU shard shape (rows [lo, hi) of the seeded N-row problem) through the C ABI."""
    from tgp.pytorch_amd import ops
    from tgp.pytorch_amd.synthetic import synthetic_problem
    dev = _dev()
    prob = synthetic_problem(N, 8, 1000, seed=0, flow="tanh5x6", S=32)
    hi = N if hi is None else hi
    p = {k: t.to(dev) for k, t in prob["params"].items()}
    flow = ops.FlowSpec(prob["program"], p["theta"].numel(), 0, dev)
    out, g, st, _ = ops.elbo_step(prob["X"][lo:hi].to(dev), prob["Y"][lo:hi].to(dev), p["Z"], p["raw_lengthscale"],
                                  p["raw_outputscale"], p["m"], p["Lam"], p["log_var_noise"], float(N), flow=flow,
                                  theta=p["theta"], S=32, kl_scale=1.0 if mb_global is None else 0.5, mb_global=mb_global,
                                  plan=plan)
    torch.cuda.synchronize()
    assert int(st[0]) == 0 and int(st[1]) == 0
    return out.cpu(), {k: t.cpu() for k, t in g.items()}


@pytest.mark.parametrize("N,D,flow,S", [(6100, 3, "sal2", 16), (6100, 13, None, 8)])
def test_big_m1000_other_input_dimensions(N, D, flow, S):
    """M = 1000 (the chunk products on 128 x 128 tiles) at the narrowest and the widest augmented-coordinate block (D = 3: DP = 4;
    D = 13: DP = 16) -- the airline shape above is D = 8 -- against the oracle."""
    # (1000 inducing points in 3 dimensions at the generator's lengthscale of 2 are numerically singular: 0.5 there; even so
    #  cond(K_MM) ~ 1e10, and two correct float64 evaluations of dELBO/dZ agree to ~4e-7)
    g = _oracle_case(N, D, 1000, flow, S, seed=4, lengthscale=0.5 if D == 3 else None)
    out, grads, status, _ = run_hip(g)
    assert int(status[0]) == 0 and int(status[1]) == 0
    compare(out, grads, g, tol_grad=1e-6 if D == 3 else TOL_GRAD)


def test_big_minibatch_step_is_bit_reproducible():
    """10 000 rows at M = 1000 (the reference's Airline minibatch): split-K slabs, the two-stream chunk pipeline and the
    riding inverse all reduce in a fixed order -- four steps, the same bits."""
    o1, g1 = _shard_step(10000)
    for _ in range(3):
        o2, g2 = _shard_step(10000)
        assert torch.equal(o1, o2)
        for k in g1:
            assert torch.equal(g1[k], g2[k]), k


def test_big_full_shard_properties():
    """The real per-GPU shard of BASELINE configs[4]: N = 250 000 rows, D = 8, M = 1000, StepTanhL 5x6, S = 32 (16 natural
    row chunks).  Too large for the CPU oracle, so the size-independent properties: bit reproducibility; shard additivity
    (the whole shard == its two 125 000-row halves summed, each with KL weight 1/2 -- what two ranks would all-reduce);
    natural chunking == chunking forced to 8 192 rows (tgp_model.plan)."""
    N = 250000
    out, grads = _shard_step(N)
    out2, grads2 = _shard_step(N)
    assert torch.equal(out, out2)
    for k in grads:
        assert torch.equal(grads[k], grads2[k]), k
    assert bool(torch.isfinite(out).all()) and float(torch.triu(grads["Lam"], 1).abs().max()) == 0.0
    # two half shards (ranks 0 and 1 of a 2-rank job): ELL adds, KL is replicated, gradients add with KL weight 1/2 each
    oa, ga = _shard_step(N, 0, N // 2, mb_global=N)
    ob, gb = _shard_step(N, N // 2, N, mb_global=N)
    assert rel_err(oa[1] + ob[1], out[1]) < 1e-9 and rel_err(oa[2], out[2]) < 1e-12
    for k in grads:
        assert rel_err(ga[k] + gb[k], grads[k]) < 1e-8, (k, rel_err(ga[k] + gb[k], grads[k]))
    # forced chunking: 31 chunks of 8 192 rows instead of 16 of 15 744 (tgp_model.plan)
    from tgp.pytorch_amd import lib
    fo, fg = _shard_step(N, plan=lib.plan_chunk_rows(8192))
    forced = {"out": fo, "grads": fg}
    assert rel_err(forced["out"][:3], out[:3]) < 1e-10
    for k in grads:
        assert rel_err(forced["grads"][k], grads[k]) < 1e-8, (k, rel_err(forced["grads"][k], grads[k]))
