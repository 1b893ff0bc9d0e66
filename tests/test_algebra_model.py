"""CPU: the restructured algebra the HIP kernels implement == the oracle (reference-shaped) path,
values and every gradient, on the golden fixtures; plus row-shard additivity."""
import pytest
import torch

import algebra_model as am
from conftest import load_golden, rel_err
from oracle import tgp_oracle as orc

CASES = ["tiny_svgp", "tiny_sal2", "tiny_tanh3x2", "tiny_idsal3", "ragged_sal2", "boston_like_svgp", "med_sal2",
         "med_tanh3x2", "init_sal2_identity"]


@pytest.mark.parametrize("name", CASES)
def test_algebra_matches_reference_fixture(name):
    g = load_golden(name)
    (elbo, ell, kld), grads, rs = am.elbo_and_grads(g["X"], g["Y"], g["params"], float(g["N_total"]), g["program"],
                                                    g["xs"], g["ws"], g.get("rowp"))
    assert rel_err(elbo, g["ELBO"]) < 1e-9
    assert rel_err(kld, g["KLD"]) < 1e-12
    assert rel_err(rs["mu"], g["mu"]) < 1e-8
    assert float((rs["v"] - g["v"]).abs().max()) < 1e-8 * float(g["v"].abs().max())
    for key in ("Z", "m", "Lam", "raw_outputscale", "raw_lengthscale", "log_var_noise"):
        assert rel_err(grads[key], g["g_" + key]) < 2e-7, key     # cond(K_MM) ~ 1e7 amplifies eps in both codes
    if g["program"] is not None:
        assert rel_err(grads["theta"], g["g_theta"]) < 1e-8
    if "rowp" in g:
        assert rel_err(grads["rowp"], g["g_rowp"]) < 1e-8


def test_row_shard_additivity():
    """Row statistics are plain sums over rows: 8 shards summed == unsharded (the multi-GPU contract)."""
    g = load_golden("med_sal2")
    p = g["params"]
    st = am.prepare(p)
    full = am.rows(g["X"], g["Y"], st, p, float(g["N_total"]), g["program"], g["xs"], g["ws"])
    N = g["X"].shape[0]
    keys = ["ell", "etab", "thetab", "G", "wb", "s2b_direct", "T0", "T1", "T2"]
    acc = {k: 0.0 for k in keys}
    for r in range(8):
        sl = slice(r * N // 8, (r + 1) * N // 8)
        # each shard sees MB = its own rows but must scale by N_total / global MB:
        part = am.rows(g["X"][sl], g["Y"][sl], st, p, float(g["N_total"]) * (sl.stop - sl.start) / N, g["program"],
                       g["xs"], g["ws"])
        for k in keys:
            acc[k] = acc[k] + part[k]
    for k in keys:
        assert rel_err(acc[k], full[k]) < 1e-11, k
    # KL gradient counted once when every rank applies kl_scale = 1/world
    gfull = am.backward_mm(st, full, p)
    gsum = None
    for r in range(8):
        sl = slice(r * N // 8, (r + 1) * N // 8)
        part = am.rows(g["X"][sl], g["Y"][sl], st, p, float(g["N_total"]) * (sl.stop - sl.start) / N, g["program"],
                       g["xs"], g["ws"])
        gr = am.backward_mm(st, part, p, kl_scale=1.0 / 8)
        gsum = gr if gsum is None else {k: gsum[k] + gr[k] for k in gr}
    for k in gfull:
        assert rel_err(gsum[k], gfull[k]) < 1e-9, k
