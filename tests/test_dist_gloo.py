"""CPU, world_size = 2 over gloo: the data-parallel contract of the step -- rows are sharded,
every rank holds KL-weighted partial gradients, ONE all-reduce of the flat buffer gives every rank the full
gradient and the full (ELBO, ELL, KL).  The per-shard numbers come from the algebra model (the checker), the code
under test is engine.allreduce_flat / engine.shard_rows, i.e. exactly what runs over RCCL on the GPUs."""
import os
import socket

import torch
import torch.multiprocessing as mp

import algebra_model as am
from conftest import load_golden, rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from tgp.pytorch_amd.engine import allreduce_flat, shard_rows
    g = load_golden("med_sal2")
    p = g["params"]
    N = g["X"].shape[0]
    lo, hi = shard_rows(N, world, rank)
    st = am.prepare(p)
    # a shard sees N_total / MB_global as its scale: pass N_total * (rows in shard) / N to the single-shard model
    rs = am.rows(g["X"][lo:hi], g["Y"][lo:hi], st, p, float(g["N_total"]) * (hi - lo) / N, g["program"], g["xs"], g["ws"])
    grads = am.backward_mm(st, rs, p, kl_scale=1.0 / world)
    keys = ["Z", "raw_lengthscale", "raw_outputscale", "m", "Lam", "log_var_noise", "theta"]
    flat = torch.cat([grads[k].reshape(-1) for k in keys])
    n = flat.numel()
    buf = torch.cat([flat, torch.stack([rs["ell"] - st["kl"], rs["ell"], st["kl"], torch.zeros(())])])
    allreduce_flat(buf, n, world)
    if rank == 0:
        ret["flat"], ret["out"], ret["sizes"] = buf[:n].clone(), buf[n:].clone(), [grads[k].numel() for k in keys]
    torch.distributed.destroy_process_group()


def test_two_rank_allreduce_reproduces_the_unsharded_step():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    g = load_golden("med_sal2")
    names = ["g_Z", "g_raw_lengthscale", "g_raw_outputscale", "g_m", "g_Lam", "g_log_var_noise", "g_theta"]
    ref = torch.cat([g[k].reshape(-1) for k in names])
    assert rel_err(ret["flat"], ref) < 1e-9
    assert rel_err(ret["out"][0], g["ELBO"]) < 1e-10 and rel_err(ret["out"][1], g["ELL"]) < 1e-10
    assert rel_err(ret["out"][2], g["KLD"]) < 1e-12


def test_shard_rows_partition():
    from tgp.pytorch_amd.engine import shard_rows
    for N, W in ((8611, 8), (10, 3), (7, 8)):
        spans = [shard_rows(N, W, r) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == N
        assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
