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


def _bootstrap_worker(rank, world, port, ret):
    """RcclComm's guarded bootstrap over gloo: rank 1's local step fails (a library that refuses tgp_comm_load); BOTH
    ranks must raise RcclUnavailable -- nobody is left in the id broadcast or in ncclCommInitRank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from tgp.pytorch_amd import engine, lib as L

    class FakeLib:
        def tgp_comm_load(self, path):
            return -104 if rank == 1 else 0

        def tgp_comm_unique_id(self, p):
            return 0

        def tgp_last_error(self):
            return b"no RCCL here"

        def tgp_comm_init(self, *a):
            raise AssertionError("tgp_comm_init must not be reached when a rank failed its bootstrap")
    real = L.load
    L.load = lambda: FakeLib()
    try:
        engine.RcclComm(world, rank, None, timeout_s=20)
        ret[rank] = "built"
    except engine.RcclUnavailable as e:
        ret[rank] = "unavailable: " + str(e)
    finally:
        L.load = real
    torch.distributed.destroy_process_group()


def test_rccl_bootstrap_fails_on_every_rank_together():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bootstrap_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret[0].startswith("unavailable") and "another rank" in ret[0], ret[0]
    assert ret[1].startswith("unavailable") and "this rank" in ret[1], ret[1]


def _rendezvous_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from tgp.pytorch_amd import engine
    import time
    t0 = time.time()
    try:
        if rank == 0:
            engine._store_rendezvous("tgp_test_rendezvous/late", world, 1.0)     # rank 1 never arrives at this tag
            ret[rank] = "passed"
        else:
            ret[rank] = "absent"
    except TimeoutError as e:
        ret[rank] = "timeout after %.1f s: %s" % (time.time() - t0, e)
    engine._store_rendezvous("tgp_test_rendezvous/both", world, 30.0)            # both arrive: returns
    torch.distributed.destroy_process_group()


def test_store_rendezvous_times_out_instead_of_hanging():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rendezvous_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret[0].startswith("timeout") and "1 of 2 ranks" in ret[0], ret[0]
    assert ret[1] == "absent"
