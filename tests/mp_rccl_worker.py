"""Worker of tests/test_gpu_multirank.py::test_engine_graphs_with_a_live_rccl_group (not collected by pytest).

RCCL refuses two ranks on one device, so the one GPU of the test box can host a process group of ONE rank only.  That is
enough to run the part of the multi-rank path the gloo rehearsal cannot: the engine's graph captures and replays with an
NCCL (= RCCL) process group alive in the process -- its watchdog thread, its collectives enqueued between the two graphs
(`split`) and recorded inside the graph (`full`, the --capture-allreduce placement).  The engine is told world_size = 2
(rank 0's half of the rows), so it takes the multi-rank code path; the group it reduces over has one member, so the
collective is the identity and eager steps, split-graph replays and captured-collective replays must agree bit for bit.

    python tests/mp_rccl_worker.py <out.json>
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[2] if len(sys.argv) > 2 else "29611")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from tgp.pytorch_amd import synthetic
    from tgp.pytorch_amd.engine import ElboEngine, shard_rows

    N, D, M, S = 1500, 4, 40, 16
    prob = synthetic.synthetic_problem(N, D, M, seed=3, flow="tanh3x2", S=S)
    lo, hi = shard_rows(N, 2, 0)

    def make(collective="torch"):
        return ElboEngine(prob["X"][lo:hi], prob["Y"][lo:hi], prob["params"], N_total=float(N), flow_blocks=prob["program"],
                          S=S, device=dev, world_size=2, rank=0, mb_global=N, collective=collective)

    def run(mode):
        eng = make()
        # collectives the watchdog thread has not reaped yet when the capture starts: the situation of a training loop that
        # captures right after its eager warm-up steps (with torch's default capture mode this kills the process)
        junk = torch.ones(1024, dtype=torch.float64, device=dev)
        for _ in range(200):
            torch.distributed.all_reduce(junk)
        if mode == "split":
            eng.capture()
        elif mode == "full":
            eng.capture(with_allreduce=True)
        hist = []
        for _ in range(6):
            (eng.step if mode == "eager" else eng.replay)()
            hist.append(list(eng.scalars()))
        eng.check_status()
        torch.cuda.synchronize()
        return eng.graph, hist, eng.fp.data.clone()

    # The default of a multi-rank engine over RCCL: build the ABI communicator, check it against torch.distributed on a
    # seeded buffer, promote it -- then ONE graph with the collective inside, U steps per launch.  The engine weights KL
    # with world_size = 2 while the group has one member, so the reduced scalars are [ELL - KL/2, ELL, KL/2]: a history
    # logged from un-reduced slots would differ between replay_many and step-by-step replays.
    eng = make(None)
    info = dict(eng.collective_info)
    promoted = eng.comm is not None
    for _ in range(3):
        eng.step()
    eng.capture(unroll=4)
    pg_graph, pg_unroll = eng.graph, eng.unroll
    hm = torch.zeros(9, 3, dtype=torch.float64, device=dev)
    eng.replay_many(9, hm)
    torch.cuda.synchronize()
    eng2 = make(None)
    for _ in range(3):
        eng2.step()
    eng2.capture(unroll=1)
    h1 = []
    for _ in range(9):
        eng2.replay()
        h1.append(list(eng2.scalars()))
    many_equals_single = bool(torch.equal(hm.cpu(), torch.tensor(h1, dtype=torch.float64))) and bool(torch.equal(eng.fp.data, eng2.fp.data))
    kl_halved = abs(h1[-1][0] - (h1[-1][1] - h1[-1][2])) < 1e-9 * abs(h1[-1][0])
    eng.close(); eng2.close()

    g_e, h_e, p_e = run("eager")
    g_s, h_s, p_s = run("split")
    g_f, h_f, p_f = run("full")
    res = {"backend": torch.distributed.get_backend(), "world": torch.distributed.get_world_size(),
           "rccl": list(torch.cuda.nccl.version()), "graphs": [g_e, g_s, g_f],
           "split_equals_eager": h_s == h_e and bool(torch.equal(p_s, p_e)),
           "full_equals_eager": h_f == h_e and bool(torch.equal(p_f, p_e)), "history": h_e,
           "auto": {"info": info, "promoted": promoted, "graph": pg_graph, "unroll": pg_unroll,
                    "many_equals_single": many_equals_single, "elbo_is_ell_minus_reduced_kl": kl_halved}}
    with open(out_path, "w") as fh:
        json.dump(res, fh)
    torch.distributed.destroy_process_group()
    print("RCCL_WORKER_OK")


if __name__ == "__main__":
    main()
