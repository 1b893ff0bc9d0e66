"""Worker of tests/test_gpu_multirank.py (not collected by pytest): one rank of a world of W processes that SHARE the
one GPU of the test box.  Backend gloo (RCCL refuses two ranks on one device); everything else is the multi-GPU code
path as it runs over RCCL: engine.shard_rows -> ElboEngine(world_size=W, rank=r) -> capture() into the two graphs
around the collective -> replay().  Rank 0 also runs the unsharded engine and writes the comparison as JSON.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node W --master-addr 127.0.0.1 --master-port P \
        tests/mp_engine_worker.py <out.json> <workload> <steps> [graph|eager]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, workload, steps, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from tgp.pytorch_amd import ops, synthetic
    from tgp.pytorch_amd.engine import ElboEngine, shard_rows

    shapes = {"tanh3x2_small": (1500, 4, 40, 16, "tanh3x2"), "sal2_power": (8611, 4, 100, 32, "sal2"),
              "svgp_big": (700, 5, 150, 8, None), "idsal3": (900, 4, 30, 12, "idsal3")}
    N, D, M, S, flow = shapes[workload]
    prob = synthetic.synthetic_problem(N, D, M, seed=3, flow=flow, S=S)
    mlp = None
    if flow == "idsal3":
        spec = ops.MlpSpec(D, 50, 2, 6, act="relu", drop_p=0.25, seed=0)
        g = torch.Generator().manual_seed(11)
        W = 0.3 * (2 * torch.rand(6 * spec.weights_per_net, generator=g, dtype=torch.float64) - 1)
        mlp = (spec, W)
        prob["rowp"] = None
    lo, hi = shard_rows(N, world, rank)

    def make(world_size, r, rows):
        kw = {}
        if mlp is not None:
            kw = dict(mlp=mlp[0], mlp_weights=mlp[1].clone(), mlp_training=False)
        return ElboEngine(prob["X"][rows], prob["Y"][rows], prob["params"], N_total=float(N), flow_blocks=prob["program"],
                          S=S, device=dev, world_size=world_size, rank=r, mb_global=N, **kw)

    eng = make(world, rank, slice(lo, hi))
    hist = []
    if mode == "graph":
        eng.capture()
        assert eng.graph == "split", eng.graph
    for _ in range(steps):
        (eng.replay if mode == "graph" else eng.step)()
        hist.append(list(eng.scalars()))
    eng.check_status()
    torch.cuda.synchronize()
    # every rank must hold the same parameters after the steps (replicated Adam on the reduced gradient)
    mine = eng.fp.data.clone()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    torch.distributed.all_gather(gathered, mine)
    result = None
    if rank == 0:
        ref = make(1, 0, slice(0, N))
        rh = []
        for _ in range(steps):
            ref.step()
            rh.append(list(ref.scalars()))
        ref.check_status()

        def rel(a, b):
            return float((a - b).abs().max() / (b.abs().max() + 1e-300))
        result = {
            "world": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(), "graph": eng.graph,
            "hist_rel": rel(torch.tensor(hist, dtype=torch.float64), torch.tensor(rh, dtype=torch.float64)),
            "param_rel": rel(eng.fp.data, ref.fp.data),
            "ranks_identical": all(bool(torch.equal(gathered[0], t)) for t in gathered[1:]),
            "history": hist,
        }
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(result, fh)
        print("MULTIRANK_OK", json.dumps({k: result[k] for k in ("world", "backend", "graph", "hist_rel", "param_rel", "ranks_identical")}))


if __name__ == "__main__":
    main()
