#!/usr/bin/env python
"""bench.py -- ELBO-steps/sec of the TGP sparse-variational hot path on MI355X.

    python bench.py --gpus 1 --steps 2000 --warmup 100
    python bench.py --gpus N --steps K --warmup W        (N > 1 and no RANK in the environment: bench.py starts its own N
                                                          ranks -- a fresh `python -m torch.distributed.run` child, created
                                                          BEFORE this process touches the GPU; it never execs)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W           (the same thing with the launcher on the outside)

One "step" = forward ELBO + backward + Adam update on one full Power-sized minibatch (reference semantics,
code/dsp/trainers/trainer_base.py:337-342), float64 like the reference's main.py.  Default workload =
BASELINE.json configs[2]: TGP on Power (N=8611 training rows, D=4), M=100, 3-block tanh flow (StepTanhL 3x2),
S=32 Gauss-Hermite nodes; synthetic seeded data of that shape, data resident in HBM.
Multi-GPU (one process per GPU, torch.distributed over RCCL): rows are sharded, ONE all-reduce of the flat
[gradient | ELBO, ELL, KL] buffer per step, replicated Adam.  `--scaling strong` (default for the Power / Boston
workloads): the SAME problem's rows are split W ways -- north_star's "Power at 1/2/4/8 MI355X"; `value` = steps of the
whole 8611-row problem per second (at this size it cannot scale: the replicated M x M work and the collective do not
shrink, and the line says so by its number).  `--scaling weak` (default for the airline workloads, whose per-GPU shard
is the workload): every rank owns a workload-sized row shard of a W-times larger minibatch, `value` counts
workload-sized shard-steps per second over the whole job, and the metric string names the W x N-row minibatch it
steps.  The documented BASELINE configs[4] line is
    python bench.py --gpus 8 --workload tgp_airline_tanh5x6 --steps 20 --warmup 3     (N = 2 M rows, 250 k per GPU, weak)
`--capture-allreduce` records the collective inside the HIP graph instead of between two graphs.

Timing: after W warm-up steps the K steps are timed `--repeats` times (default 5), each repeat bracketed by a barrier +
synchronize and reduced with MAX over ranks; the line reports the MEDIAN repeat (`ms_per_step`, `value`) and the
spread (`ms_per_step_min`, `ms_per_step_max`).

Extra objects on the JSON line (rank 0):
  roofline      dominant kernel = the fused row kernel k_rows; achieved = algorithmic FLOP per launch
                 / its average duration measured with HIP events on the launch stream.
  cpu_baseline  the oracle (CPU restatement of the reference's op sequence, eager PyTorch float64 + torch Adam)
                timed on this host's cores on a bounded number of steps of the same workload (N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense FP64 matrix peak (AMD datasheet; the microarch guide lists no f64 row)

WORKLOADS = {
    # name: (N, D, M, S, flow, blocks B, flop-equivalents per block-node c)   [SURVEY.md 8(d)]
    "tgp_power_tanh3x2": dict(N=8611, D=4, M=100, S=32, flow="tanh3x2", B=3, c=20),   # BASELINE.json configs[2]
    "tgp_power_sal2": dict(N=8611, D=4, M=100, S=32, flow="sal2", B=2, c=20),          # reference default for Power TGP
    "svgp_power": dict(N=8611, D=4, M=100, S=32, flow=None, B=0, c=0),                 # BASELINE.json configs[1]
    "svgp_boston": dict(N=455, D=13, M=5, S=32, flow=None, B=0, c=0),                  # BASELINE.json configs[0]
    # BASELINE.json configs[3]: input-dependent SAL x 3, a_n, b_n from 6 MLPs 4 -> 50 -> 50 -> 1 (relu, dropout 0.25)
    "idtgp_power_sal3": dict(N=8611, D=4, M=100, S=32, flow="idsal3", B=3, c=20, mlp=dict(H=50, L=2, p=0.25)),
    # BASELINE.json configs[4], one GPU's shard of the 2M-row full batch (8 x 250k), reference's airline flow 5x6
    "tgp_airline_tanh5x6": dict(N=250000, D=8, M=1000, S=32, flow="tanh5x6", B=5, c=52),
    "tgp_airline_mb10k": dict(N=10000, D=8, M=1000, S=32, flow="tanh5x6", B=5, c=52),    # C5b: minibatch 10k rows
    # one GPU's 1/8 of that minibatch (SURVEY 8d, C5b "1250 rows/GPU"): what a rank of the 8-GPU run computes per step --
    # the replicated M x M phases and 1 250 rows (run it with `--gpus 8 --scaling strong --workload tgp_airline_mb10k`
    # on a node; this single-GPU workload shows the per-rank step without the collective)
    "tgp_airline_mb10k_rank8": dict(N=1250, D=8, M=1000, S=32, flow="tanh5x6", B=5, c=52),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


BASELINE_METRIC = "ELBO-steps/sec (N x M kernel + chol + flow), Power M=100 S=32"


def metric_name(workload, w, n_global):
    """BASELINE.json's metric string ONLY for the configuration it is quoted on -- TGP on Power, the whole 8611-row problem
    per step (one GPU, or its rows split over the ranks); anything else says what it steps."""
    if workload == "tgp_power_tanh3x2" and n_global == w["N"]:
        return BASELINE_METRIC
    if n_global == w["N"]:
        return "ELBO-steps/sec (N x M kernel + chol + flow), workload %s" % workload
    # weak scaling: W workload-sized row shards of one W x N-row minibatch per step; `value` counts shard-steps
    # (tgp_airline_tanh5x6 on 8 GPUs = BASELINE configs[4]: 8 x 250 000 = the 2 M-row full batch)
    return ("ELBO shard-steps/sec (N x M kernel + chol + flow), workload %s: %d shards of %d rows = one %d-row "
            "minibatch per step" % (workload, n_global // w["N"], w["N"], n_global))


def rows_kernel_flops(w):
    """Algorithmic FLOP of one k_rows launch: the row-dependent terms of SURVEY.md 8(d), forward x3."""
    N, D, M, S = w["N"], w["D"], w["M"], w["S"]
    fwd = N * M * (3 * D + 1) + 2 * M * M * N + 6 * M * N + S * N * w["B"] * w["c"]
    return 3.0 * fwd


# ---- the builder's PREDICTION of a multi-GPU line (VERDICT r4 #5b): per-kernel times measured on ONE MI355X (profiles/r06_*,
# microseconds) + a stated estimate of the collective, so that the first SCALE record can be judged against a number.
# Every multi-GPU entry is UNMEASURED ON HARDWARE: no multi-GPU node was available to the builder in any round.
EXPECT = {
    # fused path (M <= 128): prepare + [rows(N_rank)] + slab reduction + the M x M adjoint launch; with more than one rank the
    # Adam update is a launch of its own behind the collective (one rank: inside the adjoint launch)
    # rows: in-step duration of the row kernel the library picks at that many rows per rank (k_rows<..,10> at 8611, k_rows4 with
    # 8-wave workgroups at 4306, 4-wave below; profiles/r06_rows_kernel_time.txt back-to-back + 1.3 us in-step: the kernel
    # statistics of the 1-GPU step); bwd: k_bwd without the update in it
    "tgp_power_tanh3x2": dict(prep=24.3, rows={8611: 43.6, 4306: 41.5, 2153: 33.4, 1077: 33.2}, reduce=4.9, bwd=20.2, bwd_w=19.4, adam=4.6),
    "tgp_power_sal2": dict(prep=24.3, rows={8611: 40.2, 4306: 37.9, 2153: 30.9, 1077: 30.7}, reduce=4.9, bwd=20.2, bwd_w=19.4, adam=4.6),
    "svgp_power": dict(prep=24.3, rows={8611: 33.8, 4306: 32.3, 2153: 26.2, 1077: 26.1}, reduce=4.5, bwd=20.2, bwd_w=19.4, adam=4.6),
    # general-M path: the single-GPU step of the workload (weak scaling: every rank runs it on its own shard) and, for the
    # minibatch split 8 ways, the measured per-rank share (tgp_airline_mb10k_rank8)
    "tgp_airline_tanh5x6": dict(ms=30.7),
    "tgp_airline_mb10k": dict(ms=2.15, strong_ms={8: 1.117}),
}


def allreduce_estimate_us(world, doubles):
    """RCCL all-reduce of `doubles` float64 over xGMI inside the captured step (tgp_allreduce_f64 on the compute stream).
    ESTIMATE, not a measurement: a latency term (ring / tree hops of ~4 us each: 2 (W - 1) steps) plus the ring's wire time
    2 (W - 1) / W x bytes at 0.7 x 153 GB/s per link."""
    if world <= 1:
        return 0.0
    hops = 2 * (world - 1)
    return 8.0 + 2.0 * hops + (2.0 * (world - 1) / world) * doubles * 8 / (0.7 * 153e9) * 1e6


def expected_line(workload, w, world, scaling, n_doubles, measured_ms_1gpu=None):
    """config.expected: the predicted ms/step and value of this (workload, world, scaling)."""
    e = EXPECT.get(workload)
    ar = allreduce_estimate_us(world, n_doubles)
    if e is not None and "rows" in e and scaling == "strong":
        nr = -(-w["N"] // world)
        key = min(e["rows"], key=lambda k: abs(k - nr))
        bwd = e["bwd"] if world == 1 else e["bwd_w"]
        us = e["prep"] + e["rows"][key] + e["reduce"] + bwd + (e["adam"] if world > 1 else 0.0) + ar
        basis = ("k_prep_a %.1f + row kernel at %d rows/rank %.1f + k_reduce %.1f + k_bwd %.1f%s + all-reduce of %d doubles "
                 "%.1f us (ESTIMATE: 8 us + 2 us per ring step + wire time; unmeasured on hardware)"
                 % (e["prep"], nr, e["rows"][key], e["reduce"], bwd, " + k_adam_dev %.1f" % e["adam"] if world > 1 else "",
                    n_doubles, ar))
        return {"ms_per_step": us * 1e-3, "value": 1e6 / us, "basis": basis, "measured_on": "1 x MI355X per-kernel times, profiles/r06_*"}
    if measured_ms_1gpu is None and e is not None and "ms" in e:
        measured_ms_1gpu = e.get("strong_ms", {}).get(world) if scaling == "strong" and world > 1 else e["ms"]
    if measured_ms_1gpu is not None:
        us = measured_ms_1gpu * 1e3 + ar
        units = world if scaling == "weak" else 1
        return {"ms_per_step": us * 1e-3, "value": units * 1e6 / us,
                "basis": "single-rank step of this rank's rows %.3f ms + all-reduce of %d doubles %.1f us (ESTIMATE, unmeasured on "
                         "hardware)" % (measured_ms_1gpu, n_doubles, ar),
                "measured_on": "1 x MI355X"}
    return None


def make_problem(w, seed):
    from tgp.pytorch_amd.synthetic import synthetic_problem
    return synthetic_problem(w["N"], w["D"], w["M"], seed=seed, flow=w["flow"], S=w["S"])


def make_mlp(w, seed):
    """ops.MlpSpec + packed weights for the workload's per-row parameter networks: torch.nn.Linear-style uniform
    init, last layer scaled down and biased so that the flow starts near the identity (a = 0, b = 1)."""
    from tgp.pytorch_amd import ops
    nn_ = 2 * w["B"]
    spec = ops.MlpSpec(w["D"], w["mlp"]["H"], w["mlp"]["L"], nn_, act="relu", drop_p=w["mlp"]["p"], seed=seed)
    g = torch.Generator().manual_seed(100 + seed)
    W = torch.empty(nn_, spec.weights_per_net, dtype=torch.float64)
    for k in range(nn_):
        o, nin = 0, spec.D
        for _ in range(spec.L):
            n = spec.H * nin + spec.H
            W[k, o:o + n] = (2 * torch.rand(n, generator=g, dtype=torch.float64) - 1) / math.sqrt(nin)
            o += n
            nin = spec.H
        W[k, o:o + spec.H] = 0.1 * (2 * torch.rand(spec.H, generator=g, dtype=torch.float64) - 1) / math.sqrt(spec.H)
        W[k, o + spec.H] = float(k % 2)        # a nets -> 0, b nets -> 1
    return spec, W.reshape(-1)


def torch_mlps(X, W, spec):
    """The same nets as plain torch ops (eval mode) -- the CPU baseline's version of the per-row parameters."""
    outs, pw = [], spec.weights_per_net
    for k in range(spec.nnets):
        wk = W[k * pw:(k + 1) * pw]
        o, h, nin = 0, X, spec.D
        for _ in range(spec.L):
            h = torch.relu(h @ wk[o:o + spec.H * nin].reshape(spec.H, nin).T + wk[o + spec.H * nin:o + spec.H * nin + spec.H])
            o += spec.H * nin + spec.H
            nin = spec.H
        outs.append(h @ wk[o:o + spec.H] + wk[o + spec.H])
    return torch.stack(outs, 1)


def cpu_baseline(prob, budget_s=15.0, max_steps=400, mlp=None):
    """Oracle step (reference-shaped eager PyTorch-CPU float64 + autograd + torch Adam) on the host cores.
    The thread count is calibrated first (a 256-thread pool is pathological for these small ops: one step took
    38 s); `cores` reports the count actually used."""
    from oracle import tgp_oracle as orc
    leaves = {k: v.clone().requires_grad_(True) for k, v in prob["params"].items()}
    groups = [{"params": list(leaves.values())}]
    Wn = None
    if mlp is not None:
        Wn = mlp[1].clone().requires_grad_(True)
        groups.append({"params": [Wn], "weight_decay": 1e-5})
    opt = torch.optim.Adam(groups, lr=0.01)
    N = prob["X"].shape[0]
    big = leaves["m"].numel() > 128
    ns = min(N, 16384) if big else N      # bounded sample: the reference materialises (N,M) matrices many times over
    X_s, Y_s = prob["X"][:ns], prob["Y"][:ns]

    def one():
        rowp = torch_mlps(X_s, Wn, mlp[0]) if mlp is not None else None
        elbo, _, _ = orc.elbo(X_s, Y_s, leaves["Z"], leaves["raw_lengthscale"], leaves["raw_outputscale"],
                              leaves["m"], leaves["Lam"], leaves["log_var_noise"], prob["N_total"], prob["program"],
                              leaves.get("theta"), prob["xs"], prob["ws"], rowp)
        opt.zero_grad()
        (-elbo).backward()
        opt.step()

    ncpu = os.cpu_count() or 1
    # Thread count: every candidate is timed for >= 1.5 s (a share of the budget) and the best measured RATE is kept --
    # best-of-three single steps flipped the choice between 16 and 32 threads from run to run, and the reported
    # baseline with it (20 vs 33 steps/s on the same host).  All candidates are reported in `sample`.
    cand = sorted({min(8, ncpu), min(32, ncpu), min(64, ncpu)} if big else {1, min(4, ncpu), min(8, ncpu), min(16, ncpu), min(32, ncpu)})
    per = max(1.5, 0.4 * budget_s / len(cand))
    rates = {}
    for nt in cand:
        torch.set_num_threads(nt)
        one()                                     # warm the pool
        t0 = time.perf_counter()
        k = 0
        while True:
            one()
            k += 1
            dtc = time.perf_counter() - t0
            if dtc >= per or (big and k >= 1):
                break
        rates[nt] = k / dtc
        if dtc / k > 3.0:
            break
    best = max(rates, key=rates.get)
    torch.set_num_threads(best)
    one()
    # The reported rate is the MEDIAN of three timed windows at the calibrated thread count (one window moved
    # +-35 % with the host's other tenants from line to line); all three are listed.
    wins, n_tot, t_tot = [], 0, 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        n = 0
        while n < max_steps // 3 + 1 and (time.perf_counter() - t0 < budget_s * 0.2 or n == 0):
            one()
            n += 1
        dtw = time.perf_counter() - t0
        wins.append(n / dtw * ns / N)
        n_tot += n
        t_tot += dtw
    med = sorted(wins)[1]
    cal = ", ".join("%d thr %.1f/s" % (nt, rates[nt] * ns / N) for nt in sorted(rates))
    return {"value": med, "unit": "ELBO-steps/s", "cores": best, "kind": "port", "host_cpus": ncpu,
            "windows": [round(x, 2) for x in wins],
            "sample": "median of 3 timed windows (%s steps/s), %d steps on the first %d of %d rows%s (oracle/tgp_oracle.py, float64, "
                      "torch.optim.Adam, %d threads: the best of the calibrated rates [%s], each timed >= %.1f s), %.1f s"
                      % (", ".join("%.1f" % x for x in wins), n_tot, ns, N, ", rate scaled by rows" if ns < N else "", best, cal, per, t_tot)}


def launch_ranks(n):
    """Parent of a multi-GPU run: `python -m torch.distributed.run` as a child process (one rank per GPU, rendezvous on
    127.0.0.1 at a free port), stdout / stderr inherited.  Returns the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this stack
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: starting %d ranks: %s" % (n, " ".join(cmd[1:])))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="tgp_power_tanh3x2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--repeats", type=int, default=5, help="timed repeats of the K steps; the median is reported")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="default: strong for the Power / Boston workloads (north_star: the same problem on 1/2/4/8 GPUs), "
                         "weak for the airline workloads (a rank's 250 k rows are 1/8 of configs[4])")
    ap.add_argument("--capture-allreduce", action="store_true", help="capture the collective inside the HIP graph")
    ap.add_argument("--collective", default="auto", choices=["auto", "torch", "abi"],
                    help="auto (default): over RCCL with more than one rank the engine checks the C ABI's tgp_allreduce_f64 "
                         "against torch.distributed.all_reduce on a seeded buffer and, when every rank agrees, uses it (RCCL on "
                         "the compute stream, inside ONE captured graph, U steps per launch); otherwise, and with `torch`, "
                         "torch.distributed.all_reduce between two graphs; `abi` skips the check")
    ap.add_argument("--replicas", type=int, default=1, help="(1 GPU only) K independent training runs of the workload on K "
                    "streams, value = their aggregate steps/s: how the chip is filled when several UCI splits train at once")
    ap.add_argument("--allreduce-only", action="store_true", help="time tgp_allreduce_f64 (RCCL on the compute stream, captured in a "
                    "HIP graph) at the two buffer sizes the steps exchange -- 10 540 doubles (Power, fused path) and 1.0 M (C5) -- and "
                    "print one JSON line: the number that replaces config.expected's ESTIMATE of the collective on the first "
                    "multi-GPU lease (with one rank: a 1-rank RCCL group, the launch overhead only)")
    ap.add_argument("--comm-timeout", type=float, default=120.0, help="bound (s) of the communicator bootstrap; a rank that cannot "
                    "complete it ends the whole job with exit code 3")
    ap.add_argument("--plan", type=int, default=0, help="tgp_model.plan of the engine's calls (include/tgp_hip.h TGP_PLAN_*): A/B of "
                    "equivalent kernels, e.g. 1 = the 16-rows-per-wave row kernel, 16 = the general-M chunk pipeline without its two-buffer overlap; 0 = the library's choice")
    ap.add_argument("--traffic-json", default=None, help="per-launch HBM bytes of the dominant kernel from a rocprofv3 "
                    "--pmc pass, keyed by the source hash of the library it was measured on (default: "
                    "profiles/rows_traffic.json); a summary of other sources is rejected and roofline.traffic is null")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "weak" if args.workload.startswith("tgp_airline") else "strong"

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` as the driver runs it: start the N ranks ourselves.  The parent has made NO GPU call
        # yet (importing torch does not initialise HIP) and never execs: a fresh child runs the launcher, every rank
        # re-enters this file with RANK / LOCAL_RANK / WORLD_SIZE set, rank 0 prints the JSON line on the inherited
        # stdout, and the parent exits with the child's code.
        return launch_ranks(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    # TGP_BENCH_BACKEND=gloo is a rehearsal hook: it lets the multi-rank control flow (shards, split graphs, the
    # all-reduce of the flat buffer, max-over-ranks timing) run with several ranks on ONE GPU, where RCCL refuses
    # duplicate devices.  The measured configuration is always nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("TGP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit("bench.py: %d ranks but %d visible GPUs (RCCL needs one device per rank; TGP_BENCH_BACKEND=gloo "
                         "rehearses the multi-rank control flow on fewer)" % (world, ndev))
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)

    # Everything from here to the JSON line can fail on ONE rank only (a communicator bootstrap that times out, RCCL missing, a
    # hand-off timeout): such a rank must not leave the others waiting in a collective.  It says why, tears its process group
    # down in a bounded way and leaves with code 3 through a fresh exit (nothing re-execs); the launcher (torch.distributed.run)
    # then ends the remaining ranks, so every rank of a failed job exits non-zero.
    try:
        return run_bench(args, world, rank, dev, backend)
    except SystemExit:
        raise
    except BaseException as e:                     # TimeoutError, RcclUnavailable, HandoffTimeoutError, TgpError, ...
        fail_all_ranks(e, rank, world)


def fail_all_ranks(exc, rank, world):
    import threading
    import traceback
    log("bench.py: rank %d of %d failed: %s: %s" % (rank, world, type(exc).__name__, exc))
    log("".join(traceback.format_exception(type(exc), exc, exc.__traceback__)[-6:]))
    if torch.distributed.is_initialized():
        th = threading.Thread(target=torch.distributed.destroy_process_group, daemon=True)
        th.start()
        th.join(10.0)                              # (a peer stuck in ncclCommInitRank can hold the teardown: do not wait for it)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(3)


def allreduce_only(args, world, rank, dev):
    """`--allreduce-only`: tgp_allreduce_f64 of the two exchange sizes, U = 20 collectives per captured graph, median of 9 replays."""
    from tgp.pytorch_amd.engine import RcclComm, CAPTURE_MODE
    cw = torch.distributed.get_world_size() if world > 1 else 1
    comm = RcclComm(cw, rank, None, timeout_s=args.comm_timeout)
    U, out = 20, []
    for n in (10540, 1000 * 1000 + 8000 + 1000 + 64):       # [grads | ELBO, ELL, KL]: Power tanh3x2; C5 (Lam 1e6 + Z 8e3 + m 1e3 + ...)
        buf = torch.ones(n, dtype=torch.float64, device=dev)
        for _ in range(3):
            comm.allreduce(buf)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
            for _ in range(U):
                comm.allreduce(buf)
                buf.mul_(1.0 / cw)                 # keeps the values finite; one tiny launch between two collectives, as a step has
        g.replay()
        ts = []
        for _ in range(9):
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / U * 1e6)
        us = sorted(ts)[len(ts) // 2]
        if world > 1:
            t = torch.tensor([us], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            us = float(t[0])
        out.append({"doubles": n, "us_per_allreduce": us, "estimate_us": allreduce_estimate_us(world, n),
                    "ratio_to_estimate": (us / allreduce_estimate_us(world, n)) if world > 1 else None})
    comm.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "tgp_allreduce_f64 in a captured graph, us per all-reduce (+ one scaling launch)", "n_gpus": world,
                          "unit": "us", "higher_is_better": False, "sizes": out,
                          "note": "the 8 us + 2 us per ring step + wire-time ESTIMATE behind config.expected is UNMEASURED ON HARDWARE "
                                  "for more than one rank until this line is produced on a multi-GPU node"}), flush=True)
    return 0


def run_bench(args, world, rank, dev, backend):
    from tgp.pytorch_amd.engine import ElboEngine

    from tgp.pytorch_amd.engine import shard_rows

    if args.allreduce_only:
        return allreduce_only(args, world, rank, dev)
    w = WORKLOADS[args.workload]
    mlp = make_mlp(w, seed=0) if "mlp" in w else None       # same networks on every rank
    if args.scaling == "weak":
        prob = make_problem(w, seed=rank)          # every rank: its own workload-sized shard, same parameters (seed 0)
        params = make_problem(w, seed=0)["params"] if rank else prob["params"]
        Xr, Yr, n_global = prob["X"], prob["Y"], w["N"] * world
    else:
        prob = make_problem(w, seed=0)             # one problem, rows split W ways
        params = prob["params"]
        lo, hi = shard_rows(w["N"], world, rank)
        Xr, Yr, n_global = prob["X"][lo:hi], prob["Y"][lo:hi], w["N"]
    eng = ElboEngine(Xr, Yr, params, N_total=float(n_global), flow_blocks=prob["program"],
                     S=w["S"], device=dev, world_size=world, rank=rank, mb_global=n_global,
                     mlp=mlp[0] if mlp else None, mlp_weights=mlp[1] if mlp else None,
                     collective=(None if args.collective == "auto" else args.collective) if world > 1 else "torch",
                     comm_timeout_s=args.comm_timeout, plan=args.plan)
    log("collective: %s" % json.dumps(eng.collective_info))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # independent replicas (other seeds, own parameters / workspaces / graphs / streams); replica 0 is `eng`
    extra = []
    if args.replicas > 1 and world == 1:
        from tgp.pytorch_amd import ops
        for k in range(1, args.replicas):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                pk = make_problem(w, seed=k)
                ops._ws_cache.clear()
                ek = ElboEngine(pk["X"], pk["Y"], pk["params"], N_total=float(w["N"]), flow_blocks=pk["program"], S=w["S"],
                                device=dev, mlp=mlp[0] if mlp else None, mlp_weights=mlp[1].clone() if mlp else None)
                for _ in range(3):
                    ek.step()
                if not args.no_graph:
                    ek.capture()
            extra.append((ek, st))
        torch.cuda.synchronize()

    # ---- warm-up (eager), capture, then the timed region --------------------------------------------------------
    run = eng.step
    for _ in range(min(args.warmup, 10)):
        eng.step()
    eng.check_status()
    if not args.no_graph:
        eng.capture(with_allreduce=True if (args.capture_allreduce and world > 1) else None)
        run = eng.replay
    if extra:
        run0 = run

        def run():
            run0()
            for ek, st in extra:
                with torch.cuda.stream(st):
                    (ek.step if args.no_graph else ek.replay)()
    # K steps: single-rank graph runs go through the engine's unrolled graph (U steps per graph launch, the remainder one
    # by one -- engine.replay_many; `config.launch` names U); everything else step by step
    many = (not args.no_graph) and not extra and (world == 1 or eng.comm is not None) and getattr(eng, "gU", None) is not None

    def run_steps(k):
        if many:
            eng.replay_many(k)
        else:
            for _ in range(k):
                run()
    run_steps(args.warmup)
    # A short run (the driver's --steps 20 at 0.11 ms is 2 ms of GPU work) is repeated until >= 50 ms have been timed in
    # total: `steps` stays as given, the median repeat is still what is reported, `repeats` says how many there were.
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    probe = time.perf_counter() - t0
    nrep = max(args.repeats, 1)
    if probe * nrep < 0.05:
        nrep = min(int(0.05 / max(probe, 1e-6)) + 1, 2000)
        if world > 1:                              # the same count on every rank
            t = torch.tensor([float(nrep)], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            nrep = int(t[0])
    dts = []
    for _ in range(nrep):
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t[0])
        dts.append(dt)
    dts.sort()
    dt = dts[len(dts) // 2]                       # median repeat: the reported K steps
    eng.check_status()
    elbo, ell, kl = eng.scalars()
    if not (elbo == elbo):
        raise SystemExit("non-finite ELBO after the timed region")

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events around the row-kernel launch, same stream -----------------
        eng.elbo(1)
        n_ev = min(max(args.steps // 4, 50), 500) if w["M"] <= 128 else 5
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
        torch.cuda.synchronize()
        for a, b in evs:
            a.record()
            eng.elbo(2)
            b.record()
        torch.cuda.synchronize()
        ks = sorted(a.elapsed_time(b) for a, b in evs)
        k_ms = sum(ks) / len(ks)
        flop = rows_kernel_flops(dict(w, N=int(Xr.shape[0])))      # this rank's rows
        big = w["M"] > 128
        kname = ("rows phase of the general-M path: per 16k-row chunk K_NM tile kernel + 6 k_gemm launches (4 triangular, "
                 "SYRK, statistics) + flow quadrature" if big else
                 "k_rows (fused K_NM + 4 triangular GEMMs + flow quadrature + SYRK)")
        achieved = flop / (k_ms * 1e-3) / 1e12
        # HBM bytes per launch of the dominant kernel: from a rocprofv3 --pmc pass (separate FETCH_SIZE / WRITE_SIZE passes,
        # tools/probes/profile_round.sh) of THE SAME SOURCES -- the summary carries the source hash of the library it was
        # measured on and is rejected when the library loaded now was built from anything else; never a bare constant
        traffic = None
        tj = args.traffic_json or os.path.join(ROOT, "profiles", "rows_traffic.json")
        if os.path.exists(tj):
            try:
                from tgp.pytorch_amd import lib as _lib
                rec = json.load(open(tj))
                if rec.get("source_hash") == _lib.load().tgp_source_hash().decode():
                    traffic = rec.get("bytes_per_launch", {}).get(args.workload)
                else:
                    log("note: %s was measured on sources %s, this library is %s: roofline.traffic = null"
                        % (tj, rec.get("source_hash"), _lib.load().tgp_source_hash().decode()))
            except Exception:
                traffic = None
        units = (world if args.scaling == "weak" else 1) * (1 + len(extra))
        pg = {"world_size": torch.distributed.get_world_size() if world > 1 else 1,
              "backend": torch.distributed.get_backend() if world > 1 else None,
              "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
              "allreduce": ("none" if world == 1 else ("in-graph" if eng.graph == "full" else "between two graphs")),
              "collective": ("tgp_allreduce_f64 (C ABI, compute stream)" if eng.comm is not None else "torch.distributed"),
              "selfcheck": eng.collective_info.get("selfcheck"), "selfcheck_detail": eng.collective_info.get("why"),
              "allreduce_doubles": eng.fp.n + eng.fp.extra}
        result = {
            # BASELINE.json's metric string for the configuration it is quoted on; other workloads say what they are
            "metric": metric_name(args.workload, w, n_global),
            "value": units * args.steps / dt, "unit": "ELBO-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": len(dts), "ms_per_step_min": 1e3 * dts[0] / args.steps, "ms_per_step_max": 1e3 * dts[-1] / args.steps,
            "config": {"workload": args.workload, "rows_per_gpu": int(Xr.shape[0]), "D": w["D"], "M": w["M"], "S": w["S"],
                       "flow": w["flow"], "global_rows_per_step": n_global, "parallelism": "row-shard x%d" % world,
                       "launch": "eager" if args.no_graph else ("hipgraph, %s steps per graph launch" % ("%d / %d" % (eng.unroll_long, eng.unroll) if getattr(eng, "gL", None) is not None else "%d" % eng.unroll) if many else "hipgraph"), "final_elbo": elbo, "process_group": pg,
                       "replicas": 1 + len(extra),
                       "expected": expected_line(args.workload, w, world, args.scaling, eng.fp.n + eng.fp.extra,
                                                 measured_ms_1gpu=(1e3 * dt / args.steps) if world == 1 else None)},
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "kernel_ms": k_ms, "kernel_ms_min": ks[0], "flop_per_launch": flop},
        }
        exp = result["config"]["expected"]
        if exp is not None:
            # the prediction beside the measurement: > 1 = faster than predicted.  For one rank the basis IS this run's own step
            # (ratio 1 by construction on the general-M workloads); the multi-rank entries are what a SCALE record falsifies.
            exp["measured_value"] = result["value"]
            exp["measured_over_expected"] = result["value"] / exp["value"]
            log("expected %.1f %s (%.4f ms/step), measured %.1f (%.4f ms/step): measured / expected = %.3f%s"
                % (exp["value"], result["unit"], exp["ms_per_step"], result["value"], result["ms_per_step"],
                   exp["measured_over_expected"], "" if world == 1 else "  [the expectation's collective term is an ESTIMATE, "
                   "unmeasured on hardware: `bench.py --gpus %d --allreduce-only` measures it]" % world))
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(prob, args.cpu_seconds, mlp=mlp)
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()
        eng.close()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    sys.exit(main() or 0)
