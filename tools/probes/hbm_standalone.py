"""Stand-alone distance / flow kernels at Airline scale, for the rocprofv3 passes north_star asks for ("achieved HBM GB/s
on the distance/flow kernels"):   rocprofv3 --kernel-trace --stats ...  /  --pmc FETCH_SIZE  /  --pmc WRITE_SIZE
    python3 tools/probes/hbm_standalone.py [reps]
Launches, each `reps` times after one warm-up:  tgp_knm_f64 (N = 2 M, D = 8, M = 100: 1.6 GB written; N = 250 k, M = 1000:
2 GB), tgp_kmm_f64 (M = 1000), tgp_flow_eval_f64 (S = 32 nodes x N = 2 M, StepTanhL 5x6: 0.5 GB in, 3 x 0.5 GB out),
tgp_ell_flow_f64 (N = 2 M rows, S = 32, same flow: VALU-bound -- its traffic is the per-row vectors only).
Prints HIP-event times and the ALGORITHMIC bytes of each launch (tools/probes/hbm_summary.py divides the PMC bytes and
these by the rocprofv3 kernel durations)."""
import os
import sys
import json
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tgp.pytorch_amd import ops
from tgp.pytorch_amd.synthetic import synthetic_problem

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
out = {}


def timed(name, fn, alg_bytes):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    out[name] = {"ms": ms, "algorithmic_bytes": alg_bytes, "GBps_algorithmic": alg_bytes / ms / 1e6}
    print("%-42s %8.3f ms  %7.1f MB algorithmic  %7.0f GB/s" % (name, ms, alg_bytes / 1e6, alg_bytes / ms / 1e6), flush=True)


g = torch.Generator().manual_seed(0)
N, D = 2000000, 8
X = torch.randn(N, D, generator=g, dtype=torch.float64).to(dev)
rl = torch.full((D,), 1.5, dtype=torch.float64, device=dev)
ro = torch.tensor([0.7], dtype=torch.float64, device=dev)
for M, n in ((100, N), (1000, 250000)):
    Z = torch.randn(M, D, generator=g, dtype=torch.float64).to(dev)
    K = torch.empty(n, M, dtype=torch.float64, device=dev)
    lib = ops.L.load()
    Xn = X[:n].contiguous()

    def knm():
        ops.L.check(lib.tgp_knm_f64(ops.L.ptr(Xn), ops.L.ptr(Z), ops.L.ptr(rl), ops.L.ptr(ro), n, M, D, ops.L.ptr(K),
                                    ops.L.stream_ptr()), "tgp_knm_f64")
    timed("tgp_knm_f64 N=%d M=%d D=8" % (n, M), knm, 8.0 * n * M + 8.0 * (n + M) * D)
    del K
Z = torch.randn(1000, D, generator=g, dtype=torch.float64).to(dev)
timed("tgp_kmm_f64 M=1000 D=8", lambda: ops.kmm(Z, rl, ro), 8.0 * 1000 * 1000 + 8.0 * 1000 * D)

prob = synthetic_problem(1024, 8, 16, seed=0, flow="tanh5x6", S=32)
theta = prob["params"]["theta"].to(dev)
flow = ops.FlowSpec(prob["program"], theta.numel(), 0, dev)
S = 32
f = torch.randn(S, N, generator=g, dtype=torch.float64).to(dev)
timed("tgp_flow_eval_f64 S=32 N=2000000 tanh5x6", lambda: ops.flow_eval(f, flow, theta), 4 * 8.0 * S * N)
del f
Y = torch.randn(N, generator=g, dtype=torch.float64).to(dev)
mu = torch.randn(N, generator=g, dtype=torch.float64).to(dev)
v = (torch.rand(N, generator=g, dtype=torch.float64) + 0.05).to(dev)
lvn = torch.tensor([-2.0], dtype=torch.float64, device=dev)
timed("tgp_ell_flow_f64 N=2000000 S=32 tanh5x6", lambda: ops.ell_flow(Y, mu, v, lvn, flow, theta, S), 5 * 8.0 * N)
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
