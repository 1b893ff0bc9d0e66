"""RCCL on the one GPU of the box, world_size 1 (what can be exercised without a multi-GPU node): process-group
initialisation as bench.py does it, an eager all_reduce of the flat gradient buffer's size, the same call captured in a
HIP graph (the `--capture-allreduce` placement) and replayed.  Not a scaling measurement: a 1-rank all-reduce moves no
bytes over xGMI; it shows that the API path the multi-rank engine uses initialises, captures and replays on this stack."""
import os
import sys
import time
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29591")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("backend", dist.get_backend(), "world", dist.get_world_size(), "RCCL/NCCL version", torch.cuda.nccl.version(), flush=True)
g = torch.randn(10545 + 3, dtype=torch.float64, device=dev)          # [grads | ELBO, ELL, KL] of the Power-sized step
ref = g.clone()
dist.all_reduce(g)
torch.cuda.synchronize()
assert torch.equal(g, ref)
t0 = time.perf_counter()
for _ in range(200):
    dist.all_reduce(g)
torch.cuda.synchronize()
print("eager all_reduce of %d doubles: %.1f us per call (host-paced)" % (g.numel(), (time.perf_counter() - t0) / 200 * 1e6), flush=True)
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    dist.all_reduce(g)          # warm-up on the capture stream
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "thread_local"
print("capture_error_mode =", mode, flush=True)
try:
    with torch.cuda.graph(graph, capture_error_mode=mode):
        g.mul_(1.0)
        dist.all_reduce(g)
        g.add_(0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    print("captured [mul, all_reduce, add] replay: %.1f us per replay" % ((time.perf_counter() - t0) / 200 * 1e6))
except Exception as e:          # noqa: BLE001  (a probe: report, do not hide)
    print("capture of all_reduce FAILED on this stack:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
