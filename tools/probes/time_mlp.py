import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tgp.pytorch_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for N in (8611, 128 * 256):
    X = torch.randn(N, 4, dtype=torch.float64, device=dev)
    for nn in (1, 2, 3, 6):
        spec = ops.MlpSpec(4, 50, 2, nn, act="relu", drop_p=0.25, seed=1)
        W = 0.3 * torch.randn(nn * spec.weights_per_net, dtype=torch.float64, device=dev)
        G = torch.randn(N, nn, dtype=torch.float64, device=dev)
        step = torch.zeros(2, dtype=torch.int32, device=dev)
        tf = timeit(lambda: ops.mlp_forward(spec, X, W, True, step))
        tb = timeit(lambda: ops.mlp_backward(spec, X, W, G, True, step))
        tf0 = timeit(lambda: ops.mlp_forward(spec, X, W, False, step))
        print("N=%6d nets=%d chunks=%4d  fwd %.1f us (no dropout %.1f)  bwd+reduce %.1f us" % (N, nn, (N + 63) // 64 * nn, tf, tf0, tb))
