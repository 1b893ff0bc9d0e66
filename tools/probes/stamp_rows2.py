"""Per-phase times of the team-split row kernel k_rows2 (diagnostic build with -DTGP_STAMPS: tools/probes/build_stamp.sh).
Thread 0 of workgroup 0 stamps s_memrealtime (100 MHz) right after each workgroup barrier."""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
os.environ["TGP_ALLOW_STALE_LIB"] = "1"
import tgp.pytorch_amd.lib as L
L.LIB_PATH = os.path.join(ROOT, "tools/probes/stamp/libtgp_hip.so")
L.load().tgp_set_rows_kernel(1)
from tgp.pytorch_amd.engine import ElboEngine
from tgp.pytorch_amd import synthetic as orc
names = ["stage", "K", "gemm1", "gemm2", "mu,v+flow", "Bvb", "gemm3", "gemm4", "E+T", "A-tile", "G,s", "tail"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8611
for flow in ("tanh3x2", "sal2", None):
    prob = orc.synthetic_problem(N, 4, 100, seed=0, flow=flow, S=32)
    eng = ElboEngine(prob["X"], prob["Y"], prob["params"], N_total=float(N), flow_blocks=prob["program"], S=32)
    warm = len(sys.argv) > 2 and sys.argv[2] == "warm"
    for _ in range(5):
        eng.elbo()
    if warm:            # the row kernel alone, three times in a row: operands already in every XCD's L2
        for _ in range(3):
            eng.elbo(2)
    torch.cuda.synchronize()
    h = eng.ws[8:8 + 24].cpu().tolist()
    t = h[0:10] + [h[18], h[10], h[17]]
    d = [(t[i + 1] - t[i]) * 0.01 for i in range(12)]
    tot = (h[17] - h[0]) * 0.01
    print(flow, "N=%d total %.1f us, clock %.2f GHz :" % (N, tot, (h[23] - h[11]) / (tot * 1000.0)),
          "  ".join("%s %.1f" % (names[i], d[i]) for i in range(12)))
