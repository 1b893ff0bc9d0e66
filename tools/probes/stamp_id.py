import os, sys, torch
os.environ.setdefault("TGP_ALLOW_STALE_LIB", "1")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tgp.pytorch_amd.lib as L
L.LIB_PATH = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/probes/stamp/libtgp_hip.so")
from tgp.pytorch_amd.engine import ElboEngine
from tgp.pytorch_amd import synthetic as orc
PLAN = L.PLAN_ROWS_K16 if os.environ.get("TGP_ROWS_RW") == "16" else 0     # read here, passed as tgp_model.plan
names = ["stage", "x+hdr", "Kexp", "gemm1", "gemm2", "mu/v", "flow", "gemm3+4", "phase1(T)", "phase2(G,s)", "tail"]
for flow in ("idsal3",):
    prob = orc.synthetic_problem(8611, 4, 100, seed=0, flow=flow, S=32)
    eng = ElboEngine(prob["X"], prob["Y"], prob["params"], N_total=8611.0, flow_blocks=prob["program"], S=32, rowp=prob["rowp"], plan=PLAN)
    for _ in range(5):
        eng.elbo()
    torch.cuda.synchronize()
    hdr = eng.ws[8:8 + 11].cpu().tolist()
    d = [(hdr[i + 1] - hdr[i]) * 0.01 for i in range(10)]
    print(flow, os.environ.get("TGP_ROWS_RW"), "total %.1f us :" % ((hdr[10] - hdr[0]) * 0.01), "  ".join("%s %.1f" % (names[i + 1], d[i]) for i in range(10)))
