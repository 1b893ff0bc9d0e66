import os, sys, torch
os.environ.setdefault("TGP_ALLOW_STALE_LIB", "1")   # the stamped build carries no source hash (build_stamp.sh)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tgp.pytorch_amd.lib as L
L.LIB_PATH = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/probes/stamp/libtgp_hip.so")
from tgp.pytorch_amd.engine import ElboEngine
from tgp.pytorch_amd import synthetic as orc
names = ["stage", "x+hdr", "Kexp", "gemm1", "gemm2", "mu/v", "flow", "gemm3+4", "phase1(T)", "phase2(G,s)", "tail"]
for flow in ("tanh3x2", "sal2", None):
    prob = orc.synthetic_problem(8611, 4, 100, seed=0, flow=flow, S=32)
    eng = ElboEngine(prob["X"], prob["Y"], prob["params"], N_total=8611.0, flow_blocks=prob["program"], S=32)
    for _ in range(5):
        eng.elbo()
    torch.cuda.synchronize()
    hdr = eng.ws[8:8 + 11].cpu().tolist()
    d = [(hdr[i + 1] - hdr[i]) * 0.01 for i in range(10)]
    ex = eng.ws[8 + 17:8 + 19].cpu().tolist()
    print("phase2 detail: write+sync %.1f  G tiles %.1f  s tiles %.1f" % ((ex[0]-hdr[8])*0.01, (ex[1]-ex[0])*0.01, (hdr[9]-ex[1])*0.01))
    ck = eng.ws[8 + 11].item(), eng.ws[8 + 23].item()
    print("shader clock during k_rows: %.0f cycles in %.1f us = %.2f GHz" % (ck[1] - ck[0], (hdr[10] - hdr[0]) * 0.01, (ck[1] - ck[0]) / ((hdr[10] - hdr[0]) * 10.0)))
    print(flow, "total %.1f us :" % ((hdr[10] - hdr[0]) * 0.01), "  ".join("%s %.1f" % (names[i + 1], d[i]) for i in range(10)))
