"""Timeline of the fused step launch (diagnostic build with -DTGP_STAMPS): the chain block's window ends and row block
0's phase boundaries on the same 100 MHz clock."""
import os, sys, torch
os.environ.setdefault("TGP_ALLOW_STALE_LIB", "1")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tgp.pytorch_amd.lib as L
L.LIB_PATH = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/probes/stamp/libtgp_hip.so")
from tgp.pytorch_amd.engine import ElboEngine
from tgp.pytorch_amd import synthetic as orc
prob = orc.synthetic_problem(int(os.environ.get("NROWS", "8611")), 4, 100, seed=0, flow="tanh3x2", S=32)
eng = ElboEngine(prob["X"], prob["Y"], prob["params"], N_total=float(os.environ.get("NROWS", "8611")), flow_blocks=prob["program"], S=32)
for _ in range(5):
    eng.elbo()
torch.cuda.synchronize()
rows = eng.ws[8:8 + 11].cpu().tolist()
ch = eng.ws[32:32 + 17].cpu().tolist()
t0 = min(ch[0], rows[0])
print("chain block: start %.2f  col0 filled %.2f" % ((ch[0] - t0) * 0.01, (ch[1] - t0) * 0.01))
for j in range(7):
    print("  window %d: barrier B1 at %.2f us, U done at %.2f" % (j, (ch[2 + 2 * j] - t0) * 0.01, (ch[3 + 2 * j] - t0) * 0.01))
print("  terminal publish %.2f" % ((ch[16] - t0) * 0.01))
names = ["start", "staged", "K tile", "A done", "B done", "mu/v", "flow", "C,Kbar", "T", "G,s", "end"]
print("row block 0: " + "  ".join("%s %.2f" % (n, (r - t0) * 0.01) for n, r in zip(names, rows)))
# per-wave window work of the chain block: offsets of the tgp::Plan (see make_plan) are not visible from Python; the
# library reports the debug area's offset through the header word 61
off = int(eng.ws[61].item())
d = eng.ws[off:off + 7 * 16].cpu().tolist()
for j in range(7):
    print("  window %d: " % j + "  ".join("w%d %.2f-%.2f (drain %.2f tiles %.2f)" % (w, (d[(j * 4 + w) * 4] - t0) * 0.01, (d[(j * 4 + w) * 4 + 3] - t0) * 0.01,
          (d[(j * 4 + w) * 4 + 1] - t0) * 0.01, (d[(j * 4 + w) * 4 + 2] - t0) * 0.01) for w in range(4)))
lg = eng.ws[off + 120:off + 120 + 96].cpu().tolist()
print("window 3 task log (task id @ us): " + " | ".join("w%d " % w + " ".join("%d@%.2f" % (int(lg[w * 24 + 2 * k]), (lg[w * 24 + 2 * k + 1] - t0) * 0.01) for k in range(8) if lg[w * 24 + 2 * k + 1] > 0) for w in range(1, 4)))
