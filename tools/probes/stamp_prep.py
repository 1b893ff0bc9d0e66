"""Per-piece shader-clock sums of k_prep_a's block 0 (diagnostic build with -DTGP_STAMPS: tools/probes/build_stamp.sh)."""
import os, sys, torch
os.environ.setdefault("TGP_ALLOW_STALE_LIB", "1")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tgp.pytorch_amd.lib as L
L.LIB_PATH = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/probes/stamp/libtgp_hip.so")
from tgp.pytorch_amd.engine import ElboEngine
from tgp.pytorch_amd import synthetic as orc
prob = orc.synthetic_problem(8611, 4, 100, seed=0, flow="tanh3x2", S=32)
eng = ElboEngine(prob["X"], prob["Y"], prob["params"], N_total=8611.0, flow_blocks=prob["program"], S=32)
for _ in range(5):
    eng.elbo()
torch.cuda.synchronize()
w0 = eng.ws[32:42].cpu().tolist()
w2 = eng.ws[42:58].cpu().tolist()
names = ["fill col 0 + barrier", "P: LDS loads", "P: potrf_panel16", "P: stores", "wait B1", "U work", "wait B2", "-", "-", "tail"]
print("wave 0 (chain), shader cycles summed over the block columns:")
for n, v in zip(names, w0):
    if n != "-":
        print("  %-22s %8.0f" % (n, v))
print("  total %.0f cycles" % sum(w0))
print("window work per wave (cycles; j = 0 | summed over j >= 1): " + "  ".join("w%d %.0f|%.0f" % (k, w2[2 * k], w2[2 * k + 1]) for k in range(8)))
