"""Role timeline of the M x M backward launch k_bwd (stamped build, tools/probes/build_stamp.sh): s_memrealtime (100 MHz) by
thread 0 of column block 0, Lam block 0, row block 0 and the final block, relative to the start of k_reduce's block 0, on a
replayed Power TGP step with the update in the launch.  Usage: python tools/probes/stamp_bwd.py [flow]"""
import os, sys, torch
os.environ.setdefault("TGP_ALLOW_STALE_LIB", "1")
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import tgp.pytorch_amd.lib as L
L.LIB_PATH = os.environ.get("TGP_STAMP_LIB", os.path.join(ROOT, "tools/probes/stamp/libtgp_hip.so"))
from tgp.pytorch_amd.engine import ElboEngine
from tgp.pytorch_amd import synthetic
flow = sys.argv[1] if len(sys.argv) > 1 else "tanh3x2"
NN = 8611
prob = synthetic.synthetic_problem(NN, 4, 100, seed=0, flow=flow, S=32)
eng = ElboEngine(prob["X"], prob["Y"], prob["params"], N_total=float(NN), flow_blocks=prob["program"], S=32)
for _ in range(3):
    eng.step()
eng.capture()
eng.replay_many(30)
torch.cuda.synchronize()
MT, MP, DP = 7, 112, 4
P = eng.fp.sizes.get("theta", 0)
mm = MP * MP
ntri = MT * (MT + 1) // 2
rup = lambda x, a: (x + a - 1) // a * a
slab_len = rup(ntri * 256 + MP * 16 + MP + 4 + P, 16)
o = 64 + 16 + 16 + MP * DP + MP + MP + 2 * rup(P + 1, 16) + 9 * mm + MT * 256 + mm + slab_len + 2 * MT * MP * (DP + 2)
d = eng.ws[o + 200:o + 200 + 17].cpu().tolist()
t0 = d[16]
us = lambda i: (d[i] - t0) * 0.01
print("relative to the start of k_reduce (us):")
print("column block 0 : start %.2f  G staged %.2f  Lbar in LDS %.2f  Q published %.2f" % (us(0), us(1), us(2), us(3)))
print("Lam block 0    : start %.2f  end %.2f" % (us(4), us(5)))
print("row block 0    : start %.2f  prefetch issued %.2f  Q count seen %.2f  Y in LDS %.2f  PP published %.2f" % (us(6), us(7), us(8), us(9), us(10)))
print("final block    : start %.2f  PP count seen %.2f  terms in LDS %.2f  Adam done %.2f  end %.2f" % (us(11), us(12), us(13), us(14), us(15)))
