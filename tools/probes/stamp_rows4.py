"""Per-wave phase timeline of k_rows4 (stamped build, tools/probes/build_stamp.sh): s_memrealtime (100 MHz) at the phase
boundaries of every wave of workgroup 0 of the training launch, at N = 2153 and 4306 rows (a quarter / half of Power:
4- and 8-wave workgroups).  Usage: python tools/probes/stamp_rows4.py [flow]"""
import os, sys, torch
os.environ.setdefault("TGP_ALLOW_STALE_LIB", "1")
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import tgp.pytorch_amd.lib as L
L.LIB_PATH = os.path.join(ROOT, "tools/probes/stamp/libtgp_hip.so")
from tgp.pytorch_amd.engine import ElboEngine
from tgp.pytorch_amd import synthetic
flows = sys.argv[1:] or ["tanh3x2", "sal2"]
SIZES = [(2153, 4), (4306, 8)]      # (rows, waves per workgroup the library picks there at MT = 7)
names = ["stage", "K strip", "gemm1", "gemm2", "mu/v", "flow", "gemm3", "gemm4", "stats", "tail"]
for flow, (NN, nw) in [(f, s_) for f in flows for s_ in SIZES]:
    flow = None if flow == "none" else flow
    prob = synthetic.synthetic_problem(NN, 4, 100, seed=0, flow=flow, S=32)
    eng = ElboEngine(prob["X"], prob["Y"], prob["params"], N_total=float(NN), flow_blocks=prob["program"], S=32)
    for _ in range(5):
        eng.elbo()
    torch.cuda.synchronize()
    # Plan.dbg: offsets as in make_plan (tgp_dev.hpp); P = shared flow parameters
    MT, MP, DP = 7, 112, 4
    P = 0 if flow is None else eng.fp.sizes.get("theta", 0)
    mm = MP * MP
    ntri = MT * (MT + 1) // 2
    rup = lambda x, a: (x + a - 1) // a * a
    slab_len = rup(ntri * 256 + MP * 16 + MP + 4 + P, 16)
    o = 64 + 16 + 16 + MP * DP + MP + MP + 2 * rup(P + 1, 16) + 9 * mm + MT * 256 + mm + slab_len + 2 * MT * MP * (DP + 2)
    d = eng.ws[o:o + 256].cpu().tolist()
    t0 = min(d[w * 20] for w in range(nw))
    print("== %s, N = %d, %d waves per workgroup" % (flow, NN, nw))
    for w in range(nw):
        s = d[w * 20:w * 20 + 11]
        print("wave %2d +%.2f : " % (w, (s[0] - t0) * 0.01) + "  ".join("%s %.2f" % (names[i], (s[i + 1] - s[i]) * 0.01) for i in range(10)) + "   total %.2f" % ((s[10] - s[0]) * 0.01))
