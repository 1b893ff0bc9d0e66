"""Window-by-window timeline of k_big_potrf (diagonal block kb = 1 of the last factorisation), diagnostic build with
-DTGP_STAMPS (tools/probes/build_stamp.sh): per wave, when it entered each window, when its own work was done and when
the window's barrier let it go (s_memrealtime, 100 MHz -> us relative to wave 0's kernel entry)."""
import ctypes, os, sys, torch
os.environ.setdefault("TGP_ALLOW_STALE_LIB", "1")
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import tgp.pytorch_amd.lib as L
L.LIB_PATH = os.path.join(R, "tools/probes/stamp/libtgp_hip.so")
from tgp.pytorch_amd.engine import ElboEngine
from tgp.pytorch_amd import synthetic as orc
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
prob = orc.synthetic_problem(10000, 8, M, seed=0, flow="tanh3x2", S=16)
eng = ElboEngine(prob["X"], prob["Y"], prob["params"], N_total=10000.0, flow_blocks=prob["program"], S=16)
for _ in range(3):
    eng.elbo()
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * (8 * 48))()
lib.tgp_debug_potrf_stamps.argtypes = [ctypes.c_void_p]
rc = lib.tgp_debug_potrf_stamps(buf)
assert rc == 0, rc
t = [[buf[w * 48 + i] for i in range(48)] for w in range(8)]
t0 = t[0][0]
us = lambda x: (x - t0) / 100.0
print("kernel entry -> block in LDS: %.2f us;  end at %.2f us" % (us(t[0][1]), us(t[0][40])))
print("window:   entry | own work done per wave (w0 = pass; w1 = pass while > 64 panel rows) | barrier exit | +phase U")
for j in range(10):
    ent = us(t[0][2 + 3 * j])
    done = " ".join("%6.2f" % (us(t[w][3 + 3 * j]) - ent) for w in range(8))
    ex = us(t[0][4 + 3 * j]) - ent
    nxt = (us(t[0][2 + 3 * (j + 1)]) if j < 9 else us(t[0][40])) - ent
    print("  j=%d  %6.2f | %s | %5.2f | %5.2f" % (j, ent, done, ex, nxt))

