import os, sys, torch
os.environ["TGP_ALLOW_STALE_LIB"] = "1"   # the stamped build carries no source hash
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tgp.pytorch_amd.lib as L
L.LIB_PATH = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/probes/stamp/libtgp_hip.so")
import bench
from tgp.pytorch_amd.engine import ElboEngine
w = bench.WORKLOADS["idtgp_power_sal3"]
prob = bench.make_problem(w, 0)
spec, W = bench.make_mlp(w, 0)
eng = ElboEngine(prob["X"], prob["Y"], prob["params"], float(prob["N_total"]), flow_blocks=prob["program"], S=w["S"], mlp=spec, mlp_weights=W)
for _ in range(3):
    eng.forward_backward()
torch.cuda.synchronize()
st = eng.mlp_ws[-16:].cpu().tolist()
nch = (eng.N + 63) // 64
per = (nch * spec.nnets + 511) // 512
G = (nch + per - 1) // per                       # mlp_groups(N, nnets, MLP_SLOTS_BWD)
off = G * spec.nnets * spec.weights_per_net      # the 16 spare doubles behind the partials
st = eng.mlp_ws[off:off + 10].cpu().tolist()
# stamps: 0 kernel start; 1 after the LAST chunk's input staging + barrier; then the phases of that chunk
names = ["weights + earlier chunks + inputs", "forward chain", "output layer + delta_L + strip + barrier", "dwo out",
         "delta-prop + dW2", "barrier + strip + barrier", "dW1"]
print("k_mlp_bwd workgroup 0:", "  ".join("%s %.1f" % (names[i], (st[i + 1] - st[i]) * 0.01) for i in range(7)),
      " total %.1f us" % ((st[7] - st[0]) * 0.01))
