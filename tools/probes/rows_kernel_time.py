"""Time of the ROWS phase alone (tgp_elbo_step_phases_f64 with TGP_PHASE_ROWS) for a list of row counts -- run it under
different TGP_ROWS4 settings to compare the 16-rows-per-wave kernel (TGP_ROWS4=0) with k_rows4.  The environment variables are
read HERE (this probe) and become tgp_model.plan of the engine's calls; the library itself reads no environment.
Usage: TGP_ROWS4=<0|4|8|unset> [TGP_ROWS_RW=16] python tools/probes/rows_kernel_time.py [flow] N1 N2 ..."""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from tgp.pytorch_amd.engine import ElboEngine
from tgp.pytorch_amd import synthetic, lib
_r4, _rw = os.environ.get("TGP_ROWS4"), os.environ.get("TGP_ROWS_RW")
PLAN = {None: 0, "0": lib.PLAN_ROWS_K, "4": lib.PLAN_ROWS4_NW4, "8": lib.PLAN_ROWS4_NW8}[_r4]
if _rw == "16":
    PLAN = lib.PLAN_ROWS_K16          # (k_rows at 16 rows per wave; with TGP_ROWS4=0 that is what the old pair of switches selected)
args = sys.argv[1:]
flow = "tanh3x2"
if args and not args[0].isdigit():
    flow = None if args[0] == "none" else args[0]
    args = args[1:]
out = []
for N in [int(a) for a in args] or [8611]:
    prob = synthetic.synthetic_problem(N, 4, 100, seed=0, flow=flow, S=32)
    eng = ElboEngine(prob["X"], prob["Y"], prob["params"], N_total=float(N), flow_blocks=prob["program"], S=32, plan=PLAN)
    eng.elbo()                       # prepare + rows + backward once (fills L, Lq, ...)
    for _ in range(20):
        eng.elbo(phases=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(10):
            eng.elbo(phases=2)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps // 10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    out.append("N=%d %.1f us" % (N, e0.elapsed_time(e1) * 1e3 / reps))
print("TGP_ROWS4=%s flow=%s : " % (os.environ.get("TGP_ROWS4", "auto"), flow) + "   ".join(out))
