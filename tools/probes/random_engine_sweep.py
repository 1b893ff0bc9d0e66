"""One-off randomized sweep of the RESIDENT ENGINE (fused Adam inside the backward launch, HIP-graph replay) against the oracle
stepped with torch.optim.Adam on the CPU: 4 steps, scalars of every step to 1e-8, Z after the last step to 1e-7 (the bars of
tests/test_gpu_big.py::test_big_engine_adam_history_matches_oracle).  Shared flow parameters only (the per-row networks have
their own tests).  Usage: python tools/probes/random_engine_sweep.py [n_cases] [seed]"""
import os, sys, random, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from oracle import tgp_oracle as orc       # checker only
from tgp.pytorch_amd.engine import ElboEngine
from tgp.pytorch_amd import ops
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
flows = [None, "sal1", "sal2", "tanh1x1", "tanh3x2", "tanh5x6"]
rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-300))
bad = 0
for c in range(n_cases):
    M = rng.choice([5, 16, 17, 30, 64, 100, 112, 127, 128])
    D = rng.choice([3, 4, 5, 8, 13, 16])
    flow = rng.choice(flows)
    S = rng.choice([8, 20, 32])
    N = rng.choice([455, 1077, 2153, 3968, 4306, 8611]) if c % 2 == 0 else rng.randint(M, 2500)
    t = time.time()
    prob = orc.synthetic_problem(N, D, M, seed=200 + c, flow=flow, S=S)
    K = orc.scale_rbf(prob["params"]["Z"], prob["params"]["Z"], prob["params"]["raw_lengthscale"], prob["params"]["raw_outputscale"])
    cond = float(torch.linalg.cond(K))
    leaves = {k: t_.clone().requires_grad_(True) for k, t_ in prob["params"].items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=0.01)
    ref = []
    try:
        for _ in range(4):
            elbo, ell, kl = orc.elbo(prob["X"], prob["Y"], leaves["Z"], leaves["raw_lengthscale"], leaves["raw_outputscale"],
                                     leaves["m"], leaves["Lam"], leaves["log_var_noise"], prob["N_total"], prob["program"],
                                     leaves.get("theta"), prob["xs"], prob["ws"])
            ref.append([float(elbo.detach()), float(ell.detach()), float(kl.detach())])
            opt.zero_grad(); (-elbo).backward(); opt.step()
    except Exception as e:
        print("case %2d N=%5d D=%2d M=%3d S=%2d flow=%-8s oracle raised %s (cond %.1e): skipped" % (c, N, D, M, S, flow, type(e).__name__, cond)); continue
    ops._ws_cache.clear()
    eng = ElboEngine(prob["X"], prob["Y"], prob["params"], float(prob["N_total"]), flow_blocks=prob["program"], S=S, device="cuda:0")
    eng.capture()
    hist = []
    for _ in range(4):
        eng.replay(); hist.append(list(eng.scalars()))
    torch.cuda.synchronize()
    st = eng.status.cpu().tolist()
    e1 = rel(torch.tensor(hist, dtype=torch.float64), torch.tensor(ref, dtype=torch.float64))
    e2 = rel(eng.fp.view("Z").cpu(), leaves["Z"].detach())
    ok = st[0] == 0 and e1 < 1e-8 and e2 < 1e-7
    tag = "ok" if ok else ("ill-conditioned, not judged" if cond > 1e11 or st[0] > 0 else "FAIL")
    bad += tag == "FAIL"
    print("case %2d N=%5d D=%2d M=%3d S=%2d flow=%-8s fused_adam=%d %s  scalars %.1e  Z %.1e  status %s cond %.1e (%.1f s)"
          % (c, N, D, M, S, flow, int(eng.fused_adam), tag, e1, e2, st[:3], cond, time.time() - t), flush=True)
print("failures:", bad)
