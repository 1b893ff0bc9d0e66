#!/bin/bash
# ID_TGP on Power, M = 100, split 1, 15000 epochs, over cg.config_seed = 0..7 (KMeans restarts, network initialisation,
# dropout stream): the spread of the point-estimate / Bayesian test NLL and RMSE beside the reference's README line
# (/root/reference/README.md:64-65: PE 2.712 3.592, BA 2.672 3.533).   tools/run_idtgp_sweep.sh <outfile> [M] [epochs]
O=${1:-gpurun_out/r03/idtgp_sweep_M100.txt}
M=${2:-100}
EP=${3:-15000}
export TGP_DATA_ROOT=${TGP_DATA_ROOT:-scratch/uci}
mkdir -p $(dirname $O); : > $O.raw
for seed in 0 1 2 3 4 5 6 7; do
python - <<PY 2>&1 | grep "^Dataset" | sed "s/^/seed $seed: /" | tee -a $O.raw
import sys, runpy
import tgp.pytorch_amd.config as cg
cg.config_seed = $seed
cg.set_seed($seed)
sys.argv = ["main", "--model", "ID_TGP", "--dataset", "power", "--train_test_seed_split", "1", "--num_inducing", "$M", "--epochs", "$EP"]
runpy.run_module("tgp.pytorch_amd.main", run_name="__main__")
PY
done
python - <<PY > $O
import re, statistics as st
rows = {"POINT ESTIMATE": [], "BAYESIAN": []}
for line in open("$O.raw"):
    m = re.search(r"(POINT ESTIMATE|BAYESIAN) FLOW , Test Negative LOGL ([-\d.naif]+), Test RMSE ([-\d.naif]+)", line)
    if m:
        rows[m.group(1)].append((float(m.group(2)), float(m.group(3))))
print("# ID_TGP, Power split 1, M = $M, $EP epochs, cg.config_seed = 0..7, one MI355X (tools/run_idtgp_sweep.sh); apply_linear order of this build: Linear -> act -> Dropout")
print("# reference README.md:64-65 (one run, GTX-980):  PE 2.712 3.592   BA 2.672 3.533" if "$M" == "100" else "# reference README.md:68-69: PE 2.744 3.732   BA 2.725 3.681")
for k, v in rows.items():
    fin = [(a, b) for a, b in v if a == a and b == b and abs(b) != float("inf")]
    nll, rm = [a for a, _ in fin], [b for _, b in fin]
    print("%-15s runs %d (finite %d)  NLL mean %.3f sd %.3f min %.3f max %.3f | RMSE mean %.3f sd %.3f min %.3f max %.3f" % (
        k, len(v), len(fin), st.mean(nll), st.pstdev(nll), min(nll), max(nll), st.mean(rm), st.pstdev(rm), min(rm), max(rm)))
print(open("$O.raw").read())
PY
cat $O
