#!/bin/bash
# Reproduces the reference's README result table (/root/reference/README.md:58-69: Power, M = 100 and 5, SVGP / TGP /
# ID_TGP point-estimate and Bayesian flow) with the drop-in main on one MI355X.  The UCI CSVs and split pickles are the
# reference's DATA files; point TGP_DATA_ROOT at a directory holding power.csv + splits_idx_power.pkl.
#   tools/run_readme_table.sh <outdir> [epochs]
O=${1:-gpurun_out/readme}
EP=${2:-15000}
mkdir -p $O
export TGP_DATA_ROOT=${TGP_DATA_ROOT:-scratch/uci}
for M in 100 5; do
  for model in SVGP TGP ID_TGP; do
    SECONDS=0
    python -m tgp.pytorch_amd.main --model $model --dataset power --train_test_seed_split 1 \
        --num_inducing $M --epochs $EP > $O/main_${model}_M$M.log 2>&1
    echo "$model M=$M: ${SECONDS}s wall" | tee -a $O/main_${model}_M$M.log
    grep -h "^Dataset" $O/main_${model}_M$M.log
  done
done
grep -h "^Dataset" $O/main_*.log > $O/readme_table_lines.txt
