#!/bin/bash
# gpurun_out/<tag>/ summaries -> profiles/<tag>_* (the files the docs cite).   tools/probes/copy_profiles.sh r02
TAG=${1:-r02}
R=$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)
O=$R/gpurun_out/$TAG; P=$R/profiles
for w in tgp_power_tanh3x2 tgp_power_sal2 svgp_power svgp_boston idtgp_power_sal3 tgp_airline_tanh5x6 tgp_airline_mb10k; do
  [ -s $O/bench_$w.json ] && cp $O/bench_$w.json $P/${TAG}_bench_$w.json
  [ -s $O/bench_rows2_$w.json ] && cp $O/bench_rows2_$w.json $P/${TAG}_bench_teamsplit_$w.json
done
cp $O/bench_tgp_power_tanh3x2_with_traffic.json $P/${TAG}_bench_tgp_power_tanh3x2_with_pmc_traffic.json
cp $O/prof_graph_kernel_stats.csv $P/${TAG}_kernel_stats_tanh3x2_hipgraph.csv
cp $O/prof_graph_rows2_kernel_stats.csv $P/${TAG}_kernel_stats_tanh3x2_hipgraph_teamsplit.csv
cp $O/prof_eager_kernel_stats.csv $P/${TAG}_kernel_stats_tanh3x2_eager.csv
[ -s $O/prof_graph_idtgp_kernel_stats.csv ] && cp $O/prof_graph_idtgp_kernel_stats.csv $P/${TAG}_kernel_stats_idtgp_sal3_hipgraph.csv
cp $O/prof_big_kernel_stats.csv $P/${TAG}_big_kernel_stats_airline_tanh5x6_hipgraph.csv
cp $O/pmc_hbm_traffic_per_kernel.csv $P/${TAG}_pmc_hbm_traffic_per_kernel.csv
cp $O/big_pmc_mfma_util_per_kernel.csv $P/${TAG}_big_pmc_mfma_util_per_kernel.csv
[ -s $O/pmc_mfma_util_per_kernel.csv ] && cp $O/pmc_mfma_util_per_kernel.csv $P/${TAG}_pmc_mfma_util_per_kernel.csv
cp $O/tests.log $P/${TAG}_gpu_tests.log
[ -s $O/stamp_rows2.txt ] && cp $O/stamp_rows2.txt $P/${TAG}_teamsplit_phase_stamps.txt
[ -s $O/mlp_stamps.txt ] && cp $O/mlp_stamps.txt $P/${TAG}_mlp_phase_stamps.txt
ls -la $P | grep ${TAG}_ | wc -l
