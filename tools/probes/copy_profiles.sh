#!/bin/bash
# gpurun_out/<tag>/ summaries -> profiles/<tag>_* (the files the docs cite).   tools/probes/copy_profiles.sh r03
TAG=${1:-r06}
R=$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)
O=$R/gpurun_out/$TAG; P=$R/profiles
for w in tgp_power_tanh3x2 tgp_power_tanh3x2_driver_cmdline tgp_power_sal2 svgp_power svgp_boston idtgp_power_sal3 tgp_airline_tanh5x6 tgp_airline_mb10k tgp_airline_mb10k_rank8 selflaunch_2ranks_1gpu_gloo_strong; do
  [ -s $O/bench_$w.json ] && cp $O/bench_$w.json $P/${TAG}_bench_$w.json
done
[ -s $O/rows_traffic.json ] && cp $O/rows_traffic.json $P/rows_traffic.json      # bench.py's default --traffic-json (hash-keyed)
cp $O/prof_graph_kernel_stats.csv $P/${TAG}_kernel_stats_tanh3x2_hipgraph.csv
cp $O/prof_eager_kernel_stats.csv $P/${TAG}_kernel_stats_tanh3x2_eager.csv
[ -s $O/prof_graph_idtgp_kernel_stats.csv ] && cp $O/prof_graph_idtgp_kernel_stats.csv $P/${TAG}_kernel_stats_idtgp_sal3_hipgraph.csv
cp $O/prof_big_kernel_stats.csv $P/${TAG}_big_kernel_stats_airline_tanh5x6_hipgraph.csv
[ -s $O/prof_big_eager_kernel_stats.csv ] && cp $O/prof_big_eager_kernel_stats.csv $P/${TAG}_big_kernel_stats_airline_tanh5x6_eager.csv
[ -s $O/prof_mb_kernel_stats.csv ] && cp $O/prof_mb_kernel_stats.csv $P/${TAG}_big_kernel_stats_airline_mb10k_hipgraph.csv
cp $O/pmc_hbm_traffic_per_kernel.csv $P/${TAG}_pmc_hbm_traffic_per_kernel.csv
cp $O/big_pmc_mfma_util_per_kernel.csv $P/${TAG}_big_pmc_mfma_util_per_kernel.csv
[ -s $O/pmc_mfma_util_per_kernel.csv ] && cp $O/pmc_mfma_util_per_kernel.csv $P/${TAG}_pmc_mfma_util_per_kernel.csv
[ -s $O/pmc_hbm_big.csv ] && cp $O/pmc_hbm_big.csv $P/${TAG}_pmc_hbm_big_airline_tanh5x6.csv
[ -s $O/pmc_hbm_standalone.csv ] && cp $O/pmc_hbm_standalone.csv $P/${TAG}_pmc_hbm_standalone_distance_flow.csv
for f in hbm_standalone bayes_eval_timing prep_phase_stamps rows_phase_stamps potrf_panel_rate gemm_bench wg_placement big_potrf_window_timeline rows4_phase_stamps rows_kernel_time bwd_role_stamps; do [ -s $O/$f.txt ] && cp $O/$f.txt $P/${TAG}_$f.txt; done
for w in tgp_airline_mb10k tgp_airline_mb10k_rank8 idtgp_power_sal3 tgp_power_tanh3x2; do [ -s $O/timeline_$w.txt ] && cp $O/timeline_$w.txt $P/${TAG}_timeline_$w.txt; done
cp $O/tests.log $P/${TAG}_gpu_tests.log
[ -s $O/allreduce_only_1rank.json ] && cp $O/allreduce_only_1rank.json $P/${TAG}_allreduce_only_1rank.json
[ -s $O/ab_previous_round.txt ] && cp $O/ab_previous_round.txt $P/${TAG}_ab_previous_round.txt
[ -s $O/pmc_valu_standalone.csv ] && cp $O/pmc_valu_standalone.csv $P/${TAG}_pmc_valu_standalone_distance_flow.csv
[ -s $O/pmc_valu_rows.csv ] && cp $O/pmc_valu_rows.csv $P/${TAG}_pmc_valu_per_kernel.csv
ls -la $P | grep ${TAG}_ | wc -l
