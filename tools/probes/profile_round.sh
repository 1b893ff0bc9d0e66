#!/bin/bash
# The command sequence behind profiles/<tag>_* (run on the GPU box through gpurun; raw outputs under gpurun_out/<tag>,
# summaries are copied into profiles/ by hand).   tools/probes/profile_round.sh r02
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
O=gpurun_out/$TAG
cd $R; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/tests.log
cat $O/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for w in tgp_power_tanh3x2 tgp_power_sal2 svgp_power svgp_boston idtgp_power_sal3; do
  python bench.py --workload $w --cpu-seconds 8 > $O/bench_$w.json 2> $O/bench_$w.err
  cut -c1-230 $O/bench_$w.json
  TGP_ROWS2=1 python bench.py --workload $w --no-cpu-baseline > $O/bench_rows2_$w.json 2> $O/bench_rows2_$w.err
  cut -c1-230 $O/bench_rows2_$w.json
done
python bench.py --workload tgp_airline_tanh5x6 --steps 20 --warmup 3 --cpu-seconds 8 > $O/bench_tgp_airline_tanh5x6.json 2> $O/bench_airline.err
cut -c1-230 $O/bench_tgp_airline_tanh5x6.json
python bench.py --workload tgp_airline_mb10k --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_tgp_airline_mb10k.json 2> $O/bench_mb10k.err
cut -c1-230 $O/bench_tgp_airline_mb10k.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/prof_graph $R/$O/prof_graph_idtgp $R/$O/prof_graph_rows2 $R/$O/prof_eager $R/$O/pmc_f $R/$O/pmc_w $R/$O/prof_big $R/$O/pmc_big
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph -- python3 $R/bench.py --steps 500 --warmup 50 --repeats 1 --no-cpu-baseline > $R/$O/prof_graph.log 2>&1
export TGP_ROWS2=1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph_rows2 -- python3 $R/bench.py --steps 500 --warmup 50 --repeats 1 --no-cpu-baseline > $R/$O/prof_graph_rows2.log 2>&1
unset TGP_ROWS2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph_idtgp -- python3 $R/bench.py --workload idtgp_power_sal3 --steps 500 --warmup 50 --repeats 1 --no-cpu-baseline > $R/$O/prof_graph_idtgp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_eager -- python3 $R/bench.py --steps 300 --warmup 30 --repeats 1 --no-cpu-baseline --no-graph > $R/$O/prof_eager.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_f -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-cpu-baseline --no-graph > $R/$O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_w -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-cpu-baseline --no-graph > $R/$O/pmc_w.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_big -- python3 $R/bench.py --workload tgp_airline_tanh5x6 --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline > $R/$O/prof_big.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES --output-format csv -d $R/$O/pmc_big -- python3 $R/bench.py --workload tgp_airline_tanh5x6 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-graph > $R/$O/pmc_big.log 2>&1
cd $R
for d in prof_graph prof_graph_idtgp prof_graph_rows2 prof_eager prof_big; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -12 "$f" | cut -c1-150; cp "$f" $O/${d}_kernel_stats.csv; done
FC=$(find $O/pmc_f -name "*counter_collection.csv" | head -1); WC=$(find $O/pmc_w -name "*counter_collection.csv" | head -1)
python tools/probes/pmc_summary.py FETCH_SIZE=$FC WRITE_SIZE=$WC > $O/pmc_hbm_traffic_per_kernel.csv; head -20 $O/pmc_hbm_traffic_per_kernel.csv | cut -c1-150
# per-launch HBM bytes of the dominant kernel from THIS build's PMC passes -> bench --traffic-json (same session)
python - <<PY
import csv, json
f = {}; w = {}
for line in open("$O/pmc_hbm_traffic_per_kernel.csv"):
    if line.startswith("#") or line.startswith("counter"): continue
    r = next(csv.reader([line]))
    (f if r[0] == "FETCH_SIZE" else w)[r[1]] = float(r[3])
k = [n for n in f if "k_rows<" in n][0]
json.dump({"tgp_power_tanh3x2": (2 * f[k] + w[k]) * 1024.0}, open("$O/rows_traffic.json", "w"))
print("k_rows HBM bytes per launch:", (2 * f[k] + w[k]) * 1024.0)
PY
python bench.py --steps 2000 --warmup 100 --cpu-seconds 8 --traffic-json $O/rows_traffic.json > $O/bench_tgp_power_tanh3x2_with_traffic.json 2> /dev/null; cut -c1-300 $O/bench_tgp_power_tanh3x2_with_traffic.json
BC=$(find $O/pmc_big -name "*counter_collection.csv" | head -1)
python tools/probes/pmc_mfma_summary.py "$BC" "python3 bench.py --workload tgp_airline_tanh5x6 --steps 3 --warmup 1 --no-graph" > $O/big_pmc_mfma_util_per_kernel.csv; head -14 $O/big_pmc_mfma_util_per_kernel.csv | cut -c1-170
cd /tmp; rm -rf $R/$O/pmc_m
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES --output-format csv -d $R/$O/pmc_m -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-cpu-baseline --no-graph > $R/$O/pmc_m.log 2>&1
cd $R
MC=$(find $O/pmc_m -name "*counter_collection.csv" | head -1)
python tools/probes/pmc_mfma_summary.py "$MC" "python3 bench.py --steps 100 --warmup 10 --no-graph (tgp_power_tanh3x2)" > $O/pmc_mfma_util_per_kernel.csv; head -12 $O/pmc_mfma_util_per_kernel.csv | cut -c1-170
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +30M -delete
python tools/probes/stamp_rows2.py 8611 > $O/stamp_rows2.txt 2>&1; cat $O/stamp_rows2.txt | tail -3
python tools/probes/mlp_stamps.py > $O/mlp_stamps.txt 2>&1; tail -1 $O/mlp_stamps.txt
python tools/probes/time_mlp.py 2>&1 | grep "^N=" >> $O/mlp_stamps.txt
