#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/final/tests.log
cat gpurun_out/final/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for w in tgp_power_tanh3x2 tgp_power_sal2 svgp_power svgp_boston idtgp_power_sal3; do
  python bench.py --workload $w --cpu-seconds 8 > gpurun_out/final/bench_$w.json 2> gpurun_out/final/bench_$w.err
  cut -c1-230 gpurun_out/final/bench_$w.json
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/final/prof_graph $R/gpurun_out/final/prof_eager $R/gpurun_out/final/pmc_f $R/gpurun_out/final/pmc_w
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_graph -- python3 $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $R/gpurun_out/final/prof_graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_eager -- python3 $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-graph > $R/gpurun_out/final/prof_eager.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final/pmc_f -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-graph > $R/gpurun_out/final/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final/pmc_w -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-graph > $R/gpurun_out/final/pmc_w.log 2>&1
cd $R
for d in prof_graph prof_eager; do f=$(find gpurun_out/final/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -9 "$f" | cut -c1-150; done
find gpurun_out/final -name "*kernel_trace.csv" -size +8M -delete
find gpurun_out/final -name "*counter_collection.csv" -size +20M -delete
ls gpurun_out/final/pmc_f/*/ | head
