#!/bin/bash
# The command sequence behind profiles/r01_v6_* (run on the GPU box through gpurun; outputs under gpurun_out/v6).
R=$GRAFT_REPO_ROOT
O=gpurun_out/v6
cd $R; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/tests.log
cat $O/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for w in tgp_power_tanh3x2 tgp_power_sal2 svgp_power svgp_boston idtgp_power_sal3; do
  python bench.py --workload $w --cpu-seconds 8 > $O/bench_$w.json 2> $O/bench_$w.err
  cut -c1-230 $O/bench_$w.json
done
python bench.py --workload tgp_airline_tanh5x6 --steps 20 --warmup 3 --cpu-seconds 8 > $O/bench_tgp_airline_tanh5x6.json 2> $O/bench_airline.err
cut -c1-230 $O/bench_tgp_airline_tanh5x6.json
python bench.py --workload tgp_airline_mb10k --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_tgp_airline_mb10k.json 2> $O/bench_mb10k.err
cut -c1-230 $O/bench_tgp_airline_mb10k.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/prof_graph $R/$O/prof_eager $R/$O/pmc_f $R/$O/pmc_w $R/$O/prof_big
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_graph -- python3 $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $R/$O/prof_graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_eager -- python3 $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-graph > $R/$O/prof_eager.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_f -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-graph > $R/$O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_w -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-graph > $R/$O/pmc_w.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_big -- python3 $R/bench.py --workload tgp_airline_tanh5x6 --steps 5 --warmup 1 --no-cpu-baseline > $R/$O/prof_big.log 2>&1
cd $R
for d in prof_graph prof_eager prof_big; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; head -12 "$f" | cut -c1-150; done
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +30M -delete
ls -la $O/pmc_f/*/ | head
