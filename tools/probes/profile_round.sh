#!/bin/bash
# The command sequence behind profiles/<tag>_* (run on the GPU box through gpurun; raw outputs under gpurun_out/<tag>,
# tools/probes/copy_profiles.sh copies the summaries into profiles/).   tools/probes/profile_round.sh r03
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
O=gpurun_out/$TAG
cd "$R" || exit 1
mkdir -p "$O"
D="$R/$O"                       # every rocprofv3 output directory below is a fresh sub-directory of this one
fresh() { mkdir -p "$D/$1.new" && mv "$D/$1.new" "$D/$1.$$" 2>/dev/null; echo "$D/$1.$$"; }
SH=$(python -c "from tgp.pytorch_amd import lib; print(lib.source_hash())")
echo "source hash $SH"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|warnings summary" | tail -4 > $O/tests.log   # (RCCL prints its banner after pytest's summary)
cat $O/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
# ---- per-launch HBM bytes of the dominant kernel (separate FETCH_SIZE / WRITE_SIZE passes), keyed by the source hash ----
cd /tmp && export TMPDIR=/tmp
PF=$(fresh pmc_f); PW=$(fresh pmc_w)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $PF -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-cpu-baseline --no-graph > $D/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $PW -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-cpu-baseline --no-graph > $D/pmc_w.log 2>&1
cd $R
FC=$(find $PF -name "*counter_collection.csv" | head -1); WC=$(find $PW -name "*counter_collection.csv" | head -1)
python tools/probes/pmc_summary.py FETCH_SIZE=$FC WRITE_SIZE=$WC > $O/pmc_hbm_traffic_per_kernel.csv; head -20 $O/pmc_hbm_traffic_per_kernel.csv | cut -c1-150
python - <<PY
import csv, json
f = {}; w = {}
for line in open("$O/pmc_hbm_traffic_per_kernel.csv"):
    if line.startswith("#") or line.startswith("counter"): continue
    r = next(csv.reader([line]))
    (f if r[0] == "FETCH_SIZE" else w)[r[1]] = float(r[3])
k = [n for n in f if "k_rows<" in n][0]
rec = {"source_hash": "$SH", "kernel": k, "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes), bench.py --steps 100 --warmup 10 --no-graph; bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 (gfx950 correction, MI355X_MICROARCH.md HBM section)",
       "bytes_per_launch": {"tgp_power_tanh3x2": (2 * f[k] + w[k]) * 1024.0}}
json.dump(rec, open("$O/rows_traffic.json", "w"), indent=1)
print("k_rows HBM bytes per launch:", rec["bytes_per_launch"])
PY
# ---- bench lines (the headline one carries roofline.traffic from the summary above: same sources) ----
python bench.py --steps 2000 --warmup 100 --cpu-seconds 8 --traffic-json $O/rows_traffic.json > $O/bench_tgp_power_tanh3x2.json 2> $O/bench_tgp_power_tanh3x2.err
cut -c1-330 $O/bench_tgp_power_tanh3x2.json
python bench.py --steps 20 --warmup 5 --cpu-seconds 8 --traffic-json $O/rows_traffic.json > $O/bench_tgp_power_tanh3x2_driver_cmdline.json 2> /dev/null
for w in tgp_power_sal2 svgp_power svgp_boston idtgp_power_sal3; do
  python bench.py --workload $w --cpu-seconds 8 > $O/bench_$w.json 2> $O/bench_$w.err
  cut -c1-230 $O/bench_$w.json
done
python bench.py --workload tgp_airline_tanh5x6 --steps 20 --warmup 3 --cpu-seconds 8 > $O/bench_tgp_airline_tanh5x6.json 2> $O/bench_airline.err
cut -c1-230 $O/bench_tgp_airline_tanh5x6.json
for w in tgp_airline_mb10k tgp_airline_mb10k_rank8; do
  python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  cut -c1-230 $O/bench_$w.json
done
TGP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline 2> /dev/null | grep "^{" > $O/bench_selflaunch_2ranks_1gpu_gloo_strong.json
cut -c1-230 $O/bench_selflaunch_2ranks_1gpu_gloo_strong.json
python bench.py --allreduce-only 2> /dev/null | grep "^{" > $O/allreduce_only_1rank.json; cut -c1-300 $O/allreduce_only_1rank.json
# ---- the same bench lines on the previous round's tree, alternately, on THIS box (a git worktree of its commit under ab_r05/, built there)
[ -d ab_r05 ] && bash tools/probes/ab_r05.sh > $O/ab_previous_round.txt 2>&1 && cat $O/ab_previous_round.txt
# ---- kernel stats ----
cd /tmp
PG=$(fresh prof_graph); PI=$(fresh prof_graph_idtgp); PE=$(fresh prof_eager); PB=$(fresh prof_big); PM=$(fresh prof_mb); PBE=$(fresh prof_big_eager)
rocprofv3 --kernel-trace --stats --output-format csv -d $PG -- python3 $R/bench.py --steps 500 --warmup 50 --repeats 1 --no-cpu-baseline > $D/prof_graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $PI -- python3 $R/bench.py --workload idtgp_power_sal3 --steps 500 --warmup 50 --repeats 1 --no-cpu-baseline > $D/prof_graph_idtgp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $PE -- python3 $R/bench.py --steps 300 --warmup 30 --repeats 1 --no-cpu-baseline --no-graph > $D/prof_eager.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $PB -- python3 $R/bench.py --workload tgp_airline_tanh5x6 --steps 5 --warmup 1 --repeats 1 --no-cpu-baseline > $D/prof_big.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $PM -- python3 $R/bench.py --workload tgp_airline_mb10k --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline > $D/prof_mb.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $PBE -- python3 $R/bench.py --workload tgp_airline_tanh5x6 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-graph > $D/prof_big_eager.log 2>&1
# ---- MFMA counters ----
QB=$(fresh pmc_big); QM=$(fresh pmc_m)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES --output-format csv -d $QB -- python3 $R/bench.py --workload tgp_airline_tanh5x6 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-graph > $D/pmc_big.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES --output-format csv -d $QM -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-cpu-baseline --no-graph > $D/pmc_m.log 2>&1
# ---- HBM counters of the general-M path and of the stand-alone distance / flow kernels (north_star's evidence) ----
QBF=$(fresh pmc_bf); QBW=$(fresh pmc_bw); HS=$(fresh hs_stats); HF=$(fresh hs_f); HW=$(fresh hs_w)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $QBF -- python3 $R/bench.py --workload tgp_airline_tanh5x6 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-graph > $D/pmc_bf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $QBW -- python3 $R/bench.py --workload tgp_airline_tanh5x6 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-graph > $D/pmc_bw.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $HS -- python3 $R/tools/probes/hbm_standalone.py 3 > $D/hs_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $HF -- python3 $R/tools/probes/hbm_standalone.py 3 > $D/hs_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $HW -- python3 $R/tools/probes/hbm_standalone.py 3 > $D/hs_w.log 2>&1
cd $R
stats() { find $1 -name "*kernel_stats.csv" | head -1; }
cc() { find $1 -name "*counter_collection.csv" | head -1; }
for pair in "prof_graph:$PG" "prof_graph_idtgp:$PI" "prof_eager:$PE" "prof_big:$PB" "prof_mb:$PM" "prof_big_eager:$PBE"; do n=${pair%%:*}; d=${pair#*:}; f=$(stats $d); echo "== $n"; head -12 "$f" | cut -c1-150; cp "$f" $O/${n}_kernel_stats.csv; done
python tools/probes/pmc_mfma_summary.py "$(cc $QB)" "python3 bench.py --workload tgp_airline_tanh5x6 --steps 3 --warmup 1 --no-graph" > $O/big_pmc_mfma_util_per_kernel.csv; head -14 $O/big_pmc_mfma_util_per_kernel.csv | cut -c1-170
python tools/probes/pmc_mfma_summary.py "$(cc $QM)" "python3 bench.py --steps 100 --warmup 10 --no-graph (tgp_power_tanh3x2)" > $O/pmc_mfma_util_per_kernel.csv; head -12 $O/pmc_mfma_util_per_kernel.csv | cut -c1-170
python tools/probes/hbm_summary.py $O/prof_big_eager_kernel_stats.csv "$(cc $QBF)" "$(cc $QBW)" "python3 bench.py --workload tgp_airline_tanh5x6 --steps 3 --warmup 1 --no-graph (eager launches: the durations are those of the same eager run, still with the helper-stream overlap of the chunk pipeline)" > $O/pmc_hbm_big.csv; head -24 $O/pmc_hbm_big.csv | cut -c1-190
python tools/probes/hbm_summary.py "$(stats $HS)" "$(cc $HF)" "$(cc $HW)" "python3 tools/probes/hbm_standalone.py 3" > $O/pmc_hbm_standalone.csv; cat $O/pmc_hbm_standalone.csv | cut -c1-190
python tools/probes/hbm_standalone.py 5 2>&1 | grep -v amdgpu > $O/hbm_standalone.txt; cat $O/hbm_standalone.txt
python tools/probes/time_bayes_eval.py 2>&1 | grep -v amdgpu > $O/bayes_eval_timing.txt; cat $O/bayes_eval_timing.txt
python tools/probes/stamp_prep.py 2>&1 | grep -v amdgpu > $O/prep_phase_stamps.txt; head -12 $O/prep_phase_stamps.txt
python tools/probes/stamp_run.py 2>&1 | grep -v amdgpu > $O/rows_phase_stamps.txt; tail -4 $O/rows_phase_stamps.txt
python tools/probes/stamp_big.py 2>&1 | grep -v amdgpu > $O/big_potrf_window_timeline.txt; head -14 $O/big_potrf_window_timeline.txt
python tools/probes/stamp_rows4.py 2>&1 | grep -v "amdgpu\|^ROCm\|^Hostname\|^Librccl" > $O/rows4_phase_stamps.txt; head -6 $O/rows4_phase_stamps.txt
# ---- the three row kernels side by side: ROWS phase alone at shard sizes and at the full batch ----
( for r in 0 auto; do if [ $r = auto ]; then unset TGP_ROWS4; else export TGP_ROWS4=$r TGP_ROWS_RW=16; fi
    python tools/probes/rows_kernel_time.py tanh3x2 455 1077 2153 3968 4306 7936 8611 2>&1 | grep "TGP_ROWS4="
    if [ $r = auto ]; then for f in sal2 none; do python tools/probes/rows_kernel_time.py $f 1077 2153 4306 8611 2>&1 | grep "TGP_ROWS4="; done; fi
    unset TGP_ROWS4 TGP_ROWS_RW; done ) > $O/rows_kernel_time.txt; cat $O/rows_kernel_time.txt
python tools/probes/stamp_bwd.py 2>&1 | grep -v amdgpu > $O/bwd_role_stamps.txt; cat $O/bwd_role_stamps.txt
bash tools/probes/valu_pass.sh > $O/valu_pass.log 2>&1; cp gpurun_out/r04v/pmc_valu_standalone.csv $O/pmc_valu_standalone.csv; cp gpurun_out/r04v/pmc_valu_rows.csv $O/pmc_valu_rows.csv
./tools/probes/potrf_panel_rate > $O/potrf_panel_rate.txt 2>&1; cat $O/potrf_panel_rate.txt
# ---- kernel timelines of one replayed step (tools/probes/timeline.py): start offset, duration, queue, gap per kernel ----
cd /tmp
for w in tgp_airline_mb10k tgp_airline_mb10k_rank8 idtgp_power_sal3 tgp_power_tanh3x2; do
  T=$(fresh tl_$w)
  rocprofv3 --kernel-trace --output-format csv -d $T -- python3 $R/bench.py --workload $w --steps 40 --warmup 10 --repeats 1 --no-cpu-baseline > $D/tl_$w.log 2>&1
  f=$(find $T -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/probes/timeline.py "$f" > $D/timeline_$w.txt 2>&1
  head -3 $D/timeline_$w.txt
done
cd $R
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +30M -delete
