R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r04v; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r04v/hs -- python3 $R/tools/probes/hbm_standalone.py 3 > $R/gpurun_out/r04v/hs.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r04v/rows -- python3 $R/bench.py --steps 100 --warmup 10 --repeats 1 --no-cpu-baseline --no-graph > $R/gpurun_out/r04v/rows.log 2>&1
cd $R
f1=$(find gpurun_out/r04v/hs -name "*counter_collection.csv" | head -1); f2=$(find gpurun_out/r04v/rows -name "*counter_collection.csv" | head -1)
python tools/probes/pmc_valu_summary.py $f1 "python3 tools/probes/hbm_standalone.py 3" k_ell_flow=64000000 k_flow_eval=64000000 > gpurun_out/r04v/pmc_valu_standalone.csv; cat gpurun_out/r04v/pmc_valu_standalone.csv | cut -c1-200
python tools/probes/pmc_valu_summary.py $f2 "python3 bench.py --steps 100 --warmup 10 --no-graph (tgp_power_tanh3x2)" k_rows=275552 > gpurun_out/r04v/pmc_valu_rows.csv; cat gpurun_out/r04v/pmc_valu_rows.csv | cut -c1-200
tail -3 gpurun_out/r04v/hs.log
find gpurun_out/r04v -name "*counter_collection.csv" -size +20M -delete
