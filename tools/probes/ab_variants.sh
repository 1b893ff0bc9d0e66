#!/bin/bash
# bench lines of the shipped library and of variant builds under tools/probes/variants/<name>/libtgp_hip.so, alternately, on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', round(r['ms_per_step'],5), 'ms/step', round(r['value'],1), 'steps/s  rows', round(r['roofline']['kernel_ms'],5))"; }
for rep in 1 2; do
  python $R/bench.py --steps 2000 --warmup 100 --no-cpu-baseline 2>/dev/null | one shipped
  for d in $R/tools/probes/variants/*/; do n=$(basename $d); python $R/tools/probes/bench_with_lib.py $d/libtgp_hip.so --steps 2000 --warmup 100 --no-cpu-baseline 2>/dev/null | one $n; done
done
