#!/bin/bash
# bench lines of the shipped library and of variant builds under tools/probes/variants/<name>/libtgp_hip.so, alternately, on one box:
#   bash tools/probes/ab_variants.sh [workload:steps:warmup ...]        (default: the headline, 2000 steps)
R=${GRAFT_REPO_ROOT:-/root/repo}
[ $# -eq 0 ] && set -- tgp_power_tanh3x2:2000:100
one() { python -c "import sys,json; s=sys.stdin.read(); r=json.loads(s) if s.strip() else None; print('$1', 'no bench line' if r is None else '%.5f ms/step %.1f steps/s  kernel %.5f' % (r['ms_per_step'], r['value'], r['roofline']['kernel_ms']))"; }
for w in "$@"; do
  n=${w%%:*}; rest=${w#*:}; st=${rest%%:*}; wu=${rest#*:}
  for rep in 1 2; do
    python $R/bench.py --workload $n --steps $st --warmup $wu --no-cpu-baseline 2>/dev/null | one "$n shipped"
    for d in $R/tools/probes/variants/*/; do v=$(basename $d); python $R/tools/probes/bench_with_lib.py $d/libtgp_hip.so --workload $n --steps $st --warmup $wu --no-cpu-baseline 2>/dev/null | one "$n $v"; done
  done
done
