#!/bin/bash
# same-box A/B of two tgp_model.plan values: bash tools/probes/ab_plan.sh <planA> <planB> workload:steps:warmup ...
A=$1; B=$2; shift; shift
for w in "$@"; do
  n=${w%%:*}; rest=${w#*:}; st=${rest%%:*}; wu=${rest#*:}
  for rep in 1 2 3; do for pl in $A $B; do
    python bench.py --workload $n --steps $st --warmup $wu --no-cpu-baseline --plan $pl 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$n plan $pl', round(r['ms_per_step'],5), 'ms/step', round(r['value'],1), 'rows kernel', round(r['roofline']['kernel_ms'],5))"
  done; done
done
