#!/bin/bash
# same-box A/B against the round-5 tree for the workloads that run on k_rows4 / other row-kernel variants
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', '$2', round(r['ms_per_step'],5), 'ms/step', round(r['value'],1), 'steps/s')"; }
for w in svgp_boston:4000:200 svgp_power:2000:100 tgp_power_sal2:2000:100 idtgp_power_sal3:1000:100; do
  n=${w%%:*}; rest=${w#*:}; st=${rest%%:*}; wu=${rest#*:}
  for rep in 1 2; do
    (cd $R/ab_r05 && python bench.py --workload $n --steps $st --warmup $wu --no-cpu-baseline 2>/dev/null | one r05 $n)
    (cd $R && python bench.py --workload $n --steps $st --warmup $wu --no-cpu-baseline 2>/dev/null | one r06 $n)
  done
done
