#!/bin/bash
# same-box A/B of this tree against the round-5 tree (a git worktree of ef38016 under ab_r05/, built there): bash tools/probes/ab_r05.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', '$2', round(r['ms_per_step'],4), 'ms/step', round(r['value'],1), 'steps/s')"; }
for w in tgp_airline_tanh5x6:10:2 tgp_airline_mb10k:100:10 tgp_airline_mb10k_rank8:100:10 tgp_power_tanh3x2:2000:100; do
  n=${w%%:*}; rest=${w#*:}; st=${rest%%:*}; wu=${rest#*:}
  for rep in 1 2; do
    (cd $R/ab_r05 && python bench.py --workload $n --steps $st --warmup $wu --no-cpu-baseline 2>/dev/null | one r05 $n)
    (cd $R && python bench.py --workload $n --steps $st --warmup $wu --no-cpu-baseline 2>/dev/null | one r06 $n)
  done
done
