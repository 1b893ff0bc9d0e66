#!/bin/bash
# quick per-kernel stats of one bench workload under graph replay: tools/probes/quick_stats.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/$TAG/prof.$$
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --steps 500 --warmup 50 --repeats 1 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/prof.log 2>&1
f=$(find $D -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/$TAG/kernel_stats.csv
head -12 "$f" | cut -c1-160
