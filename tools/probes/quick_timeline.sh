#!/bin/bash
# kernel timeline of one replayed step of a bench workload: tools/probes/quick_timeline.sh <tag> <steps> [bench args]
TAG=$1; STEPS=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/$TAG/tl.$$
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --steps $STEPS --warmup 3 --repeats 1 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/tl.log 2>&1
f=$(find $D -name "*kernel_trace.csv" | head -1)
python3 $R/tools/probes/timeline.py "$f" > $R/gpurun_out/$TAG/timeline.txt 2>&1
cp "$(find $D -name "*kernel_stats.csv" | head -1)" $R/gpurun_out/$TAG/kernel_stats.csv
find $D -name "*kernel_trace.csv" -size +8M -delete
head -90 $R/gpurun_out/$TAG/timeline.txt
