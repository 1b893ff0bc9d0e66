"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv (one file per counter pass):
python tools/probes/pmc_summary.py FETCH_SIZE=<csv> WRITE_SIZE=<csv> > profiles/..._pmc_hbm_traffic_per_kernel.csv"""
import csv
import sys
from collections import defaultdict

print("# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2), bench.py --steps 100 --warmup 10 --no-graph, workload tgp_power_tanh3x2")
print("# units: KB as reported; HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section)")
print("counter,kernel,dispatches,mean_per_dispatch_KB")
for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == name and "tgp::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print('%s,"%s",%d,%.1f' % (name, k, len(v), sum(v) / len(v)))
