"""Per-kernel VALU issue fraction from one rocprofv3 --pmc pass (counter_collection.csv):
   python tools/probes/pmc_valu_summary.py <counter_collection.csv> "<command>" [kernel=units ...] > out.csv
The roof of a flow kernel is f64 VALU issue, not HBM: one SIMD issues one 64-lane f64 VALU instruction per 4 cycles
(78.6 TFLOP/s = 1024 SIMDs x 64 lanes x 2 flop / 4 cycles x 2.4 GHz), so
   valu_issue_frac = SQ_INSTS_VALU * 4 / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
is the share of the chip's f64 vector issue slots the kernel's VALU instructions took if every one of them cost an f64 slot
(an upper bound: 32-bit integer / move instructions cost half), and SQ_ACTIVE_INST_VALU (quad-cycles, MI355X_MICROARCH.md) * 4
the cycles the vector pipes were actually busy.  `kernel=units` (e.g. k_ell_flow=64000000) adds instructions per unit."""
import csv
import sys
from collections import defaultdict

units = dict(a.split("=") for a in sys.argv[3:])
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "tgp::" in r["Kernel_Name"]:
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- %s" % sys.argv[2])
print("# valu_issue_frac = SQ_INSTS_VALU * 4 / (GUI_ACTIVE/8 * 1024 SIMDs)   valu_busy_frac = SQ_ACTIVE_INST_VALU * 4 / (GUI_ACTIVE/8 * 1024)")
print("kernel,dispatches,mean_SQ_INSTS_VALU,mean_SQ_ACTIVE_INST_VALU,mean_GRBM_GUI_ACTIVE,valu_issue_frac,valu_busy_frac,insts_per_unit")
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    n = len(c.get("GRBM_GUI_ACTIVE", []))
    if not n:
        continue
    m = lambda name: sum(c.get(name, [0])) / max(len(c.get(name, [0])), 1)
    ins, act, gui = m("SQ_INSTS_VALU"), m("SQ_ACTIVE_INST_VALU"), m("GRBM_GUI_ACTIVE")
    per = gui / 8 if gui else 0
    u = [float(v) for kk, v in units.items() if kk in k]
    print('"%s",%d,%.0f,%.0f,%.0f,%.3f,%.3f,%s' % (k[:110], n, ins, act, gui, ins * 4 / (per * 1024) if per else 0,
                                                   act * 4 / (per * 1024) if per else 0, "%.1f" % (ins * 64 / u[0]) if u else ""))
