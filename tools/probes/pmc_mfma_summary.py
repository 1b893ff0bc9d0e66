"""Per-kernel MFMA busy fraction from one rocprofv3 --pmc pass (counter_collection.csv):
   python tools/probes/pmc_mfma_summary.py <counter_collection.csv> "<command line the pass profiled>" > out.csv
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the share of the chip's matrix pipes
that were busy while the kernel ran (a kernel on few CUs is low by construction: see active_simd_equiv)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "tgp::" in r["Kernel_Name"]:
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES -- %s" % sys.argv[2])
print("# mfma_busy_frac = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs); busy_simd_equiv = MFMA_BUSY / (GUI_ACTIVE/8) = matrix pipes busy on average")
print("kernel,dispatches,mean_SQ_VALU_MFMA_BUSY_CYCLES,mean_GRBM_GUI_ACTIVE,mean_MFMA_MOPS_F64,mfma_busy_frac,busy_simd_equiv")
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    n = len(c.get("GRBM_GUI_ACTIVE", []))
    if not n:
        continue
    m = lambda name: sum(c.get(name, [0])) / max(len(c.get(name, [0])), 1)
    busy, gui = m("SQ_VALU_MFMA_BUSY_CYCLES"), m("GRBM_GUI_ACTIVE")
    per = gui / 8 if gui else 0
    print('"%s",%d,%.0f,%.0f,%.0f,%.3f,%.1f' % (k[:110], n, busy, gui, m("SQ_INSTS_VALU_MFMA_MOPS_F64"), busy / (per * 1024) if per else 0,
                                             busy / per if per else 0))
