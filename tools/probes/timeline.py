"""Timeline of ONE steady-state step from a rocprofv3 --kernel-trace CSV: start offset, duration, queue and the gap to
the previous kernel's end on the same queue, plus the busy/idle split of the step (union of kernel intervals).

usage: timeline.py <kernel_trace.csv> [anchor kernel substring = k_big_hdr | k_prep_a]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
    names = [k[2] for k in ks]
    anchor = sys.argv[2] if len(sys.argv) > 2 else ("k_big_hdr" if any("k_big_hdr" in n for n in names) else "k_prep_a")
    starts = [i for i, k in enumerate(ks) if anchor in k[2]]
    if len(starts) < 4:
        print("anchor not found often enough", anchor, len(starts))
        return
    a, b = starts[-3], starts[-2]          # a step well inside the timed region
    t0 = ks[a][0]
    step = ks[a:b]
    print(f"step = {len(step)} kernels, {(ks[b][0] - t0) / 1e3:.1f} us start to next start")
    last_end = {}
    ivs = []
    for s, e, n, q in step:
        short = n.split("(")[0].replace("void ", "").replace("tgp::", "")[:46]
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        ivs.append((s, e))
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{q:>3}  gap {gap:6.1f}  {short}")
    ivs.sort()
    busy, cur_s, cur_e = 0, ivs[0][0], ivs[0][1]
    for s, e in ivs[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"union of kernel intervals {busy / 1e3:.1f} us of {(ks[b][0] - t0) / 1e3:.1f}")


main()
