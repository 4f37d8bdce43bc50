#!/usr/bin/env python3
"""Timeline of the big-table step from a rocprofv3 --kernel-trace csv: per step, start / end of every kernel relative to the
item-side kernel's start (which kernels run beside which, where the main stream idles).
    python tools/trace_c3.py <dir with *_kernel_trace.csv> [first_step] [nsteps]"""
import csv
import glob
import sys

d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)


def short(nm):
    nm = nm.replace("void tfr::", "").replace("tfr::", "")
    return nm[:nm.index("(")] if "(" in nm else nm


items = [e for e in ev if "k_seg_reduce" in e[2] and ", true, true, " in e[2]]
print("item-side launches:", len(items))
per = [(items[k + 1][0] - items[k][0]) / 1e3 for k in range(len(items) - 1)]
print("step period us (item start to item start): median %.1f min %.1f max %.1f" % (sorted(per)[len(per) // 2], min(per), max(per)))
for k in range(first, min(first + n, len(items) - 1)):
    t0, t1 = items[k][0], items[k + 1][0]
    print("---- step %d (period %.1f us)" % (k, (t1 - t0) / 1e3))
    for s, e, nm, q in ev:
        if e > t0 and s < t1:
            print("  %8.1f .. %8.1f  (%6.1f)  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short(nm)[:60]))
