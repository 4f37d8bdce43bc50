#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc passes (one directory per pass, each holding *_counter_collection.csv) to
profiles/rNN_pmc_summary.csv: per kernel and counter, the per-dispatch mean / min / max.

    python tools/summarize_pmc.py gpurun_out/prof_r01/pmc_* > profiles/r01_pmc_summary.csv

bench.py launches the dim-128 forward (k_forward<32, 4, 0, 4>) in three groups - 110 launches with
uniform ids, 110 with Zipf(1.05) item ids, 34 on eight batches at once - so that kernel's dispatches
are also reported per group, in dispatch order ("[uniform]" / "[zipf]" / "[8x_batch]"), next to the
combined row.  Older runs without the third group (220 dispatches) split in two."""
import csv
import glob
import re
import sys
from collections import defaultdict

SPLIT = "k_forward<32, 4, 0, 4>"


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main(dirs):
    rows = defaultdict(list)                        # (kernel, counter) -> [(dispatch id, value)]
    for d in dirs:
        for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                rows[(short(r["Kernel_Name"]), r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "counter", "dispatches", "mean", "min", "max"])

    def emit(k, c, vals):
        out.writerow([k, c, len(vals), "%.6g" % (sum(vals) / len(vals)), "%.6g" % min(vals), "%.6g" % max(vals)])
    for (k, c) in sorted(rows):
        seq = [v for _, v in sorted(rows[(k, c)])]
        emit(k, c, seq)
        if SPLIT in k:
            groups = {254: (("uniform", 110), ("zipf", 110), ("8x_batch", 34)), 220: (("uniform", 110), ("zipf", 110))}.get(len(seq), ())
            lo = 0
            for name, cnt in groups:
                emit(k + " [%s]" % name, c, seq[lo:lo + cnt])
                lo += cnt


if __name__ == "__main__":
    main(sys.argv[1:])
