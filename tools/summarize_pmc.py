#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc passes (one directory per pass, each holding *_counter_collection.csv) to
profiles/rNN_pmc_summary.csv: per kernel and counter, the per-dispatch mean / min / max.

    python tools/summarize_pmc.py gpurun_out/prof_r01/pmc_* > profiles/r01_pmc_summary.csv

bench.py launches the dim-128 forward (k_forward<32, 4, 0, 4>) in two equal groups - uniform ids,
then Zipf(1.05) item ids - so that kernel's dispatches are reported as two rows, split at half in
dispatch order ("[uniform]" / "[zipf]"), next to the combined one."""
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
        if SPLIT in k and len(seq) >= 2 and len(seq) % 2 == 0:
            emit(k + " [uniform]", c, seq[:len(seq) // 2])
            emit(k + " [zipf]", c, seq[len(seq) // 2:])


if __name__ == "__main__":
    main(sys.argv[1:])
