#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc passes of ONE workload (one directory per pass, each holding *_counter_collection.csv)
to profiles/rNN_pmc_<workload>.csv: per kernel and counter, the per-dispatch mean / min / max.

    python tools/summarize_pmc.py gpurun_out/prof_r02/pmc_c3_* > profiles/r02_pmc_c3.csv

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a 16-byte-per-lane read,
so read bytes = TCC_EA0_RDREQ x 128 B (MI355X_MICROARCH.md, "HBM")."""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"^tfr::", "", name)


def main(dirs):
    rows = defaultdict(list)                        # (kernel, counter) -> [(dispatch id, value)]
    for d in dirs:
        for path in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                rows[(short(r["Kernel_Name"]), r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = csv.writer(sys.stdout)
    out.writerow(["kernel", "counter", "dispatches", "mean", "min", "max"])
    for (k, c) in sorted(rows):
        seq = [v for _, v in sorted(rows[(k, c)])]
        out.writerow([k, c, len(seq), "%.6g" % (sum(seq) / len(seq)), "%.6g" % min(seq), "%.6g" % max(seq)])


if __name__ == "__main__":
    main(sys.argv[1:])
