#!/usr/bin/env python3
"""Copy the reductions tools/profile_round.sh left under gpurun_out/prof_<round>/ into profiles/ (tracked):
    python tools/collect_profiles.py r03
-> profiles/<round>_kernel_stats_<workload>.csv, <round>_pmc_<workload>.csv, <round>_bench_under_rocprof_<workload>.json"""
import glob
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + R)
dst = os.path.join(root, "profiles")
n = 0
for path in sorted(glob.glob(os.path.join(src, "*"))):
    base = os.path.basename(path)
    if base.startswith("kernel_stats_") and base.endswith(".csv"):
        out = "%s_%s" % (R, base)
    elif base.startswith("pmc_") and base.endswith(".csv"):
        out = "%s_%s" % (R, base)
    elif base.startswith("bench_") and base.endswith(".json") and os.path.getsize(path) > 2:
        out = "%s_bench_under_rocprof_%s" % (R, base[len("bench_"):])
    else:
        continue
    shutil.copyfile(path, os.path.join(dst, out))
    n += 1
    print(out)
print(n, "files")
