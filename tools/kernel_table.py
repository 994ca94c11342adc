"""Per-kernel averages of a rocprofv3 --kernel-trace --stats run: python tools/kernel_table.py <dir> [substring ...]"""
import csv
import glob
import os
import sys

root = sys.argv[1]
want = sys.argv[2:]
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("kjarni::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if want and not any(w in n for w in want):
            continue
        print("  %-62s calls %6s avg %10.2f us  total %10.2f ms" % (n[:62], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                     float(r["TotalDurationNs"]) / 1e6))
