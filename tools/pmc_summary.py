#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, the mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
pat = sys.argv[2] if len(sys.argv) > 2 else "meanfield_step"
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(f"{root}/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
for name in sorted(acc):
    s, n = acc[name]
    print(f"{name:40s} mean/dispatch {s / n:16.1f}   dispatches {n}")
