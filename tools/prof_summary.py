#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace / PMC) into a small per-kernel
summary: python tools/prof_summary.py <dir> > summary.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = name.split("(")[0]
    return name[-90:]


def main(d):
    for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
        with open(path) as f:
            for r in csv.DictReader(f):
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                a = agg[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += dur
                a[2] = min(a[2], dur)
                a[3] = max(a[3], dur)
        print("== kernel trace:", os.path.relpath(path, d))
        print("%-92s %8s %14s %12s %12s %12s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us"))
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("%-92s %8d %14.1f %12.1f %12.1f %12.1f" % (k, a[0], a[1], a[1] / a[0], a[2], a[3]))
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        with open(path) as f:
            for r in csv.DictReader(f):
                a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        print("== counters:", os.path.relpath(path, d))
        for k, cs in sorted(agg.items()):
            for c, a in sorted(cs.items()):
                print("%-92s %-22s dispatches %6d  sum %.6g  per_dispatch %.6g" % (k, c, a[0], a[1], a[1] / a[0]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
