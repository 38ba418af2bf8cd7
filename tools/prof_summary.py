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
        durs = defaultdict(list)
        with open(path) as f:
            for r in csv.DictReader(f):
                durs[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print("== kernel trace:", os.path.relpath(path, d))
        print("%-92s %8s %14s %12s %12s %12s %12s %12s" % ("kernel", "calls", "total_us", "avg_us", "median_us", "p90_us", "min_us", "max_us"))
        for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
            v.sort()
            n = len(v)
            med = v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
            p90 = v[min(n - 1, int(0.9 * (n - 1) + 0.5))]
            print("%-92s %8d %14.1f %12.1f %12.1f %12.1f %12.1f %12.1f" % (k, n, sum(v), sum(v) / n, med, p90, v[0], v[-1]))
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
