#!/usr/bin/env python3
"""Per-kernel launch count and average / minimum duration from a rocprofv3 --kernel-trace directory:
   kernel_times.py <dir> [substring ...]"""
import collections
import csv
import glob
import os
import sys

root, pats = sys.argv[1], sys.argv[2:]
d = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if pats and not any(p in k for p in pats):
            continue
        d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':88s} {'launches':>8s} {'avg_us':>10s} {'min_us':>10s} {'total_ms':>10s}")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:88]:88s} {len(v):8d} {sum(v) / len(v):10.1f} {min(v):10.1f} {sum(v) / 1e3:10.2f}")
