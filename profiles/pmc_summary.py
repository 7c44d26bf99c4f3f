#!/usr/bin/env python3
"""Sum the counters of profiles/pmc_passes.sh per kernel:  pmc_summary.py <dir> [kernel substring ...]
Prints CSV: kernel, launches, then one column per counter (sum over launches).  FETCH_SIZE / WRITE_SIZE are KiB;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so the fetched bytes are 2 * 1024 * FETCH_SIZE."""
import collections
import csv
import glob
import os
import sys

root, pats = sys.argv[1], sys.argv[2:]
vals = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if pats and not any(p in k for p in pats):
            continue
        vals[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
names = sorted({c for k in vals for c in vals[k]})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches"] + names)
for k in sorted(vals, key=lambda k: -vals[k].get("SQ_WAVE_CYCLES", 0)):
    n = max(len(launches[(k, c)]) for c in vals[k])
    w.writerow([k[:100], n] + [("%.6g" % vals[k][c]) if c in vals[k] else "" for c in names])
