#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output into the small CSV summaries kept in this directory.

  kernel stats : python profiles/summarize_rocpd.py stats  <kt_results.db>            > x_kernel_stats.csv
  PMC counters : python profiles/summarize_rocpd.py pmc    <fetch.db> <write.db> ...   > x_hbm_counters.csv

The stats CSV has the columns of `rocprofv3 --stats` (name, calls, total/avg/min/max ns, %).
The pmc CSV sums each counter per kernel name and adds per-launch averages.  FETCH_SIZE and
WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests of
wide coalesced reads at 64 bytes, so the `corrected_bytes` column doubles it (checked in the same
run on the device-to-device copy of a known size); WRITE_SIZE is taken as is.
"""
import csv
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([name, calls, int(tot), round(avg, 1), int(mn), int(mx), round(100.0 * tot / total, 3)])


def pmc(paths):
    w = csv.writer(sys.stdout)
    w.writerow(["Counter", "Kernel", "Launches", "SumKiB", "AvgKiBPerLaunch", "corrected_bytes_per_launch", "AvgDurationNs"])
    for path in paths:
        db = sqlite3.connect(path)
        rows = db.execute("select counter_name, kernel_name, count(*), sum(value), avg(value), avg(duration) "
                          "from counters_collection group by counter_name, kernel_name order by sum(value) desc").fetchall()
        for cname, kname, n, tot, avg, dur in rows:
            factor = 2.0 if cname == "FETCH_SIZE" else 1.0
            w.writerow([cname, kname, n, round(tot, 1), round(avg, 1), int(avg * 1024 * factor), round(dur, 1)])


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("stats", "pmc"):
        sys.exit(__doc__)
    stats(sys.argv[2]) if sys.argv[1] == "stats" else pmc(sys.argv[2:])
