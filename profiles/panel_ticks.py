#!/usr/bin/env python3
"""Sum the per-phase stopwatch lines the tuning build prints (GBA_PANEL_PROF=1) from a bench stderr log."""
import re
import sys

names = ["npanels", "search", "swait", "prefix", "products", "pwait", "emit", "panels", "kernel"]
tot = [0] * 9
pat = re.compile(r"numeric panels (\d+); .*search (\d+) \(\+wait (\d+)\) prefix (\d+) products (\d+) \(\+wait (\d+)\) emit (\d+) panels (\d+) kernel (\d+)")
for line in open(sys.argv[1]):
    m = pat.search(line)
    if m:
        for i in range(9):
            tot[i] += int(m.group(i + 1))
print(dict(zip(names, tot)))
print({n: round(t / max(tot[8], 1), 3) for n, t in zip(names[1:], tot[1:])})
print("us per panel", tot[7] / max(tot[0], 1) / 100)
