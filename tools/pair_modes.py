#!/usr/bin/env python3
"""Durations of the pair kernel's launches in a rocprofv3 kernel-trace database, split into the launches that only read their pair
list and those that rewrite it first (the two populations differ by tens of microseconds):  python tools/pair_modes.py OUTDIR"""
import glob, os, sqlite3, statistics, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "kt", "**", "*results.db"), recursive=True)[0]
c = sqlite3.connect(f)
rows = c.execute("select name,start,end from kernels order by start").fetchall()
for key in ("k_pair_flat", "k_pair_half", "k_neighbors_tiles"):
    v = sorted((r[2] - r[1]) / 1e3 for r in rows if key in r[0])
    if not v:
        continue
    # split at the largest gap between consecutive sorted durations in the middle 80 %
    lo, hi = v, []
    if len(v) > 8:
        k0, k1 = len(v) // 10, len(v) - len(v) // 10
        k = max(range(max(k0, 1), k1), key=lambda i: v[i] - v[i - 1])
        if v[k] - v[k - 1] > 0.08 * v[k]:
            lo, hi = v[:k], v[k:]
    print("%-20s launches %4d  min %.1f  median %.1f  mean %.1f  max %.1f us" % (key, len(v), v[0], statistics.median(v), statistics.mean(v), v[-1]))
    if hi:
        print("   fast population: %4d launches, median %.1f us, mean %.1f;  slow population: %4d launches, median %.1f us, mean %.1f" %
              (len(lo), statistics.median(lo), statistics.mean(lo), len(hi), statistics.median(hi), statistics.mean(hi)))
