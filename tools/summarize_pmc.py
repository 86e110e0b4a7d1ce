#!/usr/bin/env python3
"""Per-kernel averages of every counter found in the rocpd databases under a tools/pmc_gpu.sh output directory."""
import glob, os, sqlite3, sys, collections
d = sys.argv[1]
only = sys.argv[2:] or None
tab = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)):
    c = sqlite3.connect(f)
    # PMC_MAXDUR_US: only dispatches shorter than this (the pair kernel's launches that rewrite their list are a population of their own)
    lim = float(os.environ.get("PMC_MAXDUR_US", "0")) * 1e3
    where = ("where duration < %f " % lim) if lim > 0 else ""
    for name, ctr, n, v, dur in c.execute("select kernel_name,counter_name,count(*),sum(value),sum(duration) from counters_collection " + where + "group by kernel_name,counter_name"):
        k = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        tab[k][ctr] = v / n
        tab[k]["_us"] = dur / n / 1e3
        tab[k]["_n"] = n
ks = sorted(tab, key=lambda k: -tab[k]["_us"] * tab[k]["_n"])[:14]
ctrs = sorted({c for k in ks for c in tab[k] if not c.startswith("_")})
for k in ks:
    if only and not any(o in k for o in only):
        continue
    print("%-40s n=%d avg_us=%.1f" % (k, tab[k]["_n"], tab[k]["_us"]))
    for c in ctrs:
        if c in tab[k]:
            print("    %-40s %16.1f" % (c, tab[k][c]))
