#!/usr/bin/env python3
"""counter / known bytes per kernel of tools/calib/calib_fetch (rocpd databases under the given directory)."""
import glob, os, sqlite3, sys, collections
BYTES = 512 << 20
tab = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*results.db"), recursive=True)):
    c = sqlite3.connect(f)
    for name, ctr, n, v, dur in c.execute("select kernel_name,counter_name,count(*),sum(value),sum(duration) from counters_collection group by kernel_name,counter_name"):
        k = name.replace("void ", "").split("(")[0][:48]
        tab[k][ctr] = v / n
        tab[k]["us"] = dur / n / 1e3
print("%-44s %10s %14s %14s   (bytes moved per launch: %d)" % ("kernel", "us", "FETCH KB/KB", "WRITE KB/KB", BYTES))
for k in sorted(tab):
    t = tab[k]
    print("%-44s %10.1f %14.3f %14.3f" % (k, t.get("us", 0.0), t.get("FETCH_SIZE", float("nan")) * 1024 / BYTES, t.get("WRITE_SIZE", float("nan")) * 1024 / BYTES))
