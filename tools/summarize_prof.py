#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory (rocprofv3 rocpd sqlite databases): per-kernel calls / average
duration from the kernel trace, and per-launch FETCH_SIZE / WRITE_SIZE from the two separate PMC passes.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; the gfx950 correction of MI355X_MICROARCH.md (wide
coalesced reads are tallied at half their bytes) is NOT applied here, both raw and x2 figures are printed."""
import glob, os, sqlite3, sys, collections

d = sys.argv[1]

def db(sub):
    r = glob.glob(os.path.join(d, sub, "**", "*results.db"), recursive=True)
    return sqlite3.connect(r[0]) if r else None

def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]

c = db("kt")
if c:
    print("== kernel trace (rocprofv3 --kernel-trace --stats) ==")
    print("%-72s %7s %12s %7s" % ("kernel", "calls", "avg_us", "pct"))
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-72s %7d %12.3f %7.2f" % (short(name), calls, avg, pct))
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    c = db(sub)
    if not c:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for name, val, dur in c.execute("select kernel_name,value,duration from counters_collection where counter_name=?", (ctr,)):
        a = agg[short(name)]
        a[0] += 1; a[1] += val; a[2] += dur
    print("== %s per launch (separate --pmc pass) ==" % ctr)
    print("%-72s %7s %14s %14s %10s" % ("kernel", "calls", "KiB_raw", "MB_raw", "avg_us"))
    for k, (n, v, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print("%-72s %7d %14.1f %14.3f %10.2f" % (k, n, v / n, v / n * 1024 / 1e6, t / n / 1e3))
