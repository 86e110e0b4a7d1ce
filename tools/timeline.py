#!/usr/bin/env python3
"""GPU timeline of one MD step from a rocprofv3 kernel-trace database: union-busy time, idle gaps and what surrounds them."""
import glob, os, sqlite3, sys
d = sys.argv[1]
f = glob.glob(os.path.join(d, "kt", "**", "*results.db"), recursive=True)[0]
c = sqlite3.connect(f)
rows = c.execute("select name,start,end,stream_id from kernels order by start").fetchall() if False else None
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
q = "select name,start,end,%s from kernels order by start" % qcol
rows = c.execute(q).fetchall()
qmap = {}
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:38]
# steps are delimited by k_evb_blend launches
blend = [i for i, r in enumerate(rows) if "k_evb_blend" in r[0]]
a, b = blend[-3], blend[-2]          # one full step between two blends (skip the last, possibly truncated)
if len(sys.argv) > 2 and sys.argv[2] == "rebuild":      # the last complete step that rebuilt the Verlet list
    for k in range(len(blend) - 2, 0, -1):
        if any("k_neighbors" in r[0] for r in rows[blend[k - 1]:blend[k] + 1]):
            a, b = blend[k - 1], blend[k]
            break
if len(sys.argv) > 2 and sys.argv[2] in ("graph", "direct"):      # the last complete step without a rebuild that went out as linear graphs (k_lg_tick inside) / as stream launches
    want = sys.argv[2] == "graph"
    for k in range(len(blend) - 2, 0, -1):
        names = [r[0] for r in rows[blend[k - 1]:blend[k] + 1]]
        if any("k_neighbors" in n for n in names) or len(names) < 20:
            continue
        if any("k_lg_tick" in n for n in names) == want:
            a, b = blend[k - 1], blend[k]
            break
seg = rows[a:b + 1]
t0 = seg[0][1]
busy = 0; cur_end = seg[0][1]; gaps = []
last = seg[0]
for r in seg:
    if r[1] > cur_end:
        gaps.append((r[1] - cur_end, short(last[0]), short(r[0]), (cur_end - t0) / 1e3))
        busy += 0
    if r[2] > cur_end:
        busy += r[2] - max(r[1], cur_end); cur_end = r[2]; last = r
span = seg[-1][2] - seg[0][1]
print("step span %.1f us, union busy %.1f us, idle %.1f us, kernels %d" % (span / 1e3, busy / 1e3, (span - busy) / 1e3, len(seg)))
for g in sorted(gaps, reverse=True)[:14]:
    print("  gap %7.1f us at t=%7.1f  after %-38s before %s" % (g[0] / 1e3, g[3], g[1], g[2]))
print("-- sequence --")
for r in seg:
    print("%8.1f %8.1f  q%d  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, qmap.setdefault(r[3], len(qmap)), short(r[0])))
