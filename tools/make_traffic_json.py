#!/usr/bin/env python3
"""traffic.json from a tools/profile_round.sh output directory: per-launch HBM traffic of the kernels bench.py may report as dominant
(FETCH_SIZE and WRITE_SIZE from their own --pmc passes, KiB -> bytes; FETCH_SIZE x 2 on gfx950 as MI355X_MICROARCH.md prescribes - the
counter tallies 128-byte requests at 64 bytes.  tools/calib/ checked that on this library's own access widths: coalesced reads of 4, 8, 16
and 32 bytes per lane all report exactly half their bytes, 32-byte gathers report 64 bytes per record, stores and fp64 atomics report
WRITE_SIZE exactly - profiles/r02_counter_calibration.txt).
'_build' is the sha256 of the sources of the measured kernels (TRAFFIC_SOURCES): bench.py reports traffic = null when its own sources differ.
    python tools/make_traffic_json.py PROF_DIR [OUT.json]"""
import glob, hashlib, json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "traffic.json")
phase_of = {"k_pair_flat": "real space (pair loop)", "k_pair_half": "real space (pair loop)", "k_pair": "real space (pair loop)", "k_pme_spread_tiles": "Q_grid (spread)",
            "k_pme_gather_tiles": "Q grid derivatives (gather)", "k_pme_gather_tiles_wl": "Q grid derivatives (gather)", "k_molecules": "intra-molecular (non-bonded + bonded)"}


TRAFFIC_SOURCES = ("ctx.hpp", "devmath.hpp", "gather_dev.hpp", "molecule.hip", "molecule_dev.hpp", "neighbor.hip", "pair.hip", "pair_flat.hip", "pme.hip")      # same list in bench.py


def build_id():      # the sources of the kernels this file covers (pair loop, PME spread / gather, per-molecule terms) and of the list they read
    h = hashlib.sha256()
    for name in TRAFFIC_SOURCES:
        h.update(open(os.path.join(ROOT, "reactive_pb_nn_md_amd", "csrc", name), "rb").read())
    return h.hexdigest()


out = {}
for sub, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(os.path.join(d, sub, "**", "*results.db"), recursive=True)[0]
    c = sqlite3.connect(f)
    for name, n, v in c.execute("select kernel_name,count(*),sum(value) from counters_collection where counter_name=? group by kernel_name", (key,)):
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        if short in phase_of:
            e = out.setdefault(phase_of[short], {"kernel": short, "source": os.path.basename(d.rstrip("/"))})
            if e["kernel"] != short:
                continue
            e[key.lower() + "_bytes_per_launch"] = e.get(key.lower() + "_bytes_per_launch", 0.0) + v / n * 1024.0
FETCH_CORRECTION = 2.0
for e in out.values():
    e["fetch_size_raw_bytes_per_launch"] = e.get("fetch_size_bytes_per_launch", 0.0)
    e["fetch_size_bytes_per_launch"] = FETCH_CORRECTION * e["fetch_size_raw_bytes_per_launch"]
    e["hbm_bytes_per_launch"] = e["fetch_size_bytes_per_launch"] + e.get("write_size_bytes_per_launch", 0.0)
out["_workload"] = {"n_atoms": 98220, "n_diabats": 8, "pme_grid": 64, "note": "bench.py default workload (configs[2])"}
out["_fetch_correction"] = FETCH_CORRECTION
out["_build"] = build_id()
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
