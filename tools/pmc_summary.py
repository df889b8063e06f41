"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs per kernel (KB per launch).
usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json [workload tag: c2 | c4 | c5 | amp16f]]
The json carries the hash of the kernel sources (xpoint_amd.build.source_hash) so that bench.py can tell when the numbers are stale."""
import collections, csv, re, sys

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]

def agg(path, counter):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        d[k][0] += float(r["Counter_Value"]); d[k][1] += 1
    return d

import json, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from xpoint_amd.build import source_hash
f = agg(sys.argv[1], "FETCH_SIZE"); w = agg(sys.argv[2], "WRITE_SIZE")
if len(sys.argv) > 3:   # machine-readable copy for bench.py's roofline.traffic
    js = {k: {"launches": f[k][1], "fetch_kb": f[k][0] / f[k][1], "write_kb": w.get(k, [0.0, 1])[0] / max(w.get(k, [0.0, 1])[1], 1)} for k in f}
    for k in js:
        js[k]["hbm_bytes_per_launch"] = (2 * js[k]["fetch_kb"] + js[k]["write_kb"]) * 1024
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                       "(gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)",
               "source_hash": source_hash(), "workload": (sys.argv[4] if len(sys.argv) > 4 else "c2"), "kernels": js}, open(sys.argv[3], "w"), indent=1)
print("# per-launch averages; FETCH_SIZE/WRITE_SIZE are in KB as rocprofv3 reports them.")
print("# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request for wide coalesced")
print("# reads, so read bytes ~= 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact for 16-B/lane streaming stores.")
print(f"{'kernel':48s} {'launches':>8s} {'FETCH_KB':>12s} {'WRITE_KB':>12s} {'HBM_MB(2F+W)':>14s}")
for k in sorted(f, key=lambda k: -(2 * f[k][0] + w.get(k, [0, 1])[0])):
    fa = f[k][0] / f[k][1]; wa = w.get(k, [0.0, 1])[0] / max(w.get(k, [0.0, 1])[1], 1)
    print(f"{k:48s} {f[k][1]:8d} {fa:12.1f} {wa:12.1f} {(2 * fa + wa) * 1024 / 1e6:14.2f}")
