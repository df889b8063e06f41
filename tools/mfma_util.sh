#!/bin/bash
# MFMA utilisation per kernel of one single-stream bench step: SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles)
cd /tmp; export TMPDIR=/tmp
# XP_MFMA_UTIL_OVERLAP=" ": keep the default multi-stream schedule (config c5: its streaming step needs it); XP_MFMA_UTIL_ARGS: extra bench.py arguments (e.g. "--precision-class amp16f"); XP_MFMA_UTIL_TAG: suffix of the output files
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAGSFX=${XP_MFMA_UTIL_TAG:+_$XP_MFMA_UTIL_TAG}; OUT=$R/gpurun_out/mfma_util$TAGSFX; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-backend ${XP_MFMA_UTIL_OVERLAP:---no-overlap} --no-h2d $XP_MFMA_UTIL_ARGS > $OUT/p1.log 2>&1 || echo "pass failed: $(tail -2 $OUT/p1.log)"
python3 - "$OUT" "$TAGSFX" "$R" > $R/gpurun_out/mfma_utilisation$TAGSFX.txt <<'PY'
import collections, csv, glob, os, re, sys
d = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]); k = re.match(r"([A-Za-z0-9_:]+(<[^>]*>)?)", k).group(1)
        e = d[k][r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1] += 1
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE on bench.py --no-overlap (per-launch averages)")
print("# utilisation = MFMA busy cycles (summed over the 1024 SIMDs) / (1024 x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs")
print(f"{'kernel':46s} {'launches':>8s} {'cycles':>10s} {'MFMA instr':>12s} {'MFMA busy':>14s} {'utilisation':>12s}")
rows = []
for k, c in d.items():
    if "SQ_INSTS_MFMA" not in c or c["SQ_INSTS_MFMA"][0] == 0: continue
    n = c["GRBM_GUI_ACTIVE"][1]; cyc = c["GRBM_GUI_ACTIVE"][0] / n / 8.0
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / n; ins = c["SQ_INSTS_MFMA"][0] / n
    rows.append((busy * n, k, n, cyc, ins, busy, busy / (1024.0 * cyc)))
for _, k, n, cyc, ins, busy, u in sorted(rows, reverse=True):
    print(f"{k:46s} {n:8d} {cyc:10.0f} {ins:12.0f} {busy:14.0f} {u:12.3f}")
import json
sys.path.insert(0, sys.argv[3])
from xpoint_amd.build import source_hash
json.dump({"source_hash": source_hash(), "workload": (sys.argv[2].lstrip("_") or "c2"),
           "note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on bench.py --no-overlap; mfma_busy_frac = busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), per-launch average",
           "kernels": {k: {"launches": n, "kernel_cycles": cyc, "mfma_instr": ins, "mfma_busy_frac": u} for _, k, n, cyc, ins, busy, u in rows}},
          open(os.path.join(os.path.dirname(sys.argv[1]), "pmc_mfma" + sys.argv[2] + ".json"), "w"), indent=1)
PY
cat $R/gpurun_out/mfma_utilisation$TAGSFX.txt
