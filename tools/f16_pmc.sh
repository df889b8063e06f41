#!/bin/bash
# Matrix-pipe utilisation of the fp16-storage GEMM per SHAPE (the bench-step average of tools/mfma_util.sh mixes HBM-bound K = 96 layers with the K >= 768
# ones): rocprofv3 --pmc on tools/gemm_bench.py (GB_F16=1) for the shapes in $1 (indices of its table; default: the deep-stage layers).
#   usage (GPU box, repo root): bash tools/f16_pmc.sh "9,12,13,14,15"  ->  gpurun_out/gemm_f16_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/f16_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export GB_F16=1 GB_ONLY=${1:-9,12,13,14,15}
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/gemm_bench.py > $OUT/p1.log 2>&1 || echo "pass failed: $(tail -2 $OUT/p1.log)"
python3 - "$OUT" > $R/gpurun_out/gemm_f16_pmc.txt <<'PY'
import collections, csv, glob, os, sys
rows = collections.defaultdict(dict)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "gemm_f16" not in r["Kernel_Name"]: continue
        rows[(r["Dispatch_Id"], r["Grid_Size"])][r["Counter_Name"]] = float(r["Counter_Value"])
# group dispatches by their MFMA instruction count (one value per shape)
g = collections.defaultdict(list)
for (d, grid), c in rows.items():
    if "SQ_INSTS_MFMA" in c and "GRBM_GUI_ACTIVE" in c: g[(int(c["SQ_INSTS_MFMA"]), grid)].append(c)
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE on tools/gemm_bench.py GB_F16=1; one line per shape (grouped by MFMA count / grid)")
print("# MFMA-busy fraction = busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); executed rate = MFMAs x 32768 FLOP / kernel time at the 2.4 GHz the counter implies")
print(f"{'MFMA instr':>12s} {'grid (threads)':>16s} {'launches':>9s} {'kernel cycles':>14s} {'MFMA busy frac':>15s} {'FLOP/cycle/CU':>14s}")
for (ins, grid), cs in sorted(g.items()):
    cyc = sum(c["GRBM_GUI_ACTIVE"] for c in cs) / len(cs) / 8.0
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c in cs) / len(cs)
    print(f"{ins:12d} {grid:>16s} {len(cs):9d} {cyc:14.0f} {busy / (1024.0 * cyc):15.3f} {ins * 32768.0 / cyc / 256.0:14.0f}")
PY
cat $R/gpurun_out/gemm_f16_pmc.txt; grep "^M" $OUT/p1.log
