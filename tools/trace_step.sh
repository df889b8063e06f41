#!/bin/bash
# kernel trace of a short single-stream bench run; prints the launch sequence of the LAST step (name, duration) — what runs between the kernels we wrote
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_step; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-backend --no-overlap --no-h2d > /dev/null 2>&1
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])[:60] for r in rows]
# last occurrence of the stem kernel = start of the last forward
idx = max(i for i, n in enumerate(names) if n.startswith("stem_conv"))
# step = from a few launches before the stem (input copies) to the end
start = idx
while start > 0 and "copyBuffer" in names[start - 1] or "elementwise" in names[start - 1].lower(): start -= 1
tot = 0
for r, n in zip(rows[start:], names[start:]):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if "copy" in n.lower() or "elementwise" in n.lower() or "fill" in n.lower() or d > 60: print(f"{d:9.1f} us  {n}")
print("launches", len(rows) - start, "sum us", round(tot, 1))
PY
