#!/bin/bash
# rocprofv3 --kernel-trace --stats of one single-stream bench run; prints the kernels whose names match $1 (regex).   usage: bash tools/kstats.sh "match_|nms_"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-backend --no-overlap --no-h2d ${KSTATS_ARGS} > /tmp/kst_bench.log 2>&1 || { echo "bench.py failed under rocprofv3:"; tail -5 /tmp/kst_bench.log; exit 1; }
python3 - "$1" <<'PY'
import csv, glob, re, sys
f = glob.glob("/tmp/kst/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[1], r["Name"]): print(f'{r["Name"][:72]:72s} x{r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
