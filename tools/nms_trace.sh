#!/bin/bash
# per-launch durations of the NMS kernels in one single-stream bench step
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/nmst
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/nmst -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-backend --no-overlap --no-h2d > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$R/gpurun_out/nmst/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'nms_' in r['Kernel_Name']]
rows = rows[-9:]
for r in rows:
    print(r['Kernel_Name'][:40], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'us  grid', r.get('Grid_Size_X', r.get('Grid_Size')))
PY
